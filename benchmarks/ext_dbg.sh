for d in 0 1 3 4 8 16 24 31; do echo "dbg=$d"; SGL_AMD_EXTEND_DEBUG=$d python benchmarks/micro.py 2>&1 | grep extend | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['B'], d['prefix'], d['ext'], round(d['us'],1))"; done
