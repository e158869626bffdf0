for v in 0 1 2; do
  if [ $v = 0 ]; then unset SGL_AMD_TICKET_PROBE; else export SGL_AMD_TICKET_PROBE=$v; fi
  echo "== probe $v"; timeout 60 python benchmarks/r02_exp17_ticketed_qkv.py 2>&1 | grep ticket_8x4
done
