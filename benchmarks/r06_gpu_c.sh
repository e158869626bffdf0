F="--no-cpu-baseline --no-kernel-roofline --no-parity --no-reference-scheduler"
for i in 1 2; do
for m in 0 1; do
SGLANG_AMD_BENCH_SYNC_PREFILL=$m timeout 600 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sync=$m', round(d['value'],1), {k:round(v,2) for k,v in d['phase_ms'].items()}, round(d['ms_per_decode_step'],4), round(d['prefill_mfma']['frac'],4))
"
done
done
