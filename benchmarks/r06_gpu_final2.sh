# round 6, end-of-round confirmation on a second box: the whole GPU suite, smoke(), the default bench line again
set -u
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_pytest_gpu_tail_final.txt; tail -3 gpurun_out/r06_pytest_gpu_tail_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py > gpurun_out/r06_bench_line_final_b.json 2> gpurun_out/r06_bench_final_b.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench_line_final_b.json').read().strip().splitlines()[-1])
print(round(d['value'],1), d['phase_ms'], round(d['roofline']['frac'],4), d['roofline']['traffic'], round(d['prefill_mfma']['frac'],4), round(d['reference_scheduler'].get('tokens_per_s',0),1), round(d['sampler']['us_per_call'],1))
P
