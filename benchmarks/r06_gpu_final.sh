# round 6, final evidence on one box: smoke(), the default bench line, rocprofv3 kernel statistics of the same command
set -u
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r06_smoke.txt
timeout 1200 python bench.py > gpurun_out/r06_bench_line_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r06_bench_line_final.json').read().strip().splitlines()[-1])
print(round(d['value'],1), d['phase_ms'], round(d['roofline']['frac'],4), d['roofline']['traffic'], round(d['prefill_mfma']['frac'],4), round(d['reference_scheduler'].get('tokens_per_s',0),1), round(d['sampler']['us_per_call'],1), d['cpu_baseline'].get('value'))
P
bash benchmarks/gpu_prof.sh --no-reference-scheduler --no-parity > /dev/null 2>&1
cp gpurun_out/kernel_stats.txt gpurun_out/r06_bench_kernel_stats.txt
head -16 gpurun_out/r06_bench_kernel_stats.txt | cut -c1-160
