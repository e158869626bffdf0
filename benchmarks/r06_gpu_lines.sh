# round 6: bench lines of the other configurations on the end-of-round tree (one box; not the headline line)
set -u
mkdir -p gpurun_out
F="--no-cpu-baseline --no-reference-scheduler"
run() { name=$1; shift; timeout 900 python bench.py $F "$@" > gpurun_out/r06_bench_line_$name.json 2> gpurun_out/r06_bench_line_$name.err; python - "$name" <<'P'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r06_bench_line_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"]), d["unit"], "decode", round(d["roofline"]["frac"], 3), "step ms", round(d["roofline"]["ms_per_decode_step"], 3), "prefill", round(d["prefill_mfma"]["frac"], 3), d["config"]["parallelism"])
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r06_bench_line_{n}.err").read()[-400:])
P
}
run mixtral_tp1 --model mixtral-8x7b
run fp8kv --kv-cache-dtype fp8_e4m3
run opsurface --operator-surface
run b128 --groups 8
run rankof8 --rank-of 8
run rankof2 --rank-of 2
run 70b_rankof8 --model llama-3-70b --rank-of 8
run 70b_tp1 --model llama-3-70b --steps 2
