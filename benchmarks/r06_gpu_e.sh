timeout 1200 python -m pytest tests/test_reference_model_gpu.py -x -q -k "sampling" 2>&1 | tail -8
python - <<'P'
import json
d=json.load(open('gpurun_out/reference_model_scheduler_more_sampling-wide-vocab.json'))
print(d['plugin_counts']['sampler'], d['sampling'])
P
cp gpurun_out/reference_model_scheduler_more_sampling-wide-vocab.json gpurun_out/r06_reference_scheduler_sampling_wide_vocab.json
