export TMPDIR=/tmp SGLANG_USE_AITER=0
run() { tag=$1; shift; timeout 300 python tests/golden/ref_model.py --run scheduler --spec-ngram 4 --job 2,2,16,8,12 --json gpurun_out/spec_$tag.json "$@" > gpurun_out/spec_$tag.out 2> gpurun_out/spec_$tag.err; echo "$tag rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/spec_$tag.json')); print('$tag', d['spec']['drafter'], d['oracle'], d['spec']['forward_modes_in_the_timed_job'], d['graph_replays_in_the_timed_job'])" 2>&1 | tail -1; }
run overlap_graph --overlap
run overlap_eager --overlap --server-args '{"disable_cuda_graph": true}'
run normal_graph
run normal_eager --server-args '{"disable_cuda_graph": true}'
