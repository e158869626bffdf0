# the bench job under the reference's Scheduler (overlap loop) with the plug-in, kernel trace: what runs between two graph replays
export TMPDIR=/tmp
REPO=$(pwd)
rm -rf /tmp/prof_refsched
(cd /tmp && SGLANG_USE_AITER=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_refsched -o run -- python $REPO/tests/golden/ref_model.py --run scheduler --dims llama3_8b --job 4,16,896,128,128 --overlap --json /tmp/refsched.json > $REPO/gpurun_out/prof_refsched.log 2>&1)
tail -2 gpurun_out/prof_refsched.log | cut -c1-300
python benchmarks/step_timeline.py /tmp/prof_refsched gpurun_out/r06_reference_scheduler_step_timeline.txt argmax_merge_kernel 100 | head -60
