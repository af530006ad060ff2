#!/bin/bash
# Round 5, SimOTA assignment rebuilt (loss_step.hip: assign_compact2 / assign_topk2 / assign_resolve2): one gpurun call
#   1. the loss-step tests (reference fixtures, oracle, crowds, edge cases, both launch sets bit-identical) + the one-shot
#      all-reduce tests (last commit of the previous session: peer-access check),
#   2. bench.py alternated twice between YUNET_ASSIGN_V2=0 and =1 (same box, per-op times of the loss step),
#   3. a kernel trace of the new launches (per-kernel breakdown).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 400 python -m pytest tests/test_loss_step_gpu.py tests/test_oneshot_gpu.py -m gpu -q --durations=5 > gpurun_out/r05_assign_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_assign_tests.log)
tail -15 gpurun_out/r05_assign_tests.log
if ! grep -q "pytest rc=0" gpurun_out/r05_assign_tests.log; then echo "TESTS FAILED: A/B skipped"; fi
TAG=r05_assign_ab SHOW="assign loss" bash tools/ubench/bench_ab.sh "YUNET_ASSIGN_V2=0" "YUNET_ASSIGN_V2=1"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_assign -o assign -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --exact-steps --no-cpu-baseline --no-other-configs --no-exact-bwd --no-live-traffic --no-gpu-eager --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_assign.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_assign -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && { cp $f gpurun_out/r05_assign_kernel_stats.csv; grep -i "assign\|loss" $f | cut -c1-60,200- | head; grep -i "assign\|loss_" $f | awk -F'","' '{print $1, $2, $4}' | cut -c1-160; }
find gpurun_out/prof_assign -type f ! -name '*stats.csv' -delete 2>/dev/null
