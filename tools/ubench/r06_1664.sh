cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
run() { echo "== $*"; (env "$@" SHAPES_ALL=1 REPS=300 timeout 120 tools/ubench/bwd_ab.bin $L $L) 2>&1 | grep -v "max|\|yardstick" | grep "16->64\|16->16\|64->16"; }
(run A=0; run NOBN=1; run BATCH=32) > gpurun_out/r06_1664.log 2>&1
cat gpurun_out/r06_1664.log
