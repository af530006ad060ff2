cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
run() { echo "== $*"; (env "$@" ONLY=80 REPS=300 timeout 120 tools/ubench/bwd_ab.bin $L $L) 2>&1 | grep -v "max|\|yardstick"; }
(run A=0; run NOBN=1; run BATCH=32; run BATCH=64; run BATCH=128; run BATCH=512; run NOBN=1 ABL=32; run ABL=16; run NOBN=1 ABL=16) > gpurun_out/r06_bw.log 2>&1
cat gpurun_out/r06_bw.log
