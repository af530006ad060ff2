cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=tools/ubench/libyunet_prof.so
run() { echo "== $*"; (env "$@" PROF=1 ONLY=${ONLY:-80} REPS=100 timeout 120 tools/ubench/bwd_ab.bin $L $L) 2>&1 | grep -v "max|\|yardstick"; }
(run A=0; run ABL=32; run ABL=16; run NOBN=1; run ABL=15; run ABL=47) > gpurun_out/${TAG:-r06_prof}.log 2>&1
cat gpurun_out/${TAG:-r06_prof}.log
