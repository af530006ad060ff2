cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=tools/ubench/libyunet_prof.so
(PROF=1 REPS=200 timeout 120 tools/ubench/bwd_ab.bin $L $L) 2>&1 | grep -v "max|\|yardstick" > gpurun_out/r06_bwd64_prof_all.log
cat gpurun_out/r06_bwd64_prof_all.log
