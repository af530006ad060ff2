cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=tools/ubench/libyunet_prof.so
(PROF=1 SHAPES_ALL=1 REPS=100 timeout 120 tools/ubench/bwd_ab.bin $L $L) 2>&1 | grep -v "max|\|yardstick" | grep -B1 "16->64\|16->16\|64->16" > gpurun_out/r06_1664_prof.log
cat gpurun_out/r06_1664_prof.log
