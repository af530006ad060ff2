cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
(SHAPES_ALL=1 REPS=300 timeout 200 tools/ubench/bwd_ab.bin $L tools/ubench/libyunet_tlate.so $L) 2>&1 | grep -v "max|" | grep "16->64\|64->16" > gpurun_out/r06_tile_late.log
cat gpurun_out/r06_tile_late.log
