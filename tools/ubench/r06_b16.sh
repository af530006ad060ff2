cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
(SHAPES_ALL=1 ONLY=160 REPS=300 timeout 200 tools/ubench/bwd_ab.bin $L tools/ubench/libyunet_b16x.so tools/ubench/libyunet_b16dy.so tools/ubench/libyunet_b16xy.so $L) 2>&1 | grep -v "max|" | grep "16->16" > gpurun_out/r06_b16_nt.log
cat gpurun_out/r06_b16_nt.log
