cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp libfacedetection.train_amd/libyunet_hip.so /tmp/libyunet_fp32.so
(SLOTS=8 REPS=1000 timeout 300 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 tools/ubench/libyunet_notab.so libfacedetection.train_amd/libyunet_hip.so) 2>&1 | grep -v "max|" > gpurun_out/r05_prologue_expt.log
cat gpurun_out/r05_prologue_expt.log
