cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp libfacedetection.train_amd/libyunet_hip.so /tmp/libyunet_fp32.so
(REPS=500 timeout 300 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 tools/ubench/libyunet_old.so libfacedetection.train_amd/libyunet_hip.so) > gpurun_out/r05_bwd64_addr2_ab.log 2>&1
cat gpurun_out/r05_bwd64_addr2_ab.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "dp_bwd or fused_pooling or tap_gradient or upadd" 2>&1 | tail -4
TAG=r05_bench_ab_addr2 SHOW="dp_bwd64 upadd_bwd" tools/ubench/bench_ab.sh "YUNET_HIP_LIB=$PWD/tools/ubench/libyunet_old.so" "YUNET_UPADD_COARSE=1" 
