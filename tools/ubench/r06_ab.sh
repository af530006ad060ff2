# same-box A/B of the 64->64 backward: torch-free harness (REPS launches per shape), then bench.py alternated
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${TAG:-r06_bwd64_ab}
cp libfacedetection.train_amd/libyunet_hip.so /tmp/libyunet_fp32.so
(REPS=${REPS:-500} timeout 300 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 tools/ubench/libyunet_base.so libfacedetection.train_amd/libyunet_hip.so $EXTRA_LIBS) 2>&1 | grep -v "max|" > gpurun_out/$TAG.log
cat gpurun_out/$TAG.log
