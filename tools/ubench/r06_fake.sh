cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
cp $L /tmp/a.so; cp $L /tmp/b.so
(ONLY=80 REPS=300 timeout 200 tools/ubench/bwd_ab.bin $L tools/ubench/libyunet_fakeld.so tools/ubench/libyunet_fake2.so /tmp/a.so) 2>&1 | grep -v "max|" > gpurun_out/r06_bwd64_fake_ld.log
echo "== ABL=32 (no issue at all)" >> gpurun_out/r06_bwd64_fake_ld.log
(ABL=32 ONLY=80 REPS=300 timeout 200 tools/ubench/bwd_ab.bin /tmp/b.so /tmp/b.so) 2>&1 | grep -v "max|" >> gpurun_out/r06_bwd64_fake_ld.log
cat gpurun_out/r06_bwd64_fake_ld.log
