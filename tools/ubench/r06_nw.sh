cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
cp $L /tmp/nw8.so; cp $L /tmp/nw4.so; cp $L /tmp/nw8b.so
: > gpurun_out/r06_bwd64_nw.log
for o in 80 40; do
(ONLY=$o REPS=300 timeout 200 tools/ubench/bwd_ab.bin /tmp/nw8.so:YUNET_BWD64_NW=8 /tmp/nw4.so:YUNET_BWD64_NW=4 /tmp/nw8b.so:YUNET_BWD64_NW=8) 2>&1 | grep -v "max|" >> gpurun_out/r06_bwd64_nw.log
done
cat gpurun_out/r06_bwd64_nw.log
