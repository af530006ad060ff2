cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=libfacedetection.train_amd/libyunet_hip.so
cp $L /tmp/nw8.so; cp $L /tmp/nw4.so
: > gpurun_out/r06_bwd64_nw_small.log
for o in 20 10; do
(ONLY=$o REPS=500 timeout 200 tools/ubench/bwd_ab.bin /tmp/nw8.so:YUNET_BWD64_NW=8 /tmp/nw4.so:YUNET_BWD64_NW=4 $L) 2>&1 | grep -v "max|" >> gpurun_out/r06_bwd64_nw_small.log
done
cat gpurun_out/r06_bwd64_nw_small.log
