#!/bin/bash
# forward 64->64 at the 20x20 / 10x10 levels: packed tile kernel (default) vs the wave-streaming kernel (option fwd64s = 2)
# with different band heights
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
P=libfacedetection.train_amd/libyunet_hip.so
L=$OUT/${TAG:-r04_fwd_small}.log; : > $L
cp $P /tmp/libyunet_s2.so
for only in 20 10; do
  for rows in 0 3 5 10 20; do
    echo "== ${only}x${only}, rows $rows (0 = by shape)" >> $L
    ONLY=$only YUNET_FWD64S_ROWS=$rows SLOTS=8 FWD=1 REPS=${REPS:-500} timeout 60 tools/ubench/bwd_ab.bin $P /tmp/libyunet_s2.so:YUNET_FWD64S=2 2>&1 | grep "fwd" >> $L
  done
done
cat $L
