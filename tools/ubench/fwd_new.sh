#!/bin/bash
# forward 64->64: tile kernel (YUNET_FWD64S=0) vs wave-streaming kernel, optional ROWS sweep
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
P=libfacedetection.train_amd/libyunet_hip.so
cp $P /tmp/libyunet_old.so
L=$OUT/${TAG:-r04_fwd_new}.log; : > $L
V="/tmp/libyunet_old.so:YUNET_FWD64S=0 $P"
for r in $ROWS; do cp $P /tmp/libyunet_r$r.so; V="$V /tmp/libyunet_r$r.so:YUNET_FWD64S_ROWS=$r"; done
SLOTS=8 FWD=1 REPS=${REPS:-300} timeout 120 tools/ubench/bwd_ab.bin $V >> $L 2>&1
echo "rc=$?" >> $L
cat $L
