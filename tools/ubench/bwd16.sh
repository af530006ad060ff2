#!/bin/bash
# backward 16->16 at 160x160: tile kernel (YUNET_BWD16S=0, reads z) vs the wave-streaming kernel that recomputes z
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
P=libfacedetection.train_amd/libyunet_hip.so
L=$OUT/${TAG:-r04_bwd16}.log; : > $L
cp $P /tmp/libyunet_t16.so
cp $P /tmp/libyunet_s16.so
V="/tmp/libyunet_t16.so:YUNET_BWD16S=0 /tmp/libyunet_s16.so:YUNET_BWD16S=1"
for r in $ROWS; do cp $P /tmp/libyunet_r$r.so; V="$V /tmp/libyunet_r$r.so:YUNET_BWD16S_ROWS=$r"; done
ZFWD=1 SHAPES_ALL=1 ONLY=160 SLOTS=8 REPS=${REPS:-200} timeout 120 tools/ubench/bwd_ab.bin $V >> $L 2>&1
echo "rc=$?" >> $L
cat $L
