#!/bin/bash
# Round-4 first GPU minutes: A/B of the variants queued at the end of round 3.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
P=libfacedetection.train_amd/libyunet_hip.so
cp $P /tmp/libyunet_ws.so
L=$OUT/r04_queued_ab.log; : > $L
echo "== forward: product, ilv, pk, product, ilv, pk" >> $L
SLOTS=8 FWD=1 REPS=500 timeout 120 tools/ubench/bwd_ab.bin $P tools/ubench/libyunet_ilv.so tools/ubench/libyunet_pk.so $P tools/ubench/libyunet_ilv.so tools/ubench/libyunet_pk.so >> $L 2>&1
echo "== forward: product vs WS (YUNET_FWD_WS=1)" >> $L
SLOTS=8 FWD=1 REPS=500 timeout 60 tools/ubench/bwd_ab.bin $P /tmp/libyunet_ws.so:YUNET_FWD_WS=1 $P /tmp/libyunet_ws.so:YUNET_FWD_WS=1 >> $L 2>&1
echo "rc=$?" >> $L
echo "== backward: product, ilv, product, ilv" >> $L
SLOTS=8 REPS=500 timeout 120 tools/ubench/bwd_ab.bin $P tools/ubench/libyunet_ilv.so $P tools/ubench/libyunet_ilv.so 2>&1 | grep -v "max|" >> $L
cat $L
