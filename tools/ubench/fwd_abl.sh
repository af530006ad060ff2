#!/bin/bash
# forward 64->64 unit: ablation sweep + per-phase clocks (product library)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
P=${LIB:-libfacedetection.train_amd/libyunet_hip.so}
L=$OUT/${TAG:-r04_fwd_abl}.log; : > $L
for a in ${ABLS:-0 1 2 4 8 3 6 7 12 14 15}; do
  echo "== ABL=$a" >> $L
  ABL=$a SLOTS=8 FWD=1 REPS=300 timeout 60 tools/ubench/bwd_ab.bin $P 2>&1 | grep -E "80x80|40x40" >> $L
done
[ -n "$NOPROF" ] && { cat $L; exit 0; }
echo "== PROF" >> $L
PROF=1 SLOTS=8 FWD=1 REPS=50 timeout 60 tools/ubench/bwd_ab.bin $P >> $L 2>&1
cat $L
