#!/bin/bash
# A/B of two builds of libyunet_hip.so inside one GPU session (same box, alternating runs):
#   tools/ab.sh <base.so> <new.so> [kbench args...]
base=$(realpath "$1"); new=$(realpath "$2"); shift 2
mkdir -p gpurun_out
for round in 1 2 3; do
  for tag in base new; do
    lib=$base; [ $tag = new ] && lib=$new
    echo "== $tag (round $round)"
    YUNET_HIP_LIB=$lib timeout 300 python tools/kbench.py "$@" 2>&1 | grep -E "fwd|bwd"
  done
done
