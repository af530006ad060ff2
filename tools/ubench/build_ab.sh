#!/bin/bash
# Builds the torch-free backward A/B harness and variant libraries of libyunet_hip.so:
#   tools/ubench/build_ab.sh [NAME "EXTRA_HIPCC_FLAGS"] ...
# -> tools/ubench/bwd_ab.bin, and for every NAME a libyunet_NAME.so next to it whose fp32 conv objects (conv_bwd / conv_fwd / conv_fwd64 / conv_bwd16 / conv_fwd16 / conv_stem) were
#    compiled with the extra flags (the other objects are shared with the product build).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/libfacedetection.train_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value"
make -C $CSRC -j8 >/dev/null
$HIPCC --offload-arch=gfx950 -O2 -Wno-unused-value $ROOT/tools/ubench/bwd_ab.cpp -o $ROOT/tools/ubench/bwd_ab.bin -ldl 2>/dev/null
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  $HIPCC $FLAGS $extra -c $CSRC/conv_bwd.hip -o /tmp/conv_bwd_$name.o &
  $HIPCC $FLAGS $extra -c $CSRC/conv_fwd.hip -o /tmp/conv_fwd_$name.o &
  $HIPCC $FLAGS $extra -c $CSRC/conv_fwd64.hip -o /tmp/conv_fwd64_$name.o &
  $HIPCC $FLAGS $extra -c $CSRC/conv_bwd16.hip -o /tmp/conv_bwd16_$name.o &
  $HIPCC $FLAGS $extra -c $CSRC/conv_fwd16.hip -o /tmp/conv_fwd16_$name.o &
  $HIPCC $FLAGS $extra -c $CSRC/conv_stem.hip -o /tmp/conv_stem_$name.o &
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC $CSRC/loss_step.o /tmp/conv_fwd_$name.o /tmp/conv_fwd64_$name.o /tmp/conv_bwd_$name.o /tmp/conv_bwd16_$name.o \
      $CSRC/conv_fwd_bf16.o $CSRC/conv_bwd_bf16.o $CSRC/conv_fwd16_bf16.o $CSRC/conv_bwd16_bf16.o $CSRC/conv_fwd64_bf16.o $CSRC/conv_stem_bf16.o $CSRC/augment.o $CSRC/detect.o $CSRC/collective.o /tmp/conv_stem_$name.o /tmp/conv_fwd16_$name.o $CSRC/api.o \
      -o $ROOT/tools/ubench/libyunet_$name.so
  echo built libyunet_$name.so
done
