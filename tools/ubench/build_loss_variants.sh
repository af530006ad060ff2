#!/bin/bash
# variant libraries that differ in loss_step.o only: tools/ubench/build_loss_variants.sh NAME "FLAGS" [NAME "FLAGS" ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/libfacedetection.train_amd/csrc
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -ffp-contract=off"
make -C $CSRC >/dev/null
OBJS=$(cd $CSRC && ls *.o | grep -v '^loss_step.o$' | sed "s#^#$CSRC/#")
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  $HIPCC $FLAGS $extra -c $CSRC/loss_step.hip -o /tmp/loss_step_$name.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC /tmp/loss_step_$name.o $OBJS -o $ROOT/tools/ubench/libyunet_$name.so
  echo built libyunet_$name.so
done
