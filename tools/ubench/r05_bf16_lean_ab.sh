#!/bin/bash
# bf16 mode: dp_bwd64 consistent with the bf16 forward (5 matrix products per tile, no low plane of a) vs the full split
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_bf16_lean_ab.log; : > $L
for rep in 1 2; do for v in default lean; do
lib=$GRAFT_REPO_ROOT/tools/ubench/libyunet_$v.so; [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/libfacedetection.train_amd/libyunet_hip.so
echo "== [$rep] $v" >> $L
YUNET_HIP_LIB=$lib timeout 200 python bench.py --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-exact-bwd --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print(b['ms_per_step'], b['value'], 'loss', b['final_loss'])
for k,v in b['kernels'].items():
    if 'bwd64' in k: print('   ',k,v['ms'])
" >> $L 2>&1
done; done; cat $L
