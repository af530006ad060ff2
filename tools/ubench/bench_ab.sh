#!/bin/bash
# same-box A/B of bench.py under different dispatcher switches: tools/ubench/bench_ab.sh "ENV=V ..." "ENV=V ..." ...
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
L=$OUT/${TAG:-r04_bench_ab}.log; : > $L
SHOW=${SHOW:-"bwd16s stem"}
for rep in 1 2; do
for cfg in "$@"; do
  echo "== [$rep] $cfg" >> $L
  env $cfg timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-exact-bwd --no-live-traffic 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print(b['ms_per_step'], b['value'])
ks=b.get('kernels',{})
for k,v in ks.items():
    if any(w in k for w in '''$SHOW'''.split()): print('   ',k,v)
" >> $L 2>&1
done
done
cat $L
