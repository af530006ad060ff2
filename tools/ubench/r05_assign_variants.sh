#!/bin/bash
# per-kernel times of the loss step under variant libraries (tools/ubench/build_loss_variants.sh)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
lib=$GRAFT_REPO_ROOT/tools/ubench/libyunet_$v.so; [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/libfacedetection.train_amd/libyunet_hip.so
YUNET_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_var_$v -o assign -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --exact-steps --no-cpu-baseline --no-other-configs --no-exact-bwd --no-live-traffic --no-gpu-eager --no-roofline > /tmp/prof_var_$v.log 2>&1
f=$(find /tmp/prof_var_$v -name '*kernel_stats.csv' | head -1)
echo "== $v"
[ -n "$f" ] && python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'assign' in n:
        k = n.split('assign_')[1].split('(')[0]
        print(f"{k:30s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
