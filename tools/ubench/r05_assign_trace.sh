#!/bin/bash
# kernel trace of a short bench run: per-kernel statistics of the loss step (gpurun_out/r05_assign_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-1}; do
YUNET_ASSIGN_V2=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_assign_$v -o assign -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --exact-steps --no-cpu-baseline --no-other-configs --no-exact-bwd --no-live-traffic --no-gpu-eager --no-roofline > /tmp/prof_assign_$v.log 2>&1
f=$(find /tmp/prof_assign_$v -name '*kernel_stats.csv' | head -1)
echo "== YUNET_ASSIGN_V2=$v"
[ -n "$f" ] && { grep -v "at::native" $f > $GRAFT_REPO_ROOT/gpurun_out/r05_assign_kernel_stats_v$v.csv; python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'assign_' in n or 'loss_' in n:
        k = (n.split('assign_')[1] if 'assign_' in n else 'loss_' + n.split('loss_')[1]).split('(')[0]
        print(f"{k:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
}
done
