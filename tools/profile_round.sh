#!/bin/bash
# Collect the evidence a round is judged on (run on the GPU box through gpurun):
#   1. bench.py JSON line (N=1)                         -> gpurun_out/rNN_bench.json
#   2. rocprofv3 --kernel-trace --stats of bench.py      -> gpurun_out/rNN_kernel_stats.csv, rNN_trace_gaps.json
#   3. PMC passes (FETCH_SIZE / WRITE_SIZE, separately)  -> gpurun_out/rNN_pmc_{fetch,write}.csv
#      over tools/kbench.py on the dominant kernels + a calibration copy of known size
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# (the driver's own line: --steps 20 --warmup 5; bench.py extends the timed window to >= 0.5 s itself)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$R -o $R -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --exact-steps --no-cpu-baseline --no-roofline --no-exact-bwd --no-other-configs > /tmp/prof_$R.log 2>&1
# keep the kernels of the step: the at::native rows (thousands of calls) are synthetic.render_faces painting the
# input batches with torch ops BEFORE the warm-up, the __amd_rocclr_copyBuffer rows its small host-to-device
# copies (VERDICT r2 hygiene #10)
for f in $(find /tmp/prof_$R -name "*kernel_stats.csv"); do
  python - "$f" "$OUT/${R}_kernel_stats.csv" <<'PY'
import sys
rows = open(sys.argv[1]).read().splitlines()
keep = [rows[0]] + [r for r in rows[1:] if '_kernel' in r and 'at::native' not in r]
open(sys.argv[2], 'w').write('\n'.join(keep) + '\n')
PY
done
#   2b. idle time between the kernels of a step, from the same trace (tools/trace_gaps.py) -> gpurun_out/rNN_trace_gaps.json
for f in $(find /tmp/prof_$R -name "*kernel_trace.csv" | head -1); do
  python $GRAFT_REPO_ROOT/tools/trace_gaps.py "$f" "$OUT/${R}_trace_gaps.json" > /dev/null 2>&1 || true
done
#      3a. over the bench step itself (per-launch means of every kernel of the step)
#      3b. over tools/kbench.py --calib (a 256 MiB device copy of known size: counter calibration)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${R}_$c -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --exact-steps --no-cpu-baseline --no-roofline --no-exact-bwd --no-other-configs > /tmp/pmc_$c.log 2>&1
  for f in $(find /tmp/pmc_${R}_$c -name "*counter_collection.csv"); do cp $f $OUT/${R}_pmc_$c.csv; done
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcc_${R}_$c -o p -- \
      python $GRAFT_REPO_ROOT/tools/kbench.py --only dp64 --reps 2 --calib > /tmp/pmcc_$c.log 2>&1
  for f in $(find /tmp/pmcc_${R}_$c -name "*counter_collection.csv"); do cp $f $OUT/${R}_pmc_calib_$c.csv; done
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${R}_pmc_FETCH_SIZE.csv $OUT/${R}_pmc_WRITE_SIZE.csv $OUT/${R}_pmc_traffic.json
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${R}_pmc_calib_FETCH_SIZE.csv $OUT/${R}_pmc_calib_WRITE_SIZE.csv $OUT/${R}_pmc_calib.json
# QUICK=1: the bench line, the kernel statistics, the PMC traffic passes and the bf16 line only (steps 4 and 6 -- unit
# utilisation and the torch-free kernel A/B -- keep their files from the previous full run)
#   4. utilisation of the dominant kernels (derived metrics, one per pass): MFMA / VALU / LDS
if [ -z "$QUICK" ]; then
UT=""
for c in MfmaUtil VALUBusy LdsUtil LDSBankConflict; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/util_${R}_$c -o p -- \
      python $GRAFT_REPO_ROOT/tools/kbench.py --only dp64 --reps 2 > /tmp/util_$c.log 2>&1
  for f in $(find /tmp/util_${R}_$c -name "*counter_collection.csv"); do cp $f $OUT/${R}_util_$c.csv; UT="$UT $OUT/${R}_util_$c.csv"; done
done
python $GRAFT_REPO_ROOT/tools/pmc_mean.py $OUT/${R}_util.json $UT
fi
#   5. the second bench line (bf16 activations / bf16 forward matrix instruction, fp32 gradients)
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline --no-gpu-eager 2>/dev/null | tail -1 > $OUT/${R}_bench_bf16.json
[ -n "$QUICK" ] && exit 0
#   6. torch-free kernel A/B at sustained clocks (tools/ubench/bwd_ab), every variant a copy of THIS library (the
#      environment switches are read once per library instance): the exact-fp32 matrix path (= the yardstick),
#      dp_bwd64 with 8 waves / 8 x 16 tiles and with
#      4 waves / 8 x 8 tiles, and the dispatch default; then forward + backward with one and with eight replicas
#      of the BatchNorm sum blocks (SLOTS, YunetBN::slots: the end-of-kernel atomics of the small levels)
for v in fp32 nw8 nw4 tile; do cp libfacedetection.train_amd/libyunet_hip.so /tmp/libyunet_$v.so; done
REPS=1000 timeout 300 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 \
    /tmp/libyunet_nw8.so:YUNET_BWD64_NW=8 /tmp/libyunet_nw4.so:YUNET_BWD64_NW=4 libfacedetection.train_amd/libyunet_hip.so \
    > $OUT/${R}_bwd_ab.log 2>&1
SHAPES_ALL=1 ONLY=160 REPS=300 timeout 200 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 libfacedetection.train_amd/libyunet_hip.so \
    >> $OUT/${R}_bwd_ab.log 2>&1
echo "== forward 64->64: tile kernel (YUNET_FWD64S=0) vs the wave-streaming kernel (default)" >> $OUT/${R}_bwd_ab.log
SLOTS=8 FWD=1 REPS=500 timeout 200 tools/ubench/bwd_ab.bin /tmp/libyunet_tile.so:YUNET_FWD64S=0 libfacedetection.train_amd/libyunet_hip.so \
    >> $OUT/${R}_bwd_ab.log 2>&1
for sl in 8; do
  echo "== SLOTS=$sl (forward, then backward)" >> $OUT/${R}_bwd_ab.log
  SLOTS=$sl FWD=1 REPS=500 timeout 200 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so libfacedetection.train_amd/libyunet_hip.so 2>&1 \
      | grep -v yardstick >> $OUT/${R}_bwd_ab.log
  SLOTS=$sl REPS=500 timeout 200 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 libfacedetection.train_amd/libyunet_hip.so 2>&1 \
      | grep -v yardstick | grep -v "max|" >> $OUT/${R}_bwd_ab.log
done
#   7. per-phase cycle counters of dp_bwd64 (a -DDP_BWD_PROF build made by tools/ubench/build_ab.sh prof "-DDP_BWD_PROF"),
#      whole kernel and with every GEMM / depthwise phase and the global traffic ablated (ABL=63: what staging costs)
if [ -f tools/ubench/libyunet_prof.so ]; then
  for a in 0 15 63; do
    echo "== ABL=$a" >> $OUT/${R}_bwd_phase_cycles.log
    PROF=1 ABL=$a REPS=50 timeout 200 tools/ubench/bwd_ab.bin /tmp/libyunet_fp32.so:YUNET_BWD_FP32MMA=1 tools/ubench/libyunet_prof.so \
        2>&1 | grep -v "total 0" >> $OUT/${R}_bwd_phase_cycles.log
  done
fi
ls -la $OUT | tail -8
