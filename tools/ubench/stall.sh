#!/bin/bash
# SQ issue / stall counters of the forward (FWD=1) or backward unit in the torch-free harness.
#   TAG=name [FWD=1] [ONLY=80] [LIB=...] tools/ubench/stall.sh
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/${LIB:-libfacedetection.train_amd/libyunet_hip.so}
TAG=${TAG:-r04_stall}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  SLOTS=8 REPS=3 timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/st_${TAG}_$i -o p -- \
      $GRAFT_REPO_ROOT/tools/ubench/bwd_ab.bin $P > /tmp/st_$i.log 2>&1 || tail -3 /tmp/st_$i.log
  for f in $(find /tmp/st_${TAG}_$i -name "*counter_collection.csv"); do cp $f $OUT/${TAG}_$i.csv; done
done
python $GRAFT_REPO_ROOT/tools/pmc_mean.py $OUT/${TAG}.json $OUT/${TAG}_*.csv
rm -f $OUT/${TAG}_*.csv
