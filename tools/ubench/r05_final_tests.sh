#!/bin/bash
# Round-5 evidence, part 1 (one gpurun call): the whole GPU suite + bench.py's N = 2 code path with two ranks on ONE card
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export YUNET_PRECISION_JSON=$PWD/gpurun_out/r05_precision.json; rm -f $YUNET_PRECISION_JSON
(timeout 1200 python -m pytest tests -m gpu -q -s --durations=12 > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_pytest_gpu.log)
tail -4 gpurun_out/r05_pytest_gpu.log
L=gpurun_out/r05_two_rank_one_gpu.log
cat > $L <<'TXT'
# bench.py's N = 2 path exercised end to end on ONE MI355X (two ranks share the card over a gloo process group; no second
# GPU on this pool, so this is a functional check of the rank code -- the RCCL window, then the SAME window through the
# one-shot all-reduce with its self-check, time-out cap and watchdog -- not a scaling number).
#   YUNET_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --steps 6 --warmup 3 --batch 64
TXT
YUNET_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 6 --warmup 3 --batch 64 2> gpurun_out/r05_two_rank.err | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('value', b['value'], 'ms_per_step', b['ms_per_step'], 'steps', b['steps'], 'first_window', b['first_window'])
print('dist.backend', b['dist']['backend'], 'comm_ms_per_step', b['dist']['comm_ms_per_step'])
print('dist.oneshot', b['dist'].get('oneshot'))
print('final_loss', b['final_loss'])
" >> $L 2>&1
cat $L; tail -3 gpurun_out/r05_two_rank.err
