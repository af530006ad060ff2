#!/usr/bin/env bash
# One process per GPU over RCCL -- the command line of the reference's tools/dist_train.sh:11-21
# (CONFIG GPUS [PORT] [train.py arguments ...]; NNODES / NODE_RANK / MASTER_ADDR from the environment).
# torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; train.py --launcher pytorch reads them.
#   tools/dist_train.sh configs/yunet_n.py 8
#   YUNET_ONESHOT_AR=1 tools/dist_train.sh configs/yunet_n.py 8 29511 --work-dir work_dirs/n8
set -e
CONFIG=$1
GPUS=$2
PORT=$3
if [ -z "$CONFIG" ] || [ -z "$GPUS" ]; then
    echo "usage: $0 CONFIG GPUS [PORT] [train.py arguments ...]" >&2
    exit 2
fi
case "$PORT" in
    ''|*[!0-9]*) EXTRA=("${@:3}"); PORT=${MASTER_PORT:-29500} ;;     # third word is not a port: a train.py argument
    *) EXTRA=("${@:4}") ;;
esac
NNODES=${NNODES:-1}
NODE_RANK=${NODE_RANK:-0}
MASTER_ADDR=${MASTER_ADDR:-"127.0.0.1"}
HERE=$(cd "$(dirname "$0")" && pwd)
# dmabuf IPC: RCCL (and the one-shot all-reduce's peer-mapped inboxes) between processes
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
PYTHONPATH="$HERE/..":$PYTHONPATH \
exec python -m torch.distributed.run \
    --nnodes=$NNODES \
    --node-rank=$NODE_RANK \
    --master-addr=$MASTER_ADDR \
    --nproc-per-node=$GPUS \
    --master-port=$PORT \
    "$HERE/train.py" \
    "$CONFIG" \
    --seed 0 \
    --launcher pytorch "${EXTRA[@]}"
