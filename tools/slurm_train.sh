#!/usr/bin/env bash
# The reference's tools/slurm_train.sh: PARTITION JOB_NAME CONFIG WORK_DIR [train.py arguments ...];
# GPUS / GPUS_PER_NODE / CPUS_PER_TASK / SRUN_ARGS from the environment.  One task per GPU;
# train.py --launcher slurm maps SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID (yunet_amd.parallel.launcher_env).
set -x
PARTITION=$1
JOB_NAME=$2
CONFIG=$3
WORK_DIR=$4
GPUS=${GPUS:-8}
GPUS_PER_NODE=${GPUS_PER_NODE:-8}
CPUS_PER_TASK=${CPUS_PER_TASK:-5}
SRUN_ARGS=${SRUN_ARGS:-""}
PY_ARGS=${@:5}
HERE=$(cd "$(dirname "$0")" && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
PYTHONPATH="$HERE/..":$PYTHONPATH \
srun -p ${PARTITION} \
    --job-name=${JOB_NAME} \
    --gres=gpu:${GPUS_PER_NODE} \
    --ntasks=${GPUS} \
    --ntasks-per-node=${GPUS_PER_NODE} \
    --cpus-per-task=${CPUS_PER_TASK} \
    --kill-on-bad-exit=1 \
    ${SRUN_ARGS} \
    python -u "$HERE/train.py" ${CONFIG} --work-dir=${WORK_DIR} --launcher="slurm" ${PY_ARGS}
