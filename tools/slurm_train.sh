#!/usr/bin/env bash
# Slurm launch of tools/train.py, one task per GPU.  Same command line as the reference's tools/slurm_train.sh
# (positional: partition, job name, config, work dir, then anything for train.py; GPUS, GPUS_PER_NODE, CPUS_PER_TASK
# and SRUN_ARGS from the environment) so existing job scripts keep working.  `--launcher slurm` makes train.py derive
# RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR from SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID / SLURM_NODELIST
# (yunet_amd.parallel.launcher_env).
if [ "$#" -lt 4 ]; then
    echo "usage: [GPUS=8] [GPUS_PER_NODE=8] [CPUS_PER_TASK=5] [SRUN_ARGS=...] $0 PARTITION JOB_NAME CONFIG WORK_DIR [train.py arguments ...]" >&2
    exit 2
fi
here=$(cd "$(dirname "$0")" && pwd)
partition=$1; job=$2; config=$3; workdir=$4
shift 4
ngpu=${GPUS:-8}
per_node=${GPUS_PER_NODE:-8}
[ "$per_node" -gt "$ngpu" ] && per_node=$ngpu
# dmabuf IPC between the ranks of a node: RCCL and the one-shot all-reduce's peer-mapped inboxes need it
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
export PYTHONPATH="$here/..${PYTHONPATH:+:$PYTHONPATH}"
set -x
# shellcheck disable=SC2086   # SRUN_ARGS is a word list by contract
srun --partition="$partition" --job-name="$job" \
     --ntasks="$ngpu" --ntasks-per-node="$per_node" --gres=gpu:"$per_node" \
     --cpus-per-task="${CPUS_PER_TASK:-5}" --kill-on-bad-exit=1 ${SRUN_ARGS:-} \
     python -u "$here/train.py" "$config" --work-dir="$workdir" --launcher=slurm "$@"
