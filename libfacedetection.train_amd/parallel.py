"""Data parallelism: one process per GPU, RCCL over xGMI through torch.distributed.

The reference wraps the detector in mmcv's MMDistributedDataParallel (a torch DDP subclass,
mmdet/apis/train.py:152-163, mmdet/utils/util_distribution.py:5-59; backend 'nccl',
configs/yunet_n.py:18).  Per step that is one ~300 KB gradient bucket plus six scalar
collectives.  Here the gradient already lives in ONE flat buffer, so the
exchange is exactly one all-reduce of 75,856 / 54,608 floats issued on the backward stream
right after the last backward kernel, plus one 4-byte all-reduce for num_pos; the logging
scalars travel as a single 5-float message.  BatchNorm statistics stay per-rank
(no SyncBN; broadcast_buffers=False in the reference).
"""
import os
import re

# dmabuf IPC (RCCL / peer-mapped inboxes across processes): the runtime reads this when it initialises, so it is set when
# this module is imported -- tools/train.py imports it before anything touches the GPU -- not inside init_dist
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402


def first_slurm_host(nodelist):
    """First host of a SLURM node list without scontrol: 'node[01-04,07],gpu3' -> 'node01', 'a,b' -> 'a'.  Nested or
    multi-dimensional bracket forms raise (better than handing torch.distributed an unparsed 'node[01-04]')."""
    m = re.match(r'^([^,\[\]]+)(?:\[([^\]]+)\])?([^,\[\]]*)', nodelist.strip())
    if not m or not m.group(1):
        raise ValueError(f'cannot take the first host of SLURM node list {nodelist!r}; set MASTER_ADDR')
    prefix, ranges, suffix = m.group(1), m.group(2), m.group(3)
    if ranges is None:
        return prefix
    if '[' in suffix or ']' in suffix:
        raise ValueError(f'cannot take the first host of SLURM node list {nodelist!r}; set MASTER_ADDR')
    first = ranges.split(',')[0].split('-')[0]
    if not first.isdigit():
        raise ValueError(f'cannot take the first host of SLURM node list {nodelist!r}; set MASTER_ADDR')
    return prefix + first + suffix


def launcher_env(launcher, env=None, first_host=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT for the three launchers mmcv's init_dist knows
    (mmcv/runner/dist_utils.py, un-vendored): 'pytorch' -- torch.distributed.run / launch export them;
    'slurm' -- SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID, the master is the first host of SLURM_NODELIST (port
    29500 unless MASTER_PORT is set); 'mpi' -- OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK.  Pure function of `env`
    (tests); `first_host` resolves a node list to its first host name (default: `scontrol show hostname`)."""
    env = os.environ if env is None else env
    out = {}
    if launcher == 'slurm':
        out['RANK'] = str(int(env['SLURM_PROCID']))
        out['WORLD_SIZE'] = str(int(env['SLURM_NTASKS']))
        out['LOCAL_RANK'] = str(int(env.get('SLURM_LOCALID', 0)))
        if 'MASTER_ADDR' not in env:
            nodes = env.get('SLURM_NODELIST') or env.get('SLURM_JOB_NODELIST') or '127.0.0.1'
            if first_host is None:
                def first_host(n):
                    import subprocess
                    try:
                        return subprocess.check_output(['scontrol', 'show', 'hostname', n], text=True).split()[0]
                    except Exception:          # noqa: BLE001 -- no scontrol: expand the bracket form here
                        return first_slurm_host(n)
            out['MASTER_ADDR'] = first_host(nodes)
    elif launcher == 'mpi':
        out['RANK'] = str(int(env['OMPI_COMM_WORLD_RANK']))
        out['WORLD_SIZE'] = str(int(env['OMPI_COMM_WORLD_SIZE']))
        out['LOCAL_RANK'] = str(int(env.get('OMPI_COMM_WORLD_LOCAL_RANK', 0)))
        if 'MASTER_ADDR' not in env:
            # mmcv's _init_dist_mpi raises KeyError here too: a multi-node job that fell back to 127.0.0.1 would
            # rendezvous every node with itself
            raise KeyError('MASTER_ADDR: the mpi launcher needs the address of rank 0 in the environment')
    elif launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    if 'MASTER_ADDR' not in env and 'MASTER_ADDR' not in out:
        out['MASTER_ADDR'] = '127.0.0.1'
    if 'MASTER_PORT' not in env:
        out['MASTER_PORT'] = '29500'
    return out


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """tools/train.py:163-170 -> mmcv init_dist.  'nccl' is RCCL on ROCm."""
    if dist.is_initialized():
        return
    os.environ.update(launcher_env(launcher))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', rank))
    if backend == 'nccl' and torch.cuda.is_available():
        dev = torch.device('cuda', local % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(dev)
        kwargs.setdefault('device_id', dev)                      # bind the communicator to this GPU
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class YuNetDistributedDataParallel(nn.Module):
    """Drop-in for MMDistributedDataParallel on this path: `train_step` scatters the batch
    to the local device and calls the module; gradient averaging is done by the module's
    engine (one flat all-reduce) instead of DDP's autograd-hook reducer."""

    def __init__(self, module, device_ids=None, broadcast_buffers=False,
                 find_unused_parameters=False, process_group=None):
        super().__init__()
        self.module = module
        self.device = torch.device('cuda', device_ids[0]) if device_ids else None
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        module.set_data_parallel(world, process_group)
        if world > 1:   # start from identical weights, like DDP's initial broadcast
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def _to_device(self, data):
        """scatter_kwargs of MMDistributedDataParallel for one device.  Per-image GT lists keep
        their type: a `synthetic.GTList` / `pipelines.DeviceGT` carries the same GT padded to
        [N, Gmax, ...] plus the per-image counts, and the engine stages those directly -- rebuilding
        such a list as a plain list would drop them (and, for DeviceGT, whose items are padded
        views, turn every all-zero padding row into a fake face)."""
        if self.device is None:
            return data
        dev = self.device

        def move(t):
            return t if t.device == dev else t.to(dev, non_blocking=True)

        out = {}
        for k, v in data.items():
            if torch.is_tensor(v):
                out[k] = move(v)
            elif isinstance(v, (list, tuple)) and v and torch.is_tensor(v[0]):
                padded, counts = getattr(v, 'padded', None), getattr(v, 'counts', None)
                moved = [move(t) for t in v]
                if padded is not None and counts is not None:
                    moved = type(v)(moved)            # GTList subclass: same type, same extras
                    moved.padded, moved.counts = move(padded), move(counts)
                out[k] = moved
            else:
                out[k] = v
        return out

    def train_step(self, data, optimizer):
        return self.module.train_step(self._to_device(data), optimizer)

    def val_step(self, data, optimizer):
        return self.module.val_step(self._to_device(data), optimizer)


def build_ddp(model, device='cuda', *args, **kwargs):
    """mmdet/utils/util_distribution.py:36-59."""
    return YuNetDistributedDataParallel(model, *args, **kwargs)
