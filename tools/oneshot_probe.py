"""Latency of the one-shot all-reduce (csrc/collective.hip) with two processes sharing one GPU -- the only multi-process
configuration a one-GPU box offers: both ranks' kernels run on the same device, so this measures the kernel's own
cost (launch, stores into the peer's inbox, flag wait, reduction), not an xGMI crossing.
    python tools/oneshot_probe.py          -> one JSON line per message size
"""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    import yunet_amd  # noqa: F401
    from yunet_amd.oneshot import OneShotAllReduce
    comm = OneShotAllReduce(dev, 320 * 1024)
    res = {}
    for name, n in (('num_pos (4 B)', 1), ('bucket B (50 KB)', 12800), ('whole gradient (303 KB)', 75861)):
        t = torch.ones(n, device=dev)
        for _ in range(20):
            comm.all_reduce_(t, mean=True)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 300
        e0.record()
        for _ in range(reps):
            comm.all_reduce_(t, mean=True)
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(1000.0 * e0.elapsed_time(e1) / reps, 2)        # microseconds per call, back to back
    out[rank] = dict(us_per_call=res, status=comm.status())
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    out = mp.Manager().dict()
    mp.spawn(worker, args=(2, _free_port(), out), nprocs=2, join=True)
    print(json.dumps({'what': 'one-shot all-reduce, 2 processes on one MI355X, back-to-back calls', 'rank0': out[0], 'rank1': out[1]}))
