"""-m gpu: the one-shot all-reduce (csrc/collective.hip, yunet_amd/oneshot.py) with TWO processes that share the one
GPU of the test box: each maps the other's inbox through hipIpc*, stores its messages there and sums the slots.
Checks: bit-exact sums against host arithmetic for the step's three message sizes and odd ones, many calls in a row
(both slot parities, ranks running ahead of each other), unaligned buffers, the mean, a message that is too large,
the engine's two-rank step through it (identical to the process-group path: a + b is commutative), and the
time-out path (a rank that never sends)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return torch.device('cuda', 0)


def _inputs(n, rank, it):
    g = torch.Generator().manual_seed(1000 * it + rank)
    return torch.randn(n, generator=g)


def _messages_worker(rank, world, port, out):
    dev = _init(rank, world, port)
    from yunet_amd.oneshot import OneShotAllReduce
    comm = OneShotAllReduce(dev, 320 * 1024)
    ok_verify = comm.verify()
    errs = []
    it = 0
    for n in (1, 3, 5, 1024, 12800, 75861, 80 * 1024):        # num_pos | odd | bucket B | whole gradient | a full slot
        for rep in range(3):
            it += 1
            xs = [_inputs(n, r, it) for r in range(world)]
            want = xs[0].clone()
            for x in xs[1:]:
                want += x
            t = xs[rank].to(dev)
            comm.all_reduce_(t)
            if rank == 1 and rep == 1:
                torch.cuda.synchronize()       # let the ranks drift apart: rank 0 runs ahead into the next call
            if not torch.equal(t.cpu(), want):
                errs.append(('sum', n, rep, float((t.cpu() - want).abs().max())))
    # mean, and a buffer that is not 16-byte aligned
    base = torch.zeros(4099, device=dev)
    xs = [_inputs(4098, r, 777) for r in range(world)]
    t = base[1:]
    t.copy_(xs[rank])
    comm.all_reduce_(t, mean=True)
    want = (xs[0] + xs[1]) * 0.5
    if not torch.equal(t.cpu(), want):
        errs.append(('mean-unaligned', float((t.cpu() - want).abs().max())))
    if float(base[0]) != 0.0:
        errs.append(('wrote outside the buffer',))
    # 200 back-to-back calls without any host synchronisation: both parities, the counter, ranks out of step
    acc = torch.full((257,), float(rank + 1), device=dev)
    for _ in range(200):
        comm.all_reduce_(acc, mean=True)        # mean of (1, 2) = 1.5 on both ranks from the first call on
    torch.cuda.synchronize()
    if not torch.equal(acc.cpu(), torch.full((257,), 1.5)):
        errs.append(('chain', acc[:4].tolist()))
    too_big = False
    try:
        comm.all_reduce_(torch.zeros(comm.max_bytes // 4 + 1, device=dev))
    except ValueError:
        too_big = True
    out[rank] = dict(verify=ok_verify, errs=errs, status=comm.status(), too_big=too_big)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_messages_bit_exact_between_two_processes_on_one_gpu():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_messages_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert out[r]['verify'], 'self-check against the process group failed'
        assert out[r]['errs'] == [], out[r]['errs']
        assert out[r]['status'] == 0 and out[r]['too_big']


def _timeout_worker(rank, world, port, out):
    dev = _init(rank, world, port)
    from yunet_amd.oneshot import OneShotAllReduce
    import yunet_amd._lib as L
    comm = OneShotAllReduce(dev, 4096)
    t = torch.full((16,), float(rank + 1), device=dev)
    prev = L.set_option('oneshot_timeout_ms', 1000)      # the default is minutes
    if rank == 0:
        comm.all_reduce_(t)          # rank 1 never sends: the wait gives up, sets the status word and poisons the buffer
    torch.cuda.synchronize()
    L.set_option('oneshot_timeout_ms', prev)
    out[rank] = dict(status=comm.status(), nan=bool(torch.isnan(t).all()), value=t.cpu().tolist(), default_ms=prev)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_missing_peer_times_out_and_reports():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_timeout_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    # loud on both channels: the host status word names the call, the buffer is NaN (never the un-reduced input)
    assert out[0]['status'] == 1 and out[0]['nan'], out[0]
    assert out[1]['status'] == 0 and out[1]['value'] == [2.0] * 16
    assert out[0]['default_ms'] >= 60000, 'the default time-out is minutes, like the process group\'s'

    
def _engine_timeout_worker(rank, world, port, out):
    """The training path must not continue on local gradients: the engine reads the status word every step."""
    dev = _init(rank, world, port)
    import yunet_amd
    import yunet_amd._lib as L
    import yunet_amd.synthetic as S
    from yunet_amd.optim import FusedSGD
    from yunet_amd.parallel import YuNetDistributedDataParallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yunet_amd.Config.fromfile(os.path.join(root, 'configs', 'yunet_s.py'))
    model = yunet_amd.build_detector(cfg.model).to(dev).train()
    ddp = YuNetDistributedDataParallel(model, device_ids=[0])
    model.bind_engine(dev)
    assert model.engine.enable_oneshot(verify=True)
    opt = FusedSGD(model, lr=1e-4, momentum=0.9, weight_decay=5e-4)
    b = S.to_device(S.make_batch(4, 160, 160, 100 + rank), dev)
    L.set_option('oneshot_timeout_ms', 1000)
    raised, first_loss = None, None
    try:
        for it in range(3):
            if rank == 1 and it >= 1:
                break                # rank 1 stops after the first step: rank 0's second step waits in vain
            o = ddp.train_step(b, opt)
            opt.zero_grad()
            o['loss'].backward()
            opt.step()
            torch.cuda.synchronize()
            if it == 0:
                first_loss = float(o['log_vars']['loss'])
    except RuntimeError as e:
        raised = str(e)
    out[rank] = dict(raised=raised, first_loss=first_loss, status=model.engine.oneshot_status())
    dist.barrier()
    model.engine.disable_oneshot()
    dist.destroy_process_group()


def test_engine_raises_when_a_peer_goes_missing():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_engine_timeout_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0]['first_loss'] == out[0]['first_loss'], 'the first (complete) step is finite'
    assert out[0]['raised'] and 'one-shot all-reduce' in out[0]['raised'], out[0]
    assert out[0]['status'] > 0 and out[1]['raised'] is None


def _engine_worker(rank, world, port, out):
    dev = _init(rank, world, port)
    import yunet_amd
    import yunet_amd.synthetic as S
    import yunet_oracle as O
    from yunet_amd.optim import FusedSGD
    from yunet_amd.parallel import YuNetDistributedDataParallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yunet_amd.Config.fromfile(os.path.join(root, 'configs', 'yunet_s.py'))
    res = {}
    for mode in ('group', 'oneshot'):
        model = yunet_amd.build_detector(cfg.model)
        model.load_state_dict(O.init_state(O.yunet_arch('s'), seed=11), strict=True)
        model.to(dev).train()
        ddp = YuNetDistributedDataParallel(model, device_ids=[0])
        model.bind_engine(dev)             # (normally bound by the first step)
        if mode == 'oneshot':
            assert model.engine.enable_oneshot(verify=True)
        opt = FusedSGD(model, lr=1e-4, momentum=0.9, weight_decay=5e-4)
        losses = []
        for it in range(3):
            batch = S.to_device(S.make_batch(4, 160, 160, S.batch_seed(rank, it)), dev)
            r = ddp.train_step(batch, opt)
            opt.zero_grad()
            r['loss'].backward()
            opt.step()
            losses.append(float(r['log_vars']['loss']))
        torch.cuda.synchronize()
        res[mode] = dict(grad=model.engine.params.grad.detach().clone().cpu(),
                         params=model.engine.params.data.detach().clone().cpu(),
                         npos=float(model.engine.plan.norm[0].item()), losses=losses,
                         status=model.engine.oneshot_status())
        model.engine.disable_oneshot()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_engine_step_through_the_oneshot_allreduce_equals_the_process_group():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_engine_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    for mode in ('group', 'oneshot'):
        assert torch.equal(a[mode]['params'], b[mode]['params']), f'{mode}: ranks diverged'
        assert torch.equal(a[mode]['grad'], b[mode]['grad'])
    assert a['oneshot']['status'] == 0 and b['oneshot']['status'] == 0
    # two addends: the sum does not depend on the order, so the two transports give the same bits
    assert torch.equal(a['group']['grad'], a['oneshot']['grad'])
    assert torch.equal(a['group']['params'], a['oneshot']['params'])
    assert a['group']['npos'] == a['oneshot']['npos'] and a['group']['losses'] == a['oneshot']['losses']
