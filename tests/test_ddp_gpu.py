"""-m gpu: the N>1 training path end to end with two ranks sharing the one GPU of the test box
(gloo carries the collectives here; on a multi-GPU node the same code runs over RCCL).  Checks
that both ranks hold identical averaged gradients / parameters after a step and that the
result equals the hand-computed mean of the two single-rank gradients."""
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


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    import yunet_amd
    import yunet_amd.synthetic as S
    import yunet_oracle as O
    from yunet_amd.optim import FusedSGD
    from yunet_amd.parallel import YuNetDistributedDataParallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yunet_amd.Config.fromfile(os.path.join(root, 'configs', 'yunet_s.py'))
    model = yunet_amd.build_detector(cfg.model)
    model.load_state_dict(O.init_state(O.yunet_arch('s'), seed=7 + rank), strict=True)   # differ on purpose
    model.to(dev).train()
    ddp = YuNetDistributedDataParallel(model, device_ids=[0])          # broadcasts rank 0's weights
    opt = FusedSGD(model, lr=1e-4, momentum=0.9, weight_decay=5e-4)
    batch = S.to_device(S.make_batch(4, 160, 160, S.batch_seed(rank, 0)), dev)
    res = ddp.train_step(batch, opt)
    opt.zero_grad()
    res['loss'].backward()
    g_avg = model.engine.params.grad.detach().clone().cpu()
    # single-rank gradients of both batches with the SAME world-mean num_pos, computed locally
    npos_mean = float(model.engine.plan.norm[0].item())
    g_single = []
    for r in range(world):
        b2 = S.to_device(S.make_batch(4, 160, 160, S.batch_seed(r, 0)), dev)
        eng = model.engine
        eng.world_size = 1
        eng.forward(b2['img'], b2['gt_bboxes'], b2['gt_keypointss'])      # norm[0] = local num_pos
        # re-run the loss phase with the world-mean normaliser, as the 2-rank step used
        eng.plan.norm[0] = npos_mean
        eng._exec(eng.plan.c_fwd_b, 'fwd_b')
        eng.backward()
        g_single.append(eng.params.grad.detach().clone().cpu())
        eng.world_size = world
    opt.step()
    torch.cuda.synchronize()
    out[rank] = dict(g_avg=g_avg, g_mean=(g_single[0] + g_single[1]) / 2,
                     params=model.engine.params.data.detach().cpu(), npos=npos_mean,
                     loss=float(res['log_vars']['loss']))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_average_gradients_and_stay_in_sync():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a['g_avg'], b['g_avg']), 'ranks disagree on the all-reduced gradient'
    assert torch.equal(a['params'], b['params']), 'ranks diverged after the optimizer step'
    assert a['npos'] == b['npos'] and a['loss'] == pytest.approx(b['loss'])
    scale = float(a['g_mean'].abs().max())
    assert float((a['g_avg'] - a['g_mean']).abs().max()) <= 1e-4 * scale
