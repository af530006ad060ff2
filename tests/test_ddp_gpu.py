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


def _nccl_world1(port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    import yunet_amd
    import yunet_amd.synthetic as S
    import yunet_oracle as O
    from yunet_amd.optim import FusedSGD
    from yunet_amd.parallel import YuNetDistributedDataParallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yunet_amd.Config.fromfile(os.path.join(root, 'configs', 'yunet_n.py'))
    res = {}
    for mode in ('plain', 'bucketed'):
        model = yunet_amd.build_detector(cfg.model)
        model.load_state_dict(O.init_state(O.yunet_arch('n'), seed=3), strict=True)
        model.to(dev).train()
        ddp = YuNetDistributedDataParallel(model, device_ids=[0])
        opt = FusedSGD(model, lr=1e-3, momentum=0.9, weight_decay=5e-4)
        logs = None
        for it in range(2):
            batch = S.to_device(S.make_batch(8, 160, 160, S.batch_seed(0, it)), dev)
            if mode == 'bucketed':
                eng = model._ensure_engine(dev)
                eng.always_bucket = True          # two backward segments + RCCL (ncclAvg) at world size 1
                eng.world_size = 1
                eng.comm_timing = True            # events around the three collectives (bench.py: N > 1)
            o = ddp.train_step(batch, opt)
            if mode == 'bucketed':
                # world size 1: the step logs from plan.losses; also route them through the gradient
                # buffer head the way world > 1 does
                model._log_pending = None
            opt.zero_grad()
            o['loss'].backward()
            opt.step()
            logs = {k: float(v) for k, v in o['log_vars'].items()}
        torch.cuda.synchronize()
        eng = model.engine
        if mode == 'bucketed':
            res['comm'] = eng.comm_report(2)
        res[mode] = dict(grad=eng.params.grad.detach().cpu().clone(), params=eng.params.data.detach().cpu().clone(),
                         head=eng.params.log_head.detach().cpu().clone(), logs=logs,
                         split=eng.plan.split_off)
    # a bare RCCL collective with the AVG operator on a slice of a larger buffer (what bucket A is)
    buf = torch.arange(64, dtype=torch.float32, device=dev)
    dist.all_reduce(buf[8:], op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    res['avg_identity'] = bool(torch.equal(buf.cpu(), torch.arange(64, dtype=torch.float32)))
    res['backend'] = dist.get_backend()
    out.update(res)
    dist.destroy_process_group()


def test_rccl_backend_bucketed_allreduce_world1():
    """The `nccl` (= RCCL) backend itself, on the one GPU of the test box: process-group init bound
    to the device, ncclAvg on buffer slices, and the two-segment backward with the bucket-A
    collective on the side stream -- bit-identical to the single-segment backward."""
    mgr = mp.Manager()
    out = mgr.dict()
    p = mp.get_context('spawn').Process(target=_nccl_world1, args=(_free_port(), out))
    p.start()
    p.join(600)
    assert p.exitcode == 0, p.exitcode
    assert out['backend'] == 'nccl' and out['avg_identity']
    a, b = out['plain'], out['bucketed']
    assert b['split'] is not None and b['split'] > 0
    assert torch.equal(a['grad'], b['grad']), 'two-segment backward differs from the single list'
    assert torch.equal(a['params'], b['params'])
    for k, v in a['logs'].items():
        assert v == pytest.approx(b['logs'][k], rel=1e-6)
    # loss_finalize mirrors cls, bbox, obj, kps, total into the head of the gradient buffer
    assert b['head'][4] == pytest.approx(b['logs']['loss'], rel=1e-6)
    # comm_report: every collective of the step was timed on its own stream (engine.comm_timing, bench.py N > 1)
    comm = out['comm']
    assert {'num_pos', 'bucket_a', 'bucket_b', 'wait_a', 'exposed_ms_per_step', 'overlapped_ms_per_step'} <= set(comm)
    assert comm['exposed_ms_per_step'] == pytest.approx(comm['num_pos'] + comm['bucket_b'] + comm['wait_a'], abs=1e-3)
    assert 0 < comm['exposed_ms_per_step'] < 50 and comm['bucket_a'] > 0
    print('[rccl world 1] comm ms per step:', comm)


def test_two_gpus_rccl_if_available():
    """Runs only where the box has >= 2 GPUs: bench.py --gpus 2 self-launches over RCCL."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (the round-end scaling run covers N = 2, 4, 8)')
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5',
                        '--warmup', '2', '--batch', '64'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['dist']['backend'] == 'nccl' and len(res['per_rank_images_per_sec']) == 2


def _free_port_cli():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """bench.py's N > 1 code on EVERY 1-GPU box: two ranks share the card over a gloo process group --
    the bucketed window with its per-collective timing, then the SAME window through the one-shot all-reduce (self-check,
    peer time-out cap, watchdog).  A functional check of the rank code the driver's scaling run executes, not a scaling
    number (profiles/r05_two_rank_one_gpu.log was this command run by hand)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YUNET_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port_cli()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6',
           '--warmup', '3', '--batch', '64']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, f'rank 0 must print ONE JSON line, got {len(lines)}'
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['steps'] >= 6 and res['value'] > 0
    assert res['dist']['backend'] == 'gloo' and len(res['per_rank_images_per_sec']) == 2
    comm = res['dist']['comm_ms_per_step']
    assert {'num_pos', 'bucket_a', 'bucket_b', 'wait_a'} <= set(comm), comm
    one = res['dist']['oneshot']
    assert one.get('status') == 0, one
    assert one['value'] > 0 and {'num_pos', 'bucket_a', 'bucket_b', 'wait_a'} <= set(one['comm_ms_per_step'])


def test_dist_train_sh_two_ranks_over_gloo(tmp_path):
    """tools/dist_train.sh (the reference's launcher line, tools/dist_train.sh:11-21) end to end: 2 ranks sharing the card over
    gloo, 3 iterations of YuNet_s through tools/train.py --launcher pytorch; the ranks must end with IDENTICAL parameters
    (the point of the gradient all-reduce)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS='2', YUNET_DUMP_PARAM_SUM=str(tmp_path))
    cmd = [os.path.join(root, 'tools', 'dist_train.sh'), os.path.join(root, 'configs', 'yunet_s.py'), '2',
           str(_free_port_cli()), '--work-dir', str(tmp_path), '--max-iters', '3', '--no-validate', '--cfg-options',
           'dist_params.backend=gloo', 'data.samples_per_gpu=16', 'data.train.iters_per_epoch=3',
           'checkpoint_config.interval=1', 'log_config.interval=1']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    sums = sorted(f for f in os.listdir(tmp_path) if f.startswith('param_sum_rank'))
    assert len(sums) == 2, os.listdir(tmp_path)
    a, b = (open(os.path.join(tmp_path, f)).read() for f in sums)
    assert a == b and float(a.split()[0]) == float(a.split()[0]), (a, b)            # identical on both ranks, finite
    assert os.path.exists(os.path.join(tmp_path, 'yunet_s.py'))                   # rank 0 dumped the config (tools/train.py:171)


def test_ddp_through_device_pipeline_keeps_gt_counts():
    """the DDP scatter used to rebuild GT lists as plain lists, so a DeviceGT's padded
    zero rows became fake faces.  Train one step through pipelines.DevicePipeline behind the DDP
    wrapper and check the engine staged the real per-image counts."""
    import yunet_amd
    import yunet_amd.runner as R
    from yunet_amd.optim import FusedSGD
    from yunet_amd.parallel import YuNetDistributedDataParallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yunet_amd.Config.fromfile(os.path.join(root, 'configs', 'yunet_s.py'))
    dev = torch.device('cuda', 0)
    model = yunet_amd.build_detector(cfg.model).to(dev).train()
    ddp = YuNetDistributedDataParallel(model, device_ids=[0])
    pipe = [dict(type='LoadImageFromFile', to_float32=True),
            dict(type='LoadAnnotations', with_bbox=True, with_keypoints=True),
            dict(type='RandomSquareCrop', crop_choice=[0.5, 0.7, 0.9, 1.1, 1.3, 1.5]),
            dict(type='Resize', img_scale=(160, 160), keep_ratio=False),
            dict(type='RandomFlip', flip_ratio=0.5),
            dict(type='Normalize', mean=[0., 0., 0.], std=[1., 1., 1.], to_rgb=False),
            dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img'])]
    ds = R.SyntheticSourceImages(pipe, samples_per_gpu=8, pool=8)
    batch = ds.batch(0, dev)
    counts = batch['gt_bboxes'].counts.clone()
    assert int(counts.max()) < batch['gt_bboxes'].padded.shape[1]       # there ARE padded rows
    opt = FusedSGD(model, lr=1e-4)
    o = ddp.train_step(batch, opt)
    o['loss'].backward()
    torch.cuda.synchronize()
    assert torch.equal(model.engine.plan.gt_count.cpu(), counts.cpu().int())
    assert int(model.engine.plan.gt_inds.max()) <= int(counts.max())
