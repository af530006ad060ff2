"""CPU, world_size 2, gloo: the N>1 path of the engine -- the flat gradient all-reduce, the
num_pos reduce_mean, the batched logging all-reduce -- against the reference's formulas
(mmdet/core/utils/dist_utils.py:68-74, mmdet/models/detectors/base.py:210-215)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import yunet_oracle as O


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
    import yunet_amd
    import yunet_amd.engine as E
    torch.manual_seed(1234 + rank)
    eng = E.YuNetEngine(O.yunet_arch('s'), 'cpu', world_size=world, process_group=None)
    # (1) gradient mean over ranks: ONE all-reduce of the flat buffer
    g_local = torch.randn(eng.layout.numel)
    eng.params.grad.copy_(g_local)
    eng.params.log_head[:5] = torch.tensor([1.0, 2.0, 3.0, 4.0, 10.0]) * (rank + 1)
    eng.allreduce_grads()            # mean over ranks of [logged scalars | flat gradient], one collective
    gathered = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(gathered, g_local)
    ok_grad = torch.allclose(eng.params.grad, sum(gathered) / world, atol=1e-6) and \
        torch.allclose(eng.params.log_head[:5], torch.tensor([1.0, 2.0, 3.0, 4.0, 10.0]) * 1.5)
    # (2) reduce_mean(num_pos): every rank contributes num_pos/world, SUM
    npos = torch.tensor([17.0, 4.0])[rank]
    norm = torch.tensor([float(npos) / world, 0.0, float(npos), 0.0])
    eng.reduce_num_pos(norm)
    ok_norm = abs(float(norm[0]) - (17.0 + 4.0) / world) < 1e-6 and float(norm[2]) == float(npos)
    # (3) _parse_losses: log_vars are world-averaged, the loss used for backward is local
    cfg = yunet_amd.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'configs', 'yunet_s.py'))
    m = yunet_amd.build_detector(cfg.model)
    losses = dict(loss_cls=torch.tensor(1.0 + rank), loss_bbox=torch.tensor(2.0),
                  loss_obj=torch.tensor(3.0 * (rank + 1)), loss_kps=torch.tensor(0.5))
    loss, lv = m._parse_losses(losses)
    ok_log = abs(float(loss) - sum(float(v) for v in losses.values())) < 1e-6 and \
        abs(lv['loss_cls'] - 1.5) < 1e-6 and abs(lv['loss_obj'] - 4.5) < 1e-6 and \
        abs(lv['loss'] - (6.5 + 10.5) / 2) < 1e-6
    # (4) the wrapper starts every rank from rank 0's weights
    from yunet_amd.parallel import YuNetDistributedDataParallel
    with torch.no_grad():
        for p in m.parameters():
            p.add_(float(rank))
    w = YuNetDistributedDataParallel(m)
    first = next(m.parameters()).detach().clone()
    got = [torch.zeros_like(first) for _ in range(world)]
    dist.all_gather(got, first)
    ok_bcast = torch.equal(got[0], got[1]) and w.module._world == world
    out[rank] = (ok_grad, ok_norm, ok_log, ok_bcast)
    dist.destroy_process_group()


def test_world_size_2_collectives():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        assert all(out[r]), (r, out[r])


def test_ddp_scatter_keeps_padded_gt_lists():
    """MMDistributedDataParallel.scatter stand-in: a GTList / DeviceGT must come out as the same
    type with its padded companion and counts (the engine stages those; the list items of a
    DeviceGT are padded views whose zero rows are NOT faces)."""
    import yunet_amd
    import yunet_amd.synthetic as S
    from yunet_amd.parallel import YuNetDistributedDataParallel
    from yunet_amd.pipelines import DeviceGT
    cfg = yunet_amd.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'configs', 'yunet_s.py'))
    w = YuNetDistributedDataParallel(yunet_amd.build_detector(cfg.model))
    w.device = torch.device('cpu')
    b = S.make_batch(3, 64, 64, 5)
    out = w._to_device(b)
    for k in ('gt_bboxes', 'gt_keypointss'):
        assert type(out[k]) is S.GTList and out[k].padded is not None
        assert torch.equal(out[k].padded, b[k].padded) and torch.equal(out[k].counts, b[k].counts)
    # DeviceGT: items are [gmax, ...] views of the padded tensor
    cnt = torch.tensor([2, 0, 5], dtype=torch.int32)
    pb = torch.zeros(3, 64, 4)
    for i, c in enumerate(cnt.tolist()):
        pb[i, :c] = torch.rand(c, 4) * 50
    dg = DeviceGT([pb[i] for i in range(3)])
    dg.padded, dg.counts = pb, cnt
    got = w._to_device(dict(gt_bboxes=dg))['gt_bboxes']
    assert type(got) is DeviceGT and torch.equal(got.counts, cnt) and got.padded.shape == (3, 64, 4)
    # pad_gt (YuNet_Head.loss API path) honours the counts of such a list
    from yunet_amd.yunet_head import pad_gt
    gk = DeviceGT([torch.zeros(64, 5, 3) for _ in range(3)])
    gk.padded, gk.counts = torch.zeros(3, 64, 5, 3), cnt
    gb2, _, c2 = pad_gt(got, gk, 'cpu')
    assert c2.tolist() == cnt.tolist() and gb2.shape[1] == 5


class _FakeSet:
    """7 images whose pixel (0, 0) carries the image index."""
    def __init__(self, n=7):
        import numpy as np
        self.data_infos = [dict(filename=f'ev/img_{i}.jpg') for i in range(n)]
        self._np = np

    def __len__(self):
        return len(self.data_infos)

    def load_image(self, i):
        img = self._np.zeros((40, 56, 3), dtype=self._np.uint8)
        img[0, 0, 0] = i
        return img


class _FakeDetector(torch.nn.Module):
    def forward(self, return_loss=False, rescale=True, img=None, img_metas=None):
        import numpy as np
        i = int(img[0][0, 0, 0, 0])
        return [[np.array([[i, i, i + 1, i + 1, 0.5]], dtype=np.float32)]]


def _eval_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from yunet_amd.evaluation import multi_gpu_test
    res = multi_gpu_test(_FakeDetector(), _FakeSet(), 'cpu', scale=None)
    out[rank] = None if res is None else [float(r[0][0, 0]) for r in res]
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_restores_dataset_order():
    """DistEvalHook's multi_gpu_test (mmdet/apis/test.py): images interleaved over the ranks, results gathered on
    rank 0 in dataset order."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_eval_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0] and out[1] is None


def _world4_worker(rank, world, port, out):
    """World 4: the two gradient buckets of engine.backward() and the deferred num_pos
    normaliser with RAGGED per-rank positives (one rank has none), on the CPU process group."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import yunet_amd.engine as E
    res = {}
    for kind in ('n', 's'):
        eng = E.YuNetEngine(O.yunet_arch(kind), 'cpu', world_size=world, process_group=None)
        lay = eng.layout
        # the cut of Plan.__init__: bucket A = the tail of the flat buffer from the first pyramid tap on
        split_unit = f"backbone.model{min(eng.arch['out_idx'])}.conv1"
        split_off = lay.units[split_unit]['off']
        cut = E.LOG_HEAD + split_off
        g = torch.Generator().manual_seed(77 + rank)
        local = torch.randn(E.LOG_HEAD + lay.numel, generator=g)
        every = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(every, local)
        want = torch.stack(every).sum(0) / world
        gb = eng.params.grad_buf
        gb.copy_(local)
        eng._allreduce_mean(gb[cut:])            # bucket A (side stream on the GPU)
        eng._allreduce_mean(gb[:cut])            # bucket B + the logged scalars
        split = gb.clone()
        gb.copy_(local)
        eng.allreduce_grads()                    # the unsplit form
        res[kind] = dict(cut=cut, frac_a=float(lay.numel - split_off) / lay.numel,
                         split_ok=bool(torch.allclose(split, want, atol=1e-6)),
                         same_as_unsplit=bool(torch.allclose(split, gb, atol=1e-6)))   # (gloo chunks by message size: not bitwise)
    # deferred normaliser: rank r holds num_pos_r / world; SUM over ranks = reduce_mean(num_pos) (yunet_head.py:493-497)
    for name, npos in (('ragged', [37.0, 0.0, 5.0, 122.0]), ('none', [0.0, 0.0, 0.0, 0.0])):
        norm = torch.tensor([npos[rank] / world, 0.0, npos[rank], 0.0])
        eng.reduce_num_pos(norm)
        n_total = max(float(norm[0]), 1.0)       # loss_finalize_ex: 1 / max(num_total, 1)
        want_total = max(sum(npos) / world, 1.0)
        res[name] = (abs(n_total - want_total) < 1e-6, float(norm[2]) == npos[rank])
    out[rank] = res
    dist.destroy_process_group()


def test_world_size_4_buckets_and_ragged_num_pos():
    world = 4
    out = mp.Manager().dict()
    mp.spawn(_world4_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        for kind in ('n', 's'):
            k = out[r][kind]
            assert k['split_ok'] and k['same_as_unsplit'], (r, kind, k)
            assert 0.5 < k['frac_a'] < 1.0, k          # most parameters ride in the overlapped bucket
        assert all(out[r]['ragged']) and all(out[r]['none']), (r, out[r])
    assert out[0]['n']['cut'] == out[3]['n']['cut']


def _bn_bcast_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import yunet_amd
    from yunet_amd.runner import broadcast_bn_buffers
    cfg = yunet_amd.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'configs', 'yunet_s.py'))
    m = yunet_amd.build_detector(cfg.model)
    with torch.no_grad():
        for n, b in m.named_buffers():
            if n.endswith('running_mean') or n.endswith('running_var'):
                b.fill_(float(rank + 1))
    sent = broadcast_bn_buffers(m)
    vals = sorted({float(b.flatten()[0]) for n, b in m.named_buffers()
                   if n.endswith('running_mean') or n.endswith('running_var')})
    out[rank] = (sent, vals)
    dist.destroy_process_group()


def test_eval_hook_broadcasts_rank0_bn_buffers():
    """DistEvalHook._do_evaluate (mmdet/core/evaluation/eval_hooks.py:101-107): before the sharded test every rank
    takes rank 0's BatchNorm running statistics (per-rank BN otherwise evaluates `world` different models)."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_bn_bcast_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        sent, vals = out[r]
        assert sent > 0 and vals == [1.0], (r, sent, vals)
