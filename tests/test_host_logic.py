"""CPU: host-side mirror of the reference interface -- registry/config, module surface,
flat parameter layout, plan construction, LR schedule, synthetic data."""
import math
import os

import pytest
import torch

import yunet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(kind):
    import yunet_amd
    cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', f'yunet_{kind}.py'))
    return yunet_amd.build_detector(cfg.model), cfg


@pytest.mark.parametrize('kind,n_params,n_tensors', [('n', 75856, 154), ('s', 54608, 136)])
def test_model_surface(kind, n_params, n_tensors):
    """Parameter count / names / shapes are the checkpoint contract (README.md:146-147)."""
    m, cfg = build(kind)
    assert sum(p.numel() for p in m.parameters()) == n_params
    assert len(list(m.parameters())) == n_tensors
    ref = O.init_state(O.yunet_arch(kind))
    sd = m.state_dict()
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert m.arch() == {**O.yunet_arch(kind)}
    # reference init: bias 0.02, BN gamma 1 / beta 0 (yunet_backbone.py:21-31)
    assert float(sd['backbone.model0.conv1.bias'][0]) == pytest.approx(0.02)
    assert float(sd['neck.lateral_convs.1.bn.weight'][3]) == 1.0


def test_registry_names_and_errors():
    import yunet_amd
    for name in ('YuNet', 'YuNetBackbone', 'TFPN', 'YuNet_Head', 'CrossEntropyLoss', 'EIoULoss',
                 'DIoULoss', 'SmoothL1Loss'):
        assert yunet_amd.MODELS.get(name) is not None, name
    assert yunet_amd.BBOX_ASSIGNERS.get('SimOTAAssigner') is not None
    assert yunet_amd.PRIOR_GENERATORS.get('MlvlPointGenerator') is not None
    with pytest.raises(KeyError):
        yunet_amd.build_detector(dict(type='NoSuchDetector'))
    with pytest.raises(TypeError):
        yunet_amd.build_from_cfg(dict(foo=1), yunet_amd.MODELS)
    # round 4: SimOTAAssigner takes the reference's constructor arguments (candidate_topk up to 16)
    a = yunet_amd.build_assigner(dict(type='SimOTAAssigner', candidate_topk=5, iou_weight=2.0, cls_weight=0.5))
    assert (a.candidate_topk, a.iou_weight, a.cls_weight, a.center_radius) == (5, 2.0, 0.5, 2.5)
    with pytest.raises(NotImplementedError):
        yunet_amd.build_assigner(dict(type='SimOTAAssigner', candidate_topk=17))
    m, _ = build('s')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.forward_train(torch.zeros(1, 3, 160, 160), [{}], [torch.zeros(1, 4)], [torch.zeros(1)],
                        [torch.zeros(1, 5, 3)])


def test_config_overrides_and_cli_surface():
    import yunet_amd
    cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'yunet_n.py'))
    cfg.merge_from_dict({'optimizer.lr': 0.02, 'data.samples_per_gpu': 8})
    assert cfg.optimizer.lr == 0.02 and cfg.data.samples_per_gpu == 8
    assert cfg.model.bbox_head.loss_bbox.type == 'EIoULoss'
    assert cfg.dist_params.backend == 'nccl'


def test_flat_layout_is_a_bijection():
    import yunet_amd.engine as E
    for kind in ('n', 's'):
        m, _ = build(kind)
        lay = E.ParamLayout(m.arch())
        names = [k for k, _ in m.named_parameters()]
        assert set(lay.entries) == set(names)
        cover = torch.zeros(lay.numel, dtype=torch.int32)
        for k, (off, shape) in lay.entries.items():
            cover[off:off + math.prod(shape)] += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1       # no gap, no overlap
        # fused head rows: cls | bbox | obj | kps
        off0 = lay.units['head.0']['off']
        assert lay.entries['bbox_head.multi_level_cls.0.conv1.weight'][0] == off0
        assert lay.entries['bbox_head.multi_level_bbox.0.conv1.weight'][0] == off0 + 64
        assert lay.entries['bbox_head.multi_level_obj.0.conv1.weight'][0] == off0 + 5 * 64
        assert lay.entries['bbox_head.multi_level_kps.0.conv1.weight'][0] == off0 + 6 * 64


@pytest.mark.parametrize('fuse_taps', [True, False])
def test_plan_builds_on_cpu_and_orders_accumulation(monkeypatch, fuse_taps):
    """Op lists can be built without a GPU (pointers only).  Gradients with two backward writers (the lateral
    outputs: head chain + TFPN merge; unfused form: also the taps, pool + TFPN) must have exactly one overwriting and
    one accumulating writer.  Default form: a tap's gradient has ONE writer -- the pool backward, which takes the
    merge's share as `extra` (p[3]) from a merge backward that ran before it with dxa = NULL."""
    import yunet_amd._lib as L
    import yunet_amd.engine as E
    if not fuse_taps:
        monkeypatch.setenv('YUNET_NO_UPADD_POOL_FUSION', '1')
    for kind, h in (('n', 320), ('s', 160)):
        eng = E.YuNetEngine(O.yunet_arch(kind), 'cpu')
        eng.use_lanes = True            # executor lanes for the head chains (off by default: no measured gain)
        plan = eng.get_plan(2, h, h, 3)
        assert plan.P == sum((h // s) ** 2 for s in (8, 16, 32))
        ops = plan.fwd_a + plan.fwd_b + plan.bwd
        assert all(1 <= op.opcode <= L.OP_JOIN for op in ops)
        # every unit's weight-gradient partials are reduced by the one batched op at the end
        red = [op for op in plan.bwd if op.opcode == L.OP_REDUCE_BATCH]
        assert len(red) == 1 and red[0].i[0] == len(plan.reduce_jobs)
        assert red[0].i[1] == sum((w + 63) // 64 for _, _, _, w, _ in plan.reduce_jobs)
        tab = plan.reduce_table.numpy()
        assert [int(r[2] & 0xffffffff) for r in tab] == [j[2] for j in plan.reduce_jobs]     # rows
        assert [int(r[2] >> 32) for r in tab] == [j[3] for j in plan.reduce_jobs]            # width
        n_dp = sum(1 for op in plan.fwd_a if op.opcode == L.OP_DP_FWD)
        assert n_dp == len(E.ParamLayout.dp_units(eng.arch)) + 3
        assert sum(1 for op in plan.bwd if op.opcode == L.OP_DP_BWD) == n_dp
        writers = {}
        for op in plan.bwd:
            if op.opcode == L.OP_DP_BWD:
                writers.setdefault(op.dp.dx, []).append(op.dp.accumulate_dx)
            elif op.opcode == L.OP_POOL_BWD:
                writers.setdefault(op.p[2], []).append(op.i[4])
            elif op.opcode == L.OP_UPADD_BWD:
                if op.p[3]:
                    writers.setdefault(op.p[3], []).append(op.i[4])
                writers.setdefault(op.p[4], []).append(op.i[5])
        multi = [w for w in writers.values() if len(w) > 1]
        assert len(multi) == (2 if fuse_taps else 4)        # two lateral outputs (+ two backbone taps)
        pools = [op for op in plan.bwd if op.opcode == L.OP_POOL_BWD]
        merges = [op for op in plan.bwd if op.opcode == L.OP_UPADD_BWD]
        assert len(pools) == 2 and len(merges) == 2
        if fuse_taps:
            # each merge hands its gradient (p[2] = dout) to the pool backward of the same tap (p[0] = z), later in the list
            for m in merges:
                assert not m.p[3]
                match = [q for q in pools if q.p[0] == m.p[0]]
                assert len(match) == 1 and match[0].p[3] == m.p[2] and match[0].i[4] == 0
                ops_b = list(plan.bwd)
                assert [o.p[0] == m.p[0] and o.opcode == L.OP_UPADD_BWD for o in ops_b].index(True) < \
                    [o.p[0] == m.p[0] and o.opcode == L.OP_POOL_BWD for o in ops_b].index(True)
        else:
            assert all(not q.p[3] for q in pools) and all(m.p[3] for m in merges)
        for w in writers.values():
            assert w[0] == 0 and all(a == 1 for a in w[1:])
        # executor lanes: the head chains of levels 1 and 2 run on side streams.  Every list forks a lane before
        # its first op on it and joins it before the list ends; in backward a lane is joined BEFORE the
        # upsample-add that accumulates into the gradient that lane's chain wrote first (accumulate flag 1).
        for ops_l in (plan.fwd_a, plan.bwd, plan.fwd_eval):
            open_l = 0
            for op in ops_l:
                if op.opcode == L.OP_FORK:
                    open_l |= op.i[0]
                elif op.opcode == L.OP_JOIN:
                    assert op.i[0] & open_l == op.i[0]
                    open_l &= ~op.i[0]
                else:
                    lane = op.i[L.OP_LANE]
                    assert 0 <= lane <= L.MAX_LANES and (lane == 0 or (open_l >> lane) & 1), 'op on a lane that is not forked'
            assert open_l == 0, 'a list ends with a lane still open'
        assert plan.lanes_used == 0b110
        lane_ops = [op for op in plan.fwd_a if op.i[L.OP_LANE] > 0]
        assert len(lane_ops) == 2 * (2 if kind == 'n' else 1) and all(op.opcode == L.OP_DP_FWD for op in lane_ops)
        wrote = {}                      # gradient buffer -> lane whose backward wrote it first
        open_l = 0
        for op in plan.bwd:
            if op.opcode == L.OP_FORK:
                open_l |= op.i[0]
            elif op.opcode == L.OP_JOIN:
                open_l &= ~op.i[0]
            elif op.opcode == L.OP_DP_BWD and op.i[L.OP_LANE] > 0:
                wrote[op.dp.dx] = op.i[L.OP_LANE]
            elif op.opcode == L.OP_UPADD_BWD:
                for ptr, acc in ((op.p[3], op.i[4]), (op.p[4], op.i[5])):
                    if ptr in wrote:
                        assert acc == 1 and not (open_l >> wrote[ptr]) & 1, 'accumulating into a gradient whose lane is not joined'
        assert len([1 for op in plan.bwd if op.opcode == L.OP_UPADD_BWD]) == 2
    with pytest.raises(ValueError, match='multiples of 32'):
        E.YuNetEngine(O.yunet_arch('n'), 'cpu').get_plan(1, 100, 100, 1)


def test_plan_bn_sum_blocks_are_disjoint_replica_tiles():
    """Layout of the BatchNorm sums (YunetBN::slots): every layer owns one [slots, 2c] forward block and one
    backward block, the blocks tile the single fp64 buffer the step zeroes with one memset, the descriptors handed
    to the kernels and the rows of the bn_batch tables point at those blocks with the same replica count."""
    import ctypes as C
    import yunet_amd._lib as L
    import yunet_amd.engine as E
    eng = E.YuNetEngine(O.yunet_arch('n'), 'cpu')
    plan = eng.get_plan(2, 320, 320, 3)
    R = E.BN_SLOTS
    assert R == 8
    base, total = plan.stats.data_ptr(), plan.stats.numel()
    spans = []
    for name, b in plan.bn.items():
        for key in ('stats', 'bstats'):
            v = b[key]
            assert v.shape == (R, 2 * b['c']) and v.is_contiguous() and v.dtype == torch.float64
            lo = (v.data_ptr() - base) // 8
            spans.append((lo, lo + v.numel(), name, key))
    spans.sort()
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), 'blocks overlap or leave holes'
    # the one memset of the step covers the whole buffer
    ms = plan.ops_memset_stats
    assert ms.opcode == L.OP_MEMSET and ms.p[0] == base
    assert (ms.i[0] & 0xffffffff) | (ms.i[1] << 32) == total * 8
    # descriptors inside the op records
    blocks = {(b['stats'].data_ptr(), b['bstats'].data_ptr()): b['c'] for b in plan.bn.values()}
    seen = 0
    for op in plan.fwd_a + plan.bwd:
        if op.opcode in (L.OP_DP_FWD, L.OP_DP_BWD):
            for bn, on in ((op.dp.in_bn, op.dp.in_transform == L.T_BNRELU), (op.dp.out_bn, op.dp.out_has_bn)):
                if on:
                    assert (bn.stats, bn.bstats) in blocks and bn.slots == R
                    seen += 1
    assert seen > 40
    # bn_batch tables: offset (in doubles), c, count, ..., slots
    for tab, which in ((plan.bn_table_f, 'stats'), (plan.bn_table_b, 'bstats')):
        rows = tab.numpy()
        assert rows.shape == (len(plan.bn), 7) and (rows[:, 6] == R).all()
        want = sorted((b[which].data_ptr() - base) // 8 for b in plan.bn.values())
        assert sorted(rows[:, 0].tolist()) == want


def test_plan_cache_is_bounded_lru(monkeypatch):
    """get_plan keeps the most recently used shapes (testing at original image sizes walks through hundreds of
    shapes, each plan owning all buffers of its shape); a dropped plan stays valid for whoever still holds it."""
    import yunet_amd.engine as E
    monkeypatch.setattr(E, 'MAX_PLANS', 3)
    eng = E.YuNetEngine(O.yunet_arch('s'), 'cpu')
    ps = [eng.get_plan(1, 32 * i, 64, 1) for i in range(1, 6)]
    assert [k[1] for k in eng.plans] == [96, 128, 160]
    assert eng.get_plan(1, 96, 64, 1) is ps[2] and [k[1] for k in eng.plans] == [128, 160, 96]     # a hit refreshes
    again = eng.get_plan(1, 32, 64, 1)
    assert again is not ps[0] and len(eng.plans) == 3 and ps[0].P == again.P                      # rebuilt, old one intact
    assert eng.get_plan(1, 32, 64, 100) is not again                                               # Gmax is part of the key


def test_lr_schedule_matches_mmcv_semantics():
    """regular lr = 0.01*0.1^(#steps<=epoch); linear warm-up over 1500 iters
    from ratio 0.001; checkpoint optimizer lr after both steps is 1e-4."""
    from yunet_amd.runner import StepLrWarmup
    s = StepLrWarmup(0.01, step=[400, 544], warmup='linear', warmup_iters=1500, warmup_ratio=0.001)
    assert s.lr_at(0, 0) == pytest.approx(0.01 * 0.001)
    assert s.lr_at(0, 750) == pytest.approx(0.01 * (1 - 0.5 * 0.999))
    assert s.lr_at(3, 1500) == pytest.approx(0.01)
    assert s.lr_at(400, 10 ** 6) == pytest.approx(0.001)
    assert s.lr_at(560, 10 ** 6) == pytest.approx(1e-4)


def test_synthetic_batches_are_deterministic_and_well_formed():
    import yunet_amd.synthetic as S
    a, b = S.make_batch(6, 320, 320, 99), S.make_batch(6, 320, 320, 99)
    assert torch.equal(a['img'], b['img'])
    assert float(a['img'].min()) >= 0 and float(a['img'].max()) < 255
    for ga, gb_, ka in zip(a['gt_bboxes'], b['gt_bboxes'], a['gt_keypointss']):
        assert torch.equal(ga, gb_) and 1 <= ga.shape[0] <= S.MAX_GT
        assert (ga[:, 2] > ga[:, 0]).all() and (ga[:, 3] > ga[:, 1]).all()
        assert float(ga.min()) >= 0 and float(ga[:, 2].max()) <= 320
        assert ka.shape[1:] == (5, 3) and set(ka[..., 2].unique().tolist()) <= {0.0, 1.0}
    pad = a['gt_bboxes'].padded
    for i, g in enumerate(a['gt_bboxes']):
        assert torch.equal(pad[i, :g.shape[0]], g) and int(a['gt_bboxes'].counts[i]) == g.shape[0]
    assert sum(S.WIDER_VAL_FACES_HIST) == 3226


def test_checkpoint_roundtrip(tmp_path):
    from yunet_amd.optim import FusedSGD
    from yunet_amd import runner as R
    m, _ = build('s')
    opt = FusedSGD(m, lr=0.01, momentum=0.9, weight_decay=5e-4)
    path = os.path.join(tmp_path, 'ck.pth')
    R.save_checkpoint(m, opt, path, dict(epoch=3, iter=77))
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ck) == {'meta', 'state_dict', 'optimizer'}      # reference checkpoint format
    m2, _ = build('s')
    meta = R.load_checkpoint(m2, path)
    assert meta['epoch'] == 3
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def test_device_pipeline_builds_from_reference_style_config_and_rejects_variants():
    """The PIPELINES registry accepts the reference's train pipeline list unchanged; variants the
    device stage does not implement fail at build time, not silently at run time."""
    import yunet_amd
    from yunet_amd.pipelines import DevicePipeline
    cfg = yunet_amd.Config.fromfile('configs/yunet_n.py')
    pipe = DevicePipeline(cfg.data.train.pipeline, seed=3, gmax=128)
    assert pipe.out_size == 320 and pipe.cfg.n_choice == 6 and pipe.cfg.gmax == 128
    assert list(pipe.cfg.crop_choice)[:6] == [0.5, 0.7, 0.9, 1.1, 1.3, 1.5]
    assert pipe.cfg.flip_ratio == 0.5 and pipe.cfg.pad_value == 128.0 and pipe.cfg.max_attempts == 250
    base = [dict(p) for p in cfg.data.train.pipeline]

    def variant(i, **kw):
        v = [dict(p) for p in base]
        v[i].update(kw)
        return v
    for bad in (variant(3, keep_ratio=True), variant(3, img_scale=(320, 256)), variant(3, img_scale=(300, 300)),
                variant(4, direction='vertical'), variant(5, mean=[104., 117., 123.]), variant(5, to_rgb=True),
                variant(2, crop_choice=None, crop_ratio_range=(0.3, 1.0)), base[:-1], base[::-1]):
        with pytest.raises((NotImplementedError, ValueError)):
            DevicePipeline(bad)
    with pytest.raises(NotImplementedError, match='DevicePipeline'):
        pipe.steps[2](dict())                      # the carriers have no per-sample CPU implementation


def test_loss_reduction_is_validated_for_the_fused_path():
    import yunet_amd
    cfg = yunet_amd.Config.fromfile('configs/yunet_s.py')
    cfg.model.bbox_head.loss_bbox.reduction = 'mean'
    model = yunet_amd.build_detector(cfg.model)
    with pytest.raises(NotImplementedError, match='loss_bbox.reduction'):
        model.bbox_head.loss_cfg()


def test_bench_byte_model_matches_survey_figures():
    """bench.py's algorithmic-bytes model over the real op lists reproduces SURVEY.md 8(d):
    YuNet_n 320x320 = 26.43 MB forward + 40.51 MB backward per image (unit-boundary traffic)."""
    import importlib.util
    import os
    import yunet_amd._lib as L
    import yunet_amd.engine as E
    spec = importlib.util.spec_from_file_location(
        'bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = 4
    conv = (L.OP_STEM_FWD, L.OP_DP_FWD, L.OP_POOL_FWD, L.OP_UPADD_FWD)

    def per_image(plan):
        return (sum(bench.op_bytes(op, L) for op in plan.fwd_a if op.opcode in conv) / n / 1e6,
                sum(bench.op_bytes(op, L) for op in plan.bwd) / n / 1e6)
    # the reference's op graph (separate pooling kernels, YUNET_NO_POOL_FUSION=1): SURVEY's figures
    os.environ['YUNET_NO_POOL_FUSION'] = '1'
    try:
        plan = E.YuNetEngine(O.yunet_arch('n'), 'cpu').get_plan(n, 320, 320, 3)
    finally:
        del os.environ['YUNET_NO_POOL_FUSION']
    fwd, bwd = per_image(plan)
    assert fwd == pytest.approx(26.43, rel=0.02)
    # SURVEY charges every backward unit 2*in + out, the stem included; nobody writes a gradient for the image (a leaf),
    # so the kernels' own model is 3*320*320*4 B = 1.23 MB per image lower and the reference-graph
    # model (the numerator of step_frac) keeps SURVEY's figure
    leaf = 3 * 320 * 320 * 4 / 1e6
    assert bwd == pytest.approx(40.51 - leaf, rel=0.02)
    assert sum(bench.op_bytes_reference_graph(op, L) for op in plan.bwd) / n / 1e6 == pytest.approx(40.51, rel=0.02)
    stem_b = [op for op in plan.bwd if op.opcode == L.OP_STEM_BWD][0]
    assert bench.op_bytes_reference_graph(stem_b, L) - bench.op_bytes(stem_b, L) == n * 3 * 320 * 320 * 4
    # the default plan folds the two big pools into their producers / consumers: the
    # boundaries pooled tensor -> pool kernel -> full-size gradient disappear from the byte model
    plan = E.YuNetEngine(O.yunet_arch('n'), 'cpu').get_plan(n, 320, 320, 3)
    assert sum(op.opcode == L.OP_POOL_FWD for op in plan.fwd_a) == 2
    assert sum(op.opcode == L.OP_POOL_BWD for op in plan.bwd) == 2
    fwd, bwd = per_image(plan)
    assert fwd == pytest.approx(23.35, rel=0.02)
    assert bwd == pytest.approx(30.88 - leaf, rel=0.02)
    # ... and bench.py's reference-graph model charges the fused units as unit + pooling kernel again
    ref = sum(bench.op_bytes_reference_graph(op, L) for op in list(plan.fwd_a) + list(plan.bwd) if op.opcode in conv or
              op.opcode in (L.OP_STEM_BWD, L.OP_DP_BWD, L.OP_POOL_BWD, L.OP_UPADD_BWD)) / n / 1e6
    assert ref == pytest.approx(26.43 + 40.51, rel=0.02)
    # 135.6 M MAC forward per image, 76.5 % of it in the pointwise GEMMs
    pw_flops = sum(bench.op_flops(op, L) for op in plan.fwd_a if op.opcode == L.OP_DP_FWD)
    assert pw_flops / n / 2 / 1e6 == pytest.approx(135.6 * 0.765, rel=0.03)
    # the committed PMC table is found and keyed by the names bench.py prints
    # (a kernel FAMILY: {template instance: launches per step}, the mean is weighted by the launches)
    traffic, src = bench.pmc_traffic({'dp_bwd64_kernel<8,false,false>': 1})
    assert traffic is not None and traffic > 1e8 and src.endswith('_pmc_traffic.json')
    both, _ = bench.pmc_traffic({'dp_bwd64_kernel<8,false,false>': 1, 'dp_bwd64_kernel<4,false,false>': 4})
    small, _ = bench.pmc_traffic({'dp_bwd64_kernel<4,false,false>': 4})
    assert both == pytest.approx((traffic + 4 * small) / 5, rel=1e-6)
    assert bench.pmc_traffic({'no_such_kernel<1>': 1}) == (None, None)
    assert bench.family_of('dp_bwd64_kernel<8,false,true>') == 'dp_bwd64_kernel' and bench.family_of('loss_kernel') == 'loss_kernel'


def test_bbox_mapping_back_matches_reference():
    """aug_test's un-flip + un-scale == mmdet/core/bbox/transforms.py:bbox_mapping_back."""
    import importlib.util
    import torch
    from yunet_amd.yunet import bbox_mapping_back
    g = torch.Generator().manual_seed(0)
    b = torch.rand(7, 4, generator=g) * 300
    b[:, 2:] += b[:, :2]
    ref_path = '/root/reference/mmdet/core/bbox/transforms.py'
    ref = None
    if os.path.exists(ref_path):
        spec = importlib.util.spec_from_file_location('ref_bbox_transforms', ref_path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    for flip, direction in ((False, 'horizontal'), (True, 'horizontal'), (True, 'vertical'), (True, 'diagonal')):
        for sf in ([0.5, 0.5, 0.5, 0.5], [1.25, 0.8, 1.25, 0.8]):
            meta = dict(img_shape=(480, 640, 3), scale_factor=sf, flip=flip, flip_direction=direction)
            got = bbox_mapping_back(b, meta)
            # definition: flip about the view's width / height, then divide by the scale factor
            want = b.clone()
            if flip and direction in ('horizontal', 'diagonal'):
                want[:, 0], want[:, 2] = 640 - b[:, 2], 640 - b[:, 0]
            if flip and direction in ('vertical', 'diagonal'):
                want[:, 1], want[:, 3] = 480 - b[:, 3], 480 - b[:, 1]
            want = want / torch.tensor(sf)
            assert torch.equal(got, want)
            if ref is not None:
                assert torch.equal(got, ref.bbox_mapping_back(b, (480, 640, 3), sf, flip, direction))


def test_cfg_options_create_missing_sections():
    """--cfg-options fp16.loss_scale=512. on a config without an fp16 section (mmcv Config.merge_from_dict)."""
    import yunet_amd
    cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'yunet_s.py'))
    assert cfg.get('fp16') is None
    cfg.merge_from_dict({'fp16.loss_scale': 512.0, 'data.samples_per_gpu': 16, 'a.b.c': 1})
    assert cfg.fp16.loss_scale == 512.0 and cfg.data.samples_per_gpu == 16 and cfg.a.b.c == 1
