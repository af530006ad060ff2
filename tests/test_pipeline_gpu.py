"""-m gpu: the device input pipeline (csrc/augment.hip through the C ABI) against
(a) fixtures produced by the unmodified reference transforms (crop window, kept boxes, box /
keypoint values, flip: bit-exact) and (b) the numpy oracle on seeded batches (everything
bit-exact, pixels included -- the oracle's bilinear is a restatement of cv2's, see its header)."""
import numpy as np
import pytest
import torch

import pipeline_oracle as P
from test_pipeline_oracle import load_case

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REF_PIPELINE = [
    dict(type='LoadImageFromFile', to_float32=True),
    dict(type='LoadAnnotations', with_bbox=True, with_keypoints=True),
    dict(type='RandomSquareCrop', crop_choice=[0.5, 0.7, 0.9, 1.1, 1.3, 1.5]),
    dict(type='Resize', img_scale=(640, 640), keep_ratio=False),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='Normalize', mean=[0., 0., 0.], std=[1., 1., 1.], to_rgb=False),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore', 'gt_keypointss']),
]


def make_pipe(S, seed, gmax=64):
    from yunet_amd.pipelines import DevicePipeline
    cfg = [dict(p) for p in REF_PIPELINE]
    cfg[3]['img_scale'] = (S, S)
    return DevicePipeline(cfg, seed=seed, gmax=gmax)


def run_device(srcs, S, seed, iteration, gmax=64):
    from yunet_amd.pipelines import SourceBatch
    pipe = make_pipe(S, seed, gmax)
    sb = SourceBatch.from_lists([s[0] for s in srcs], [s[1] for s in srcs], [s[2] for s in srcs], DEV)
    out = pipe(sb, iteration)
    torch.cuda.synchronize()
    return pipe, out


@pytest.mark.parametrize('name', ['pipeline_s160.npz', 'pipeline_s320.npz'])
def test_device_pipeline_vs_reference_fixtures(name):
    g, seed, it, S, srcs = load_case(name)
    pipe, out = run_device(srcs, S, seed, it)
    assert pipe.check() == []
    params = pipe.params.cpu().numpy()
    cnt = out['gt_bboxes'].counts.cpu().numpy()
    gb, gk = out['gt_bboxes'].padded.cpu().numpy(), out['gt_keypointss'].padded.cpu().numpy()
    img = out['img'].cpu().numpy()
    for i in range(len(srcs)):
        cw, flip, draws, kept = [int(v) for v in g[f'meta_{i}']]
        assert (int(params[i, 2]), int(params[i, 3]), int(params[i, 5]), int(params[i, 4])) == \
            (cw, flip, draws, kept), f'decision differs from the reference (image {i})'
        assert int(cnt[i]) == kept
        assert np.array_equal(gb[i, :kept], g[f'boxes_{i}']), f'boxes differ (image {i})'
        assert np.array_equal(gk[i, :kept], g[f'kps_{i}']), f'keypoints differ (image {i})'
        assert not gb[i, kept:].any() and not gk[i, kept:].any()
        im = img[i]
        assert np.array_equal(im[:, :16, :16], g[f'img_corner_{i}'])
        assert np.array_equal(im[:, S // 2 - 8:S // 2 + 8, S // 2 - 8:S // 2 + 8], g[f'img_center_{i}'])
        dig = np.stack([im.astype(np.float64).sum((1, 2)), (im.astype(np.float64) ** 2).sum((1, 2))])
        assert np.array_equal(dig, g[f'img_digest_{i}'])


@pytest.mark.parametrize('S,n,seed,it', [(160, 24, 3, 0), (320, 16, 4, 17), (640, 6, 5, 2)])
def test_device_pipeline_vs_oracle(S, n, seed, it):
    rng = np.random.default_rng(100 + seed)
    srcs = []
    for i in range(n):
        h, w = int(rng.integers(60, 700)), int(rng.integers(60, 700))
        g = int(rng.integers(1, 40)) if i % 5 else -int(rng.integers(1, 3))
        srcs.append(P.synth_image(rng, h, w, g))
    pipe, out = run_device(srcs, S, seed, it)
    assert pipe.check() == []
    params = pipe.params.cpu().numpy()
    gb, gk = out['gt_bboxes'].padded.cpu().numpy(), out['gt_keypointss'].padded.cpu().numpy()
    cnt = out['gt_bboxes'].counts.cpu().numpy()
    img = out['img'].cpu().numpy()
    flips = 0
    for i, (im_u8, boxes, kps) in enumerate(srcs):
        r = P.augment_image(im_u8, boxes, kps, seed, it, i, S, pipe.steps[2].crop_choice)
        assert np.array_equal(params[i, :4], r['params']), f'window / flip differ (image {i})'
        k = r['boxes'].shape[0]
        assert int(cnt[i]) == k and int(params[i, 4]) == k
        assert np.array_equal(gb[i, :k], r['boxes']) and np.array_equal(gk[i, :k], r['kps'])
        assert np.array_equal(img[i], r['img']), f'pixels differ from the oracle (image {i})'
        flips += int(params[i, 3])
    assert 0 < flips < n


def test_device_pipeline_edge_cases():
    """No GT / unreachable GT -> status 1 and check() raises; more kept boxes than gmax ->
    first gmax kept, status 2; a 1.5x window on a tiny image pads with 128."""
    rng = np.random.default_rng(9)
    img0, b0, k0 = P.synth_image(rng, 100, 120, 3)
    many = P.synth_image(rng, 400, 400, 200)
    tiny = (np.full((40, 40, 3), 9, np.uint8), np.array([[10, 10, 30, 30]], np.float32),
            np.full((1, 5, 3), -1, np.float32))
    srcs = [(img0, b0[:0], k0[:0]), many, tiny]
    pipe, out = run_device(srcs, 160, 21, 0, gmax=64)
    params = pipe.params.cpu().numpy()
    assert params[0, 6] == 1 and params[0, 2] == 0 and int(out['gt_bboxes'].counts[0]) == 0
    assert float(out['img'][0].min()) == 128.0 and float(out['img'][0].max()) == 128.0
    with pytest.raises(ValueError, match='no window'):
        pipe.check()
    r = P.augment_image(many[0], many[1], many[2], 21, 0, 1, 160, pipe.steps[2].crop_choice)
    if r['boxes'].shape[0] > 64:
        assert params[1, 6] == 2 and int(out['gt_bboxes'].counts[1]) == 64
        assert np.array_equal(out['gt_bboxes'].padded[1].cpu().numpy(), r['boxes'][:64])
    r = P.augment_image(tiny[0], tiny[1], tiny[2], 21, 0, 2, 160, pipe.steps[2].crop_choice)
    assert np.array_equal(out['img'][2].cpu().numpy(), r['img'])
    vals = np.unique(out['img'][2].cpu().numpy())
    if params[2, 2] > 40:
        assert 128.0 in vals and 9.0 in vals


def test_device_pipeline_feeds_train_step_full_batch():
    """BASELINE shape: 256 sources -> 320x320 batch -> one training step; repeatable per
    (seed, iteration), different across iterations, flips ~ Bernoulli(0.5), boxes inside [0, S]."""
    import yunet_amd
    from yunet_amd.optim import FusedSGD
    from yunet_amd.pipelines import SourceBatch
    rng = np.random.default_rng(1)
    srcs = [P.synth_image(rng, int(rng.integers(200, 500)), int(rng.integers(200, 500)), int(rng.integers(1, 30)))
            for _ in range(256)]
    sb = SourceBatch.from_lists([s[0] for s in srcs], [s[1] for s in srcs], [s[2] for s in srcs], DEV)
    pipe = make_pipe(320, 5)
    a = pipe(sb, 3)
    pa = pipe.params.clone()
    b = pipe(sb, 3)
    assert torch.equal(pa, pipe.params) and torch.equal(a['img'], b['img'])
    assert torch.equal(a['gt_bboxes'].padded, b['gt_bboxes'].padded)
    c = pipe(sb, 4)
    assert not torch.equal(pa[:, :3], pipe.params[:, :3])
    assert pipe.check() == []
    assert 90 < int(pa[:, 3].sum()) < 166
    gb = a['gt_bboxes'].padded
    assert float(gb.min()) >= 0 and float(gb.max()) <= 320
    assert int(a['gt_bboxes'].counts.min()) >= 1
    torch.manual_seed(0)
    cfg = yunet_amd.Config.fromfile('configs/yunet_n.py')
    model = yunet_amd.build_detector(cfg.model).to(DEV).train()
    opt = FusedSGD(model, lr=1e-4, momentum=0.9, weight_decay=5e-4)
    out = model.train_step(c, opt)
    opt.zero_grad()
    out['loss'].backward()
    opt.step()
    torch.cuda.synchronize()
    loss = float(out['log_vars']['loss'])
    assert loss == loss and 0 < loss < 1e4


def test_host_fed_sources_equal_resident_sources():
    """runner.SyntheticSourceImages(host_fed=True): the batch's decoded sources live in pinned host memory and travel on a
    copy stream into one of two device buffers under the previous step -- the pipeline's outputs must be bit-identical to
    the resident mode's for every iteration (double-buffer hand-over, events), and `report()` carries the upload timing."""
    import os
    import yunet_amd
    import yunet_amd.runner as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yunet_amd.Config.fromfile(os.path.join(root, 'configs', 'yunet_s.py'))
    kw = dict(samples_per_gpu=12, pool=5, seed=3, src_hw=((300, 420), (512, 384), (200, 200)))
    a = R.SyntheticSourceImages(cfg.train_pipeline, **kw)
    b = R.SyntheticSourceImages(cfg.train_pipeline, host_fed=True, timing=True, **kw)
    for it in range(5):
        ba, bb = a.batch(it, 'cuda'), b.batch(it, 'cuda')
        torch.cuda.synchronize()
        assert torch.equal(ba['img'], bb['img']), it
        assert torch.equal(ba['gt_bboxes'].padded, bb['gt_bboxes'].padded) and torch.equal(ba['gt_bboxes'].counts, bb['gt_bboxes'].counts)
        assert torch.equal(ba['gt_keypointss'].padded, bb['gt_keypointss'].padded)
    rep = b.report(skip=0)
    assert rep['batches_timed'] == 5 and rep['h2d_ms'] > 0 and rep['pipeline_ms'] > 0 and rep['h2d_bytes'] > 12 * 200 * 200 * 3 - 1
