"""not-gpu: size-independent properties of the oracles (hypothesis): what the HIP path is checked
against must itself satisfy the invariants of the algorithms it restates."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import crafted as C
import pipeline_oracle as P
import yunet_oracle as O

SET = settings(max_examples=25, deadline=None)


@SET
@given(st.integers(1, 6), st.integers(0, 10_000))
def test_iou_is_symmetric_bounded_and_one_on_the_diagonal(n, seed):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * 100
    b = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 50 + 0.5], 1)
    iou = O.pairwise_iou(b, b)
    assert torch.allclose(iou, iou.t())
    assert float(iou.min()) >= 0 and float(iou.max()) <= 1 + 1e-6
    assert torch.allclose(iou.diagonal(), torch.ones(n), atol=1e-5)


@SET
@given(st.sampled_from([(64, 64), (96, 160), (160, 160)]), st.integers(0, 10_000))
def test_decode_and_keypoint_encode_are_consistent_with_the_priors(hw, seed):
    h, w = hw
    sizes = C.featmap_sizes(h, w)
    pri = O.grid_priors(sizes, [8, 16, 32])
    assert pri.shape[0] == sum(a * b for a, b in sizes)
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(pri.shape[0], 4, generator=g) * 0.5
    box = O.bbox_decode(pri, pred)
    cx, cy = (box[:, 0] + box[:, 2]) / 2, (box[:, 1] + box[:, 3]) / 2
    assert torch.allclose(cx, pred[:, 0] * pri[:, 2] + pri[:, 0], atol=1e-3)
    assert torch.allclose(cy, pred[:, 1] * pri[:, 3] + pri[:, 1], atol=1e-3)
    assert bool((box[:, 2] > box[:, 0]).all() and (box[:, 3] > box[:, 1]).all())
    kps = torch.rand(pri.shape[0], 10, generator=g) * h
    enc = O.kps_encode(pri, kps)
    dec = enc.reshape(-1, 5, 2) * pri[:, None, 2:] + pri[:, None, :2]
    assert torch.allclose(dec.reshape(-1, 10), kps, atol=1e-3)


@SET
@given(st.integers(1, 12), st.integers(0, 10_000))
def test_simota_invariants(g_count, seed):
    """Every prior belongs to at most one GT, a GT with any candidate gets >= 1 prior, at most
    candidate_topk + ties priors per GT before conflict resolution, fg <=> gt_inds > 0."""
    import yunet_amd.synthetic as S
    h = w = 160
    b = S.make_batch(1, h, w, seed, max_gt=g_count, with_img=False)
    flat = C.crafted_preds(b['gt_bboxes'], b['gt_keypointss'], h, w, seed + 1)[0]
    sizes = C.featmap_sizes(h, w)
    pri = O.grid_priors(sizes, [8, 16, 32])
    off = torch.cat([pri[:, :2] + pri[:, 2:] * 0.5, pri[:, 2:]], 1)
    scores = flat[:, 0].sigmoid() * flat[:, 5].sigmoid()
    dec = O.bbox_decode(pri, flat[:, 1:5])
    gt = b['gt_bboxes'][0]
    gi, labels, ovl = O.simota_assign(scores, off, dec, gt, b['gt_labels'][0])
    fg = gi > 0
    assert int(gi.max()) <= gt.shape[0] and int(gi.min()) >= 0
    assert torch.equal(fg, labels >= 0)
    assert bool((ovl[~fg] == -1e5).all()) and bool((ovl[fg] >= 0).all())
    per_gt = torch.bincount(gi[fg], minlength=gt.shape[0] + 1)[1:]
    assert int(per_gt.max()) <= 10 + 4                   # dynamic_k <= candidate_topk (+ exact cost ties)
    # a GT whose box or centre region contains a prior centre is never left without priors unless a
    # conflict moved its only candidates to another GT: at least one GT is matched when any is valid
    cx, cy = off[:, 0], off[:, 1]
    in_any = ((cx[:, None] > gt[None, :, 0]) & (cx[:, None] < gt[None, :, 2]) &
              (cy[:, None] > gt[None, :, 1]) & (cy[:, None] < gt[None, :, 3])).any()
    if bool(in_any):
        assert int(fg.sum()) >= 1


@SET
@given(st.integers(30, 300), st.integers(30, 300), st.integers(1, 20), st.integers(0, 10_000),
       st.sampled_from([64, 160]))
def test_pipeline_oracle_invariants(h, w, g, seed, S):
    rng = np.random.default_rng(seed)
    img, boxes, kps = P.synth_image(rng, h, w, g)
    r = P.augment_image(img, boxes, kps, seed, 3, 0, S, [0.5, 0.7, 0.9, 1.1, 1.3, 1.5])
    left, top, cw, flip = [int(v) for v in r['params']]
    assert cw == int([0.5, 0.7, 0.9, 1.1, 1.3, 1.5][[int(c * min(h, w)) for c in
                                                      [0.5, 0.7, 0.9, 1.1, 1.3, 1.5]].index(cw)] * min(h, w))
    assert r['boxes'].shape[0] == int(r['mask'].sum()) >= 1
    assert r['boxes'].min() >= 0 and r['boxes'].max() <= S
    assert bool((r['boxes'][:, 2] >= r['boxes'][:, 0]).all() and (r['boxes'][:, 3] >= r['boxes'][:, 1]).all())
    assert r['img'].shape == (3, S, S) and r['img'].min() >= 0 and r['img'].max() <= 255
    # flipping twice is the identity on boxes and keypoints
    b2, k2 = P.flip_gt(*P.flip_gt(r['boxes'], r['kps'], S), S)
    assert np.allclose(b2, r['boxes'], atol=1e-4) and np.allclose(k2, r['kps'], atol=1e-4)
    # same (seed, iteration, image) -> same decision; another image index -> independent stream
    r2 = P.augment_image(img, boxes, kps, seed, 3, 0, S, [0.5, 0.7, 0.9, 1.1, 1.3, 1.5])
    assert np.array_equal(r2['params'], r['params']) and np.array_equal(r2['img'], r['img'])


@SET
@given(st.integers(5, 40), st.integers(5, 40), st.integers(4, 48), st.integers(0, 10_000))
def test_bilinear_resize_is_linear_and_bounded(h, w, S, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(0, 255, (h, h, 2)).astype(np.float32)       # square, as after RandomSquareCrop
    b = rng.uniform(0, 255, (h, h, 2)).astype(np.float32)
    ra, rb = P.resize_linear(a, S), P.resize_linear(b, S)
    assert np.allclose(P.resize_linear(0.25 * a + 0.5 * b, S), 0.25 * ra + 0.5 * rb, atol=1e-3)
    assert ra.min() >= a.min() - 1e-3 and ra.max() <= a.max() + 1e-3
