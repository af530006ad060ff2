"""not-gpu: the detection oracle (oracle/detect_oracle.py) against fixtures produced by the
unmodified reference `simple_test` in eval mode (oracle/make_golden_detect.py).  The NMS inside
that reference run is the oracle's own restatement of mmcv.ops.batched_nms (compiled op, not
installable here) -- everything before and after it is the reference's code."""
import os

import numpy as np
import pytest
import torch

import detect_oracle as D

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load_case(name):
    g = np.load(os.path.join(GOLD, name))
    kind, size, n, seed = str(g['kind']), int(g['size']), int(g['n']), int(g['seed'])
    arch, sd = D.make_state(kind, seed, size)
    img = D.structured_images(n, size, seed)
    return g, arch, sd, img


@pytest.mark.parametrize('name', ['detect_s_160.npz', 'detect_n_320.npz'])
def test_detect_oracle_matches_reference(name):
    g, arch, sd, img = load_case(name)
    flat, sizes = D.eval_flat(img, sd, arch)
    assert int(((flat[..., 0].sigmoid() * flat[..., 5].sigmoid()) >= 0.02).sum()) == int(g['candidates'])
    res = D.get_bboxes(flat, sizes, arch['strides'], float(g['score_thr']), float(g['iou_thr']))
    for i, (d, k) in enumerate(res):
        ref = g[f'dets_{i}']
        assert d.shape == ref.shape and k.shape == (len(ref), 10)
        assert np.allclose(d.numpy(), ref, rtol=1e-5, atol=1e-4)
        assert bool((d[:-1, 4] >= d[1:, 4]).all())           # descending score


def test_nms_known_answers():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 5], [21, 21, 29, 29]],
                         dtype=torch.float32)
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6, 0.95])
    # IoU(0,1) = 81/119 = 0.68 > 0.45 -> 1 dropped; IoU(0,3) = 0.5 -> dropped; IoU(4,2) = 0.64 -> 2 dropped
    assert D.nms_greedy(boxes, scores, 0.45).tolist() == [4, 0]
    assert D.nms_greedy(boxes, scores, 0.60).tolist() == [4, 0, 3]
    assert D.nms_greedy(boxes, scores, 0.70).tolist() == [4, 0, 1, 2, 3]
    # IoU exactly at the threshold is NOT suppressed (strict >)
    b = torch.tensor([[0, 0, 2, 2], [0, 0, 2, 1]], dtype=torch.float32)
    assert D.nms_greedy(b, torch.tensor([0.9, 0.8]), 0.5).tolist() == [0, 1]
    # ties keep the lower index first; empty input
    assert D.nms_greedy(torch.tensor([[0, 0, 1, 1], [5, 5, 6, 6]], dtype=torch.float32),
                        torch.tensor([0.5, 0.5]), 0.45).tolist() == [0, 1]
    assert D.nms_greedy(torch.zeros(0, 4), torch.zeros(0), 0.45).numel() == 0


@pytest.mark.parametrize('kind,h,w', [('n', 256, 384), ('s', 320, 192), ('n', 640, 640)])
def test_rectangular_inputs_match_live_reference(kind, h, w):
    """Beyond the two square fixtures: rectangular inputs through the
    UNMODIFIED `simple_test` (eval forward, flatten, sigmoid, decode, threshold, bbox2result; NMS = the restated
    greedy algorithm, as for the fixtures) and through the oracle."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    import ref_stub
    import yunet_oracle as O
    if not ref_stub.available():
        pytest.skip('reference tree not present')
    ns = ref_stub.load_reference()
    ns.mods['mmdet.models.dense_heads.yunet_head'].batched_nms = D.batched_nms
    ns.mods['mmdet.models.detectors.yunet'].bbox2result = importlib.import_module('mmdet.core.bbox.transforms').bbox2result
    model, _ = ref_stub.build_detector(f'yunet_{kind}.py')
    arch, sd = D.make_state(kind, 5, size=160)          # weights calibrated to fire on the structured images
    model.load_state_dict(sd, strict=True)
    model.eval()
    assert arch == O.yunet_arch(kind)
    # (same host, same torch kernels on both sides: no need to keep the decisions away from their thresholds
    #  as the committed fixtures do for the GPU's expf)
    gen = torch.Generator().manual_seed(1000 + h)
    big = D.structured_images(2, max(h, w), h + w)
    img = (big[:, :, :h, :w] * 0.8 + torch.rand(2, 3, h, w, generator=gen) * 50).contiguous()
    flat, sizes = D.eval_flat(img, sd, arch)
    assert sizes == [(h // s, w // s) for s in arch['strides']]
    metas = [dict(img_shape=(h, w, 3), scale_factor=np.ones(4, np.float32)) for _ in range(2)]
    with torch.no_grad():
        res = model.simple_test(img, metas, rescale=False)
    got = D.get_bboxes(flat, sizes, arch['strides'], 0.02, 0.45)
    total = 0
    for i, (d, _) in enumerate(got):
        ref = res[i][0]
        assert d.shape == ref.shape, (i, d.shape, ref.shape)
        assert np.allclose(d.numpy(), ref, rtol=1e-5, atol=1e-4)
        total += len(ref)
    assert total >= 1
