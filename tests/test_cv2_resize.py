"""cv2.resize(uint8, INTER_LINEAR) of the test pipeline: the device code (yunet_amd.imresize, integer tensor
arithmetic) against the per-pixel restatement of OpenCV's fixed-point algorithm (oracle/cv2_resize_oracle.py),
hand-derived known answers, and the properties the algorithm implies.  CPU tensors here; the arithmetic is integer,
so the same code gives the same bytes on the GPU (tests/test_detect_gpu.py runs it there)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import cv2_resize_oracle as O  # noqa: E402

from yunet_amd import imresize as R  # noqa: E402


def product(img, dsize):
    a = np.asarray(img, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return R.resize_linear_u8(torch.from_numpy(a.copy()), dsize).numpy()


# Hand-derived from modules/imgproc/src/resize.cpp (see the oracle's header), e.g. the first one, column 2:
#   fx = (2 + .5) * 4/3 - .5 = 2.8333 -> sx = 2, a = (341, 1707); H = 200 * 341 + 255 * 1707 = 503485; H >> 4 = 31467;
#   (2048 * 31467) >> 16 = 983; (983 + 0 + 2) >> 2 = 246
KATS = [
    ([[0, 100, 200, 255]], (3, 1), [[17, 150, 246]]),
    ([[10, 250]], (4, 1), [[10, 70, 190, 250]]),                  # upscale: columns outside snap to the border pixel
    ([[10], [250]], (1, 4), [[10], [70], [190], [250]]),          # rows: taps clamped, the same values
    ([[1, 3], [5, 8]], (1, 1), [[4]]),                            # exactly 2 x 2: (1 + 3 + 5 + 8 + 2) >> 2
    ([[7, 7, 7], [7, 7, 7]], (5, 7), np.full((7, 5), 7)),
]


@pytest.mark.parametrize('src,dsize,want', KATS)
def test_known_answers(src, dsize, want):
    want = np.asarray(want, np.uint8)
    assert np.array_equal(O.resize_linear_u8(np.asarray(src, np.uint8), dsize), want)
    assert np.array_equal(product(src, dsize)[:, :, 0], want)


@pytest.mark.parametrize('seed', range(12))
def test_device_code_equals_the_restatement(seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 48))
    c = int(rng.choice([1, 3]))
    dh, dw = int(rng.integers(1, 56)), int(rng.integers(1, 56))
    if seed % 4 == 0:                       # keep-ratio sizes as Resize(keep_ratio=True) produces them
        dw, dh = O.rescale_size(w, h, (32, 32))
        dw, dh = max(dw, 1), max(dh, 1)
    if seed == 5:                           # the INTER_AREA substitution
        h, w = 2 * dh, 2 * dw
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    if seed % 3 == 0:
        img[rng.random((h, w, c)) < 0.3] = 255       # saturated regions: the >> 4 / >> 16 truncations at full scale
    want = O.resize_linear_u8(img, (dw, dh))
    got = product(img, (dw, dh))
    assert got.shape == (dh, dw, c) and np.array_equal(got, want), (h, w, dh, dw)


def test_properties():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(product(img, (53, 37)), img)                                  # same size: a copy
    for v in (0, 1, 128, 254, 255):                                                     # constants survive every scale
        assert np.all(product(np.full((9, 14, 3), v, np.uint8), (31, 5)) == v)
    big = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    area = (big[0::2, 0::2].astype(int) + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2
    assert np.array_equal(product(big, (48, 32)), area.astype(np.uint8))
    # against exact bilinear with the same half-pixel geometry: the 11-bit coefficients and the two truncating
    # shifts cost at most one grey level after rounding
    for (dw, dh) in ((80, 50), (30, 21), (96, 20), (17, 64)):
        got = product(big, (dw, dh)).astype(np.float64)
        sy = np.clip((np.arange(dh) + 0.5) * 64 / dh - 0.5, 0, 63)
        sx = np.clip((np.arange(dw) + 0.5) * 96 / dw - 0.5, 0, 95)
        y0, x0 = np.floor(sy).astype(int), np.floor(sx).astype(int)
        y1, x1 = np.minimum(y0 + 1, 63), np.minimum(x0 + 1, 95)
        fy, fx = (sy - y0)[:, None, None], (sx - x0)[None, :, None]
        b = big.astype(np.float64)
        ref = (b[y0][:, x0] * (1 - fx) + b[y0][:, x1] * fx) * (1 - fy) + (b[y1][:, x0] * (1 - fx) + b[y1][:, x1] * fx) * fy
        assert np.abs(got - ref).max() <= 1.0, (dw, dh)
    with pytest.raises(ValueError):
        product(img, (0, 5))


def test_rescale_size_is_mmcv_imrescale():
    # mmcv.imrescale((long, short)): factor = min(long / max(h, w), short / min(h, w)); int(side * factor + 0.5)
    assert R.rescale_size(1024, 768, (640, 640)) == (640, 480)
    assert R.rescale_size(768, 1024, (640, 640)) == (480, 640)
    assert R.rescale_size(1024, 683, (1100, 1650)) == (1649, 1100)      # factor 1100 / 683: 1649.2 + .5, 1100.0 + .5
    assert R.rescale_size(333, 500, (320, 320)) == (213, 320)
    assert R.rescale_size(333, 500, (320, 320)) == O.rescale_size(333, 500, (320, 320))


def test_prepare_test_image_resizes_in_uint8():
    """The device test pipeline (Resize keep_ratio -> Normalize(0, 1) -> Pad): the pixels are the fixed-point
    resize of the uint8 image, the rest zero padding up to a multiple of 32; scale None keeps the image."""
    from yunet_amd import evaluation as E
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)
    x, meta = E.prepare_test_image(img[:, ::-1], (64, 64), 'cpu')          # a negatively strided view, as decoders return
    nw, nh = O.rescale_size(70, 45, (64, 64))
    assert (nw, nh) == (64, 41) and meta['img_shape'] == (nh, nw, 3) and meta['ori_shape'] == (45, 70, 3)
    assert x.dtype == torch.float32 and x.shape == (1, 3, 64, 64) and meta['pad_shape'] == (64, 64, 3)
    want = O.resize_linear_u8(np.ascontiguousarray(img[:, ::-1]), (nw, nh))
    assert np.array_equal(x[0, :, :nh, :nw].permute(1, 2, 0).numpy(), want.astype(np.float32))
    assert float(x[0, :, nh:].abs().max()) == 0.0
    assert np.allclose(meta['scale_factor'], [nw / 70, nh / 45, nw / 70, nh / 45])
    y, m2 = E.prepare_test_image(img, None, 'cpu')
    assert y.shape == (1, 3, 64, 96) and np.array_equal(y[0, :, :45, :70].permute(1, 2, 0).numpy(), img.astype(np.float32))
    z, _ = E.prepare_test_image(img[:, ::-1], (64, 64), 'cpu', resize='float')
    assert float((z - x).abs().max()) <= 1.5          # the fp32 bilinear it replaces: within the rounding of one grey level
