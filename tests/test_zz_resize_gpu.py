"""GPU: the uint8 fixed-point resize of the test pipeline (yunet_amd.imresize) gives the bytes of the CPU run of the
same code -- which tests/test_cv2_resize.py holds against the per-pixel restatement of OpenCV's algorithm."""
import numpy as np
import pytest
import torch

from yunet_amd import evaluation as E
from yunet_amd import imresize as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('h,w,dw,dh', [(45, 70, 64, 41), (480, 640, 320, 240), (333, 500, 213, 320), (768, 1024, 1650, 1238),
                                       (64, 96, 48, 32), (7, 5, 5, 7)])
def test_resize_on_the_device_equals_the_cpu_run(h, w, dw, dh):
    rng = np.random.default_rng(h * 1000 + w)
    img = torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    want = R.resize_linear_u8(img, (dw, dh))
    got = R.resize_linear_u8(img.cuda(), (dw, dh))
    assert got.is_cuda and got.dtype == torch.uint8 and torch.equal(got.cpu(), want)


def test_prepare_test_image_on_the_device():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    xc, mc = E.prepare_test_image(img, (320, 320), 'cpu')
    xg, mg = E.prepare_test_image(img, (320, 320), 'cuda')
    assert xg.is_cuda and torch.equal(xg.cpu(), xc) and mg['img_shape'] == mc['img_shape'] and mg['pad_shape'] == mc['pad_shape']
