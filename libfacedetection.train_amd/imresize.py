"""cv2.resize(uint8, INTER_LINEAR) on the device, in its own integer arithmetic.

The reference's TEST pipeline resizes the decoded uint8 image (tools/test_widerface.py:77-96 ->
configs/yunet_n.py:57-86: LoadImageFromFile -> Resize(keep_ratio=True) -> mmcv.imrescale -> cv2.resize), so the
pixels the detector sees come out of OpenCV's 11-bit fixed-point bilinear, not out of a float interpolation:

    coefficients  fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx   (columns: sx < 0 -> (0, 0);
                  sx >= w - 1 -> (w - 1, 0); rows: taps clamped instead),  a = round_half_even((1 - fx, fx) * 2048)
    columns       H = S[sx] * a0 + S[sx + 1] * a1                                     (int32, scaled by 2^11)
    rows          out = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2
    special cases same size: copy;  exactly 2 x in both directions: (a + b + c + d + 2) >> 2  (INTER_AREA fast path)

(OpenCV modules/imgproc/src/resize.cpp; restated with per-pixel loops in oracle/cv2_resize_oracle.py, which the
tests compare this file with -- cv2 itself is not installed here, see that file's header.)

The two coefficient tables (one per axis, a few KB) are built on the host in exactly OpenCV's float / double steps;
everything per pixel is integer tensor arithmetic on the image's device, so CPU and GPU results are identical.
The image travels to the device as uint8 (1 B per sample instead of 4).
"""
import numpy as np
import torch

COEF_SCALE = 2048


def _axis_tables(dst, src, border_rule):
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if border_rule:
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= src - 1
        f[hi], s[hi] = 0.0, src - 1
    c0 = (np.float32(1.0) - f).astype(np.float32)
    w0 = np.rint(c0 * np.float32(COEF_SCALE)).astype(np.int32)      # cvRound: ties to even
    w1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    t0 = np.clip(s, 0, src - 1)
    t1 = np.clip(s + 1, 0, src - 1)
    return t0, t1, w0, w1


def resize_linear_u8(img, dsize):
    """img: uint8 tensor [h, w, c] (any device); dsize = (width, height), cv2's order -> uint8 [height, width, c]."""
    assert img.dtype == torch.uint8 and img.dim() == 3
    h, w, _ = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if dw < 1 or dh < 1:
        raise ValueError(f'resize_linear_u8: empty destination size {dsize}')
    if (dw, dh) == (w, h):
        return img.clone()
    dev = img.device
    x = img.to(torch.int32)
    if w == 2 * dw and h == 2 * dh:
        s = x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2]
        return ((s + 2) >> 2).to(torch.uint8)
    x0, x1, a0, a1 = _axis_tables(dw, w, True)
    y0, y1, b0, b1 = _axis_tables(dh, h, False)

    def t(a, dt=torch.int64):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)

    a0, a1 = t(a0, torch.int32).view(1, dw, 1), t(a1, torch.int32).view(1, dw, 1)
    hpass = x.index_select(1, t(x0)) * a0 + x.index_select(1, t(x1)) * a1            # [h, dw, c], x 2^11
    hpass = hpass >> 4
    b0, b1 = t(b0, torch.int32).view(dh, 1, 1), t(b1, torch.int32).view(dh, 1, 1)
    out = (((hpass.index_select(0, t(y0)) * b0) >> 16) + ((hpass.index_select(0, t(y1)) * b1) >> 16) + 2) >> 2
    return out.clamp_(0, 255).to(torch.uint8)


def rescale_size(w, h, scale):
    """mmcv.imrescale's target size for a (long edge, short edge) scale: factor = min(long / max(h, w),
    short / min(h, w)), each side int(side * factor + 0.5) (mmcv/image/geometric.py rescale_size / _scale_size)."""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)
