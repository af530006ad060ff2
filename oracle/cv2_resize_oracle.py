"""TEST INFRASTRUCTURE ONLY -- CPU restatement of cv2.resize(src_u8, dsize, interpolation=INTER_LINEAR) for
8-bit images: the pixel arithmetic of the reference's TEST pipeline.  Nothing in the product imports this.

Where the reference reaches it: tools/test_widerface.py:77-96 sets MultiScaleFlipAug.img_scale, the pipeline of
configs/yunet_n.py:57-86 runs LoadImageFromFile (uint8 BGR) -> Resize(keep_ratio=True)
(mmdet/datasets/pipelines/transforms.py:225-258: mmcv.imrescale -> mmcv.imresize -> cv2.resize, 'bilinear' =
cv2.INTER_LINEAR) -> Normalize(mean 0, std 1) -> Pad.  The image is still uint8 when it is resized, so OpenCV takes
its FIXED-POINT path, not the float one the train pipeline (to_float32=True) takes.

Third-party code, not under /root/reference and not installed here: OpenCV (mmcv 1.3.17 requires opencv-python >= 3;
the arithmetic below is modules/imgproc/src/resize.cpp as of OpenCV 3.4 / 4.x, unchanged across those versions):

  cv::resize            dsize == ssize -> copy.  inv_scale = dsize / ssize (double), scale = 1. / inv_scale.
                        INTER_LINEAR with scale_x == scale_y == 2 exactly is replaced by INTER_AREA, whose fast 2 x 2
                        path for uchar is (a + b + c + d + 2) >> 2.
  resize(), coefficient loops
                        fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx;
                        sx < 0 -> fx = 0, sx = 0;  sx >= width - 1 -> fx = 0, sx = width - 1;
                        ialpha = saturate_cast<short>((1.f - fx, fx) * INTER_RESIZE_COEF_SCALE)  (2048, cvRound =
                        round half to even).  The same for rows, WITHOUT the border rule: the row taps are
                        clip(sy + k, 0, height) instead.
  HResizeLinear<uchar, int, short, 2048>
                        D[dx] = S[sx] * a0 + S[sx + cn] * a1           (int, scaled by 2^11; dx >= xmax: S[sx] * 2048)
  VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>, VResizeLinearVec_32s8u>
                        dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
                        (the scalar tail and the SIMD body compute the same expression)

PARITY UNPINNED: cv2 cannot be imported here, so no output of the real library pins this file; it follows the
published source, and tests/test_cv2_resize.py holds hand-derived known answers and the properties the algorithm
implies (identity, constant images, exact 2 x 2 averaging, at most one grey level from exact bilinear + rounding).
Written with explicit per-pixel loops on purpose: it is the checker of the vectorised product code
(libfacedetection.train_amd/imresize.py), not a second copy of it.
"""
import math

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _cv_round(v):
    """cvRound of a float: nearest integer, ties to even (lrint under the default rounding mode)."""
    return int(np.rint(np.float32(v)))


def _axis_tables(dst, src, border_rule):
    scale = 1.0 / (float(dst) / float(src))
    ofs, coef = [], []
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)          # double expression, then the cast
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if border_rule:
            if s < 0:
                f, s = np.float32(0.0), 0
            if s >= src - 1:
                f, s = np.float32(0.0), src - 1
        c0 = np.float32(1.0) - f
        ofs.append(s)
        coef.append((_cv_round(c0 * np.float32(COEF_SCALE)), _cv_round(f * np.float32(COEF_SCALE))))
    return ofs, coef


def resize_linear_u8(img, dsize):
    """img uint8 [h, w] or [h, w, c]; dsize = (width, height) as cv2 takes it -> uint8 [height, width(, c)]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w, cn = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (w, h):
        out = img.copy()
        return out[:, :, 0] if squeeze else out
    out = np.zeros((dh, dw, cn), np.uint8)
    if w == 2 * dw and h == 2 * dh:                      # INTER_LINEAR -> INTER_AREA, ResizeAreaFast 2 x 2
        for y in range(dh):
            for x in range(dw):
                for c in range(cn):
                    s = int(img[2 * y, 2 * x, c]) + int(img[2 * y, 2 * x + 1, c]) + \
                        int(img[2 * y + 1, 2 * x, c]) + int(img[2 * y + 1, 2 * x + 1, c])
                    out[y, x, c] = (s + 2) >> 2
        return out[:, :, 0] if squeeze else out
    xofs, ialpha = _axis_tables(dw, w, True)
    yofs, ibeta = _axis_tables(dh, h, False)
    rows = {}                                            # source row -> its horizontal pass (int, x 2048)

    def hrow(sy):
        if sy not in rows:
            r = np.zeros((dw, cn), np.int64)
            for dx in range(dw):
                sx = xofs[dx]
                a0, a1 = ialpha[dx]
                for c in range(cn):
                    if sx + 1 < w:
                        r[dx, c] = int(img[sy, sx, c]) * a0 + int(img[sy, sx + 1, c]) * a1
                    else:                                # dx >= xmax
                        r[dx, c] = int(img[sy, sx, c]) * COEF_SCALE
            rows[sy] = r
        return rows[sy]

    def clip(v):
        return 0 if v < 0 else (v if v < h else h - 1)

    for dy in range(dh):
        s0, s1 = hrow(clip(yofs[dy])), hrow(clip(yofs[dy] + 1))
        b0, b1 = ibeta[dy]
        for dx in range(dw):
            for c in range(cn):
                v = (((b0 * (int(s0[dx, c]) >> 4)) >> 16) + ((b1 * (int(s1[dx, c]) >> 4)) >> 16) + 2) >> 2
                out[dy, dx, c] = min(max(v, 0), 255)
    return out[:, :, 0] if squeeze else out


def rescale_size(w, h, scale):
    """mmcv.image.geometric.rescale_size for a (long, short) tuple scale: the keep-ratio size of Resize."""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)
