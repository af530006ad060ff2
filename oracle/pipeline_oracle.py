"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the reference TRAIN input pipeline
for the device augmentation path.  Nothing in the product imports this.

Restated steps (configs/yunet_n.py:36-56 of the reference):
  RandomSquareCrop(crop_choice=[0.5 .. 1.5])   mmdet/datasets/pipelines/transforms.py:975-1169
  Resize(img_scale=(S, S), keep_ratio=False)   transforms.py:242-299 (+ mmcv.imresize -> cv2.resize)
  RandomFlip(flip_ratio=0.5), 5-point swap     transforms.py:425-546
  Normalize(mean 0, std 1, to_rgb=False)       exact no-op
  DefaultFormatBundle / collate                HWC -> CHW, ragged GT -> padded [N, Gmax, .] + counts

Randomness.  The reference draws from numpy's global MT19937 stream; a device kernel cannot
reproduce that stream, so the DECISION LOGIC is restated over a counter-based generator
(`rand_u32(seed, image, counter)`, pure 32-bit integer arithmetic, identical in csrc/augment.hip).
Pinning: `oracle/make_golden_pipeline.py` runs the UNMODIFIED reference transform classes with
`np.random.choice / randint` redirected to this same generator, so crop windows, kept boxes,
transformed boxes / keypoints and flip decisions are pinned bit-for-bit
(tests/golden/pipeline_*.npz).

PARITY UNPINNED for the pixel values of the Resize step only: the reference calls
cv2.resize(INTER_LINEAR) and cv2 is not installed in this environment.  `resize_linear` restates
OpenCV's documented float32 bilinear (half-pixel centres, edge clamp, horizontal pass then
vertical pass, no fused multiply-add); the device kernel is bit-exact against THIS restatement.
"""
import numpy as np

M32 = 0xFFFFFFFF
FLIP_ORDER = (1, 0, 2, 4, 3)          # transforms.py:497


# ------------------------------------------------------------------ counter-based generator
def mix32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def stream_key(seed, iteration, image):
    """Per (run seed, iteration, image) key; `rand_u32(key, ctr)` is the ctr-th draw."""
    k = mix32((seed & M32) ^ ((iteration * 0x27D4EB2F) & M32))
    return mix32(k ^ ((image * 0x9E3779B9) & M32))


def rand_u32(key, ctr):
    return mix32(key ^ ((ctr * 0x85EBCA6B + 0xC2B2AE35) & M32))


def bounded(u32, n):
    """floor(u * n) for u = u32 / 2^32 -- the integer in [0, n)."""
    return (u32 * n) >> 32


class Stream:
    """The draws of one image, in order."""

    def __init__(self, seed, iteration, image):
        self.key = stream_key(seed, iteration, image)
        self.ctr = 0

    def next_u32(self):
        v = rand_u32(self.key, self.ctr)
        self.ctr += 1
        return v

    def randint(self, low, high):          # numpy.random.randint(low, high): [low, high)
        return low + bounded(self.next_u32(), high - low)

    def choice_index(self, n):             # numpy.random.choice(a) without p
        return bounded(self.next_u32(), n)

    def uniform(self):                     # numpy.random.random_sample()
        return self.next_u32() / 4294967296.0


# ------------------------------------------------------------------ RandomSquareCrop decision
def centers_in_patch(boxes, patch):
    """transforms.py:1079-1086 (strict inequalities)."""
    c = (boxes[:, :2] + boxes[:, 2:]) / 2
    return (c[:, 0] > patch[0]) & (c[:, 1] > patch[1]) & (c[:, 0] < patch[2]) & (c[:, 1] < patch[3])


def decide_crop(h, w, boxes, crop_choice, st, max_attempts=250, max_retries=64):
    """transforms.py:1032-1090.  Returns (left, top, cw) or None when no window with a box centre
    inside was found (the reference would loop forever; max_retries bounds it)."""
    if boxes.shape[0] == 0:
        return None
    for _ in range(max_retries):
        # max(crop_choice) = 1.5 > 1.0, so a new scale is drawn on every retry (:1044-1052)
        scale = float(crop_choice[st.choice_index(len(crop_choice))])
        for _ in range(max_attempts):
            cw = int(scale * min(w, h))
            ch = cw
            if w == cw:
                left = 0
            elif w > cw:
                left = st.randint(0, w - cw)
            else:
                left = st.randint(w - cw, 0)
            if h == ch:
                top = 0
            elif h > ch:
                top = st.randint(0, h - ch)
            else:
                top = st.randint(h - ch, 0)
            patch = (left, top, left + cw, top + ch)
            if centers_in_patch(boxes, patch).any():
                return left, top, cw
    return None


# ------------------------------------------------------------------ GT transforms
def crop_gt(boxes, kps, left, top, cw):
    """transforms.py:1091-1125: keep boxes whose centre is inside, clip to the window, shift."""
    patch = np.array([left, top, left + cw, top + cw], dtype=np.int64)
    mask = centers_in_patch(boxes, patch)
    b = boxes[mask].copy()
    b[:, 2:] = np.minimum(b[:, 2:], patch[2:]).astype(np.float32)
    b[:, :2] = np.maximum(b[:, :2], patch[:2]).astype(np.float32)
    b = (b.astype(np.float64) - np.tile(patch[:2], 2)).astype(np.float32)
    k = kps[mask].copy()
    k[:, :, :2] = np.minimum(k[:, :, :2], patch[2:]).astype(np.float32)
    k[:, :, :2] = np.maximum(k[:, :, :2], patch[:2]).astype(np.float32)
    k[:, :, 0] = (k[:, :, 0].astype(np.float64) - patch[0]).astype(np.float32)
    k[:, :, 1] = (k[:, :, 1].astype(np.float64) - patch[1]).astype(np.float32)
    return b, k, mask


def resize_gt(boxes, kps, cw, S):
    """transforms.py:277-299 with mmcv.imresize's w_scale = S / w (python floats -> float32)."""
    sf = np.array([S / cw, S / cw, S / cw, S / cw], dtype=np.float32)
    b = boxes * sf
    b[:, 0::2] = np.clip(b[:, 0::2], 0, S)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, S)
    k = kps.copy()
    k[:, :, 0] *= sf[0]
    k[:, :, 1] *= sf[1]
    k[:, :, 0] = np.clip(k[:, :, 0], 0, S)
    k[:, :, 1] = np.clip(k[:, :, 1], 0, S)
    return b, k


def flip_gt(boxes, kps, S):
    """transforms.py:457-503 (horizontal)."""
    b = boxes.copy()
    b[:, 0] = S - boxes[:, 2]
    b[:, 2] = S - boxes[:, 0]
    k = kps[:, list(FLIP_ORDER), :].copy()
    k[:, :, 0] = S - k[:, :, 0]
    return b.astype(np.float32), k.astype(np.float32)


# ------------------------------------------------------------------ image
def crop_image(img, left, top, cw, pad=128.0):
    """transforms.py:1127-1147: square window, everything outside the source is `pad`."""
    h, w = img.shape[:2]
    out = np.full((cw, cw, img.shape[2]), pad, dtype=np.float32)
    x0, y0, x1, y1 = max(0, left), max(0, top), min(w, left + cw), min(h, top + cw)
    tx, ty = max(0, -left), max(0, -top)
    out[ty:ty + (y1 - y0), tx:tx + (x1 - x0)] = img[y0:y1, x0:x1]
    return out


def linear_coeffs(dst, src):
    """OpenCV resize INTER_LINEAR coefficient table for one axis (float32)."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, (np.float32(1.0) - f).astype(np.float32), f


def resize_linear(img, S):
    """float32 bilinear resize to S x S: horizontal pass, then vertical pass, no FMA."""
    h, w = img.shape[:2]
    sx, sx1, a0, a1 = linear_coeffs(S, w)
    sy, sy1, b0, b1 = linear_coeffs(S, h)
    img = img.astype(np.float32)
    rows = img[:, sx] * a0[None, :, None] + img[:, sx1] * a1[None, :, None]      # [h, S, c]
    rows = rows.astype(np.float32)
    out = rows[sy] * b0[:, None, None] + rows[sy1] * b1[:, None, None]
    return out.astype(np.float32)


# ------------------------------------------------------------------ whole pipeline
def augment_image(img_u8, boxes, kps, seed, iteration, image, S, crop_choice, flip_ratio=0.5,
                  pad=128.0):
    """One image through crop -> resize -> flip.  img_u8 [h, w, 3] uint8 (the reference converts
    to float32 at load).  Returns dict(img [3,S,S] f32, boxes, kps, params, kept mask)."""
    h, w = img_u8.shape[:2]
    st = Stream(seed, iteration, image)
    dec = decide_crop(h, w, boxes, crop_choice, st)
    if dec is None:
        raise ValueError('no crop window contains a box centre (image without usable GT)')
    left, top, cw = dec
    b, k, mask = crop_gt(boxes, kps, left, top, cw)
    b, k = resize_gt(b, k, cw, S)
    # RandomFlip: np.random.choice(['horizontal', None], p=[r, 1 - r]) = cdf search of one uniform
    flip = st.uniform() < flip_ratio
    im = resize_linear(crop_image(img_u8.astype(np.float32), left, top, cw, pad), S)
    if flip:
        b, k = flip_gt(b, k, S)
        im = im[:, ::-1]
    return dict(img=np.ascontiguousarray(im.transpose(2, 0, 1)), boxes=b, kps=k,
                params=np.array([left, top, cw, int(flip)], dtype=np.int32), mask=mask)


def collate(results, gmax):
    """Ragged per-image GT -> padded [N, gmax, .] + counts (the engine's staged-GT format);
    boxes beyond gmax are dropped (first gmax kept, order preserved)."""
    n = len(results)
    gb = np.zeros((n, gmax, 4), np.float32)
    gk = np.zeros((n, gmax, 5, 3), np.float32)
    cnt = np.zeros(n, np.int32)
    for i, r in enumerate(results):
        c = min(gmax, r['boxes'].shape[0])
        gb[i, :c], gk[i, :c], cnt[i] = r['boxes'][:c], r['kps'][:c], c
    return gb, gk, cnt


# ------------------------------------------------------------------ synthetic sources (tests)
def synth_image(rng, h, w, g):
    """A uint8 image and g face boxes with 5 landmarks (some invisible: -1 rows as in labelv2).
    g < 0: |g| tiny boxes squeezed into the top-left corner, so most crop windows miss them
    (exercises the retry loops of RandomSquareCrop)."""
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    if g < 0:
        g = -g
        x1 = rng.uniform(0, 0.04 * w, size=g)
        y1 = rng.uniform(0, 0.04 * h, size=g)
        bw = bh = np.full(g, 6.0)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        kps = np.full((g, 5, 3), -1.0, np.float32)
        return img, boxes, kps
    side = np.exp(rng.uniform(np.log(6), np.log(min(h, w) * 0.5), size=g))
    asp = rng.uniform(1.0, 1.4, size=g)
    bw, bh = side, np.minimum(side * asp, h - 1)
    x1 = rng.uniform(0, w - bw)
    y1 = rng.uniform(0, h - bh)
    boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
    kps = np.zeros((g, 5, 3), np.float32)
    kps[:, :, 0] = (x1[:, None] + rng.uniform(0, 1, (g, 5)) * bw[:, None])
    kps[:, :, 1] = (y1[:, None] + rng.uniform(0, 1, (g, 5)) * bh[:, None])
    kps[:, :, 2] = 1.0
    inv = rng.uniform(size=g) < 0.3
    kps[inv] = -1.0
    return img, boxes, kps
