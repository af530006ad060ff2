"""Synthetic WIDER-Face-shaped training batches (SURVEY.md §8(d) input contract).

The reference's CPU data pipeline (mmdet/datasets/pipelines/transforms.py:975-1169,
formatting.py:206-249) cannot run without the WIDER images, so batches enter at the
``forward_train`` boundary (mmdet/models/detectors/yunet.py:21-51): ``img`` is a
stacked ``[N,3,H,W]`` fp32 tensor of raw 0-255 values (mean 0 / std 1,
configs/yunet_n.py:27), GT is delivered as per-image ragged lists.

All randomness comes from a CPU ``torch.Generator`` so that the CPU oracle and the
GPU path see identical bytes.
"""
import math

import torch

# faces-per-image histogram of data/widerface/labelv2/val/labelv2.txt (3,226 images),
# counts clipped to [1, 64]; entry i = number of images with i+1 faces (last = ">=64").
WIDER_VAL_FACES_HIST = [
    1122, 416, 220, 186, 159, 108, 87, 80, 66, 58, 57, 34, 52, 39, 28, 35, 31, 30, 25,
    20, 11, 18, 17, 17, 15, 9, 17, 14, 10, 8, 12, 6, 7, 5, 3, 9, 9, 4, 4, 11, 3, 0, 3, 4,
    3, 2, 3, 2, 3, 2, 4, 5, 3, 3, 1, 5, 3, 2, 1, 3, 2, 2, 2, 106]
MAX_GT = 64


class GTList(list):
    """A per-image list of GT tensors (what the reference's collate delivers,
    mmdet/datasets/pipelines/formatting.py:232-238) that also carries the same data padded to
    [N, Gmax, ...] plus the per-image counts, prepared by the data source on the CPU so the
    training step does not loop over images on the host."""
    padded = None      # [N, Gmax, ...] tensor
    counts = None      # [N] int32 tensor


def _pad(lst, tail, gmax):
    n = len(lst)
    out = torch.zeros((n, gmax) + tail, dtype=torch.float32)
    cnt = torch.zeros(n, dtype=torch.int32)
    for i, t in enumerate(lst):
        g = int(t.shape[0])
        out[i, :g] = t
        cnt[i] = g
    return out, cnt


def attach_padding(batch, max_gt=None):
    """Turn the ragged GT lists of a batch dict into GTLists with padded companions."""
    gmax = 64
    need = max([int(b.shape[0]) for b in batch['gt_bboxes']] + [1])
    while gmax < need:
        gmax *= 2
    gb = GTList(batch['gt_bboxes'])
    gb.padded, gb.counts = _pad(batch['gt_bboxes'], (4,), gmax)
    gk = GTList(batch['gt_keypointss'])
    gk.padded, gk.counts = _pad(batch['gt_keypointss'], (5, 3), gmax)
    batch['gt_bboxes'], batch['gt_keypointss'] = gb, gk
    return batch


def batch_seed(rank, it):
    return 1234 + 1000 * rank + it


def make_gt(num_imgs, height, width, gen, max_gt=MAX_GT, hist=None):
    """Ragged GT lists: boxes xyxy fp32, labels int64 (all 0), keypoints [G,5,3]."""
    hist = torch.tensor((hist or WIDER_VAL_FACES_HIST)[:max_gt], dtype=torch.float32)
    counts = torch.multinomial(hist, num_imgs, replacement=True, generator=gen) + 1
    scale = height / 320.0
    lo, hi = math.log(4.0 * scale), math.log(160.0 * scale)
    gt_bboxes, gt_labels, gt_kps = [], [], []
    for n in range(num_imgs):
        g = int(counts[n])
        u = torch.rand(g, 8, generator=gen)
        w = torch.exp(lo + (hi - lo) * u[:, 0])
        h = w * (1.0 + 0.4 * u[:, 1])
        w = torch.clamp(w, max=width - 1.0)
        h = torch.clamp(h, max=height - 1.0)
        x1 = u[:, 2] * (width - w)
        y1 = u[:, 3] * (height - h)
        boxes = torch.stack([x1, y1, x1 + w, y1 + h], dim=1).float()
        kp = torch.rand(g, 5, 2, generator=gen)
        kx = x1[:, None] + kp[..., 0] * w[:, None]
        ky = y1[:, None] + kp[..., 1] * h[:, None]
        vis = (u[:, 4] < 0.7).float()[:, None].expand(g, 5)
        kps = torch.stack([kx, ky, vis], dim=-1).float()
        gt_bboxes.append(boxes.contiguous())
        gt_labels.append(torch.zeros(g, dtype=torch.int64))
        gt_kps.append(kps.contiguous())
    return gt_bboxes, gt_labels, gt_kps


def render_faces(img, gt_bboxes, gt_kps):
    """Paint a face-like pattern over every GT box of a noise batch, in place: a bright ellipse
    inscribed in the box with dark dots at the five landmarks.  Pure noise images carry no information
    about the boxes, so nothing can be learned from them; with this pattern a few hundred SGD
    iterations give a detector whose predicted boxes overlap their GT (IoU 0.5+), i.e. SimOTA's
    dynamic_k > 1 and real conflicts -- what a trained checkpoint exercises.
    Works on any device; larger faces are painted first so that small ones stay visible."""
    n, _, h, w = img.shape
    ys = torch.arange(h, device=img.device, dtype=torch.float32).view(1, h, 1)
    xs = torch.arange(w, device=img.device, dtype=torch.float32).view(1, 1, w)
    for i in range(n):
        b = gt_bboxes[i].to(img.device, torch.float32)
        k = gt_kps[i].to(img.device, torch.float32)
        if b.numel() == 0:
            continue
        order = torch.argsort((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), descending=True)
        b, k = b[order], k[order]
        cx, cy = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
        rx, ry = (b[:, 2] - b[:, 0]) / 2, (b[:, 3] - b[:, 1]) / 2
        inside = ((xs - cx.view(-1, 1, 1)) / rx.view(-1, 1, 1)) ** 2 + \
                 ((ys - cy.view(-1, 1, 1)) / ry.view(-1, 1, 1)) ** 2 <= 1.0            # [G,h,w]
        tone = 200.0 + 40.0 * torch.linspace(0, 1, b.shape[0], device=img.device)     # distinguishable faces
        rank = torch.arange(1, b.shape[0] + 1, device=img.device, dtype=torch.float32).view(-1, 1, 1)
        last = (inside.float() * rank).amax(dim=0)        # later (smaller) faces overwrite earlier ones
        hit = last > 0
        face = tone[(last - 1).clamp(min=0).long()]
        rad = torch.clamp(rx / 5.0, min=0.8).view(-1, 1, 1, 1)
        dots = ((xs.unsqueeze(0) - k[:, :, 0].reshape(-1, 5, 1, 1)) ** 2 +
                (ys.unsqueeze(0) - k[:, :, 1].reshape(-1, 5, 1, 1)) ** 2 <= rad ** 2)  # [G,5,h,w]
        dots = (dots & (k[:, :, 2] > 0).reshape(-1, 5, 1, 1)).any(dim=1).any(dim=0)
        face = torch.where(dots & hit, torch.full_like(face, 25.0), face)
        img[i] = torch.where(hit.unsqueeze(0), face.unsqueeze(0).expand(3, h, w) * 0.85 + img[i] * 0.15, img[i])
    return img


def make_batch(num_imgs, height, width, seed, max_gt=MAX_GT, with_img=True, structured=False):
    """One synthetic batch on CPU: dict(img, img_metas, gt_bboxes, gt_labels, gt_keypointss).
    structured=True paints render_faces() patterns over the noise (trained-weights bench fixture)."""
    gen = torch.Generator().manual_seed(int(seed))
    gt_bboxes, gt_labels, gt_kps = make_gt(num_imgs, height, width, gen, max_gt)
    img = None
    if with_img:
        img = torch.rand(num_imgs, 3, height, width, generator=gen) * 255.0
        if structured:
            render_faces(img, gt_bboxes, gt_kps)
    metas = [dict(img_shape=(height, width, 3), pad_shape=(height, width, 3),
                  scale_factor=1.0, flip=False, filename=f'synthetic_{seed}_{i}')
             for i in range(num_imgs)]
    return attach_padding(dict(img=img, img_metas=metas, gt_bboxes=gt_bboxes,
                               gt_labels=gt_labels, gt_keypointss=gt_kps))


def to_device(batch, device):
    out = dict(batch)
    if batch['img'] is not None:
        out['img'] = batch['img'].to(device, non_blocking=True)
    for k in ('gt_bboxes', 'gt_labels', 'gt_keypointss'):
        moved = [t.to(device, non_blocking=True) for t in batch[k]]
        src = batch[k]
        if isinstance(src, GTList) and src.padded is not None:
            moved = GTList(moved)
            moved.padded = src.padded.to(device, non_blocking=True)
            moved.counts = src.counts.to(device, non_blocking=True)
        out[k] = moved
    return out
