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


def make_batch(num_imgs, height, width, seed, max_gt=MAX_GT, with_img=True):
    """One synthetic batch on CPU: dict(img, img_metas, gt_bboxes, gt_labels, gt_keypointss)."""
    gen = torch.Generator().manual_seed(int(seed))
    gt_bboxes, gt_labels, gt_kps = make_gt(num_imgs, height, width, gen, max_gt)
    img = None
    if with_img:
        img = torch.rand(num_imgs, 3, height, width, generator=gen) * 255.0
    metas = [dict(img_shape=(height, width, 3), pad_shape=(height, width, 3),
                  scale_factor=1.0, flip=False, filename=f'synthetic_{seed}_{i}')
             for i in range(num_imgs)]
    return dict(img=img, img_metas=metas, gt_bboxes=gt_bboxes, gt_labels=gt_labels,
                gt_keypointss=gt_kps)


def to_device(batch, device):
    out = dict(batch)
    if batch['img'] is not None:
        out['img'] = batch['img'].to(device, non_blocking=True)
    for k in ('gt_bboxes', 'gt_labels', 'gt_keypointss'):
        out[k] = [t.to(device, non_blocking=True) for t in batch[k]]
    return out
