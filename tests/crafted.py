"""Crafted head predictions for loss-step tests.

A randomly initialised (or noise-fed) network predicts boxes with IoU ~ 0 against
every GT, which pins SimOTA's dynamic_k at 1 and never exercises the k > 1 /
conflict-resolution paths.  These helpers synthesise a
``[N, P, 16]`` prediction tensor whose decoded boxes near each GT are noisy copies
of that GT, so IoUs spread over 0.2..0.9.
"""
import math

import torch


def featmap_sizes(height, width, strides=(8, 16, 32)):
    return [(height // s, width // s) for s in strides]


def priors_xy_stride(height, width, strides=(8, 16, 32)):
    out = []
    for s in strides:
        h, w = height // s, width // s
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
        out.append(torch.stack([xs.reshape(-1) * s, ys.reshape(-1) * s,
                                torch.full((h * w,), s)], dim=-1).float())
    return torch.cat(out)


def crafted_preds(gt_bboxes, gt_kpss, height, width, seed, strides=(8, 16, 32),
                  box_noise=0.25, logit_mean=0.5):
    """[N, P, 16] fp32: cls | dx dy dw dh | obj | 10 kps."""
    gen = torch.Generator().manual_seed(int(seed))
    pri = priors_xy_stride(height, width, strides)
    P = pri.shape[0]
    N = len(gt_bboxes)
    flat = torch.randn(N, P, 16, generator=gen) * 0.5
    flat[..., 0] -= 2.0
    flat[..., 5] -= 2.0
    px, py, ps = pri[:, 0], pri[:, 1], pri[:, 2]
    for n in range(N):
        gb = gt_bboxes[n].float()
        gk = gt_kpss[n].float()
        for g in range(gb.shape[0]):
            x1, y1, x2, y2 = [float(v) for v in gb[g]]
            gcx, gcy = (x1 + x2) / 2, (y1 + y2) / 2
            gw, gh = max(x2 - x1, 1e-3), max(y2 - y1, 1e-3)
            cx, cy = px + 0.5 * ps, py + 0.5 * ps
            near = ((cx - gcx).abs() < torch.clamp(0.6 * torch.tensor(gw), min=1.0) + 1.6 * ps) & \
                   ((cy - gcy).abs() < torch.clamp(0.6 * torch.tensor(gh), min=1.0) + 1.6 * ps)
            idx = torch.nonzero(near).squeeze(1)
            if idx.numel() == 0:
                continue
            k = idx.numel()
            nz = torch.randn(k, 16, generator=gen)
            s = ps[idx]
            flat[n, idx, 1] = (gcx - px[idx]) / s + box_noise * nz[:, 1] * min(1.0, gw / 16)
            flat[n, idx, 2] = (gcy - py[idx]) / s + box_noise * nz[:, 2] * min(1.0, gh / 16)
            flat[n, idx, 3] = torch.log(gw / s) + box_noise * nz[:, 3]
            flat[n, idx, 4] = torch.log(gh / s) + box_noise * nz[:, 4]
            flat[n, idx, 0] = logit_mean + nz[:, 0]
            flat[n, idx, 5] = logit_mean + nz[:, 5]
            enc = (gk[g, :, :2].reshape(1, 10) - torch.stack([px[idx], py[idx]], 1).repeat(1, 5)) \
                / s[:, None]
            flat[n, idx, 6:] = enc + 0.2 * nz[:, 6:]
    return flat.contiguous()


def pad_gt(gt_bboxes, gt_kpss, gmax=None):
    """Ragged lists -> padded [N,Gmax,4], [N,Gmax,5,3], counts[N] (int32)."""
    n = len(gt_bboxes)
    gmax = gmax or max(1, max(int(b.shape[0]) for b in gt_bboxes))
    boxes = torch.zeros(n, gmax, 4)
    kps = torch.zeros(n, gmax, 5, 3)
    cnt = torch.zeros(n, dtype=torch.int32)
    for i in range(n):
        g = int(gt_bboxes[i].shape[0])
        boxes[i, :g] = gt_bboxes[i]
        kps[i, :g] = gt_kpss[i]
        cnt[i] = g
    return boxes, kps, cnt


def unpad_gt(boxes, kps, cnt):
    gb = [boxes[i, :int(cnt[i])].clone() for i in range(boxes.shape[0])]
    gk = [kps[i, :int(cnt[i])].clone() for i in range(boxes.shape[0])]
    gl = [torch.zeros(int(cnt[i]), dtype=torch.int64) for i in range(boxes.shape[0])]
    return gb, gl, gk


def crowded_gt(counts, h, w, seed):
    """Ragged GT with PRESCRIBED face counts (the labelv2 reader feeds real WIDER lists: p99 178,
    max 709 faces per image), same box / landmark distributions as synthetic.make_gt."""
    import math
    gen = torch.Generator().manual_seed(seed)
    lo, hi = math.log(4.0 * h / 320.0), math.log(160.0 * h / 320.0)
    gb, gl, gk = [], [], []
    for g in counts:
        u = torch.rand(g, 8, generator=gen)
        bw = torch.exp(lo + (hi - lo) * u[:, 0] * (0.55 if g > 64 else 1.0))   # crowds are small faces
        bh = bw * (1.0 + 0.4 * u[:, 1])
        bw, bh = bw.clamp(max=w - 1.0), bh.clamp(max=h - 1.0)
        x1, y1 = u[:, 2] * (w - bw), u[:, 3] * (h - bh)
        gb.append(torch.stack([x1, y1, x1 + bw, y1 + bh], 1).float().contiguous())
        kp = torch.rand(g, 5, 2, generator=gen)
        vis = (u[:, 4] < 0.7).float()[:, None].expand(g, 5)
        gk.append(torch.stack([x1[:, None] + kp[..., 0] * bw[:, None], y1[:, None] + kp[..., 1] * bh[:, None], vis],
                              -1).float().contiguous())
        gl.append(torch.zeros(g, dtype=torch.int64))
    return gb, gl, gk
