#!/usr/bin/env python
"""What the SimOTA walk sees at the bench batch (trained fixture, structured faces): per image G, V and the share of
(valid prior, GT) pairs that are 'candidates' (box overlaps the GT, or prior in its box-and-centre region)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import yunet_amd
import yunet_amd.synthetic as S
from yunet_amd.optim import FusedSGD
dev = torch.device('cuda', 0)
cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'yunet_n.py'))
torch.manual_seed(0)
model = yunet_amd.build_detector(cfg.model).to(dev); model.train()
model.load_state_dict(torch.load(os.path.join(ROOT, 'tests', 'golden', 'yunet_n_synth_trained.pth'), map_location='cpu', weights_only=False)['state_dict'], strict=True)
opt = FusedSGD(model, lr=1e-5, momentum=0.9, weight_decay=5e-4)
b = S.make_batch(256, 320, 320, S.batch_seed(0, 0), with_img=False)
gen = torch.Generator(device=dev).manual_seed(S.batch_seed(0, 0))
img = torch.rand(256, 3, 320, 320, generator=gen, device=dev) * 255.0
b['img'] = S.render_faces(img, b['gt_bboxes'], b['gt_keypointss'])
b = S.to_device(b, dev)
out = model.train_step(b, opt)
torch.cuda.synchronize()
eng = model.engine
plan = next(iter(eng.plans.values())) if hasattr(eng, 'plans') else eng.plan
flat = plan.flat.clone()
N, P, _ = flat.shape
strides = [8, 16, 32]
pri = []
for s in strides:
    h = 320 // s
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(h), indexing='ij')
    pri.append(torch.stack([xs.flatten() * s, ys.flatten() * s, torch.full((h * h,), s)], 1).float())
pri = torch.cat(pri).to(dev)
px, py, s = pri[:, 0], pri[:, 1], pri[:, 2]
cx, cy = px + s * 0.5, py + s * 0.5
rows = []
tot_pairs = tot_cand = 0
for n in range(N):
    gt = b['gt_bboxes'][n]
    G = gt.shape[0]
    f = flat[n]
    l = cx[:, None] - gt[None, :, 0]; t = cy[:, None] - gt[None, :, 1]; r = gt[None, :, 2] - cx[:, None]; bb = gt[None, :, 3] - cy[:, None]
    inbox = torch.stack([l, t, r, bb], 0).min(0).values > 0
    gcx = (gt[:, 0] + gt[:, 2]) / 2; gcy = (gt[:, 1] + gt[:, 3]) / 2; rs = 2.5 * s
    incen = torch.stack([cx[:, None] - (gcx[None] - rs[:, None]), cy[:, None] - (gcy[None] - rs[:, None]),
                         (gcx[None] + rs[:, None]) - cx[:, None], (gcy[None] + rs[:, None]) - cy[:, None]], 0).min(0).values > 0
    valid = (inbox | incen).any(1)
    bx = f[:, 1] * s + px; by = f[:, 2] * s + py; bw = torch.exp(f[:, 3]) * s; bh = torch.exp(f[:, 4]) * s
    x1, y1, x2, y2 = bx - bw / 2, by - bh / 2, bx + bw / 2, by + bh / 2
    w = (torch.minimum(x2[:, None], gt[None, :, 2]) - torch.maximum(x1[:, None], gt[None, :, 0])).clamp(min=0)
    h = (torch.minimum(y2[:, None], gt[None, :, 3]) - torch.maximum(y1[:, None], gt[None, :, 1])).clamp(min=0)
    cand = ((w * h > 0) | (inbox & incen))[valid]
    V = int(valid.sum())
    rows.append((G, V, int(cand.sum()), int(cand.sum(0).max()) if G else 0))
    tot_pairs += V * G; tot_cand += int(cand.sum())
rows.sort()
print('images', N, 'pairs', tot_pairs, 'candidates', tot_cand, 'share', round(tot_cand / tot_pairs, 3))
print('G histogram (G: images):', {g: sum(1 for r in rows if r[0] == g) for g in sorted({r[0] for r in rows})})
print('heaviest images (G, V, candidate pairs, max candidates of one GT):', rows[-12:])
print('mean V', sum(r[1] for r in rows) / N, 'sum_G', sum(r[0] for r in rows))
