import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import torch
import crafted as C, helpers as Hh, yunet_oracle as O
import test_loss_step_gpu as T
h = 640; counts = [709, 65, 178, 12]; gmax = 1024; seed = 202
gb, gl, gk = T._crowded_gt(counts, h, h, seed)
flat = C.crafted_preds(gb, gk, h, h, seed + 1)
fl = flat.clone().requires_grad_(True)
ol, oaux = O.loss_step(fl, gb, gl, gk, C.featmap_sizes(h, h), O.yunet_arch('n'))
sum(ol.values()).backward()
gbp, gkp, cnt = C.pad_gt(gb, gk, gmax=gmax)
gt_inds, ovl, img_stats, labels, losses, dflat, norm = T.run_hip(flat, gbp, gkp, cnt, h, h, 'EIoULoss')
d = (dflat - fl.grad).abs()
print('max', float(d.max()), 'scale', float(fl.grad.abs().max()))
for ch in range(16):
    print(ch, float(d[..., ch].max()), float(fl.grad[..., ch].abs().max()))
idx = torch.nonzero(d > 1e-6)
print('n bad', idx.shape[0])
for r in idx[:20]:
    n, p, c = [int(v) for v in r]
    print(n, p, c, 'gt', int(gt_inds[n, p]), 'ref gt', int(oaux['gt_inds'][n, p]), 'hip', float(dflat[n, p, c]), 'ref', float(fl.grad[n, p, c]), 'ovl', float(ovl[n,p]))
print('losses', losses, {k: float(v) for k, v in ol.items()}, 'norm', norm)
