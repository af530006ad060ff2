"""-m gpu: the hot path at BASELINE.json's FULL sizes (YuNet_n 320x320 bs 256, 640x640 bs 64,
YuNet_s 320x320 bs 512), checked through size-independent properties plus an oracle spot check
on a handful of images (the CPU oracle needs seconds per image batch at these sizes).

Properties used:
  * the assignment of an image depends on that image alone: a batch, its permutation, and its
    sub-batches give bit-identical gt_inds per image (integer bar: exact);
  * sum(num_pos per image) == count(gt_inds > 0), every positive prior lies inside its GT's box
    or center region is NOT required by SimOTA, but its index must be a valid GT (1..G);
  * padded GT rows beyond gt_count never matter;
  * a second run of the same step from the same state reproduces the integer outputs exactly and
    the losses to 1e-6 relative (fp64 atomics only change the summation order of BN sums).
"""
import pytest
import torch

import crafted as C
import helpers as Hh
import yunet_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
CONFIGS = [('n', 320, 256, 11), ('n', 640, 64, 12), ('s', 320, 512, 13)]


def _assign(flat, gb, gk, cnt, h, w):
    import yunet_amd.kernels as k
    sizes = C.featmap_sizes(h, w)
    gt_inds, ovl, img_stats, _ = k.assign(flat, gb, gk, cnt, sizes, [8, 16, 32])
    torch.cuda.synchronize()
    return gt_inds, ovl, img_stats


@pytest.mark.parametrize('kind,h,n,seed', CONFIGS)
def test_assignment_properties_full_batch(kind, h, n, seed):
    import yunet_amd.synthetic as S
    b = S.make_batch(n, h, h, seed, with_img=False)
    flat = C.crafted_preds(b['gt_bboxes'], b['gt_keypointss'], h, h, seed + 1)
    gb, gk, cnt = C.pad_gt(b['gt_bboxes'], b['gt_keypointss'])
    f, gbd, gkd = flat.to(DEV).contiguous(), gb.to(DEV).contiguous(), gk.to(DEV).contiguous()
    cntd = cnt.to(DEV).int().contiguous()
    gi, ovl, st = _assign(f, gbd, gkd, cntd, h, h)

    # (1) bookkeeping invariants
    pos = gi > 0
    assert torch.equal(st[:, 0].long(), pos.sum(1)), 'num_pos per image != count(gt_inds > 0)'
    assert int(gi.max()) <= int(cnt.max()) and int(gi.min()) >= 0
    assert bool((gi <= cntd[:, None]).all()), 'a prior is assigned to a padded (non-existent) GT'
    assert bool((ovl[pos] >= 0).all()) and bool((ovl[~pos] == -1e5).all())   # (dynamic_k >= 1 can pick an IoU-0 prior)

    # (2) permutation: per-image results move with the image, bit for bit
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
    pd = perm.to(DEV)
    gi_p, ovl_p, st_p = _assign(f[pd].contiguous(), gbd[pd].contiguous(), gkd[pd].contiguous(),
                                cntd[pd].contiguous(), h, h)
    assert torch.equal(gi_p, gi[pd]) and torch.equal(ovl_p, ovl[pd]) and torch.equal(st_p, st[pd])

    # (3) sub-batches (ragged split) reproduce the rows of the full batch
    for lo, hi in ((0, 1), (1, 8), (n // 2 - 3, n // 2 + 4), (n - 5, n)):
        gi_s, _, _ = _assign(f[lo:hi].contiguous(), gbd[lo:hi].contiguous(), gkd[lo:hi].contiguous(),
                             cntd[lo:hi].contiguous(), h, h)
        assert torch.equal(gi_s, gi[lo:hi])

    # (4) garbage in the padded GT rows is ignored
    gb2, gk2 = gbd.clone(), gkd.clone()
    mask = torch.arange(gb.shape[1], device=DEV)[None, :] >= cntd[:, None]
    gb2[mask] = torch.tensor([3.0, 3.0, 150.0, 150.0], device=DEV)
    gk2[mask] = 77.0
    gi_g, _, _ = _assign(f, gb2, gk2, cntd, h, h)
    assert torch.equal(gi_g, gi)

    # (5) oracle spot check on 6 images spread over the batch (exact, modulo fp32 near-ties)
    sizes = C.featmap_sizes(h, h)
    arch = O.yunet_arch(kind)
    idx = [0, 1, n // 3, n // 2, n - 2, n - 1]
    sub = torch.tensor(idx)
    _, oaux = O.loss_step(flat[sub], [b['gt_bboxes'][i] for i in idx], [b['gt_labels'][i] for i in idx],
                          [b['gt_keypointss'][i] for i in idx], sizes, arch)
    ref = oaux['gt_inds'].int()
    got = gi.cpu()[sub]
    bad = [i for j, i in enumerate(idx) if not torch.equal(got[j], ref[j])]
    unexplained = [i for i in bad if not Hh.image_near_tie(flat[i], b['gt_bboxes'][i], sizes)]
    assert not unexplained, f'assignment mismatch not explained by an fp32 near-tie: images {unexplained}'
    assert len(bad) <= 1


@pytest.mark.parametrize('kind,h,n,seed', CONFIGS)
def test_full_step_runs_and_repeats(kind, h, n, seed):
    """One whole training step (fwd + SimOTA + losses + bwd + SGD) at the full configuration:
    finite, consistent, and repeatable from the same state."""
    import yunet_amd
    import yunet_amd.synthetic as S
    from yunet_amd.optim import FusedSGD

    def run():
        torch.manual_seed(0)
        cfg = yunet_amd.Config.fromfile(f'configs/yunet_{kind}.py')
        model = yunet_amd.build_detector(cfg.model).to(DEV).train()
        opt = FusedSGD(model, lr=0.01, momentum=0.9, weight_decay=0.0005)
        batch = S.to_device(S.make_batch(n, h, h, seed), DEV)
        out = model.train_step(batch, opt)
        opt.zero_grad()
        out['loss'].backward()
        plan = model.engine.plan
        gt_inds = plan.gt_inds.clone()
        grad = model.engine.params.grad.clone()
        opt.step()
        torch.cuda.synchronize()
        logs = {k: float(v) for k, v in out['log_vars'].items()}
        return logs, gt_inds, grad, model.engine.params.data.clone()

    logs1, gi1, g1, p1 = run()
    logs2, gi2, g2, p2 = run()
    for k, v in logs1.items():
        assert v == v and abs(v) < 1e6, (k, v)                       # finite
        assert abs(v - logs2[k]) <= 1e-6 * max(1.0, abs(v)), (k, v, logs2[k])
    assert abs(logs1['loss'] - sum(logs1[k] for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps'))) \
        <= 1e-5 * max(1.0, abs(logs1['loss']))
    assert bool(torch.isfinite(g1).all()) and bool(torch.isfinite(p1).all())
    assert float(g1.abs().max()) > 0
    assert torch.equal(gi1, gi2), 'integer assignment is not reproducible'
    assert int((gi1 > 0).sum()) > 0
    assert float((g1 - g2).abs().max()) <= 1e-5 * float(g1.abs().max())
    assert float((p1 - p2).abs().max()) <= 1e-6 * float(p1.abs().max())
