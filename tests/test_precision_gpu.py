"""-m gpu: what the two-way bf16 split of the 64 -> 64 BACKWARD GEMMs costs.

The forward 64-channel units multiply through an exact three-way bf16 split (fp32-accurate, 2.4e-7).  The backward
64 -> 64 units (dp_bwd64, csrc/conv_bwd.hip) form p, dW1 and da as hi*hi + hi*lo + lo*hi -- every operand carries 16
significand bits, a product is good to ~2^-17.  The dispatcher option `bwd_fp32mma` runs the same units on the exact-fp32
matrix instruction (bench.py's `exact_fp32_bwd` line).  This file PROVES the bound instead of asserting it in prose:

  * element level, at the bench's batch (256 images, 80 x 80 and 40 x 40 maps): a per-element relative-error histogram of
    dx and dW1 of the split kernel against the exact-fp32 kernel (which test_dp_bwd_exact_fp32mma pins to fp64 at 2e-5);
  * trajectory level: 50 SGD iterations of YuNet_n 320 x 320 bs 32 from the trained fixture on both paths, and -- the
    yardstick -- on the exact path from parameters perturbed in the LAST fp32 bit.  Training is a chaotic map (SimOTA
    assignments flip on 1e-7 score differences); the statement is that the split path diverges from the exact one no
    faster than fp32 rounding noise itself does.

The measured numbers are printed (pytest -s) and land in profiles/r05_precision.json via tools/profile_round.sh.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAINED = os.path.join(ROOT, 'tests', 'golden', 'yunet_n_synth_trained.pth')


def _emit(tag, rec):
    print(f'[precision] {tag}: ' + json.dumps(rec))
    out = os.environ.get('YUNET_PRECISION_JSON')
    if out:
        try:
            cur = json.load(open(out)) if os.path.exists(out) else {}
        except Exception:
            cur = {}
        cur[tag] = rec
        json.dump(cur, open(out, 'w'), indent=1)


def _hist(got, ref):
    """Per-element relative error |got - ref| / max(|ref|, 1e-3 rms(ref)) -> quantiles and tail fractions."""
    got, ref = got.double().flatten(), ref.double().flatten()
    rms = float(ref.pow(2).mean().sqrt())
    rel = (got - ref).abs() / ref.abs().clamp_min(1e-3 * rms)
    # torch.quantile is capped at 16 M elements: a strided sample of the big tensors
    smp = rel[:: max(1, rel.numel() // 4_000_000)]
    q = torch.quantile(smp, torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64, device=smp.device))
    return dict(n=int(rel.numel()), rms_ref=rms, median=float(q[0]), p90=float(q[1]), p99=float(q[2]), p999=float(q[3]),
                max=float(rel.max()), frac_gt_1e4=float((rel > 1e-4).double().mean()),
                frac_gt_1e3=float((rel > 1e-3).double().mean()),
                max_over_tensor_max=float((got - ref).abs().max() / ref.abs().max()))


@pytest.mark.parametrize('n,h,w', [(256, 80, 80), (256, 40, 40)])
def test_split_bf16_backward_error_histogram_at_bench_batch(n, h, w):
    import yunet_amd._lib as L
    import yunet_amd.kernels as k
    g = torch.Generator(device=DEV).manual_seed(h)
    c = 64
    x = torch.randn(n, h, w, c, generator=g, device=DEV) * 2 + 0.5
    z = torch.randn(n, h, w, c, generator=g, device=DEV) * 1.5
    dy = torch.randn(n, h, w, c, generator=g, device=DEV) * (torch.rand(n, h, w, c, generator=g, device=DEV) > 0.5)
    w_pw = torch.randn(c, c, generator=g, device=DEV) * 0.2
    b_pw = torch.randn(c, generator=g, device=DEV) * 0.1
    w_dw = torch.randn(c, 9, generator=g, device=DEV) * 0.3
    b_dw = torch.randn(c, generator=g, device=DEV) * 0.1

    def stats(t):
        t2 = t.double().reshape(-1, c)
        return torch.cat([t2.sum(0), (t2 * t2).sum(0)]).contiguous()

    def run(exact):
        prev = L.set_option('bwd_fp32mma', 1 if exact else 0)
        try:
            in_bn = k.BN(stats(x), torch.rand(c, device=DEV) * 0 + 1.0, torch.zeros(c, device=DEV) + 0.1, n * h * w,
                         bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV))
            bst = torch.cat([dy.double().reshape(-1, c).sum(0), torch.zeros(c, dtype=torch.float64, device=DEV)])
            out_bn = k.BN(stats(z), torch.ones(c, device=DEV), torch.zeros(c, device=DEV), n * h * w, bstats=bst.contiguous())
            dx, dw1, db1, dw2, _ = k.dp_bwd(x, w_pw, b_pw, w_dw, b_dw, z, dy, in_bn, out_bn)
            torch.cuda.synchronize()
            return dx, dw1.reshape(c, c).clone(), dw2.reshape(c, 9).clone(), in_bn.bstats.clone()
        finally:
            L.set_option('bwd_fp32mma', prev)

    dx_e, dw1_e, dw2_e, bs_e = run(True)
    dx_s, dw1_s, dw2_s, bs_s = run(False)
    rec = dict(dx=_hist(dx_s, dx_e), dW1=_hist(dw1_s, dw1_e), dW2=_hist(dw2_s, dw2_e), bn_sums=_hist(bs_s, bs_e))
    _emit(f'hist_{n}x{h}x{w}', rec)
    # The stated bound: a product is good to 2^-17 = 7.6e-6 of |a||b|.  Measured (MI355X, round 5,
    # profiles/r05_precision.json): against the TENSOR's maximum every output is within 9e-6 (dx 7.2e-6, dW1 7.0e-6,
    # dW2 5.2e-6, BN sums 8.8e-6); per element against the element's OWN magnitude (floor 1e-3 rms) the median is 5e-7
    # (dx: 64-term sums average the error down) / 5.7e-6 (dW1, dW2: heavily cancelling sums over N*H*W pixels, the
    # error stays at the size of one product's), the 99th percentile 1.5e-4 / 3.4e-4 -- elements that are themselves
    # 100x below the tensor's rms.  Bars = measured x ~2:
    assert rec['dx']['median'] <= 2e-6 and rec['dx']['p99'] <= 3e-4 and rec['dx']['max_over_tensor_max'] <= 1.5e-5
    assert rec['dW1']['median'] <= 1.2e-5 and rec['dW1']['p99'] <= 8e-4 and rec['dW1']['max_over_tensor_max'] <= 1.5e-5
    assert rec['dW2']['max_over_tensor_max'] <= 1.5e-5 and rec['bn_sums']['max_over_tensor_max'] <= 2e-5


def _batches(iters, bs=32):
    """The 50 structured batches of the trajectory test, rendered once and kept on the device (the three runs see the same data)."""
    import yunet_amd.synthetic as S
    return [S.to_device(S.make_batch(bs, 320, 320, 91_000 + it, structured=True), DEV) for it in range(iters)]


def _train(path, batches, perturb=False, lr=1e-3):
    import yunet_amd
    import yunet_amd._lib as L
    import yunet_amd.synthetic as S
    from yunet_amd.optim import FusedSGD
    prev = L.set_option('bwd_fp32mma', 1 if path == 'exact' else 0)
    try:
        cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'yunet_n.py'))
        model = yunet_amd.build_detector(cfg.model)
        sd = torch.load(TRAINED, map_location='cpu', weights_only=False)['state_dict']
        if perturb:
            # the fp32-noise twin: every parameter moved by +-1 unit in its last place (relative 2^-23 .. 2^-24)
            gen = torch.Generator().manual_seed(99)
            sd = {k_: (torch.nextafter(v, v + torch.where(torch.rand(v.shape, generator=gen) < 0.5, -1.0, 1.0))
                       if v.is_floating_point() and 'running' not in k_ else v) for k_, v in sd.items()}
        model.load_state_dict(sd, strict=True)
        model.to(DEV).train()
        opt = FusedSGD(model, lr=lr, momentum=0.9, weight_decay=5e-4)
        theta0 = model.engine.params.data.clone() if model.engine is not None else None
        losses = []
        for b in batches:
            out = model.train_step(b, opt)
            if theta0 is None:
                theta0 = model.engine.params.data.clone()
            opt.zero_grad()
            out['loss'].backward()
            opt.step()
            losses.append(float(out['log_vars']['loss']))
        torch.cuda.synchronize()
        return theta0.double(), model.engine.params.data.clone().double(), losses
    finally:
        L.set_option('bwd_fp32mma', prev)


def test_split_bf16_backward_trajectory_50_iterations_vs_exact_fp32():
    """50 SGD iterations (momentum 0.9, lr 1e-3, weight decay 5e-4) of YuNet_n 320 x 320 bs 32 from the trained fixture:
    default (split-bf16 64 -> 64 backward) vs bwd_fp32mma = 1, with the exact path restarted from last-bit-perturbed
    parameters as the yardstick.  Measured (round 5, profiles/r05_precision.json): after 50 iterations the parameters have
    travelled 0.134 (l2); the split path ends 0.00669 away from the exact path = 5.0 % of the distance travelled, the
    exact path restarted one unit-in-the-last-place away ends 0.00637 away = 4.7 %: SimOTA's discrete assignment
    amplifies ANY perturbation to that level within 50 iterations, and the split-bf16 gradients are not distinguishable
    from fp32 rounding noise (ratio 1.05).  Largest loss difference at any iteration: 0.16 % (split) vs 0.20 % (twin).
    A second run measured 5.0 % vs 5.9 % (ratio 0.85).
    Stated bound: distance(split, exact) <= 3 x distance(twin, exact) + 0.2 % of the travel, <= 15 % of the travel;
    loss curves within 2 % at every iteration."""
    iters = 50
    batches = _batches(iters)
    t0, th_e, l_e = _train('exact', batches)
    _, th_s, l_s = _train('split', batches)
    _, th_n, l_n = _train('exact', batches, perturb=True)
    travel = float((th_e - t0).norm())
    d_split = float((th_s - th_e).norm())
    d_noise = float((th_n - th_e).norm())
    rel_loss_split = max(abs(a - b) / abs(b) for a, b in zip(l_s, l_e))
    rel_loss_noise = max(abs(a - b) / abs(b) for a, b in zip(l_n, l_e))
    rec = dict(iters=iters, travel=travel, d_split=d_split, d_noise=d_noise, d_split_over_travel=d_split / travel,
               d_noise_over_travel=d_noise / travel, max_rel_loss_diff_split=rel_loss_split,
               max_rel_loss_diff_noise=rel_loss_noise, loss_first=l_e[0], loss_last=l_e[-1],
               loss_last_split=l_s[-1], loss_last_noise=l_n[-1])
    _emit('trajectory_n320_bs32', rec)
    assert all(v == v for v in l_s + l_e + l_n)
    # (two runs of this test measured d_split / d_noise = 1.05 and 0.85: the fp64 atomics of the BatchNorm sums order
    # differently from run to run and every trajectory is one sample of the same chaotic spread -- hence the wide bars)
    assert d_split <= 3.0 * d_noise + 2e-3 * travel, rec
    assert d_split <= 0.15 * travel, rec
    assert rel_loss_split <= 2e-2, rec
