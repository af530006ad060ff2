"""-m gpu: every HIP kernel of libyunet_hip.so, called through the C ABI, against a torch
fp64 CPU restatement of the same op (floating-point kernels) -- tolerance stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def K():
    import yunet_amd.kernels as k
    return k


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def stats_of(z_nhwc):
    z = z_nhwc.double().reshape(-1, z_nhwc.shape[-1])
    return torch.cat([z.sum(0), (z * z).sum(0)]).contiguous()


def bn_ref(x_nchw, gamma, beta, eps=1e-5):
    """train-mode BN in fp64 -> (bn output, xhat)"""
    mean = x_nchw.mean(dim=(0, 2, 3), keepdim=True)
    var = x_nchw.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    xhat = (x_nchw - mean) / torch.sqrt(var + eps)
    return xhat * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1), xhat


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_err_ch(a, b, dim=1):
    """worst per-channel error: max |a - b| over a channel / max |b| over THAT channel (the max-norm of rel_err hides
    errors in channels whose values are small next to the tensor's largest)"""
    a, b = a.double().cpu(), b.double().cpu()
    dims = [d for d in range(a.dim()) if d != dim]
    num = (a - b).abs().amax(dim=dims)
    den = b.abs().amax(dim=dims)
    keep = den > 0
    return float((num[keep] / den[keep]).max()) if bool(keep.any()) else 0.0


def mk_unit(cin, cout, g):
    w_pw = torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / (cin + cout)) ** 0.5
    b_pw = torch.randn(cout, generator=g) * 0.1
    w_dw = torch.randn(cout, 1, 3, 3, generator=g) * 0.3
    b_dw = torch.randn(cout, generator=g) * 0.1
    return w_pw, b_pw, w_dw, b_dw


SHAPES = [(2, 10, 10), (3, 20, 40), (1, 24, 16), (2, 5, 5),
          (5, 10, 10), (4, 20, 20), (19, 10, 10), (6, 7, 13)]     # N >= 4 and H, W <= 20: packed-canvas tiling
# Shapes that select the remaining template instances / code paths of the kernels the bench runs:
#   (2, 64, 96), (1, 160, 160): W >= 64 and H >= 32 -> the 16x32 "big" tile of the 16->16 units;
#   (40, 80, 80): 2000 tiles of 8x16 > the persistent grid (256 workgroups backward, <= 1024
#   forward) -> every workgroup walks >= 3 tiles through the issue(t + gridDim.x) prefetch path;
#   (70, 20, 20): packed canvas with 5 canvas rows -> 30 x 10 = 300 packed tiles > 256 workgroups;
#   (40, 40, 40): width a multiple of 8 but not of 16 -> dp_bwd64 on 8 x 8 tiles with 4 waves, two workgroups per
#   CU (round 3): 1000 tiles > the 512-workgroup grid, the one-float-per-thread remainder of its halo.
#   (3, 40, 72), (5, 50, 70): big-tile class with ragged strips / bands of the wave-streaming 16 -> 16 backward
#   (conv_bwd16.hip: 28-column strips -> 28 + 28 + 16 and 28 + 28 + 14 columns; band heights that do not divide H)
BIG_SHAPES = {(16, 16): [(2, 64, 96), (1, 160, 160), (9, 96, 128), (3, 40, 72), (5, 50, 70)],
              (64, 64): [(40, 80, 80), (70, 20, 20), (40, 40, 40)],
              (16, 64): [(40, 80, 80)],
              (64, 16): [(40, 80, 80), (70, 20, 20)]}


def shapes_for(cin, cout):
    return SHAPES + BIG_SHAPES.get((cin, cout), [])


CHANNELS = [(16, 16), (16, 32), (16, 64), (32, 32), (32, 64), (64, 64), (64, 16)]


@pytest.mark.parametrize('cin,cout', CHANNELS)
@pytest.mark.parametrize('with_in_bn', [False, True])
def test_dp_fwd(cin, cout, with_in_bn):
    k = K()
    g = torch.Generator().manual_seed(cin * 100 + cout)
    for (n, h, w) in shapes_for(cin, cout):
        x = torch.randn(n, cin, h, w, generator=g) * 3 + 1.5
        w_pw, b_pw, w_dw, b_dw = mk_unit(cin, cout, g)
        gamma = torch.rand(cin, generator=g) + 0.5
        beta = torch.randn(cin, generator=g) * 0.2
        xd = x.double()
        a = F.relu(bn_ref(xd, gamma.double(), beta.double())[0]) if with_in_bn else xd
        p = F.conv2d(a, w_pw.double(), b_pw.double())
        zr = F.conv2d(p, w_dw.double(), b_dw.double(), padding=1, groups=cout)
        xg = nhwc(x).to(DEV)
        in_bn = None
        if with_in_bn:
            in_bn = k.BN(stats_of(xg).to(DEV), gamma.to(DEV), beta.to(DEV), n * h * w)
        out_stats = torch.zeros(2 * cout, dtype=torch.float64, device=DEV)
        out_bn = k.BN(out_stats, torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV),
                      n * h * w) if cout != 16 or cin != 64 else None
        z = k.dp_fwd(xg, w_pw.to(DEV).view(cout, cin).contiguous(), b_pw.to(DEV),
                     w_dw.to(DEV).view(cout, 9).contiguous(), b_dw.to(DEV), in_bn, out_bn)
        torch.cuda.synchronize()
        assert rel_err(nchw(z.cpu()), zr) < 2e-5, (n, h, w)
        assert rel_err_ch(nchw(z.cpu()), zr) < 8e-5, ('per output channel', n, h, w, rel_err_ch(nchw(z.cpu()), zr))
        if out_bn is not None:
            sr = stats_of(nhwc(zr))
            assert rel_err(out_stats, sr) < 2e-5


@pytest.fixture
def exact_fp32_backward():
    """Dispatcher option bwd_fp32mma = 1 (include/yunet_hip.h): every backward GEMM of the 64 -> 64 units on the
    exact-fp32 matrix instruction -- the variant bench.py times as `exact_fp32_bwd` (that number
    came from a dispatch no test drove)."""
    import yunet_amd._lib as L
    prev = L.set_option('bwd_fp32mma', 1)
    yield
    L.set_option('bwd_fp32mma', prev)


@pytest.mark.parametrize('mode', ['bn_bn', 'id_bn', 'bn_nobn'])
def test_dp_bwd_exact_fp32mma(mode, exact_fp32_backward):
    """test_dp_bwd[64-64-*] with bwd_fp32mma = 1: dp_bwd_kernel<64,64,8,16,PACKED,GEMM=0> on the plain 80 x 80 /
    40 x 40 maps and the packed 20 x 20 canvas (BIG_SHAPES walks > 256 tiles per workgroup grid).  Same fp64 yardstick;
    the bar is the forward kernels' 2e-5 (the split-bf16 default gets 5e-5)."""
    _dp_bwd_case(64, 64, mode, tol=2e-5)


def test_fused_pooling_exact_fp32mma(exact_fp32_backward):
    """The pooled-dy 64 -> 64 instance (80 x 80 -> 40 x 40 in the shipped nets) with bwd_fp32mma = 1:
    dp_bwd_kernel<64,64,8,16,false,0,true> (before round 5 this instance ignored the option)."""
    test_fused_pooling(64, 64, 20, 80, 80)


@pytest.mark.parametrize('cin,cout', CHANNELS)
@pytest.mark.parametrize('mode', ['bn_bn', 'id_bn', 'bn_nobn'])
def test_dp_bwd(cin, cout, mode):
    """dx / dW1 / db1 / dW2 / db2 and the producer's BN-backward sums vs fp64 autograd."""
    _dp_bwd_case(cin, cout, mode, tol=5e-5)


def _dp_bwd_case(cin, cout, mode, tol):
    k = K()
    g = torch.Generator().manual_seed(7 + cin * 100 + cout)
    in_bn_on = mode.startswith('bn')
    out_bn_on = mode.endswith('_bn')
    for (n, h, w) in shapes_for(cin, cout):
        x = (torch.randn(n, cin, h, w, generator=g) * 2 + 0.5).double()
        w_pw, b_pw, w_dw, b_dw = [t.double().requires_grad_(True) for t in mk_unit(cin, cout, g)]
        gi, bi = (torch.rand(cin, generator=g) + 0.5).double(), (torch.randn(cin, generator=g) * .2).double()
        go, bo = (torch.rand(cout, generator=g) + 0.5).double(), (torch.randn(cout, generator=g) * .2).double()
        r = torch.randn(n, cout, h, w, generator=g).double()
        if in_bn_on:
            b_in, xhat_in = bn_ref(x, gi, bi)
            b_in = b_in.detach().requires_grad_(True)
            a = F.relu(b_in)
        else:
            b_in = x.clone().requires_grad_(True)
            a = b_in
        z = F.conv2d(F.conv2d(a, w_pw, b_pw), w_dw, b_dw, padding=1, groups=cout)
        if out_bn_on:
            zb, xhat_out = bn_ref(z, go, bo)
            zb.retain_grad()
            y = F.relu(zb)
            (y * r).sum().backward()
            dy_ref = zb.grad           # grad wrt BN output, ReLU mask applied
        else:
            scale = torch.rand(cout, generator=g).double() + 0.5
            (z * r * scale.view(1, -1, 1, 1)).sum().backward()
            dy_ref = r
        # ---- kernel
        xg = nhwc(x.float()).to(DEV)
        in_bn = out_bn = None
        if in_bn_on:
            in_bn = k.BN(stats_of(xg), gi.float().to(DEV), bi.float().to(DEV), n * h * w,
                         bstats=torch.zeros(2 * cin, dtype=torch.float64, device=DEV))
        zg = nhwc(z.detach().float()).to(DEV)
        dyg = nhwc(dy_ref.float()).to(DEV)
        dy_scale = None
        if out_bn_on:
            bst = torch.cat([dy_ref.sum(dim=(0, 2, 3)), (dy_ref * xhat_out.detach()).sum(dim=(0, 2, 3))])
            out_bn = k.BN(stats_of(zg), go.float().to(DEV), bo.float().to(DEV), n * h * w,
                          bstats=bst.to(DEV).contiguous())
        else:
            dy_scale = scale.float().to(DEV)
        dx, dw1, db1, dw2, db2 = k.dp_bwd(
            xg, w_pw.detach().float().to(DEV).view(cout, cin).contiguous(), b_pw.detach().float().to(DEV),
            w_dw.detach().float().to(DEV).view(cout, 9).contiguous(), b_dw.detach().float().to(DEV),
            zg, dyg, in_bn, out_bn, dy_scale=dy_scale)
        torch.cuda.synchronize()
        assert rel_err(nchw(dx.cpu()), b_in.grad) < tol, ('dx', n, h, w)
        # per input channel; bar = 4x the max-norm bar: a channel's own maximum is up to ~4x
        # below the tensor's at these channel counts
        assert rel_err_ch(nchw(dx.cpu()), b_in.grad) < 4 * tol, ('dx per channel', n, h, w, rel_err_ch(nchw(dx.cpu()), b_in.grad))
        assert rel_err_ch(dw1.reshape(cout, cin), w_pw.grad.reshape(cout, cin), dim=0) < 4 * tol, ('dw1 per output channel', n, h, w)
        assert rel_err(dw1, w_pw.grad) < tol, ('dw1', n, h, w)
        assert rel_err(db1, b_pw.grad) < tol, ('db1', n, h, w)
        assert rel_err(dw2, w_dw.grad) < tol, ('dw2', n, h, w)
        if not out_bn_on:   # with BN the dw-bias gradient is identically zero (noise only)
            assert rel_err(db2, b_dw.grad) < tol, ('db2', n, h, w)
        else:
            assert float(db2.abs().max()) < 1e-3 * float(dy_ref.abs().sum())
        if in_bn_on:
            ref_b = torch.cat([b_in.grad.sum(dim=(0, 2, 3)), (b_in.grad * xhat_in).sum(dim=(0, 2, 3))])
            assert rel_err(in_bn.bstats, ref_b) < tol


def test_dp_bwd_accumulate():
    k = K()
    g = torch.Generator().manual_seed(3)
    n, h, w, cin, cout = 2, 12, 20, 16, 16
    x = nhwc(torch.randn(n, cin, h, w, generator=g)).to(DEV)
    w_pw, b_pw, w_dw, b_dw = [t.to(DEV) for t in mk_unit(cin, cout, g)]
    z = torch.randn(n, h, w, cout, generator=g).to(DEV)
    dy = torch.randn(n, h, w, cout, generator=g).to(DEV)
    args = (x, w_pw.view(cout, cin).contiguous(), b_pw, w_dw.view(cout, 9).contiguous(), b_dw, z, dy)
    dx0 = k.dp_bwd(*args)[0]
    base = torch.randn(n, h, w, cin, generator=g).to(DEV)
    dx1 = base.clone()
    k.dp_bwd(*args, dx=dx1, accumulate_dx=True)
    torch.cuda.synchronize()
    assert torch.allclose(dx1, base + dx0, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('n,h,w', [(2, 32, 32), (3, 64, 96), (1, 20, 36), (2, 320, 320), (5, 66, 150)])
def test_stem_fwd_bwd(n, h, w):
    k = K()
    g = torch.Generator().manual_seed(h)
    img = torch.rand(n, 3, h, w, generator=g) * 255
    wt = (torch.randn(16, 3, 3, 3, generator=g) * 0.05).double().requires_grad_(True)
    b = (torch.randn(16, generator=g) * 0.1).double().requires_grad_(True)
    gamma, beta = (torch.rand(16, generator=g) + 0.5).double(), torch.randn(16, generator=g).double() * 0.2
    z = F.conv2d(img.double(), wt, b, stride=2, padding=1)
    zb, xhat = bn_ref(z, gamma, beta)
    zb.retain_grad()
    r = torch.randn(z.shape, generator=g).double()
    (F.relu(zb) * r).sum().backward()
    stats = torch.zeros(32, dtype=torch.float64, device=DEV)
    zg = k.stem_fwd(img.to(DEV), wt.detach().float().to(DEV), b.detach().float().to(DEV), stats)
    torch.cuda.synchronize()
    assert rel_err(nchw(zg.cpu()), z.detach()) < 2e-5
    assert rel_err(stats, stats_of(nhwc(z.detach()))) < 2e-5
    dy = zb.grad
    bst = torch.cat([dy.sum(dim=(0, 2, 3)), (dy * xhat.detach()).sum(dim=(0, 2, 3))]).to(DEV)
    bn = k.BN(stats, gamma.float().to(DEV), beta.float().to(DEV), n * (h // 2) * (w // 2), bstats=bst)
    dw, db = k.stem_bwd(img.to(DEV), zg, nhwc(dy.float()).to(DEV), bn)
    torch.cuda.synchronize()
    assert rel_err(dw, wt.grad) < 1e-4
    assert float(db.abs().max()) < 1e-3 * float(dy.abs().sum())
    # the variant a training step runs: both products on the matrix cores, z recomputed from the image (yunet_stem_bwd_rz)
    dw2, db2 = k.stem_bwd(img.to(DEV), zg, nhwc(dy.float()).to(DEV), bn, wt.detach().float().to(DEV).contiguous(),
                          b.detach().float().to(DEV))
    torch.cuda.synchronize()
    assert rel_err(dw2, wt.grad) < 1e-4
    assert float(db2.abs().max()) < 1e-3 * float(dy.abs().sum())
    # and the VALU tile kernel behind the option (its aligned 16-byte patch loads want W % 4 == 0; the engine feeds
    # multiples of 32)
    if w % 4:
        return
    import yunet_amd._lib as L
    prev = L.set_option('stem_mma', 0)
    try:
        stats0 = torch.zeros(32, dtype=torch.float64, device=DEV)
        zg0 = k.stem_fwd(img.to(DEV), wt.detach().float().to(DEV), b.detach().float().to(DEV), stats0)
        torch.cuda.synchronize()
        assert rel_err(nchw(zg0.cpu()), z.detach()) < 2e-5 and rel_err(stats0, stats_of(nhwc(z.detach()))) < 2e-5
    finally:
        L.set_option('stem_mma', prev)


@pytest.mark.parametrize('c', [16, 32, 64])
def test_pool_fwd_bwd(c):
    k = K()
    g = torch.Generator().manual_seed(c)
    n, h, w = 3, 12, 20
    z = (torch.randn(n, c, h, w, generator=g) * 2).double()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).double(), torch.randn(c, generator=g).double() * .3
    zb, xhat = bn_ref(z, gamma, beta)
    zb = zb.detach().requires_grad_(True)
    out = F.max_pool2d(F.relu(zb), 2)
    r = torch.randn(out.shape, generator=g).double()
    (out * r).sum().backward()
    zg = nhwc(z.float()).to(DEV)
    bn = k.BN(stats_of(zg), gamma.float().to(DEV), beta.float().to(DEV), n * h * w,
              bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV))
    og = k.pool_fwd(zg, bn)
    dx = k.pool_bwd(zg, bn, nhwc(r.float()).to(DEV))
    torch.cuda.synchronize()
    assert rel_err(nchw(og.cpu()), out.detach()) < 2e-5
    assert rel_err(nchw(dx.cpu()), zb.grad) < 2e-5
    ref_b = torch.cat([zb.grad.sum(dim=(0, 2, 3)), (zb.grad * xhat).sum(dim=(0, 2, 3))])
    assert rel_err(bn.bstats, ref_b) < 5e-5


@pytest.mark.parametrize('ci,c,n,h,w', [(16, 16, 2, 64, 96), (16, 16, 1, 160, 160), (64, 64, 3, 40, 48),
                                        (64, 64, 20, 80, 80), (32, 64, 3, 40, 48), (64, 64, 12, 40, 40)])
def test_fused_pooling(ci, c, n, h, w):
    """unit P -> BN -> ReLU -> max_pool2d(2) -> unit Q without a pooling kernel and without a full-size
    gradient of P's output: P's forward also writes the raw window winners + their positions (some gammas are
    negative or zero: minimum / first element), Q reads them through the BN+ReLU input transform of P's
    BatchNorm, Q's backward writes the masked pooled gradient and the BN-backward sums, P's backward
    (pool_idx) expands it while staging.  Against fp64 autograd through the same graph.
    (20, 80, 80) walks the prefetch path of both kernels."""
    k = K()
    g = torch.Generator().manual_seed(c + h + ci)
    x = (torch.randn(n, ci, h, w, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    P = [t.double().requires_grad_(True) for t in mk_unit(ci, c, g)]
    Q = [t.double().requires_grad_(True) for t in mk_unit(c, c, g)]
    gp, bp = (torch.rand(c, generator=g) + 0.5).double(), (torch.randn(c, generator=g) * .3).double()
    gp[1], gp[c - 3] = -gp[1], -0.7          # falling BN: the window MINIMUM wins
    gp[5] = 0.0                              # constant channel: first element, like F.max_pool2d
    gq, bq = (torch.rand(c, generator=g) + 0.5).double(), (torch.randn(c, generator=g) * .3).double()
    zp = F.conv2d(F.conv2d(x, P[0], P[1]), P[2], P[3], padding=1, groups=c)
    zbp, xhat_p = bn_ref(zp, gp, bp)
    zbp.retain_grad()
    pooled, ref_idx = F.max_pool2d(F.relu(zbp), 2, return_indices=True)
    pooled.retain_grad()
    zq = F.conv2d(F.conv2d(pooled, Q[0], Q[1]), Q[2], Q[3], padding=1, groups=c)
    zbq, xhat_q = bn_ref(zq, gq, bq)
    zbq.retain_grad()
    r = torch.randn(zq.shape, generator=g).double()
    (F.relu(zbq) * r).sum().backward()

    def dev(t, *shape):
        t = t.detach().float().to(DEV)
        return t.view(*shape).contiguous() if shape else t
    xg = nhwc(x.detach().float()).to(DEV)
    pw = [dev(P[0], c, ci), dev(P[1]), dev(P[2], c, 9), dev(P[3])]
    qw = [dev(Q[0], c, c), dev(Q[1]), dev(Q[2], c, 9), dev(Q[3])]
    bn_p = k.BN(torch.zeros(2 * c, dtype=torch.float64, device=DEV), dev(gp), dev(bp), n * h * w,
                bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV))
    zpg, praw, idx = k.dp_fwd(xg, *pw, None, bn_p, pool=True)
    torch.cuda.synchronize()
    assert rel_err(nchw(zpg.cpu()), zp.detach()) < 2e-5
    # the raw winners, transformed, are the pooled activations
    mean, var = zp.detach().mean(dim=(0, 2, 3)), zp.detach().var(dim=(0, 2, 3), unbiased=False)
    act = F.relu((nchw(praw.cpu()).double() - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
                 * gp.view(1, -1, 1, 1) + bp.view(1, -1, 1, 1))
    assert rel_err(act, pooled.detach()) < 2e-5
    # window positions vs F.max_pool2d's indices
    ry, rx = ref_idx // w, ref_idx % w
    ref_pos = ((ry % 2) * 2 + rx % 2).to(torch.uint8)
    live = pooled.detach() > 1e-4                       # dead windows (all zero after the ReLU) carry no gradient
    live[:, 5] = True                                   # ... but the constant channel must take position 0 everywhere
    agree = (nchw(idx.cpu()) == ref_pos) | ~live
    assert float(agree.double().mean()) > 0.9999        # (fp32 vs fp64 near-ties may pick the other element)
    assert bool((nchw(idx.cpu())[:, 5] == 0).all())
    # Q forward through the ordinary BN+ReLU input transform of P's BatchNorm
    bn_q = k.BN(torch.zeros(2 * c, dtype=torch.float64, device=DEV), dev(gq), dev(bq), n * (h // 2) * (w // 2))
    zqg = k.dp_fwd(praw, *qw, bn_p, bn_q)
    torch.cuda.synchronize()
    assert rel_err(nchw(zqg.cpu()), zq.detach()) < 5e-5
    # backward
    dq = zbq.grad                                        # grad wrt Q's BN output, ReLU mask applied
    bn_q.bstats = torch.cat([dq.sum(dim=(0, 2, 3)), (dq * xhat_q.detach()).sum(dim=(0, 2, 3))]).to(DEV).contiguous()
    dpool, qdw1, qdb1, qdw2, _ = k.dp_bwd(praw, *qw, zqg, nhwc(dq.float()).to(DEV), bn_p, bn_q)
    dxp, pdw1, pdb1, pdw2, _ = k.dp_bwd(xg, *pw, zpg, dpool, None, bn_p, pool_idx=idx)
    torch.cuda.synchronize()
    tol = 5e-5                    # the bar of the plain units (test_dp_bwd); 1e-4 until round 3
    masked = pooled.grad * (pooled.detach() > 0)
    assert rel_err(nchw(dpool.cpu()), masked) < tol
    ref_b = torch.cat([zbp.grad.sum(dim=(0, 2, 3)), (zbp.grad * xhat_p.detach()).sum(dim=(0, 2, 3))])
    assert rel_err(bn_p.bstats, ref_b) < tol
    for name, got, want in (('q.dw1', qdw1, Q[0].grad), ('q.db1', qdb1, Q[1].grad), ('q.dw2', qdw2, Q[2].grad),
                            ('p.dx', nchw(dxp.cpu()), x.grad), ('p.dw1', pdw1, P[0].grad), ('p.db1', pdb1, P[1].grad),
                            ('p.dw2', pdw2, P[2].grad)):
        assert rel_err(got, want) < tol, (name, float(rel_err(got, want)))


def test_upadd_fwd_bwd():
    k = K()
    g = torch.Generator().manual_seed(11)
    n, h, w, c = 2, 12, 20, 64
    za = (torch.randn(n, c, h, w, generator=g) * 2).double()
    zb_ = (torch.randn(n, c, h // 2, w // 2, generator=g) * 2).double()
    ga, ba = (torch.rand(c, generator=g) + 0.5).double(), torch.randn(c, generator=g).double() * .3
    gb, bb = (torch.rand(c, generator=g) + 0.5).double(), torch.randn(c, generator=g).double() * .3
    ya, xha = bn_ref(za, ga, ba)
    yb, xhb = bn_ref(zb_, gb, bb)
    ya = ya.detach().requires_grad_(True)
    yb = yb.detach().requires_grad_(True)
    out = F.relu(ya) + F.interpolate(F.relu(yb), scale_factor=2., mode='nearest')
    r = torch.randn(out.shape, generator=g).double()
    (out * r).sum().backward()
    zag, zbg = nhwc(za.float()).to(DEV), nhwc(zb_.float()).to(DEV)
    bna = k.BN(stats_of(zag), ga.float().to(DEV), ba.float().to(DEV), n * h * w,
               bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV))
    bnb = k.BN(stats_of(zbg), gb.float().to(DEV), bb.float().to(DEV), n * h * w // 4,
               bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV))
    og = k.upadd_fwd(zag, bna, zbg, bnb)
    dxa, dxb = k.upadd_bwd(zag, bna, zbg, bnb, nhwc(r.float()).to(DEV))
    torch.cuda.synchronize()
    assert rel_err(nchw(og.cpu()), out.detach()) < 2e-5
    assert rel_err(nchw(dxa.cpu()), ya.grad) < 2e-5
    assert rel_err(nchw(dxb.cpu()), yb.grad) < 2e-5
    assert rel_err(bna.bstats, torch.cat([ya.grad.sum(dim=(0, 2, 3)), (ya.grad * xha).sum(dim=(0, 2, 3))])) < 5e-5
    assert rel_err(bnb.bstats, torch.cat([yb.grad.sum(dim=(0, 2, 3)), (yb.grad * xhb).sum(dim=(0, 2, 3))])) < 5e-5


@pytest.mark.parametrize('n,h,w,c', [(2, 12, 20, 64), (3, 40, 40, 64)])
def test_tap_gradient_written_once(n, h, w, c):
    """A pyramid tap y = relu(bn(z)) feeds max_pool2d (yunet_backbone.py:39-40) AND the identity branch of the TFPN
    merge (tfpn.py:39-40); autograd adds the two gradients.  upadd_bwd(skip_a) + pool_bwd(extra = the merge's gradient)
    write the tap's gradient once: against fp64 autograd through both consumers, and against the unfused kernel pair
    (dx identical bit for bit -- per element the same single addition -- the BatchNorm sums to fp64 rounding)."""
    k = K()
    g = torch.Generator().manual_seed(h + c)
    za = (torch.randn(n, c, h, w, generator=g) * 2).double()
    zb_ = (torch.randn(n, c, h // 2, w // 2, generator=g) * 2).double()
    ga, ba = (torch.rand(c, generator=g) + 0.5).double(), torch.randn(c, generator=g).double() * .3
    gb, bb = (torch.rand(c, generator=g) + 0.5).double(), torch.randn(c, generator=g).double() * .3
    ga[3] = -ga[3]
    ya, xha = bn_ref(za, ga, ba)
    yb, xhb = bn_ref(zb_, gb, bb)
    ya = ya.detach().requires_grad_(True)
    yb = yb.detach().requires_grad_(True)
    merged = F.relu(ya) + F.interpolate(F.relu(yb), scale_factor=2., mode='nearest')
    pooled = F.max_pool2d(F.relu(ya), 2)
    r_m = torch.randn(merged.shape, generator=g).double()
    r_p = torch.randn(pooled.shape, generator=g).double()
    ((merged * r_m).sum() + (pooled * r_p).sum()).backward()
    zag, zbg = nhwc(za.float()).to(DEV), nhwc(zb_.float()).to(DEV)
    dm, dp = nhwc(r_m.float()).to(DEV), nhwc(r_p.float()).to(DEV)

    def bns():
        return (k.BN(stats_of(zag), ga.float().to(DEV), ba.float().to(DEV), n * h * w,
                     bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV)),
                k.BN(stats_of(zbg), gb.float().to(DEV), bb.float().to(DEV), n * h * w // 4,
                     bstats=torch.zeros(2 * c, dtype=torch.float64, device=DEV)))
    # the unfused pair: the merge writes its share, the pool kernel re-reads z and accumulates
    bna0, bnb0 = bns()
    dxa0, dxb0 = k.upadd_bwd(zag, bna0, zbg, bnb0, dm)
    k.pool_bwd(zag, bna0, dp, dx=dxa0, accumulate=True)
    # the fused pair
    bna1, bnb1 = bns()
    none_a, dxb1 = k.upadd_bwd(zag, bna1, zbg, bnb1, dm, skip_a=True)
    assert none_a is None
    torch.cuda.synchronize()
    assert float(bna1.bstats.abs().max()) == 0.0          # the merge left the tap's sums alone
    dxa1 = torch.full(zag.shape, float('nan'), device=DEV)
    k.pool_bwd(zag, bna1, dp, dx=dxa1, extra=dm)
    torch.cuda.synchronize()
    assert torch.equal(dxa1, dxa0) and torch.equal(dxb1, dxb0)
    assert rel_err(bna1.bstats, bna0.bstats) < 1e-12 and torch.equal(bnb1.bstats, bnb0.bstats)
    assert rel_err(nchw(dxa1.cpu()), ya.grad) < 2e-5
    assert rel_err(nchw(dxb1.cpu()), yb.grad) < 2e-5
    assert rel_err(bna1.bstats, torch.cat([ya.grad.sum(dim=(0, 2, 3)), (ya.grad * xha).sum(dim=(0, 2, 3))])) < 5e-5
    # accumulate on top of an existing gradient (a third consumer that ran first)
    base = torch.randn(zag.shape, device=DEV)
    dxa2 = base.clone()
    k.pool_bwd(zag, bna1, dp, dx=dxa2, accumulate=True, extra=dm)
    torch.cuda.synchronize()
    assert torch.equal(dxa2, base + dxa1)
    # round 5: skip_a runs the dedicated coarse-gradient kernel (upadd_bwd_coarse_kernel); the general kernel behind the
    # option gives the same bits, and accumulate_b adds onto an existing coarse gradient
    import yunet_amd._lib as L
    prev = L.set_option('upadd_coarse', 0)
    try:
        bna3, bnb3 = bns()
        _, dxb3 = k.upadd_bwd(zag, bna3, zbg, bnb3, dm, skip_a=True)
        torch.cuda.synchronize()
    finally:
        L.set_option('upadd_coarse', prev)
    assert torch.equal(dxb3, dxb1) and torch.equal(bnb3.bstats, bnb1.bstats)
    baseb = torch.randn(zbg.shape, device=DEV)
    dxb4 = baseb.clone()
    bna4, bnb4 = bns()
    k.upadd_bwd(zag, bna4, zbg, bnb4, dm, dxb=dxb4, acc_b=True, skip_a=True)
    torch.cuda.synchronize()
    assert torch.equal(dxb4, baseb + dxb1)


def _replicas(v, slots, g):
    """[2C] sums -> [slots, 2C] replica blocks that add up to v (unequal shares, one of them negative)."""
    wts = torch.rand(slots, 1, generator=g, dtype=torch.float64) + 0.1
    wts = wts / wts.sum()
    wts[0] -= 0.5
    wts[1] += 0.5
    return (wts.to(v.device) * v.view(1, -1)).contiguous()


@pytest.mark.parametrize('cin,cout,n,h,w', [(64, 64, 70, 20, 20), (64, 64, 3, 40, 48), (16, 64, 2, 32, 32),
                                            (64, 16, 5, 20, 20), (16, 16, 2, 64, 96)])
def test_bn_sum_replicas(cin, cout, n, h, w):
    """YunetBN::slots: the sum blocks as several replicas (workgroup b adds into replica b % slots, readers add
    the replicas up) give the results of the single-block layout -- forward, backward, pool, upsample-add."""
    k = K()
    g = torch.Generator().manual_seed(cin + cout + h)
    S = 4
    x = nhwc(torch.randn(n, cin, h, w, generator=g) * 2 + 0.5).to(DEV)
    w_pw, b_pw, w_dw, b_dw = [t.to(DEV) for t in mk_unit(cin, cout, g)]
    w_pw, w_dw = w_pw.view(cout, cin).contiguous(), w_dw.view(cout, 9).contiguous()
    gi, bi = (torch.rand(cin, generator=g) + 0.5).to(DEV), (torch.randn(cin, generator=g) * .2).to(DEV)
    go, bo = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * .2).to(DEV)
    dy = nhwc(torch.randn(n, cout, h, w, generator=g)).to(DEV)
    sx = stats_of(x)
    res = {}
    for slots in (1, S):
        rep = (lambda v: v.clone()) if slots == 1 else (lambda v: _replicas(v, slots, g))
        zero = lambda c: torch.zeros(slots * 2 * c, dtype=torch.float64, device=DEV)
        in_bn = k.BN(rep(sx), gi, bi, n * h * w, bstats=zero(cin), slots=slots)
        out_bn = k.BN(zero(cout), go, bo, n * h * w, slots=slots) if cout != 16 or cin != 64 else None
        z = k.dp_fwd(x, w_pw, b_pw, w_dw, b_dw, in_bn, out_bn)
        torch.cuda.synchronize()
        sz = stats_of(z)
        if out_bn is not None:
            got = out_bn.stats.view(slots, -1)
            assert rel_err(got.sum(0), sz) < 2e-5
            if slots > 1 and n * h * w >= 4 * 512:   # more workgroups than replicas: every replica was written
                assert bool((got[:, cout:] > 0).all())
        dy_scale = None
        if out_bn is not None:
            dyd = dy.double()
            mean = sz[:cout] / (n * h * w)
            var = sz[cout:] / (n * h * w) - mean * mean
            xhat = (z.double() - mean) / torch.sqrt(var + 1e-5)
            mask = (xhat * go.double() + bo.double()) > 0
            dym = dyd * mask
            bst = torch.cat([dym.sum(dim=(0, 1, 2)), (dym * xhat).sum(dim=(0, 1, 2))])
            out_bn = k.BN(rep(sz), go, bo, n * h * w, bstats=rep(bst), slots=slots)
        else:
            dy_scale = torch.ones(cout, device=DEV)
        r = k.dp_bwd(x, w_pw, b_pw, w_dw, b_dw, z, dy, in_bn, out_bn, dy_scale=dy_scale)
        torch.cuda.synchronize()
        res[slots] = [z] + [t.clone() for t in r] + [in_bn.bstats.view(slots, -1).sum(0)]
        if cin == cout:        # the element-wise kernels that read and write the same blocks
            bn_p = k.BN(rep(sx), gi, bi, n * h * w, bstats=zero(cin), slots=slots)
            pooled = k.pool_fwd(x, bn_p)
            dpool = k.pool_bwd(x, bn_p, torch.ones_like(pooled) * 0.5)
            bn_a = k.BN(rep(sx), gi, bi, n * h * w, bstats=zero(cin), slots=slots)
            bn_b = k.BN(rep(stats_of(pooled)), gi, bi, n * h * w // 4, bstats=zero(cin), slots=slots)
            up = k.upadd_fwd(x, bn_a, pooled, bn_b)
            dxa, dxb = k.upadd_bwd(x, bn_a, pooled, bn_b, dy[..., :cin].contiguous())
            torch.cuda.synchronize()
            res[slots] += [pooled, dpool, bn_p.bstats.view(slots, -1).sum(0), up, dxa, dxb,
                           bn_a.bstats.view(slots, -1).sum(0), bn_b.bstats.view(slots, -1).sum(0)]
    for i, (a, b) in enumerate(zip(res[1], res[S])):
        if i == 5 and not (cout == 16 and cin == 64):
            continue      # d(depthwise bias) in front of a BatchNorm is identically zero: rounding noise in both
        # the replica shares add up to the sums to 1e-16; the coefficients derived from them agree to fp32 rounding
        assert rel_err(b, a) < 5e-6, i


def test_bn_running_and_param_grad():
    k = K()
    g = torch.Generator().manual_seed(5)
    c, cnt = 64, 777
    x = torch.randn(cnt, c, generator=g).double() * 3 + 2
    stats = torch.cat([x.sum(0), (x * x).sum(0)]).to(DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    k.bn_update_running(stats, rm, rv, cnt)
    torch.cuda.synchronize()
    assert rel_err(rm, 0.1 * x.mean(0)) < 1e-5
    assert rel_err(rv, 0.9 + 0.1 * x.var(0, unbiased=True)) < 1e-5
    dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    k.bn_param_grad(stats, dg, db)
    torch.cuda.synchronize()
    assert rel_err(db, x.sum(0)) < 1e-6 and rel_err(dg, (x * x).sum(0)) < 1e-6


@pytest.mark.parametrize('momentum,dampening,nesterov', [(0.9, 0.0, False), (0.9, 0.0, True), (0.9, 0.3, False), (0.0, 0.0, False)])
def test_sgd_matches_torch(momentum, dampening, nesterov):
    """yunet_sgd_step_ex against torch.optim.SGD itself: the shipped hyper-parameters (configs/yunet_n.py:1) and the
    constructor's other arguments (Nesterov, dampening, no momentum)."""
    k = K()
    g = torch.Generator().manual_seed(9)
    n = 75856
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.01, momentum=momentum, weight_decay=5e-4, dampening=dampening, nesterov=nesterov)
    p = p0.clone().to(DEV)
    buf = torch.zeros(n, device=DEV)
    lr = torch.tensor([0.01], device=DEV)
    for it in range(4):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        k.sgd_step(p, gr.to(DEV), buf, lr, momentum, 5e-4, first=(it == 0), dampening=dampening, nesterov=nesterov)
    torch.cuda.synchronize()
    assert rel_err(p, ref.detach()) < 1e-6
    import yunet_amd._lib as L
    rc = L.load().yunet_sgd_step_ex(p.data_ptr(), p.data_ptr(), buf.data_ptr(), n, lr.data_ptr(), 0.0, 0.0, 1, 0.0, 1.0, 0, None)
    assert rc == L.EINVAL       # Nesterov without momentum: torch raises ValueError


def test_reduce_partials_batch():
    """One launch over a table of partial buffers == the per-buffer reductions, bit for bit
    (same summation order), incl. ragged widths and the accumulate flag."""
    k = K()
    g = torch.Generator().manual_seed(2)
    shapes = [(256, 4816), (768, 448), (37, 1), (256, 1360), (5, 64), (256, 65)]
    parts = [torch.randn(r, w, generator=g).to(DEV) for r, w in shapes]
    base = [torch.randn(w, generator=g).to(DEV) for _, w in shapes]
    acc = [False, True, False, True, True, False]
    want = [b.clone() for b in base]
    for p, o, a in zip(parts, want, acc):
        k.reduce_partials(p, o, accumulate=a)
    got = [b.clone() for b in base]
    tab = k.reduce_partials_batch(parts, got, acc)
    torch.cuda.synchronize()
    for w_, g_ in zip(want, got):
        assert torch.equal(w_, g_)
    del tab


def test_reduce_partials():
    k = K()
    g = torch.Generator().manual_seed(1)
    part = torch.randn(1000, 4816, generator=g).to(DEV)
    out = torch.zeros(4816, device=DEV)
    k.reduce_partials(part, out)
    torch.cuda.synchronize()
    assert rel_err(out, part.double().sum(0)) < 1e-5


@pytest.mark.parametrize('cin,cout,bias', [(16, 16, 0.1), (16, 16, 30.0), (64, 64, 30.0)])
def test_dp_bwd_weight_gradient_precision_with_large_activation_means(cin, cout, bias):
    """BatchNorm behind the unit makes sum(dz) = 0 in exact arithmetic; with a large mean of the
    pointwise output p the depthwise weight gradient sum(p * dz) then hinges on how exactly dz
    sums to zero.  The (hi, lo) means / fp64 sums of the kernels keep its error at or below the
    error of torch's own fp32 autograd against fp64 (the reference's arithmetic)."""
    import torch.nn.functional as F
    k = K()
    g = torch.Generator().manual_seed(0)
    n, h, w = 8, 40, 40
    x = torch.randn(n, cin, h, w, generator=g) * 2 + 0.5
    w_pw, b_pw, w_dw, b_dw = mk_unit(cin, cout, g)
    b_pw = b_pw + bias
    go, bo = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * .2
    r = torch.randn(n, cout, h, w, generator=g)
    res = {}
    for name, dt in (('f64', torch.float64), ('f32', torch.float32)):
        ws = [t.to(dt).clone().requires_grad_(True) for t in (w_pw, b_pw, w_dw, b_dw)]
        z = F.conv2d(F.conv2d(x.to(dt), ws[0], ws[1]), ws[2], ws[3], padding=1, groups=cout)
        zb = F.batch_norm(z, None, None, go.to(dt), bo.to(dt), True, 0.1, 1e-5)
        zb.retain_grad()
        (F.relu(zb) * r.to(dt)).sum().backward()
        res[name] = (ws[2].grad.double(), zb.grad, z.detach())
    dy64, z64 = res['f64'][1], res['f64'][2]
    _, xhat = bn_ref(z64, go.double(), bo.double())
    xg, zg = nhwc(x).to(DEV), nhwc(res['f32'][2]).to(DEV)
    dyg = nhwc(res['f32'][1].float()).to(DEV)
    bst = torch.cat([dy64.sum(dim=(0, 2, 3)), (dy64 * xhat).sum(dim=(0, 2, 3))]).to(DEV)
    out_bn = k.BN(stats_of(zg), go.to(DEV), bo.to(DEV), n * h * w, bstats=bst.contiguous())
    _, _, _, dw2, _ = k.dp_bwd(xg, w_pw.to(DEV).view(cout, cin).contiguous(), b_pw.to(DEV),
                               w_dw.to(DEV).view(cout, 9).contiguous(), b_dw.to(DEV), zg, dyg, None, out_bn)
    torch.cuda.synchronize()
    ref = res['f64'][0]
    scale = float(ref.abs().max())
    err_torch32 = float((res['f32'][0] - ref).abs().max())
    err_hip = float((dw2.cpu().double().view_as(ref) - ref).abs().max())
    # 16-channel variants meet torch's own fp32 error.  The 64 -> 64 variant keeps its 36 dW2
    # accumulators in fp32 registers across ~200 pixels per thread; with |p| ~ 30 (30x what the
    # BN-normalised network produces) that costs up to 0.6 % here -- bounded, and recorded as a known
    # limit rather than hidden.
    bound = max(2.0 * err_torch32, 2e-5 * scale) if cout < 64 else 1e-2 * scale
    assert err_hip <= bound, (err_hip, err_torch32, scale)


@pytest.mark.parametrize('n,shift', [(1, 0), (3, 0), (4, 0), (1027, 0), (1027, 1), (256 * 2100 * 16, 0), (70001, 3)])
def test_add(n, shift):
    """yunet_add (YUNET_OP_ADD, ABI 9): out = a + b, exact; 16-byte path, its tail, and unaligned buffers."""
    k = K()
    g = torch.Generator().manual_seed(n)
    a = torch.randn(n + shift, generator=g).to(DEV)[shift:]
    b = torch.randn(n + shift, generator=g).to(DEV)[shift:]
    guard = torch.full((n + shift + 8,), 7.0, device=DEV)
    out = guard[shift:shift + n]
    k.add(a, b, out)
    torch.cuda.synchronize()
    assert torch.equal(out, a + b)
    assert float(guard[shift + n:].min()) == 7.0 and (shift == 0 or float(guard[:shift].min()) == 7.0)
    # in place (out = a), the form the engine uses
    a2 = a.clone()
    k.add(a2, b, a2)
    torch.cuda.synchronize()
    assert torch.equal(a2, a + b)
