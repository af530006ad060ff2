"""Precision stress for the depthwise weight gradient: large mean(p), BN behind the unit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
import yunet_amd.kernels as k
from test_kernels_gpu import nhwc, nchw, stats_of, bn_ref, mk_unit
DEV = 'cuda'
g = torch.Generator().manual_seed(0)
for (n, h, w, cin, cout, bias) in [(8, 40, 40, 16, 16, 0.1), (8, 40, 40, 16, 16, 30.0), (8, 40, 40, 64, 64, 30.0)]:
    x = (torch.randn(n, cin, h, w, generator=g) * 2 + 0.5)
    w_pw, b_pw, w_dw, b_dw = mk_unit(cin, cout, g)
    b_pw = b_pw + bias
    go, bo = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * .2
    r = torch.randn(n, cout, h, w, generator=g)
    res = {}
    for name, dt in (('f64', torch.float64), ('f32', torch.float32)):
        ws = [t.to(dt).clone().requires_grad_(True) for t in (w_pw, b_pw, w_dw, b_dw)]
        xx = x.to(dt)
        z = F.conv2d(F.conv2d(xx, ws[0], ws[1]), ws[2], ws[3], padding=1, groups=cout)
        zb = F.batch_norm(z, None, None, go.to(dt), bo.to(dt), True, 0.1, 1e-5)
        zb.retain_grad()
        (F.relu(zb) * r.to(dt)).sum().backward()
        res[name] = (ws[2].grad.double(), zb.grad, z.detach())
    dy64, z64 = res['f64'][1], res['f64'][2]
    _, xhat = bn_ref(z64, go.double(), bo.double())
    xg = nhwc(x).to(DEV)
    zg = nhwc(res['f32'][2]).to(DEV)           # the fp32 forward output, as the engine would have
    dyg = nhwc(res['f32'][1].float()).to(DEV)
    bst = torch.cat([dy64.sum(dim=(0, 2, 3)), (dy64 * xhat).sum(dim=(0, 2, 3))]).to(DEV)
    out_bn = k.BN(stats_of(zg), go.to(DEV), bo.to(DEV), n * h * w, bstats=bst.contiguous())
    dx, dw1, db1, dw2, db2 = k.dp_bwd(xg, w_pw.to(DEV).view(cout, cin).contiguous(), b_pw.to(DEV),
                                      w_dw.to(DEV).view(cout, 9).contiguous(), b_dw.to(DEV), zg, dyg, None, out_bn)
    torch.cuda.synchronize()
    ref = res['f64'][0]
    e32 = float((res['f32'][0] - ref).abs().max())
    ehip = float((dw2.cpu().double() - ref).abs().max())
    print(f'bias {bias:5.1f} c {cin}->{cout}: max|dW2| {float(ref.abs().max()):.4f}  torch-fp32 err {e32:.3e}  hip err {ehip:.3e}  db2 hip {float(db2.abs().max()):.3e}')
