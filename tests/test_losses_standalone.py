"""not-gpu: the stand-alone `forward` of the loss modules (reference signature
forward(pred, target, weight, avg_factor, reduction_override)) against the unmodified reference
classes when the reference tree is present, and against closed-form values otherwise."""
import pytest
import torch

import ref_stub


def boxes(n, g):
    xy = torch.rand(n, 2, generator=g) * 100
    wh = torch.rand(n, 2, generator=g) * 50 + 1
    return torch.cat([xy, xy + wh], 1)


def test_closed_form_values():
    from yunet_amd import losses as L
    b = torch.tensor([[0., 0., 10., 10.]])
    assert float(L.EIoULoss(reduction='sum')(b, b)) == pytest.approx(0.0, abs=1e-6)
    assert float(L.DIoULoss(reduction='sum')(b, b)) == pytest.approx(0.0, abs=1e-6)
    half = torch.tensor([[0., 0., 10., 5.]])                         # IoU 0.5 -> x = 0.5 >= 0.1
    assert float(L.EIoULoss(reduction='sum')(half, b)) == pytest.approx(0.5 - 0.05, abs=1e-5)
    d = torch.tensor([[0.05, 1.0]])
    want = 0.5 * 0.05 ** 2 / (1 / 9) + (1.0 - 0.5 / 9)
    assert float(L.SmoothL1Loss(beta=1 / 9, reduction='sum')(d, torch.zeros(1, 2))) == pytest.approx(want, rel=1e-6)
    z = torch.zeros(4, 1)
    assert float(L.CrossEntropyLoss(use_sigmoid=True, reduction='mean')(z, torch.ones(4, 1))) == \
        pytest.approx(0.6931472, rel=1e-6)
    with pytest.raises(ValueError):
        L.SmoothL1Loss(reduction='sum')(d, d, avg_factor=2.0)


@pytest.mark.skipif(not ref_stub.available(), reason='reference tree not present')
def test_against_reference_classes():
    from yunet_amd import losses as L
    ns = ref_stub.load_reference()
    g = torch.Generator().manual_seed(0)
    p, t = boxes(64, g), boxes(64, g)
    p[:8] = t[:8] + 0.3
    w1, w4 = torch.rand(64, generator=g), torch.rand(64, 4, generator=g)
    pairs = [(L.EIoULoss(loss_weight=5.0, reduction='sum'), ns.losses.iou.EIoULoss(loss_weight=5.0, reduction='sum')),
             (L.EIoULoss(reduction='mean'), ns.losses.iou.EIoULoss(reduction='mean')),
             (L.DIoULoss(loss_weight=2.0, reduction='sum'), ns.losses.iou.DIoULoss(loss_weight=2.0, reduction='sum')),
             # round 5: the rest of the family (YuNet_Head's own default is IoULoss(mode='square', eps=1e-16))
             (L.IoULoss(mode='square', eps=1e-16, reduction='sum', loss_weight=5.0),
              ns.losses.iou.IoULoss(mode='square', eps=1e-16, reduction='sum', loss_weight=5.0)),
             (L.IoULoss(mode='linear'), ns.losses.iou.IoULoss(mode='linear')),
             (L.IoULoss(), ns.losses.iou.IoULoss()),
             (L.GIoULoss(reduction='sum'), ns.losses.iou.GIoULoss(reduction='sum')),
             (L.CIoULoss(reduction='sum', loss_weight=2.0), ns.losses.iou.CIoULoss(reduction='sum', loss_weight=2.0))]
    cases = [dict(), dict(weight=w1), dict(weight=w4), dict(reduction_override='none'),
             dict(weight=w1, avg_factor=7.0, reduction_override='mean'), dict(weight=torch.zeros(64))]
    for mine, ref in pairs:
        for kw in cases:
            pa, pb = p.clone().requires_grad_(True), p.clone().requires_grad_(True)
            a, b = mine(pa, t, **kw), ref(pb, t, **kw)
            a.sum().backward()
            b.sum().backward()
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), (type(mine).__name__, list(kw))
            assert torch.allclose(pa.grad, pb.grad, rtol=1e-5, atol=1e-6)
    x, y = torch.randn(40, 10, generator=g), torch.randn(40, 10, generator=g) * 0.2
    for kw in [dict(), dict(weight=torch.rand(40, 1, generator=g)),
               dict(weight=torch.rand(40, 1, generator=g), avg_factor=3.0)]:
        a = L.SmoothL1Loss(beta=1 / 9, loss_weight=0.1)(x, y, **kw)
        b = ns.losses.sl1.SmoothL1Loss(beta=1 / 9, loss_weight=0.1)(x, y, **kw)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    lg, soft = torch.randn(30, 1, generator=g), torch.rand(30, 1, generator=g)
    mine = L.CrossEntropyLoss(use_sigmoid=True, reduction='sum')
    ref = ns.losses.ce.CrossEntropyLoss(use_sigmoid=True, reduction='sum')
    for kw in [dict(), dict(weight=torch.rand(30, 1, generator=g)), dict(avg_factor=4.0, reduction_override='mean')]:
        assert torch.allclose(mine(lg, soft, **kw), ref(lg, soft, **kw), rtol=1e-6, atol=1e-6)
    hard = torch.randint(0, 2, (30,), generator=g)          # 1 = background for a 1-class sigmoid head
    hard[3] = -100                                         # ignored
    a = L.CrossEntropyLoss(use_sigmoid=True, reduction='mean')(lg, hard)
    b = ns.losses.ce.CrossEntropyLoss(use_sigmoid=True, reduction='mean')(lg, hard)
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
