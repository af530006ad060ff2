"""-m gpu: the registered building blocks are differentiable on their own, like the reference's nn.Modules
(mmdet/models/utils/yunet_layer.py:30-36,57-62,79-82, backbones/yunet_backbone.py:33-41, necks/tfpn.py:33-45,
dense_heads/yunet_head.py:175-247) -- VERDICT r5 missing 2 / next 5.

  * a random cotangent backpropagated through YuNetBackbone + TFPN (one autograd node per unit on yunet_dp_fwd /
    yunet_dp_bwd / yunet_stem_*) against the oracle's parameter gradients (fp64 evaluation of the same stack);
  * the per-module training path (backbone -> neck -> head.forward -> head.loss -> backward) against the fused engine
    on the same weights and batch: same losses, same assignment, same parameter gradients;
  * eval-mode BatchNorm (running statistics, frozen) through the same nodes;
  * three SGD steps of backbone + neck under a toy torch head with torch.optim.SGD.
"""
import os

import pytest
import torch
import torch.nn.functional as F

import yunet_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(kind, sd=None):
    import yunet_amd
    cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', f'yunet_{kind}.py'))
    model = yunet_amd.build_detector(cfg.model)
    if sd is not None:
        model.load_state_dict(sd, strict=True)
    return model


def _oracle_backbone_neck_grads(sd, arch, img, cots, dtype, training=True):
    """Parameter gradients of sum_i <feat_i, cot_i> through the oracle's backbone + neck in `dtype`."""
    keys = [k for k in O.param_keys(sd) if k.startswith('backbone.') or k.startswith('neck.')]
    work = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaf = {k: work[k].clone().requires_grad_(True) for k in keys}
    work.update(leaf)
    feats = O.neck_forward(O.backbone_forward(img.to(dtype), work, arch, training), work, arch, training)
    total = sum((f * c.to(dtype)).sum() for f, c in zip(feats, cots))
    grads = torch.autograd.grad(total, [leaf[k] for k in keys])
    return keys, [f.detach() for f in feats], dict(zip(keys, grads)), work


@pytest.mark.parametrize('kind', ['n', 's'])
@pytest.mark.parametrize('training', [True, False])
def test_backbone_neck_cotangent_vs_oracle(kind, training):
    arch = O.yunet_arch(kind)
    sd = O.init_state(arch, seed=21)
    g = torch.Generator().manual_seed(5)
    # non-trivial BN affine / running statistics (init_state has gamma 1, beta 0, mean 0, var 1)
    for k in list(sd):
        if k.endswith('bn.weight') or k.endswith('bn1.weight'):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
        elif k.endswith('bn.bias') or k.endswith('bn1.bias'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.2
        elif k.endswith('running_var'):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 30 + 5
        elif k.endswith('running_mean'):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.5
    n, h = 4, 160
    img = torch.rand(n, 3, h, h, generator=g) * 255.0
    if not training:
        # frozen BatchNorm: take the statistics a train-mode pass leaves behind, so that activations stay in range
        warm = {k: v.clone() for k, v in sd.items()}
        for _ in range(12):
            O.neck_forward(O.backbone_forward(img, warm, arch, True), warm, arch, True)
        for k in warm:
            if k.endswith('running_mean') or k.endswith('running_var'):
                sd[k] = warm[k].clone()
    sizes = [h // 8, h // 16, h // 32]
    cots = [torch.randn(n, 64, s, s, generator=g) for s in sizes]
    keys, feats64, g64, _ = _oracle_backbone_neck_grads(sd, arch, img, cots, torch.float64, training)
    _, _, g32, _ = _oracle_backbone_neck_grads(sd, arch, img, cots, torch.float32, training)

    model = _build(kind, sd)
    bb, neck = model.backbone.to(DEV), model.neck.to(DEV)
    bb.train(training), neck.train(training)
    feats = neck(bb(img.to(DEV)))
    assert all(f.requires_grad for f in feats)
    for f, r in zip(feats, feats64):
        assert float((f.detach().cpu().double() - r).abs().max()) <= 2e-4 * float(r.abs().max())
    sum((f * c.to(DEV)).sum() for f, c in zip(feats, cots)).backward()
    torch.cuda.synchronize()
    named = dict(list(('backbone.' + k, p) for k, p in bb.named_parameters()) +
                 list(('neck.' + k, p) for k, p in neck.named_parameters()))
    assert sorted(named) == sorted(keys)
    scale = max(float(v.abs().max()) for v in g64.values())
    worst, worst_ref = (0.0, None), 0.0
    for k in keys:
        got = named[k].grad
        assert got is not None, f'{k} received no gradient'
        e = float((got.cpu().double() - g64[k]).abs().max()) / scale
        worst = max(worst, (e, k))
        worst_ref = max(worst_ref, float((g32[k].double() - g64[k]).abs().max()) / scale)
    print(f'[modules {kind} training={training}] worst gradient error {worst[0]:.2e} ({worst[1]}) of the largest gradient; '
          f'oracle fp32 vs fp64: {worst_ref:.2e}')
    # 5e-5 of the largest gradient, or three times what the oracle's own fp32 evaluation is away from fp64
    # (train-mode BatchNorm over 20 layers amplifies rounding), whichever is larger
    assert worst[0] <= max(5e-5, 3 * worst_ref), worst
    if training:       # nn.BatchNorm2d bookkeeping
        ref = {k: v.clone() for k, v in sd.items()}
        O.neck_forward(O.backbone_forward(img, ref, arch, True), ref, arch, True)
        got_sd = {('backbone.' + k): v for k, v in bb.state_dict().items()}
        got_sd.update({('neck.' + k): v for k, v in neck.state_dict().items()})
        for k, v in ref.items():
            if k.endswith('running_mean') or k.endswith('running_var'):
                if k in got_sd:
                    assert float((got_sd[k].cpu() - v).abs().max()) <= 1e-4 * float(v.abs().max()) + 1e-6, k
            elif k.endswith('num_batches_tracked') and k in got_sd:
                assert int(got_sd[k]) == int(v), k


@pytest.mark.parametrize('kind', ['n', 's'])
def test_per_module_training_path_equals_the_fused_engine(kind):
    """backbone -> neck -> head.forward -> head.loss (one autograd node per unit + the fused loss step) computes what
    YuNet.forward_train (ONE node: the engine) computes."""
    import yunet_amd.synthetic as S
    arch = O.yunet_arch(kind)
    sd = O.init_state(arch, seed=3)
    b = S.make_batch(8, 160, 160, 77)
    bd = S.to_device(b, DEV)

    fused = _build(kind, sd).to(DEV).train()
    losses_f = fused.forward_train(**bd)
    sum(losses_f.values()).backward()
    torch.cuda.synchronize()
    gi_f = fused.engine.plan.gt_inds.clone()
    grads_f = {k: p.grad.detach().clone() for k, p in fused.named_parameters()}

    mod = _build(kind, sd).to(DEV).train()
    feats = mod.extract_feat(bd['img'])
    losses_m = mod.bbox_head.forward_train(feats, bd['img_metas'], bd['gt_bboxes'], bd['gt_labels'], bd['gt_keypointss'])
    sum(losses_m.values()).backward()
    torch.cuda.synchronize()
    assert torch.equal(mod.bbox_head.last_gt_inds, gi_f), 'the two paths assigned different priors'
    for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps'):
        assert float(losses_m[k]) == pytest.approx(float(losses_f[k]), rel=1e-4), k
    scale = max(float(v.abs().max()) for v in grads_f.values())
    for k, p in mod.named_parameters():
        assert p.grad is not None, k
        err = float((p.grad - grads_f[k]).abs().max()) / scale
        assert err <= 1e-4, (k, err)
    # BatchNorm running statistics moved identically
    sf, sm = fused.state_dict(), mod.state_dict()
    for k in sf:
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert float((sf[k] - sm[k]).abs().max()) <= 1e-4 * float(sf[k].abs().max()) + 1e-6, k


def test_backbone_and_neck_train_under_a_toy_head():
    """Three SGD steps (torch.optim.SGD over the modules' own parameters) of YuNetBackbone + TFPN under a foreign head:
    the loss falls and every parameter moves."""
    import yunet_amd
    torch.manual_seed(0)
    cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', 'yunet_n.py'))
    bb = yunet_amd.builder.build_backbone(cfg.model.backbone).to(DEV).train()
    neck = yunet_amd.builder.build_neck(cfg.model.neck).to(DEV).train()
    head = torch.nn.Conv2d(64, 3, 1).to(DEV)                       # a toy head in plain torch
    params = list(bb.parameters()) + list(neck.parameters()) + list(head.parameters())
    opt = torch.optim.SGD(params, lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(4, 3, 128, 128, generator=g) * 255).to(DEV)
    target = [torch.randn(4, 3, s, s, generator=g).to(DEV) for s in (16, 8, 4)]
    before = [p.detach().clone() for p in params]
    hist = []
    for _ in range(3):
        opt.zero_grad()
        feats = neck(bb(img))
        loss = sum(F.mse_loss(head(f), t) for f, t in zip(feats, target))
        loss.backward()
        opt.step()
        hist.append(float(loss))
    assert all(h == h for h in hist) and hist[-1] < hist[0], hist
    # (a conv bias in front of a train-mode BatchNorm has an identically zero gradient: only rounding noise moves it)
    names = ([k for k, _ in bb.named_parameters()] + [k for k, _ in neck.named_parameters()] + ['head.weight', 'head.bias'])
    still = [k for k, p, q in zip(names, params, before)
             if not float((p.detach() - q).abs().max()) > 0 and not (k.endswith('.bias') and 'bn' not in k)]
    assert not still, f'parameter tensors that did not move: {still}'


def test_single_unit_and_head_nodes():
    """ConvDPUnit with / without BN and the fused per-level head node against torch autograd on the same arithmetic."""
    import yunet_amd.functional as Fh
    from yunet_amd.yunet_layer import ConvDPUnit
    g = torch.Generator().manual_seed(9)
    for cin, cout, bn in ((16, 16, True), (64, 64, True), (16, 64, True), (64, 16, False)):
        m = ConvDPUnit(cin, cout, withBNRelu=bn).to(DEV).train()
        x = torch.randn(2, cin, 24, 40, generator=g).to(DEV).requires_grad_(True)
        r = torch.randn(2, cout, 24, 40, generator=g).to(DEV)
        (m(x) * r).sum().backward()
        got = [x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
        x.grad = None
        for p in m.parameters():
            p.grad = None
        z = F.conv2d(F.conv2d(x.double(), m.conv1.weight.double(), m.conv1.bias.double()), m.conv2.weight.double(),
                     m.conv2.bias.double(), padding=1, groups=cout)
        if bn:
            z = F.relu(F.batch_norm(z, None, None, m.bn.weight.double(), m.bn.bias.double(), training=True, eps=m.bn.eps))
        ref = torch.autograd.grad((z * r.double()).sum(), [x] + list(m.parameters()))
        for a, e, name in zip(got, ref, ['x'] + [k for k, _ in m.named_parameters()]):
            if bn and name == 'conv2.bias':          # identically zero behind a train-mode BatchNorm (noise only)
                continue
            assert float((a.double() - e).abs().max()) <= 5e-5 * float(e.abs().max()) + 1e-7, (cin, cout, name)
    # the four heads of a level as one node: gradients reach each unit's own parameters
    units = [ConvDPUnit(64, c, False).to(DEV) for c in (1, 4, 1, 10)]
    x = torch.randn(2, 64, 20, 20, generator=g).to(DEV).requires_grad_(True)
    r = torch.randn(2, 16, 20, 20, generator=g).to(DEV)
    (Fh.fused_dp_units(units, x) * r).sum().backward()
    zs = [F.conv2d(F.conv2d(x.double(), u.conv1.weight.double(), u.conv1.bias.double()), u.conv2.weight.double(),
                   u.conv2.bias.double(), padding=1, groups=u.out_channels) for u in units]
    ps = [p for u in units for p in u.parameters()]
    ref = torch.autograd.grad((torch.cat(zs, 1) * r.double()).sum(), [x] + ps)
    for a, e in zip([x.grad] + [p.grad for p in ps], ref):
        assert float((a.double() - e).abs().max()) <= 5e-5 * float(e.abs().max()) + 1e-7
