"""-m gpu: checkpoint -> resume keeps the SGD momentum (the restored buffer used to be
replaced by zeros on the first step), for this repo's checkpoint format and for the reference's
torch.optim.SGD optimizer state (what weights/yunet_*.pth carry)."""
import pytest
import torch

import yunet_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _mk(seed=5):
    import yunet_amd
    from yunet_amd.optim import FusedSGD
    cfg = yunet_amd.Config.fromfile('configs/yunet_s.py')
    m = yunet_amd.build_detector(cfg.model)
    m.load_state_dict(O.init_state(O.yunet_arch('s'), seed=seed), strict=True)
    m.to(DEV).train()
    return m, FusedSGD(m, lr=0.01, momentum=0.9, weight_decay=5e-4)


def _step(m, opt, it):
    import yunet_amd.synthetic as S
    b = S.to_device(S.make_batch(4, 160, 160, S.batch_seed(0, it)), DEV)
    out = m.train_step(b, opt)
    opt.zero_grad()
    out['loss'].backward()
    opt.step()
    return float(out['log_vars']['loss'])


def test_resume_equals_uninterrupted_run(tmp_path):
    import yunet_amd.runner as R
    m, opt = _mk()
    for it in range(2):
        _step(m, opt, it)
    path = str(tmp_path / 'ck.pth')
    R.save_checkpoint(m, opt, path, dict(epoch=1, iter=2))
    loss_a = _step(m, opt, 2)
    torch.cuda.synchronize()
    want = m.engine.params.data.detach().cpu().clone()
    # a fresh process would do exactly this
    m2, opt2 = _mk(seed=99)
    meta = R.load_checkpoint(m2, path, opt2)
    assert meta['iter'] == 2
    loss_b = _step(m2, opt2, 2)
    torch.cuda.synchronize()
    got = m2.engine.params.data.detach().cpu()
    assert loss_b == pytest.approx(loss_a, rel=1e-6)
    # with zeroed momentum the update would differ by ~0.9 * lr * |buf|: orders of magnitude above this bar
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert opt2._steps == 3


def test_reference_format_optimizer_state_is_loaded():
    """{'state': {i: {'momentum_buffer'}}, 'param_groups'} of torch.optim.SGD over model.parameters()."""
    m, opt = _mk()
    _step(m, opt, 0)
    ref_opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(0)
    state = {}
    for i, p in enumerate(m.parameters()):
        state[i] = dict(momentum_buffer=torch.randn(p.shape, generator=g))
    sd = dict(state=state, param_groups=[dict(lr=1e-4, momentum=0.9, weight_decay=5e-4,
                                              params=list(range(len(state))))])
    opt.load_state_dict(sd)
    assert opt.param_groups[0]['lr'] == 1e-4
    flat = opt._buf.cpu()
    for i, (name, p) in enumerate(m.named_parameters()):
        off, shape = m.engine.layout.entries[name]
        assert torch.equal(flat[off:off + p.numel()].view(shape), state[i]['momentum_buffer']), name
    del ref_opt


def test_resume_reference_format_checkpoint_through_train_detector(tmp_path):
    """train_detector -> EpochBasedRunner.resume() loads the optimizer BEFORE any forward,
    i.e. before the model has bound its engine; a reference-format checkpoint (torch.optim.SGD state, as in
    weights/yunet_*.pth) used to raise there.  The per-parameter momentum buffers now wait for the engine and
    are laid out at the first step: the resumed run equals the uninterrupted one."""
    import yunet_amd
    import yunet_amd.runner as R
    # uninterrupted: 2 steps, remember state after the 2nd, do a 3rd
    m, opt = _mk()
    for it in range(2):
        _step(m, opt, it)
    torch.cuda.synchronize()
    params = list(m.named_parameters())
    flat = opt._buf.detach().cpu()
    state = {}
    for i, (name, p) in enumerate(params):
        off, shape = m.engine.layout.entries[name]
        state[i] = dict(momentum_buffer=flat[off:off + p.numel()].view(shape).clone())
    ck = dict(meta=dict(epoch=0, iter=2), state_dict={k: v.detach().cpu() for k, v in m.state_dict().items()},
              optimizer=dict(state=state, param_groups=[dict(lr=0.01, momentum=0.9, weight_decay=5e-4, dampening=0,
                                                             nesterov=False, params=list(range(len(state))))]))
    path = str(tmp_path / 'ref_format.pth')
    torch.save(ck, path)
    loss_a = _step(m, opt, 2)
    torch.cuda.synchronize()
    want = m.engine.params.data.detach().cpu().clone()

    # the train_detector order: model -> optimizer -> runner.resume() -> first forward
    cfg = yunet_amd.Config.fromfile('configs/yunet_s.py')
    m2 = yunet_amd.build_detector(cfg.model).to(DEV).train()
    assert m2.engine is None, 'the engine is expected to bind lazily (otherwise this test checks nothing)'
    from yunet_amd.optim import build_optimizer
    opt2 = build_optimizer(m2, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=5e-4))
    runner = R.EpochBasedRunner(m2, opt2, None, lambda *a, **k: None, None, max_epochs=1)
    runner.resume(path)
    assert runner.iter == 2 and opt2._pending is not None
    loss_b = _step(m2, opt2, 2)
    torch.cuda.synchronize()
    got = m2.engine.params.data.detach().cpu()
    assert opt2._pending is None and opt2._steps == 2
    assert loss_b == pytest.approx(loss_a, rel=1e-6)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
