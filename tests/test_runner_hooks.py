"""CPU: the hook-based EpochBasedRunner (mmcv runner surface the reference's train_detector drives,
mmdet/apis/train.py:169-246): hook order per iteration, step-LR + warm-up applied before each
iteration, CheckpointHook + latest.pth, auto-resume (mmdet/utils/misc.py:11-42), TensorBoard event
files, Fp16OptimizerHook's loss-scale bookkeeping."""
import os
from collections import OrderedDict

import pytest
import torch

import yunet_amd
import yunet_amd.runner as R
import yunet_amd.tb_events as T


class ToyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0]))
        self.calls = []

    def train_step(self, data, optimizer):
        self.calls.append(('train_step', optimizer.param_groups[0]['lr']))
        loss = (self.w * data['x']).sum() ** 2
        return dict(loss=loss, log_vars=OrderedDict(loss=loss.detach(), loss_cls=loss.detach() * 0.5),
                    num_samples=1)


class ToySource:
    iters_per_epoch = 4

    def batch(self, it, device=None):
        return dict(x=torch.tensor([1.0 + it, 0.5]))


def _cfg(tmp_path, **extra):
    c = yunet_amd.Config(dict(
        optimizer=dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0),
        optimizer_config=dict(grad_clip=None),
        lr_config=dict(policy='step', warmup='linear', warmup_iters=6, warmup_ratio=0.001, step=[1, 2]),
        runner=dict(type='EpochBasedRunner', max_epochs=3),
        checkpoint_config=dict(interval=1),
        log_config=dict(interval=2, hooks=[dict(type='TextLoggerHook'), dict(type='TensorboardLoggerHook')]),
        work_dir=str(tmp_path)))
    c.merge_from_dict(extra)
    return c


def _runner(tmp_path, lines, **extra):
    cfg = _cfg(tmp_path, **extra)
    m = ToyModel()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    opt.param_groups[0]['initial_lr'] = 0.01
    r = R.EpochBasedRunner(m, opt, str(tmp_path), lines.append, dict(seed=1), max_epochs=3)
    r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, cfg.checkpoint_config, cfg.log_config)
    return r, m, opt


def test_runner_applies_lr_schedule_hooks_and_writes_checkpoints(tmp_path):
    lines = []
    r, m, opt = _runner(tmp_path, lines)
    hist = r.run([ToySource()], device='cpu')
    # lr seen by every train_step == StepLrWarmup(epoch, iter) (configs/yunet_n.py:4-10 semantics)
    sched = R.StepLrWarmup(0.01, step=[1, 2], warmup='linear', warmup_iters=6, warmup_ratio=0.001)
    want = [sched.lr_at(it // 4, it) for it in range(12)]
    got = [lr for _, lr in m.calls]
    assert got == pytest.approx(want) and got[0] == pytest.approx(0.01 * 0.001) and got[-1] == pytest.approx(1e-4)
    assert r.epoch == 3 and r.iter == 12
    # priorities: lr hook before the optimizer hook, loggers last
    kinds = [type(h).__name__ for h in r.hooks]
    assert kinds.index('StepLrUpdaterHook') < kinds.index('OptimizerHook') < kinds.index('CheckpointHook') \
        < kinds.index('TextLoggerHook')
    # logged every 2 iterations, python floats only
    assert len(hist) == 6 and all(isinstance(v, float) for v in hist[0].values() if not isinstance(v, int))
    assert len(lines) == 6 and lines[0].startswith('Epoch [1][2] lr:')
    # checkpoints: epoch_k.pth + latest.pth in the reference's format
    for k in (1, 2, 3):
        ck = torch.load(tmp_path / f'epoch_{k}.pth', weights_only=False)
        assert set(ck) == {'meta', 'state_dict', 'optimizer'} and ck['meta']['epoch'] == k and ck['meta']['iter'] == 4 * k
    assert R.find_latest_checkpoint(str(tmp_path)) == str(tmp_path / 'latest.pth')
    os.remove(tmp_path / 'latest.pth')
    (tmp_path / 'notes.pth').write_bytes(b'')                    # no trailing number: ignored
    assert R.find_latest_checkpoint(str(tmp_path)) == str(tmp_path / 'epoch_3.pth')
    with pytest.warns(UserWarning):
        assert R.find_latest_checkpoint(str(tmp_path / 'missing')) is None
    # TensorBoard events: every logged scalar, CRCs valid
    ev_dir = tmp_path / 'tf_logs'
    files = os.listdir(ev_dir)
    assert len(files) == 1 and files[0].startswith('events.out.tfevents.')
    ev = T.read_events(str(ev_dir / files[0]))
    assert [e for e in ev if e[1] == 'train/loss'][0][0] == 2
    assert {e[1] for e in ev} >= {'train/loss', 'train/loss_cls', 'learning_rate', 'train/time'}
    assert len([e for e in ev if e[1] == 'train/loss']) == 6


def test_resume_continues_epoch_iter_and_momentum(tmp_path):
    lines = []
    r, m, opt = _runner(tmp_path, lines)
    r._max_epochs = 2
    r.run([ToySource()], device='cpu')
    w_after_2 = m.w.detach().clone()
    r._max_epochs = 3
    r.run([ToySource()], device='cpu')
    want = m.w.detach().clone()
    # fresh objects, resume from epoch_2.pth
    os.remove(tmp_path / 'latest.pth')
    os.remove(tmp_path / 'epoch_3.pth')
    r2, m2, opt2 = _runner(tmp_path / 'b', lines)
    r2.resume(R.find_latest_checkpoint(str(tmp_path)))
    assert (r2.epoch, r2.iter) == (2, 8) and torch.equal(m2.w.detach(), w_after_2)
    r2.run([ToySource()], device='cpu')
    assert torch.allclose(m2.w.detach(), want, rtol=1e-6, atol=0)


class _FakeEngine:
    """engine.params.grad of the real model is a persistent view of the flat buffer; the toy's follows
    whatever .grad tensor autograd currently holds"""

    def __init__(self, p):
        class P:
            grad = property(lambda s: p.grad)
        self.params = P()


def test_fp16_hook_loss_scale_bookkeeping():
    """Static scale: applied to the loss, removed before the step; dynamic: a non-finite gradient skips
    the step and halves the scale."""
    m = ToyModel()
    m.precision = None
    m.set_precision = lambda p: setattr(m, 'precision', p)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    ref = ToyModel()
    ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    run = type('Rn', (), {})()
    run.model, run.optimizer = m, opt
    h = R.Fp16OptimizerHook(loss_scale=512.)
    h.before_run(run)
    assert m.precision == 'bf16'
    data = dict(x=torch.tensor([1.0, 0.5]))
    run.outputs = m.train_step(data, opt)
    m.engine = _FakeEngine(m.w)
    h.after_train_iter(run)
    out = ref.train_step(data, ref_opt)
    ref_opt.zero_grad(); out['loss'].backward(); ref_opt.step()
    assert torch.allclose(m.w.detach(), ref.w.detach(), rtol=1e-6)
    hd = R.Fp16OptimizerHook(loss_scale='dynamic')
    assert hd.scale == 2. ** 16
    before = m.w.detach().clone()
    run.outputs = dict(loss=(m.w * torch.tensor([float('inf'), 1.0])).sum())
    hd.after_train_iter(run)
    assert hd.scale == 2. ** 15 and torch.equal(m.w.detach(), before)


def test_lr_policies_closed_forms():
    """The policies of mmcv's LrUpdaterHook family besides the shipped 'step' (mmcv/runner/hooks/lr_updater.py,
    un-vendored: restated formulas, hand-evaluated here)."""
    import math
    L = R.LrSchedule
    assert L(0.1, policy='fixed').lr_at(7, 123) == 0.1
    assert L(0.1, policy='step', step=3, gamma=0.5).lr_at(7, 0) == pytest.approx(0.1 * 0.5 ** 2)
    assert L(0.1, policy='step', step=[2, 5], gamma=0.1, min_lr=0.005).lr_at(6, 0) == pytest.approx(0.005)
    assert L(0.1, policy='step', step=[100], by_epoch=False).lr_at(0, 150) == pytest.approx(0.01)
    assert L(0.1, policy='exp', gamma=0.9).lr_at(3, 0) == pytest.approx(0.1 * 0.9 ** 3)
    assert L(0.1, policy='inv', gamma=0.5, power=2.0).lr_at(2, 0) == pytest.approx(0.1 * (1 + 1.0) ** -2)
    assert L(0.1, policy='poly', power=2.0, min_lr=0.01).lr_at(5, 0, max_epochs=10) == pytest.approx((0.1 - 0.01) * 0.25 + 0.01)
    cos = L(0.1, policy='CosineAnnealing', min_lr_ratio=0.1)
    assert cos.lr_at(0, 0, max_epochs=10) == pytest.approx(0.1) and cos.lr_at(10, 0, max_epochs=10) == pytest.approx(0.01)
    assert cos.lr_at(5, 0, max_epochs=10) == pytest.approx(0.01 + 0.5 * 0.09 * (math.cos(math.pi / 2) + 1))
    # warm-up types (by iteration, on top of the regular lr)
    assert L(0.1, policy='fixed', warmup='constant', warmup_iters=10, warmup_ratio=0.2).lr_at(0, 3) == pytest.approx(0.02)
    assert L(0.1, policy='fixed', warmup='linear', warmup_iters=10, warmup_ratio=0.2).lr_at(0, 5) == pytest.approx(0.1 * (1 - 0.5 * 0.8))
    assert L(0.1, policy='fixed', warmup='exp', warmup_iters=10, warmup_ratio=0.01).lr_at(0, 5) == pytest.approx(0.1 * 0.01 ** 0.5)
    assert L(0.1, policy='fixed', warmup='linear', warmup_iters=10, warmup_ratio=0.2).lr_at(0, 10) == 0.1
    with pytest.raises(NotImplementedError):
        L(0.1, policy='cyclic')
    with pytest.raises(ValueError):
        L(0.1, policy='CosineAnnealing')
    with pytest.raises(ValueError):
        L(0.1, policy='poly').lr_at(1, 0)


def test_runner_runs_a_cosine_schedule(tmp_path):
    lines = []
    r, m, opt = _runner(tmp_path, lines, lr_config=dict(policy='CosineAnnealing', min_lr_ratio=0.0, by_epoch=False))
    r.run([ToySource()])
    lrs = [lr for tag, lr in m.calls if tag == 'train_step']
    import math
    assert len(lrs) == 12 and r.max_iters == 12
    for it, lr in enumerate(lrs):
        assert lr == pytest.approx(0.5 * 0.01 * (math.cos(math.pi * it / 12) + 1))


def test_lr_config_rejects_what_it_does_not_implement():
    """An lr_config key this restatement ignores would give a different schedule than mmcv without a word."""
    from yunet_amd.runner import LrSchedule
    LrSchedule(0.01, policy='step', step=[2, 4], warmup='exp', warmup_iters=3, warmup_ratio=0.1, warmup_by_epoch=False)
    for bad in (dict(warmup_by_epoch=True), dict(periods=[1, 2]), dict(cyclic_times=3), dict(gamma=[0.1, 0.5])):
        with pytest.raises(NotImplementedError):
            LrSchedule(0.01, policy='step', step=[2, 4], **bad)
    with pytest.raises(ValueError, match='"exp"'):
        LrSchedule(0.01, policy='step', step=1, warmup='cosine')
