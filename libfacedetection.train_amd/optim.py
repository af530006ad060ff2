"""Fused SGD over the engine's flat parameter buffer (one kernel per step).

torch.optim.SGD semantics with the reference's hyper-parameters (configs/yunet_n.py:1:
lr 0.01, momentum 0.9, weight_decay 5e-4 on EVERY parameter -- no paramwise_cfg).  The
learning rate lives in device memory so LR schedules never force a host sync.
"""
import torch

from . import kernels as K


class FusedSGD:
    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=0.0, dampening=0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError('dampening / nesterov are not used by the reference configs')
        self.model = model
        self.defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, initial_lr=lr, params=list(model.parameters()))]
        self._buf = None
        self._lr_dev = None
        self._lr_val = None
        self._steps = 0

    def _bind(self):
        eng = self.model.engine
        if eng is None:
            raise RuntimeError('FusedSGD.step() before the first forward_train: no gradients yet')
        if self._buf is None or self._buf.device != eng.device or \
                self._buf.numel() != eng.params.data.numel():
            self._buf = torch.zeros_like(eng.params.data)
            self._lr_dev = torch.zeros(1, device=eng.device)
            self._lr_val = None
            self._steps = 0
        return eng

    def zero_grad(self, set_to_none=False):
        """Gradients are overwritten (not accumulated) by the fused backward; nothing to do."""
        return None

    @torch.no_grad()
    def step(self, closure=None):
        eng = self._bind()
        g = self.param_groups[0]
        if self._lr_val != g['lr']:
            self._lr_dev.fill_(float(g['lr']))
            self._lr_val = g['lr']
        K.sgd_step(eng.params.data, eng.params.grad, self._buf, self._lr_dev, g['momentum'],
                   g['weight_decay'], 1.0, first=(self._steps == 0))
        self._steps += 1

    def state_dict(self):
        return dict(param_groups=[{k: v for k, v in self.param_groups[0].items() if k != 'params'}],
                    momentum_buffer=None if self._buf is None else self._buf.clone(),
                    steps=self._steps)

    def load_state_dict(self, sd):
        self.param_groups[0].update(sd['param_groups'][0])
        if sd.get('momentum_buffer') is not None:
            self._buf = sd['momentum_buffer'].clone()
            self._lr_dev = torch.zeros(1, device=self._buf.device)
            self._lr_val = None
        self._steps = sd.get('steps', 0)


def build_optimizer(model, cfg):
    """optimizer = dict(type='SGD', ...) -> FusedSGD (mmdet/apis/train.py:167)."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t != 'SGD':
        raise NotImplementedError(f'optimizer type {t}: the reference configs use SGD')
    target = model.module if hasattr(model, 'module') else model
    return FusedSGD(target, **cfg)
