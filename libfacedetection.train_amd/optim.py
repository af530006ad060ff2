"""Fused SGD over the engine's flat parameter buffer (one kernel per step).

torch.optim.SGD semantics with the reference's hyper-parameters (configs/yunet_n.py:1:
lr 0.01, momentum 0.9, weight_decay 5e-4 on EVERY parameter -- no paramwise_cfg).  The
learning rate lives in device memory so LR schedules never force a host sync.
"""
import torch

from . import kernels as K


class FusedSGD:
    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=0.0, dampening=0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):        # torch/optim/sgd.py
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        self.model = model
        self.defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening, nesterov=nesterov)
        self.param_groups = [dict(self.defaults, initial_lr=lr, params=list(model.parameters()))]
        self._buf = None
        self._lr_dev = None
        self._lr_val = None
        self._steps = 0
        self._pending = None       # torch-format momentum buffers waiting for the engine's flat layout
        self.grad_scale = 1.0      # multiplies the gradient inside the update kernel (loss-scale removal)

    def _bind(self):
        eng = self.model.engine
        if eng is None:
            raise RuntimeError('FusedSGD.step() before the first forward_train: no gradients yet')
        n = eng.params.data.numel()
        if self._pending is not None:
            self._layout_pending(eng)
        if self._buf is not None and self._buf.numel() == n and self._buf.device != eng.device:
            # a momentum buffer restored from a checkpoint (map_location='cpu') or left behind by
            # model.to(other device): move it, keep the step count -- do NOT restart the momentum
            self._buf = self._buf.to(eng.device, torch.float32).contiguous()
            self._lr_dev, self._lr_val = None, None
        if self._buf is None or self._buf.numel() != n:
            self._buf = torch.zeros_like(eng.params.data)
            self._lr_dev, self._lr_val = None, None
            self._steps = 0
        if self._lr_dev is None or self._lr_dev.device != eng.device:
            self._lr_dev = torch.zeros(1, device=eng.device)
            self._lr_val = None
        return eng

    def _layout_pending(self, eng):
        """torch.optim.SGD per-parameter momentum buffers -> the engine's flat layout (parameters are views
        of eng.params.data, so a parameter's offset is the distance of its storage from the flat base)."""
        params = list(self.model.parameters())
        flat = torch.zeros_like(eng.params.data)
        base = eng.params.data.data_ptr()
        for i, mb in self._pending:
            p = params[i]
            off = (p.data_ptr() - base) // 4
            if off < 0 or off + p.numel() > flat.numel():
                raise RuntimeError('FusedSGD: a parameter is not a view of the engine\'s flat buffer')
            flat[off:off + p.numel()].copy_(mb)
        self._buf = flat
        self._pending = None
        self._lr_dev, self._lr_val = None, None

    def zero_grad(self, set_to_none=False):
        """Gradients are overwritten (not accumulated) by the fused backward; nothing to do."""
        return None

    @torch.no_grad()
    def step(self, closure=None):
        eng = self._bind()
        if getattr(eng, '_os_main', None) is not None or getattr(eng, '_os_side', None) is not None:
            # opt-in one-shot all-reduce: a peer that never arrived left NaN in the gradient and a status word on the
            # host.  Find out BEFORE the update is applied (and before an after_train_iter CheckpointHook could save the
            # poisoned parameters as "the last checkpoint"): wait for this step's collectives, then read the words
            # (costs the launch overlap of one step, only on this path)
            torch.cuda.current_stream(eng.device).synchronize()
            eng._check_oneshot()
        g = self.param_groups[0]
        if self._lr_val != g['lr']:
            self._lr_dev.fill_(float(g['lr']))
            self._lr_val = g['lr']
        K.sgd_step(eng.params.data, eng.params.grad, self._buf, self._lr_dev, g['momentum'],
                   g['weight_decay'], float(self.grad_scale), first=(self._steps == 0),
                   dampening=g.get('dampening', 0.0), nesterov=g.get('nesterov', False))
        self._steps += 1

    def state_dict(self):
        if self._pending is not None and self.model.engine is not None:
            self._layout_pending(self.model.engine)
        return dict(param_groups=[{k: v for k, v in self.param_groups[0].items() if k != 'params'}],
                    momentum_buffer=None if self._buf is None else self._buf.clone(),
                    steps=self._steps)

    def load_state_dict(self, sd):
        """Own format (state_dict above) or the reference's torch.optim.SGD format
        ({'state': {i: {'momentum_buffer': tensor}}, 'param_groups': [...]}, as stored in
        weights/yunet_*.pth): per-parameter momentum buffers are laid out into the flat buffer in
        model.parameters() order, which is the order torch numbers them in."""
        import warnings
        groups = sd.get('param_groups') or [{}]
        self.param_groups[0].update({k: v for k, v in groups[0].items() if k != 'params'})
        self._lr_dev, self._lr_val = None, None
        self._pending = None
        if sd.get('momentum_buffer') is not None:
            self._buf = sd['momentum_buffer'].detach().clone().float().reshape(-1)
            self._steps = int(sd.get('steps', 1))
        elif isinstance(sd.get('state'), dict) and sd['state']:
            params = list(self.model.parameters())
            bufs = []
            for i, p in enumerate(params):
                st = sd['state'].get(i)
                mb = None if st is None else st.get('momentum_buffer')
                if mb is None or mb.numel() != p.numel():
                    warnings.warn('optimizer state does not match the model: momentum restarts from zero')
                    bufs = None
                    break
                bufs.append((p, mb))
            if bufs is not None:
                # The flat layout is defined by the engine, which a model binds lazily at its first
                # forward -- EpochBasedRunner.resume() (train_detector with resume_from / auto_resume)
                # loads the optimizer BEFORE any forward.  Keep the per-parameter buffers and lay them out
                # in _bind(), once the engine exists.
                self._pending = [(i, mb.detach().clone().float().reshape(-1)) for i, (_, mb) in enumerate(bufs)]
                self._buf = None
                self._steps = 1
                if self.model.engine is not None:
                    self._layout_pending(self.model.engine)
        else:
            warnings.warn('checkpoint carries no SGD momentum: it restarts from zero')
            self._buf = None
            self._steps = 0


def build_optimizer(model, cfg):
    """optimizer = dict(type='SGD', ...) -> FusedSGD (mmdet/apis/train.py:167)."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t != 'SGD':
        raise NotImplementedError(f'optimizer type {t}: the reference configs use SGD')
    target = model.module if hasattr(model, 'module') else model
    return FusedSGD(target, **cfg)
