"""Minimal training loop standing in for the mmcv runner the reference drives
(mmdet/apis/train.py:117-246; mmcv EpochBasedRunner + OptimizerHook + StepLrUpdaterHook +
CheckpointHook + TextLoggerHook are un-vendored).

Per iteration, exactly the reference's order: lr hook -> model.train_step(batch, optimizer)
-> optimizer.zero_grad() -> loss.backward() -> optimizer.step() -> log buffer.
"""
import math
import os
import time
import warnings

import torch

from . import synthetic
from .optim import build_optimizer
from .parallel import build_ddp, get_dist_info


class LrSchedule:
    """lr_config = dict(policy=..., warmup=..., warmup_iters, warmup_ratio, ...) of mmcv's LrUpdaterHook family
    (mmcv/runner/hooks/lr_updater.py, un-vendored: restated from its published formulas, SURVEY App. C -- the shipped
    configs use policy='step' with by-epoch steps and a linear by-iteration warm-up, configs/yunet_n.py:4-10, and that
    case is the one checked against the shipped checkpoint's optimizer state).

    policy: 'step' (step int | list, gamma, min_lr) | 'fixed' | 'exp' (gamma) | 'poly' (power, min_lr) |
            'inv' (gamma, power) | 'CosineAnnealing' (min_lr | min_lr_ratio);  by_epoch (default True) selects the
    progress variable (epoch / iteration) of the regular schedule;  warmup: None | 'constant' | 'linear' | 'exp',
    always by iteration (mmcv's default warmup_by_epoch=False)."""

    POLICIES = ('step', 'fixed', 'exp', 'poly', 'inv', 'CosineAnnealing')

    def __init__(self, base_lr, step=None, gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1, policy='step',
                 by_epoch=True, min_lr=None, min_lr_ratio=None, power=1.0, warmup_by_epoch=False, **unknown):
        if policy not in self.POLICIES:
            raise NotImplementedError(f'lr policy {policy!r}: implemented are {self.POLICIES}')
        if warmup not in (None, 'constant', 'linear', 'exp'):
            raise ValueError(f'"{warmup}" is not a supported type for warming up, valid types are "constant", "linear" '
                             f'and "exp"')
        # an option this restatement does not implement must not be swallowed: the schedule would silently differ from
        # mmcv's
        if warmup_by_epoch:
            raise NotImplementedError('lr_config: warmup_by_epoch=True is not implemented (warm-up runs by iteration)')
        if unknown:
            raise NotImplementedError(f'lr_config: unsupported option(s) {sorted(unknown)} for policy {policy!r}')
        if policy == 'step' and isinstance(gamma, (list, tuple)):
            raise NotImplementedError('lr_config: a per-step gamma list is not implemented')
        if policy == 'step' and step is None:
            raise ValueError('lr policy "step" needs step=<int | list>')
        if policy == 'CosineAnnealing' and (min_lr is None) == (min_lr_ratio is None):
            raise ValueError('CosineAnnealing: exactly one of min_lr / min_lr_ratio')
        self.base_lr, self.policy, self.by_epoch = base_lr, policy, by_epoch
        self.step = step if isinstance(step, int) or step is None else list(step)
        self.gamma, self.power, self.min_lr, self.min_lr_ratio = gamma, power, min_lr, min_lr_ratio
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio

    def regular_lr(self, epoch, it, max_epochs=None, max_iters=None):
        progress = epoch if self.by_epoch else it
        max_progress = max_epochs if self.by_epoch else max_iters
        b = self.base_lr
        if self.policy == 'fixed':
            return b
        if self.policy == 'step':
            k = progress // self.step if isinstance(self.step, int) else sum(1 for s in self.step if progress >= s)
            lr = b * self.gamma ** k
            return max(lr, self.min_lr) if self.min_lr is not None else lr
        if self.policy == 'exp':
            return b * self.gamma ** progress
        if self.policy == 'inv':
            return b * (1 + self.gamma * progress) ** (-self.power)
        if max_progress is None:
            raise ValueError(f'lr policy {self.policy!r} needs the run length (max_epochs / max_iters)')
        if self.policy == 'poly':
            m = self.min_lr or 0.0
            return (b - m) * (1 - progress / max_progress) ** self.power + m
        target = b * self.min_lr_ratio if self.min_lr_ratio is not None else self.min_lr
        return target + 0.5 * (b - target) * (math.cos(math.pi * progress / max_progress) + 1)

    def lr_at(self, epoch, it, max_epochs=None, max_iters=None):
        lr = self.regular_lr(epoch, it, max_epochs, max_iters)
        if self.warmup is not None and it < self.warmup_iters:
            if self.warmup == 'constant':
                lr = lr * self.warmup_ratio
            elif self.warmup == 'linear':
                k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
                lr = lr * (1 - k)
            else:
                lr = lr * self.warmup_ratio ** (1 - it / self.warmup_iters)
        return lr


class StepLrWarmup(LrSchedule):
    """The shipped schedule (policy='step', linear warm-up); kept as a name for tools/make_trained_fixture.py."""

    def __init__(self, base_lr, step, gamma=0.1, warmup='linear', warmup_iters=0, warmup_ratio=0.1, policy='step', **kw):
        super().__init__(base_lr, step=step, gamma=gamma, warmup=warmup, warmup_iters=warmup_iters,
                         warmup_ratio=warmup_ratio, policy=policy, **kw)


class SyntheticWiderFace:
    """Endless synthetic WIDER-Face-shaped batches; stands in for
    RetinaFaceDataset + the cv2 pipeline, which need the WIDER images (out of scope)."""

    def __init__(self, img_scale=(640, 640), samples_per_gpu=16, iters_per_epoch=403, rank=0,
                 max_gt=synthetic.MAX_GT, resident=0, **_):
        self.h, self.w = img_scale[1], img_scale[0]
        self.bs, self.iters_per_epoch, self.rank, self.max_gt = \
            samples_per_gpu, iters_per_epoch, rank, max_gt
        # resident = k > 0: k batches are generated once, kept on the device and cycled (what bench.py feeds: the step
        # with inputs already in HBM); 0: a fresh batch is drawn on the host every iteration
        self.resident, self._pool = int(resident), {}

    def batch(self, it, device=None):
        if self.resident > 0 and device is not None:
            k = it % self.resident
            if k not in self._pool:
                b = synthetic.make_batch(self.bs, self.h, self.w, synthetic.batch_seed(self.rank, k), self.max_gt)
                self._pool[k] = synthetic.to_device(b, device)
            return self._pool[k]
        b = synthetic.make_batch(self.bs, self.h, self.w, synthetic.batch_seed(self.rank, it),
                                 self.max_gt)
        return synthetic.to_device(b, device) if device is not None else b


class SyntheticSourceImages:
    """Synthetic DECODED sources (uint8 HWC at WIDER-like sizes, ragged GT) pushed through the reference's train
    pipeline on the GPU (pipelines.DevicePipeline: RandomSquareCrop -> Resize -> RandomFlip -> collate) every iteration
    -- the role of RetinaFaceDataset + the cv2 worker processes, minus image decoding.

    host_fed=False: the decoded sources stay resident in HBM (WIDER-Face train, 12 880 images, is ~30 GB decoded: it
    fits 288 GB many times over), an iteration costs the two pipeline launches.
    host_fed=True : the sources live in PINNED HOST memory and every iteration's batch is uploaded on a copy stream into
    one of two device buffers while the previous step runs (what a host data loader has to pay: the decoded uint8
    source of an image is 1 - 2.4 MB, more than its 320 x 320 fp32 crop).  `timing=True` records events around the
    upload and the pipeline (`report()`)."""

    def __init__(self, pipeline, samples_per_gpu=16, iters_per_epoch=403, rank=0, pool=64,
                 src_hw=((768, 1024), (1024, 683), (500, 375), (683, 1024)), max_gt=synthetic.MAX_GT,
                 seed=0, host_fed=False, timing=False, **_):
        from .pipelines import DevicePipeline
        self.pipe = DevicePipeline(pipeline, seed=seed + 7919 * rank, gmax=64 if max_gt <= 64 else 128)
        self.bs, self.iters_per_epoch, self.rank = samples_per_gpu, iters_per_epoch, rank
        self.pool, self.src_hw, self.max_gt, self.seed = pool, src_hw, max_gt, seed
        self.host_fed, self.timing = bool(host_fed), bool(timing)
        self._src = None
        self._ev = []

    def _build(self, device):
        import numpy as np
        from .pipelines import SourceBatch
        rng = np.random.default_rng(self.seed + 1000 * self.rank)
        gen = torch.Generator().manual_seed(self.seed + 1000 * self.rank)
        imgs, boxes, kps = [], [], []
        for i in range(self.pool):
            h, w = self.src_hw[i % len(self.src_hw)]
            imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
            b, _, k = synthetic.make_gt(1, h, w, gen, self.max_gt)
            boxes.append(b[0])
            kps.append(k[0])
        idx = [i % self.pool for i in range(self.bs)]
        return SourceBatch.from_lists([imgs[i] for i in idx], [boxes[i] for i in idx],
                                      [kps[i] for i in idx], 'cpu' if self.host_fed else device)

    def _build_host_fed(self, device):
        """Pinned host copy of the batch's sources + two device buffers + the copy stream."""
        from .pipelines import SourceBatch
        host = self._build(device)
        self._host_src = host.src.pin_memory()
        meta = [t.to(device) for t in (host.src_off, host.src_hw, host.boxes, host.kps, host.gt_off)]
        self._bufs = [torch.empty_like(self._host_src, device=device) for _ in range(2)]
        self._views = [SourceBatch(b, *meta) for b in self._bufs]
        self._copy = torch.cuda.Stream(device=device)
        self._uploaded = [None, None]          # event: buffer b holds iteration's sources
        self._consumed = [None, None]          # event: the pipeline that read buffer b has run
        self._pending = {}

    def _upload(self, it):
        b = it % 2
        with torch.cuda.stream(self._copy):
            if self._consumed[b] is not None:
                self._copy.wait_event(self._consumed[b])
            e0 = torch.cuda.Event(enable_timing=self.timing)
            e0.record(self._copy)
            self._bufs[b].copy_(self._host_src, non_blocking=True)
            e1 = torch.cuda.Event(enable_timing=self.timing)
            e1.record(self._copy)
        self._uploaded[b] = e1
        self._pending[it] = (e0, e1)

    def batch(self, it, device=None):
        if device is None:
            raise RuntimeError('SyntheticSourceImages augments on the GPU: a device is required')
        cur = torch.cuda.current_stream()
        if not self.host_fed:
            if self._src is None:
                self._src = self._build(device)
            # (the two pipeline launches on a side stream under the previous step were measured: 4.96 -> 4.98 ms per
            #  iteration, nothing -- the step's persistent kernels fill the GPU, the pipeline's 0.31 ms is work, not latency)
            p0 = self._mark(cur)
            out = self.pipe(self._src, it)
            if self.timing:
                self._ev.append((None, (p0, self._mark(cur))))
            return out
        if self._src is None:
            self._src = True
            self._build_host_fed(device)
        if it not in self._pending:
            self._upload(it)
        b = it % 2
        h2d = self._pending.pop(it)
        cur.wait_event(self._uploaded[b])
        p0 = self._mark(cur)
        out = self.pipe(self._views[b], it)
        done = torch.cuda.Event(enable_timing=self.timing)
        done.record(cur)
        self._consumed[b] = done
        if self.timing:
            self._ev.append((h2d, (p0, done)))
        self._upload(it + 1)                   # the next batch travels while this step runs
        return out

    def _mark(self, stream):
        if not self.timing:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def report(self, skip=5):
        """Mean milliseconds per batch of the upload and of the two pipeline launches (timing=True; synchronises)."""
        if not self.timing:
            return {}
        torch.cuda.synchronize()
        ev = self._ev[skip:]
        out = {'batches_timed': len(ev)}
        h = [a.elapsed_time(b) for (hd, _) in ev if hd is not None for a, b in [hd]]
        p = [a.elapsed_time(b) for (_, (a, b)) in ev if a is not None]
        if h:
            out['h2d_ms'] = sum(h) / len(h)
            out['h2d_bytes'] = int(self._host_src.numel())
            out['h2d_GBs'] = out['h2d_bytes'] / (out['h2d_ms'] * 1e-3) / 1e9
        if p:
            out['pipeline_ms'] = sum(p) / len(p)
        self._ev = []
        return out


def save_checkpoint(model, optimizer, path, meta):
    """Reference format: torch.save({'meta', 'state_dict', 'optimizer'})."""
    target = model.module if hasattr(model, 'module') else model
    sd = {k: v.detach().cpu() for k, v in target.state_dict().items()}
    torch.save(dict(meta=meta, state_dict=sd, optimizer=optimizer.state_dict()), path)


def load_checkpoint(model, path, optimizer=None, strict=True):
    ck = torch.load(path, map_location='cpu', weights_only=False)
    target = model.module if hasattr(model, 'module') else model
    target.load_state_dict(ck['state_dict'] if 'state_dict' in ck else ck, strict=strict)
    if optimizer is not None:
        if isinstance(ck.get('optimizer'), dict):
            optimizer.load_state_dict(ck['optimizer'])
        else:
            import warnings
            warnings.warn(f'{path}: no optimizer state in the checkpoint, momentum restarts from zero')
    return ck.get('meta', {})


# ================================================================================ hooks / runner
PRIORITY = dict(HIGHEST=0, VERY_HIGH=10, HIGH=30, ABOVE_NORMAL=40, NORMAL=50, BELOW_NORMAL=60, LOW=70,
                VERY_LOW=90, LOWEST=100)


class Hook:
    """mmcv.runner.Hook: the stage methods an EpochBasedRunner calls."""
    stages = ('before_run', 'before_train_epoch', 'before_train_iter', 'after_train_iter',
              'after_train_epoch', 'after_run')

    def before_run(self, runner): pass                    # noqa: E704
    def after_run(self, runner): pass                     # noqa: E704
    def before_epoch(self, runner): pass                  # noqa: E704
    def after_epoch(self, runner): pass                   # noqa: E704
    def before_iter(self, runner): pass                   # noqa: E704
    def after_iter(self, runner): pass                    # noqa: E704

    def before_train_epoch(self, runner):
        self.before_epoch(runner)

    def after_train_epoch(self, runner):
        self.after_epoch(runner)

    def before_train_iter(self, runner):
        self.before_iter(runner)

    def after_train_iter(self, runner):
        self.after_iter(runner)

    def every_n_iters(self, runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False

    def every_n_epochs(self, runner, n):
        return (runner.epoch + 1) % n == 0 if n > 0 else False


class StepLrUpdaterHook(Hook):
    """lr_config = dict(policy='step', ...) (configs/yunet_n.py:4-10): sets the lr of every param group
    before each iteration (by-epoch steps, by-iteration linear warm-up).  The same hook carries the other policies of
    LrSchedule (mmcv names them {Fixed,Exp,Poly,Inv,CosineAnnealing}LrUpdaterHook)."""

    def __init__(self, base_lr=None, **cfg):
        self.cfg, self.base_lr, self.sched = cfg, base_lr, None

    def before_run(self, runner):
        base = self.base_lr if self.base_lr is not None else runner.optimizer.param_groups[0].get(
            'initial_lr', runner.optimizer.param_groups[0]['lr'])
        self.sched = LrSchedule(base, **self.cfg)

    def before_train_iter(self, runner):
        lr = self.sched.lr_at(runner.epoch, runner.iter, getattr(runner, 'max_epochs', None), getattr(runner, 'max_iters', None))
        for g in runner.optimizer.param_groups:
            g['lr'] = lr


class OptimizerHook(Hook):
    """optimizer_config = dict(grad_clip=None): zero_grad -> backward -> step (mmcv OptimizerHook)."""

    def __init__(self, grad_clip=None, **_):
        self.grad_clip = grad_clip

    def after_train_iter(self, runner):
        runner.optimizer.zero_grad()
        runner.outputs['loss'].backward()
        if self.grad_clip is not None:
            params = [p for p in runner.model.parameters() if p.requires_grad and p.grad is not None]
            torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)
        runner.optimizer.step()


class Fp16OptimizerHook(OptimizerHook):
    """cfg.fp16 = dict(loss_scale=512.) (mmdet/apis/train.py:181-185 -> mmcv Fp16OptimizerHook).
    The reference's mixed precision is fp16 storage with loss scaling.  On MI355X the reduced
    precision mode of this path is bf16 activation storage + bf16 matrix instructions in forward with
    fp32 gradients and master weights (BASELINE.json configs[2]); bf16 keeps fp32's exponent range,
    so the loss scale is applied and removed exactly (a power of two) and only the overflow check of
    `loss_scale='dynamic'` ever changes anything: a step with non-finite gradients is skipped and the
    scale halves, 2000 clean steps double it (mmcv LossScaler defaults)."""

    def __init__(self, grad_clip=None, loss_scale=512., distributed=True, **_):
        super().__init__(grad_clip)
        self.dynamic = loss_scale == 'dynamic' or isinstance(loss_scale, dict)
        init = 2. ** 16 if loss_scale == 'dynamic' else (loss_scale.get('init_scale', 2. ** 16)
                                                          if isinstance(loss_scale, dict) else float(loss_scale))
        self.scale, self.growth_interval, self._good = float(init), 2000, 0

    def before_run(self, runner):
        target = runner.model.module if hasattr(runner.model, 'module') else runner.model
        target.set_precision('bf16')

    def after_train_iter(self, runner):
        opt = runner.optimizer
        opt.zero_grad()
        (runner.outputs['loss'] * self.scale).backward()
        target = runner.model.module if hasattr(runner.model, 'module') else runner.model
        grad = target.engine.params.grad
        # mmcv's GradScaler-based hook skips a step with non-finite gradients for static scales too
        # (only a dynamic scale is then halved)
        if not bool(torch.isfinite(grad).all()):
            if self.dynamic:
                self.scale = max(self.scale / 2.0, 1.0)
                self._good = 0
            return                                             # skip the step, like LossScaler
        if hasattr(opt, 'grad_scale') and self.grad_clip is None:
            opt.grad_scale = 1.0 / self.scale                  # FusedSGD: folded into the update kernel
        else:
            # clipping measures the norm of the UNSCALED gradient (max_norm must not shrink by the loss
            # scale): remove the scale in place first
            grad.mul_(1.0 / self.scale)
            if hasattr(opt, 'grad_scale'):
                opt.grad_scale = 1.0
        if self.grad_clip is not None:
            params = [p for p in runner.model.parameters() if p.requires_grad and p.grad is not None]
            torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)
        opt.step()
        if self.dynamic:
            self._good += 1
            if self._good % self.growth_interval == 0:
                self.scale *= 2.0


class CheckpointHook(Hook):
    """checkpoint_config = dict(interval=N) : epoch_{k}.pth + a `latest.pth` copy (rank 0)."""

    def __init__(self, interval=1, by_epoch=True, max_keep_ckpts=-1, **_):
        self.interval, self.by_epoch, self.max_keep = interval, by_epoch, max_keep_ckpts

    def after_train_epoch(self, runner):
        if runner.rank != 0 or not runner.work_dir or not self.every_n_epochs(runner, self.interval):
            return
        runner.save_checkpoint(runner.work_dir, f'epoch_{runner.epoch + 1}.pth')
        if self.max_keep > 0:
            old = runner.epoch + 1 - self.max_keep * self.interval
            path = os.path.join(runner.work_dir, f'epoch_{old}.pth')
            if old > 0 and os.path.exists(path):
                os.remove(path)


class EvalHook(Hook):
    """mmdet/core/evaluation/eval_hooks.py:24-66 on this path: every `interval` epochs (by_epoch) or iterations the
    detector runs in eval mode over the validation set (the device test pipeline + get_bboxes), rank 0 calls
    dataset.evaluate(results, metric=...) and files the numbers in the log (runner.log_buffer entries tagged
    mode='val').  The shipped configs set interval=1001 (> max_epochs): the hook is registered, as in the reference,
    and never fires there.  Distributed (DistEvalHook): the images are sharded over the ranks and the results gathered
    on rank 0 (evaluation.multi_gpu_test), as the reference does -- no rank idles behind a barrier."""

    def __init__(self, dataset, interval=1, by_epoch=True, metric='mAP', start=None, scale=(640, 640),
                 max_images=None, save_best=None, distributed=False, **eval_kwargs):
        self.dataset, self.interval, self.by_epoch, self.metric = dataset, int(interval), by_epoch, metric
        self.start, self.scale, self.max_images, self.distributed = start, scale, max_images, distributed
        self.eval_kwargs = {k: v for k, v in eval_kwargs.items() if k in ('iou_thr',)}
        dropped = sorted(k for k in eval_kwargs if k not in self.eval_kwargs)
        if save_best is not None:
            dropped.append('save_best')
        if dropped:      # (mmcv's EvalHook options this path does not implement: say so instead of dropping them silently)
            warnings.warn(f'EvalHook: evaluation options {dropped} are not supported on this path and are ignored')
        self.best, self.results = None, []

    def _should(self, runner, count):
        if self.start is not None and count < self.start:
            return False
        return count > 0 and count % self.interval == 0

    def after_train_epoch(self, runner):
        if self.by_epoch and self._should(runner, runner.epoch + 1):
            self._evaluate(runner)

    def after_train_iter(self, runner):
        if not self.by_epoch and self._should(runner, runner.iter + 1):
            self._evaluate(runner)

    def _evaluate(self, runner):
        from .evaluation import multi_gpu_test, single_gpu_test
        target = runner.model.module if hasattr(runner.model, 'module') else runner.model
        dev = torch.device(runner.device) if not isinstance(runner.device, torch.device) else runner.device
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        sharded = self.distributed and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if sharded:
            # BatchNorm running statistics are per rank (no SyncBN, broadcast_buffers=False): the reference's
            # DistEvalHook broadcasts rank 0's running_var / running_mean before multi_gpu_test
            # (mmdet/core/evaluation/eval_hooks.py:101-107) so that every shard is evaluated by the SAME model --
            # the one rank 0 checkpoints.  Here the buffers of all layers are two flat tensors.
            broadcast_bn_buffers(target)
            dets = multi_gpu_test(target, self.dataset, dev, self.scale, self.max_images)      # None off rank 0
        else:
            dets = single_gpu_test(target, self.dataset, dev, self.scale, self.max_images) if runner.rank == 0 else None
        if dets is not None:
            res = self.dataset.evaluate(dets, metric=self.metric, **self.eval_kwargs)
            done = runner.iter if self.by_epoch else runner.iter + 1          # iterations completed so far
            entry = dict(mode='val', epoch=runner.epoch + 1, iter=done, **{k: float(v) for k, v in res.items()})
            runner.log_buffer.append(entry)
            self.results.append(entry)
            runner.logger('Epoch(val) [%d][%d]\t%s' % (runner.epoch + 1, len(dets),
                                                       ', '.join(f'{k}: {float(v):.4f}' for k, v in res.items())))
            if self.best is None or res.get('mAP', 0.0) > self.best:
                self.best = res.get('mAP', 0.0)


def broadcast_bn_buffers(model, src=0):
    """dist.broadcast of every BatchNorm running_var / running_mean from rank `src` (DistEvalHook._do_evaluate,
    mmdet/core/evaluation/eval_hooks.py:101-107).  Returns the number of tensors sent (0 without a process group)."""
    if not (torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
        return 0
    eng = getattr(model, 'engine', None)
    if eng is not None:          # the fused path: all layers' buffers live in two flat device tensors
        bufs = [eng.params.running_var, eng.params.running_mean]
    else:                        # a model that has not been bound to a device yet: its module buffers
        bufs = [b for n, b in model.named_buffers() if n.endswith('running_var') or n.endswith('running_mean')]
    gloo = torch.distributed.get_backend() != 'nccl'
    for b in bufs:
        if gloo and b.is_cuda:   # CPU process group (tests, ranks sharing one GPU): through the host
            h = b.detach().cpu()
            torch.distributed.broadcast(h, src)
            b.copy_(h)
        else:
            torch.distributed.broadcast(b, src)
    return len(bufs)


class LoggerHook(Hook):
    def __init__(self, interval=50, by_epoch=True, **_):
        self.interval, self.by_epoch = interval, by_epoch
        self._t0 = None

    def before_run(self, runner):
        self._t0 = time.time()

    def after_train_iter(self, runner):
        if not self.every_n_iters(runner, self.interval):
            return
        # log_vars are lazily copied device scalars backed by a 256-row staging ring: freeze them
        # into python floats when they are logged, never keep the lazy objects
        lv = {k: float(v) for k, v in runner.outputs['log_vars'].items()}
        dt = (time.time() - self._t0) / self.interval
        self._t0 = time.time()
        rec = dict(lv, iter=runner.iter + 1, epoch=runner.epoch + 1, lr=runner.current_lr()[0], time=dt)
        if not runner.log_buffer or runner.log_buffer[-1].get('mode') == 'val' or runner.log_buffer[-1]['iter'] != rec['iter']:
            runner.log_buffer.append(rec)            # once per iteration, however many logger hooks
        if runner.rank == 0:
            self.log(runner, rec)

    def log(self, runner, rec):
        raise NotImplementedError


class TextLoggerHook(LoggerHook):
    def log(self, runner, rec):
        keys = [k for k in rec if k not in ('iter', 'epoch', 'lr', 'time')]
        runner.logger(f"Epoch [{rec['epoch']}][{rec['iter']}] lr: {rec['lr']:.3e}, time: {rec['time']:.4f}, " +
                      ', '.join(f'{k}: {rec[k]:.4f}' for k in keys))


class TensorboardLoggerHook(LoggerHook):
    """configs/yunet_n.py:14-17.  Writes TensorBoard event files directly (tfevents record framing +
    Event / Summary protobuf wire format, masked CRC-32C) -- no tensorboard package needed."""

    def __init__(self, log_dir=None, interval=50, **kw):
        super().__init__(interval, **kw)
        self.log_dir, self.writer = log_dir, None

    def before_run(self, runner):
        super().before_run(runner)
        if runner.rank == 0:
            from .tb_events import EventWriter
            d = self.log_dir or os.path.join(runner.work_dir or '.', 'tf_logs')
            self.writer = EventWriter(d)

    def log(self, runner, rec):
        if self.writer is None:
            return
        for k, v in rec.items():
            if k not in ('iter', 'epoch'):
                self.writer.add_scalar(('train/' + k) if k != 'lr' else 'learning_rate', v, rec['iter'])
        self.writer.flush()

    def after_run(self, runner):
        if self.writer is not None:
            self.writer.close()


HOOKS = dict(TextLoggerHook=TextLoggerHook, TensorboardLoggerHook=TensorboardLoggerHook,
             CheckpointHook=CheckpointHook, OptimizerHook=OptimizerHook, Fp16OptimizerHook=Fp16OptimizerHook)


class EpochBasedRunner:
    """The slice of mmcv.runner.EpochBasedRunner the reference's train_detector drives
    (mmdet/apis/train.py:169-246): register_training_hooks / register_hook / resume /
    load_checkpoint / run, and per iteration  before_train_iter hooks -> model.train_step ->
    after_train_iter hooks (OptimizerHook: zero_grad, backward, step)."""

    def __init__(self, model, optimizer=None, work_dir=None, logger=print, meta=None, max_epochs=None,
                 max_iters=None):
        self.model, self.optimizer, self.work_dir, self.logger = model, optimizer, work_dir, logger
        self.meta = dict(meta or {})
        self.rank, self.world_size = get_dist_info()
        self._max_epochs, self._max_iters = max_epochs, max_iters
        self.epoch, self.iter, self.inner_iter = 0, 0, 0
        self.hooks, self.outputs, self.log_buffer = [], None, []
        self.data_source, self.device = None, 'cuda'

    @property
    def max_epochs(self):
        return self._max_epochs

    @property
    def max_iters(self):
        """mmcv EpochBasedRunner: max_epochs * len(data_loader) once the data source is known."""
        if self._max_iters is not None:
            return self._max_iters
        per = getattr(self.data_source, 'iters_per_epoch', None)
        return self._max_epochs * per if (self._max_epochs is not None and per) else None

    def current_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def register_hook(self, hook, priority='NORMAL'):
        hook.priority = PRIORITY[priority] if isinstance(priority, str) else int(priority)
        pos = len(self.hooks)
        while pos > 0 and self.hooks[pos - 1].priority > hook.priority:
            pos -= 1
        self.hooks.insert(pos, hook)

    def register_training_hooks(self, lr_config, optimizer_config=None, checkpoint_config=None, log_config=None,
                                momentum_config=None, custom_hooks_config=None):
        if lr_config is not None:
            cfg = dict(lr_config)
            policy = cfg.pop('policy', 'step')
            if policy not in LrSchedule.POLICIES:
                raise NotImplementedError(f'lr policy {policy!r}: implemented are {LrSchedule.POLICIES}')
            self.register_hook(StepLrUpdaterHook(policy=policy, **cfg), 'VERY_HIGH')
        if isinstance(optimizer_config, Hook):
            self.register_hook(optimizer_config, 'ABOVE_NORMAL')
        else:
            oc = dict(optimizer_config or {})
            self.register_hook(HOOKS[oc.pop('type', 'OptimizerHook')](**oc), 'ABOVE_NORMAL')
        if checkpoint_config is not None:
            self.register_hook(CheckpointHook(**checkpoint_config), 'NORMAL')
        if log_config is not None:
            for h in log_config.get('hooks', [dict(type='TextLoggerHook')]):
                h = dict(h)
                self.register_hook(HOOKS[h.pop('type')](interval=log_config.get('interval', 50), **h), 'VERY_LOW')
        for h in custom_hooks_config or []:
            h = dict(h)
            prio = h.pop('priority', 'NORMAL')
            self.register_hook(HOOKS[h.pop('type')](**h), prio)

    def call_hook(self, stage):
        for h in self.hooks:
            getattr(h, stage)(self)

    def save_checkpoint(self, out_dir, filename, create_latest=True):
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, filename)
        save_checkpoint(self.model, self.optimizer, path, dict(self.meta, epoch=self.epoch + 1, iter=self.iter))
        if create_latest:
            import shutil
            shutil.copyfile(path, os.path.join(out_dir, 'latest.pth'))

    def load_checkpoint(self, path):
        return load_checkpoint(self.model, path)

    def resume(self, path):
        m = load_checkpoint(self.model, path, self.optimizer)
        self.epoch, self.iter = m.get('epoch', 0), m.get('iter', 0)
        self.logger(f'resumed epoch {self.epoch}, iter {self.iter} from {path}')

    def train(self, data_source):
        self.model.train()
        self.call_hook('before_train_epoch')
        for i in range(data_source.iters_per_epoch):
            self.inner_iter = i
            self.call_hook('before_train_iter')
            batch = data_source.batch(self.iter, self.device)
            self.outputs = self.model.train_step(batch, self.optimizer)
            self.call_hook('after_train_iter')
            self.iter += 1
            if self._max_iters is not None and self.iter >= self._max_iters:
                return False
        self.call_hook('after_train_epoch')
        self.epoch += 1
        return True

    def run(self, data_sources, workflow=(('train', 1),), device='cuda'):
        self.device = device
        src = data_sources[0] if isinstance(data_sources, (list, tuple)) else data_sources
        self.data_source = src
        self.call_hook('before_run')
        while self.epoch < self._max_epochs:
            if not self.train(src):
                break
        self.call_hook('after_run')
        return self.log_buffer


def find_latest_checkpoint(path, suffix='pth'):
    """mmdet/utils/misc.py:11-42: `latest.pth` if present, else the checkpoint with the largest
    trailing number (epoch_12.pth, iter_5000.pth)."""
    import glob
    import warnings
    if not os.path.exists(path):
        warnings.warn('The path of checkpoints does not exist.')
        return None
    if os.path.exists(os.path.join(path, f'latest.{suffix}')):
        return os.path.join(path, f'latest.{suffix}')
    best, best_path = -1, None
    for ck in glob.glob(os.path.join(path, f'*.{suffix}')):
        try:
            n = int(os.path.basename(ck).split('_')[-1].split('.')[0])
        except ValueError:
            continue
        if n > best:
            best, best_path = n, ck
    if best_path is None:
        warnings.warn('There are no checkpoints in the path.')
    return best_path


def update_data_root(cfg, log=print):
    """mmdet/utils/misc.py:45-76 (called first thing by tools/train.py:113 and tools/test.py:138): with the
    environment variable MMDET_DATASETS set, every string under cfg.data that contains cfg.data_root gets that part
    replaced by the variable's value, and cfg.data_root becomes that value.  Like the reference, nested dicts are
    walked, lists (pipelines) are not."""
    dst = os.environ.get('MMDET_DATASETS')
    if dst is None or cfg.get('data_root') is None or cfg.get('data') is None:
        return cfg
    log(f'MMDET_DATASETS has been set to be {dst}.Using {dst} as data root.')
    src = cfg['data_root']

    def walk(node):
        for k, v in list(node.items()):
            if isinstance(v, dict):
                walk(v)
            elif isinstance(v, str) and src in v:
                node[k] = v.replace(src, dst)
    walk(cfg['data'])
    cfg['data_root'] = dst
    return cfg


def auto_scale_lr(cfg, distributed, log=print):
    """mmdet/apis/train.py:71-113: the linear scaling rule, applied when the config carries
    `auto_scale_lr = dict(enable=True, base_batch_size=B)` (tools/train.py --auto-scale-lr switches `enable` on):
    optimizer.lr *= (GPUs x samples_per_gpu) / B.  Returns the learning rate in effect."""
    asl = cfg.get('auto_scale_lr')
    if not asl or not asl.get('enable', False):
        log('Automatic scaling of learning rate (LR) has been disabled.')
        return cfg.optimizer['lr']
    base = asl.get('base_batch_size')
    if base is None:
        return cfg.optimizer['lr']
    if distributed:
        from .parallel import get_dist_info
        num_gpus = get_dist_info()[1]
    else:
        num_gpus = len(cfg.get('gpu_ids') or [0])
    data = cfg.get('data') or {}
    spg = (data.get('train_dataloader') or {}).get('samples_per_gpu', data.get('samples_per_gpu'))
    batch = num_gpus * spg
    log(f'Training with {num_gpus} GPU(s) with {spg} samples per GPU. The total batch size is {batch}.')
    if batch != base:
        scaled = (batch / base) * cfg.optimizer['lr']
        log(f"LR has been automatically scaled from {cfg.optimizer['lr']} to {scaled}")
        cfg.optimizer['lr'] = scaled
    else:
        log(f"The batch size match the base batch size: {base}, will not scaling the LR ({cfg.optimizer['lr']}).")
    return cfg.optimizer['lr']


def train_detector(model, dataset, cfg, distributed=False, validate=False, timestamp=None,
                   meta=None, max_iters=None, device='cuda', log=print):
    """mmdet/apis/train.py:117-246 surface: DDP wrap, optimizer, EpochBasedRunner, fp16 / optimizer /
    lr / checkpoint / logger hooks from the config, auto-resume / resume / load_from, run.
    Returns the logged history (one dict of python floats per logging interval)."""
    model = model.to(device)
    model.train()
    if distributed:
        model = build_ddp(model, device, device_ids=[torch.cuda.current_device()],
                          broadcast_buffers=False)
    auto_scale_lr(cfg, distributed, log if cfg.get('auto_scale_lr') else (lambda *a: None))
    optimizer = build_optimizer(model, cfg.optimizer)
    runner = EpochBasedRunner(model, optimizer, cfg.get('work_dir'), log, meta,
                              max_epochs=cfg.runner['max_epochs'], max_iters=max_iters)
    fp16_cfg = cfg.get('fp16', None)
    opt_cfg = dict(cfg.get('optimizer_config') or {})
    if fp16_cfg is not None:
        optimizer_config = Fp16OptimizerHook(**opt_cfg, **fp16_cfg, distributed=distributed)
    else:
        optimizer_config = opt_cfg
    log_config = cfg.get('log_config')
    if max_iters is not None and max_iters <= 64 and log_config is not None:
        log_config = dict(log_config, interval=1)          # short smoke runs: log every iteration
    runner.register_training_hooks(cfg.lr_config, optimizer_config, cfg.get('checkpoint_config'), log_config,
                                   cfg.get('momentum_config'), cfg.get('custom_hooks'))
    # EvalHook (mmdet/apis/train.py:204-232): validate=True and a validation set in the config
    if validate and cfg.get('data') is not None and cfg.get('data').get('val') is not None:
        from .builder import build_dataset
        val_cfg = dict(cfg.get('data')['val'])
        val_cfg['test_mode'] = True
        eval_cfg = dict(cfg.get('evaluation', {}))
        eval_cfg['by_epoch'] = dict(cfg.runner).get('type', 'EpochBasedRunner') != 'IterBasedRunner'
        scale = None
        for t in val_cfg.get('pipeline') or []:
            if t.get('type') == 'MultiScaleFlipAug':
                scale = tuple(t['img_scale']) if not isinstance(t['img_scale'], list) else tuple(t['img_scale'][0])
        if os.path.exists(val_cfg.get('ann_file', '')):
            runner.register_hook(EvalHook(build_dataset(val_cfg), scale=scale or (640, 640), distributed=distributed,
                                          **eval_cfg), 'LOW')
        else:
            log(f"validate: {val_cfg.get('ann_file')} not found -- no EvalHook registered")
    resume_from = cfg.get('resume_from')
    if not resume_from and cfg.get('auto_resume') and cfg.get('work_dir'):
        resume_from = find_latest_checkpoint(cfg.work_dir)
    if resume_from:
        runner.resume(resume_from)
    elif cfg.get('load_from'):
        runner.load_checkpoint(cfg.load_from)
    return runner.run([dataset], cfg.get('workflow', [('train', 1)]), device=device)
