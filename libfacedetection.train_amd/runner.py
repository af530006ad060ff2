"""Minimal training loop standing in for the mmcv runner the reference drives
(mmdet/apis/train.py:117-246; mmcv EpochBasedRunner + OptimizerHook + StepLrUpdaterHook +
CheckpointHook + TextLoggerHook are un-vendored).

Per iteration, exactly the reference's order: lr hook -> model.train_step(batch, optimizer)
-> optimizer.zero_grad() -> loss.backward() -> optimizer.step() -> log buffer.
"""
import os
import time

import torch

from . import synthetic
from .optim import build_optimizer
from .parallel import build_ddp, get_dist_info


class StepLrWarmup:
    """lr_config = dict(policy='step', warmup='linear', warmup_iters, warmup_ratio, step=[..])
    with by_epoch steps (configs/yunet_n.py:4-10)."""

    def __init__(self, base_lr, step, gamma=0.1, warmup='linear', warmup_iters=0,
                 warmup_ratio=0.1, policy='step', **_):
        assert policy == 'step'
        self.base_lr, self.step, self.gamma = base_lr, list(step), gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio

    def lr_at(self, epoch, it):
        lr = self.base_lr * self.gamma ** sum(1 for s in self.step if epoch >= s)
        if self.warmup == 'linear' and it < self.warmup_iters:
            k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
            lr = lr * (1 - k)
        return lr


class SyntheticWiderFace:
    """Endless synthetic WIDER-Face-shaped batches; stands in for
    RetinaFaceDataset + the cv2 pipeline, which need the WIDER images (out of scope)."""

    def __init__(self, img_scale=(640, 640), samples_per_gpu=16, iters_per_epoch=403, rank=0,
                 max_gt=synthetic.MAX_GT, **_):
        self.h, self.w = img_scale[1], img_scale[0]
        self.bs, self.iters_per_epoch, self.rank, self.max_gt = \
            samples_per_gpu, iters_per_epoch, rank, max_gt

    def batch(self, it, device=None):
        b = synthetic.make_batch(self.bs, self.h, self.w, synthetic.batch_seed(self.rank, it),
                                 self.max_gt)
        return synthetic.to_device(b, device) if device is not None else b


class SyntheticSourceImages:
    """Synthetic DECODED sources (uint8 HWC at WIDER-like sizes, ragged GT) kept resident on the
    device and pushed through the reference's train pipeline on the GPU (pipelines.DevicePipeline:
    RandomSquareCrop -> Resize -> RandomFlip -> collate) every iteration -- the role of
    RetinaFaceDataset + the cv2 worker processes, minus image decoding."""

    def __init__(self, pipeline, samples_per_gpu=16, iters_per_epoch=403, rank=0, pool=64,
                 src_hw=((768, 1024), (1024, 683), (500, 375), (683, 1024)), max_gt=synthetic.MAX_GT,
                 seed=0, **_):
        from .pipelines import DevicePipeline
        self.pipe = DevicePipeline(pipeline, seed=seed + 7919 * rank, gmax=64 if max_gt <= 64 else 128)
        self.bs, self.iters_per_epoch, self.rank = samples_per_gpu, iters_per_epoch, rank
        self.pool, self.src_hw, self.max_gt, self.seed = pool, src_hw, max_gt, seed
        self._src = None

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
                                      [kps[i] for i in idx], device)

    def batch(self, it, device=None):
        if device is None:
            raise RuntimeError('SyntheticSourceImages augments on the GPU: a device is required')
        if self._src is None:
            self._src = self._build(device)
        return self.pipe(self._src, it)


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


def train_detector(model, dataset, cfg, distributed=False, validate=False, timestamp=None,
                   meta=None, max_iters=None, device='cuda', log=print):
    """mmdet/apis/train.py:117 surface.  Returns the logged history: one dict of python floats
    per logging interval (every iteration for short runs with max_iters <= 64)."""
    rank, world = get_dist_info()
    model = model.to(device)
    model.train()
    if distributed:
        model = build_ddp(model, device, device_ids=[torch.cuda.current_device()],
                          broadcast_buffers=False)
    optimizer = build_optimizer(model, cfg.optimizer)
    sched = StepLrWarmup(cfg.optimizer['lr'], **cfg.lr_config)
    max_epochs = cfg.runner['max_epochs']
    interval = cfg.log_config['interval'] if 'log_config' in cfg else 50
    ck_interval = cfg.checkpoint_config['interval'] if 'checkpoint_config' in cfg else 0
    work_dir = cfg.get('work_dir')
    start_epoch, it = 0, 0
    if cfg.get('resume_from'):
        m = load_checkpoint(model, cfg.resume_from, optimizer)
        start_epoch, it = m.get('epoch', 0), m.get('iter', 0)
    elif cfg.get('load_from'):
        load_checkpoint(model, cfg.load_from)
    history = []
    t0 = time.time()
    for epoch in range(start_epoch, max_epochs):
        for _ in range(dataset.iters_per_epoch):
            lr = sched.lr_at(epoch, it)
            for g in optimizer.param_groups:
                g['lr'] = lr
            batch = dataset.batch(it, device)
            out = model.train_step(batch, optimizer)
            optimizer.zero_grad()
            out['loss'].backward()
            optimizer.step()
            it += 1
            if it % interval == 0 or (max_iters is not None and max_iters <= 64):
                # log_vars are lazily copied device scalars backed by a 256-row staging ring:
                # freeze them into python floats when they are logged, never keep the lazy objects
                lv = {k: float(v) for k, v in out['log_vars'].items()}
                history.append(dict(lv, iter=it, epoch=epoch + 1, lr=lr))
            if rank == 0 and it % interval == 0:
                dt = (time.time() - t0) / interval
                t0 = time.time()
                log(f'Epoch [{epoch + 1}][{it}] lr: {lr:.3e}, time: {dt:.4f}, ' +
                    ', '.join(f'{k}: {v:.4f}' for k, v in lv.items()))
            if max_iters is not None and it >= max_iters:
                return history
        if rank == 0 and work_dir and ck_interval and (epoch + 1) % ck_interval == 0:
            os.makedirs(work_dir, exist_ok=True)
            save_checkpoint(model, optimizer, os.path.join(work_dir, f'epoch_{epoch + 1}.pth'),
                            dict(meta or {}, epoch=epoch + 1, iter=it))
    return history
