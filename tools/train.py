#!/usr/bin/env python
"""Train YuNet on MI355X -- same command line as the reference's tools/train.py
(tools/train.py:24-104): CONFIG [--work-dir] [--resume-from] [--seed] [--deterministic]
[--cfg-options k=v ...] [--launcher {none,pytorch,slurm,mpi}] [--local_rank].

Multi-GPU (one process per GPU, RCCL):
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
      tools/train.py configs/yunet_n.py --launcher pytorch
"""
import argparse
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import yunet_amd  # noqa: E402
from yunet_amd import runner as R  # noqa: E402
from yunet_amd.parallel import get_dist_info, init_dist  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='Train a detector')
    p.add_argument('config')
    p.add_argument('--work-dir')
    p.add_argument('--resume-from')
    p.add_argument('--auto-resume', action='store_true')
    p.add_argument('--no-validate', action='store_true')
    p.add_argument('--gpu-id', type=int, default=0)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--diff-seed', action='store_true')
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--cfg-options', nargs='+', default=[])
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', '--local-rank', type=int, default=0)
    p.add_argument('--max-iters', type=int, default=None, help='stop early (smoke runs)')
    a = p.parse_args()
    os.environ.setdefault('LOCAL_RANK', str(a.local_rank))
    return a


def main():
    args = parse_args()
    cfg = yunet_amd.Config.fromfile(args.config)
    opts = {}
    for kv in args.cfg_options:
        k, v = kv.split('=', 1)
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
        opts[k] = v
    cfg.merge_from_dict(opts)
    cfg['work_dir'] = args.work_dir or cfg.get('work_dir') or os.path.join(
        './work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    if args.resume_from:
        cfg['resume_from'] = args.resume_from
    if args.auto_resume:                      # tools/train.py:118 -> mmdet/apis/train.py:236-240
        cfg['auto_resume'] = True
    distributed = args.launcher != 'none'
    if distributed:
        init_dist(args.launcher, **cfg.get('dist_params', dict(backend='nccl')))
    else:
        torch.cuda.set_device(args.gpu_id)
    rank, world = get_dist_info()
    seed = args.seed if args.seed is not None else 0
    torch.manual_seed(seed + (rank if args.diff_seed else 0))
    model = yunet_amd.build_detector(cfg.model)
    model.init_weights()
    dcfg = cfg.data.train
    kw = {k: v for k, v in dcfg.items() if k != 'type'}
    if dcfg.get('type') == 'SyntheticWiderFace':
        ds = R.SyntheticWiderFace(samples_per_gpu=cfg.data.samples_per_gpu, rank=rank, **kw)
    elif dcfg.get('type') == 'SyntheticSourceImages':      # device-side reference pipeline
        ds = R.SyntheticSourceImages(samples_per_gpu=cfg.data.samples_per_gpu, rank=rank, seed=seed, **kw)
    elif dcfg.get('type') == 'RetinaFaceDataset':          # labelv2 annotations + image files (PIL decode)
        from yunet_amd.datasets import RetinaFaceSource
        dataset = yunet_amd.build_dataset(dcfg)
        ds = RetinaFaceSource(dataset, dcfg['pipeline'], samples_per_gpu=cfg.data.samples_per_gpu, rank=rank,
                              world=world, seed=seed)
    else:
        raise SystemExit('data sources: RetinaFaceDataset (labelv2 + image files, augmented on the GPU), '
                         'SyntheticWiderFace (ready batches) or SyntheticSourceImages (decoded synthetic '
                         'sources + the reference train pipeline on the GPU)')
    meta = dict(config=args.config, seed=seed, CLASSES=('face',))
    R.train_detector(model, ds, cfg, distributed=distributed, validate=not a.no_validate, meta=meta,
                     max_iters=args.max_iters)


if __name__ == '__main__':
    main()
