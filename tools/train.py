#!/usr/bin/env python
"""Train YuNet on MI355X -- the command line of the reference's tools/train.py (tools/train.py:24-104):

    CONFIG [--work-dir DIR] [--resume-from CKPT] [--auto-resume] [--no-validate]
           [--gpu-id N | --gpus N | --gpu-ids N ...] [--seed S] [--diff-seed] [--deterministic]
           [--cfg-options K=V ... | --options K=V ...] [--launcher {none,pytorch,slurm,mpi}] [--local_rank R]
           [--auto-scale-lr]

Multi-GPU (one process per GPU, RCCL):
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
      tools/train.py configs/yunet_n.py --launcher pytorch
"""
import argparse
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import yunet_amd  # noqa: E402
from yunet_amd import runner as R  # noqa: E402
from yunet_amd.parallel import get_dist_info, init_dist  # noqa: E402
from yunet_amd.registry import DictAction  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train a detector')
    p.add_argument('config', help='train config file path')
    p.add_argument('--work-dir', help='the dir to save logs and models')
    p.add_argument('--resume-from', help='the checkpoint file to resume from')
    p.add_argument('--auto-resume', action='store_true', help='resume from the latest checkpoint automatically')
    p.add_argument('--no-validate', action='store_true', help='whether not to evaluate the checkpoint during training')
    g = p.add_mutually_exclusive_group()
    g.add_argument('--gpus', type=int, help='(deprecated, use --gpu-id) number of gpus (non-distributed training)')
    g.add_argument('--gpu-ids', type=int, nargs='+', help='(deprecated, use --gpu-id) ids of gpus (non-distributed training)')
    g.add_argument('--gpu-id', type=int, default=0, help='id of gpu to use (non-distributed training)')
    p.add_argument('--seed', type=int, default=None, help='random seed')
    p.add_argument('--diff-seed', action='store_true', help='different seeds for different ranks')
    p.add_argument('--deterministic', action='store_true',
                   help='accepted for compatibility: the HIP kernels of the step do not depend on a cuDNN mode')
    p.add_argument('--options', nargs='+', action=DictAction, help='(deprecated) use --cfg-options')
    p.add_argument('--cfg-options', nargs='+', action=DictAction,
                   help='override settings of the config: key=value pairs, key="[a,b]" or key=a,b for lists, '
                        'nested lists / tuples like key="[(a,b),(c,d)]"; no white space')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none', help='job launcher')
    p.add_argument('--local_rank', '--local-rank', type=int, default=0)
    p.add_argument('--auto-scale-lr', action='store_true', help='enable automatically scaling LR')
    p.add_argument('--max-iters', type=int, default=None, help='stop early (smoke runs; not in the reference)')
    a = p.parse_args(argv)
    if 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = str(a.local_rank)
    if a.options and a.cfg_options:
        raise ValueError('--options and --cfg-options cannot be both specified, '
                         '--options is deprecated in favor of --cfg-options')
    if a.options:
        warnings.warn('--options is deprecated in favor of --cfg-options')
        a.cfg_options = a.options
    return a


def prepare_config(args):
    """tools/train.py:107-160: config file + command-line overrides -> the configuration of the run."""
    cfg = yunet_amd.Config.fromfile(args.config)
    R.update_data_root(cfg)                       # MMDET_DATASETS (tools/train.py:112-113)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    if args.auto_scale_lr:
        asl = cfg.get('auto_scale_lr')
        if asl is not None and 'enable' in asl and 'base_batch_size' in asl:
            asl['enable'] = True
        else:
            warnings.warn('Can not find "auto_scale_lr" or "auto_scale_lr.enable" or '
                          '"auto_scale_lr.base_batch_size" in your configuration file.')
    # work_dir is determined in this priority: CLI > segment in file > filename
    if args.work_dir is not None:
        cfg['work_dir'] = args.work_dir
    elif cfg.get('work_dir', None) is None:
        cfg['work_dir'] = os.path.join('./work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    if args.resume_from is not None:
        cfg['resume_from'] = args.resume_from
    cfg['auto_resume'] = args.auto_resume
    if args.gpus is not None:
        cfg['gpu_ids'] = [0]
        warnings.warn('`--gpus` is deprecated because we only support single GPU mode in non-distributed '
                      'training. Use `gpus=1` now.')
    if args.gpu_ids is not None:
        cfg['gpu_ids'] = args.gpu_ids[0:1]
        warnings.warn('`--gpu-ids` is deprecated, please use `--gpu-id`. Because we only support single GPU mode '
                      'in non-distributed training. Use the first GPU in `gpu_ids` now.')
    if args.gpus is None and args.gpu_ids is None:
        cfg['gpu_ids'] = [args.gpu_id]
    return cfg


def build_source(cfg, rank, world, seed):
    """The training data source named by data.train.type (the role of build_dataset + build_dataloader)."""
    dcfg = cfg.data.train
    kw = {k: v for k, v in dcfg.items() if k != 'type'}
    if dcfg.get('type') == 'SyntheticWiderFace':
        return R.SyntheticWiderFace(samples_per_gpu=cfg.data.samples_per_gpu, rank=rank, **kw)
    if dcfg.get('type') == 'SyntheticSourceImages':      # device-side reference pipeline
        return R.SyntheticSourceImages(samples_per_gpu=cfg.data.samples_per_gpu, rank=rank, seed=seed, **kw)
    if dcfg.get('type') == 'RetinaFaceDataset':          # labelv2 annotations + image files (PIL decode)
        from yunet_amd.datasets import RetinaFaceSource
        dataset = yunet_amd.build_dataset(dcfg)
        return RetinaFaceSource(dataset, dcfg['pipeline'], samples_per_gpu=cfg.data.samples_per_gpu, rank=rank,
                                world=world, seed=seed)
    raise SystemExit('data sources: RetinaFaceDataset (labelv2 + image files, augmented on the GPU), '
                     'SyntheticWiderFace (ready batches) or SyntheticSourceImages (decoded synthetic '
                     'sources + the reference train pipeline on the GPU)')


def main(argv=None):
    args = parse_args(argv)
    cfg = prepare_config(args)
    distributed = args.launcher != 'none'
    if distributed:
        init_dist(args.launcher, **cfg.get('dist_params', dict(backend='nccl')))
        cfg['gpu_ids'] = list(range(get_dist_info()[1]))
    else:
        torch.cuda.set_device(cfg['gpu_ids'][0])
    rank, world = get_dist_info()
    os.makedirs(os.path.abspath(cfg['work_dir']), exist_ok=True)
    if rank == 0:
        cfg.dump(os.path.join(cfg['work_dir'], os.path.basename(args.config)))      # tools/train.py:171
    timestamp = time.strftime('%Y%m%d_%H%M%S', time.localtime())
    seed = args.seed if args.seed is not None else 0
    seed = seed + rank if args.diff_seed else seed
    torch.manual_seed(seed)
    cfg['seed'] = seed
    model = yunet_amd.build_detector(cfg.model)
    model.init_weights()
    ds = build_source(cfg, rank, world, args.seed if args.seed is not None else 0)
    main.last_source = ds                      # (measurement tools read the source's own timing: tools/train_e2e.py)
    meta = dict(config=cfg.pretty_text, seed=seed, exp_name=os.path.basename(args.config), CLASSES=('face',))
    hist = R.train_detector(model, ds, cfg, distributed=distributed, validate=not args.no_validate,
                            timestamp=timestamp, meta=meta, max_iters=args.max_iters)
    dump = os.environ.get('YUNET_DUMP_PARAM_SUM')
    if dump:
        # launcher tests: every rank leaves a checksum of its parameters (data-parallel ranks must end identical)
        with torch.no_grad():
            flat = torch.cat([p.detach().double().reshape(-1) for p in model.parameters()])
            text = f'{float(flat.sum())!r} {float(flat.abs().sum())!r} {float((flat * flat).sum())!r}'
        with open(os.path.join(dump, f'param_sum_rank{rank}.txt'), 'w') as f:
            f.write(text)
    return hist


if __name__ == '__main__':
    main()
