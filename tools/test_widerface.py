#!/usr/bin/env python
"""WIDER-Face evaluation of a checkpoint -- same command line as the reference's
tools/test_widerface.py (CONFIG CHECKPOINT [--out DIR] [--save-preds] [--thr T] [--mode M]):

    mode 0 (640, 640) | 1 (1100, 1650) | 2 origin size, padded to a multiple of 32 | >30 (mode, mode)

Per image: PIL decode (BGR) -> keep-ratio resize of the uint8 image on the GPU in cv2.resize's fixed-point
arithmetic (yunet_amd/imresize.py) -> zero pad -> eval forward + get_bboxes with rescale=True (HIP kernels) ->
x y w h score rows -> wider_evaluation (easy / medium / hard AP at IoU 0.5).  `--eval-only PRED_DIR` skips
inference and scores saved prediction files.

The resize restates OpenCV's published algorithm and is tested against a per-pixel restatement of it; cv2 itself
is not installed here, so the library's own output does not pin it.
Mode 1 pads 1100 x 1650 up to 1120 x 1664 (the network needs multiples of 32).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import yunet_amd  # noqa: E402
from yunet_amd import evaluation as E  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='YuNet WIDER-Face test (and eval)')
    p.add_argument('config')
    p.add_argument('checkpoint', nargs='?')
    p.add_argument('--out', default='./work_dirs/wout')
    p.add_argument('--save-preds', action='store_true')
    p.add_argument('--thr', type=float, default=-1.)
    p.add_argument('--mode', type=int, default=0)
    p.add_argument('--local_rank', '--local-rank', type=int, default=0)      # accepted like the reference's (single process)
    p.add_argument('--gt-path', default=None, help='directory of the wider_*_val.mat files '
                   '(default: <dir of data.test.ann_file>/gt)')
    p.add_argument('--eval-only', default=None, metavar='PRED_DIR')
    p.add_argument('--max-images', type=int, default=None)
    return p.parse_args()


def target_scale(mode):
    if mode == 0:
        return (640, 640)
    if mode == 1:
        return (1100, 1650)
    if mode == 2:
        return None
    if mode > 30:
        return (mode, mode)
    raise SystemExit(f'--mode {mode}')


prepare = E.prepare_test_image      # the test pipeline on the device (shared with the EvalHook)


def main():
    a = parse_args()
    cfg = yunet_amd.Config.fromfile(a.config)
    tcfg = dict(cfg.data.test)
    gt_path = a.gt_path or os.path.join(os.path.dirname(tcfg['ann_file']), 'gt')
    if a.eval_only:
        results = E.read_predictions(a.eval_only)
    else:
        if not a.checkpoint:
            raise SystemExit('a checkpoint is required unless --eval-only is given')
        if a.thr != -1.:
            cfg.model.test_cfg.score_thr = a.thr
        dev = torch.device('cuda', 0)
        model = yunet_amd.build_detector(cfg.model)
        ck = torch.load(a.checkpoint, map_location='cpu', weights_only=False)
        model.load_state_dict(ck['state_dict'] if 'state_dict' in ck else ck, strict=True)
        model.to(dev).eval()
        tcfg['test_mode'] = True
        ds = yunet_amd.build_dataset(tcfg)
        scale = target_scale(a.mode)
        results = {}
        n = len(ds) if a.max_images is None else min(len(ds), a.max_images)
        for i in range(n):
            name = ds.data_infos[i]['filename']
            img, meta = prepare(ds.load_image(i), scale, dev)
            meta['ori_filename'] = name
            res = model(return_loss=False, rescale=True, img=[img], img_metas=[[meta]])[0][0]
            event, fn = name.split('/')[-2], name.split('/')[-1]
            stem = fn[:-4] if fn.endswith('.jpg') else os.path.splitext(fn)[0]
            xywh = res.copy()
            xywh[:, 2] -= xywh[:, 0]
            xywh[:, 3] -= xywh[:, 1]
            results.setdefault(event, {})[stem] = xywh.astype(np.float64)
            if a.save_preds:
                E.write_predictions(a.out, event, stem, res)
            if (i + 1) % 200 == 0:
                print(f'[{i + 1}/{n}]')
    aps = E.wider_evaluation(results, gt_path, 0.5)
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, 'aps'), 'w') as f:
        f.write('%f,%f,%f\n' % (aps[0], aps[1], aps[2]))
    print('APS:', aps)


if __name__ == '__main__':
    main()
