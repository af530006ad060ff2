#!/usr/bin/env python
"""WIDER-Face evaluation of a checkpoint -- same command line as the reference's
tools/test_widerface.py (CONFIG CHECKPOINT [--out DIR] [--save-preds] [--thr T] [--mode M]):

    mode 0 (640, 640) | 1 (1100, 1650) | 2 origin size, padded to a multiple of 32 | >30 (mode, mode)

Per image: PIL decode (BGR) -> keep-ratio bilinear resize on the GPU -> zero pad -> eval forward +
get_bboxes with rescale=True (HIP kernels) -> x y w h score rows -> wider_evaluation (easy / medium /
hard AP at IoU 0.5).  `--eval-only PRED_DIR` skips inference and scores saved prediction files.

Not pinned against the reference: its Resize runs cv2's uint8 fixed-point bilinear on the CPU
(cv2 is not installed here); this tool resizes in fp32, results can differ in the last grey level.
Mode 1 pads 1100 x 1650 up to 1120 x 1664 (the network needs multiples of 32).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

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


def prepare(img_bgr, scale, device):
    """uint8 [h,w,3] -> (float32 [1,3,H,W] on the device, img_meta).  mmcv.imrescale semantics for the
    size (rescale_size: factor = min(long/long_edge, short/short_edge), rounded), Pad to the target
    (or to a multiple of 32), right / bottom, zeros."""
    h, w = img_bgr.shape[:2]
    x = torch.from_numpy(img_bgr).to(device).permute(2, 0, 1)[None].float()
    if scale is None:
        nh, nw = h, w
    else:
        f = min(max(scale) / max(h, w), min(scale) / min(h, w))
        nw, nh = int(w * float(f) + 0.5), int(h * float(f) + 0.5)
        x = F.interpolate(x, size=(nh, nw), mode='bilinear', align_corners=False)
    ph = max(nh, 0 if scale is None else scale[0] if nh <= scale[0] else nh)
    pw = max(nw, 0 if scale is None else scale[1] if nw <= scale[1] else nw)
    ph, pw = (ph + 31) // 32 * 32, (pw + 31) // 32 * 32
    x = F.pad(x, (0, pw - nw, 0, ph - nh)).contiguous()
    sf = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
    meta = dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3), scale_factor=sf,
                flip=False, flip_direction='horizontal')
    return x, meta


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
