"""RetinaFace / WIDER-Face `labelv2` annotations and decoded image sources.

Reference: mmdet/datasets/retinaface.py:17-150 (`RetinaFaceDataset`: `_parse_ann_line`,
`load_annotations`, `get_ann_info`) on the text format

    # <relative image path> <width> <height>
    x1 y1 x2 y2 [ 5 x (kx ky kflag) [score] | ignore-flag ]

The reference feeds every sample through cv2 worker processes (LoadImageFromFile + the
augmentation pipeline on the CPU).  Here the dataset only DECODES (PIL; BGR channel order like
cv2.imread) and hands decoded uint8 sources to `pipelines.DevicePipeline`, which runs
RandomSquareCrop -> Resize -> RandomFlip -> collate as HIP kernels
(`RetinaFaceSource.batch`, the counterpart of `runner.SyntheticSourceImages`).
"""
import os

import numpy as np
import torch

from .builder import DATASETS

NK = 5


def parse_ann_line(line, min_size=None, test_mode=False):
    """One annotation row -> dict(bbox [4], kps [5,3], ignore, cat)  (retinaface.py:29-56).
    Landmark rows are (x, y, flag): all -1 -> weight 0, else weight 1; a row with a single
    fifth value carries an ignore flag instead of landmarks."""
    values = [float(x) for x in line.strip().split()]
    bbox = np.array(values[0:4], dtype=np.float32)
    kps = np.zeros((NK, 3), dtype=np.float32)
    ignore = False
    if min_size is not None:
        if test_mode:
            raise AssertionError('min_size is a training-time filter')
        if bbox[2] - bbox[0] < min_size or bbox[3] - bbox[1] < min_size:
            ignore = True
    if len(values) > 5:
        kps = np.array(values[4:19], dtype=np.float32).reshape(NK, 3)
        for row in kps:
            if (row == -1).all():
                row[2] = 0.0
            else:
                if row[2] < 0:
                    raise AssertionError(f'negative landmark flag in {line!r}')
                row[2] = 1.0
    elif len(values) == 5:
        ignore = ignore or values[4] == 1
    elif not test_mode:
        raise AssertionError('boxes without landmarks / flags are test annotations')
    return dict(bbox=bbox, kps=kps, ignore=ignore, cat='FG')


def load_labelv2(ann_file, min_size=None, test_mode=False):
    """-> [dict(filename, width, height, objs=[...])]  (retinaface.py:58-99).  Images without
    any object are dropped in training mode."""
    infos, cur = [], None
    with open(ann_file) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith('#'):
                name, w, h = line[1:].strip().split()[:3]
                cur = dict(filename=name, width=int(w), height=int(h), objs=[])
                infos.append(cur)
                continue
            if cur is None:
                raise AssertionError(f'{ann_file}: annotation row before the first "# image" header')
            cur['objs'].append(parse_ann_line(line, min_size, test_mode))
    # the reference keys images by name: a repeated header REPLACES the earlier entry in place
    by_name = {}
    for it in infos:
        by_name[it['filename']] = it
    infos = list(by_name.values())
    if not test_mode:
        infos = [it for it in infos if it['objs']]
    return infos


def ann_info(info):
    """retinaface.py:101-150: kept / ignored boxes, labels, landmarks of one image."""
    keep = [o for o in info['objs'] if not o['ignore']]
    ign = [o for o in info['objs'] if o['ignore']]
    return dict(
        bboxes=np.array([o['bbox'] for o in keep], dtype=np.float32).reshape(-1, 4),
        labels=np.zeros(len(keep), dtype=np.int64),
        keypointss=np.array([o['kps'] for o in keep], dtype=np.float32).reshape(-1, NK, 3),
        bboxes_ignore=np.array([o['bbox'] for o in ign], dtype=np.float32).reshape(-1, 4),
        labels_ignore=np.zeros(len(ign), dtype=np.int64))


def imread_bgr(path):
    """cv2.imread(path, IMREAD_COLOR) stand-in: uint8 [h, w, 3], BGR.  Like cv2 (and mmcv.imread's
    'color' flag) the EXIF orientation tag is APPLIED, so a rotated JPEG decodes in the frame its
    annotations were made in."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im)
        return np.ascontiguousarray(np.asarray(im.convert('RGB'), dtype=np.uint8)[:, :, ::-1])


@DATASETS.register_module()
class RetinaFaceDataset:
    """configs/yunet_n.py:28-35,87-100: dict(type='RetinaFaceDataset', ann_file=..., img_prefix=...,
    pipeline=[...], min_size=...)."""
    CLASSES = ('FG',)

    def __init__(self, ann_file, img_prefix='', pipeline=None, min_size=None, test_mode=False,
                 gt_path=None, **_):
        self.ann_file, self.img_prefix, self.pipeline_cfg = ann_file, img_prefix, pipeline
        self.min_size, self.test_mode, self.gt_path = min_size, test_mode, gt_path
        self.NK = NK
        self.cat2label = {c: i for i, c in enumerate(self.CLASSES)}
        self.data_infos = load_labelv2(ann_file, min_size, test_mode)
        if not test_mode:
            # CustomDataset._filter_imgs (mmdet/datasets/custom.py:119-122, 176-185): training drops
            # images whose shorter side is below 32 px
            self.data_infos = [it for it in self.data_infos if min(it['width'], it['height']) >= 32]
        self.flag = np.array([1 if it['width'] / it['height'] > 1 else 0 for it in self.data_infos],
                             dtype=np.uint8)      # CustomDataset._set_group_flag

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, idx):
        return ann_info(self.data_infos[idx])

    def load_image(self, idx):
        return imread_bgr(os.path.join(self.img_prefix, self.data_infos[idx]['filename']))

    def evaluate(self, results, metric='mAP', logger=None, iou_thr=0.5, **_):
        """CustomDataset.evaluate (mmdet/datasets/custom.py:310-367) for the metric the shipped configs ask
        for (`evaluation = dict(interval=..., metric='mAP')`): AP of the face class at each IoU threshold."""
        from collections import OrderedDict
        from .evaluation import eval_map_single_class
        if not isinstance(metric, str):
            assert len(metric) == 1
            metric = metric[0]
        if metric != 'mAP':
            raise KeyError(f'metric {metric} is not supported')
        anns = [self.get_ann_info(i) for i in range(len(results))]
        thrs = [iou_thr] if isinstance(iou_thr, float) else list(iou_thr)
        res, aps = OrderedDict(), []
        for t in thrs:
            ap, _ = eval_map_single_class(results, anns, t)
            aps.append(ap)
            res[f'AP{int(t * 100):02d}'] = round(ap, 3)
        res['mAP'] = sum(aps) / len(aps)
        return res

    def __getitem__(self, idx):
        """Decoded sample (what LoadImageFromFile + LoadAnnotations deliver, before augmentation)."""
        info = self.data_infos[idx]
        ann = self.get_ann_info(idx)
        return dict(img=self.load_image(idx), filename=info['filename'],
                    ori_shape=(info['height'], info['width'], 3), gt_bboxes=ann['bboxes'],
                    gt_labels=ann['labels'], gt_keypointss=ann['keypointss'],
                    gt_bboxes_ignore=ann['bboxes_ignore'])


class RetinaFaceSource:
    """Training data source over a RetinaFaceDataset: iteration `it` of rank r is batch it % iters_per_epoch of the
    reference's DistributedGroupSampler(dataset, samples_per_gpu, world, r, seed) in epoch it // iters_per_epoch
    (samplers.py; the order `tools/dist_train.sh` feeds, also used at world size 1, where the reference's
    GroupSampler draws from numpy's global generator and is not reproducible).  Samples are decoded with PIL on
    the host and augmented on the GPU by the config's own pipeline (pipelines.DevicePipeline)."""

    def __init__(self, dataset, pipeline, samples_per_gpu=16, rank=0, world=1, seed=0, max_gt=64, workers=4):
        from .pipelines import DevicePipeline
        self.ds, self.bs, self.rank, self.world, self.seed = dataset, samples_per_gpu, rank, world, seed
        self.pipe = DevicePipeline(pipeline, seed=seed + 7919 * rank, gmax=64 if max_gt <= 64 else 128)
        from .samplers import DistributedGroupSampler
        self.sampler = DistributedGroupSampler(dataset, samples_per_gpu, world, rank, seed=seed)
        self.iters_per_epoch = max(1, len(self.sampler) // samples_per_gpu)
        self._perm_epoch, self._perm = None, None
        # decode ahead: the samples of iteration it + 1 are decoded by a small thread pool (PIL releases
        # the GIL while decoding) while the GPU runs iteration it -- the role of the reference's
        # DataLoader workers (workers_per_gpu, mmdet/datasets/builder.py:87-190)
        self.workers = max(0, int(workers))
        self._pool = None
        self._ahead = {}

    def _indices(self, it):
        epoch, k = divmod(it, self.iters_per_epoch)
        if self._perm_epoch != epoch:
            self.sampler.set_epoch(epoch)
            self._perm = list(iter(self.sampler))
            self._perm_epoch = epoch
        return self._perm[k * self.bs:(k + 1) * self.bs]

    def _decoded(self, it):
        if self.workers == 0:
            return [self.ds[i] for i in self._indices(it)]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(self.workers)
        for k in (it, it + 1):                              # this iteration (if not already queued) and the next
            if k not in self._ahead:
                self._ahead[k] = [self._pool.submit(self.ds.__getitem__, i) for i in self._indices(k)]
        futs = self._ahead.pop(it)
        for k in [k for k in self._ahead if k < it]:        # a caller that jumps around: drop stale work
            del self._ahead[k]
        return [f.result() for f in futs]

    def batch(self, it, device=None):
        if device is None:
            raise RuntimeError('RetinaFaceSource augments on the GPU: a device is required')
        from .pipelines import SourceBatch
        samples = self._decoded(it)
        src = SourceBatch.from_lists([s['img'] for s in samples], [s['gt_bboxes'] for s in samples],
                                     [s['gt_keypointss'] for s in samples], device)
        return self.pipe(src, it)
