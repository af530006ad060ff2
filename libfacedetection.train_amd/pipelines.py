"""The reference's TRAIN input pipeline (configs/yunet_n.py:36-56) as a device stage.

    pipe = DevicePipeline(cfg.data.train.pipeline, seed=0, gmax=64)   # the reference's own list
    src = SourceBatch.from_lists(images_u8_hwc, gt_bboxes, gt_keypointss, device)
    batch = pipe(src, iteration)        # dict(img, img_metas, gt_bboxes, gt_labels, gt_keypointss)
    out = model.train_step(batch, optimizer)

`RandomSquareCrop -> Resize(keep_ratio=False) -> RandomFlip -> Normalize(0, 1) ->
DefaultFormatBundle -> Collect` run as two HIP kernels (csrc/augment.hip); the classes below carry
the configuration under the reference's registry names (mmdet/datasets/pipelines/transforms.py,
formatting.py, loading.py) so the reference's config files build unchanged.  Decoding image files
(LoadImageFromFile) is outside this stage: sources arrive as uint8 HWC arrays.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .builder import PIPELINES
from .registry import build_from_cfg
from .synthetic import GTList


class _Carrier:
    """Configuration carrier: the arithmetic lives in the fused device stage."""

    def __init__(self, **kw):
        self.cfg = dict(kw)
        for k, v in kw.items():
            setattr(self, k, v)

    def __call__(self, results):
        raise NotImplementedError(
            f'{type(self).__name__} is executed inside DevicePipeline (HIP kernels); there is no '
            'per-sample CPU implementation in this framework')

    def __repr__(self):
        return f'{type(self).__name__}({self.cfg})'


@PIPELINES.register_module()
class LoadImageFromFile(_Carrier):
    def __init__(self, to_float32=False, color_type='color', file_client_args=None):
        super().__init__(to_float32=to_float32, color_type=color_type)


@PIPELINES.register_module()
class LoadAnnotations(_Carrier):
    def __init__(self, with_bbox=True, with_label=True, with_keypoints=False, with_mask=False,
                 with_seg=False, **kw):
        super().__init__(with_bbox=with_bbox, with_label=with_label, with_keypoints=with_keypoints)


@PIPELINES.register_module()
class RandomSquareCrop(_Carrier):
    """transforms.py:975-1169."""

    def __init__(self, crop_ratio_range=None, crop_choice=None, bbox_clip_border=True):
        if crop_choice is None or crop_ratio_range is not None:
            raise NotImplementedError('only RandomSquareCrop(crop_choice=[...]) (the shipped configs)')
        if not bbox_clip_border:
            raise NotImplementedError('bbox_clip_border=False')
        super().__init__(crop_choice=[float(c) for c in crop_choice])


@PIPELINES.register_module()
class Resize(_Carrier):
    """transforms.py:52-330, the keep_ratio=False / single-scale case of the train configs."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True,
                 bbox_clip_border=True, backend='cv2', interpolation='bilinear', override=False):
        if isinstance(img_scale, list):
            if len(img_scale) != 1:
                raise NotImplementedError('multi-scale Resize')
            img_scale = img_scale[0]
        if keep_ratio or ratio_range is not None or img_scale is None or img_scale[0] != img_scale[1] \
                or interpolation != 'bilinear' or not bbox_clip_border:
            raise NotImplementedError('only Resize(img_scale=(S, S), keep_ratio=False, bilinear)')
        super().__init__(img_scale=(int(img_scale[0]), int(img_scale[1])))


@PIPELINES.register_module()
class RandomFlip(_Carrier):
    """transforms.py:378-546 (horizontal)."""

    def __init__(self, flip_ratio=None, direction='horizontal'):
        if direction != 'horizontal' or not isinstance(flip_ratio, float):
            raise NotImplementedError('only RandomFlip(flip_ratio=<float>, direction="horizontal")')
        super().__init__(flip_ratio=flip_ratio)


@PIPELINES.register_module()
class Normalize(_Carrier):
    def __init__(self, mean, std, to_rgb=True):
        if any(float(m) != 0.0 for m in mean) or any(float(s) != 1.0 for s in std) or to_rgb:
            raise NotImplementedError('the YuNet configs feed raw 0-255 BGR (mean 0, std 1, to_rgb=False)')
        super().__init__(mean=list(mean), std=list(std), to_rgb=to_rgb)


@PIPELINES.register_module()
class DefaultFormatBundle(_Carrier):
    def __init__(self, **kw):
        super().__init__()


@PIPELINES.register_module()
class Collect(_Carrier):
    def __init__(self, keys, meta_keys=None):
        super().__init__(keys=list(keys))


class DeviceGT(GTList):
    """GT of a device-augmented batch: `padded` [N, Gmax, ...] and `counts` [N] live on the device
    (what the loss step stages directly); the list items are the padded per-image views -- rows at
    or beyond counts[i] are zero."""


class SourceBatch:
    """A batch of decoded source images and their annotations, resident on the device:
    src uint8 (concatenated HWC images), src_off int64 [N], src_hw int32 [N,2],
    boxes fp32 [sum G,4], kps fp32 [sum G,5,3], gt_off int32 [N+1]."""

    def __init__(self, src, src_off, src_hw, boxes, kps, gt_off):
        self.src, self.src_off, self.src_hw = src, src_off, src_hw
        self.boxes, self.kps, self.gt_off = boxes, kps, gt_off
        self.n = int(src_hw.shape[0])

    @classmethod
    def from_lists(cls, images, gt_bboxes, gt_keypointss, device):
        """images: list of uint8 [h, w, 3] arrays / tensors (BGR as decoded); gt lists per image."""
        imgs = [np.ascontiguousarray(np.asarray(im.cpu() if torch.is_tensor(im) else im, dtype=np.uint8))
                for im in images]
        for im in imgs:
            if im.ndim != 3 or im.shape[2] != 3:
                raise ValueError('source images must be uint8 [h, w, 3]')
        sizes = np.array([im.size for im in imgs], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        src = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs]))
        hw = torch.tensor([[im.shape[0], im.shape[1]] for im in imgs], dtype=torch.int32)
        cnt = [int(np.asarray(b).shape[0]) for b in gt_bboxes]
        goff = torch.tensor(np.concatenate([[0], np.cumsum(cnt)]), dtype=torch.int32)
        tot = max(1, sum(cnt))
        boxes = torch.zeros(tot, 4)
        kps = torch.zeros(tot, 5, 3)
        if sum(cnt):
            boxes[:sum(cnt)] = torch.cat([torch.as_tensor(np.asarray(b), dtype=torch.float32).reshape(-1, 4)
                                          for b in gt_bboxes])
            kps[:sum(cnt)] = torch.cat([torch.as_tensor(np.asarray(k), dtype=torch.float32).reshape(-1, 5, 3)
                                        for k in gt_keypointss])
        d = torch.device(device)
        return cls(src.to(d), torch.from_numpy(off).to(d), hw.to(d), boxes.to(d), kps.to(d), goff.to(d))


class DevicePipeline:
    """Builds from the reference's pipeline list and runs it as two kernel launches per batch."""

    ORDER = ['LoadImageFromFile', 'LoadAnnotations', 'RandomSquareCrop', 'Resize', 'RandomFlip',
             'Normalize', 'DefaultFormatBundle', 'Collect']

    def __init__(self, pipeline, seed=0, gmax=64, pad_value=128.0, max_attempts=250, max_retries=64):
        steps = [build_from_cfg(p, PIPELINES) if isinstance(p, dict) else p for p in pipeline]
        names = [type(s).__name__ for s in steps]
        if names != self.ORDER:
            raise NotImplementedError(f'DevicePipeline implements exactly {self.ORDER}; got {names}')
        self.steps = steps
        by = dict(zip(names, steps))
        if not by['LoadAnnotations'].with_keypoints:
            raise NotImplementedError('LoadAnnotations(with_keypoints=True) is required')
        self.out_size = by['Resize'].img_scale[0]
        if self.out_size % 32:
            raise ValueError('Resize img_scale must be a multiple of 32 for the YuNet stack')
        choice = by['RandomSquareCrop'].crop_choice
        if not 1 <= len(choice) <= 8:
            raise NotImplementedError('crop_choice must have 1..8 entries')
        cfg = L.YunetAugCfg()
        cfg.out_size, cfg.n_choice = self.out_size, len(choice)
        for i, c in enumerate(choice):
            cfg.crop_choice[i] = c
        cfg.flip_ratio, cfg.pad_value, cfg.seed = by['RandomFlip'].flip_ratio, pad_value, seed & 0xFFFFFFFF
        cfg.max_attempts, cfg.max_retries, cfg.gmax = max_attempts, max_retries, gmax
        self.cfg = cfg
        self.gmax = gmax
        self.params = None

    def __call__(self, src, iteration):
        lib = L.load()
        n, S, dev = src.n, self.out_size, src.src.device
        if dev.type != 'cuda':
            raise RuntimeError('DevicePipeline needs device-resident sources: HIP kernels only, no CPU fallback')
        img = torch.empty(n, 3, S, S, device=dev, dtype=torch.float32)
        gb = torch.empty(n, self.gmax, 4, device=dev, dtype=torch.float32)
        gk = torch.empty(n, self.gmax, 5, 3, device=dev, dtype=torch.float32)
        cnt = torch.empty(n, device=dev, dtype=torch.int32)
        params = torch.empty(n, 8, device=dev, dtype=torch.int32)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        L.check(lib.yunet_aug_decide(p(src.src_hw), p(src.boxes), p(src.kps), p(src.gt_off), C.byref(self.cfg),
                                     int(iteration) & 0xFFFFFFFF, n, p(params), p(gb), p(gk), p(cnt), stream),
                'yunet_aug_decide')
        L.check(lib.yunet_aug_pixels(p(src.src), p(src.src_off), p(src.src_hw), p(params), C.byref(self.cfg), n,
                                     p(img), stream), 'yunet_aug_pixels')
        self.params = params
        boxes, kps = DeviceGT(list(gb)), DeviceGT(list(gk))
        boxes.padded, boxes.counts = gb, cnt
        kps.padded, kps.counts = gk, cnt
        labels = GTList([torch.zeros(self.gmax, dtype=torch.int64, device=dev)] * n)
        metas = [dict(img_shape=(S, S, 3), pad_shape=(S, S, 3), batch_input_shape=(S, S)) for _ in range(n)]
        return dict(img=img, img_metas=metas, gt_bboxes=boxes, gt_labels=labels, gt_keypointss=kps)

    def check(self):
        """Synchronising status check of the last batch: raises like the reference would misbehave
        (it loops forever on an image whose boxes no crop window can contain)."""
        st = self.params[:, 6].cpu()
        bad = (st == 1).nonzero().flatten().tolist()
        if bad:
            raise ValueError(f'RandomSquareCrop found no window containing a box centre for images {bad} '
                             '(images without GT must be filtered by the dataset, retinaface.py)')
        return (st == 2).nonzero().flatten().tolist()       # images whose GT was truncated to gmax
