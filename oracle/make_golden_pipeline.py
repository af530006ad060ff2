#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/pipeline_*.npz by running the UNMODIFIED
reference transform classes (RandomSquareCrop -> Resize -> RandomFlip -> Normalize,
configs/yunet_n.py:36-48) on seeded synthetic images, with numpy's random draws redirected to the
counter-based generator of oracle/pipeline_oracle.py (so the device path, which uses the same
generator, can be compared bit for bit).

    python oracle/make_golden_pipeline.py            # needs /root/reference

What the fixtures pin: crop windows, kept-box masks, transformed boxes / keypoints, flip flags
(all produced by the reference's own code).  The image is produced with `mmcv.imresize` bound to
pipeline_oracle.resize_linear because cv2 is absent here -- image pixels are therefore NOT an
independent check of the Resize interpolation (see the header of pipeline_oracle.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pipeline_oracle as P   # noqa: E402
import ref_stub               # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
CROP_CHOICE = [0.5, 0.7, 0.9, 1.1, 1.3, 1.5]


class RedirectedRandom:
    """np.random.choice / randint / random_sample served from a pipeline_oracle.Stream."""

    def __init__(self):
        self.st = None
        self.saved = None

    def __enter__(self):
        self.saved = (np.random.choice, np.random.randint, np.random.random_sample)
        np.random.choice, np.random.randint = self.choice, self.randint
        np.random.random_sample = self.random_sample
        return self

    def __exit__(self, *a):
        np.random.choice, np.random.randint, np.random.random_sample = self.saved

    def choice(self, a, size=None, replace=True, p=None):
        assert size is None
        if p is None:
            return a[self.st.choice_index(len(a))]
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return a[int(cdf.searchsorted(self.st.uniform(), side='right'))]   # numpy's own algorithm

    def randint(self, low, high=None, size=None):
        assert size is None and high is not None
        return self.st.randint(int(low), int(high))

    def random_sample(self, size=None):
        assert size is None
        return self.st.uniform()


def run_reference(T, imgs, boxes, kps, seed, iteration, S):
    crop = T.RandomSquareCrop(crop_choice=CROP_CHOICE)
    resize = T.Resize(img_scale=(S, S), keep_ratio=False)
    flip = T.RandomFlip(flip_ratio=0.5)
    norm = T.Normalize(mean=[0., 0., 0.], std=[1., 1., 1.], to_rgb=False)
    out = []
    with RedirectedRandom() as rr:
        for i, (im, b, k) in enumerate(zip(imgs, boxes, kps)):
            rr.st = P.Stream(seed, iteration, i)
            res = dict(img=im.astype(np.float32), img_shape=im.shape, ori_shape=im.shape,
                       img_fields=['img'], bbox_fields=['gt_bboxes'], keypoints_fields=['gt_keypointss'],
                       gt_bboxes=b.copy(), gt_labels=np.zeros(len(b), np.int64), gt_keypointss=k.copy())
            res = crop(res)
            cw = res['img'].shape[0]
            res = resize(res)
            res = flip(res)
            res = norm(res)
            out.append(dict(img=np.ascontiguousarray(res['img'].transpose(2, 0, 1)).astype(np.float32),
                            boxes=res['gt_bboxes'].astype(np.float32),
                            kps=res['gt_keypointss'].astype(np.float32),
                            labels=res['gt_labels'], cw=cw, flip=bool(res['flip']),
                            draws=rr.st.ctr))
    return out


def main():
    if not ref_stub.available():
        raise SystemExit('needs the reference tree')

    def imresize(img, size, return_scale=False, interpolation='bilinear', out=None, backend=None):
        h, w = img.shape[:2]
        assert size[0] == size[1] and interpolation == 'bilinear'
        r = P.resize_linear(img, size[0])
        return (r, size[0] / w, size[1] / h) if return_scale else r

    def imflip(img, direction='horizontal'):
        assert direction == 'horizontal'
        return np.flip(img, axis=1)

    T = ref_stub.load_pipeline_transforms(imresize=imresize, imflip=imflip)
    os.makedirs(OUT, exist_ok=True)
    for name, seed, iteration, S, shapes in [
            ('pipeline_s160', 7, 0, 160, [(120, 200, 3), (333, 250, 9), (97, 97, 1), (480, 640, 24),
                                          (400, 600, -1), (600, 400, -1), (300, 300, -2)]),
            ('pipeline_s320', 11, 5, 320, [(768, 1024, 40), (500, 375, 2), (1024, 683, 64), (240, 320, 5),
                                           (333, 500, 1), (1024, 1024, 17)])]:
        rng = np.random.default_rng(seed)
        imgs, boxes, kps = zip(*[P.synth_image(rng, h, w, g) for h, w, g in shapes])
        ref = run_reference(T, imgs, boxes, kps, seed, iteration, S)
        pack = dict(seed=seed, iteration=iteration, S=S, n=len(shapes),
                    crop_choice=np.array(CROP_CHOICE, np.float64))
        for i, r in enumerate(ref):
            pack[f'src_shape_{i}'] = np.array(imgs[i].shape[:2] + (int(imgs[i].astype(np.int64).sum()),), np.int64)
            pack[f'src_g_{i}'] = np.int64(shapes[i][2])
            pack[f'src_boxes_{i}'] = boxes[i]
            pack[f'src_kps_{i}'] = kps[i]
            pack[f'img_{i}'] = r['img']
            pack[f'boxes_{i}'] = r['boxes']
            pack[f'kps_{i}'] = r['kps']
            pack[f'meta_{i}'] = np.array([r['cw'], int(r['flip']), r['draws'], len(r['boxes'])], np.int64)
        # keep the fixture small: the uint8 sources are regenerated from the seed by
        # pipeline_oracle.synth_image (shape + byte sum stored to detect generator drift); of the
        # output image a digest (sum / sum of squares per channel, float64) and two 16x16 windows
        for i, r in enumerate(ref):
            im = pack.pop(f'img_{i}')
            pack[f'img_digest_{i}'] = np.stack([im.astype(np.float64).sum((1, 2)),
                                                (im.astype(np.float64) ** 2).sum((1, 2))])
            pack[f'img_corner_{i}'] = im[:, :16, :16].copy()
            pack[f'img_center_{i}'] = im[:, S // 2 - 8:S // 2 + 8, S // 2 - 8:S // 2 + 8].copy()
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **pack)
        print(name, [(int(r['cw']), r['flip'], r['draws'], len(r['boxes'])) for r in ref])


if __name__ == '__main__':
    main()
