#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- tests/golden/detect_*.npz from the UNMODIFIED reference detector:
`model.eval(); model.simple_test(img, img_metas)` (mmdet/models/detectors/yunet.py:53-81 ->
yunet_head.py:290-416 -> bbox2result), with the one compiled op it needs, mmcv.ops.batched_nms,
replaced by the restated greedy NMS of oracle/detect_oracle.py (see its header).

    python oracle/make_golden_detect.py            # needs /root/reference
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detect_oracle as D   # noqa: E402
import ref_stub             # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def main():
    if not ref_stub.available():
        raise SystemExit('needs the reference tree')
    ns = ref_stub.load_reference()
    head_mod = ns.mods['mmdet.models.dense_heads.yunet_head']
    det_mod = ns.mods['mmdet.models.detectors.yunet']
    head_mod.batched_nms = D.batched_nms
    det_mod.bbox2result = importlib.import_module('mmdet.core.bbox.transforms').bbox2result
    os.makedirs(OUT, exist_ok=True)
    for name, kind, size, n in [('detect_n_320', 'n', 320, 3), ('detect_s_160', 's', 160, 4)]:
        best = None
        for seed in range(40):
            arch, sd = D.make_state(kind, seed, size)
            img = D.structured_images(n, size, seed)
            flat, sizes = D.eval_flat(img, sd, arch)
            ds, di, dt = D.stability(flat, sizes, arch['strides'], 0.02, 0.45)
            cnt = [len(d) for d, _ in D.get_bboxes(flat, sizes, arch['strides'])]
            cand = int(((flat[..., 0].sigmoid() * flat[..., 5].sigmoid()) >= 0.02).sum())
            # decisions well away from their thresholds, NMS that really suppresses
            if ds > 2e-5 and di > 2e-5 and dt > 2e-7 and cand > 1.5 * sum(cnt) and min(cnt) > 20:
                best = (seed, arch, sd, img, cand, cnt)
                break
        assert best is not None, 'no stable seed found'
        seed, arch, sd, img, cand, cnt = best
        model, _ = ref_stub.build_detector(f'yunet_{kind}.py')
        missing = model.load_state_dict(sd, strict=True)
        model.eval()
        metas = [dict(img_shape=(size, size, 3), scale_factor=np.ones(4, np.float32)) for _ in range(n)]
        with torch.no_grad():
            res = model.simple_test(img, metas, rescale=False)
        pack = dict(kind=kind, size=size, n=n, seed=seed, score_thr=0.02, iou_thr=0.45,
                    candidates=cand)
        for i, r in enumerate(res):
            assert len(r) == 1
            pack[f'dets_{i}'] = r[0].astype(np.float32)
        oracle = D.get_bboxes(*D.eval_flat(img, sd, arch), arch['strides'])
        for i, (d, _) in enumerate(oracle):       # the restatement agrees with the reference run
            assert np.allclose(d.numpy(), pack[f'dets_{i}'], rtol=1e-5, atol=1e-4), i
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **pack)
        print(name, 'seed', seed, 'candidates', cand, 'kept', [len(pack[f'dets_{i}']) for i in range(n)], missing)


if __name__ == '__main__':
    main()
