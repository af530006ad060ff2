"""TEST INFRASTRUCTURE ONLY -- synthetic WIDER-Face evaluation sets.

Builds a small ground-truth set in the protocol's own file format (wider_face_val.mat +
wider_{easy,medium,hard}_val.mat, the nested cell arrays mmdet/core/evaluation/widerface.py:63-83
reads) plus a matching synthetic prediction set, so that the unmodified reference evaluation and
`libfacedetection.train_amd/evaluation.py` can be run on identical inputs
(oracle/make_golden_wider.py, tests/test_evaluation.py)."""
import os

import numpy as np


def synth_events(seed=0, n_events=3, imgs_per_event=5):
    """-> (events, pred).  events as evaluation.load_wider_gt returns them; pred in the layout
    wider_evaluation takes ({event: {image: [n,5] x y w h score, descending score}})."""
    rng = np.random.default_rng(seed)
    events, pred = [], {}
    for e in range(n_events):
        ev = dict(name=f'{e}--Event_{e}', images=[])
        pred[ev['name']] = {}
        for j in range(imgs_per_event):
            g = int(rng.integers(0, 9)) if (e + j) % 4 else 0          # some images without faces
            xy = rng.uniform(0, 500, (g, 2))
            wh = np.exp(rng.uniform(np.log(6), np.log(200), (g, 1))) * np.array([[1.0, 1.25]])
            boxes = np.round(np.concatenate([xy, wh], 1))               # integer pixel boxes like WIDER
            size = boxes[:, 2] if g else np.zeros(0)
            keep = dict(easy=np.nonzero(size >= 50)[0] + 1, medium=np.nonzero(size >= 20)[0] + 1,
                        hard=np.nonzero(size >= 8)[0] + 1)
            name = f'{e}_Event_{e}_{j}'
            ev['images'].append(dict(name=name, boxes=boxes.astype(np.float64),
                                     keep={k: v.astype(np.int64) for k, v in keep.items()}))
            # predictions: jittered copies of most GTs (some twice), plus false positives
            rows = []
            for b in boxes:
                for _ in range(int(rng.integers(0, 3))):
                    jit = b + rng.normal(0, 0.12, 4) * np.array([b[2], b[3], b[2], b[3]])
                    rows.append(list(jit) + [float(rng.uniform(0.3, 0.99))])
            for _ in range(int(rng.integers(0, 6)) if j != 2 else 0):    # one image without predictions
                rows.append(list(rng.uniform(0, 500, 2)) + list(rng.uniform(8, 120, 2)) +
                            [float(rng.uniform(0.02, 0.7))])
            arr = np.asarray(rows, dtype=np.float64).reshape(-1, 5)
            arr = arr[np.argsort(-arr[:, 4], kind='stable')]
            pred[ev['name']][name] = arr
        events.append(ev)
    return events, pred


def _cell(items):
    c = np.empty((len(items), 1), dtype=object)
    for i, v in enumerate(items):
        c[i, 0] = v
    return c


def write_mats(events, gt_dir):
    """The four protocol files, in the cell layout scipy.io.loadmat gives back for the real ones."""
    from scipy.io import savemat
    os.makedirs(gt_dir, exist_ok=True)
    savemat(os.path.join(gt_dir, 'wider_face_val.mat'), dict(
        event_list=_cell([np.array([ev['name']]) for ev in events]),
        file_list=_cell([_cell([np.array([im['name']]) for im in ev['images']]) for ev in events]),
        face_bbx_list=_cell([_cell([im['boxes'].reshape(-1, 4) for im in ev['images']]) for ev in events])))
    for k in ('easy', 'medium', 'hard'):
        savemat(os.path.join(gt_dir, f'wider_{k}_val.mat'), dict(
            gt_list=_cell([_cell([im['keep'][k].reshape(-1, 1).astype(np.int32) for im in ev['images']])
                           for ev in events])))
