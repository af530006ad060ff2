"""TEST INFRASTRUCTURE ONLY -- runs the UNMODIFIED reference WIDER-Face evaluation
(/root/reference/mmdet/core/evaluation/widerface.py: wider_evaluation) on a synthetic ground-truth /
prediction set and stores the three APs in tests/golden/wider_eval.npz.

The reference file predates numpy 1.24 (`np.float`, `np.int`): the two aliases are restored on the
numpy module for the duration of the run; the file itself is imported from where it lies.

    python oracle/make_golden_wider.py
"""
import copy
import importlib.util
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import wider_fixture as WF  # noqa: E402

REF = os.environ.get('YUNET_REFERENCE_ROOT', '/root/reference')


def reference_module():
    if not hasattr(np, 'float'):
        np.float = float
    if not hasattr(np, 'int'):
        np.int = int
    spec = importlib.util.spec_from_file_location(
        'ref_widerface', os.path.join(REF, 'mmdet', 'core', 'evaluation', 'widerface.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules['ref_widerface'] = mod      # its worker pool pickles functions by module name
    spec.loader.exec_module(mod)
    return mod


def run_reference(seed, **kw):
    events, pred = WF.synth_events(seed, **kw)
    ref = reference_module()
    with tempfile.TemporaryDirectory() as d:
        WF.write_mats(events, d)
        with np.errstate(all='ignore'):
            aps = ref.wider_evaluation(copy.deepcopy(pred), d, 0.5)
    return [float(a) for a in aps]


def real_gt_preds(gt_dir, seed=0):
    """Deterministic synthetic predictions over the REAL protocol ground truth
    (data/widerface/labelv2/val/gt): 70 % of the faces, jittered, random scores."""
    sys.path.insert(0, ROOT)
    import yunet_amd.evaluation as E
    ev = E.load_wider_gt(gt_dir)
    rng = np.random.default_rng(seed)
    pred = {}
    for e in ev:
        pred[e['name']] = {}
        for im in e['images']:
            b = im['boxes']
            k = rng.uniform(size=len(b)) < 0.7
            jit = b[k] + rng.normal(0, 0.05, (int(k.sum()), 4)) * b[k][:, [2, 3, 2, 3]]
            sc = rng.uniform(0.2, 1, len(jit))
            o = np.argsort(-sc, kind='stable')
            pred[e['name']][im['name']] = np.concatenate([jit, sc[:, None]], 1)[o]
    return ev, pred


def main():
    out = {}
    gt_dir = os.path.join(REF, 'data', 'widerface', 'labelv2', 'val', 'gt')
    if '--real' in sys.argv and os.path.isdir(gt_dir):
        _, pred = real_gt_preds(gt_dir)
        with np.errstate(all='ignore'):
            aps = reference_module().wider_evaluation(copy.deepcopy(pred), gt_dir, 0.5)
        np.savez(os.path.join(ROOT, 'tests', 'golden', 'wider_eval_real.npz'), aps=np.array([float(a) for a in aps]))
        print('real GT', aps)
        return
    for seed, kw in ((0, dict()), (1, dict(n_events=4, imgs_per_event=7)), (2, dict(n_events=2, imgs_per_event=3))):
        out[f'aps_{seed}'] = np.array(run_reference(seed, **kw))
        out[f'cfg_{seed}'] = np.array([kw.get('n_events', 3), kw.get('imgs_per_event', 5)])
        print(seed, out[f'aps_{seed}'])
    np.savez(os.path.join(ROOT, 'tests', 'golden', 'wider_eval.npz'), **out)


if __name__ == '__main__':
    main()
