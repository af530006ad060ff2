"""CPU: WIDER-Face evaluation (libfacedetection.train_amd/evaluation.py) pinned against the
unmodified reference function (mmdet/core/evaluation/widerface.py:271 wider_evaluation):
committed APs from oracle/make_golden_wider.py, a live comparison when the reference tree is
present, and properties of the protocol."""
import copy
import os

import numpy as np
import pytest

import helpers as Hh
import wider_fixture as WF


def _ours(events, pred, tmp_path=None):
    import yunet_amd.evaluation as E
    if tmp_path is not None:
        WF.write_mats(events, str(tmp_path))
        return E.wider_evaluation(copy.deepcopy(pred), str(tmp_path), 0.5)
    return E.wider_evaluation(copy.deepcopy(pred), events, 0.5)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_aps_match_reference_fixture(seed, tmp_path):
    g = Hh.load_golden('wider_eval.npz')
    ne, ni = [int(v) for v in g[f'cfg_{seed}']]
    events, pred = WF.synth_events(seed, n_events=ne, imgs_per_event=ni)
    aps = _ours(events, pred, tmp_path)          # through the .mat files, like the reference
    assert np.allclose(aps, g[f'aps_{seed}'], rtol=0, atol=1e-12), (aps, g[f'aps_{seed}'])
    assert np.allclose(_ours(events, pred), aps, rtol=0, atol=0)


@pytest.mark.skipif(not os.path.isdir('/root/reference/mmdet'), reason='reference tree not present')
def test_aps_match_live_reference():
    import make_golden_wider as MG
    import yunet_amd.evaluation as E
    for seed in (5, 6):
        ref = MG.run_reference(seed, n_events=3, imgs_per_event=6)
        events, pred = WF.synth_events(seed, n_events=3, imgs_per_event=6)
        assert np.allclose(E.wider_evaluation(copy.deepcopy(pred), events), ref, rtol=0, atol=1e-12)


def test_perfect_predictions_give_ap_one_and_text_roundtrip(tmp_path):
    import yunet_amd.evaluation as E
    events, _ = WF.synth_events(3)
    pred = {}
    for ev in events:
        pred[ev['name']] = {}
        for im in ev['images']:
            b = im['boxes']
            sc = np.linspace(0.9, 0.5, len(b)) if len(b) else np.zeros(0)
            pred[ev['name']][im['name']] = np.concatenate([b, sc[:, None]], 1).reshape(-1, 5)
    aps = E.wider_evaluation(copy.deepcopy(pred), events)
    assert aps == pytest.approx([1.0, 1.0, 1.0], abs=1e-9)
    # prediction text files (tools/test_widerface.py --save-preds format) round-trip
    d = str(tmp_path / 'preds')
    for ev, imgs in pred.items():
        for name, arr in imgs.items():
            xyxy = arr.copy()
            xyxy[:, 2] += xyxy[:, 0]
            xyxy[:, 3] += xyxy[:, 1]
            E.write_predictions(d, ev, name, xyxy)
    back = E.read_predictions(d)
    assert E.wider_evaluation(back, events) == pytest.approx([1.0, 1.0, 1.0], abs=1e-9)
    # no predictions at all: AP 0, no crash
    empty = {ev['name']: {im['name']: np.zeros((0, 5)) for im in ev['images']} for ev in events}
    one = copy.deepcopy(empty)
    first = events[1]['images'][1]
    one[events[1]['name']][first['name']] = np.array([[1., 1., 5., 5., 0.5], [300., 300., 9., 9., 0.4]])
    with np.errstate(all='ignore'):
        assert all(a == 0.0 or a != a for a in E.wider_evaluation(one, events))


def test_image_eval_matches_sequential_definition():
    """image_eval's array form == the per-prediction loop of the protocol, on random cases."""
    import yunet_amd.evaluation as E
    rng = np.random.default_rng(0)
    for _ in range(50):
        g, n = int(rng.integers(1, 8)), int(rng.integers(1, 30))
        gt = np.concatenate([rng.uniform(0, 100, (g, 2)), rng.uniform(5, 60, (g, 2))], 1).round()
        pick = rng.integers(0, g, n)
        pr = gt[pick] + rng.normal(0, 4, (n, 4))
        pred = np.concatenate([pr, np.sort(rng.uniform(0, 1, (n, 1)), 0)[::-1]], 1)
        flag = (rng.uniform(size=g) < 0.6).astype(np.int64)
        rec, prop = E.image_eval(pred, gt, flag, 0.5)
        recall_list, want_rec, want_prop = np.zeros(g), np.zeros(n), np.ones(n)
        iou = E.pairwise_iou_xywh(pred[:, :4], gt)
        for h in range(n):
            m = int(iou[h].argmax())
            if iou[h, m] >= 0.5:
                if flag[m] == 0:
                    recall_list[m] = -1
                    want_prop[h] = -1
                elif recall_list[m] == 0:
                    recall_list[m] = 1
            want_rec[h] = (recall_list == 1).sum()
        assert np.array_equal(rec, want_rec) and np.array_equal(prop, want_prop)


REAL_GT = '/root/reference/data/widerface/labelv2/val/gt'


@pytest.mark.skipif(not os.path.isdir(REAL_GT), reason='reference tree not present')
def test_real_protocol_ground_truth_matches_reference_aps():
    """The real wider_*_val.mat files (3226 images) + deterministic synthetic predictions: APs equal the
    ones the unmodified reference function produced for the same inputs (oracle/make_golden_wider.py --real)."""
    import make_golden_wider as MG
    import yunet_amd.evaluation as E
    ev, pred = MG.real_gt_preds(REAL_GT)
    aps = E.wider_evaluation(pred, ev)
    assert np.allclose(aps, Hh.load_golden('wider_eval_real.npz')['aps'], rtol=0, atol=1e-12)
