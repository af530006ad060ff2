"""CPU: bench.py's algorithmic-byte model can never claim more bytes than the HBM counters saw.

For the newest round that committed BOTH a bench line with per-kernel model bytes (profiles/rNN_bench.json: kernels[*].
algorithmic_bytes_per_launch) and the PMC table of the same step (profiles/rNN_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE
passes), every kernel that moves at least 32 MB per launch must satisfy  model bytes <= 1.02 x counter bytes  -- a model
figure above the counters (rounds 1-5 charged the stem's backward an image gradient nobody writes: 1 049 MB against
773 MB measured) makes `families.*.frac` read higher than the kernel can be.  Small launches are exempt: their inputs were
just written by the previous kernel and partly hit in the 4 MB L2s, so the counters legitimately see less than the
unit-boundary bytes."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_BYTES = 32e6


def newest_pair():
    rounds = sorted({int(m.group(1)) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json'))
                     for m in [re.match(r'r(\d+)_pmc_traffic\.json', os.path.basename(f))] if m}, reverse=True)
    for r in rounds:
        b = os.path.join(ROOT, 'profiles', f'r{r:02d}_bench.json')
        p = os.path.join(ROOT, 'profiles', f'r{r:02d}_pmc_traffic.json')
        if not os.path.exists(b):
            continue
        kernels = json.load(open(b)).get('kernels', {})
        if any('algorithmic_bytes_per_launch' in v for v in kernels.values()):
            return r, kernels, json.load(open(p))['kernels']
    return None, None, None


def test_model_bytes_never_exceed_counter_bytes():
    r, kernels, pmc = newest_pair()
    if r is None:
        pytest.skip('no committed round carries per-kernel model bytes next to its PMC table yet')
    checked, bad = 0, []
    for name, v in kernels.items():
        model = v.get('algorithmic_bytes_per_launch', 0)
        key = name.replace(' ', '')
        if model < MIN_BYTES or key not in pmc:
            continue
        seen = pmc[key]['traffic_bytes']
        checked += 1
        if model > 1.02 * seen:
            bad.append((name, model, seen, round(model / seen, 3)))
    assert checked >= 8, f'round {r}: only {checked} kernels could be compared'
    assert not bad, f'round {r}: model bytes above the counters: {bad}'


def test_stem_backward_is_charged_without_an_image_gradient():
    """The concrete case: image read once + dy read, nothing written for the (leaf) image."""
    import importlib.util
    import ctypes as C
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import yunet_amd._lib as L
    op = L.YunetOp()
    op.opcode = L.OP_STEM_BWD
    op.i[0], op.i[1], op.i[2], op.i[11] = 256, 320, 320, L.F32
    img, out = 256 * 3 * 320 * 320 * 4, 256 * 160 * 160 * 16 * 4
    assert bench.op_bytes(op, L) == img + out
    assert bench.op_bytes_reference_graph(op, L) == 2 * img + out          # SURVEY 8d's generic 2*in + out (step_frac's numerator)
    op.opcode = L.OP_STEM_FWD
    assert bench.op_bytes(op, L) == img + out
