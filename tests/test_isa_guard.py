"""Static guard on the device code (no GPU needed: hipcc cross-compiles gfx950 here).

Round 4, DESIGN.md section 10 finding 7: a loop of the form `for (i = tid; i < n; i += NT) lds[f(i)] = global[i]` (or
`v += global[...]`) is compiled with `s_waitcnt vmcnt(0)` in front of every store / add -- one memory round trip per
trip.  The 64 x 64 pointwise weights in the prologue of dp_fwd64s were 16 such trips per workgroup (0.13 ms per step),
the weight-gradient reduction 48 per thread.  tools/dbg/serial_loads.py reads the `-S` output and reports every loop of
at most 60 instructions with one or two vector-memory loads and a full wait; this test fails when a NEW one of ten or
more instructions appears (shorter ones are the zero-trip tails of bn_sum past eight replicas).
"""
import importlib.util
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'libfacedetection.train_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'

# loops that are known and harmless: at most one or two trips (G <= Gmax ground-truth boxes per image over 256 threads,
# 27 x 16 stem weights over 256 threads in the bf16-storage stem)
KNOWN = {
    ('conv_fwd.hip', 'stem_fwd_kernel'): 1,
    ('loss_step.hip', 'assign_compact_kernel'): 1,
    ('loss_step.hip', 'assign_resolve_kernel'): 2,
    # round 5 (assign_v2): the GT boxes of an image beyond the first 512 / 1024 (zero trips at Gmax <= 512), the chunk
    # counts of an image (one trip: ceil(P / 256) <= 1024), the pair-list offset (one trip for up to 512 images), the
    # prefix scan over blocks of 64 chunk counts (one trip up to P = 16384)
    ('loss_step.hip', 'assign_compact2_kernel'): 2,
    ('loss_step.hip', 'assign_resolve2_kernel'): 2,
    ('loss_step.hip', 'assign_topk2_kernel'): 2,
}
FILES = ['api.hip', 'conv_fwd.hip', 'conv_fwd16.hip', 'conv_fwd64.hip', 'conv_bwd.hip', 'conv_bwd16.hip', 'conv_stem.hip',
         'loss_step.hip']


def _scanner():
    spec = importlib.util.spec_from_file_location('serial_loads', os.path.join(ROOT, 'tools', 'dbg', 'serial_loads.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _assembly(tmp, name):
    out = os.path.join(tmp, name.replace('.hip', '.s'))
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-unused-value', '-w']
    if name == 'loss_step.hip':
        flags.append('-ffp-contract=off')          # as the Makefile builds it
    subprocess.run([HIPCC] + flags + ['-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, name)], check=True,
                   capture_output=True, timeout=600)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not found')
def test_no_new_loop_waits_out_every_load(tmp_path):
    scan = _scanner().scan
    with ThreadPoolExecutor(4) as ex:
        paths = list(ex.map(lambda n: _assembly(str(tmp_path), n), FILES))
    seen, bad = {}, []
    for name, path in zip(FILES, paths):
        for label, n_ins, n_loads, func in scan(path):
            if n_ins < 10:
                continue
            key = next((k for k in KNOWN if k[0] == name and k[1] in func), None)
            if key is None:
                bad.append((name, label, n_ins, n_loads, func))
            else:
                seen[key] = seen.get(key, 0) + 1
    assert not bad, ('loops that issue one load per memory round trip (load everything first, then consume: '
                     f'common.h staged_table / column_slice_sum): {bad}')
    for key, n in seen.items():
        assert n <= KNOWN[key], (key, n)


def test_scanner_flags_the_pattern(tmp_path):
    """The scanner itself: a serialised loop is reported, the batched form is not."""
    scan = _scanner().scan
    src = tmp_path / 'k.s'
    src.write_text('\n'.join([
        '_Z3badPf:',
        '.LBB0_1:',
        '\tglobal_load_dword v1, v[2:3], off',
        '\tv_add_u32_e32 v2, 4, v2',
        '\ts_waitcnt vmcnt(0)',
        '\tds_write_b32 v4, v1',
        '\ts_cbranch_execnz .LBB0_1',
        '\ts_endpgm',
        '_Z4goodPf:',
        '.LBB1_1:',
        '\tglobal_load_dword v1, v[2:3], off',
        '\tglobal_load_dword v5, v[2:3], off offset:1024',
        '\tglobal_load_dword v6, v[2:3], off offset:2048',
        '\ts_waitcnt vmcnt(2)',
        '\tds_write_b32 v4, v1',
        '\ts_waitcnt vmcnt(1)',
        '\tds_write_b32 v4, v5',
        '\ts_waitcnt vmcnt(0)',
        '\tds_write_b32 v4, v6',
        '\ts_cbranch_execnz .LBB1_1',
        '\ts_endpgm', '']))
    found = scan(str(src))
    assert len(found) == 1 and found[0][0] == '.LBB0_1' and 'bad' in found[0][3]
