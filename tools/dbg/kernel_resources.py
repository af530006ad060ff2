#!/usr/bin/env python
"""Register / LDS budget of every kernel of the library, from the metadata hipcc writes into its `-S` output.

    python tools/dbg/kernel_resources.py [--md] [file.hip ...]        (default: every .hip of csrc/)

Columns: threads per workgroup (launch bound), VGPRs (+ AGPRs), resulting waves per SIMD (512 registers per SIMD lane on
gfx950: floor(512 / (vgpr + agpr rounded up to 8)), at most 8), spilled VGPRs, scratch bytes per lane, static LDS bytes
(the dynamic part is set at launch: see the launch_* functions).  DESIGN.md section 4 quotes this table.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'libfacedetection.train_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def assembly(path, tmp):
    out = os.path.join(tmp, os.path.basename(path).replace('.hip', '.s'))
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-w']
    if os.path.basename(path) in ('loss_step.hip', 'augment.hip', 'detect.hip'):
        flags.append('-ffp-contract=off')
    subprocess.run([HIPCC] + flags + ['-S', '--cuda-device-only', '-o', out, path], check=True, capture_output=True)
    return out


def kernels(asm):
    text = open(asm).read()
    meta = text[text.find('amdhsa.kernels:'):]
    rows = []
    for blk in re.split(r'\n  - \.agpr_count:', meta)[1:]:
        blk = '.agpr_count:' + blk

        def field(name, cast=int):
            m = re.search(r'\.' + name + r':\s+(\S+)', blk)
            return cast(m.group(1)) if m else None
        name = field('name', str)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(anonymous namespace\)::', '', dem)
        dem = re.sub(r'^void ', '', dem)
        dem = re.sub(r'\(.*$', '', dem)
        v, a = field('vgpr_count'), field('agpr_count') or 0
        alloc = (v + a + 7) // 8 * 8
        rows.append(dict(kernel=dem, threads=field('max_flat_workgroup_size'), vgpr=v, agpr=a,
                         waves_per_simd=min(8, 512 // max(alloc, 1)), spill=field('vgpr_spill_count'),
                         scratch=field('private_segment_fixed_size'), lds_static=field('group_segment_fixed_size'),
                         sgpr=field('sgpr_count')))
    return rows


def main(argv):
    md = '--md' in argv
    files = [a for a in argv if a.endswith('.hip')] or sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(4) as ex:
            asms = list(ex.map(lambda p: assembly(p, tmp), files))
        for f, asm in zip(files, asms):
            rows = kernels(asm)
            if md:
                print(f'\n`{os.path.basename(f)}`\n\n| kernel | threads | VGPR (+AGPR) | waves / SIMD | spilled | scratch B | static LDS B |\n|---|---|---|---|---|---|---|')
            for r in sorted(rows, key=lambda r: r['kernel']):
                regs = f"{r['vgpr']}" + (f" + {r['agpr']}" if r['agpr'] else '')
                if md:
                    print(f"| `{r['kernel']}` | {r['threads']} | {regs} | {r['waves_per_simd']} | {r['spill']} | {r['scratch']} | {r['lds_static']} |")
                else:
                    print(f"{os.path.basename(f):16s} {r['kernel'][:70]:70s} thr {r['threads']:4d}  vgpr {regs:9s} waves/SIMD {r['waves_per_simd']}  "
                          f"spill {r['spill']}  scratch {r['scratch']}  lds {r['lds_static']}")


if __name__ == '__main__':
    main(sys.argv[1:])
