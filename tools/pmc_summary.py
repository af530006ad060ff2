#!/usr/bin/env python
"""Turn rocprofv3 --pmc counter_collection CSVs into the per-kernel HBM traffic table that
bench.py reports as roofline.traffic.

    python tools/pmc_summary.py FETCH.csv WRITE.csv OUT.json [--calib-bytes 268435456]

Units / corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads show up at exactly
one half -- it is doubled here.  A device-to-device copy of known size in the same pass (any
`__amd_rocclr_copyBuffer` launch, tools/kbench.py --calib) is used to print the calibration
factors next to the table; WRITE_SIZE needed no correction in this environment.
"""
import csv
import re
import json
import sys
from collections import defaultdict


def per_kernel(path):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name') or row.get('Kernel Name')
            val = float(row.get('Counter_Value') or row.get('Counter Value'))
            m = re.search(r'(\w+_kernel)(<[^>(]*>)?', name)
            if m and 'at::native' not in name:
                key = m.group(1) + (m.group(2) or '').replace(' ', '')
            elif name.startswith('__amd_rocclr'):
                key = name
            else:
                key = 'torch:' + re.sub(r'[^\w]+', '_', name)[-40:]
            acc[key][0] += 1
            acc[key][1] += val
            acc[key][2] = max(acc[key][2], val)
    return {k: (n, s / n, mx) for k, (n, s, mx) in acc.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    calib = 256 * 1024 * 1024
    if '--calib-bytes' in sys.argv:
        calib = int(sys.argv[sys.argv.index('--calib-bytes') + 1])
    fe, wr = per_kernel(fetch), per_kernel(write)
    table = {}
    for k in sorted(set(fe) | set(wr)):
        nf, f_kib, f_max = fe.get(k, (0, 0.0, 0.0))
        nw, w_kib, w_max = wr.get(k, (0, 0.0, 0.0))
        table[k] = {
            'launches_fetch_pass': nf, 'launches_write_pass': nw,
            'FETCH_SIZE_KiB_raw': round(f_kib, 1), 'WRITE_SIZE_KiB_raw': round(w_kib, 1),
            # bytes per launch: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE
            'read_bytes': int(2 * f_kib * 1024), 'write_bytes': int(w_kib * 1024),
            'traffic_bytes': int(2 * f_kib * 1024 + w_kib * 1024),
            'max_read_bytes': int(2 * f_max * 1024), 'max_write_bytes': int(w_max * 1024),
        }
    cal = {k: v for k, v in table.items() if 'copyBuffer' in k}
    for k, v in cal.items():
        v['calibration'] = {'expected_read_bytes': calib, 'expected_write_bytes': calib,
                            'read_ratio': round(v['max_read_bytes'] / calib, 3),     # largest copy of the pass
                            'write_ratio': round(v['max_write_bytes'] / calib, 3)}
    json.dump({'note': 'HBM bytes per launch, mean over the launches of each kernel in the pass; '
                       'read = 2 x FETCH_SIZE (gfx950), write = WRITE_SIZE, KiB -> bytes',
               'kernels': table}, open(out, 'w'), indent=1)
    for k, v in sorted(table.items(), key=lambda kv: -kv[1]['traffic_bytes'])[:12]:
        print(f"{k[:48]:48s} read {v['read_bytes'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB")


if __name__ == '__main__':
    main()
