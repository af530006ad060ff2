#!/usr/bin/env python
"""Mean value of each counter per kernel from rocprofv3 --pmc counter_collection CSVs.

    python tools/pmc_mean.py OUT.json CSV [CSV ...]

Used for the derived utilisation metrics (MfmaUtil, VALUBusy, LdsUtil, LDSBankConflict, ...)
that tools/profile_round.sh collects in separate passes over tools/kbench.py.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in files:
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row['Kernel_Name']
                m = re.search(r'(\w+_kernel)(<[^>(]*>)?', name)
                if not m or 'at::native' in name:
                    continue
                key = m.group(1) + (m.group(2) or '').replace(' ', '')
                a = acc[key][row['Counter_Name']]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    table = {k: {c: round(s / n, 3) for c, (n, s) in v.items()} | {'launches': max(n for n, _ in v.values())}
             for k, v in acc.items()}
    json.dump(table, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in sorted(table.items()):
        print(f'{k[:44]:44s}', {c: x for c, x in v.items() if c != 'launches'})


if __name__ == '__main__':
    main()
