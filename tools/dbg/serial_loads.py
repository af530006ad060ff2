#!/usr/bin/env python
"""Find loops in device assembly that wait out EVERY vector-memory load before issuing the next one.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s csrc/conv_fwd64.hip
    python tools/dbg/serial_loads.py /tmp/k.s ...

A loop (label .. backward branch to it) of at most 60 instructions with one or two global / buffer / flat loads and an
`s_waitcnt vmcnt(0)` is reported: each trip costs a full memory round trip.  Round 4 found the weight-table copies of
the kernel prologues (16 trips in dp_fwd64s: -0.13 ms per step once batched, common.h: staged_table), the weight-gradient
reduction and loss_finalize this way.
"""
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        return name


def scan(path):
    lines = open(path).read().split('\n')
    labels, func = {}, None
    for idx, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            func = m.group(1)
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = (idx, func)
    out = []
    for idx, l in enumerate(lines):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if not m or m.group(1) not in labels or labels[m.group(1)][0] >= idx:
            continue
        start, func = labels[m.group(1)]
        body = [x.strip() for x in lines[start:idx] if x.startswith('\t') and not x.strip().startswith(';')]
        if len(body) > 60:
            continue
        loads = [x for x in body if re.match(r'(global_load|buffer_load|flat_load)', x)]
        waits = [x for x in body if x.startswith('s_waitcnt') and 'vmcnt(0)' in x]
        if 1 <= len(loads) <= 2 and waits:
            out.append((m.group(1), len(body), len(loads), demangle(func)[:120]))
    return out


if __name__ == '__main__':
    for p in sys.argv[1:]:
        for lab, n, nl, fn in scan(p):
            print(f'{p.split("/")[-1]}  {lab}  {n} instructions, {nl} load(s)  |  {fn}')
