#!/usr/bin/env python
"""Idle time between the kernels of one training step, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 6 --warmup 3 \
        --no-cpu-baseline --no-roofline --no-exact-bwd --no-other-configs
    python tools/trace_gaps.py /tmp/tr/.../t_kernel_trace.csv out.json

A step is delimited by its sgd_kernel launch.  For the last steps of the trace the script reports the step's span
(first kernel start -> sgd end), the union of the kernels' busy intervals over ALL queues (so work on the lanes /
the side stream counts as busy), the idle remainder, and the largest gaps with the kernels either side of them.
"""
import csv
import json
import sys


def main(path, out=None, last=4):
    rows = list(csv.DictReader(open(path)))
    ks = []
    for r in rows:
        name = r.get('Kernel_Name') or r.get('Name')
        if name is None or 'at::native' in name:
            continue
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', '0')))
    ks.sort()
    ends = [i for i, k in enumerate(ks) if 'sgd_kernel' in k[2]]
    steps = []
    for a, b in zip(ends[:-1], ends[1:]):
        seg = ks[a + 1:b + 1]
        t0, t1 = seg[0][0], max(k[1] for k in seg)
        busy, gaps, cur = 0, [], None
        for s, e, n, q in seg:
            if cur is None:
                cur = [s, e, n]
            elif s <= cur[1]:
                if e > cur[1]:
                    cur[1], cur[2] = e, n
            else:
                busy += cur[1] - cur[0]
                gaps.append((s - cur[1], cur[2][:60], n[:60]))
                cur = [s, e, n]
        busy += cur[1] - cur[0]
        steps.append(dict(span_us=(t1 - t0) / 1e3, busy_us=busy / 1e3, idle_us=(t1 - t0 - busy) / 1e3,
                          kernels=len(seg), queues=len({k[3] for k in seg}),
                          period_us=(ks[b][1] - ks[a][1]) / 1e3,
                          sum_kernel_us=sum(k[1] - k[0] for k in seg) / 1e3,
                          top_gaps=[dict(us=g[0] / 1e3, after=g[1], before=g[2])
                                    for g in sorted(gaps, reverse=True)[:8]],
                          gaps_over_1us=sum(1 for g in gaps if g[0] > 1000),
                          median_gap_us=sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0))
    steps = steps[-last:]
    res = dict(trace=path.split('/')[-1], steps=steps)
    text = json.dumps(res, indent=1)
    if out:
        open(out, 'w').write(text + '\n')
    for s in steps:
        print({k: v for k, v in s.items() if k != 'top_gaps'})
    if steps:
        for g in steps[-1]['top_gaps']:
            print('   ', g)


if __name__ == '__main__':
    main(*sys.argv[1:3])
