#!/usr/bin/env python
"""End-to-end images/s of tools/train.py: the reference's command line, 200 iterations of
configs/yunet_n.py (YuNet_n 320 x 320, 256 images per GPU) from three data sources, next to the step bench.py times:

  ready     SyntheticWiderFace      finished fp32 batches resident in HBM (what bench.py feeds: the step alone + the runner's
                                    hooks, LR schedule, logging)
  resident  SyntheticSourceImages   decoded uint8 sources resident in HBM, the reference's train pipeline (RandomSquareCrop ->
                                    Resize -> RandomFlip -> Normalize -> collate) on the GPU every iteration
  host_fed  SyntheticSourceImages   the same sources in PINNED HOST memory, each batch uploaded on a copy stream into one of
            (host_fed=True)         two device buffers while the previous step runs

    python tools/train_e2e.py [--iters 200] [--out profiles/r06_train_e2e.json]

Prints / writes one JSON object: per mode images/s over the last three logging intervals (150 iterations), the runner's
time per iteration, and the source's own events (upload ms / GB/s, pipeline ms).  Weights: the trained fixture
(tests/golden/yunet_n_synth_trained.pth) so that SimOTA works as in bench.py."""
import argparse
import importlib.util
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_train_tool():
    spec = importlib.util.spec_from_file_location('yunet_train_tool', os.path.join(ROOT, 'tools', 'train.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'yunet_n.py'))
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    import torch
    T = load_train_tool()
    fixture = os.path.join(ROOT, 'tests', 'golden', 'yunet_n_synth_trained.pth')
    modes = [('ready', ['data.train.type=SyntheticWiderFace', 'data.train.resident=2']),
             ('resident', ['data.train.type=SyntheticSourceImages', 'data.train.timing=True']),
             ('host_fed', ['data.train.type=SyntheticSourceImages', 'data.train.timing=True', 'data.train.host_fed=True'])]
    res = {'what': __doc__.split('\n')[0], 'config': os.path.basename(a.config), 'iters': a.iters, 'modes': {}}
    for name, opts in modes:
        with tempfile.TemporaryDirectory() as wd:
            argv = [a.config, '--work-dir', wd, '--max-iters', str(a.iters), '--no-validate', '--seed', '0',
                    '--cfg-options', 'log_config.interval=50', f'load_from={fixture}'] + opts
            hist = T.main(argv)
            torch.cuda.synchronize()
        rows = [r for r in hist if 'time' in r]
        bs = None
        import yunet_amd
        bs = yunet_amd.Config.fromfile(a.config).data.samples_per_gpu
        steady = rows[1:] if len(rows) > 1 else rows          # the first interval holds start-up (plan build, first launches)
        t = sum(r['time'] for r in steady) / len(steady)
        m = {'ms_per_iter': round(1000 * t, 3), 'images_per_sec': round(bs / t, 1), 'batch': bs,
             'intervals_ms_per_iter': [round(1000 * r['time'], 3) for r in rows],
             'final_loss': round(float(rows[-1]['loss']), 4)}
        src = getattr(T.main, 'last_source', None)
        if src is not None and hasattr(src, 'report'):
            m['source_events'] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in src.report().items()}
        res['modes'][name] = m
        print(name, json.dumps(m), flush=True)
        del src
        T.main.last_source = None
        torch.cuda.empty_cache()
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(json.dumps(res, indent=1) + '\n')


if __name__ == '__main__':
    main()
