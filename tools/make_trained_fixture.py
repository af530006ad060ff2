#!/usr/bin/env python
"""Train YuNet_n for a few thousand iterations on STRUCTURED synthetic batches (synthetic.render_faces)
on one MI355X and store the state_dict as tests/golden/yunet_n_synth_trained.pth -- the
"trained-checkpoint-like" weights bench.py loads so that SimOTA runs with dynamic_k > 1 (SURVEY 8d:
a randomly initialised network gives k = 1 for 90 % of the GTs).

    python tools/make_trained_fixture.py [--kind n|s] [--iters 3000] [--batch 64]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import yunet_amd  # noqa: E402
import yunet_amd.synthetic as S  # noqa: E402
from yunet_amd.optim import FusedSGD  # noqa: E402
from yunet_amd.runner import StepLrWarmup  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--kind', default='n', choices=['n', 's'])
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    a.out = a.out or os.path.join(ROOT, 'tests', 'golden', f'yunet_{a.kind}_synth_trained.pth')
    dev = torch.device('cuda', 0)
    cfg = yunet_amd.Config.fromfile(os.path.join(ROOT, 'configs', f'yunet_{a.kind}.py'))
    torch.manual_seed(0)
    model = yunet_amd.build_detector(cfg.model).to(dev).train()
    opt = FusedSGD(model, lr=0.01, momentum=0.9, weight_decay=5e-4)
    sched = StepLrWarmup(0.01, step=[10 ** 9], warmup='linear', warmup_iters=500, warmup_ratio=0.001)
    t0 = time.time()
    for it in range(a.iters):
        opt.param_groups[0]['lr'] = sched.lr_at(0, it)
        b = S.make_batch(a.batch, 320, 320, 50_000 + it, with_img=False)
        gen = torch.Generator(device=dev).manual_seed(50_000 + it)
        img = torch.rand(a.batch, 3, 320, 320, generator=gen, device=dev) * 255.0
        b['img'] = S.render_faces(img, b['gt_bboxes'], b['gt_keypointss'])
        out = model.train_step(S.to_device(b, dev), opt)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        if (it + 1) % 250 == 0:
            lv = {k: round(float(v), 3) for k, v in out['log_vars'].items()}
            plan = model.engine.plan
            pos = plan.gt_inds > 0
            print(f'[{it + 1}] {lv} matched IoU {float(plan.max_overlaps[pos].mean()):.3f} '
                  f'positives/GT {float(pos.sum()) / float(plan.gt_count.sum()):.2f} ({time.time() - t0:.0f} s)', flush=True)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.save(dict(state_dict=sd, meta=dict(iters=a.iters, batch=a.batch, data='synthetic.render_faces')), a.out)
    print('saved', a.out, os.path.getsize(a.out), 'bytes')


if __name__ == '__main__':
    main()
