#!/usr/bin/env python
"""Per-kernel micro-benchmark at the headline shapes (YuNet_n 320x320 bs 256): times single
launches with events on the launch stream and prints achieved algorithmic GB/s.

    python tools/kbench.py [--reps 10] [--only dp_fwd64]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import yunet_amd.kernels as K  # noqa: E402

DEV = 'cuda'


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def unit(ci, co):
    g = torch.Generator().manual_seed(ci * 7 + co)
    return (torch.randn(co, ci, generator=g).to(DEV) * 0.1, torch.randn(co, generator=g).to(DEV) * .1,
            torch.randn(co, 9, generator=g).to(DEV) * 0.3, torch.randn(co, generator=g).to(DEV) * .1)


def stats_like(x):
    c = x.shape[-1]
    v = x.reshape(-1, c)[:65536].double()
    n = x.numel() // c
    return torch.cat([v.mean(0) * n, (v * v).mean(0) * n]).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--prof', action='store_true')
    ap.add_argument('--calib', action='store_true', help='also run a 256 MiB device copy (PMC unit calibration)')
    ap.add_argument('--ablate', action='store_true')
    ap.add_argument('--delay', action='store_true', help='forward-kernel phase ablation')
    a = ap.parse_args()
    N = a.n
    if a.calib:
        src = torch.rand(64 * 1024 * 1024, device=DEV)      # 256 MiB read + 256 MiB written
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
    cases = [('dp64', 64, 64, 80, 80), ('dp16', 16, 16, 160, 160), ('dp16_64', 16, 64, 80, 80),
             ('dp64_40', 64, 64, 40, 40), ('head', 64, 16, 40, 40),
             ('dp64_20', 64, 64, 20, 20), ('dp64_10', 64, 64, 10, 10), ('head_10', 64, 16, 10, 10)]
    for name, ci, co, h, w in cases:
        if a.only and a.only not in name:
            continue
        x = torch.randn(N, h, w, ci, device=DEV)
        wp, bp, wd, bd = unit(ci, co)
        gam, bet = torch.ones(ci, device=DEV), torch.zeros(ci, device=DEV)
        in_bn = K.BN(stats_like(x), gam, bet, N * h * w, bstats=torch.zeros(2 * ci, dtype=torch.float64, device=DEV))
        ostats = torch.zeros(2 * co, dtype=torch.float64, device=DEV)
        has_bn = co != 16 or ci != 64
        out_bn = K.BN(ostats, torch.ones(co, device=DEV), torch.zeros(co, device=DEV), N * h * w,
                      bstats=torch.zeros(2 * co, dtype=torch.float64, device=DEV)) if has_bn else None
        z = torch.empty(N, h, w, co, device=DEV)
        t = timeit(lambda: K.dp_fwd(x, wp, bp, wd, bd, in_bn, out_bn, z=z), a.reps)
        if a.delay:
            import ctypes as C
            import yunet_amd._lib as L
            dd = K._dp_desc(x, wp, bp, wd, bd, z, in_bn, out_bn)
            for dl in (0, 1, 2, 4, 8, 3, 15):
                dd.prof = dl if dl else None
                tt = timeit(lambda: L.check(L.load().yunet_dp_fwd(C.byref(dd), K._stream()), 'fwd'), a.reps)
                print(f'   fwd ablate mask {dl:2d}: {tt:.4f} ms')
        if a.prof:
            import ctypes as C
            import yunet_amd._lib as L
            dd = K._dp_desc(x, wp, bp, wd, bd, z, in_bn, out_bn)
            pr = torch.zeros(4096, 16, dtype=torch.int64, device=DEV)
            dd.prof = pr.data_ptr()
            L.check(L.load().yunet_dp_fwd(C.byref(dd), K._stream()), 'fwd')
            torch.cuda.synchronize()
            v = pr.view(-1, 4).double()
            v = v[v.sum(1) > 0]
            print('   fwd phase cycles/wave (stage, pw, dw, endbar):', [int(x) for x in v.mean(0).tolist()], 'waves', v.shape[0], 'max total', int(v.sum(1).max()))
        px = N * h * w
        by = px * (ci + co) * 4
        print(f'{name:8s} fwd  {t:8.4f} ms  {by / t / 1e6:8.1f} GB/s  {t * 1e-3 * 2.4e9 * 256 / px:7.1f} CUcyc/px')
        dy = torch.randn(N, h, w, co, device=DEV)
        dx = torch.empty_like(x)
        if out_bn is not None:
            out_bn.stats = stats_like(z)
        blocks = K.dp_grid(N, h, w, ci, co)
        part = torch.empty(blocks, K.dp_row_width(ci, co), device=DEV)
        import ctypes as C
        import yunet_amd._lib as L
        d = K._dp_desc(x, wp, bp, wd, bd, z, in_bn, out_bn)
        d.dy, d.dx = dy.data_ptr(), dx.data_ptr()
        d.wgrad_partials, d.wgrad_blocks = part.data_ptr(), blocks
        lib = L.load()
        t = timeit(lambda: L.check(lib.yunet_dp_bwd(C.byref(d), K._stream()), 'bwd'), a.reps)
        if a.ablate:
            for m in (1, 2, 4, 8, 16, 32, 63, 62, 61, 59, 55, 47, 31):
                d.prof = m
                tt = timeit(lambda: L.check(lib.yunet_dp_bwd(C.byref(d), K._stream()), 'bwd'), a.reps)
                print(f'   ablate mask {m:2d}: {tt:.4f} ms (full {t:.4f})')
            d.prof = None
        by = px * (2 * ci + co) * 4
        print(f'{name:8s} bwd  {t:8.4f} ms  {by / t / 1e6:8.1f} GB/s  {t * 1e-3 * 2.4e9 * 256 / px:7.1f} CUcyc/px')
        out = torch.empty(part.shape[1], device=DEV)
        t = timeit(lambda: K.reduce_partials(part, out), a.reps)
        print(f'{name:8s} red  {t:8.4f} ms  rows={blocks}')
    if not a.only or 'stem' in a.only:
        img = torch.rand(N, 3, 320, 320, device=DEV) * 255
        w = torch.randn(16, 3, 3, 3, device=DEV) * 0.05
        b = torch.zeros(16, device=DEV)
        st = torch.zeros(32, dtype=torch.float64, device=DEV)
        t = timeit(lambda: K.stem_fwd(img, w, b, st), a.reps)
        by = N * (3 * 320 * 320 + 16 * 160 * 160) * 4
        print(f'stem     fwd  {t:8.4f} ms  {by / t / 1e6:8.1f} GB/s')
    if 'aug' in a.only:
        # device input pipeline: 256 WIDER-sized uint8 sources (768x1024) -> S x S fp32 batches.
        # algorithmic bytes per image: the crop window read once (cw*cw*3 B, clipped to the
        # source) + the planar fp32 output (3*S*S*4 B)
        import numpy as np
        from yunet_amd.pipelines import DevicePipeline, SourceBatch
        rng = np.random.default_rng(0)
        n, h, w = N, 768, 1024
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        gts = []
        for _ in range(n):
            g = int(rng.integers(1, 24))
            x1, y1 = rng.uniform(0, w - 80, g), rng.uniform(0, h - 80, g)
            side = rng.uniform(8, 80, g)
            gts.append((np.stack([x1, y1, x1 + side, y1 + side], 1).astype(np.float32),
                        np.ones((g, 5, 3), np.float32)))
        sb = SourceBatch.from_lists([img] * n, [g[0] for g in gts], [g[1] for g in gts], DEV)
        for S in (320, 640):
            pipe = DevicePipeline([
                dict(type='LoadImageFromFile', to_float32=True),
                dict(type='LoadAnnotations', with_bbox=True, with_keypoints=True),
                dict(type='RandomSquareCrop', crop_choice=[0.5, 0.7, 0.9, 1.1, 1.3, 1.5]),
                dict(type='Resize', img_scale=(S, S), keep_ratio=False),
                dict(type='RandomFlip', flip_ratio=0.5),
                dict(type='Normalize', mean=[0., 0., 0.], std=[1., 1., 1.], to_rgb=False),
                dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img'])], seed=1)
            it = [0]

            def run():
                it[0] += 1
                pipe(sb, it[0])
            t = timeit(run, a.reps)
            pr = pipe.params.cpu().numpy().astype(np.int64)
            x0, y0 = np.maximum(pr[:, 0], 0), np.maximum(pr[:, 1], 0)
            x1, y1 = np.minimum(pr[:, 0] + pr[:, 2], w), np.minimum(pr[:, 1] + pr[:, 2], h)
            src_b = int(((x1 - x0) * (y1 - y0) * 3).sum())
            out_b = n * 3 * S * S * 4
            print(f'aug S={S}  {t:8.4f} ms/batch  {n / t * 1e3:10.0f} img/s  '
                  f'{(src_b + out_b) / t / 1e6:8.1f} GB/s (src {src_b / 1e6:.0f} MB + out {out_b / 1e6:.0f} MB)')
    if 'det' in a.only:
        # test-time path: eval-mode forward of YuNet_n on 256 x 320x320 + get_bboxes (one workgroup
        # per image: scores, bitonic sort of the candidates, greedy NMS)
        import yunet_amd
        cfg = yunet_amd.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     'configs', 'yunet_n.py'))
        torch.manual_seed(0)
        model = yunet_amd.build_detector(cfg.model).to(DEV).eval()
        img = torch.rand(N, 3, 320, 320, device=DEV) * 255
        eng = model._ensure_engine(torch.device(DEV))
        t = timeit(lambda: eng.forward_eval(img), a.reps)
        print(f'eval fwd {t:8.4f} ms/batch  {N / t * 1e3:10.0f} img/s')
        P = 2100
        sizes = [(40, 40), (20, 20), (10, 10)]
        g = torch.Generator(device='cpu').manual_seed(0)
        for label, shift in (('sparse', -7.0), ('medium', -4.5), ('dense', 0.0)):
            flat = torch.randn(N, P, 16, generator=g)
            flat[..., 0] = flat[..., 0] * 1.5 + shift
            flat[..., 3:5] = flat[..., 3:5] * 0.5 + 0.8
            flat = flat.to(DEV)
            t = timeit(lambda: K.detect(flat, sizes, [8, 16, 32]), a.reps)
            _, _, cnt = K.detect(flat, sizes, [8, 16, 32])
            cand = int(((flat[..., 0].sigmoid() * flat[..., 5].sigmoid()) >= 0.02).sum()) / N
            print(f'detect   {t:8.4f} ms/batch  {N / t * 1e3:10.0f} img/s  {label}: {cand:.0f} candidates '
                  f'-> {float(cnt.float().mean()):.0f} kept per image')


if __name__ == '__main__':
    main()
