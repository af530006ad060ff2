#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Writes tests/golden/fullstep_<kind>_<size>_<batch>_<weights>.npz: one training step of
the CPU oracle (oracle/yunet_oracle.py -- the restatement pinned against the unmodified reference by
tests/test_oracle_vs_reference.py) at BASELINE.json's FULL batch sizes, evaluated ONCE on the build box, so that
tests/test_fullsize_gpu.py::test_full_step_vs_oracle compares the HIP step with a fixture instead of running four
full-batch oracle steps on the GPU box's host while the GPU lease idles (380 - 860 s of a 90-minute budget, VERDICT r3
weak #7 / next #4).

    python oracle/make_golden_fullstep.py [n:320:256:11:init ...]      (no arguments: the six test cases)

Per case (inputs are regenerated in the test from the same seeds: synthetic.make_batch, oracle.init_state / the
trained fixture):
    losses            [5] fp64: cls, bbox, obj, kps (oracle fp32 step) and their sum
    pos               [K,3] int32: (image, prior, 1-based GT) of every positive of the oracle's assignment
    num_priors        P
    flat_scale        max |flat| of the oracle's forward;  flat_sample = flat[:, ::STRIDE, :] (fp32)
    dflat_scale       max |d loss / d flat|;               dflat_sample likewise
    keys / offsets    parameter names and their offsets into the flat gradient vectors below
    grad64            d loss / d params of an fp64 evaluation of the conv stack fed the oracle's d loss / d flat
                      (the yardstick), stored fp32 (rounding 6e-8 of a bar of 1e-3)
    err_ref           per parameter: max |oracle fp32 gradient - grad64|  (the oracle's own fp32 noise)
    amax64            per parameter: max |grad64|
    bn_keys / bn_vals running_mean / running_var after the step (fp32), num_batches_tracked
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import crafted as C  # noqa: E402
import yunet_amd.synthetic as S  # noqa: E402
import yunet_oracle as O  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
TRAINED = os.path.join(GOLDEN, 'yunet_n_synth_trained.pth')


def trained_path(kind):
    """The trained-checkpoint-like fixture of an architecture (tools/make_trained_fixture.py); YuNet is fully
    convolutional, so the 320 x 320 fixture also drives the 640 x 640 configuration (as in bench.py: run_other_config)."""
    return os.path.join(GOLDEN, f'yunet_{kind}_synth_trained.pth')


# the three BASELINE batches from random initialisation + the three configurations bench.py TIMES (trained fixtures on
# structured faces: SimOTA with dynamic_k 7-9 and real conflicts) -- VERDICT r5 next 3
CASES = [('n', 320, 256, 11, 'init'), ('n', 640, 64, 12, 'init'), ('s', 320, 512, 13, 'init'), ('n', 320, 256, 14, 'trained'),
         ('n', 640, 64, 15, 'trained'), ('s', 320, 512, 16, 'trained')]
STRIDE = 41            # priors kept in the samples: flat[:, ::41, :]


def case_inputs(kind, h, n, seed, weights):
    arch = O.yunet_arch(kind)
    if weights == 'trained':
        sd = {k: v.float() if v.is_floating_point() else v
              for k, v in torch.load(trained_path(kind), map_location='cpu', weights_only=False)['state_dict'].items()}
    else:
        sd = O.init_state(arch, seed=seed)
    b = S.make_batch(n, h, h, seed, structured=weights == 'trained')
    return arch, sd, b


def fixture_path(kind, h, n, weights):
    return os.path.join(GOLDEN, f'fullstep_{kind}_{h}_{n}_{weights}.npz')


def make(kind, h, n, seed, weights):
    t0 = time.time()
    arch, sd, b = case_inputs(kind, h, n, seed, weights)
    sizes = C.featmap_sizes(h, h)
    keys = O.param_keys(sd)
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaf)
    maps = O.conv_stack_forward(b['img'], work, arch, True)
    flat_o = O.flatten_preds(*maps)
    flat_o.retain_grad()
    lo_t, aux_o = O.loss_step(flat_o, b['gt_bboxes'], b['gt_labels'], b['gt_keypointss'], sizes, arch)
    sum(lo_t.values()).backward()
    lo = [float(lo_t[k]) for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps')]
    lo.append(float(sum(lo_t.values())))
    grads_o = [leaf[k].grad.double() for k in keys]
    dflat_o = flat_o.grad.detach().clone()
    flat_d = flat_o.detach().clone()
    gi = aux_o['gt_inds'].int()
    bn = {k: v.detach().clone() for k, v in work.items()
          if k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked')}
    del maps, flat_o, lo_t, aux_o, leaf, work
    print(f'  fp32 oracle step {time.time() - t0:.0f} s', flush=True)

    # fp64 yardstick: the same conv stack in double precision, fed the oracle's d loss / d flat
    t1 = time.time()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaf64 = {k: sd64[k].clone().requires_grad_(True) for k in keys}
    work64 = dict(sd64)
    work64.update(leaf64)
    flat64 = O.flatten_preds(*O.conv_stack_forward(b['img'].double(), work64, arch, True))
    g64 = torch.autograd.grad((flat64 * dflat_o.double()).sum(), [leaf64[k] for k in keys])
    del flat64, work64, leaf64, sd64
    print(f'  fp64 yardstick {time.time() - t1:.0f} s', flush=True)

    offsets = np.cumsum([0] + [int(g.numel()) for g in g64]).astype(np.int64)
    pos = torch.nonzero(gi > 0)
    pos = torch.cat([pos, gi[gi > 0].long()[:, None]], 1).int().numpy()
    np.savez_compressed(
        fixture_path(kind, h, n, weights),
        losses=np.array(lo, np.float64), pos=pos, num_priors=np.int64(gi.shape[1]),
        flat_scale=np.float64(float(flat_d.abs().max())), flat_sample=flat_d[:, ::STRIDE, :].numpy().copy(),
        dflat_scale=np.float64(float(dflat_o.abs().max())), dflat_sample=dflat_o[:, ::STRIDE, :].numpy().copy(),
        stride=np.int64(STRIDE), keys=np.array(keys), offsets=offsets,
        grad64=torch.cat([g.reshape(-1) for g in g64]).float().numpy(),
        err_ref=np.array([float((a - o).abs().max()) for a, o in zip(grads_o, g64)], np.float64),
        amax64=np.array([float(o.abs().max()) for o in g64], np.float64),
        bn_keys=np.array(sorted(bn)), bn_offsets=np.cumsum([0] + [int(bn[k].numel()) for k in sorted(bn)]).astype(np.int64),
        bn_vals=torch.cat([bn[k].reshape(-1).float() for k in sorted(bn)]).numpy(),
        seed=np.int64(seed))
    print(f'{fixture_path(kind, h, n, weights)}: {os.path.getsize(fixture_path(kind, h, n, weights)) / 1e6:.2f} MB, '
          f'{len(pos)} positives, losses {lo}, {time.time() - t0:.0f} s', flush=True)


if __name__ == '__main__':
    cases = CASES
    if len(sys.argv) > 1:
        cases = []
        for a in sys.argv[1:]:
            k, h, n, s, w = a.split(':')
            cases.append((k, int(h), int(n), int(s), w))
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', os.cpu_count() or 8)))
    for c in cases:
        print('case', c, flush=True)
        make(*c)
