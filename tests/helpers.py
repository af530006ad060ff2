"""Shared test helpers (CPU side): golden loading, tie analysis."""
import os

import numpy as np
import torch

import crafted as C
import yunet_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_loss_inputs(name):
    g = load_golden(name)
    flat = torch.from_numpy(g['flat'])
    gb, gk, cnt = torch.from_numpy(g['gt_boxes']), torch.from_numpy(g['gt_kps']), \
        torch.from_numpy(g['gt_count'])
    h, w = int(g['height']), int(g['width'])
    return g, flat, gb, gk, cnt, h, w


def image_near_tie(flat_n, gt_boxes_n, sizes, strides=(8, 16, 32), rel=4e-7):
    """True if image n's SimOTA decision sits within fp32 transcendental rounding of a tie
    (k-th vs (k+1)-th cost, conflict argmin, or dynamic-k truncation)."""
    pri = O.grid_priors(sizes, list(strides))
    off = torch.cat([pri[:, :2] + pri[:, 2:] * 0.5, pri[:, 2:]], -1)
    dec = O.bbox_decode(pri, flat_n[:, 1:5])
    sc = flat_n[:, 0].sigmoid() * flat_n[:, 5].sigmoid()
    labels = torch.zeros(gt_boxes_n.shape[0], dtype=torch.int64)
    _, _, _, d = O.simota_assign(sc, off, dec, gt_boxes_n.float(), labels, return_debug=True)
    cost = torch.sort(d['cost'], dim=0).values
    for g in range(cost.shape[1]):
        k = int(d['dynamic_ks'][g])
        if k < cost.shape[0]:
            if float((cost[k, g] - cost[k - 1, g]) / cost[k - 1, g].abs()) <= rel:
                return True
    K = min(10, d['ious'].shape[0])
    s = torch.sort(d['ious'], dim=0, descending=True).values[:K].sum(0)
    frac = (s - s.round()).abs()
    frac[s == 0] = 1.0        # no overlapping candidate at all: dynamic_k = max(int(0), 1) is not a rounding question
    if float(frac.min()) < 1e-5:
        return True
    # conflict resolution: only priors matched to more than one GT take the argmin over their costs
    cost_m = d['cost']
    if cost_m.shape[1] > 1:
        order = torch.argsort(cost_m, dim=0, stable=True)
        matching = torch.zeros_like(cost_m, dtype=torch.bool)
        for g in range(cost_m.shape[1]):
            matching[order[:int(d['dynamic_ks'][g]), g], g] = True
        multi = matching.sum(1) > 1
        if bool(multi.any()):
            two = torch.sort(torch.where(matching[multi], cost_m[multi], torch.full_like(cost_m[multi], float('inf'))),
                             dim=1).values
            if float(((two[:, 1] - two[:, 0]) / two[:, 0].abs()).min()) <= rel:
                return True
    return False


def fp64_conv_grads(batch, sd, arch, dflat):
    """d(sum(flat * dflat))/d(params) of the conv stack in fp64: the yardstick against which both
    fp32 implementations (the oracle port and the HIP path) are measured."""
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    keys = O.param_keys(sd64)
    leaf = {k: sd64[k].clone().requires_grad_(True) for k in keys}
    work = dict(sd64)
    work.update(leaf)
    maps = O.conv_stack_forward(batch['img'].double(), work, arch, True)
    flat = O.flatten_preds(*maps)
    (flat * dflat.double()).sum().backward()
    return {k: leaf[k].grad for k in keys}
