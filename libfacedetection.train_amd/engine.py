"""Fused YuNet training-step engine: the whole hot path as three replayable op lists.

`forward_train` of the reference (mmdet/models/detectors/yunet.py:21-51) walks
backbone -> TFPN -> head -> loss through ~76 eager conv/BN/ReLU ops and a Python loop
over images; here the same step is a fixed sequence of HIP kernel launches, built once
per input shape as an array of `YunetOp` and replayed by ONE C call per phase
(`yunet_exec`):

    fwd_a : stem, ConvDPUnits, pools, upsample-adds, fused heads, assign, loss_norm
            [optional RCCL all-reduce of the scalar num_pos when world > 1]
    fwd_b : loss (+ d loss / d preds), loss_finalize, BN running-stat updates
    bwd   : head / ConvDPUnit / pool / upsample-add / stem backward, partial reductions,
            BN parameter grads  -> one flat gradient buffer

PyTorch supplies device memory, streams and torch.distributed only.
"""
import collections
import warnings
import ctypes as C
import math
import os

import torch

from . import _lib as L
from . import kernels as K

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# Replicas of every BatchNorm layer's fp64 sum blocks (YunetBN::slots).  A kernel ends with one fp64 atomic
# per channel and workgroup; at the packed 20 x 20 / 10 x 10 levels ~250 workgroups hit the same eight cache
# lines and that tail was 4-5 us of a 14-35 us launch.  Eight replicas remove it (tools/ubench/bwd_ab.cpp,
# DESIGN.md section 7); the readers add the replicas up.
BN_SLOTS = max(1, int(os.environ.get('YUNET_BN_SLOTS', '8')))
MAX_PLANS = max(2, int(os.environ.get('YUNET_MAX_PLANS', '16')))     # plans kept per engine (see get_plan)
LOG_HEAD = 8          # floats in front of the flat gradient: cls, bbox, obj, kps, total, 3 spare


# ============================================================================ parameters
class ParamLayout:
    """Flat fp32 parameter buffer <-> reference state_dict names (OIHW shapes).

    Every kernel unit owns a contiguous block  W_pw | b_pw | W_dw | b_dw [| gamma | beta]
    (stem: W | b | gamma | beta) so a unit's weight-gradient partial row reduces straight
    into the flat gradient buffer.  The four per-level head ConvDPUnits
    (cls 1, bbox 4, obj 1, kps 10 channels; yunet_head.py:149-156) are stored as ONE fused
    64->16 unit whose rows are the concatenation cls|bbox|obj|kps -- each reference
    tensor is still a contiguous slice, so the state_dict contract is unchanged.
    """

    def __init__(self, arch):
        self.arch = arch
        self.entries = {}      # state_dict key -> (offset, shape)
        self.units = {}        # unit name -> dict(off=..., cin, cout, bn=bool)
        self.buffers = {}      # BN running buffers: key -> (index into bn list, kind)
        self.bn_names = []     # BN prefixes in forward order
        off = 0
        st = arch['stage_channels']
        cin0, cmid, _ = st[0]
        assert cin0 == 3 and cmid == 16, 'stem kernel is specialised for 3->16'
        # stem
        self.units['stem'] = dict(off=off, cin=3, cout=cmid)
        off = self._add('backbone.model0.conv1.weight', off, (cmid, 3, 3, 3))
        off = self._add('backbone.model0.conv1.bias', off, (cmid,))
        off = self._add('backbone.model0.bn1.weight', off, (cmid,))
        off = self._add('backbone.model0.bn1.bias', off, (cmid,))
        self.bn_names.append('backbone.model0.bn1')
        for prefix, ci, co, with_bn in self.dp_units(arch, heads=False):
            self.units[prefix] = dict(off=off, cin=ci, cout=co, bn=with_bn)
            off = self._add(prefix + '.conv1.weight', off, (co, ci, 1, 1))
            off = self._add(prefix + '.conv1.bias', off, (co,))
            off = self._add(prefix + '.conv2.weight', off, (co, 1, 3, 3))
            off = self._add(prefix + '.conv2.bias', off, (co,))
            if with_bn:
                off = self._add(prefix + '.bn.weight', off, (co,))
                off = self._add(prefix + '.bn.bias', off, (co,))
                self.bn_names.append(prefix + '.bn')
        fc = arch['feat_channels']
        kp = 2 * arch['kps_num']
        self.head_rows = [('cls', 0, 1), ('bbox', 1, 4), ('obj', 5, 1), ('kps', 6, kp)]
        self.head_cout = 6 + kp
        assert self.head_cout == 16, 'fused head kernel is specialised for 16 outputs'
        # Towers (stacked_convs > 0): the cls map comes from the cls tower, bbox / obj / kps from the reg tower.  Each tower
        # feeds its OWN fused 64 -> 16 unit -- `headc.{l}` holds the cls rows, `head.{l}` the other 15 -- whose foreign rows
        # stay zero (they are not parameters: no state_dict entry, zero gradient through a masked dy_scale, weight decay
        # of zero is zero); the two [N,P,16] outputs are added (YUNET_OP_ADD).
        self.towers = arch.get('stacked_convs', 0) > 0
        for l in range(len(arch['strides'])):
            for name, rows in ((f'head.{l}', [r for r in self.head_rows if not (self.towers and r[0] == 'cls')]),
                               (f'headc.{l}', [r for r in self.head_rows if r[0] == 'cls'] if self.towers else [])):
                if not rows:
                    continue
                self.units[name] = dict(off=off, cin=fc, cout=self.head_cout, bn=False)
                base = off
                for nm, r0, nr in rows:
                    self.entries[f'bbox_head.multi_level_{nm}.{l}.conv1.weight'] = \
                        (base + r0 * fc, (nr, fc, 1, 1))
                base += self.head_cout * fc
                for nm, r0, nr in rows:
                    self.entries[f'bbox_head.multi_level_{nm}.{l}.conv1.bias'] = (base + r0, (nr,))
                base += self.head_cout
                for nm, r0, nr in rows:
                    self.entries[f'bbox_head.multi_level_{nm}.{l}.conv2.weight'] = \
                        (base + r0 * 9, (nr, 1, 3, 3))
                base += self.head_cout * 9
                for nm, r0, nr in rows:
                    self.entries[f'bbox_head.multi_level_{nm}.{l}.conv2.bias'] = (base + r0, (nr,))
                base += self.head_cout
                off = base
        self.numel = off

    def _add(self, key, off, shape):
        self.entries[key] = (off, tuple(shape))
        return off + int(math.prod(shape))

    @staticmethod
    def dp_units(arch, heads=True):
        out = []
        st = arch['stage_channels']
        out.append(('backbone.model0.conv2', st[0][1], st[0][2], True))
        for i in range(1, len(st)):
            ci, co = st[i]
            out.append((f'backbone.model{i}.conv1', ci, ci, True))
            out.append((f'backbone.model{i}.conv2', ci, co, True))
        for i, c in enumerate(arch['neck_channels']):
            out.append((f'neck.lateral_convs.{i}', c, c, True))
        fc = arch['feat_channels']
        for l in range(len(arch['strides'])):
            for j in range(arch['shared_stacked_convs']):
                out.append((f'bbox_head.multi_level_share_convs.{l}.{j}', fc, fc, True))
        for tower in ('cls', 'reg'):          # per-level towers (yunet_head.py:126-140), state_dict order
            for l in range(len(arch['strides'])):
                for j in range(arch.get('stacked_convs', 0)):
                    out.append((f'bbox_head.multi_level_{tower}_convs.{l}.{j}', fc, fc, True))
        return out

    def unit_ptrs(self, flat, name):
        """device pointers (w_pw, b_pw, w_dw, b_dw, gamma, beta) of a ConvDPUnit block."""
        u = self.units[name]
        ci, co = u['cin'], u['cout']
        base = flat.data_ptr() + 4 * u['off']
        o = [0, co * ci, co * ci + co, co * ci + co + co * 9, co * ci + 2 * co + co * 9,
             co * ci + 3 * co + co * 9]
        return [base + 4 * x for x in o]

    def unit_width(self, name):
        u = self.units[name]
        return K.dp_row_width(u['cin'], u['cout'])


class FlatParams:
    """Owns the flat parameter / gradient / momentum buffers and the BN running stats."""

    def __init__(self, layout, device):
        self.layout = layout
        self.device = device
        self.data = torch.zeros(layout.numel, device=device, dtype=torch.float32)
        # One allocation  [ log scalars (LOG_HEAD floats) | gradient of every parameter ]: the five
        # logged loss values sit in front of the gradient so that they travel in the gradient
        # all-reduce of the last bucket ("logging scalars folded into the same buffer").
        self.grad_buf = torch.zeros(LOG_HEAD + layout.numel, device=device, dtype=torch.float32)
        self.log_head = self.grad_buf[:LOG_HEAD]
        self.grad = self.grad_buf[LOG_HEAD:]
        nb = len(layout.bn_names)
        self.bn_channels = []
        for name in layout.bn_names:
            self.bn_channels.append(layout.entries[name + '.weight'][1][0])
        tot = sum(self.bn_channels)
        self.running_mean = torch.zeros(tot, device=device, dtype=torch.float32)
        self.running_var = torch.ones(tot, device=device, dtype=torch.float32)
        self.num_batches_tracked = torch.zeros(nb, device=device, dtype=torch.int64)
        self.bn_offset = {}
        o = 0
        for name, c in zip(layout.bn_names, self.bn_channels):
            self.bn_offset[name] = o
            o += c

    def view(self, key, of=None):
        off, shape = self.layout.entries[key]
        buf = self.data if of is None else of
        return buf[off:off + int(math.prod(shape))].view(shape)

    def load_state_dict(self, sd):
        with torch.no_grad():
            for key in self.layout.entries:
                self.view(key).copy_(sd[key].to(self.device, torch.float32))
            for i, name in enumerate(self.layout.bn_names):
                o, c = self.bn_offset[name], self.bn_channels[i]
                if name + '.running_mean' in sd:
                    self.running_mean[o:o + c].copy_(sd[name + '.running_mean'])
                    self.running_var[o:o + c].copy_(sd[name + '.running_var'])
                if name + '.num_batches_tracked' in sd:
                    self.num_batches_tracked[i] = int(sd[name + '.num_batches_tracked'])

    def state_dict(self):
        sd = {}
        for key in self.layout.entries:
            sd[key] = self.view(key).detach().clone()
        for i, name in enumerate(self.layout.bn_names):
            o, c = self.bn_offset[name], self.bn_channels[i]
            sd[name + '.running_mean'] = self.running_mean[o:o + c].clone()
            sd[name + '.running_var'] = self.running_var[o:o + c].clone()
            sd[name + '.num_batches_tracked'] = self.num_batches_tracked[i].clone()
        return sd


# ================================================================================ engine
class _T:
    """An activation tensor of the plan: raw buffer (+ the BN its consumers must apply)."""

    def __init__(self, buf, n, h, w, c, bn=None, img_stride=None):
        self.buf, self.n, self.h, self.w, self.c, self.bn = buf, n, h, w, c, bn
        self.img_stride = img_stride if img_stride is not None else h * w * c
        self.grad = None
        self.grad_written = False
        self.bn_count = n * h * w   # elements per channel behind the BN statistics of `bn`
        self.pooled_into = None     # (pooled tensor, window-position bytes) when max_pool2d is this tensor's only consumer
        self.plain_pool = False     # a pool_bwd op of its own will write this tensor's gradient (_pool, unfused form)
        self.fuse_upadd = False     # ... and applies the TFPN merge's share of it too (_upadd: `upadd_share` = that gradient)
        self.upadd_share = None


class _BNRef:
    def __init__(self, name, c, count, stats, bstats, gamma_ptr, beta_ptr):
        self.name, self.c, self.count = name, c, count
        self.stats, self.bstats = stats, bstats      # (tensor views, fp64 [2c])
        self.gamma_ptr, self.beta_ptr = gamma_ptr, beta_ptr

    def c_struct(self):
        return L.YunetBN(self.stats.data_ptr(), self.bstats.data_ptr(), self.gamma_ptr,
                         self.beta_ptr, self.count, BN_EPS, BN_SLOTS)


_NULL_BN = L.YunetBN(None, None, None, None, 1, BN_EPS)


class Plan:
    """Buffers + op lists for one (N, H, W, Gmax) shape."""

    def __init__(self, eng, n, h, w, gmax):
        self.eng, self.n, self.h, self.w, self.gmax = eng, n, h, w, gmax
        arch, lay, fp, dev = eng.arch, eng.layout, eng.params, eng.device
        # activation storage of this plan: fp32, or bf16 ("bf16 fwd / fp32 grads", BASELINE configs[2]):
        # every tensor a forward kernel writes except the head output; gradients stay fp32
        self.act_dtype = torch.bfloat16 if eng.precision == 'bf16' else torch.float32
        self.act_flag = L.BF16 if eng.precision == 'bf16' else L.F32
        self.keep = []          # python objects that must outlive the op arrays
        self.producer = {}      # id(tensor) -> (cin, cout) of the ConvDPUnit that wrote it
        self.fwd_op_of = {}     # id(tensor) -> its OP_DP_FWD record
        self.fwd_a, self.fwd_b, self.bwd = [], [], []
        self.bwd_nodes = []     # (lane, closure generating backward ops, lane to JOIN first | None), in fwd order
        self._lane = 0          # executor lane of the ops being appended (0 = the caller's stream)
        self._join_before = None
        self.tensors = {}       # unit name -> (input _T, output _T): introspection / debugging
        f32 = dict(device=dev, dtype=torch.float32)

        # ---- BN statistic buffers: one fp64 block, zeroed by a single memset per step
        # (every layer's sums are BN_SLOTS replicas [BN_SLOTS, 2c]: YunetBN::slots)
        tot_c = sum(fp.bn_channels) * BN_SLOTS
        self.stats = torch.zeros(4 * tot_c, device=dev, dtype=torch.float64)
        self.bn = {}
        o = 0
        for name, c in zip(lay.bn_names, fp.bn_channels):
            g_off = lay.entries[name + '.weight'][0]
            b_off = lay.entries[name + '.bias'][0]
            c2 = 2 * c * BN_SLOTS
            self.bn[name] = dict(c=c, stats=self.stats[o:o + c2].view(BN_SLOTS, 2 * c),
                                 bstats=self.stats[2 * tot_c + o:2 * tot_c + o + c2].view(BN_SLOTS, 2 * c),
                                 gamma=fp.data.data_ptr() + 4 * g_off,
                                 beta=fp.data.data_ptr() + 4 * b_off,
                                 dgamma=fp.grad.data_ptr() + 4 * g_off,
                                 dbeta=fp.grad.data_ptr() + 4 * b_off)
            o += c2
        self.ops_memset_stats = self._op(L.OP_MEMSET, p=[self.stats.data_ptr()],
                                         i=self._split64(self.stats.numel() * 8))
        self.fwd_a.append(self.ops_memset_stats)

        # ---- loss-step geometry
        strides = arch['strides']
        self.sizes = [(h // s, w // s) for s in strides]
        self.P = sum(a * b for a, b in self.sizes)
        self.levels = K.make_levels(self.sizes, strides)
        self.flat = torch.empty(n, self.P, 16, **f32)
        self.dflat = torch.empty(n, self.P, 16, **f32)
        # towers (stacked_convs > 0): the cls tower's head unit writes its own [N,P,16] (zeros outside the cls channel)
        self.towers = lay.towers
        self.flat_c = torch.empty(n, self.P, 16, **f32) if self.towers else None
        # ... and each tower's unit takes dy_scale masked to its own channels (engine.backward refreshes them)
        self.mask_cls = torch.tensor([1.0] + [0.0] * 15, **f32)
        self.dy_scale_cls = self.mask_cls.clone()
        self.dy_scale_reg = 1.0 - self.mask_cls

        # ---- conv stack
        self.img_ptr_ops = []
        st = arch['stage_channels']
        cmid = st[0][1]
        z0 = self._new_t(n, h // 2, w // 2, cmid, bn_name='backbone.model0.bn1')
        self._stem(z0)
        cur = self._dp(z0, 'backbone.model0.conv2')
        taps = []
        for i in range(len(st)):
            if i > 0:
                cur = self._dp(cur, f'backbone.model{i}.conv1')
                cur = self._dp(cur, f'backbone.model{i}.conv2')
            if i in arch['out_idx']:
                taps.append(cur)
            if i in arch['downsample_idx']:
                # the pool is the stage output's only consumer unless the stage is also tapped by the neck
                cur = self._pool(cur, sole_consumer=i not in arch['out_idx'])
        # The head chain of a pyramid level (share convs -> fused head) depends on that level's lateral conv only,
        # and the levels are mutually independent (yunet_head.py:175-247 loops over them): the chains of the coarser
        # levels CAN run on executor lanes (side streams) next to the rest of the top-down pathway; level 0 stays on
        # the caller's stream.  Off by default (YUNET_LANES=1 turns it on): measured on YuNet_n 320 x 320 bs 256 the
        # step is 5.375 ms with lanes vs 5.340 ms without -- a packed 20 x 20 launch already has 882 tiles for the
        # 256 CUs, only the 10 x 10 launches (242 tiles, 23 us each) leave CUs idle, and the backward kernel's
        # 149 KB of LDS admits one workgroup per CU, so concurrent launches queue instead of sharing CUs.
        feats = list(taps)
        level_of = {i: l for l, i in enumerate(arch['neck_out_idx'])}
        bases, b0 = [], 0
        for hh, ww in self.sizes:
            bases.append(b0)
            b0 += hh * ww
        lanes_ok = (bool(os.environ.get('YUNET_LANES')) or bool(getattr(eng, 'use_lanes', False))) and not self.towers
        self.lanes_used = 0

        def head_chain(i):
            l = level_of.get(i)
            if l is None:
                return
            f = feats[i]
            assert (f.h, f.w) == self.sizes[l], 'feature sizes vs strides'
            lane = l if (lanes_ok and 0 < l <= L.MAX_LANES) else 0
            if lane:
                self.fwd_a.append(self._op(L.OP_FORK, i=[1 << lane]))
                self.lanes_used |= 1 << lane
                f.head_lane = lane
            self._lane = lane
            for j in range(arch['shared_stacked_convs']):
                f = self._dp(f, f'bbox_head.multi_level_share_convs.{l}.{j}')
            if self.towers:
                # yunet_head.py:191-207: two towers on the same feature; cls from one, bbox / obj / kps from the other
                fc_, fr_ = f, f
                for j in range(arch['stacked_convs']):
                    fc_ = self._dp(fc_, f'bbox_head.multi_level_cls_convs.{l}.{j}')
                for j in range(arch['stacked_convs']):
                    fr_ = self._dp(fr_, f'bbox_head.multi_level_reg_convs.{l}.{j}')
                self._head(fr_, l, bases[l])
                self._head(fc_, l, bases[l], cls_tower=True)
            else:
                self._head(f, l, bases[l])
            self._lane = 0
        # Round 5: the head chains wait until the whole top-down pathway is built, and the share convs of the levels --
        # mutually independent plain 64 -> 64 units -- are emitted next to each other as ONE group
        # (YunetOp.i[OP_GROUP]): the executor launches them as one grid (yunet_dp_fwd_group).  On the 20 x 20 / 10 x 10
        # levels a launch of their own is a prologue, one or two bands per wave and a drain (27 / 13 us for 13 + 3 MB).
        # Needs no lanes (one stream); the tower head (stacked_convs > 0) keeps the per-level order.
        grouped = (not lanes_ok and not self.towers and not os.environ.get('YUNET_NO_HEAD_GROUP')
                   and arch['shared_stacked_convs'] >= 1)
        for i in range(len(feats) - 1, 0, -1):
            feats[i] = self._dp(feats[i], f'neck.lateral_convs.{i}')
            if not grouped:
                head_chain(i)
            # the backward of this merge ACCUMULATES into feats[i]'s gradient after the level's head chain (on its
            # lane) has written it: the executor joins that lane first
            self._join_before = getattr(feats[i], 'head_lane', None)
            feats[i - 1] = self._upadd(feats[i - 1], feats[i])
            self._join_before = None
        feats[0] = self._dp(feats[0], 'neck.lateral_convs.0')
        if not grouped:
            head_chain(0)
        else:
            order = [i for i in range(len(feats)) if level_of.get(i) is not None]        # finest (largest) map first
            cur = {}
            for i in order:
                assert (feats[i].h, feats[i].w) == self.sizes[level_of[i]], 'feature sizes vs strides'
                cur[i] = feats[i]
            for j in range(arch['shared_stacked_convs']):
                for s0 in range(0, len(order), L.DP_GROUP_MAX):
                    part = order[s0:s0 + L.DP_GROUP_MAX]
                    first = len(self.fwd_a)
                    for i in part:
                        cur[i] = self._dp(cur[i], f'bbox_head.multi_level_share_convs.{level_of[i]}.{j}')
                    if len(part) >= 2:
                        assert len(self.fwd_a) == first + len(part)
                        self.fwd_a[first].i[L.OP_GROUP] = len(part)
            for i in order:
                self._head(cur[i], level_of[i], bases[level_of[i]])
        if self.lanes_used:
            self.fwd_a.append(self._op(L.OP_JOIN, i=[self.lanes_used]))
        if self.towers:      # flat = (reg tower's bbox | obj | kps channels) + (cls tower's cls channel): exact zeros elsewhere
            nel = n * self.P * 16
            self.fwd_a.append(self._op(L.OP_ADD, p=[self.flat.data_ptr(), self.flat_c.data_ptr(), self.flat.data_ptr()],
                                       i=[nel & 0xffffffff if (nel & 0xffffffff) < 2 ** 31 else (nel & 0xffffffff) - 2 ** 32,
                                          nel >> 32]))

        # ---- loss step
        self.gt_boxes = torch.zeros(n, gmax, 4, **f32)
        self.gt_kps = torch.zeros(n, gmax, 5, 3, **f32)
        self.gt_count = torch.zeros(n, device=dev, dtype=torch.int32)
        self.gt_inds = torch.empty(n, self.P, device=dev, dtype=torch.int32)
        self.max_overlaps = torch.empty(n, self.P, **f32)
        self.img_stats = torch.empty(n, 2, **f32)
        self.scratch = torch.empty(n, self.P, 12, **f32)     # yunet_assign work arrays
        self.norm = torch.zeros(4, **f32)
        self.losses = torch.zeros(8, **f32)        # cls, bbox, obj, kps, total (3 spare)
        lib = L.load()
        self.loss_blocks = lib.yunet_loss_blocks(n, self.P)
        self.loss_partials = torch.empty(self.loss_blocks, 4, **f32)
        self.dy_scale = torch.ones(16, **f32)
        self.dy_norm = torch.ones(16, **f32)      # deferred normaliser: per-channel 1 / num_total (loss_finalize_ex)
        self.dy_up = torch.ones(16, **f32)        # deferred mode: the upstream loss scales; dy_scale = dy_up * dy_norm
        self.deferred = False
        op = self._op(L.OP_ASSIGN,
                      p=[self.flat.data_ptr(), self.gt_boxes.data_ptr(), self.gt_kps.data_ptr(),
                         None, self.gt_count.data_ptr(), self.gt_inds.data_ptr(), None,
                         self.max_overlaps.data_ptr(), self.img_stats.data_ptr(),
                         self.scratch.data_ptr()],
                      i=[n, self.P, gmax, int(arch.get('candidate_topk', 10))],
                      f=[arch['center_radius'], float(arch.get('iou_weight', 3.0)), float(arch.get('cls_weight', 1.0))])
        op.lv = self.levels
        self.assign_idx = len(self.fwd_a)
        self.fwd_a.append(op)
        self.fwd_a.append(self._op(L.OP_LOSS_NORM, p=[self.img_stats.data_ptr(),
                                                     self.norm.data_ptr()],
                                   i=[n], f=[1.0 / eng.world_size]))
        op = self._op(L.OP_LOSS,
                      p=[self.flat.data_ptr(), self.gt_inds.data_ptr(),
                         self.max_overlaps.data_ptr(), self.gt_boxes.data_ptr(),
                         self.gt_kps.data_ptr(), self.norm.data_ptr(), self.dflat.data_ptr(),
                         self.loss_partials.data_ptr()],
                      i=[n, self.P, gmax, self.loss_blocks])
        op.lv = self.levels
        op.loss = K.make_loss_cfg(arch['loss_bbox'], arch['loss_cls_weight'],
                                  arch['loss_bbox_weight'], arch['loss_obj_weight'],
                                  arch['loss_kps_weight'], float(arch.get('loss_bbox_eps', 1e-6)),
                                  float(arch.get('loss_bbox_smooth_point', 0.1)), arch['kps_beta'],
                                  arch.get('loss_bbox_mode'))
        self.fwd_b.append(op)
        self.fwd_b.append(self._op(L.OP_LOSS_FINALIZE, p=[self.loss_partials.data_ptr(),
                                                         self.losses.data_ptr(),
                                                         fp.log_head.data_ptr()],
                                   i=[self.loss_blocks]))
        # BN running statistics (nn.BatchNorm2d momentum 0.1) and, in backward, d(gamma)/d(beta)
        # of ALL BatchNorm layers: one launch each, driven by a small device table
        rows_f, rows_b = [], []
        so = 0
        for name, c in zip(lay.bn_names, fp.bn_channels):
            cnt = self.bn_count[name]
            g_off = lay.entries[name + '.weight'][0]
            b_off = lay.entries[name + '.bias'][0]
            rows_f.append([so, c, cnt, fp.bn_offset[name], g_off, b_off, BN_SLOTS])
            rows_b.append([2 * tot_c + so, c, cnt, fp.bn_offset[name], g_off, b_off, BN_SLOTS])
            so += 2 * c * BN_SLOTS
        self.bn_table_f = torch.tensor(rows_f, dtype=torch.int32).to(dev)
        self.bn_table_b = torch.tensor(rows_b, dtype=torch.int32).to(dev)
        self.fwd_b.append(self._op(
            L.OP_BN_BATCH, p=[self.bn_table_f.data_ptr(), self.stats.data_ptr(),
                              fp.running_mean.data_ptr(), fp.running_var.data_ptr(), None],
            i=[len(rows_f), 0], f=[BN_MOMENTUM]))

        # ---- backward: reverse of the forward nodes; the head chains of the lanes first (their gradients exist
        # from the start), each on its lane, then the caller's stream walks the rest and joins a lane right before
        # the first op that accumulates into a gradient that lane wrote
        self.reduce_jobs = []   # (partials ptr, grad ptr, rows, width, accumulate) of every unit
        marks = []              # after each backward node: (#ops, #reduce jobs)
        if self.lanes_used:
            self.bwd.append(self._op(L.OP_FORK, i=[self.lanes_used]))
        joined = 0
        order = [nd for nd in reversed(self.bwd_nodes) if nd[0] > 0] + [nd for nd in reversed(self.bwd_nodes) if nd[0] == 0]
        for lane, node, join in order:
            if join and not (joined >> join) & 1:
                self.bwd.append(self._op(L.OP_JOIN, i=[1 << join]))
                joined |= 1 << join
            self._lane = lane
            node()
            self._lane = 0
            marks.append((len(self.bwd), len(self.reduce_jobs)))
        if self.lanes_used & ~joined:
            self.bwd.append(self._op(L.OP_JOIN, i=[self.lanes_used & ~joined]))
        kernels_bwd = list(self.bwd)
        # all weight-gradient partial reductions in ONE launch (table lives on the device)
        self.reduce_table, chunk = K.reduce_job_table(self.reduce_jobs, dev)
        self.bwd.append(self._op(L.OP_REDUCE_BATCH, p=[self.reduce_table.data_ptr()],
                                 i=[len(self.reduce_jobs), chunk]))
        self.bwd.append(self._op(
            L.OP_BN_BATCH, p=[self.bn_table_b.data_ptr(), self.stats.data_ptr(), None, None,
                              fp.grad.data_ptr()], i=[len(rows_b), 1], f=[0.0]))
        self.c_fwd_a = self._carray(self.fwd_a)
        self.c_fwd_b = self._carray(self.fwd_b)
        self.c_bwd = self._carray(self.bwd)
        # N > 1: the same two phases with the num_pos normaliser DEFERRED -- the loss kernel leaves the cls / bbox /
        # obj terms un-normalised and does not read norm[0], so its all-reduce runs on the side stream BESIDE the
        # loss kernel; loss_finalize applies 1 / max(num_total, 1) to the logged losses and writes the per-channel
        # factor the fused head units take as dy_scale (engine.forward / backward).  x 1.0 is exact: the gradients
        # are bit-identical to the undeferred form.
        assert self.fwd_b[0].opcode == L.OP_LOSS and self.fwd_b[1].opcode == L.OP_LOSS_FINALIZE
        loss_def, fin_def = L.YunetOp(), L.YunetOp()
        C.memmove(C.byref(loss_def), C.byref(self.fwd_b[0]), C.sizeof(L.YunetOp))
        C.memmove(C.byref(fin_def), C.byref(self.fwd_b[1]), C.sizeof(L.YunetOp))
        loss_def.loss.defer_num_total = 1
        fin_def.p[3] = self.norm.data_ptr()
        fin_def.p[4] = self.dy_norm.data_ptr()
        self.c_fwd_b_loss = self._carray([loss_def])
        self.c_fwd_b_rest = self._carray([fin_def] + self.fwd_b[2:])
        self.own_gt = (self.gt_boxes, self.gt_kps, self.gt_count)     # the plan's own staging buffers (bind_gt)
        self._gt_bound = tuple(t.data_ptr() for t in self.own_gt)

        # ---- world > 1: the same backward in TWO segments so that the gradient all-reduce of the
        # first bucket (head, neck and the backbone stages from the first pyramid tap on: the tail
        # of the flat buffer, ~85 % of the parameters) runs on a side stream underneath the backward
        # kernels of the early, high-resolution stages (stem .. model2: most of the backward time).
        # The reference gets this overlap from the DDP reducer (mmdet/apis/train.py:156-161).
        # A unit's weight-gradient partials and the d(gamma)/d(beta) of its BatchNorm are final once
        # its own backward kernel has run (the BN sums come from its consumers, which ran earlier).
        self.split_off = None
        split_unit = f"backbone.model{min(arch['out_idx'])}.conv1"
        if split_unit in lay.units and lay.units[split_unit]['off'] > 0:
            split_off = lay.units[split_unit]['off']
            gbase = fp.grad.data_ptr()
            done = [j for j in self.reduce_jobs if j[1] - gbase >= 4 * split_off]
            # the jobs of bucket A must be a prefix of the execution order
            n_a = len(done)
            if n_a and all(j[1] - gbase >= 4 * split_off for j in self.reduce_jobs[:n_a]):
                ops_a = max(m[0] for m in marks if m[1] <= n_a)
                tab_a, chunk_a = K.reduce_job_table(self.reduce_jobs[:n_a], dev)
                tab_b, chunk_b = K.reduce_job_table(self.reduce_jobs[n_a:], dev)
                rb_a = [r for r in rows_b if r[4] >= split_off]
                rb_b = [r for r in rows_b if r[4] < split_off]
                self.bn_table_ba = torch.tensor(rb_a, dtype=torch.int32).to(dev)
                self.bn_table_bb = torch.tensor(rb_b, dtype=torch.int32).to(dev)
                self.keep += [tab_a, tab_b]

                def tail_ops(tab, nj, ch, bnt, nb):
                    return [self._op(L.OP_REDUCE_BATCH, p=[tab.data_ptr()], i=[nj, ch]),
                            self._op(L.OP_BN_BATCH, p=[bnt.data_ptr(), self.stats.data_ptr(), None, None,
                                                       fp.grad.data_ptr()], i=[nb, 1], f=[0.0])]
                self.bwd_a = kernels_bwd[:ops_a] + tail_ops(tab_a, n_a, chunk_a, self.bn_table_ba, len(rb_a))
                self.bwd_b = kernels_bwd[ops_a:] + tail_ops(tab_b, len(self.reduce_jobs) - n_a, chunk_b,
                                                            self.bn_table_bb, len(rb_b))
                self.c_bwd_a = self._carray(self.bwd_a)
                self.c_bwd_b = self._carray(self.bwd_b)
                # one GPU: segment A's kernels alone + its reduction as a list of its own, which then runs on the side
                # stream under the kernels of segment B (engine.backward)
                self.c_bwd_a_k = self._carray(self.bwd_a[:ops_a])
                self.c_tail_a = self._carray(self.bwd_a[ops_a:])
                self.split_off, self.split_ops = split_off, ops_a

        # ---- eval(): the same conv-stack launches with BatchNorm on the running statistics
        # (op 0 fills the sums from running_mean / running_var instead of zeroing them; producers
        # do not accumulate: ConvDPUnits get out_has_bn = 0, the stem sums into a scratch block)
        self.eval_scratch = torch.zeros(64, device=dev, dtype=torch.float64)
        self.fwd_eval = [self._op(L.OP_BN_BATCH, p=[self.bn_table_f.data_ptr(), self.stats.data_ptr(),
                                                    fp.running_mean.data_ptr(), fp.running_var.data_ptr(), None],
                                  i=[len(rows_f), 2], f=[0.0])]
        for op in self.fwd_a[1:]:
            if op.opcode in (L.OP_ASSIGN, L.OP_LOSS_NORM):
                continue          # test time: no SimOTA on stale GT, gt_inds / norm stay untouched
            cp = L.YunetOp()
            C.memmove(C.byref(cp), C.byref(op), C.sizeof(L.YunetOp))
            if cp.opcode == L.OP_DP_FWD:
                cp.dp.out_has_bn = 0
            elif cp.opcode == L.OP_STEM_FWD:
                cp.p[4] = self.eval_scratch.data_ptr()
            self.fwd_eval.append(cp)
        self.c_fwd_eval = self._carray(self.fwd_eval)

    # ------------------------------------------------------------------ helpers
    bn_count = None

    @staticmethod
    def _split64(v):
        return [v & 0xffffffff if (v & 0xffffffff) < 2 ** 31 else (v & 0xffffffff) - 2 ** 32,
                (v >> 32) & 0x7fffffff]

    def _op(self, opcode, p=(), i=(), f=()):
        op = L.YunetOp()
        op.opcode = opcode
        op.i[L.OP_LANE] = self._lane if opcode not in (L.OP_FORK, L.OP_JOIN) else 0
        op.i[11] = self.act_flag          # stem / pool / upsample-add ops: activation storage type
        for k, v in enumerate(p):
            op.p[k] = v
        for k, v in enumerate(i):
            op.i[k] = int(v)
        for k, v in enumerate(f):
            op.f[k] = float(v)
        return op

    def _carray(self, ops):
        arr = (L.YunetOp * len(ops))()
        for k, op in enumerate(ops):
            C.memmove(C.byref(arr, k * C.sizeof(L.YunetOp)), C.byref(op), C.sizeof(L.YunetOp))
        return arr

    def _bn_struct(self, name, count):
        if self.bn_count is None:
            self.bn_count = {}
        self.bn_count[name] = count
        b = self.bn[name]
        return L.YunetBN(b['stats'].data_ptr(), b['bstats'].data_ptr(), b['gamma'], b['beta'],
                         count, BN_EPS, BN_SLOTS)

    def _new_t(self, n, h, w, c, bn_name=None):
        buf = torch.empty(n, h, w, c, device=self.eng.device, dtype=self.act_dtype)
        t = _T(buf, n, h, w, c, bn=bn_name)
        return t

    def _grad_of(self, t):
        """(grad buffer, accumulate flag) for a consumer's backward; first writer overwrites."""
        if t.grad is None:
            t.grad = torch.empty(t.buf.shape, device=t.buf.device, dtype=torch.float32)   # gradients: always fp32
        acc = t.grad_written
        t.grad_written = True
        return t.grad, int(acc)

    # ------------------------------------------------------------------ nodes
    def _stem(self, z0):
        lay, fp = self.eng.layout, self.eng.params
        u = lay.units['stem']
        wp = fp.data.data_ptr() + 4 * u['off']
        bp = wp + 4 * 16 * 27
        bn = self.bn['backbone.model0.bn1']
        cnt = z0.n * z0.h * z0.w
        self._bn_struct('backbone.model0.bn1', cnt)
        op = self._op(L.OP_STEM_FWD, p=[None, wp, bp, z0.buf.data_ptr(), bn['stats'].data_ptr()],
                      i=[self.n, self.h, self.w, 16])
        self.fwd_a.append(op)
        self.img_ptr_ops.append(('fwd_a', len(self.fwd_a) - 1))
        blocks = K.stem_grid(self.n, self.h, self.w)
        width = 16 * 27 + 16
        part = torch.empty(blocks, width, device=self.eng.device, dtype=torch.float32)
        self.keep.append(part)
        gptr = fp.grad.data_ptr() + 4 * u['off']

        def bwd():
            assert z0.grad is not None
            op = self._op(L.OP_STEM_BWD, p=[None, z0.buf.data_ptr(), z0.grad.data_ptr(),
                                            part.data_ptr(), wp, bp],      # wp, bp: fp32 storage recomputes z from the image
                          i=[self.n, self.h, self.w, 16, blocks])
            op.bn[0] = self._bn_struct('backbone.model0.bn1', cnt)
            self.bwd.append(op)
            self.img_ptr_ops.append(('bwd', len(self.bwd) - 1))
            self.reduce_jobs.append((part.data_ptr(), gptr, blocks, width, 0))
        self.bwd_nodes.append((self._lane, bwd, self._join_before))

    def _dp_desc(self, x, name, z, z_img_stride=None):
        lay, fp = self.eng.layout, self.eng.params
        u = lay.units[name]
        d = L.YunetDP()
        d.N, d.H, d.W, d.cin, d.cout = x.n, x.h, x.w, u['cin'], u['cout']
        assert x.c == u['cin']
        d.in_transform = L.T_BNRELU if x.bn is not None else L.T_IDENTITY
        d.out_has_bn = 1 if u['bn'] else 0
        d.x_img_stride = x.img_stride
        d.z_img_stride = z_img_stride if z_img_stride is not None else x.h * x.w * u['cout']
        d.x = x.buf.data_ptr()
        d.in_bn = self._bn_struct(x.bn, x.bn_count) if x.bn is not None else _NULL_BN
        ptrs = lay.unit_ptrs(fp.data, name)
        d.w_pw, d.b_pw, d.w_dw, d.b_dw = ptrs[0], ptrs[1], ptrs[2], ptrs[3]
        d.z = z
        d.x_dtype = self.act_flag
        d.z_dtype = self.act_flag if u['bn'] else L.F32          # the fused heads write the fp32 [N,P,16]
        bn_name = name + '.bn' if u['bn'] else None
        d.out_bn = self._bn_struct(bn_name, x.n * x.h * x.w) if bn_name else _NULL_BN
        return d, u, bn_name

    def _dp(self, x, name):
        u = self.eng.layout.units[name]
        z = self._new_t(x.n, x.h, x.w, u['cout'], bn_name=name + '.bn')
        self.producer[id(z)] = (u['cin'], u['cout'])
        self._dp_node(x, name, z.buf.data_ptr(), None, z)
        self.tensors[name] = (x, z)
        return z

    def _head(self, x, level, base, cls_tower=False):
        name = f'headc.{level}' if cls_tower else f'head.{level}'
        out = self.flat_c if cls_tower else self.flat
        zptr = out.data_ptr() + 4 * base * 16
        scale = None
        if self.towers:      # each tower's unit only sees the gradient of its own channels (its other rows stay zero)
            scale = self.dy_scale_cls if cls_tower else self.dy_scale_reg
        self._dp_node(x, name, zptr, self.P * 16, None, dy_ptr=self.dflat.data_ptr() + 4 * base * 16, dy_scale=scale)

    def _dp_node(self, x, name, zptr, z_img_stride, zt, dy_ptr=None, dy_scale=None):
        d, u, bn_name = self._dp_desc(x, name, zptr, z_img_stride)
        op = self._op(L.OP_DP_FWD)
        op.dp = d
        self.fwd_a.append(op)
        if zt is not None:
            self.fwd_op_of[id(zt)] = op          # _pool() may attach the fused pooling outputs
        blocks = K.dp_grid(x.n, x.h, x.w, u['cin'], u['cout'])
        width = K.dp_row_width(u['cin'], u['cout'])
        part = torch.empty(blocks, width, device=self.eng.device, dtype=torch.float32)
        self.keep.append(part)
        gptr = self.eng.params.grad.data_ptr() + 4 * u['off']

        def bwd():
            d2, _, _ = self._dp_desc(x, name, zptr, z_img_stride)
            if zt is not None and zt.pooled_into is not None:
                # fused max_pool2d backward: dy = the pooled gradient its consumer wrote + argmax bytes
                out, idx = zt.pooled_into
                assert out.grad is not None, f'{name}: pooled output has no gradient'
                d2.dy = out.grad.data_ptr()
                d2.pool_idx = idx.data_ptr()
                d2.dy_scale = None
            elif zt is not None:
                assert zt.grad is not None, f'{name}: output has no gradient'
                d2.dy = zt.grad.data_ptr()
                d2.dy_scale = None
            else:
                d2.dy = dy_ptr
                d2.dy_scale = (dy_scale if dy_scale is not None else self.dy_scale).data_ptr()
            gx, acc = self._grad_of(x)
            d2.dx = gx.data_ptr()
            d2.accumulate_dx = acc
            d2.wgrad_partials, d2.wgrad_blocks = part.data_ptr(), blocks
            op = self._op(L.OP_DP_BWD)
            op.dp = d2
            self.bwd.append(op)
            self.reduce_jobs.append((part.data_ptr(), gptr, blocks, width, 0))
        self.bwd_nodes.append((self._lane, bwd, self._join_before))

    def _pool(self, x, sole_consumer=False):
        # Fused pooling: when the pool is the only consumer of x and the producing unit has the
        # build for it, that unit's forward also writes the raw window winners + their positions, the
        # pool's consumer reads them through the ordinary BN+ReLU input transform of x's BatchNorm, and the
        # producer's backward expands the pooled gradient on load: no pooling kernels, no full-size dz.
        prod = self.producer.get(id(x))
        if (sole_consumer and prod is not None and not os.environ.get('YUNET_NO_POOL_FUSION') and
                L.load().yunet_dp_pool_fusion_ok(x.n, x.h, x.w, prod[0], prod[1])):
            out = self._new_t(x.n, x.h // 2, x.w // 2, x.c, bn_name=x.bn)
            out.bn_count = x.bn_count
            idx = torch.empty(x.n, x.h // 2, x.w // 2, x.c, device=self.eng.device, dtype=torch.uint8)
            self.keep.append(idx)
            fop = self.fwd_op_of[id(x)]
            fop.dp.pool_out, fop.dp.pool_idx = out.buf.data_ptr(), idx.data_ptr()
            x.pooled_into = (out, idx)
            return out
        out = self._new_t(x.n, x.h // 2, x.w // 2, x.c, bn_name=None)
        cnt = x.n * x.h * x.w
        op = self._op(L.OP_POOL_FWD, p=[x.buf.data_ptr(), out.buf.data_ptr()],
                      i=[x.n, x.h, x.w, x.c])
        op.bn[0] = self._bn_struct(x.bn, cnt)
        self.fwd_a.append(op)
        x.plain_pool = True

        def bwd():
            # a pyramid tap: its other consumer is the identity branch of the TFPN merge, whose backward ran earlier (the
            # merge comes later in the forward) and left its share of the gradient to this op -- same ReLU mask, same
            # BatchNorm sums, so the tap's gradient is written ONCE instead of written, re-read and re-written
            extra = None
            if x.fuse_upadd:
                assert x.upadd_share is not None, 'the merge backward must precede the pool backward of its tap'
                extra = x.upadd_share.data_ptr()
            gx, acc = self._grad_of(x)
            op = self._op(L.OP_POOL_BWD, p=[x.buf.data_ptr(), out.grad.data_ptr(), gx.data_ptr(), extra],
                          i=[x.n, x.h, x.w, x.c, acc])
            op.bn[0] = self._bn_struct(x.bn, cnt)
            self.bwd.append(op)
        self.bwd_nodes.append((self._lane, bwd, self._join_before))
        return out

    def _upadd(self, a, b):
        assert a.bn is not None and b.bn is not None and a.h == 2 * b.h and a.w == 2 * b.w
        out = self._new_t(a.n, a.h, a.w, a.c, bn_name=None)
        op = self._op(L.OP_UPADD_FWD, p=[a.buf.data_ptr(), b.buf.data_ptr(), out.buf.data_ptr()],
                      i=[a.n, a.h, a.w, a.c])
        op.bn[0] = self._bn_struct(a.bn, a.n * a.h * a.w)
        op.bn[1] = self._bn_struct(b.bn, b.n * b.h * b.w)
        self.fwd_a.append(op)
        # `a` is also max-pooled by a kernel of its own (built earlier in the forward = run later in the backward):
        # leave a's share of the gradient to that kernel (see _pool)
        a.fuse_upadd = a.plain_pool and not os.environ.get('YUNET_NO_UPADD_POOL_FUSION')

        def bwd():
            if a.fuse_upadd:
                a.upadd_share = out.grad
                ga_ptr, acc_a = None, 0
            else:
                ga, acc_a = self._grad_of(a)
                ga_ptr = ga.data_ptr()
            gb, acc_b = self._grad_of(b)
            op = self._op(L.OP_UPADD_BWD,
                          p=[a.buf.data_ptr(), b.buf.data_ptr(), out.grad.data_ptr(),
                             ga_ptr, gb.data_ptr()],
                          i=[a.n, a.h, a.w, a.c, acc_a, acc_b])
            op.bn[0] = self._bn_struct(a.bn, a.n * a.h * a.w)
            op.bn[1] = self._bn_struct(b.bn, b.n * b.h * b.w)
            self.bwd.append(op)
        self.bwd_nodes.append((self._lane, bwd, self._join_before))
        return out

    def bind_gt(self, boxes, kps, count):
        """Point the assignment and loss ops at padded GT tensors ([N,Gmax,4] fp32, [N,Gmax,5,3] fp32, [N] int32, contiguous, on
        this device) -- the plan's own staging buffers, or the data source's when it delivers them in this layout
        (synthetic.GTList / the device input pipeline): the step then reads them in place, like the image, instead of copying
        them first (three copy kernels with ~10 us between them in front of every step, profiles/r05_trace_gaps.json).  The
        tensors stay referenced here until the next bind."""
        ptrs = (boxes.data_ptr(), kps.data_ptr(), count.data_ptr())
        self.gt_boxes, self.gt_kps, self.gt_count = boxes, kps, count
        if ptrs == self._gt_bound:
            return
        a = self.c_fwd_a[self.assign_idx]
        assert a.opcode == L.OP_ASSIGN
        a.p[1], a.p[2], a.p[4] = ptrs
        for arr in (self.c_fwd_b, self.c_fwd_b_loss):
            assert arr[0].opcode == L.OP_LOSS
            arr[0].p[3], arr[0].p[4] = ptrs[0], ptrs[1]
        self._gt_bound = ptrs

    def set_img(self, img):
        ptr = img.data_ptr()
        for which, idx in self.img_ptr_ops:
            arr = {'fwd_a': self.c_fwd_a, 'bwd': self.c_bwd}[which]
            arr[idx].p[0] = ptr
            if which == 'fwd_a':
                self.c_fwd_eval[idx].p[0] = ptr       # same position: op 0 is replaced, not removed
            elif self.split_off is not None:          # the two-segment copy of the backward list
                if idx < self.split_ops:
                    self.c_bwd_a[idx].p[0] = ptr
                    self.c_bwd_a_k[idx].p[0] = ptr
                else:
                    self.c_bwd_b[idx - self.split_ops].p[0] = ptr


class YuNetEngine:
    """Runs the YuNet training step of one rank; see module docstring."""

    def __init__(self, arch, device, world_size=1, process_group=None):
        self.arch = dict(arch)
        self.device = torch.device(device)
        self.layout = ParamLayout(arch)
        self.params = FlatParams(self.layout, self.device)
        self.world_size = world_size
        self.process_group = process_group
        self.plans = collections.OrderedDict()      # (N, H, W, Gmax, precision) -> Plan, least recently used first
        self.plan = None
        self.always_bucket = False      # tests: run the two-segment backward + collectives at world size 1
        self.use_lanes = False          # head chains of the coarser levels on executor side streams (Plan.__init__)
        self.overlap_reduce = os.environ.get('YUNET_OVERLAP_REDUCE', '1') != '0'     # one GPU: see backward()
        # comm_timing: events around the three collectives of a step (num_pos | bucket A on the side stream |
        # bucket B + logged scalars) and around the final wait for the side stream; comm_report() turns them
        # into milliseconds per step, split into EXPOSED (on the launch stream, nothing to hide behind) and
        # overlapped.  bench.py switches it on for N > 1 so that a scaling run explains its own efficiency.
        self.comm_timing = False
        self._comm_events = []
        self.precision = 'fp32'         # 'fp32' | 'bf16' (activation storage + forward matrix instruction)
        self.lib = L.load()
        self._host_idx = {}
        # one-shot all-reduce over peer-mapped inboxes (oneshot.py / csrc/collective.hip) instead of the process
        # group's collectives: one communicator per stream that carries messages
        self._os_side = self._os_main = None

    # ------------------------------------------------------------------ GT staging
    def stage_gt(self, plan, gt_bboxes, gt_keypointss):
        """Ragged per-image lists -> padded device buffers with O(1) kernel launches."""
        n = plan.n
        pb, pk = getattr(gt_bboxes, 'padded', None), getattr(gt_keypointss, 'padded', None)
        if pb is not None and pk is not None and pb.shape[1] == plan.gmax and pb.is_cuda:
            # fast path: the data source already padded the GT (synthetic.GTList)
            cnt = gt_bboxes.counts
            own_b, own_k, own_c = plan.own_gt
            if (not os.environ.get('YUNET_COPY_GT') and pb.device == own_b.device and cnt.device == own_b.device and
                    pk.device == own_b.device and pb.dtype == torch.float32 and pk.dtype == torch.float32 and
                    cnt.dtype == torch.int32 and pb.is_contiguous() and pk.is_contiguous() and cnt.is_contiguous() and
                    tuple(pb.shape) == tuple(own_b.shape) and pk.numel() == own_k.numel() and cnt.numel() == n):
                plan.bind_gt(pb, pk, cnt)          # read in place (no copy kernels in front of the step)
                return
            plan.bind_gt(*plan.own_gt)
            plan.gt_boxes.copy_(pb, non_blocking=True)
            plan.gt_kps.copy_(pk.reshape(own_k.shape), non_blocking=True)
            plan.gt_count.copy_(cnt, non_blocking=True)
            return
        plan.bind_gt(*plan.own_gt)
        cnt_t = getattr(gt_bboxes, 'counts', None)
        if cnt_t is not None:
            # a GTList whose padded companion does not fit this plan (other Gmax / host-resident):
            # its counts are authoritative -- list items may be padded views with zero rows
            counts = [int(c) for c in cnt_t.tolist()]
            gt_bboxes = [b[:c] for b, c in zip(gt_bboxes, counts)]
            gt_keypointss = [k[:c] for k, c in zip(gt_keypointss, counts)]
        else:
            counts = [int(b.shape[0]) for b in gt_bboxes]
        tot = sum(counts)
        plan.gt_count.copy_(torch.tensor(counts, dtype=torch.int32), non_blocking=True)
        if tot == 0:
            return
        img_idx = torch.repeat_interleave(torch.arange(n), torch.tensor(counts))
        slot = torch.cat([torch.arange(c) for c in counts]) if tot else torch.zeros(0, dtype=torch.int64)
        lin = (img_idx * plan.gmax + slot).to(self.device, non_blocking=True)
        boxes = torch.cat([b.reshape(-1, 4) for b in gt_bboxes]).to(self.device, torch.float32)
        kps = torch.cat([k.reshape(-1, 15) for k in gt_keypointss]).to(self.device, torch.float32)
        plan.gt_boxes.view(-1, 4).index_copy_(0, lin, boxes)
        plan.gt_kps.view(-1, 15).index_copy_(0, lin, kps)

    def set_precision(self, precision):
        if precision not in ('fp32', 'bf16'):
            raise ValueError(f"precision {precision!r}: 'fp32' or 'bf16'")
        self.precision = precision

    def get_plan(self, n, h, w, max_gt):
        gmax = 64
        while gmax < max_gt:
            gmax *= 2
        key = (n, h, w, gmax, self.precision)
        plan = self.plans.get(key)
        if plan is None:
            if h % 32 or w % 32:
                raise ValueError('input height/width must be multiples of 32 (reference: '
                                 'max_pool2d(2) x4 + nearest x2 upsampling must line up)')
            plan = self.plans[key] = Plan(self, n, h, w, gmax)
            # a plan owns every activation / gradient buffer of its shape (hundreds of MB for one 1024 x 1408
            # image).  Training uses one or two shapes; testing at the original image sizes (tools/test_widerface.py
            # --mode 2) walks through hundreds: keep the most recently used ones, drop the rest (a dropped plan
            # stays alive while an autograd graph or `eng.plan` still refers to it)
            while len(self.plans) > MAX_PLANS:
                # a dropped plan returns its buffers to the caching allocator; work issued on the comm side stream
                # (or on executor lanes) does not hold them alive: drain the device first (rare: > 16 shapes walked)
                if self.device.type == 'cuda' and torch.cuda.is_available():
                    torch.cuda.synchronize(self.device)
                self.plans.popitem(last=False)
        else:
            self.plans.move_to_end(key)
        return plan

    def _exec(self, arr, what):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(self.lib.yunet_exec(arr, len(arr), stream), what)

    # ------------------------------------------------------------------ step phases
    def forward(self, img, gt_bboxes, gt_keypointss):
        """img [N,3,H,W] fp32 CUDA; GT ragged lists.  Returns the device tensor
        losses[4] = (loss_cls, loss_bbox, loss_obj, loss_kps)."""
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
        n, _, h, w = img.shape
        pb = getattr(gt_bboxes, 'padded', None)
        cnt_t = getattr(gt_bboxes, 'counts', None)
        if pb is not None:
            max_gt = int(pb.shape[1])
        elif cnt_t is not None:
            max_gt = max(int(cnt_t.max()), 1)
        else:
            max_gt = max([int(b.shape[0]) for b in gt_bboxes] + [1])
        self._check_oneshot()
        plan = self.get_plan(n, h, w, max_gt)
        self.plan = plan
        self._img = img
        self.stage_gt(plan, gt_bboxes, gt_keypointss)
        plan.set_img(img)
        self._exec(plan.c_fwd_a, 'yunet_exec(fwd_a)')
        was_deferred = plan.deferred
        plan.deferred = self.world_size > 1 or (self.always_bucket and torch.distributed.is_initialized())
        if was_deferred and not plan.deferred:
            plan.dy_scale.fill_(1.0)     # (tests flip the mode on one plan) no stale 1 / num_total in the head scales
        if plan.deferred:
            # reduce_mean(num_pos) (yunet_head.py:493-497) on the side stream beside the loss kernel; the launch
            # stream waits for it only in front of loss_finalize
            main, side = torch.cuda.current_stream(), self._comm_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.reduce_num_pos(plan.norm, side)
            self._exec(plan.c_fwd_b_loss, 'yunet_exec(fwd_b: loss)')
            main.wait_stream(side)
            self._exec(plan.c_fwd_b_rest, 'yunet_exec(fwd_b: finalize)')
        else:
            self._exec(plan.c_fwd_b, 'yunet_exec(fwd_b)')
        self.params.num_batches_tracked += 1
        return plan.losses

    def scale_buffer(self):
        """Where the upstream scales of the four losses go (16 head channels: cls | box x4 | obj | kps x10): the
        vector the fused head units read as dy_scale -- or, with the deferred num_pos normaliser (N > 1), its
        un-normalised half, which backward() multiplies by loss_finalize's per-channel 1 / num_total."""
        plan = self.plan
        return plan.dy_up if plan.deferred else plan.dy_scale

    def backward(self, grad_scales=None):
        """d(sum_i s_i * loss_i)/d(params) -> params.grad (overwritten)."""
        plan = self.plan
        if grad_scales is not None:
            s = [float(v) for v in grad_scales]
            vec = [s[0]] + [s[1]] * 4 + [s[2]] + [s[3]] * 10
            self.scale_buffer().copy_(torch.tensor(vec, dtype=torch.float32), non_blocking=True)
        if plan.deferred:
            # x the deferred 1 / num_total of the cls | box | obj channels (written by loss_finalize)
            torch.mul(plan.dy_up, plan.dy_norm, out=plan.dy_scale)
        if plan.towers:
            torch.mul(plan.dy_scale, plan.mask_cls, out=plan.dy_scale_cls)
            torch.sub(plan.dy_scale, plan.dy_scale_cls, out=plan.dy_scale_reg)
        if (self.world_size <= 1 and not self.always_bucket) and plan.split_off is not None and self.overlap_reduce:
            # one GPU: the weight-gradient reduction + BN parameter gradients of segment A (head, neck, late backbone
            # stages: ~85 % of the parameters, 44 us as one launch at the end of backward) on the side stream, under the
            # backward kernels of the early, high-resolution stages
            main, side = torch.cuda.current_stream(), self._comm_stream()
            self._exec(plan.c_bwd_a_k, 'yunet_exec(bwd_a kernels)')
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._exec(plan.c_tail_a, 'yunet_exec(bwd_a reduction)')
            self._exec(plan.c_bwd_b, 'yunet_exec(bwd_b)')
            main.wait_stream(side)
            return
        if (self.world_size <= 1 and not self.always_bucket) or plan.split_off is None:
            self._exec(plan.c_bwd, 'yunet_exec(bwd)')
            self.allreduce_grads()
            return
        # two segments: bucket A (tail of the flat buffer) is all-reduced on the side stream
        # while the kernels of segment B run; bucket B carries the logged scalars in its head
        main = torch.cuda.current_stream()
        side = self._comm_stream()
        gb, cut = self.params.grad_buf, LOG_HEAD + plan.split_off
        self._exec(plan.c_bwd_a, 'yunet_exec(bwd_a)')
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ea = self._comm_mark(side)
            self._allreduce_mean(gb[cut:], self._os_side)
            self._comm_mark(side, ('bucket_a', ea))
        self._exec(plan.c_bwd_b, 'yunet_exec(bwd_b)')
        eb = self._comm_mark(main)
        self._allreduce_mean(gb[:cut], self._os_main)
        ew = self._comm_mark(main, ('bucket_b', eb))
        main.wait_stream(side)
        self._comm_mark(main, ('wait_a', ew))

    def _comm_mark(self, stream, close=None):
        """comm_timing: record an event on `stream`; with close = (name, start event) file the pair."""
        if not self.comm_timing:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        if close is not None and close[1] is not None:
            self._comm_events.append((close[0], close[1], ev))
        return ev

    def comm_report(self, steps):
        """Milliseconds per step of every timed collective since the last call (synchronises).  exposed =
        num_pos + bucket B + the wait for bucket A at the end of backward: the time the launch stream spends
        in or behind communication; bucket A itself runs under the backward kernels of the early stages."""
        torch.cuda.synchronize(self.device)
        self._check_oneshot()
        tot = {}
        for name, e0, e1 in self._comm_events:
            tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
        self._comm_events = []
        per = {k: round(v / max(steps, 1), 4) for k, v in tot.items()}
        per['exposed_ms_per_step'] = round(sum(per.get(k, 0.0) for k in ('num_pos', 'bucket_b', 'wait_a')), 4)
        per['overlapped_ms_per_step'] = per.get('bucket_a', 0.0)
        return per

    def _comm_stream(self):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def enable_oneshot(self, verify=True):
        """Route the step's three messages (num_pos | gradient bucket A on the side stream | bucket B + logged
        scalars) through the one-shot all-reduce: every rank stores its message into every peer's inbox over xGMI
        and sums the slots in rank order (csrc/collective.hip).  With `verify`, each communicator first reproduces
        an all-gather of the process group bit for bit; on a mismatch nothing changes and False is returned."""
        from .oneshot import OneShotAllReduce
        nbytes = self.params.grad_buf.numel() * 4
        comms = []
        try:
            for _ in range(2):
                comms.append(OneShotAllReduce(self.device, nbytes, self.process_group))
        except (RuntimeError, ValueError) as e:
            # raised on EVERY rank together (oneshot.py: _agree; world > 8): all stay on the process group
            warnings.warn(f'one-shot all-reduce unavailable, staying on the process group: {e}')
            for c in comms:
                c.close()
            return False
        if verify and not all(c.verify() for c in comms):
            for c in comms:
                c.close()
            return False
        self._os_side, self._os_main = comms
        return True

    def disable_oneshot(self):
        for c in (self._os_side, self._os_main):
            if c is not None:
                c.close()
        self._os_side = self._os_main = None

    def oneshot_status(self):
        """0, or the sequence number of the first message whose wait for a peer timed out (host read)."""
        return max((c.status() for c in (self._os_side, self._os_main) if c is not None), default=0)

    def _check_oneshot(self):
        """Called at the start of every step (and by comm_report): a host read of the communicators' status words,
        no synchronisation.  A peer that did not arrive within `oneshot_timeout_ms` left this rank's buffer poisoned
        with NaN (csrc/collective.hip); the ranks' parameters would diverge silently if training went on, and the
        process group cannot repair a step that was already applied -- so this raises."""
        if self._os_side is None and self._os_main is None:
            return
        st = self.oneshot_status()
        if st:
            raise RuntimeError(f'one-shot all-reduce: message {st} timed out waiting for a peer (option '
                               f'oneshot_timeout_ms); the reduced buffer was poisoned with NaN.  A rank stalled or '
                               f'died -- restart from the last checkpoint (YUNET_ONESHOT_AR=0 uses the process group)')

    def _allreduce_mean(self, t, comm=None):
        """Mean over ranks, in place, on the current stream.  RCCL averages inside the collective
        (ncclAvg); other backends (gloo: CPU tests, several ranks sharing one GPU) sum, then scale.
        `comm`: the one-shot communicator of the current stream, when enabled."""
        if comm is not None:
            comm.all_reduce_(t, mean=True)
            return
        pg = self.process_group
        if torch.distributed.get_backend(pg) == 'nccl':
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.AVG, group=pg)
        else:
            torch.distributed.all_reduce(t, group=pg)
            t.div_(self.world_size)

    def reduce_num_pos(self, norm, stream=None):
        """reduce_mean(num_pos) (mmdet/core/utils/dist_utils.py:68-74, yunet_head.py:493-497):
        every rank holds num_pos/world in norm[0]; SUM over ranks.  4 bytes, latency-bound; issued on the side
        stream, beside the loss kernel (forward())."""
        if self.world_size > 1 or (self.always_bucket and torch.distributed.is_initialized()):
            e0 = self._comm_mark(stream)
            if self._os_side is not None:
                self._os_side.all_reduce_(norm[0:1], stream=stream)
            else:
                torch.distributed.all_reduce(norm[0:1], group=self.process_group)
            self._comm_mark(stream, ('num_pos', e0))

    def allreduce_grads(self):
        """DDP gradient mean in ONE collective over [logged scalars | flat gradient]
        (303 KB for yunet_n, 218 KB for yunet_s) -- the unsplit form of what backward() issues
        in two buckets."""
        if self.world_size <= 1:
            return
        self._allreduce_mean(self.params.grad_buf, self._os_main)

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def forward_eval(self, img):
        """eval(): conv stack with BatchNorm on the running statistics -> flat [N,P,16]
        (cls | dx dy dw dh | obj | 10 kps raw head outputs).  No statistics are updated."""
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
        n, _, h, w = img.shape
        plan = self.get_plan(n, h, w, 1)
        self.plan = plan
        self._img = img
        plan.set_img(img)
        self._exec(plan.c_fwd_eval, 'yunet_exec(fwd_eval)')
        return plan.flat

    @torch.no_grad()
    def forward_features(self, img):
        """Train-mode conv stack only -> flat [N,P,16] (used by tests)."""
        n, _, h, w = img.shape
        plan = self.get_plan(n, h, w, 1)
        self.plan = plan
        self._img = img          # the backward op list reads the image again (stem wgrad)
        plan.set_img(img)
        plan.bind_gt(*plan.own_gt)       # (never zero a data source's tensor)
        plan.gt_count.zero_()
        self._exec(plan.c_fwd_a, 'yunet_exec(fwd_a)')
        return plan.flat
