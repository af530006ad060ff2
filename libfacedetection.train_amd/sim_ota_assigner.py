"""SimOTAAssigner with the reference's call signature
(mmdet/core/bbox/assigners/sim_ota_assigner.py:13-257), running the HIP assign kernel.

`YuNet_Head.loss` does not call this per image (the fused kernel handles the batch); the
class exists so configs resolve and so the stand-alone API stays usable and testable."""
import torch

from . import kernels as K
from .builder import BBOX_ASSIGNERS, BBOX_SAMPLERS


class AssignResult:
    """mmdet/core/bbox/assigners/assign_result.py:43-49."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = \
            num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


@BBOX_ASSIGNERS.register_module()
class SimOTAAssigner:
    def __init__(self, center_radius=2.5, candidate_topk=10, iou_weight=3.0, cls_weight=1.0):
        if not 1 <= int(candidate_topk) <= 16:
            raise NotImplementedError('candidate_topk must be in 1..16 (per-lane candidate lists of the HIP kernel)')
        self.center_radius = center_radius
        self.candidate_topk, self.iou_weight, self.cls_weight = candidate_topk, iou_weight, cls_weight

    def assign(self, pred_scores, priors, decoded_bboxes, gt_bboxes, gt_labels,
               gt_bboxes_ignore=None, eps=1e-7):
        """pred_scores [P,1]; priors [P,4] = (cx, cy, stride, stride) with the +0.5*stride
        offset already applied (yunet_head.py:572-573); decoded_bboxes [P,4]."""
        if pred_scores.shape[-1] != 1:
            raise NotImplementedError('single-class (face) assignment only')
        dev = decoded_bboxes.device
        P = decoded_bboxes.shape[0]
        # recover the level structure from the priors' strides (levels are concatenated)
        strides_t = priors[:, 2]
        uniq, counts = torch.unique_consecutive(strides_t, return_counts=True)
        strides = [int(s) for s in uniq.tolist()]
        sizes = []
        off = 0
        for s, c in zip(strides, counts.tolist()):
            xs = priors[off:off + c, 0]
            w = int(round(float((xs.max() - xs.min()) / s))) + 1
            sizes.append((c // w, w))
            off += c
        G = int(gt_bboxes.shape[0])
        gmax = max(G, 1)
        gb = torch.zeros(1, gmax, 4, device=dev)
        gb[0, :G] = gt_bboxes.float()
        gk = torch.zeros(1, gmax, 5, 3, device=dev)
        gl = torch.zeros(1, gmax, dtype=torch.int32, device=dev)
        gl[0, :G] = gt_labels.int()
        cnt = torch.tensor([G], dtype=torch.int32, device=dev)
        gi, ovl, _, labels = K.assign(
            None, gb, gk, cnt, sizes, strides, self.center_radius, gt_labels=gl, want_labels=True,
            pre_scores=pred_scores.reshape(1, P).float().contiguous(),
            pre_boxes=decoded_bboxes.reshape(1, P, 4).float().contiguous(),
            candidate_topk=self.candidate_topk, iou_weight=self.iou_weight, cls_weight=self.cls_weight)
        if G == 0:
            ovl = torch.zeros_like(ovl)      # sim_ota_assigner.py:136-149
        return AssignResult(G, gi[0].long(), ovl[0], labels=labels[0].long())


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    """mmdet/core/bbox/samplers/pseudo_sampler.py:24-42 (positives in ascending order)."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1)
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1)
        res = type('SamplingResult', (), {})()
        res.pos_inds, res.neg_inds = pos, neg
        res.pos_assigned_gt_inds = assign_result.gt_inds[pos] - 1
        res.pos_gt_bboxes = gt_bboxes.view(-1, 4)[res.pos_assigned_gt_inds]
        res.pos_gt_labels = assign_result.labels[pos] if assign_result.labels is not None else None
        return res
