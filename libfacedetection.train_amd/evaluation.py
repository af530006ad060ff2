"""WIDER-Face average precision (easy / medium / hard) of a prediction set.

Reference: mmdet/core/evaluation/widerface.py:152-347 (`norm_score`, `image_eval`,
`img_pr_info`, `dataset_pr_info`, `voc_ap`, `wider_evaluation`), called by
tools/test_widerface.py:177 as ``wider_evaluation(results, gt_path, 0.5)``.

Same inputs and the same float64 arithmetic, organised per image as array operations instead
of the reference's per-prediction Python loop over a multiprocessing pool:

* the IoU of every (prediction, GT) pair is one matrix (the "+1" pixel convention of the
  protocol: w = x2 - x1 + 1);
* the greedy matching is sequential only through "has this kept GT been hit before", which
  is a first-occurrence mask + cumulative sum;
* the 1000-threshold precision / recall counters of an image come from one boolean
  [threshold, prediction] matrix.

`pred` is {event: {image name: float array [n, 5] = x, y, w, h, score}} (rows in descending
score, as get_bboxes delivers them); `gt_path` holds the four protocol files
wider_{face,easy,medium,hard}_val.mat.  Host side only (numpy): evaluation is not part of the
training hot path.
"""
import os

import numpy as np

THRESH_NUM = 1000


# ------------------------------------------------------------------------------- ground truth
def load_wider_gt(gt_dir):
    """-> list of events: dict(name, images=[dict(name, boxes [g,4] xywh float64,
    keep={'easy'|'medium'|'hard': 1-based index array})])."""
    from scipy.io import loadmat
    face = loadmat(os.path.join(gt_dir, 'wider_face_val.mat'))
    subsets = {k: loadmat(os.path.join(gt_dir, f'wider_{k}_val.mat'))['gt_list']
               for k in ('easy', 'medium', 'hard')}
    events = []
    for i in range(len(face['event_list'])):
        ev = dict(name=str(face['event_list'][i][0][0]), images=[])
        files, boxes = face['file_list'][i][0], face['face_bbx_list'][i][0]
        for j in range(len(files)):
            keep = {k: np.asarray(subsets[k][i][0][j][0]).reshape(-1).astype(np.int64) for k in subsets}
            ev['images'].append(dict(name=str(files[j][0][0]),
                                     boxes=np.asarray(boxes[j][0], dtype=np.float64).reshape(-1, 4),
                                     keep=keep))
        events.append(ev)
    return events


# --------------------------------------------------------------------------------- predictions
def read_predictions(pred_dir):
    """The per-image text files tools/test_widerface.py --save-preds writes
    (<event>/<image>.txt: name, count, then `x y w h score` rows) -> the `pred` dict."""
    pred = {}
    for event in sorted(os.listdir(pred_dir)):
        edir = os.path.join(pred_dir, event)
        if not os.path.isdir(edir):
            continue
        cur = {}
        for fn in sorted(os.listdir(edir)):
            with open(os.path.join(edir, fn)) as f:
                lines = f.read().splitlines()
            name = lines[0].split('/')[-1]
            rows = [[float(v) for v in ln.split(' ')] for ln in lines[2:] if ln.strip()]
            cur[name[:-4] if name.endswith('.jpg') else name] = \
                np.asarray(rows, dtype=np.float64).reshape(-1, 5)
        pred[event] = cur
    return pred


def write_predictions(pred_dir, event, image_name, boxes_xyxy_score):
    """One image in the protocol's text format (tools/test_widerface.py:150-165)."""
    os.makedirs(os.path.join(pred_dir, event), exist_ok=True)
    b = np.asarray(boxes_xyxy_score, dtype=np.float64).reshape(-1, 5)
    with open(os.path.join(pred_dir, event, image_name + '.txt'), 'w') as f:
        f.write(f'{event}/{image_name}.jpg\n{b.shape[0]}\n')
        for r in b:
            f.write('%.5f %.5f %.5f %.5f %g\n' % (r[0], r[1], r[2] - r[0], r[3] - r[1], r[4]))


def norm_score(pred):
    """Min-max normalisation of every score over the WHOLE prediction set, in place
    (widerface.py:152-174)."""
    lo, hi = 2.0, -1.0
    for ev in pred.values():
        for v in ev.values():
            if len(v):
                lo, hi = min(lo, float(np.min(v[:, -1]))), max(hi, float(np.max(v[:, -1])))
    diff = hi - lo
    for ev in pred.values():
        for v in ev.values():
            if len(v):
                v[:, -1] = (v[:, -1] - lo).astype(np.float64) / diff
    return pred


# --------------------------------------------------------------------------------- one image
def pairwise_iou_xywh(pred, gt):
    """[n_pred, n_gt] IoU with the protocol's inclusive-pixel convention (widerface.py:39-52)."""
    p = np.asarray(pred, dtype=np.float64)
    g = np.asarray(gt, dtype=np.float64)
    px2, py2 = p[:, 0] + p[:, 2], p[:, 1] + p[:, 3]
    gx2, gy2 = g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    w = np.minimum(gx2[None, :], px2[:, None]) - np.maximum(g[None, :, 0], p[:, None, 0]) + 1
    h = np.minimum(gy2[None, :], py2[:, None]) - np.maximum(g[None, :, 1], p[:, None, 1]) + 1
    inter = w * h
    ga = (gx2 - g[:, 0] + 1) * (gy2 - g[:, 1] + 1)
    pa = (px2 - p[:, 0] + 1) * (py2 - p[:, 1] + 1)
    o = inter / (ga[None, :] + pa[:, None] - inter)
    o[(w <= 0) | (h <= 0)] = 0
    return o


def image_eval(pred, gt, keep_flag, iou_thresh):
    """Greedy matching of one image (widerface.py:177-215).

    keep_flag[g] = 1 for the GTs of the current difficulty subset.  Returns
    pred_recall[h] = subset GTs recalled by predictions 0..h, and proposal[h] = -1 for
    predictions whose best GT lies outside the subset (not counted as false positives), else 1."""
    n = pred.shape[0]
    iou = pairwise_iou_xywh(pred[:, :4], gt)
    best = iou.argmax(axis=1)
    hit = iou[np.arange(n), best] >= iou_thresh
    in_subset = keep_flag[best] == 1
    proposal = np.where(hit & ~in_subset, -1.0, 1.0)
    counted = hit & in_subset
    first = np.zeros(n, dtype=bool)
    if counted.any():
        idx = np.nonzero(counted)[0]
        _, first_pos = np.unique(best[idx], return_index=True)     # first prediction per GT
        first[idx[first_pos]] = True
    return np.cumsum(first).astype(np.float64), proposal


def img_pr_info(scores, proposal, pred_recall, thresh_num=THRESH_NUM):
    """[thresh_num, 2] = (#counted predictions, #recalled GTs) at score thresholds
    1 - (t+1)/thresh_num (widerface.py:218-239): taken at the LAST prediction whose score
    passes the threshold."""
    thresh = np.array([1 - (t + 1) / thresh_num for t in range(thresh_num)])
    ok = scores[None, :] >= thresh[:, None]
    any_ok = ok.any(axis=1)
    last = scores.shape[0] - 1 - np.argmax(ok[:, ::-1], axis=1)
    cum_prop = np.cumsum(proposal == 1).astype(np.float64)
    out = np.zeros((thresh_num, 2))
    out[any_ok, 0] = cum_prop[last[any_ok]]
    out[any_ok, 1] = pred_recall[last[any_ok]]
    return out


def voc_ap(rec, prec):
    """Area under the precision envelope (widerface.py:250-268)."""
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


# ----------------------------------------------------------------------------------- dataset
def wider_evaluation(pred, gt_path, iou_thresh=0.5, return_curves=False):
    """-> [AP_easy, AP_medium, AP_hard] (widerface.py:271-347).  `pred` scores are normalised in
    place, as in the reference."""
    pred = norm_score(pred)
    events = load_wider_gt(gt_path) if isinstance(gt_path, (str, os.PathLike)) else gt_path
    aps, curves = [], []
    for setting in ('easy', 'medium', 'hard'):
        count_face = 0
        pr = np.zeros((THRESH_NUM, 2))
        for ev in events:
            plist = pred[ev['name']]
            for im in ev['images']:
                info = plist[im['name']]
                keep = im['keep'][setting]
                count_face += len(keep)
                if len(im['boxes']) == 0 or len(info) == 0:
                    continue
                flag = np.zeros(im['boxes'].shape[0], dtype=np.int64)
                if len(keep):
                    flag[keep - 1] = 1
                info = np.asarray(info, dtype=np.float64)
                rec, prop = image_eval(info, im['boxes'], flag, iou_thresh)
                pr += img_pr_info(info[:, 4], prop, rec)
        with np.errstate(divide='ignore', invalid='ignore'):
            precision = pr[:, 1] / pr[:, 0]
            recall = pr[:, 1] / count_face
        aps.append(float(voc_ap(recall, precision)))
        curves.append(np.stack([precision, recall], 1))
    return (aps, curves) if return_curves else aps


# ============================================================================ mAP (EvalHook during training)
# What `EvalHook` reports while training (mmdet/apis/train.py:226-232 -> mmdet/core/evaluation/eval_hooks.py:24-66
# -> CustomDataset.evaluate, mmdet/datasets/custom.py:310-367 -> eval_map, mmdet/core/evaluation/mean_ap.py:522-686):
# VOC-style average precision of the single face class at IoU 0.5, "area" mode, ignored GT boxes neither matched
# nor counted.  Restated for one class, without the multiprocessing pool; same arithmetic and dtypes.
def bbox_overlaps_np(b1, b2, eps=1e-6):
    """mmdet/core/evaluation/bbox_overlaps.py:5-65 (mode 'iou', no legacy +1): float32 [n, k]."""
    b1, b2 = np.asarray(b1, dtype=np.float32).reshape(-1, 4), np.asarray(b2, dtype=np.float32).reshape(-1, 4)
    if b1.shape[0] * b2.shape[0] == 0:
        return np.zeros((b1.shape[0], b2.shape[0]), dtype=np.float32)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    xs, ys = np.maximum(b1[:, None, 0], b2[None, :, 0]), np.maximum(b1[:, None, 1], b2[None, :, 1])
    xe, ye = np.minimum(b1[:, None, 2], b2[None, :, 2]), np.minimum(b1[:, None, 3], b2[None, :, 3])
    overlap = np.maximum(xe - xs, 0) * np.maximum(ye - ys, 0)
    union = np.maximum(a1[:, None] + a2[None, :] - overlap, np.float32(eps))
    return (overlap / union).astype(np.float32)


def tpfp_default(dets, gts, gts_ignore, iou_thr=0.5):
    """mean_ap.py:168-267 without area ranges: (tp, fp) float32 [m] for the detections [m, 5] of one image.
    A detection whose best-overlapping GT is an ignored box is neither; a second hit on a covered GT is fp."""
    m = dets.shape[0]
    tp, fp = np.zeros(m, dtype=np.float32), np.zeros(m, dtype=np.float32)
    ignore = np.concatenate([np.zeros(gts.shape[0], dtype=bool), np.ones(gts_ignore.shape[0], dtype=bool)])
    allgt = np.vstack([gts.reshape(-1, 4), gts_ignore.reshape(-1, 4)])
    if allgt.shape[0] == 0:
        fp[...] = 1
        return tp, fp
    ious = bbox_overlaps_np(dets[:, :4], allgt)
    best, arg = ious.max(axis=1), ious.argmax(axis=1)
    covered = np.zeros(allgt.shape[0], dtype=bool)
    for i in np.argsort(-dets[:, -1]):
        if best[i] >= iou_thr:
            g = arg[i]
            if not ignore[g]:
                if not covered[g]:
                    covered[g] = True
                    tp[i] = 1
                else:
                    fp[i] = 1
        else:
            fp[i] = 1
    return tp, fp


def average_precision_area(recalls, precisions):
    """mean_ap.py:13-57, mode 'area' (float32 accumulator like the reference)."""
    mrec = np.hstack((np.zeros(1, recalls.dtype), recalls, np.ones(1, recalls.dtype)))
    mpre = np.hstack((np.zeros(1, recalls.dtype), precisions, np.zeros(1, recalls.dtype)))
    for i in range(mpre.shape[0] - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    ind = np.where(mrec[1:] != mrec[:-1])[0]
    return np.float32(np.sum((mrec[ind + 1] - mrec[ind]) * mpre[ind + 1]))


def eval_map_single_class(det_results, annotations, iou_thr=0.5):
    """eval_map (mean_ap.py:522-686) for one class.  det_results: per image [[n, 5] array] (the per-class
    list of a detector's simple_test) or the [n, 5] array itself; annotations: per image dict(bboxes, labels,
    bboxes_ignore, labels_ignore) as RetinaFaceDataset.get_ann_info returns.  -> (mAP, dict(num_gts, num_dets,
    recall, precision, ap))."""
    assert len(det_results) == len(annotations)
    dets = [np.asarray(d[0] if isinstance(d, (list, tuple)) else d, dtype=np.float32).reshape(-1, 5) for d in det_results]
    tps, fps, num_gts = [], [], 0
    for d, ann in zip(dets, annotations):
        gts = np.asarray(ann['bboxes'], dtype=np.float32).reshape(-1, 4)
        ign = np.asarray(ann.get('bboxes_ignore', np.zeros((0, 4))), dtype=np.float32).reshape(-1, 4)
        t, f = tpfp_default(d, gts, ign, iou_thr)
        tps.append(t)
        fps.append(f)
        num_gts += gts.shape[0]
    alld = np.vstack(dets) if dets else np.zeros((0, 5), dtype=np.float32)
    order = np.argsort(-alld[:, -1])
    tp = np.cumsum(np.hstack(tps)[order]) if alld.shape[0] else np.zeros(0, dtype=np.float32)
    fp = np.cumsum(np.hstack(fps)[order]) if alld.shape[0] else np.zeros(0, dtype=np.float32)
    eps = np.finfo(np.float32).eps
    recalls = tp / np.maximum(np.array([num_gts]), eps)      # float64, like the reference's int array / float32 eps
    precisions = tp / np.maximum(tp + fp, eps)
    ap = average_precision_area(recalls, precisions)
    res = dict(num_gts=num_gts, num_dets=int(alld.shape[0]), recall=recalls, precision=precisions, ap=ap)
    return (float(ap) if num_gts > 0 else 0.0), res


def prepare_test_image(img_bgr, scale, device, resize='cv2'):
    """The test pipeline of the shipped configs on the device (configs/yunet_n.py:57-86: MultiScaleFlipAug(
    img_scale, flip=False) -> Resize(keep_ratio=True) -> Normalize(mean 0, std 1) -> Pad(size_divisor 32)):
    uint8 [h, w, 3] -> (float32 [1, 3, H, W] on the device, img_meta).  mmcv.imrescale semantics for the size
    (factor = min(long / long_edge, short / short_edge), rounded); scale None keeps the original size.
    The image is resized while it is still uint8, as the reference does, in cv2.resize's fixed-point arithmetic
    (imresize.resize_linear_u8); resize='float' is the fp32 bilinear this function used before (A/B only)."""
    import torch
    import torch.nn.functional as F
    from . import imresize
    h, w = img_bgr.shape[:2]
    x8 = torch.from_numpy(np.ascontiguousarray(img_bgr)).to(device)
    if scale is None:
        nh, nw = h, w
        x = x8.permute(2, 0, 1)[None].float()
    else:
        nw, nh = imresize.rescale_size(w, h, scale)
        if resize == 'cv2' and x8.dtype == torch.uint8:
            x = imresize.resize_linear_u8(x8, (nw, nh)).permute(2, 0, 1)[None].float()
        else:
            x = F.interpolate(x8.permute(2, 0, 1)[None].float(), size=(nh, nw), mode='bilinear', align_corners=False)
    ph = max(nh, 0 if scale is None else scale[0] if nh <= scale[0] else nh)
    pw = max(nw, 0 if scale is None else scale[1] if nw <= scale[1] else nw)
    ph, pw = (ph + 31) // 32 * 32, (pw + 31) // 32 * 32
    x = F.pad(x, (0, pw - nw, 0, ph - nh)).contiguous()
    sf = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
    meta = dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3), scale_factor=sf,
                flip=False, flip_direction='horizontal')
    return x, meta


def single_gpu_test(model, dataset, device, scale=(640, 640), max_images=None):
    """mmdet/apis/test.py single_gpu_test for this path: eval-mode forward + get_bboxes(rescale=True) per image
    of a test-mode RetinaFaceDataset -> [[dets [n, 5]]] per image (boxes in original-image coordinates)."""
    import torch
    was_training = model.training
    model.eval()
    out = []
    n = len(dataset) if max_images is None else min(len(dataset), max_images)
    with torch.no_grad():
        for i in range(n):
            img, meta = prepare_test_image(dataset.load_image(i), scale, device)
            meta['ori_filename'] = dataset.data_infos[i]['filename']
            out.append(model(return_loss=False, rescale=True, img=[img], img_metas=[[meta]])[0])
    if was_training:
        model.train()
    return out


def multi_gpu_test(model, dataset, device, scale=(640, 640), max_images=None, group=None):
    """mmdet/apis/test.py multi_gpu_test for this path (what the reference's DistEvalHook runs): rank r takes the
    images r, r + world, r + 2 world, ...; the per-image results are gathered and put back in dataset order
    (collect_results).  Returns the full list on rank 0 and None on the other ranks."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    was_training = model.training
    model.eval()
    n = len(dataset) if max_images is None else min(len(dataset), max_images)
    part = []
    with torch.no_grad():
        for i in range(rank, n, world):
            img, meta = prepare_test_image(dataset.load_image(i), scale, device)
            meta['ori_filename'] = dataset.data_infos[i]['filename']
            part.append(model(return_loss=False, rescale=True, img=[img], img_metas=[[meta]])[0])
    if was_training:
        model.train()
    parts = [None] * world
    dist.all_gather_object(parts, part, group=group)
    if rank != 0:
        return None
    out = [None] * n
    for r, p in enumerate(parts):
        for k, res in enumerate(p):
            out[r + k * world] = res
    return out

