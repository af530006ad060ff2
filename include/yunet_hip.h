/*
 * yunet_hip.h -- C ABI of libyunet_hip.so, the MI355X (gfx950) kernels of the YuNet
 * training hot path.  Plain pointers and sizes only: every pointer is a DEVICE pointer
 * unless stated otherwise, `stream` is a hipStream_t passed as void*, every function
 * returns 0 on success or a negative YUNET_E* / the hipError_t of the failed launch.
 * No ownership is transferred; nothing is allocated; nothing synchronises (the inbox set-up of the
 * one-shot all-reduce at the end of this file is the one exception, and says so).
 *
 * The reference (ShiqiYu/libfacedetection.train) has no FFI: its drop-in boundary is
 * the mmcv Registry.  The registered Python classes of
 * libfacedetection.train_amd/ bind these entry points with ctypes; each entry point
 * names the reference code it replaces (paths relative to the reference root).
 *
 * Layouts: activations are NHWC fp32; parameters keep the reference's OIHW fp32
 * shapes (pointwise [Co,Ci,1,1], depthwise [C,1,3,3], stem [16,3,3,3]); images enter
 * as the reference delivers them, NCHW fp32.
 */
#ifndef YUNET_HIP_H
#define YUNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YUNET_ABI_VERSION 11

#define YUNET_EINVAL (-1)   /* bad argument / unsupported channel count */
#define YUNET_EOPCODE (-2)  /* unknown opcode in an op list            */

/* Train-mode BatchNorm description shared by producers and consumers.
 * The producer of a tensor accumulates `stats` = {sum[C], sumsq[C]} (fp64) over all
 * N*H*W positions; consumers derive scale = gamma*invstd, shift = beta - mean*scale in
 * their prologue (nn.BatchNorm2d defaults, mmdet/models/utils/yunet_layer.py:26,60).
 * In backward, consumers accumulate `bstats` = {sum dy[C], sum dy*xhat[C]} (fp64), i.e.
 * d(beta) and d(gamma); the producer's backward turns dy into dz with them. */
typedef struct YunetBN {
    const double* stats;   /* [slots][2*C] forward sums (zeroed before the step)      */
    double* bstats;        /* [slots][2*C] backward sums (zeroed before the step) or NULL */
    const float* gamma;    /* [C]                                                     */
    const float* beta;     /* [C]                                                     */
    int32_t count;         /* N*H*W of the normalised tensor                          */
    float eps;
    int32_t slots;         /* replicas of the sum blocks (0 and 1: one).  Workgroup b adds its partial sums
                            * into replica b % slots and every reader adds the replicas up: the fp64 atomics
                            * that end a kernel spread over `slots` times as many cache lines (at the 10 x 10
                            * and 20 x 20 levels ~250 workgroups x 128 atomics on eight lines cost 4-5 us of a
                            * 14-30 us launch)                  */
    int32_t reserved_;
} YunetBN;

/* Input transform of a fused unit (how it reads its input tensor). */
enum { YUNET_T_IDENTITY = 0, YUNET_T_BNRELU = 1 };

/* Storage type of ACTIVATION tensors (raw conv outputs, pool / upsample-add outputs).  YUNET_BF16 is
 * BASELINE.json configs[2] "bf16 fwd / fp32 grads": activations are kept as bf16 in HBM and the
 * forward pointwise GEMM runs on the bf16 matrix instruction; gradients, the head output
 * [N,P,16], BatchNorm sums, parameters and the optimizer stay fp32.  (The reference's analogue is
 * its fp16 path: mmdet/apis/train.py:181-185, detectors/base.py:168, yunet_head.py:418 force_fp32.) */
enum { YUNET_F32 = 0, YUNET_BF16 = 1 };

/* One ConvDPUnit: 1x1 pointwise (bias) -> 3x3 depthwise (bias, zero-pads the pointwise
 * output) [-> BN -> ReLU applied by the consumer].
 * mmdet/models/utils/yunet_layer.py:4-36.  cin in {16,32,64}, cout in {16,32,64}. */
typedef struct YunetDP {
    int32_t N, H, W, cin, cout;
    int32_t in_transform;      /* YUNET_T_*: applied to x on load                     */
    int32_t out_has_bn;        /* 1: unit is followed by BN+ReLU (stats are produced) */
    int32_t accumulate_dx;     /* bwd: dx += instead of dx =                          */
    int64_t x_img_stride;      /* elements between images of x  (>= H*W*cin)          */
    int64_t z_img_stride;      /* elements between images of z  (>= H*W*cout)         */
    const float* x;            /* [N,H,W,cin] producer's raw output                   */
    YunetBN in_bn;             /* BN of the producer (when in_transform == BNRELU)    */
    const float* w_pw;         /* [cout,cin]                                          */
    const float* b_pw;         /* [cout]                                              */
    const float* w_dw;         /* [cout,9]                                            */
    const float* b_dw;         /* [cout]                                              */
    float* z;                  /* [N,H,W,cout] raw (pre-BN) output                    */
    YunetBN out_bn;            /* this unit's BN (stats written fwd, bstats read bwd) */
    /* backward only */
    const float* dy;           /* [N,H,W,cout] grad wrt BN output (or wrt z if no BN) */
    const float* dy_scale;     /* [cout] per-channel factor on dy (device) or NULL    */
    float* dx;                 /* [N,H,W,cin] grad wrt the BN output of the producer
                                  (ReLU mask applied) or wrt x if IDENTITY; NULL skips */
    float* wgrad_partials;     /* [nblocks, cout*cin + cout + cout*9 + cout] fp32     */
    int32_t wgrad_blocks;      /* number of partial rows (= launch grid)              */
    unsigned long long* prof;  /* optional [grid,8] per-workgroup phase cycle counters (or NULL) */
    int32_t x_dtype;           /* YUNET_F32 | YUNET_BF16: storage of x (and of z unless z_dtype says otherwise) */
    int32_t z_dtype;           /* storage of z: equals x_dtype, except the fused heads, whose z is the fp32 [N,P,16] */
    /* Fused max_pool2d(2) of this unit's BN+ReLU output (both NULL: none).  Where yunet_dp_pool_fusion_ok()
     * says so and the pool is the output's ONLY consumer:
     *   forward : pool_out [N,H/2,W/2,cout] (activation storage type) receives, per 2x2 window and channel, the
     *             RAW z that wins the window after BN+ReLU (the maximum for gamma > 0, the minimum for gamma < 0,
     *             the first element for gamma == 0) and pool_idx [N,H/2,W/2,cout] its window position 2*dy + dx
     *             (ties: the smaller position, as F.max_pool2d).  The consumer reads pool_out with
     *             in_transform = BNRELU and THIS unit's BN (count = N*H*W): relu(bn(.)) is monotone, so that is
     *             max_pool2d(relu(bn(z))) exactly, and its backward yields the BN-backward sums unchanged.
     *   backward: pool_idx != NULL: dy is the POOLED gradient [N,H/2,W/2,cout] the consumer wrote as its dx
     *             (ReLU mask applied); it reaches the recorded window position while the tile is staged -- no
     *             full-size gradient of z exists. */
    float* pool_out;
    uint8_t* pool_idx;
} YunetDP;

/* ---- conv stack (mmdet/models/utils/yunet_layer.py, backbones/yunet_backbone.py:33-41,
 *      necks/tfpn.py:33-45, dense_heads/yunet_head.py:175-247) ------------------- */

/* ABI 10: n <= YUNET_DP_GROUP_MAX mutually INDEPENDENT ConvDPUnit forwards in one launch where the kernels allow it (plain
 * 64 -> 64 units on the wave-streaming kernel: the share convs of the pyramid levels, yunet_head.py:175-247, which the
 * reference walks in a Python loop); otherwise the units are launched one after the other.  Every unit computes exactly
 * what yunet_dp_fwd computes for it (same grid, same band height); only launch boundaries disappear.  Nothing may
 * connect the units: no output of one is an input (or a BatchNorm sum block) of another. */
#define YUNET_DP_GROUP_MAX 3
int yunet_dp_fwd_group(const YunetDP* const* units, int n, void* stream);

/* Conv_head.conv1: 3x3 stride-2 conv 3->cmid (+bias), NCHW image in, NHWC raw out,
 * BN statistics accumulated (yunet_layer.py:51-52,58). cmid must be 16. */
int yunet_stem_fwd(const float* img, const float* w, const float* b, float* z,
                   double* stats, int N, int H, int W, int cmid, void* stream);
/* weight/bias gradient of the stem (no input gradient: the image is a leaf).
 * dy is the grad wrt bn1's output with the ReLU mask applied; partials
 * [blocks, cmid*27 + cmid] are reduced by yunet_reduce_partials. */
int yunet_stem_bwd(const float* img, const float* z, const float* dy, const YunetBN* bn,
                   float* wgrad_partials, int wgrad_blocks, int N, int H, int W, int cmid,
                   void* stream);

/* The same weight gradient with z RECOMPUTED from the image (w [cmid,3,3,3], b [cmid]: the stem's parameters) instead of
 * read -- 112 instead of 176 bytes per output pixel -- as two matrix products on the matrix cores.  fp32 storage only. */
int yunet_stem_bwd_rz(const float* img, const float* w, const float* b, const float* dy, const YunetBN* bn,
                      float* wgrad_partials, int wgrad_blocks, int N, int H, int W, int cmid, void* stream);

int yunet_dp_fwd(const YunetDP* d, void* stream);
int yunet_dp_bwd(const YunetDP* d, void* stream);
/* rows of wgrad_partials (= persistent grid) yunet_dp_bwd / yunet_stem_bwd use for a shape */
int yunet_dp_bwd_blocks(int N, int H, int W, int cin, int cout);
/* 1 if yunet_dp_fwd / yunet_dp_bwd accept YunetDP.pool_out / pool_idx for this shape (the unpacked
 * 16->16 units on maps >= 32x64, the 32->64 units and the unpacked 64->64 units; H, W even) */
int yunet_dp_pool_fusion_ok(int N, int H, int W, int cin, int cout);
int yunet_stem_bwd_blocks(int N, int H, int W);

/* F.max_pool2d(relu(bn(z)), 2)  (yunet_backbone.py:39-40).  out [N,H/2,W/2,C]. */
int yunet_pool_fwd(const float* z, const YunetBN* bn, float* out, int N, int H, int W, int C,
                   void* stream);
/* dy_out [N,H/2,W/2,C] -> dz-side grad wrt bn output (mask applied) [N,H,W,C]; accumulates
 * bn->bstats.  accumulate != 0: dx += . */
int yunet_pool_bwd(const float* z, const YunetBN* bn, const float* dy_out, float* dx,
                   int accumulate, int N, int H, int W, int C, void* stream);
/* The same with a second, full-size gradient of relu(bn(z)) added under the same mask: dx = mask (extra +
 * route(dy_out)), one set of bn->bstats sums.  `extra` [N,H,W,C] fp32 (NULL = yunet_pool_bwd).  Used where a pyramid
 * tap feeds both max_pool2d (yunet_backbone.py:39-40) and the identity branch of the TFPN merge (tfpn.py:39-40):
 * autograd adds the two gradients of the tap; here yunet_upadd_bwd(dxa = NULL) leaves its share to this call (ABI 8). */
int yunet_pool_bwd_add(const float* z, const YunetBN* bn, const float* dy_out, const float* extra, float* dx,
                       int accumulate, int N, int H, int W, int C, void* stream);

/* TFPN merge: out = relu(bn_a(za)) + nearest_up2(relu(bn_b(zb)))  (tfpn.py:39-40).
 * za [N,H,W,C], zb [N,H/2,W/2,C]. */
int yunet_upadd_fwd(const float* za, const YunetBN* bna, const float* zb, const YunetBN* bnb,
                    float* out, int N, int H, int W, int C, void* stream);
/* dxa == NULL: the share of the fine tensor (mask_a dout, and bna->bstats) is NOT produced -- the caller hands `dout`
 * to yunet_pool_bwd_add as `extra`; za is then not read. */
int yunet_upadd_bwd(const float* za, const YunetBN* bna, const float* zb, const YunetBN* bnb,
                    const float* dout, float* dxa, int accumulate_a, float* dxb,
                    int accumulate_b, int N, int H, int W, int C, void* stream);

/* BatchNorm running statistics: running = (1-m)*running + m*batch (unbiased var)
 * for `count` BN layers described by parallel arrays (host pointers to device ptrs). */
int yunet_bn_update_running(const double* stats, float* running_mean, float* running_var,
                            int C, int count, float momentum, void* stream);

/* Final BN parameter gradients: d(gamma) = bstats[C:2C], d(beta) = bstats[0:C] (a single-replica block). */
int yunet_bn_param_grad(const double* bstats, float* dgamma, float* dbeta, int C,
                        int accumulate, void* stream);

/* All BatchNorm layers of the model in ONE launch.  table (device, int32 [n,7]) rows:
 * {stats offset (doubles, into stats_base), C, count, running offset (floats),
 *  dgamma offset, dbeta offset (floats, into grad_base), slots (YunetBN::slots of the [slots][2C] block)}.
 * mode 0: running_mean/var update from the forward sums at stats_base + off;
 * mode 1: d(gamma), d(beta) from the backward sums at stats_base + off;
 * mode 2: eval() -- WRITES sums at stats_base + off whose mean / variance equal the running
 *         statistics, so the same forward kernels apply BatchNorm in eval mode. */
int yunet_bn_batch(const int32_t* table, int n, const double* stats_base, float* running_mean,
                   float* running_var, float momentum, float* grad_base, int mode, void* stream);

/* out[j] (+)= sum_b partials[b, j]  for j < width, deterministic order. */
int yunet_reduce_partials(const float* partials, int blocks, int width, float* out,
                          int accumulate, void* stream);

/* The same reduction for a whole table of partial buffers in ONE launch (all weight-gradient
 * reductions of a backward pass: ~20 tiny kernels otherwise).  `jobs` is a DEVICE array;
 * chunk0 = index of the job's first 64-column chunk in the grid (running sum of
 * ceil(width / 64)), total_chunks = the grid size. */
typedef struct YunetReduceJob {
    const float* partials;   /* [blocks, width] */
    float* out;              /* [width] */
    int32_t blocks, width, accumulate, chunk0;
} YunetReduceJob;
int yunet_reduce_partials_batch(const YunetReduceJob* jobs, int njobs, int total_chunks, void* stream);

/* ---- loss step (mmdet/models/dense_heads/yunet_head.py:418-604) ---------------------- */

#define YUNET_MAX_LEVELS 5
typedef struct YunetLevels {
    int32_t num_levels;
    int32_t h[YUNET_MAX_LEVELS], w[YUNET_MAX_LEVELS], stride[YUNET_MAX_LEVELS];
} YunetLevels;

/* Fused MlvlPointGenerator priors (core/anchor/point_generator.py:80-175, offset 0) +
 * _bbox_decode (yunet_head.py:376-386) + SimOTAAssigner._assign
 * (core/bbox/assigners/sim_ota_assigner.py:95-257, bbox_overlaps
 * core/bbox/iou_calculators/iou2d_calculator.py:232-253) + PseudoSampler, three launches: compaction
 * of the valid priors, top-k per (image, GT) pair, conflict resolution per image (two interchangeable sets of launches:
 * option "assign_v2" below; identical outputs).
 * flat [N,P,16] = cls | dx dy dw dh | obj | 10 kps.
 * gt_boxes [N,Gmax,4] xyxy, gt_kps [N,Gmax,5,3] (x,y,vis), gt_labels [N,Gmax] or NULL,
 * gt_count [N].  Outputs: gt_inds [N,P] int32 (1-based, 0 = background),
 * labels [N,P] int32 (-1 background) or NULL, max_overlaps [N,P] (-1e5 background),
 * img_stats [N,2] = {num_pos, sum of kps weights}.  scratch: [N,P,12] fp32 (ABI 3; 8 words before).
 * Ties at the k-th cost are broken towards the lowest prior index.
 * Limits (YUNET_EINVAL beyond them): P <= 65535 priors per image (16-bit candidate indices:
 * a 1760x1760 training crop; the shipped configs train at 320 - 640, P <= 8400), Gmax <= 4096. */
int yunet_assign(const float* flat, const float* gt_boxes, const float* gt_kps,
                 const int32_t* gt_labels, const int32_t* gt_count, const YunetLevels* lv,
                 int N, int P, int Gmax, float center_radius, int32_t* gt_inds,
                 int32_t* labels, float* max_overlaps, float* img_stats, float* scratch,
                 void* stream);

/* SimOTAAssigner's constructor arguments (sim_ota_assigner.py:25-33; ABI 6 -- rounds 1-3 compiled 10 / 3.0 / 1.0 in):
 * cost = cls_cost * cls_weight + iou_cost * iou_weight (+1e5 outside box-and-centre); dynamic k from the
 * candidate_topk largest IoUs (1 <= candidate_topk <= 16; the per-lane lists are compiled for 10 and for 16). */
typedef struct YunetAssignCfg {
    float center_radius;
    int32_t candidate_topk;
    float iou_weight, cls_weight;
} YunetAssignCfg;
/* yunet_assign_ex with explicit assigner parameters; in an op list YUNET_OP_ASSIGN reads f[0] = center_radius,
 * i[3] = candidate_topk (0: 10), f[1] = iou_weight, f[2] = cls_weight (both 0: 3.0 / 1.0). */
int yunet_assign_cfg(const float* flat, const float* pre_scores, const float* pre_boxes,
                     const float* gt_boxes, const float* gt_kps, const int32_t* gt_labels,
                     const int32_t* gt_count, const YunetLevels* lv, int N, int P, int Gmax,
                     const YunetAssignCfg* cfg, int32_t* gt_inds, int32_t* labels, float* max_overlaps,
                     float* img_stats, float* scratch, void* stream);

/* Same kernel driven through SimOTAAssigner.assign()'s own signature
 * (sim_ota_assigner.py:38-93): pred_scores [N,P] = sigmoid(cls)*sigmoid(obj) and
 * decoded_bboxes [N,P,4] are given instead of being derived from `flat` (flat may be NULL). */
int yunet_assign_ex(const float* flat, const float* pre_scores, const float* pre_boxes,
                    const float* gt_boxes, const float* gt_kps, const int32_t* gt_labels,
                    const int32_t* gt_count, const YunetLevels* lv, int N, int P, int Gmax,
                    float center_radius, int32_t* gt_inds, int32_t* labels, float* max_overlaps,
                    float* img_stats, float* scratch, void* stream);

/* norm[0] = sum_n num_pos / world (the rank-local term of reduce_mean, yunet_head.py:493-497,
 * to be all-reduced by the caller when world > 1), norm[1] = sum_n kps weight. */
int yunet_loss_norm(const float* img_stats, int N, float inv_world, float* norm, void* stream);

/* box losses of mmdet/models/losses/iou_loss.py: EIoULoss :194-227 | DIoULoss :137-172 | IoULoss :14-50 (mode linear /
 * square / log; YuNet_Head's own default is mode 'square', yunet_head.py:59-64) | GIoULoss :103-120 | CIoULoss :230-293 */
enum { YUNET_BOX_EIOU = 0, YUNET_BOX_DIOU = 1, YUNET_BOX_IOU_LINEAR = 2, YUNET_BOX_IOU_SQUARE = 3, YUNET_BOX_IOU_LOG = 4,
       YUNET_BOX_GIOU = 5, YUNET_BOX_CIOU = 6 };
typedef struct YunetLossCfg {
    int32_t box_loss;            /* YUNET_BOX_*  (losses/iou_loss.py:194-227 / 137-172) */
    float w_cls, w_box, w_obj, w_kps;   /* loss_weight of each term                    */
    float box_eps;               /* EIoULoss/DIoULoss eps (1e-6)                       */
    float smooth_point;          /* EIoU 0.1                                           */
    float kps_beta;              /* SmoothL1 beta (1/9)                                */
    int32_t defer_num_total;     /* ABI 6, multi-GPU: 1 = norm[0] is NOT read -- loss_cls / loss_bbox / loss_obj and
                                  * their d/d(flat) leave yunet_loss un-normalised; yunet_loss_finalize_ex applies
                                  * 1 / max(num_total, 1) once the all-reduce of num_pos (yunet_head.py:493-497) has
                                  * landed, so that collective runs beside the loss kernel instead of in front of it */
} YunetLossCfg;

/* The four YuNet_Head losses and d(loss_i)/d(flat) in one pass (yunet_head.py:506-532,
 * losses/cross_entropy_loss.py:85-145, iou_loss.py, smooth_l1_loss.py:10-32,
 * losses/utils.py:29-55).  norm[0] is the (all-reduced) mean num_pos, clamped to >= 1
 * inside.  dflat [N,P,16] receives d(loss_c)/d(flat) for the loss that owns channel c,
 * each with unit upstream gradient.  partials [blocks,4] -> losses via
 * yunet_loss_finalize: losses[5] = {cls, bbox, obj, kps, total}, total = ((cls+bbox)+obj)+kps
 * in fp32 -- the sum _parse_losses builds (mmdet/models/detectors/base.py:206-209).
 * `mirror` (nullable) receives the same five floats a second time: the host keeps it behind
 * the flat gradient buffer so that the logged scalars ride in the gradient all-reduce. */
int yunet_loss(const float* flat, const int32_t* gt_inds, const float* max_overlaps,
               const float* gt_boxes, const float* gt_kps, const YunetLevels* lv,
               const YunetLossCfg* cfg, const float* norm, int N, int P, int Gmax,
               float* dflat, float* partials, int blocks, void* stream);
int yunet_loss_finalize(const float* partials, int blocks, float* losses, float* mirror,
                        void* stream);
/* The same with the deferred normaliser of YunetLossCfg.defer_num_total: num_total (device, nullable) = the
 * all-reduced mean num_pos; the first three losses are multiplied by 1 / max(num_total[0], 1) and dy_norm [16]
 * (nullable) receives the per-channel factor of d(loss)/d(flat) -- that value for cls | dx dy dw dh | obj, 1 for the
 * ten kps channels (normalised by the rank-local weight sum inside yunet_loss) -- which the fused head units take
 * as YunetDP.dy_scale.  (x * 1.0 is exact: gradients are bit-identical to the undeferred form.)
 * In an op list: YUNET_OP_LOSS_FINALIZE with p[3] = num_total, p[4] = dy_norm. */
int yunet_loss_finalize_ex(const float* partials, int blocks, float* losses, float* mirror,
                           const float* num_total, float* dy_norm, void* stream);
int yunet_loss_blocks(int N, int P);

/* out[i] = a[i] + b[i] (ABI 9).  The one place the reference's head concatenates predictions that come from DIFFERENT
 * inputs: with per-level towers (YuNet_Head(stacked_convs > 0), yunet_head.py:115-147, 191-207) the cls map is computed
 * from the cls tower and bbox / obj / kps from the reg tower, then flattened side by side (:456-472).  Here each tower
 * feeds one fused 64 -> 16 head unit whose rows for the other tower's channels are zero, and the two [N,P,16] outputs
 * (exact zeros in the foreign channels) are added.  `out` may alias `a`. */
int yunet_add(const float* a, const float* b, float* out, size_t n, void* stream);

/* ---- optimizer (torch.optim.SGD semantics, configs/yunet_n.py:1) --------------------- */
/* g = grad*grad_scale + wd*p;  buf = first ? g : momentum*buf + g;  p -= lr*buf.
 * lr is read from device memory (lr_dev[0]) so schedules do not need a re-capture. */
int yunet_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n,
                   const float* lr_dev, float momentum, float weight_decay, float grad_scale,
                   int first_step, void* stream);
/* ABI 9: the remaining arguments of torch.optim.SGD (torch/optim/sgd.py _single_tensor_sgd):
 *   buf = first ? g : momentum*buf + (1 - dampening)*g;   p -= lr * (nesterov ? g + momentum*buf : buf);
 * momentum == 0: p -= lr*g and momentum_buf is not touched (may be NULL).  nesterov needs momentum > 0 and
 * dampening == 0 (YUNET_EINVAL otherwise, like torch's ValueError). */
int yunet_sgd_step_ex(float* params, const float* grads, float* momentum_buf, int64_t n,
                      const float* lr_dev, float momentum, float dampening, int nesterov,
                      float weight_decay, float grad_scale, int first_step, void* stream);

/* ---- op-list executor ---------------------------------------------------------------- */
/* A training step is a fixed sequence of the calls above; the host builds it once as an
 * array of YunetOp and replays it with one FFI call per phase. */
enum {
    YUNET_OP_STEM_FWD = 1, YUNET_OP_STEM_BWD, YUNET_OP_DP_FWD, YUNET_OP_DP_BWD,
    YUNET_OP_POOL_FWD, YUNET_OP_POOL_BWD, YUNET_OP_UPADD_FWD, YUNET_OP_UPADD_BWD,
    YUNET_OP_BN_RUNNING, YUNET_OP_BN_PARAM_GRAD, YUNET_OP_REDUCE_PARTIALS,
    YUNET_OP_ASSIGN, YUNET_OP_LOSS_NORM, YUNET_OP_LOSS, YUNET_OP_LOSS_FINALIZE,
    YUNET_OP_SGD, YUNET_OP_MEMSET, YUNET_OP_BN_BATCH, YUNET_OP_REDUCE_BATCH,
    YUNET_OP_FORK, YUNET_OP_JOIN,
    YUNET_OP_ADD      /* ABI 9: yunet_add(p[0], p[1], p[2], n = i[1] << 32 | i[0]) */
};
/* Lanes (ABI 4).  The head chains of the pyramid levels (share conv -> fused head, and their backward) are
 * mutually independent: mmdet/models/dense_heads/yunet_head.py:175-247 walks them in a Python loop, and on the
 * small levels one launch has ~200 tiles for 256 CUs.  An op with i[YUNET_OP_LANE] = L > 0 is launched on the
 * executor's side stream L (two side streams, created on first use) instead of `stream`:
 *   YUNET_OP_FORK  i[0] = bit mask of lanes: those side streams wait for everything enqueued on `stream` so far;
 *   YUNET_OP_JOIN  i[0] = bit mask of lanes: `stream` waits for everything enqueued on those side streams.
 * A list must JOIN every lane it FORKed before it ends.  yunet_exec_lanes(0) makes the executor ignore lanes
 * (everything on `stream`, FORK / JOIN become no-ops): per-launch timing, debugging. */
#define YUNET_OP_LANE 10
#define YUNET_MAX_LANES 2
/* Groups (ABI 10).  A YUNET_OP_DP_FWD op with i[YUNET_OP_GROUP] = g in 2 .. YUNET_DP_GROUP_MAX declares that it and the
 * g - 1 ops after it (all YUNET_OP_DP_FWD, same lane, same storage type) are mutually independent: the executor hands them
 * to yunet_dp_fwd_group in one call.  0 / 1: an ordinary op.  An executor call that starts in the middle of a group (a
 * one-op replay for timing) runs the ops one by one. */
#define YUNET_OP_GROUP 9
typedef struct YunetOp {
    int32_t opcode;
    int32_t i[12];
    float f[8];
    void* p[12];
    YunetBN bn[2];
    YunetDP dp;
    YunetLevels lv;
    YunetLossCfg loss;
} YunetOp;
int yunet_exec(const YunetOp* ops /* HOST array */, int n_ops, void* stream);
int yunet_exec_lanes(int enable);   /* returns the previous setting */

/* ---- the conv stack with bf16 activation storage (YUNET_BF16, see the enum above) ----------------
 * Same arguments as the entry points without the suffix; pointers to ACTIVATION tensors (z, x, pool /
 * upsample-add inputs and outputs) then address bf16 elements, everything else (image, dy, dx, the
 * heads' [N,P,16] output, weights, partials, BN sums) is fp32 / fp64 as before.  yunet_exec selects
 * them per op (YunetDP.x_dtype, YunetOp.i[11]). */
int yunet_stem_fwd_bf16(const float* img, const float* w, const float* b, float* z, double* stats,
                        int N, int H, int W, int cmid, void* stream);
int yunet_stem_bwd_bf16(const float* img, const float* z, const float* dy, const YunetBN* bn,
                        float* wgrad_partials, int wgrad_blocks, int N, int H, int W, int cmid,
                        void* stream);
int yunet_dp_fwd_bf16(const YunetDP* d, void* stream);
int yunet_dp_fwd_group_bf16(const YunetDP* const* units, int n, void* stream);
int yunet_dp_bwd_bf16(const YunetDP* d, void* stream);
int yunet_pool_fwd_bf16(const float* z, const YunetBN* bn, float* out, int N, int H, int W, int C,
                        void* stream);
int yunet_pool_bwd_bf16(const float* z, const YunetBN* bn, const float* dy_out, float* dx,
                        int accumulate, int N, int H, int W, int C, void* stream);
int yunet_pool_bwd_add_bf16(const float* z, const YunetBN* bn, const float* dy_out, const float* extra, float* dx,
                            int accumulate, int N, int H, int W, int C, void* stream);
int yunet_upadd_fwd_bf16(const float* za, const YunetBN* bna, const float* zb, const YunetBN* bnb,
                         float* out, int N, int H, int W, int C, void* stream);
int yunet_upadd_bwd_bf16(const float* za, const YunetBN* bna, const float* zb, const YunetBN* bnb,
                         const float* dout, float* dxa, int accumulate_a, float* dxb,
                         int accumulate_b, int N, int H, int W, int C, void* stream);

/* ---- detection post-processing -------------------------------------
 * YuNet_Head.get_bboxes (mmdet/models/dense_heads/yunet_head.py:290-416): priors, sigmoid scores
 * (cls * obj) >= score_thr, _bbox_decode, then mmcv.ops.batched_nms for the single face class =
 * greedy NMS (IoU with offset 0, suppress when IoU > iou_thr), survivors in descending score.
 *  flat [N,P,16] raw head outputs of an eval-mode forward; any P (candidates above the threshold
 *  are compacted first; up to 16384 of them sort in LDS, more in the scratch -- an origin-size
 *  WIDER image of 1024x1024 has P = 21504, tools/test_widerface.py --mode 2).
 *  dets [N,max_out,5] = x1 y1 x2 y2 score; kps [N,max_out,10] decoded landmarks or NULL
 *  (_kps_decode, yunet_head.py:388-393); count [N]; scratch >= yunet_detect_scratch_bytes(N, P). */
size_t yunet_detect_scratch_bytes(int N, int P);
/* mmcv.ops.batched_nms for ONE class on explicit candidates (the merge step of test-time
 * augmentation, mmdet/models/dense_heads/dense_test_mixins.py:89-103): per set n the first
 * counts[n] (or all K when counts == NULL) rows of boxes [N,K,4] / scores [N,K] with
 * score >= score_thr, greedy NMS in descending score (ties: lower index first).
 * dets [N,max_out,5], keep [N,max_out] (indices into the set, or NULL), count [N];
 * scratch >= yunet_detect_scratch_bytes(N, K). */
int yunet_nms(const float* boxes, const float* scores, const int32_t* counts, int N, int K,
              float score_thr, float iou_thr, int max_out, float* dets, int32_t* keep,
              int32_t* count, void* scratch, void* stream);
int yunet_detect(const float* flat, const YunetLevels* lv, int N, int P, float score_thr,
                 float iou_thr, int max_out, float* dets, float* kps, int32_t* count, void* scratch,
                 void* stream);

/* ---- device input pipeline ----------------------------------------
 * The reference's TRAIN pipeline (configs/yunet_n.py:36-56) on the device:
 *   RandomSquareCrop  mmdet/datasets/pipelines/transforms.py:975-1169
 *   Resize(keep_ratio=False)          transforms.py:242-299 (mmcv.imresize -> cv2 INTER_LINEAR)
 *   RandomFlip + 5-landmark swap      transforms.py:425-546
 *   collate to padded GT              mmdet/datasets/pipelines/formatting.py:206-249
 * Random draws come from a counter-based 32-bit generator keyed by (seed, iteration, image) --
 * restated in oracle/pipeline_oracle.py and pinned against the unmodified reference classes. */
typedef struct YunetAugCfg {
    int32_t out_size;        /* S of Resize(img_scale=(S, S)) */
    int32_t n_choice;        /* entries used in crop_choice, 1..8 */
    double crop_choice[8];   /* RandomSquareCrop(crop_choice=...); double: cw = int(scale * short) */
    double flip_ratio;       /* RandomFlip(flip_ratio=...) */
    float pad_value;         /* fill outside the source image (128, transforms.py:1128) */
    uint32_t seed;
    int32_t max_attempts;    /* window draws per scale (250, transforms.py:1058) */
    int32_t max_retries;     /* scale re-draws before giving up (the reference loops forever) */
    int32_t gmax;            /* rows of the padded GT outputs */
} YunetAugCfg;

/* Per image: decide scale / window / flip, transform and compact the kept boxes + keypoints.
 *  src_hw [N,2] (h, w); boxes [sum G,4] xyxy; kps [sum G,5,3]; gt_off [N+1] prefix offsets;
 *  params [N,8] int32 = left, top, cw (0 = failed), flip, kept, draws, status, 0
 *      status 0 ok, 1 no window with a box centre inside (or no GT), 2 kept > gmax (truncated);
 *  out_boxes [N,gmax,4], out_kps [N,gmax,5,3] (rows >= count zeroed), out_count [N]. */
int yunet_aug_decide(const int32_t* src_hw, const float* boxes, const float* kps, const int32_t* gt_off,
                     const YunetAugCfg* cfg, uint32_t iteration, int N, int32_t* params,
                     float* out_boxes, float* out_kps, int32_t* out_count, void* stream);
/* Crop (pad 128 outside the source) -> bilinear resize to S x S -> flip, uint8 HWC sources
 * (src + src_off[n], BGR as loaded) to planar fp32 [N,3,S,S]; `params` from yunet_aug_decide. */
int yunet_aug_pixels(const uint8_t* src, const long long* src_off, const int32_t* src_hw,
                     const int32_t* params, const YunetAugCfg* cfg, int N, float* out_img, void* stream);

/* Measurement switches of the dispatchers (ABI 6).  The library reads the environment ONCE, the first time an
 * option is needed (YUNET_NO_PACK, YUNET_BWD_FP32MMA, YUNET_BWD64_NW, YUNET_EW_GRID, YUNET_DP_FWD_BLOCKS_PER_CU);
 * after that only this call changes them -- no launch calls getenv.  Names:
 *   "no_pack"            1: the 20x20 / 10x10 levels on per-image tiles instead of the packed canvas
 *   "bwd_fp32mma"        1: every backward GEMM on the exact-fp32 matrix instruction (bench.py: exact_fp32_bwd)
 *   "bwd64_nw"           0 (by shape) | 4 | 8: waves per workgroup of the 64 -> 64 backward kernel
 *   "ew_grid"            workgroup cap of the element-wise backward kernels (0 restores the default, 768)
 *   "fwd_blocks_per_cu"  0 (occupancy API) | 1..4: resident forward workgroups per CU
 *   "fwd64s"             0: every fp32 64 -> 64 forward unit on the tile kernel | 1: the plain and fused-pooling units on
 *                        the wave-streaming kernel | 2 (default): the 20x20 / 10x10 levels too
 *   "fwd64s_rows"        0 (by shape) | rows per band of the wave-streaming kernel
 *   "bwd16s"             1 (default): the fp32 16 -> 16 backward unit on maps >= 32 x 64 on the wave-streaming kernel that
 *                        recomputes z (it does not read YunetDP.z) | 0: the tile kernel;  "bwd16s_rows": rows per band
 *   "fwd16s"             1 (default): the fp32 16 -> 16 / 16 -> 64 forward units on the wave-streaming kernels | 0: the tile kernels
 *   "stem_mma"           1 (default): the fp32 stem (yunet_stem_fwd; YUNET_OP_STEM_BWD with the stem's parameters in p[4],
 *                        p[5] -> yunet_stem_bwd_rz) as matrix products on the matrix cores | 0: the VALU tile kernels
 *   "bwd32_split"        1 (default): the 32 -> 64 backward unit (YuNet_s) on the split-bf16 matrix path | 0: exact-fp32 MFMA
 *   "upadd_coarse"       1 (default): yunet_upadd_bwd with dxa = NULL on the dedicated coarse-gradient kernel | 0: general kernel
 *   "assign_v2"          1 (default): yunet_assign* on the round-5 launches (compaction per 256-prior chunk, one workgroup per
 *                        (image, GT) pair with a candidate-pruned walk, one wave per conflict; needs Gmax < P, N <= 65535) | 0: one
 *                        workgroup per image for compaction / conflicts, every pair evaluated in full.  Same outputs, bit for bit.
 *   "fwd_group"          1 (default): yunet_dp_fwd_group puts independent plain 64 -> 64 units into one grid | 0: one launch each
 *   "oneshot_timeout_ms" how long yunet_allreduce waits for a peer (default 120 000; env YUNET_ONESHOT_TIMEOUT_MS)
 * "no_pack" and "bwd64_nw" change yunet_dp_bwd_blocks(): set them before any plan is built.
 * Returns the previous value, or YUNET_EINVAL for an unknown name / a value out of range. */
int yunet_set_option(const char* name, int value);

/* ---- gradient / num_pos exchange between the GPUs of one node (csrc/collective.hip) ---------------------
 * Replaces, for this path, what the reference gets from torch DDP over NCCL (mmdet/apis/train.py:152-163:
 * MMDistributedDataParallel averages the ~300 KB of gradients) and from reduce_mean
 * (mmdet/core/utils/dist_utils.py:68-74, called at yunet_head.py:493-497 for num_pos).  The messages are
 * latency-bound and xGMI is a point-to-point mesh, so instead of a ring every rank STORES its message into a
 * slot of every peer's inbox (peer-mapped device memory), raises a flag behind it, waits for the world's flags
 * in its own inbox and adds the slots in rank order: one kernel per rank, results bit-identical on all ranks.
 *
 * These are the only entry points that allocate: an inbox is uncached device memory that peers map through
 * hipIpc* handles (one process per GPU).  Set-up, per rank: yunet_comm_alloc -> yunet_comm_export -> exchange the
 * 64-byte handles over any host channel -> yunet_comm_open on every peer's handle -> fill a YunetComm.
 * Every rank must issue the same sequence of yunet_allreduce calls on a given YunetComm (one YunetComm per stream
 * that carries collectives).  A peer that never arrives makes the wait give up after option "oneshot_timeout_ms"
 * (default 120 000 = 2 min: a step lasts milliseconds, a peer that is two minutes late is gone): yunet_comm_status() then returns the
 * sequence number of that call and the buffer is POISONED with NaN (the shares of the blocks that gave up), so a
 * caller that never reads the status word cannot train on un-reduced gradients unnoticed. */
#define YUNET_MAX_RANKS 8
#define YUNET_IPC_HANDLE_BYTES 64
/* bytes in front of the message slots of an inbox (ABI 11: flags per (parity, rank, piece) + the local send counter;
 * messages above 64 KB travel as 2 / 4 / 8 pieces, one workgroup per peer and piece) */
#define YUNET_COMM_HEADER_BYTES 20480
typedef struct YunetComm {
    int32_t rank, world;
    uint32_t seq;                    /* calls made so far; incremented by yunet_allreduce (start at 0)          */
    int32_t reserved_;
    uint64_t slot_bytes;             /* capacity of one message: (inbox bytes - YUNET_COMM_HEADER_BYTES) / (2 * world) */
    void* inbox[YUNET_MAX_RANKS];    /* inbox of rank r as mapped into THIS process (own rank: the allocation) */
    int32_t* status;                 /* HOST word from yunet_comm_alloc                                        */
} YunetComm;
size_t yunet_comm_inbox_bytes(int world, size_t max_msg_bytes);
int yunet_comm_alloc(size_t bytes, void** inbox, int32_t** status);
int yunet_comm_free(void* inbox, int32_t* status);
int yunet_comm_export(void* inbox, void* handle64 /* HOST, YUNET_IPC_HANDLE_BYTES */);
int yunet_comm_open(const void* handle64 /* HOST */, void** mapped);
int yunet_comm_close(void* mapped);
/* buf[0..n) <- sum over ranks (mean != 0: divided by world), in place, n * 4 <= slot_bytes. */
int yunet_allreduce(YunetComm* c /* HOST */, float* buf, size_t n, int mean, void* stream);
int yunet_comm_status(const YunetComm* c /* HOST */);

int yunet_abi_version(void);
/* grid size the fused conv kernels are launched with (rows of wgrad_partials). */
int yunet_conv_blocks(void);

#ifdef __cplusplus
}
#endif
#endif /* YUNET_HIP_H */
