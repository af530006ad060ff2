// loss_step.hip -- YuNet_Head loss step on gfx950: priors + decode + SimOTA assignment,
// then the four losses with analytic gradients.  Compiled with -ffp-contract=off: the
// reference evaluates mul and add as separate fp32 roundings (eager torch), and the
// assignment is graded on integer outputs, so no FMA contraction is allowed here.
//
// Reference: mmdet/models/dense_heads/yunet_head.py:418-604,
// mmdet/core/bbox/assigners/sim_ota_assigner.py:95-257,
// mmdet/core/bbox/iou_calculators/iou2d_calculator.py:232-253,
// mmdet/core/anchor/point_generator.py:80-175, mmdet/models/losses/*.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yunet_hip.h"
#include "levels.h"

int yunet_option_assign_v2();        // api.hip: YunetOptions::assign_v2 (measurement switch, default 1)

#define ASSIGN_THREADS 1024
#define ASSIGN_WAVES (ASSIGN_THREADS / 64)
// candidate_topk of SimOTAAssigner: the per-lane lists of assign_topk_kernel are compiled for 10 (the shipped
// configs) and 16 entries; candidate_topk itself is a runtime argument <= the capacity
#define TOPK_MAX 16
#define INF_COST 100000.0f

namespace {

__device__ __forceinline__ float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

struct GT {
    float x1, y1, x2, y2;
};

// pairwise IoU exactly as bbox_overlaps (is_aligned=False, eps=1e-6)
__device__ __forceinline__ float iou_of(float bx1, float by1, float bx2, float by2, const GT& g) {
    float area1 = (bx2 - bx1) * (by2 - by1);
    float area2 = (g.x2 - g.x1) * (g.y2 - g.y1);
    float ltx = fmaxf(bx1, g.x1), lty = fmaxf(by1, g.y1);
    float rbx = fminf(bx2, g.x2), rby = fminf(by2, g.y2);
    float w = fmaxf(rbx - ltx, 0.0f), h = fmaxf(rby - lty, 0.0f);
    float overlap = w * h;
    float uni = (area1 + area2) - overlap;
    uni = fmaxf(uni, 1e-6f);
    return overlap / uni;
}

__device__ __forceinline__ bool in_gt_box(float cx, float cy, const GT& g) {
    float l = cx - g.x1, t = cy - g.y1, r = g.x2 - cx, b = g.y2 - cy;
    return fminf(fminf(l, t), fminf(r, b)) > 0.0f;
}

__device__ __forceinline__ bool in_gt_center(float cx, float cy, float s, float radius, const GT& g) {
    float gcx = (g.x1 + g.x2) / 2.0f, gcy = (g.y1 + g.y2) / 2.0f;
    float rs = radius * s;
    float l = cx - (gcx - rs), t = cy - (gcy - rs);
    float r = (gcx + rs) - cx, b = (gcy + rs) - cy;
    return fminf(fminf(l, t), fminf(r, b)) > 0.0f;
}

// cost[v,g] = (cls_cost*cls_weight + iou_cost*iou_weight) + (in_both ? 0 : 1e5)   (sim_ota_assigner.py:124-128)
struct CostW {
    float iou_w, cls_w;
};
__device__ __forceinline__ float cost_of(float cls_cost, float iou, bool in_both, CostW w) {
    float iou_cost = -logf(iou + 1e-7f);
    float c = cls_cost * w.cls_w + iou_cost * w.iou_w;
    return c + (in_both ? 0.0f : INF_COST);
}

// order-preserving map float -> uint32 (handles the slightly negative costs of iou == 1)
__device__ __forceinline__ uint32_t ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

// scratch record of a valid prior (compacted, ascending prior index)
struct __attribute__((aligned(16))) VRec {
    float x1, y1, x2, y2;  // decoded box
    float cls_cost, cx, cy, s;
};

// Work arrays of the assignment in the caller's scratch ([N,P,12] fp32 words):
//   rec [N][P] VRec (8 words) | cnt [N][P] int32 | vidx [N][P] u16 | mg [N][P] u16 | V [N] int32 | vc [N][chunks] int32
// (V: valid priors per image of the one-workgroup-per-image launches; vc: valid priors per 256-prior chunk of the
// round-5 launches, whose records live at slot 256 c + rank instead of one dense run)
struct AssignScratch {
    VRec* rec;
    int* cnt;
    uint16_t* vidx;
    uint16_t* mg;
    int* V;
    int* vc;
    uint32_t* items;     // [0] = number of (image, GT) pairs of the batch, [1 + i] = image << 16 | GT of pair i (image-major)
};
__host__ __device__ inline AssignScratch assign_scratch(float* scratch, int N, int P) {
    AssignScratch a;
    const size_t np = (size_t)N * P;
    a.rec = reinterpret_cast<VRec*>(scratch);
    a.cnt = reinterpret_cast<int*>(scratch + np * 8);
    a.vidx = reinterpret_cast<uint16_t*>(scratch + np * 9);
    a.mg = a.vidx + np;
    a.V = reinterpret_cast<int*>(scratch + np * 10);
    a.vc = reinterpret_cast<int*>(scratch + np * 11);       // ceil(P / 256) <= P entries per image
    a.items = reinterpret_cast<uint32_t*>(scratch + np * 10);   // 1 + N * Gmax <= N * P entries (round-5 launches: Gmax < P)
    return a;
}

// The assignment runs as three launches.  The middle one -- cost and IoU of every valid prior against one
// GT, top-10 lists, dynamic k -- is all the arithmetic (V x G x ~200 instructions per image) and used to run
// inside one workgroup per image, 16 waves taking the image's GTs in turn: the step waited for the image
// with the most faces while most CUs idled.  As its own launch with one wave per (image, GT) pair the pairs
// of the whole batch spread over the chip.
//
// ---- A: decode, region tests, ordered compaction of the valid priors (one workgroup per image) -----------
__global__ __launch_bounds__(ASSIGN_THREADS) void assign_compact_kernel(
    const float* __restrict__ flat, const float* __restrict__ gt_boxes, const int32_t* __restrict__ gt_count,
    Levels L, int P, int Gmax, float radius, int32_t* __restrict__ gt_inds, int32_t* __restrict__ labels,
    float* __restrict__ max_overlaps, AssignScratch ws, const float* __restrict__ pre_scores,
    const float* __restrict__ pre_boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GT* s_gt = reinterpret_cast<GT*>(smem);
    __shared__ int s_wave_off[ASSIGN_WAVES + 1];

    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = max(0, min(gt_count[n], Gmax));      // (a negative count from an unvalidated data source is an empty image)
    const float* fl = flat + (size_t)n * P * 16;
    VRec* rec_out = ws.rec + (size_t)n * P;
    int* cnt = ws.cnt + (size_t)n * P;
    uint16_t* vidx = ws.vidx + (size_t)n * P;
    uint16_t* mg = ws.mg + (size_t)n * P;

    for (int g = tid; g < G; g += ASSIGN_THREADS) {
        const float* b = gt_boxes + ((size_t)n * Gmax + g) * 4;
        s_gt[g] = GT{b[0], b[1], b[2], b[3]};
    }
    __syncthreads();

    int vbase = 0;
    for (int p0 = 0; p0 < P; p0 += ASSIGN_THREADS) {
        const int p = p0 + tid;
        bool valid = false;
        VRec rec;
        if (p < P) {
            // the prior's head outputs go in flight before the region tests (only valid priors use them: the load used to sit
            // behind the loop over the GTs, one exposed memory latency per chunk)
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 b = make_float2(0.f, 0.f);
            if (!pre_scores) {
                a = *reinterpret_cast<const float4*>(fl + (size_t)p * 16);
                b = *reinterpret_cast<const float2*>(fl + (size_t)p * 16 + 4);
            }
            float px, py, s;
            prior_of(L, p, px, py, s);
            const float cx = px + s * 0.5f, cy = py + s * 0.5f;
            for (int g = 0; g < G; ++g) {
                const GT gt = s_gt[g];
                if (in_gt_box(cx, cy, gt) || in_gt_center(cx, cy, s, radius, gt)) {
                    valid = true;
                    break;
                }
            }
            gt_inds[(size_t)n * P + p] = 0;
            max_overlaps[(size_t)n * P + p] = -INF_COST;
            if (labels) labels[(size_t)n * P + p] = -1;
            if (valid && pre_scores) {
                // stand-alone SimOTAAssigner.assign(): scores and decoded boxes are inputs
                const float4 bx = *reinterpret_cast<const float4*>(pre_boxes + ((size_t)n * P + p) * 4);
                rec.x1 = bx.x; rec.y1 = bx.y; rec.x2 = bx.z; rec.y2 = bx.w;
                rec.cls_cost = -fmaxf(logf(sqrtf(pre_scores[(size_t)n * P + p])), -100.0f);
                rec.cx = cx; rec.cy = cy; rec.s = s;
            } else if (valid) {
                // a = cls, dx, dy, dw ; b = dh, obj
                const float bx = a.y * s + px, by = a.z * s + py;
                const float bw = expf(a.w) * s, bh = expf(b.x) * s;
                rec.x1 = bx - bw / 2.0f;
                rec.y1 = by - bh / 2.0f;
                rec.x2 = bx + bw / 2.0f;
                rec.y2 = by + bh / 2.0f;
                const float score = sigmoidf_ref(a.x) * sigmoidf_ref(b.y);
                // F.binary_cross_entropy(sqrt(score), 1) = -max(log(sqrt(score)), -100)
                rec.cls_cost = -fmaxf(logf(sqrtf(score)), -100.0f);
                rec.cx = cx;
                rec.cy = cy;
                rec.s = s;
            }
        }
        const unsigned long long bal = __ballot(valid);
        const int wcnt = __popcll(bal);
        if (lane == 0) s_wave_off[wid] = wcnt;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int w = 0; w < ASSIGN_WAVES; ++w) {
                int c = s_wave_off[w];
                s_wave_off[w] = acc;
                acc += c;
            }
            s_wave_off[ASSIGN_WAVES] = acc;
        }
        __syncthreads();
        if (valid) {
            const int v = vbase + s_wave_off[wid] + __popcll(bal & ((1ull << lane) - 1ull));
            vidx[v] = (uint16_t)p;
            cnt[v] = 0;
            mg[v] = 0;
            rec_out[v] = rec;
        }
        vbase += s_wave_off[ASSIGN_WAVES];
        __syncthreads();
    }
    if (tid == 0) ws.V[n] = vbase;
}

// ---- B: one wave per (image, GT): dynamic k from the top-10 IoUs, then the k cheapest -------------------
#define TOPK_WAVES 4
template <int TOPK>
__global__ __launch_bounds__(TOPK_WAVES * 64) void assign_topk_kernel(
    const float* __restrict__ gt_boxes, const int32_t* __restrict__ gt_count, int P, int Gmax, int gblocks,
    float radius, int topk, CostW cw, AssignScratch ws) {
    const int n = blockIdx.x / gblocks;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = (blockIdx.x - n * gblocks) * TOPK_WAVES + wid;
    const int G = max(0, min(gt_count[n], Gmax));      // (a negative count from an unvalidated data source is an empty image)
    const int V = ws.V[n];
    if (g >= G || V <= 0) return;                 // wave-uniform
    const VRec* __restrict__ rec = ws.rec + (size_t)n * P;
    int* cnt = ws.cnt + (size_t)n * P;
    uint16_t* mg = ws.mg + (size_t)n * P;
    const float* b = gt_boxes + ((size_t)n * Gmax + g) * 4;
    const GT gt{b[0], b[1], b[2], b[3]};
    float ti[TOPK];
    float tc[TOPK];
    int tv[TOPK];
#pragma unroll
    for (int i = 0; i < TOPK; ++i) {
        ti[i] = -1.0f;
        tc[i] = 3.0e38f;
        tv[i] = 0x7fffffff;
    }
    VRec nxt = rec[lane < V ? lane : 0];               // the next record is in flight while one is ranked
    for (int v = lane; v < V; v += 64) {
        const VRec r = nxt;
        if (v + 64 < V) nxt = rec[v + 64];
        float iou = iou_of(r.x1, r.y1, r.x2, r.y2, gt);
        const bool both = in_gt_box(r.cx, r.cy, gt) && in_gt_center(r.cx, r.cy, r.s, radius, gt);
        float c = cost_of(r.cls_cost, iou, both, cw);
        int cv = v;
        if (iou > ti[TOPK - 1]) {
#pragma unroll
            for (int i = 0; i < TOPK; ++i) {
                const bool sw = iou > ti[i];
                const float t = ti[i];
                ti[i] = sw ? iou : t;
                iou = sw ? t : iou;
            }
        }
        if (c < tc[TOPK - 1]) {
#pragma unroll
            for (int i = 0; i < TOPK; ++i) {
                const bool sw = c < tc[i];  // strict: equal costs keep ascending v
                const float t = tc[i];
                const int u = tv[i];
                tc[i] = sw ? c : t;
                tv[i] = sw ? cv : u;
                c = sw ? t : c;
                cv = sw ? u : cv;
            }
        }
    }
    // merge: sum of the wave-wide top-10 IoUs in descending order
    float sum = 0.0f;
    const int K = V < topk ? V : topk;         // candidate_topk = min(self.candidate_topk, ious.size(0))
    for (int k = 0; k < K; ++k) {
        const float m = wave_max_f(ti[0]);
        const unsigned long long who = __ballot(ti[0] == m);
        const int winner = __ffsll((long long)who) - 1;
        if (lane == winner) {
#pragma unroll
            for (int i = 0; i < TOPK - 1; ++i) ti[i] = ti[i + 1];
            ti[TOPK - 1] = -1.0f;
        }
        sum = sum + m;
    }
    int dk = (int)sum;  // .int(): truncation toward zero
    dk = dk < 1 ? 1 : dk;
    // at most candidate_topk IoUs <= 1 were added, so dk <= candidate_topk for finite predictions.  A diverged model (inf / NaN boxes)
    // can make the sum inf and (int)inf = INT_MAX: bound the loop so that a diverged run reports NaN losses, as the
    // reference does, instead of spinning for 2^31 iterations per (image, GT) pair
    dk = dk > topk ? topk : dk;
    for (int k = 0; k < dk; ++k) {
        const unsigned long long key = ((unsigned long long)ord(tc[0]) << 32) | (uint32_t)tv[0];
        const unsigned long long kmin = wave_min_u64(key);
        if (key == kmin && tv[0] != 0x7fffffff) {
            const int v = tv[0];
            atomicAdd(&cnt[v], 1);
            mg[v] = (uint16_t)g;       // read back only where cnt ends at 1: then this is the only writer
#pragma unroll
            for (int i = 0; i < TOPK - 1; ++i) {
                tc[i] = tc[i + 1];
                tv[i] = tv[i + 1];
            }
            tc[TOPK - 1] = 3.0e38f;
            tv[TOPK - 1] = 0x7fffffff;
        }
    }
}

// ---- C: conflicts -> argmin over ALL gts; outputs; per-image statistics (one workgroup per image) -------
__global__ __launch_bounds__(ASSIGN_THREADS) void assign_resolve_kernel(
    const float* __restrict__ gt_boxes, const float* __restrict__ gt_kps, const int32_t* __restrict__ gt_labels,
    const int32_t* __restrict__ gt_count, int P, int Gmax, float radius, CostW cw, int32_t* __restrict__ gt_inds,
    int32_t* __restrict__ labels, float* __restrict__ max_overlaps, float* __restrict__ img_stats,
    AssignScratch ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GT* s_gt = reinterpret_cast<GT*>(smem);
    __shared__ float s_red[ASSIGN_WAVES][2];
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = max(0, min(gt_count[n], Gmax));      // (a negative count from an unvalidated data source is an empty image)
    const int V = ws.V[n];
    const VRec* __restrict__ rec = ws.rec + (size_t)n * P;
    const int* cnt = ws.cnt + (size_t)n * P;
    const uint16_t* vidx = ws.vidx + (size_t)n * P;
    const uint16_t* mg = ws.mg + (size_t)n * P;
    for (int g = tid; g < G; g += ASSIGN_THREADS) {
        const float* b = gt_boxes + ((size_t)n * Gmax + g) * 4;
        s_gt[g] = GT{b[0], b[1], b[2], b[3]};
    }
    __syncthreads();
    float npos = 0.0f, wsum = 0.0f;
    for (int v = tid; v < V; v += ASSIGN_THREADS) {
        const int c = cnt[v];
        if (c == 0) continue;
        const VRec r = rec[v];
        int g = mg[v];
        if (c > 1) {
            float best = 3.0e38f;
            for (int j = 0; j < G; ++j) {
                const GT gt = s_gt[j];
                const float iou = iou_of(r.x1, r.y1, r.x2, r.y2, gt);
                const bool both = in_gt_box(r.cx, r.cy, gt) && in_gt_center(r.cx, r.cy, r.s, radius, gt);
                const float cj = cost_of(r.cls_cost, iou, both, cw);
                if (cj < best) {
                    best = cj;
                    g = j;
                }
            }
        }
        const float iou = iou_of(r.x1, r.y1, r.x2, r.y2, s_gt[g]);
        const int p = vidx[v];
        gt_inds[(size_t)n * P + p] = g + 1;
        max_overlaps[(size_t)n * P + p] = iou;
        if (labels) labels[(size_t)n * P + p] = gt_labels ? gt_labels[(size_t)n * Gmax + g] : 0;
        const float* kp = gt_kps + ((size_t)n * Gmax + g) * 15;
        // torch.mean over the 5 visibility flags
        const float w = ((((kp[2] + kp[5]) + kp[8]) + kp[11]) + kp[14]) / 5.0f;
        npos += 1.0f;
        wsum += w;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        npos += __shfl_xor(npos, o, 64);
        wsum += __shfl_xor(wsum, o, 64);
    }
    if (lane == 0) {
        s_red[wid][0] = npos;
        s_red[wid][1] = wsum;
    }
    __syncthreads();
    if (tid == 0) {
        float a = 0.0f, b = 0.0f;
        for (int w = 0; w < ASSIGN_WAVES; ++w) {
            a += s_red[w][0];
            b += s_red[w][1];
        }
        img_stats[n * 2 + 0] = a;
        img_stats[n * 2 + 1] = b;
    }
}

// ===== round 5: the same three steps, balanced over the chip (option assign_v2, default) ==============================
// What the launches above cost at the bench batch (256 images, WIDER's faces-per-image histogram: 5 % of the images
// carry 64 faces): 53 + 61 + 39 us, and none of it is memory traffic -- every one of them is the dependent chain of its
// slowest wave.  `compact` and `resolve` run ONE workgroup per image whose chain grows with the image's face count (a
// thread loops over every GT), so the launch lasts as long as the most crowded image while the other CUs idle; `topk`
// evaluates the full cost (division, logarithm, two 10-deep insertion networks: ~200 dependent instructions) for every
// (valid prior, GT) pair although a GT can only ever select priors whose box it overlaps or whose centre lies in its
// box-and-centre region, fetches one record per round trip to memory, and merges the per-lane lists with reductions
// built from LDS permutes.  Here
//   A2  compacts per CHUNK of 256 priors (grid = images x chunks) with CQ = 2 threads per prior, each looping over half
//       of the GTs: valid priors of chunk c occupy the slots 256 c .. 256 c + vc[n][c] - 1 of the image's work arrays --
//       ascending prior index, so "lowest slot" is still "lowest prior index" in every tie;
//   B2  (one workgroup of four waves per (image, GT) pair, pairs from a dense list A2 writes) stages the image's records in
//       blocks of 512 through LDS, the next block's loads in flight under the current block's arithmetic; a wave spends
//       ~60 instructions per prior on the overlap / region tests, compacts the priors that can matter into a per-wave
//       list and runs the expensive part on those alone (exactness: see the kernel); wave reductions on the DPP row
//       operations; the four waves' lists are merged through LDS;
//   C2  resolves each conflict with a whole wave (one GT per lane) instead of one lane looping over every GT.
// Outputs are bit-identical to the launches above (tests/test_loss_step_gpu.py runs both).  Measured (profiles/
// r05_assign_kernels.log): A2 24 us, B2 61 us (11 fixed + 12 walk + 8 evaluation + 28 list merges), C2 13 us.
#define CCH 256                      // priors per chunk of A2
#define CCH_SHIFT 8
#ifndef A2_CQ
#define A2_CQ 2                       // measured at the bench batch: 1 -> 24.9 us, 2 -> 22.7 us, 4 -> 29.3 us
#endif
#define CQ A2_CQ                    // threads per prior of A2 (each takes the GTs q, q + CQ, ...)
#define RB 512                       // records per LDS block of B2 (= capacity of the per-wave candidate list)

__device__ __forceinline__ float overlap_of(float bx1, float by1, float bx2, float by2, const GT& g) {
    float ltx = fmaxf(bx1, g.x1), lty = fmaxf(by1, g.y1);
    float rbx = fminf(bx2, g.x2), rby = fminf(by2, g.y2);
    float w = fmaxf(rbx - ltx, 0.0f), h = fmaxf(rby - lty, 0.0f);
    return w * h;            // the `overlap` of iou_of(), same operations
}

// Wave-wide reductions on the DPP row operations of gfx9 (xor 1, xor 2 inside quads, half-row mirror, row mirror, then
// lane 15 / lane 31 broadcast into the following rows): six VALU instructions instead of six LDS permutes (~100+ clocks
// each).  max and the 64-bit min are exact and order-independent, so the result equals the shuffle version's.  All 64
// lanes must be active.
#define DPP_I(x, ctrl, rm) __builtin_amdgcn_update_dpp((x), (x), (ctrl), (rm), 0xf, false)
__device__ __forceinline__ float wave_max_f_dpp(float v) {
#define DPP_MAX_STEP(ctrl, rm) v = fmaxf(v, __int_as_float(DPP_I(__float_as_int(v), ctrl, rm)))
    DPP_MAX_STEP(0xB1, 0xf);      // quad_perm [1,0,3,2]
    DPP_MAX_STEP(0x4E, 0xf);      // quad_perm [2,3,0,1]
    DPP_MAX_STEP(0x141, 0xf);     // row_half_mirror
    DPP_MAX_STEP(0x140, 0xf);     // row_mirror
    DPP_MAX_STEP(0x142, 0xa);     // row_bcast:15 into rows 1, 3
    DPP_MAX_STEP(0x143, 0xc);     // row_bcast:31 into rows 2, 3
#undef DPP_MAX_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned long long wave_min_u64_dpp(unsigned long long k) {
    int hi = (int)(uint32_t)(k >> 32), lo = (int)(uint32_t)k;
#define DPP_MIN_STEP(ctrl, rm)                                                               \
    {                                                                                        \
        const int th = DPP_I(hi, ctrl, rm), tl = DPP_I(lo, ctrl, rm);                        \
        const bool take = (uint32_t)th < (uint32_t)hi || (th == hi && (uint32_t)tl < (uint32_t)lo); \
        hi = take ? th : hi;                                                                 \
        lo = take ? tl : lo;                                                                 \
    }
    DPP_MIN_STEP(0xB1, 0xf)
    DPP_MIN_STEP(0x4E, 0xf)
    DPP_MIN_STEP(0x141, 0xf)
    DPP_MIN_STEP(0x140, 0xf)
    DPP_MIN_STEP(0x142, 0xa)
    DPP_MIN_STEP(0x143, 0xc)
#undef DPP_MIN_STEP
    hi = __builtin_amdgcn_readlane(hi, 63);
    lo = __builtin_amdgcn_readlane(lo, 63);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int wave_sum_i_dpp(int v) {
#define DPP_ADD_STEP(ctrl, rm, bc) v += __builtin_amdgcn_update_dpp(0, v, (ctrl), (rm), 0xf, (bc))
    // (a sum is not idempotent: lanes without a DPP source must add 0, hence old = 0)
    DPP_ADD_STEP(0xB1, 0xf, false);
    DPP_ADD_STEP(0x4E, 0xf, false);
    DPP_ADD_STEP(0x141, 0xf, false);
    DPP_ADD_STEP(0x140, 0xf, false);
    DPP_ADD_STEP(0x142, 0xa, false);
    DPP_ADD_STEP(0x143, 0xc, false);
#undef DPP_ADD_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- A2: one workgroup per (image, chunk of 256 priors), CQ threads per prior ---------------------------------------------
__global__ __launch_bounds__(CCH * CQ) void assign_compact2_kernel(
    const float* __restrict__ flat, const float* __restrict__ gt_boxes, const int32_t* __restrict__ gt_count,
    Levels L, int P, int Gmax, float radius, int nchunk, int32_t* __restrict__ gt_inds, int32_t* __restrict__ labels,
    float* __restrict__ max_overlaps, AssignScratch ws, const float* __restrict__ pre_scores,
    const float* __restrict__ pre_boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GT* s_gt = reinterpret_cast<GT*>(smem);
    constexpr int NWV = CCH * CQ / 64;                            // waves of the workgroup, 64 / CQ priors each
    __shared__ int s_wcnt[NWV];
    const int n = blockIdx.x / nchunk, c = blockIdx.x - n * nchunk;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = tid & (CQ - 1);
    const int G = max(0, min(gt_count[n], Gmax));      // (a negative count from an unvalidated data source is an empty image)
    const int p = c * CCH + tid / CQ;
    const bool lead = q == 0 && p < P;                            // the thread that decodes and stores prior p
    // the prior's head outputs go in flight before anything else (only valid priors use them)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 b = make_float2(0.f, 0.f);
    if (lead && !pre_scores) {
        const float* fl = flat + ((size_t)n * P + p) * 16;
        a = *reinterpret_cast<const float4*>(fl);
        b = *reinterpret_cast<const float2*>(fl + 4);
    }
    // (the first pass does not wait for gt_count: rows beyond G are padding of the [N, Gmax, 4] array and never read back)
    for (int g = tid; g < Gmax && (g < CCH * CQ || g < G); g += CCH * CQ) {
        const float* bx = gt_boxes + ((size_t)n * Gmax + g) * 4;
        s_gt[g] = GT{bx[0], bx[1], bx[2], bx[3]};
    }
    if (c == 0) {
        // the (image, GT) pairs of the batch as one dense list for B2: this image's pairs start at the number of GTs of the
        // images before it
        const int N = gridDim.x / nchunk;
        int part = 0;
        for (int i = tid; i < n; i += CCH * CQ) part += max(0, min(gt_count[i], Gmax));
        part = wave_sum_i_dpp(part);
        if (lane == 0) s_wcnt[wid] = part;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) base += s_wcnt[w];
        for (int g = tid; g < G; g += CCH * CQ) ws.items[1 + base + g] = ((uint32_t)n << 16) | (uint32_t)g;
        if (n == N - 1 && tid == 0) ws.items[0] = (uint32_t)(base + G);
    }
    __syncthreads();
    float px = 0.f, py = 0.f, s = 1.f;
    bool mine = false;
    if (p < P) {
        prior_of(L, p, px, py, s);
        const float cx = px + s * 0.5f, cy = py + s * 0.5f;
        for (int g = q; g < G; g += CQ) {
            const GT gt = s_gt[g];
            if (in_gt_box(cx, cy, gt) || in_gt_center(cx, cy, s, radius, gt)) {
                mine = true;
                break;
            }
        }
    }
    const unsigned long long any = __ballot(mine);
    const bool valid = lead && ((any >> (lane & ~(CQ - 1))) & ((1ull << CQ) - 1ull)) != 0ull;
    VRec rec;
    if (lead) {
        const float cx = px + s * 0.5f, cy = py + s * 0.5f;
        gt_inds[(size_t)n * P + p] = 0;
        max_overlaps[(size_t)n * P + p] = -INF_COST;
        if (labels) labels[(size_t)n * P + p] = -1;
        if (valid && pre_scores) {
            // stand-alone SimOTAAssigner.assign(): scores and decoded boxes are inputs
            const float4 bx = *reinterpret_cast<const float4*>(pre_boxes + ((size_t)n * P + p) * 4);
            rec.x1 = bx.x; rec.y1 = bx.y; rec.x2 = bx.z; rec.y2 = bx.w;
            rec.cls_cost = -fmaxf(logf(sqrtf(pre_scores[(size_t)n * P + p])), -100.0f);
            rec.cx = cx; rec.cy = cy; rec.s = s;
        } else if (valid) {
            // a = cls, dx, dy, dw ; b = dh, obj
            const float bx = a.y * s + px, by = a.z * s + py;
            const float bw = expf(a.w) * s, bh = expf(b.x) * s;
            rec.x1 = bx - bw / 2.0f;
            rec.y1 = by - bh / 2.0f;
            rec.x2 = bx + bw / 2.0f;
            rec.y2 = by + bh / 2.0f;
            const float score = sigmoidf_ref(a.x) * sigmoidf_ref(b.y);
            // F.binary_cross_entropy(sqrt(score), 1) = -max(log(sqrt(score)), -100)
            rec.cls_cost = -fmaxf(logf(sqrtf(score)), -100.0f);
            rec.cx = cx; rec.cy = cy; rec.s = s;
        }
    }
    const unsigned long long bal = __ballot(valid);
    if (lane == 0) s_wcnt[wid] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        const int cw = s_wcnt[w];
        off += w < wid ? cw : 0;
        total += cw;
    }
    if (valid) {
        const size_t slot = (size_t)n * P + c * CCH + off + __popcll(bal & ((1ull << lane) - 1ull));
        ws.vidx[slot] = (uint16_t)p;
        ws.cnt[slot] = 0;
        ws.mg[slot] = 0;
        ws.rec[slot] = rec;
    }
    if (tid == 0) ws.vc[(size_t)n * nchunk + c] = total;
}

// ---- B2: one workgroup (four waves) per (image, GT) pair, pairs taken from the dense list A2 wrote ------------------------
// Which pairs can matter.  For GT g and valid prior v let ov = the overlap of v's decoded box with g and both = "v's centre is
// in g's box AND in its centre region".  If ov == 0 then iou == +0 exactly (overlap / max(union, 1e-6)), so v adds nothing to
// the sum of the top IoUs, and if also !both its cost is (cls_cost * cls_w + c0 * iou_w) + 1e5 with the constant
// c0 = -log(0 + 1e-7): three operations, no division, no logarithm.  Priors with ov > 0 || both ("candidates": 13 % of the
// pairs at the bench batch, up to half of an image's valid priors for a large face) are compacted into a per-wave LDS list
// and get the full evaluation and both insertion networks, exactly as in B.  Of the others only the ONE cheapest
// (cost, slot) key is tracked (a running minimum per lane).  That is enough whenever the dynamic_k cheapest priors contain
// at most one non-candidate: the number of list entries below the cheapest non-candidate is counted, and if fewer than
// dynamic_k - 1 lie below it (never for dynamic_k = 1; otherwise only when a GT's overlapping priors are fewer than its
// dynamic k would have them -- rare) the workgroup repeats the walk with every prior treated as a candidate, which is
// kernel B.  Each lane visits its priors in ascending slot order, so the strict comparisons of the insertion networks keep
// "lowest prior index" among equal costs; every later merge compares full (cost, slot) keys, which are unique.
// The records of the image are staged through LDS in blocks of 512 (the next block's loads in flight under the current
// block's arithmetic); wave w takes records 128 w .. 128 w + 127 of a block.  Afterwards each wave extracts its own top
// lists with wave reductions and the four lists are merged from LDS.
template <int TOPK>
#ifdef B2_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(B2_WAVES_PER_EU, B2_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(TOPK_WAVES * 64) void assign_topk2_kernel(
    const float* __restrict__ gt_boxes, int P, int Gmax, int nchunk, float radius, int topk, CostW cw, AssignScratch ws) {
    constexpr int NT = TOPK_WAVES * 64, PER = RB / NT, WB = RB / TOPK_WAVES;      // WB: records of a block per wave
    static_assert(TOPK_WAVES * TOPK <= 64, "the cross-wave merges hold one list entry per lane");
    __shared__ int s_pre[257];                                    // prefix sums of the chunk counts (nchunk <= 256)
    __shared__ float4 s_rec[RB][2];                               // a record = (x1 y1 x2 y2 | cls_cost cx cy s)
    __shared__ uint16_t s_slot[RB];
    __shared__ uint16_t s_cand[TOPK_WAVES][WB];
    __shared__ float s_ti[TOPK_WAVES * TOPK];
    __shared__ unsigned long long s_key[TOPK_WAVES * TOPK];
    __shared__ unsigned long long s_ncm[TOPK_WAVES];
    __shared__ int s_below[TOPK_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nitems = (int)ws.items[0];
    // c0 must come out of the device's logf at run time, as it does inside cost_of(): the zero is opaque to the compiler
    float zero = 0.0f;
    asm volatile("" : "+v"(zero));
    const float c0w = -logf(zero + 1e-7f) * cw.iou_w;
    uint16_t* cand = s_cand[wid];
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {          // workgroup-uniform
        const uint32_t it = ws.items[1 + item];
        const int n = (int)(it >> 16), g = (int)(it & 0xffffu);
        __syncthreads();                                          // (the previous pair's reads of s_pre / the merge arrays)
        if (wid == 0) {
            int run = 0;
            for (int c0 = 0; c0 < nchunk; c0 += 64) {
                const int c = c0 + lane;
                const int x = c < nchunk ? ws.vc[(size_t)n * nchunk + c] : 0;
                int inc = x;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(inc, o, 64);
                    inc += lane >= o ? t : 0;
                }
                if (c < nchunk) s_pre[c + 1] = run + inc;
                run += __shfl(inc, 63, 64);
            }
            if (lane == 0) s_pre[0] = 0;
        }
        __syncthreads();
        const int V = s_pre[nchunk];
        if (V <= 0) continue;                                     // workgroup-uniform
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(ws.rec + (size_t)n * P);
        const float* bx = gt_boxes + ((size_t)n * Gmax + g) * 4;
        const GT gt{bx[0], bx[1], bx[2], bx[3]};
        float ti[TOPK];
        float tc[TOPK];
        int tv[TOPK];
        int dk = 1;
        unsigned long long ncm = ~0ull;                           // cheapest non-candidate of the pair
        for (int mode = 0; mode < 2; ++mode) {                    // 0: candidates only; 1: every prior (kernel B)
            unsigned long long nc_best = ~0ull;                   // cheapest non-candidate of this lane
#pragma unroll
            for (int i = 0; i < TOPK; ++i) {
                ti[i] = 0.0f;                                     // zero IoUs are never inserted and add nothing
                tc[i] = 3.0e38f;
                tv[i] = 0x7fffffff;
            }
            // staging: thread t brings records j = b0 + t and j + NT of the image's dense order; the chunk cursor of each
            // only moves forward
            static_assert(PER == 2, "two records per thread and block");
            int chunk0 = 0, chunk1 = 0, sslot0 = 0, sslot1 = 0;
            float4 sa0 = make_float4(0.f, 0.f, 0.f, 0.f), sb0 = sa0, sa1 = sa0, sb1 = sa0;
#define B2_FETCH(b0_)                                                                              \
            {                                                                                      \
                const int ja = (b0_) + tid, jb = ja + NT;                                          \
                if (ja < V) {                                                                      \
                    while (ja >= s_pre[chunk0 + 1]) ++chunk0;                                      \
                    sslot0 = (chunk0 << CCH_SHIFT) + (ja - s_pre[chunk0]);                         \
                    sa0 = rec[2 * sslot0];                                                         \
                    sb0 = rec[2 * sslot0 + 1];                                                     \
                }                                                                                  \
                if (jb < V) {                                                                      \
                    while (jb >= s_pre[chunk1 + 1]) ++chunk1;                                      \
                    sslot1 = (chunk1 << CCH_SHIFT) + (jb - s_pre[chunk1]);                         \
                    sa1 = rec[2 * sslot1];                                                         \
                    sb1 = rec[2 * sslot1 + 1];                                                     \
                }                                                                                  \
            }
            B2_FETCH(0)
            for (int b0 = 0; b0 < V; b0 += RB) {
                if (b0 + tid < V) {
                    s_rec[tid][0] = sa0;
                    s_rec[tid][1] = sb0;
                    s_slot[tid] = (uint16_t)sslot0;
                }
                if (b0 + tid + NT < V) {
                    s_rec[tid + NT][0] = sa1;
                    s_rec[tid + NT][1] = sb1;
                    s_slot[tid + NT] = (uint16_t)sslot1;
                }
                __syncthreads();
                if (b0 + RB < V) B2_FETCH(b0 + RB)                // in flight under this block's arithmetic
                const int nrec = V - b0 < RB ? V - b0 : RB;
                int nc = 0;
#pragma unroll
                for (int u = 0; u < WB / 64; ++u) {
                    const int i = wid * WB + u * 64 + lane;
                    bool is_cand = false;
                    if (i < nrec) {
                        const float4 ra = s_rec[i][0], rb = s_rec[i][1];
                        const float ov = overlap_of(ra.x, ra.y, ra.z, ra.w, gt);
                        const bool both = in_gt_box(rb.y, rb.z, gt) && in_gt_center(rb.y, rb.z, rb.w, radius, gt);
                        is_cand = mode != 0 || ov > 0.0f || both;
                        if (!is_cand) {
                            const float cc = (rb.x * cw.cls_w + c0w) + INF_COST;
                            const unsigned long long key = ((unsigned long long)ord(cc) << 32) | (uint32_t)s_slot[i];
                            nc_best = key < nc_best ? key : nc_best;
                        }
                    }
                    const unsigned long long bal = __ballot(is_cand);
                    if (is_cand) cand[nc + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
                    nc += __popcll(bal);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (int i = lane; i < nc; i += 64) {
                    const int li = cand[i];
                    const float4 ra = s_rec[li][0], rb = s_rec[li][1];
                    float iou = iou_of(ra.x, ra.y, ra.z, ra.w, gt);
                    const bool both = in_gt_box(rb.y, rb.z, gt) && in_gt_center(rb.y, rb.z, rb.w, radius, gt);
                    float c = cost_of(rb.x, iou, both, cw);
                    int cv = s_slot[li];
                    if (iou > ti[TOPK - 1]) {
#pragma unroll
                        for (int k = 0; k < TOPK; ++k) {
                            const bool sw = iou > ti[k];
                            const float t = ti[k];
                            ti[k] = sw ? iou : t;
                            iou = sw ? t : iou;
                        }
                    }
                    if (c < tc[TOPK - 1]) {
#pragma unroll
                        for (int k = 0; k < TOPK; ++k) {
                            const bool sw = c < tc[k];  // strict: equal costs keep ascending slot
                            const float t = tc[k];
                            const int u = tv[k];
                            tc[k] = sw ? c : t;
                            tv[k] = sw ? cv : u;
                            c = sw ? t : c;
                            cv = sw ? u : cv;
                        }
                    }
                }
                __syncthreads();                                  // every read of the block is done before it is overwritten
            }
#undef B2_FETCH
            // ---- this wave's largest IoUs (descending) and its cheapest non-candidate -> LDS
            const int K = V < topk ? V : topk;         // candidate_topk = min(self.candidate_topk, ious.size(0))
            for (int k = 0; k < TOPK; ++k) {
                float m = 0.0f;
                if (k < K) {
                    m = wave_max_f_dpp(ti[0]);
                    const unsigned long long who = __ballot(ti[0] == m);
                    const int winner = __ffsll((long long)who) - 1;
                    if (lane == winner && m > 0.0f) {
#pragma unroll
                        for (int i = 0; i < TOPK - 1; ++i) ti[i] = ti[i + 1];
                        ti[TOPK - 1] = 0.0f;
                    }
                }
                if (lane == 0) s_ti[wid * TOPK + k] = m;
                if (!(m > 0.0f)) {                     // zeros from here on
                    for (int k2 = k + 1; k2 < TOPK; ++k2)
                        if (lane == 0) s_ti[wid * TOPK + k2] = 0.0f;
                    break;
                }
            }
            {
                const unsigned long long ncw = wave_min_u64_dpp(nc_best);
                if (lane == 0) s_ncm[wid] = ncw;
            }
            __syncthreads();
            // ---- dynamic k: sum of the workgroup's largest IoUs in descending order (every wave computes the same)
            float x = lane < TOPK_WAVES * TOPK ? s_ti[lane] : 0.0f;
            float sum = 0.0f;
            for (int k = 0; k < K; ++k) {
                const float m = wave_max_f_dpp(x);
                if (!(m > 0.0f)) break;                // zeros add nothing
                const unsigned long long who = __ballot(x == m);
                if (lane == __ffsll((long long)who) - 1) x = 0.0f;
                sum = sum + m;
            }
            dk = (int)sum;  // .int(): truncation toward zero
            dk = dk < 1 ? 1 : dk;
            dk = dk > topk ? topk : dk;                // (a diverged model's inf / NaN sum: see kernel B)
            ncm = s_ncm[0];
#pragma unroll
            for (int w = 1; w < TOPK_WAVES; ++w) ncm = s_ncm[w] < ncm ? s_ncm[w] : ncm;
            bool again = false;
            if (ncm != ~0ull) {                        // (none: the lists hold every prior)
                // how many list entries lie below the cheapest non-candidate?  dk - 1 of them suffice
                int below = 0;
#pragma unroll
                for (int k = 0; k < TOPK; ++k) {
                    const unsigned long long key = ((unsigned long long)ord(tc[k]) << 32) | (uint32_t)tv[k];
                    below += (tv[k] != 0x7fffffff && key < ncm) ? 1 : 0;
                }
                below = wave_sum_i_dpp(below);
                if (lane == 0) s_below[wid] = below;
                __syncthreads();
                below = 0;
#pragma unroll
                for (int w = 0; w < TOPK_WAVES; ++w) below += s_below[w];
                again = below < dk - 1;                // two or more non-candidates would be selected: walk again, every prior in full
            }
            if (!again || mode == 1) break;            // workgroup-uniform
            __syncthreads();                           // (s_ti / s_ncm are rewritten by the second walk)
        }
        // ---- the cheapest non-candidate joins one list (wave 0, lane 0) in full (cost, slot) order: its slot may be below
        // equal costs there
        if (wid == 0 && lane == 0 && ncm != ~0ull) {
            const uint32_t oc = (uint32_t)(ncm >> 32);
            float c = __uint_as_float((oc & 0x80000000u) ? (oc & 0x7fffffffu) : ~oc);
            int cv = (int)(uint32_t)ncm;
#pragma unroll
            for (int k = 0; k < TOPK; ++k) {
                const bool sw = c < tc[k] || (c == tc[k] && cv < tv[k]);
                const float t = tc[k];
                const int u = tv[k];
                tc[k] = sw ? c : t;
                tv[k] = sw ? cv : u;
                c = sw ? t : c;
                cv = sw ? u : cv;
            }
        }
        // ---- this wave's dk cheapest (cost, slot) keys -> LDS; wave 0 merges the four lists
        for (int k = 0; k < TOPK; ++k) {
            unsigned long long kmin = ~0ull;
            if (k < dk) {
                const unsigned long long key = tv[0] != 0x7fffffff ? ((unsigned long long)ord(tc[0]) << 32) | (uint32_t)tv[0] : ~0ull;
                kmin = wave_min_u64_dpp(key);
                if (key == kmin && kmin != ~0ull) {
#pragma unroll
                    for (int i = 0; i < TOPK - 1; ++i) {
                        tc[i] = tc[i + 1];
                        tv[i] = tv[i + 1];
                    }
                    tc[TOPK - 1] = 3.0e38f;
                    tv[TOPK - 1] = 0x7fffffff;
                }
            }
            if (lane == 0) s_key[wid * TOPK + k] = kmin;
        }
        __syncthreads();
        if (wid == 0) {
            unsigned long long key = lane < TOPK_WAVES * TOPK ? s_key[lane] : ~0ull;
            int* cnt = ws.cnt + (size_t)n * P;
            uint16_t* mg = ws.mg + (size_t)n * P;
            for (int k = 0; k < dk; ++k) {
                const unsigned long long kmin = wave_min_u64_dpp(key);
                if (kmin == ~0ull) break;
                if (key == kmin) {
                    const int v = (int)(uint32_t)kmin;
                    atomicAdd(&cnt[v], 1);
                    mg[v] = (uint16_t)g;       // read back only where cnt ends at 1: then this is the only writer
                    key = ~0ull;
                }
            }
        }
    }
}

// ---- C2: conflicts -> argmin over all GTs with one GT per lane; outputs; per-image statistics (one workgroup per image) ----
__global__ __launch_bounds__(ASSIGN_THREADS) void assign_resolve2_kernel(
    const float* __restrict__ gt_boxes, const float* __restrict__ gt_kps, const int32_t* __restrict__ gt_labels,
    const int32_t* __restrict__ gt_count, int P, int Gmax, int nchunk, float radius, CostW cw,
    int32_t* __restrict__ gt_inds, int32_t* __restrict__ labels, float* __restrict__ max_overlaps,
    float* __restrict__ img_stats, AssignScratch ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GT* s_gt = reinterpret_cast<GT*>(smem);
    __shared__ float s_red[ASSIGN_WAVES][2];
    __shared__ int s_vc[256];
    __shared__ uint16_t s_conf[ASSIGN_THREADS], s_res[ASSIGN_THREADS];
    __shared__ int s_nconf;
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = max(0, min(gt_count[n], Gmax));      // (a negative count from an unvalidated data source is an empty image)
    const VRec* __restrict__ rec = ws.rec + (size_t)n * P;
    const int* cnt = ws.cnt + (size_t)n * P;
    const uint16_t* vidx = ws.vidx + (size_t)n * P;
    const uint16_t* mg = ws.mg + (size_t)n * P;
    for (int g = tid; g < G; g += ASSIGN_THREADS) {
        const float* b = gt_boxes + ((size_t)n * Gmax + g) * 4;
        s_gt[g] = GT{b[0], b[1], b[2], b[3]};
    }
    for (int c = tid; c < nchunk; c += ASSIGN_THREADS) s_vc[c] = ws.vc[(size_t)n * nchunk + c];
    float npos = 0.0f, wsum = 0.0f;
    const int nslots = G > 0 ? nchunk * CCH : 0;
    for (int base = 0; base < nslots; base += ASSIGN_THREADS) {       // workgroup-uniform trip count
        if (tid == 0) s_nconf = 0;
        __syncthreads();                                              // (also: s_gt / s_vc of the prologue)
        const int s = base + tid;
        const bool live = s < nslots && (s & (CCH - 1)) < s_vc[s >> CCH_SHIFT];
        const int c = live ? cnt[s] : 0;
        int myk = 0;
        if (c > 1) {
            myk = atomicAdd(&s_nconf, 1);
            s_conf[myk] = (uint16_t)s;
        }
        __syncthreads();
        const int nconf = s_nconf;
        for (int k = wid; k < nconf; k += ASSIGN_WAVES) {
            const VRec r = rec[s_conf[k]];                            // one address per wave
            unsigned long long best = ~0ull;
            for (int j = lane; j < G; j += 64) {
                const GT gt = s_gt[j];
                const float iou = iou_of(r.x1, r.y1, r.x2, r.y2, gt);
                const bool both = in_gt_box(r.cx, r.cy, gt) && in_gt_center(r.cx, r.cy, r.s, radius, gt);
                const float cj = cost_of(r.cls_cost, iou, both, cw);
                const unsigned long long key = ((unsigned long long)ord(cj) << 32) | (uint32_t)j;   // first minimum
                best = key < best ? key : best;
            }
            best = wave_min_u64_dpp(best);
            if (lane == 0) s_res[k] = (uint16_t)(uint32_t)best;
        }
        __syncthreads();
        if (c > 0) {
            const VRec r = rec[s];
            const int g = c > 1 ? (int)s_res[myk] : (int)mg[s];
            const float iou = iou_of(r.x1, r.y1, r.x2, r.y2, s_gt[g]);
            const int p = vidx[s];
            gt_inds[(size_t)n * P + p] = g + 1;
            max_overlaps[(size_t)n * P + p] = iou;
            if (labels) labels[(size_t)n * P + p] = gt_labels ? gt_labels[(size_t)n * Gmax + g] : 0;
            const float* kp = gt_kps + ((size_t)n * Gmax + g) * 15;
            // torch.mean over the 5 visibility flags
            const float w = ((((kp[2] + kp[5]) + kp[8]) + kp[11]) + kp[14]) / 5.0f;
            npos += 1.0f;
            wsum += w;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        npos += __shfl_xor(npos, o, 64);
        wsum += __shfl_xor(wsum, o, 64);
    }
    if (lane == 0) {
        s_red[wid][0] = npos;
        s_red[wid][1] = wsum;
    }
    __syncthreads();
    if (tid == 0) {
        float a = 0.0f, b = 0.0f;
        for (int w = 0; w < ASSIGN_WAVES; ++w) {
            a += s_red[w][0];
            b += s_red[w][1];
        }
        img_stats[n * 2 + 0] = a;
        img_stats[n * 2 + 1] = b;
    }
}

__global__ void loss_norm_kernel(const float* __restrict__ img_stats, int N, float inv_world,
                                 float* __restrict__ norm) {
    __shared__ float s[2][256];
    float a = 0.0f, b = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        a += img_stats[2 * i];
        b += img_stats[2 * i + 1];
    }
    s[0][threadIdx.x] = a;
    s[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) {
            s[0][threadIdx.x] += s[0][threadIdx.x + o];
            s[1][threadIdx.x] += s[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        norm[0] = s[0][0] * inv_world;  // tensor.div_(world) before the all-reduce SUM
        norm[1] = s[1][0];
        norm[2] = s[0][0];              // rank-local positives (logging)
    }
}

// ---- forward-mode dual numbers with 4 partials: the box loss is written once, in the
// reference's evaluation order, and its gradient w.r.t. (dx,dy,dw,dh) falls out.
struct D4 {
    float v, d[4];
};
__device__ __forceinline__ D4 mk(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) {
    D4 r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator-(D4 a, D4 b) {
    D4 r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
    D4 r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
    D4 r; r.v = a.v / b.v;
    const float inv = 1.0f / b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
// torch.min / torch.max of two tensors split the gradient EVENLY between equal arguments
// (minimum / maximum backward: grad * (a < b) + grad / 2 * (a == b)); a predicted edge that coincides
// with its GT edge to the last bit does occur (one prior in ~3000 positives of a crowded 640 x 640 batch,
// tests/test_loss_step_gpu.py::test_loss_step_crowded_images_vs_oracle).  clamp(min=0) passes the
// gradient at the boundary (grad * (x >= 0)).
__device__ __forceinline__ D4 dtie(D4 a, D4 b) {
    D4 r; r.v = a.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = 0.5f * a.d[i] + 0.5f * b.d[i];
    return r;
}
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v == b.v ? dtie(a, b) : (a.v < b.v) ? a : b; }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v == b.v ? dtie(a, b) : (a.v > b.v) ? a : b; }
__device__ __forceinline__ D4 dclamp0(D4 a) { return a.v >= 0.0f ? a : mk(0.0f); }
__device__ __forceinline__ D4 dscale(D4 a, float k) {
    D4 r; r.v = a.v * k;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * k;
    return r;
}
__device__ __forceinline__ D4 dadd(D4 a, float k) { a.v = a.v + k; return a; }

// eiou_loss, mmdet/models/losses/iou_loss.py:194-227
__device__ D4 eiou(D4 px1, D4 py1, D4 px2, D4 py2, float tx1f, float ty1f, float tx2f, float ty2f,
                   float sp, float eps) {
    const D4 tx1 = mk(tx1f), ty1 = mk(ty1f), tx2 = mk(tx2f), ty2 = mk(ty2f);
    const D4 ex1 = dmin(px1, tx1), ey1 = dmin(py1, ty1);
    const D4 ix1 = dmax(px1, tx1), iy1 = dmax(py1, ty1);
    const D4 ix2 = dmin(px2, tx2), iy2 = dmin(py2, ty2);
    const D4 xmin = dmin(ix1, ix2), ymin = dmin(iy1, iy2);
    const D4 xmax = dmax(ix1, ix2), ymax = dmax(iy1, iy2);
    const D4 inter = (((ix2 - ex1) * (iy2 - ey1) + (xmin - ex1) * (ymin - ey1)) -
                      (ix1 - ex1) * (ymax - ey1)) - (xmax - ex1) * (iy1 - ey1);
    const D4 uni = dadd(((px2 - px1) * (py2 - py1) + (tx2 - tx1) * (ty2 - ty1)) - inter, eps);
    const D4 x = mk(1.0f) - inter / uni;
    if (x.v < sp) return dscale(dscale(x * x, 0.5f), 1.0f / sp);   // 0.5*x^2/sp
    return dadd(x, -0.5f * sp);
}

// diou_loss, mmdet/models/losses/iou_loss.py:137-172
__device__ D4 diou(D4 px1, D4 py1, D4 px2, D4 py2, float tx1f, float ty1f, float tx2f, float ty2f,
                   float eps) {
    const D4 tx1 = mk(tx1f), ty1 = mk(ty1f), tx2 = mk(tx2f), ty2 = mk(ty2f);
    const D4 w = dclamp0(dmin(px2, tx2) - dmax(px1, tx1));
    const D4 h = dclamp0(dmin(py2, ty2) - dmax(py1, ty1));
    const D4 overlap = w * h;
    const D4 ap = (px2 - px1) * (py2 - py1);
    const D4 ag = (tx2 - tx1) * (ty2 - ty1);
    const D4 uni = dadd((ap + ag) - overlap, eps);
    const D4 ious = overlap / uni;
    const D4 cw = dclamp0(dmax(px2, tx2) - dmin(px1, tx1));
    const D4 ch = dclamp0(dmax(py2, ty2) - dmin(py1, ty1));
    const D4 c2 = dadd(cw * cw + ch * ch, eps);
    const D4 a = (tx1 + tx2) - (px1 + px2);
    const D4 b = (ty1 + ty2) - (py1 + py2);
    const D4 rho2 = dscale(a * a, 0.25f) + dscale(b * b, 0.25f);
    return mk(1.0f) - (ious - rho2 / c2);
}

// ---- the other members of the reference's IoU-loss family (round 5: YuNet_Head's own default is IoULoss(mode='square'),
// yunet_head.py:59-64; the registry also offers GIoULoss / CIoULoss) -------------------------------------------------
__device__ __forceinline__ D4 dlog(D4 a) {
    D4 r; r.v = logf(a.v);
    const float inv = 1.0f / a.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * inv;
    return r;
}
__device__ __forceinline__ D4 datan(D4 a) {
    D4 r; r.v = atanf(a.v);
    const float k = 1.0f / (1.0f + a.v * a.v);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * k;
    return r;
}
// bbox_overlaps(pred, target, is_aligned=True, eps) of iou2d_calculator.py:214-259: IoU and the union it used
__device__ __forceinline__ D4 aligned_iou(D4 px1, D4 py1, D4 px2, D4 py2, D4 tx1, D4 ty1, D4 tx2, D4 ty2, float eps,
                                          D4* uni_out) {
    const D4 w = dclamp0(dmin(px2, tx2) - dmax(px1, tx1));
    const D4 h = dclamp0(dmin(py2, ty2) - dmax(py1, ty1));
    const D4 overlap = w * h;
    const D4 a1 = (px2 - px1) * (py2 - py1), a2 = (tx2 - tx1) * (ty2 - ty1);
    const D4 uni = dmax((a1 + a2) - overlap, mk(eps));          // torch.max(union, eps)
    *uni_out = uni;
    return overlap / uni;
}
// iou_loss, iou_loss.py:14-50: mode 0 linear | 1 square | 2 log; `eps` clamps the IoU from below
__device__ D4 iou_family(D4 px1, D4 py1, D4 px2, D4 py2, float tx1f, float ty1f, float tx2f, float ty2f, int mode, float eps) {
    D4 uni;
    D4 ious = aligned_iou(px1, py1, px2, py2, mk(tx1f), mk(ty1f), mk(tx2f), mk(ty2f), 1e-6f, &uni);
    if (!(ious.v >= eps)) ious = mk(eps);                       // .clamp(min=eps): the gradient passes at the boundary
    if (mode == 0) return mk(1.0f) - ious;
    if (mode == 1) return mk(1.0f) - ious * ious;
    D4 l = dlog(ious);
    return mk(0.0f) - l;
}
// giou_loss, iou_loss.py:103-120 (bbox_overlaps mode 'giou', iou2d_calculator.py:248-259)
__device__ D4 giou(D4 px1, D4 py1, D4 px2, D4 py2, float tx1f, float ty1f, float tx2f, float ty2f, float eps) {
    const D4 tx1 = mk(tx1f), ty1 = mk(ty1f), tx2 = mk(tx2f), ty2 = mk(ty2f);
    D4 uni;
    const D4 ious = aligned_iou(px1, py1, px2, py2, tx1, ty1, tx2, ty2, eps, &uni);
    const D4 ew = dclamp0(dmax(px2, tx2) - dmin(px1, tx1)), eh = dclamp0(dmax(py2, ty2) - dmin(py1, ty1));
    const D4 ea = dmax(ew * eh, mk(eps));
    const D4 gious = ious - (ea - uni) / ea;
    return mk(1.0f) - gious;
}
// ciou_loss, iou_loss.py:230-293 (alpha is computed under torch.no_grad)
__device__ D4 ciou(D4 px1, D4 py1, D4 px2, D4 py2, float tx1f, float ty1f, float tx2f, float ty2f, float eps) {
    const D4 tx1 = mk(tx1f), ty1 = mk(ty1f), tx2 = mk(tx2f), ty2 = mk(ty2f);
    const D4 w = dclamp0(dmin(px2, tx2) - dmax(px1, tx1));
    const D4 h = dclamp0(dmin(py2, ty2) - dmax(py1, ty1));
    const D4 overlap = w * h;
    const D4 ap = (px2 - px1) * (py2 - py1), ag = (tx2 - tx1) * (ty2 - ty1);
    const D4 uni = dadd((ap + ag) - overlap, eps);
    const D4 ious = overlap / uni;
    const D4 cw = dclamp0(dmax(px2, tx2) - dmin(px1, tx1)), ch = dclamp0(dmax(py2, ty2) - dmin(py1, ty1));
    const D4 c2 = dadd(cw * cw + ch * ch, eps);
    const D4 w1 = px2 - px1, h1 = dadd(py2 - py1, eps);
    const D4 w2 = tx2 - tx1, h2 = dadd(ty2 - ty1, eps);
    const D4 a = (tx1 + tx2) - (px1 + px2), b = (ty1 + ty2) - (py1 + py2);
    const D4 rho2 = dscale(a * a, 0.25f) + dscale(b * b, 0.25f);
    const D4 da = datan(w2 / h2) - datan(w1 / h1);
    const D4 v = dscale(da * da, 0.40528473456935109f);          // 4 / pi^2
    const float alpha = ious.v > 0.5f ? v.v / ((1.0f - ious.v) + v.v) : 0.0f;
    const D4 cious = ious - (rho2 / c2 + dscale(v, alpha));
    if (cious.v < -1.0f) return mk(2.0f);                       // 1 - clamp(cious, -1, 1): constant outside the range
    if (cious.v > 1.0f) return mk(0.0f);
    return mk(1.0f) - cious;
}

// log_sigmoid(x) = min(x,0) - log1p(exp(-|x|));  BCEWithLogits = (1-t)*x - log_sigmoid(x)
__device__ __forceinline__ float bce_logits(float x, float t) {
    const float ls = fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
    return (1.0f - t) * x - ls;
}

#define LOSS_THREADS 256

__global__ __launch_bounds__(LOSS_THREADS) void loss_kernel(
    const float* __restrict__ flat, const int32_t* __restrict__ gt_inds,
    const float* __restrict__ max_overlaps, const float* __restrict__ gt_boxes,
    const float* __restrict__ gt_kps, Levels L, YunetLossCfg cfg, const float* __restrict__ norm,
    int N, int P, int Gmax, float* __restrict__ dflat, float* __restrict__ partials) {
    // cfg.defer_num_total: norm[0] (the all-reduced positives, possibly still in flight on another stream) is NOT
    // read: the cls / bbox / obj terms and their gradients leave this kernel un-normalised (x 1.0 is exact) and
    // loss_finalize_kernel applies 1 / max(num_total, 1) to the losses and hands it to the head backward as dy_scale
    const float num_total = cfg.defer_num_total ? 1.0f : fmaxf(norm[0], 1.0f);
    const float inv_total = 1.0f / num_total;
    const float kps_den = norm[1] + 1.1920928955078125e-07f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const long long total = (long long)N * P;
    for (long long e = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * LOSS_THREADS) {
        const int n = (int)(e / P), p = (int)(e - (long long)n * P);
        const float4* src = reinterpret_cast<const float4*>(flat + e * 16);
        float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        float o[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.0f;
        const int gi = gt_inds[e];
        const float t_obj = gi > 0 ? 1.0f : 0.0f;
        const float xo = q1.y;
        acc[2] += bce_logits(xo, t_obj);
        o[5] = cfg.w_obj * (sigmoidf_ref(xo) - t_obj) * inv_total;
        if (gi > 0) {
            const int g = gi - 1;
            float px, py, s;
            prior_of(L, p, px, py, s);
            // cls: BCE with IoU soft target, positives only
            const float t = max_overlaps[e];
            acc[0] += bce_logits(q0.x, t);
            o[0] = cfg.w_cls * (sigmoidf_ref(q0.x) - t) * inv_total;
            // box: decode as duals w.r.t. (dx, dy, dw, dh)
            D4 ddx = mk(q0.y), ddy = mk(q0.z), ddw = mk(q0.w), ddh = mk(q1.x);
            ddx.d[0] = 1.f; ddy.d[1] = 1.f; ddw.d[2] = 1.f; ddh.d[3] = 1.f;
            const D4 cx = dadd(dscale(ddx, s), px), cy = dadd(dscale(ddy, s), py);
            D4 ew = mk(expf(ddw.v)), eh = mk(expf(ddh.v));
            ew.d[2] = ew.v; eh.d[3] = eh.v;
            const D4 bw = dscale(ew, s), bh = dscale(eh, s);
            const D4 hx = dscale(bw, 0.5f), hy = dscale(bh, 0.5f);   // w/2 is exact
            const D4 x1 = cx - hx, y1 = cy - hy, x2 = cx + hx, y2 = cy + hy;
            const float* tb = gt_boxes + ((size_t)n * Gmax + g) * 4;
            D4 lb;
            switch (cfg.box_loss) {      // uniform over the launch
                case YUNET_BOX_EIOU: lb = eiou(x1, y1, x2, y2, tb[0], tb[1], tb[2], tb[3], cfg.smooth_point, cfg.box_eps); break;
                case YUNET_BOX_DIOU: lb = diou(x1, y1, x2, y2, tb[0], tb[1], tb[2], tb[3], cfg.box_eps); break;
                case YUNET_BOX_GIOU: lb = giou(x1, y1, x2, y2, tb[0], tb[1], tb[2], tb[3], cfg.box_eps); break;
                case YUNET_BOX_CIOU: lb = ciou(x1, y1, x2, y2, tb[0], tb[1], tb[2], tb[3], cfg.box_eps); break;
                default: lb = iou_family(x1, y1, x2, y2, tb[0], tb[1], tb[2], tb[3], cfg.box_loss - YUNET_BOX_IOU_LINEAR, cfg.box_eps);
            }
            acc[1] += lb.v;
            // (d * w) * 1/num_total, in this order: with the deferred normaliser the last factor is applied by the head
            // backward (dy_scale) and the product must round the same way
            const float kb = cfg.w_box;
            o[1] = lb.d[0] * kb * inv_total; o[2] = lb.d[1] * kb * inv_total;
            o[3] = lb.d[2] * kb * inv_total; o[4] = lb.d[3] * kb * inv_total;
            // kps: smooth-L1 on (kps - prior_xy)/stride, weight = mean visibility
            const float* kp = gt_kps + ((size_t)n * Gmax + g) * 15;
            const float w = ((((kp[2] + kp[5]) + kp[8]) + kp[11]) + kp[14]) / 5.0f;
            const float pr[10] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            const float kk = cfg.w_kps / kps_den;
            float lk = 0.0f;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float tgt = (kp[(j >> 1) * 3 + (j & 1)] - ((j & 1) ? py : px)) / s;
                const float d = pr[j] - tgt;
                const float ad = fabsf(d);
                const bool quad = ad < cfg.kps_beta;
                lk += (quad ? 0.5f * ad * ad / cfg.kps_beta : ad - 0.5f * cfg.kps_beta) * w;
                const float gsl = quad ? d / cfg.kps_beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                o[6 + j] = gsl * w * kk;
            }
            acc[3] += lk;
        }
        float4* dst = reinterpret_cast<float4*>(dflat + e * 16);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        dst[2] = make_float4(o[8], o[9], o[10], o[11]);
        dst[3] = make_float4(o[12], o[13], o[14], o[15]);
    }
    // deterministic block reduction -> one partial row per block
    __shared__ float s_p[LOSS_THREADS / 64][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1) v += __shfl_xor(v, o2, 64);
        if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = 0.0f;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) v += s_p[w][threadIdx.x];
        // loss_i = weight * sum / normaliser
        const float wgt = threadIdx.x == 0 ? cfg.w_cls : threadIdx.x == 1 ? cfg.w_box
                          : threadIdx.x == 2 ? cfg.w_obj : cfg.w_kps;
        const float den = threadIdx.x == 3 ? kps_den : num_total;
        partials[blockIdx.x * 4 + threadIdx.x] = wgt * v / den;
    }
}

__global__ void loss_finalize_kernel(const float* __restrict__ partials, int blocks,
                                     float* __restrict__ losses, float* __restrict__ mirror,
                                     const float* __restrict__ num_total, float* __restrict__ dy_norm) {
    // 4 waves, one per loss term; fp64 accumulation in a fixed order
    __shared__ float s_l[4];
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v = 0.0;
    {
        // eight loads in flight, added in row order (a plain `v += p[...]` loop waits out every load before the next
        // is issued: up to 32 round trips per lane between the loss kernel and the first backward kernel)
        int b = lane;
        for (; b + 64 * 7 < blocks; b += 64 * 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = partials[(b + 64 * u) * 4 + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += (double)x[u];
        }
        for (; b < blocks; b += 64) v += (double)partials[b * 4 + k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    // deferred normaliser (yunet_loss with cfg.defer_num_total): cls, bbox and obj arrive un-normalised
    const float inv = num_total ? 1.0f / fmaxf(num_total[0], 1.0f) : 1.0f;
    if (lane == 0) s_l[k] = (num_total && k < 3) ? (float)v * inv : (float)v;
    if (dy_norm && threadIdx.x < 16) dy_norm[threadIdx.x] = threadIdx.x < 6 ? inv : 1.0f;   // cls | box x4 | obj | kps x10
    __syncthreads();
    // losses[4] = the total the reference builds in _parse_losses (base.py:206-209): a python sum()
    // over the dict in insertion order, i.e. ((cls + bbox) + obj) + kps in fp32
    if (threadIdx.x < 5) {
        const float tot = ((s_l[0] + s_l[1]) + s_l[2]) + s_l[3];
        const float o = threadIdx.x < 4 ? s_l[threadIdx.x] : tot;
        losses[threadIdx.x] = o;
        if (mirror) mirror[threadIdx.x] = o;
    }
}

}  // namespace

extern "C" int yunet_loss_blocks(int N, int P) {
    long long total = (long long)N * P;
    long long b = (total + LOSS_THREADS - 1) / LOSS_THREADS;
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

extern "C" int yunet_assign_cfg(const float* flat, const float* pre_scores, const float* pre_boxes,
                                const float* gt_boxes, const float* gt_kps,
                                const int32_t* gt_labels, const int32_t* gt_count,
                                const YunetLevels* lv, int N, int P, int Gmax, const YunetAssignCfg* cfg,
                                int32_t* gt_inds, int32_t* labels, float* max_overlaps,
                                float* img_stats, float* scratch, void* stream) {
    if (!lv || lv->num_levels < 1 || lv->num_levels > YUNET_MAX_LEVELS || P > 65535 || N < 1 ||
        Gmax < 1 || !cfg || cfg->candidate_topk < 1 || cfg->candidate_topk > TOPK_MAX)
        return YUNET_EINVAL;
    const float center_radius = cfg->center_radius;
    const int topk = cfg->candidate_topk;
    const CostW cw{cfg->iou_weight, cfg->cls_weight};
    Levels L = make_levels(lv);
    if (L.base[YUNET_MAX_LEVELS] != P) return YUNET_EINVAL;
    const size_t lds = (size_t)Gmax * sizeof(GT);
    if (lds > 64 * 1024) return YUNET_EINVAL;
    const AssignScratch ws = assign_scratch(scratch, N, P);
    hipStream_t st = (hipStream_t)stream;
    // the round-5 launches keep the batch's (image, GT) pair list in the scratch words the per-image launches use for V:
    // 1 + N * Gmax entries must fit into N * P words, and an entry packs image << 16 | GT
    if (yunet_option_assign_v2() && Gmax < P && N <= 65535 && Gmax <= 65535) {
        const int nchunk = (P + CCH - 1) / CCH;                   // <= 256 (P <= 65535)
        hipLaunchKernelGGL(assign_compact2_kernel, dim3(N * nchunk), dim3(CCH * CQ), lds, st, flat, gt_boxes, gt_count, L, P,
                           Gmax, center_radius, nchunk, gt_inds, labels, max_overlaps, ws, pre_scores, pre_boxes);
        const long long pairs = (long long)N * Gmax;              // upper bound of the pair list; a workgroup strides over it
        const int tgrid = (int)(pairs < 4096 ? pairs : 4096);
        if (topk <= 10)
            hipLaunchKernelGGL(assign_topk2_kernel<10>, dim3(tgrid), dim3(TOPK_WAVES * 64), 0, st, gt_boxes, P, Gmax, nchunk,
                               center_radius, topk, cw, ws);
        else
            hipLaunchKernelGGL(assign_topk2_kernel<TOPK_MAX>, dim3(tgrid), dim3(TOPK_WAVES * 64), 0, st, gt_boxes, P, Gmax, nchunk,
                               center_radius, topk, cw, ws);
        hipLaunchKernelGGL(assign_resolve2_kernel, dim3(N), dim3(ASSIGN_THREADS), lds, st, gt_boxes, gt_kps, gt_labels,
                           gt_count, P, Gmax, nchunk, center_radius, cw, gt_inds, labels, max_overlaps, img_stats, ws);
        return -(int)hipGetLastError();
    }
    hipLaunchKernelGGL(assign_compact_kernel, dim3(N), dim3(ASSIGN_THREADS), lds, st, flat, gt_boxes, gt_count, L, P,
                       Gmax, center_radius, gt_inds, labels, max_overlaps, ws, pre_scores, pre_boxes);
    const int gblocks = (Gmax + TOPK_WAVES - 1) / TOPK_WAVES;
    if (topk <= 10)
        hipLaunchKernelGGL(assign_topk_kernel<10>, dim3(N * gblocks), dim3(TOPK_WAVES * 64), 0, st, gt_boxes, gt_count, P,
                           Gmax, gblocks, center_radius, topk, cw, ws);
    else
        hipLaunchKernelGGL(assign_topk_kernel<TOPK_MAX>, dim3(N * gblocks), dim3(TOPK_WAVES * 64), 0, st, gt_boxes, gt_count,
                           P, Gmax, gblocks, center_radius, topk, cw, ws);
    hipLaunchKernelGGL(assign_resolve_kernel, dim3(N), dim3(ASSIGN_THREADS), lds, st, gt_boxes, gt_kps, gt_labels,
                       gt_count, P, Gmax, center_radius, cw, gt_inds, labels, max_overlaps, img_stats, ws);
    return -(int)hipGetLastError();
}

extern "C" int yunet_assign_ex(const float* flat, const float* pre_scores, const float* pre_boxes,
                               const float* gt_boxes, const float* gt_kps,
                               const int32_t* gt_labels, const int32_t* gt_count,
                               const YunetLevels* lv, int N, int P, int Gmax, float center_radius,
                               int32_t* gt_inds, int32_t* labels, float* max_overlaps,
                               float* img_stats, float* scratch, void* stream) {
    const YunetAssignCfg cfg{center_radius, 10, 3.0f, 1.0f};      // SimOTAAssigner's defaults (configs/yunet_*.py)
    return yunet_assign_cfg(flat, pre_scores, pre_boxes, gt_boxes, gt_kps, gt_labels, gt_count, lv, N, P, Gmax, &cfg,
                            gt_inds, labels, max_overlaps, img_stats, scratch, stream);
}

extern "C" int yunet_assign(const float* flat, const float* gt_boxes, const float* gt_kps,
                            const int32_t* gt_labels, const int32_t* gt_count,
                            const YunetLevels* lv, int N, int P, int Gmax, float center_radius,
                            int32_t* gt_inds, int32_t* labels, float* max_overlaps,
                            float* img_stats, float* scratch, void* stream) {
    return yunet_assign_ex(flat, nullptr, nullptr, gt_boxes, gt_kps, gt_labels, gt_count, lv, N, P,
                           Gmax, center_radius, gt_inds, labels, max_overlaps, img_stats, scratch,
                           stream);
}

extern "C" int yunet_loss_norm(const float* img_stats, int N, float inv_world, float* norm,
                               void* stream) {
    hipLaunchKernelGGL(loss_norm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, img_stats, N,
                       inv_world, norm);
    return -(int)hipGetLastError();
}

extern "C" int yunet_loss(const float* flat, const int32_t* gt_inds, const float* max_overlaps,
                          const float* gt_boxes, const float* gt_kps, const YunetLevels* lv,
                          const YunetLossCfg* cfg, const float* norm, int N, int P, int Gmax,
                          float* dflat, float* partials, int blocks, void* stream) {
    if (!lv || !cfg || blocks < 1) return YUNET_EINVAL;
    Levels L = make_levels(lv);
    if (L.base[YUNET_MAX_LEVELS] != P) return YUNET_EINVAL;
    hipLaunchKernelGGL(loss_kernel, dim3(blocks), dim3(LOSS_THREADS), 0, (hipStream_t)stream, flat,
                       gt_inds, max_overlaps, gt_boxes, gt_kps, L, *cfg, norm, N, P, Gmax, dflat,
                       partials);
    return -(int)hipGetLastError();
}

extern "C" int yunet_loss_finalize_ex(const float* partials, int blocks, float* losses, float* mirror,
                                      const float* num_total, float* dy_norm, void* stream) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials,
                       blocks, losses, mirror, num_total, dy_norm);
    return -(int)hipGetLastError();
}

extern "C" int yunet_loss_finalize(const float* partials, int blocks, float* losses, float* mirror,
                                   void* stream) {
    return yunet_loss_finalize_ex(partials, blocks, losses, mirror, nullptr, nullptr, stream);
}
