// detect.hip -- YuNet_Head.get_bboxes on the device:
//   priors + sigmoid scores + score threshold + box decode + greedy NMS, one image per workgroup.
// Reference: mmdet/models/dense_heads/yunet_head.py:290-416 (get_bboxes, _bbox_decode,
// _bboxes_nms) and mmcv.ops.batched_nms / nms (single class: plain greedy NMS, IoU with
// offset 0, suppression when IoU > iou_threshold, survivors in descending score order).
//
//   flat [N,P,16] = cls | dx dy dw dh | obj | 10 kps (raw head outputs, eval-mode forward)
//   score = sigmoid(cls) * sigmoid(obj); candidates: score >= score_thr
//   keys (score bits, ~prior index) sorted descending in LDS (bitonic), so equal scores keep the
//   lower prior index first; sorted boxes live in LDS (first BOX_CAP) and a global scratch;
//   a suppressed candidate's key is zeroed in place.
// Compiled with -ffp-contract=off (same single-rounding arithmetic as the torch reference).
#include "common.h"
#include "levels.h"

#define DET_THREADS 1024
#define DET_LDS_KEYS 16384       // candidates sorted in LDS: 8 B x 16384 = 128 KB
#define DET_BOX_CAP 1024         // sorted boxes cached in LDS (16 KB)

namespace {

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// mmcv nms (offset 0): inter / (area_a + area_b - inter)
__device__ __forceinline__ float nms_iou(const float4 a, const float4 b) {
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float width = fmaxf(right - left, 0.0f), height = fmaxf(bottom - top, 0.0f);
    const float inter = width * height;
    const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
    return inter / (sa + sb - inter);
}

// Sort (bitonic, descending) + decode + greedy NMS over `keys` [padn], K real candidates.  Inlined
// twice: with the keys in LDS (the usual case) and with the keys in the global scratch (an image with
// more than DET_LDS_KEYS candidates above the score threshold) -- same code, the address space of
// `keys` is known at each call site.
// Candidate sources: head outputs of one image (decode on the fly) or a plain list of boxes + scores.
struct FlatSrc {
    const float* f;          // [P,16] raw head outputs of this image
    Levels L;
    __device__ __forceinline__ bool score(int p, float thr, float& sc) const {
        sc = sigmoid_ref(f[(size_t)p * 16 + 0]) * sigmoid_ref(f[(size_t)p * 16 + 5]);
        return sc >= thr;
    }
    __device__ __forceinline__ float4 box(int p) const {
        float px, py, s;
        prior_of(L, p, px, py, s);
        const float* q = f + (size_t)p * 16;
        const float cx = q[1] * s + px, cy = q[2] * s + py;
        const float w = expf(q[3]) * s, h = expf(q[4]) * s;
        return make_float4(cx - w / 2.0f, cy - h / 2.0f, cx + w / 2.0f, cy + h / 2.0f);
    }
    __device__ __forceinline__ void extra(int p, int o, float* kn, int32_t*) const {
        if (!kn) return;                              // _kps_decode (yunet_head.py:388-393)
        float px, py, s;
        prior_of(L, p, px, py, s);
        const float* q = f + (size_t)p * 16 + 6;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            kn[o * 10 + 2 * t] = q[2 * t] * s + px;
            kn[o * 10 + 2 * t + 1] = q[2 * t + 1] * s + py;
        }
    }
};
struct BoxSrc {
    const float* boxes;      // [K,4]
    const float* scores;     // [K]
    __device__ __forceinline__ bool score(int p, float thr, float& sc) const {
        sc = scores[p];
        return sc >= thr;
    }
    __device__ __forceinline__ float4 box(int p) const {
        return *reinterpret_cast<const float4*>(boxes + (size_t)p * 4);
    }
    __device__ __forceinline__ void extra(int p, int o, float*, int32_t* keep) const {
        if (keep) keep[o] = p;
    }
};

// Order-preserving map of an fp32 score onto an unsigned key (and back): raw float bits are monotone only
// for non-negative values -- a negative score (yunet_nms accepts any score_thr, default -inf) has the sign
// bit set and would sort above every positive one, in reversed order.
__device__ __forceinline__ unsigned score_key(float s) {
    const unsigned u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

template <typename KeyPtr, typename Src>
__device__ __forceinline__ void sort_decode_nms(KeyPtr keys, int padn, int K, const Src& src, float iou_thr,
                                                int max_out, float4* __restrict__ gbox, float4* sbox,
                                                float* __restrict__ dn, float* __restrict__ kn,
                                                int32_t* __restrict__ keep_n, int32_t* __restrict__ count_n) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= padn; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < padn; i += DET_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < b : a > b) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // ---- decode the candidates in sorted order (yunet_head.py:376-386) -------------------------------
    for (int i = tid; i < K; i += DET_THREADS) {
        const int p = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const float4 b = src.box(p);
        gbox[i] = b;
        if (i < DET_BOX_CAP) sbox[i] = b;
    }
    __syncthreads();
    // ---- greedy NMS in score order ---------------------------------------------------------------------
    // `kept` lives in a register of EVERY thread: keys[i] is uniform after the barrier that ends the
    // previous iteration, so all threads count the same survivors and take the same break (a shared
    // counter written by thread 0 mid-iteration could be read before or after that write)
    if (iou_thr >= 1.0f) {
        // IoU never exceeds 1: nothing is suppressed (with_nms = False of the test-time-augmentation
        // path, dense_test_mixins.py:78-83) -- every candidate goes out, in score order, in parallel
        const int nout = K < max_out ? K : max_out;
        for (int i = tid; i < nout; i += DET_THREADS) {
            const unsigned long long key = keys[i];
            const float4 bi = i < DET_BOX_CAP ? sbox[i] : gbox[i];
            dn[i * 5 + 0] = bi.x; dn[i * 5 + 1] = bi.y; dn[i * 5 + 2] = bi.z; dn[i * 5 + 3] = bi.w;
            dn[i * 5 + 4] = key_score((unsigned)(key >> 32));
            src.extra((int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)), i, kn, keep_n);
        }
        if (tid == 0) *count_n = nout;
        return;
    }
    int kept = 0;
    for (int i = 0; i < K; ++i) {
        const unsigned long long key_i = keys[i];   // 0 = suppressed (written before the last barrier)
        if (key_i == 0ull) continue;
        const int o = kept;
        if (o >= max_out) break;
        const float4 bi = i < DET_BOX_CAP ? sbox[i] : gbox[i];
        for (int j = i + 1 + tid; j < K; j += DET_THREADS) {
            if (keys[j] == 0ull) continue;
            const float4 bj = j < DET_BOX_CAP ? sbox[j] : gbox[j];
            if (nms_iou(bi, bj) > iou_thr) keys[j] = 0ull;
        }
        if (tid == 0) {
            const unsigned long long key = key_i;
            dn[o * 5 + 0] = bi.x; dn[o * 5 + 1] = bi.y; dn[o * 5 + 2] = bi.z; dn[o * 5 + 3] = bi.w;
            dn[o * 5 + 4] = key_score((unsigned)(key >> 32));
            src.extra((int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)), o, kn, keep_n);
        }
        kept = o + 1;
        __syncthreads();
    }
    if (tid == 0) *count_n = kept;
}

// One image per workgroup.  Candidates (score >= score_thr) are COMPACTED first, so the sort runs
// over the next power of two above their number, not above P: any P fits (a 1024x1024 test image has
// P = 21504), and the usual few hundred candidates sort in a fraction of the time.
//   scratch per image: float4 boxes[P] | u64 keys[padP]      (padP = next power of two >= P)
template <bool FROM_HEAD>
__global__ __launch_bounds__(DET_THREADS) void detect_kernel(
    const float* __restrict__ flat, const float* __restrict__ in_scores, const int32_t* __restrict__ in_count,
    const Levels L, int P, int padP, int lds_keys, float score_thr,
    float iou_thr, int max_out, float* __restrict__ dets, float* __restrict__ kps_out,
    int32_t* __restrict__ keep, int32_t* __restrict__ count, unsigned char* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* skeys = reinterpret_cast<unsigned long long*>(smem);           // [lds_keys]
    float4* sbox = reinterpret_cast<float4*>(smem + (size_t)lds_keys * 8);             // [DET_BOX_CAP]
    __shared__ int s_k;
    const int n = blockIdx.x, tid = threadIdx.x;
    // FROM_HEAD: flat = [N,P,16] head outputs; else flat = [N,P,4] boxes with in_scores [N,P] and an
    // optional per-set element count (yunet_nms: the merged candidates of test-time augmentation)
    const FlatSrc hsrc{flat + (size_t)n * P * 16, L};
    const BoxSrc bsrc{flat + (size_t)n * P * 4, in_scores ? in_scores + (size_t)n * P : nullptr};
    const int Pn = (!FROM_HEAD && in_count) ? min(in_count[n], P) : P;
    unsigned char* sc = scratch + (size_t)n * ((size_t)P * 16 + (size_t)padP * 8);
    float4* gbox = reinterpret_cast<float4*>(sc);
    unsigned long long* gkeys = reinterpret_cast<unsigned long long*>(sc + (size_t)P * 16);
    const bool small = padP <= lds_keys;       // every prior fits in LDS: no global key traffic at all

    // ---- scores, threshold, compaction (order is irrelevant: the keys are sorted next) -----------------
    if (tid == 0) s_k = 0;
    __syncthreads();
    for (int p = tid; p < Pn; p += DET_THREADS) {
        float sc_;
        const bool ok = FROM_HEAD ? hsrc.score(p, score_thr, sc_) : bsrc.score(p, score_thr, sc_);
        if (ok) {
            const unsigned long long key =
                ((unsigned long long)score_key(sc_) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
            const int slot = atomicAdd(&s_k, 1);
            if (small) skeys[slot] = key;
            else gkeys[slot] = key;
        }
    }
    __syncthreads();
    const int K = s_k;
    int padn = 2;
    while (padn < K) padn <<= 1;
    float* dn = dets + (size_t)n * max_out * 5;
    float* kn = kps_out ? kps_out + (size_t)n * max_out * 10 : nullptr;
    int32_t* kp = keep ? keep + (size_t)n * max_out : nullptr;
    auto run = [&](auto keys) {
        if constexpr (FROM_HEAD) sort_decode_nms(keys, padn, K, hsrc, iou_thr, max_out, gbox, sbox, dn, kn, kp, count + n);
        else sort_decode_nms(keys, padn, K, bsrc, iou_thr, max_out, gbox, sbox, dn, kn, kp, count + n);
    };
    if (padn <= lds_keys) {
        if (!small)
            for (int i = tid; i < K; i += DET_THREADS) skeys[i] = gkeys[i];
        for (int i = K + tid; i < padn; i += DET_THREADS) skeys[i] = 0ull;
        __syncthreads();
        run(skeys);
    } else {
        for (int i = K + tid; i < padn; i += DET_THREADS) gkeys[i] = 0ull;
        __syncthreads();
        run(gkeys);
    }
}

}  // namespace

extern "C" size_t yunet_detect_scratch_bytes(int N, int P) {
    size_t padP = 2;
    while (padP < (size_t)P) padP <<= 1;
    return (size_t)N * ((size_t)P * 16 + padP * 8);
}

extern "C" int yunet_detect(const float* flat, const YunetLevels* lv, int N, int P, float score_thr,
                            float iou_thr, int max_out, float* dets, float* kps, int32_t* count,
                            void* scratch, void* stream) {
    if (!flat || !lv || !dets || !count || !scratch || N < 1 || P < 1 || P > (1 << 24) || max_out < 1)
        return YUNET_EINVAL;
    const Levels L = make_levels(lv);
    if (L.base[YUNET_MAX_LEVELS] != P) return YUNET_EINVAL;
    int padP = 2;
    while (padP < P) padP <<= 1;
    const int lds_keys = padP < DET_LDS_KEYS ? padP : DET_LDS_KEYS;
    const size_t smem = (size_t)lds_keys * 8 + (size_t)DET_BOX_CAP * 16;
    // the largest dynamic-LDS size is always the same (DET_LDS_KEYS keys + the box cache): raise the limit to it
    // once per device (common.h: per_device) instead of tracking a growing process-wide maximum
    static PerDevice attr_set;
    constexpr size_t smem_max = (size_t)DET_LDS_KEYS * 8 + (size_t)DET_BOX_CAP * 16;
    if (per_device(attr_set, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(detect_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max) == hipSuccess ? 1 : -1;
        }) < 0)
        return YUNET_EINVAL;
    hipLaunchKernelGGL(detect_kernel<true>, dim3(N), dim3(DET_THREADS), smem, (hipStream_t)stream, flat,
                       (const float*)nullptr, (const int32_t*)nullptr, L, P, padP, lds_keys, score_thr, iou_thr,
                       max_out, dets, kps, (int32_t*)nullptr, count, (unsigned char*)scratch);
    return hip_status();
}

extern "C" int yunet_nms(const float* boxes, const float* scores, const int32_t* counts, int N, int K,
                         float score_thr, float iou_thr, int max_out, float* dets, int32_t* keep,
                         int32_t* count, void* scratch, void* stream) {
    if (!boxes || !scores || !dets || !count || !scratch || N < 1 || K < 1 || K > (1 << 24) || max_out < 1)
        return YUNET_EINVAL;
    int padP = 2;
    while (padP < K) padP <<= 1;
    const int lds_keys = padP < DET_LDS_KEYS ? padP : DET_LDS_KEYS;
    const size_t smem = (size_t)lds_keys * 8 + (size_t)DET_BOX_CAP * 16;
    // the largest dynamic-LDS size is always the same (DET_LDS_KEYS keys + the box cache): raise the limit to it
    // once per device (common.h: per_device) instead of tracking a growing process-wide maximum
    static PerDevice attr_set;
    constexpr size_t smem_max = (size_t)DET_LDS_KEYS * 8 + (size_t)DET_BOX_CAP * 16;
    if (per_device(attr_set, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(detect_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_max) == hipSuccess ? 1 : -1;
        }) < 0)
        return YUNET_EINVAL;
    Levels L{};
    hipLaunchKernelGGL(detect_kernel<false>, dim3(N), dim3(DET_THREADS), smem, (hipStream_t)stream, boxes, scores,
                       counts, L, K, padP, lds_keys, score_thr, iou_thr, max_out, dets, (float*)nullptr, keep, count,
                       (unsigned char*)scratch);
    return hip_status();
}
