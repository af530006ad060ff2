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
#define DET_MAX_P 16384          // keys: 8 B x 16384 = 128 KB of LDS
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

__global__ __launch_bounds__(DET_THREADS) void detect_kernel(
    const float* __restrict__ flat, const Levels L, int P, int padn, float score_thr, float iou_thr,
    int max_out, float* __restrict__ dets, float* __restrict__ kps_out, int32_t* __restrict__ count,
    float4* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);            // [padn]
    float4* sbox = reinterpret_cast<float4*>(smem + (size_t)padn * 8);                 // [DET_BOX_CAP]
    __shared__ int s_k;
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* f = flat + (size_t)n * P * 16;
    float4* gbox = scratch + (size_t)n * P;

    // ---- scores and keys ------------------------------------------------------------------------
    if (tid == 0) s_k = 0;
    for (int p = tid; p < padn; p += DET_THREADS) {
        unsigned long long key = 0ull;
        if (p < P) {
            const float sc = sigmoid_ref(f[p * 16 + 0]) * sigmoid_ref(f[p * 16 + 5]);
            if (sc >= score_thr)
                key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
        }
        keys[p] = key;
    }
    __syncthreads();
    // ---- bitonic sort, descending -----------------------------------------------------------------
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
    // ---- number of candidates (keys are sorted: first zero key ends them) ----------------------------
    for (int i = tid; i < padn; i += DET_THREADS)
        if (keys[i] != 0ull && (i + 1 == padn || keys[i + 1] == 0ull)) s_k = i + 1;
    __syncthreads();
    const int K = s_k;
    // ---- decode the candidates in sorted order (yunet_head.py:376-386) -------------------------------
    for (int i = tid; i < K; i += DET_THREADS) {
        const int p = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        float px, py, s;
        prior_of(L, p, px, py, s);
        const float* q = f + p * 16;
        const float cx = q[1] * s + px, cy = q[2] * s + py;
        const float w = expf(q[3]) * s, h = expf(q[4]) * s;
        const float4 b = make_float4(cx - w / 2.0f, cy - h / 2.0f, cx + w / 2.0f, cy + h / 2.0f);
        gbox[i] = b;
        if (i < DET_BOX_CAP) sbox[i] = b;
    }
    __syncthreads();
    // ---- greedy NMS in score order ---------------------------------------------------------------------
    float* dn = dets + (size_t)n * max_out * 5;
    float* kn = kps_out ? kps_out + (size_t)n * max_out * 10 : nullptr;
    // `kept` lives in a register of EVERY thread: keys[i] is uniform after the barrier that ends the
    // previous iteration, so all threads count the same survivors and take the same break (a shared
    // counter written by thread 0 mid-iteration could be read before or after that write)
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
            dn[o * 5 + 4] = __uint_as_float((unsigned)(key >> 32));
            if (kn) {                               // _kps_decode (yunet_head.py:388-393)
                const int p = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
                float px, py, s;
                prior_of(L, p, px, py, s);
                const float* q = f + p * 16 + 6;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    kn[o * 10 + 2 * t] = q[2 * t] * s + px;
                    kn[o * 10 + 2 * t + 1] = q[2 * t + 1] * s + py;
                }
            }
        }
        kept = o + 1;
        __syncthreads();
    }
    if (tid == 0) count[n] = kept;
}

}  // namespace

extern "C" int yunet_detect(const float* flat, const YunetLevels* lv, int N, int P, float score_thr,
                            float iou_thr, int max_out, float* dets, float* kps, int32_t* count,
                            void* scratch, void* stream) {
    if (!flat || !lv || !dets || !count || !scratch || N < 1 || P < 1 || P > DET_MAX_P || max_out < 1)
        return YUNET_EINVAL;
    const Levels L = make_levels(lv);
    if (L.base[YUNET_MAX_LEVELS] != P) return YUNET_EINVAL;
    int padn = 2;
    while (padn < P) padn <<= 1;
    const size_t smem = (size_t)padn * 8 + (size_t)DET_BOX_CAP * 16;
    static size_t attr = 0;
    if (smem > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(detect_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = smem;
    }
    hipLaunchKernelGGL(detect_kernel, dim3(N), dim3(DET_THREADS), smem, (hipStream_t)stream, flat, L, P,
                       padn, score_thr, iou_thr, max_out, dets, kps, count, (float4*)scratch);
    return hip_status();
}
