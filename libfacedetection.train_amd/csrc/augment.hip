// augment.hip -- the reference TRAIN input pipeline on the device:
//   RandomSquareCrop -> Resize(keep_ratio=False) -> RandomFlip (+ 5-landmark swap) -> collate
// (mmdet/datasets/pipelines/transforms.py:975-1169, 242-299, 425-546; configs/yunet_n.py:36-56).
//
// Two kernels per batch:
//   aug_decide_kernel : one wavefront per image.  Draws the crop scale / window with the
//       reference's retry logic from a counter-based generator (the same 32-bit integer
//       function as oracle/pipeline_oracle.py), tests all box centres per attempt with a wave
//       ballot, then transforms the kept boxes / keypoints (clip, shift, scale, clip, flip) and
//       compacts them order-preserving into the padded [N, Gmax, .] layout the loss step stages.
//   aug_pixels_kernel : one thread per output pixel: flip -> bilinear taps in the (virtual) padded
//       crop -> source uint8 gather, written planar NCHW fp32 (what stem_fwd reads).
//
// Built with -ffp-contract=off: every float operation is the single rounded operation numpy
// performs, so boxes / keypoints are bit-identical to the reference and pixels to the oracle.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t stream_key(uint32_t seed, uint32_t iteration, uint32_t image) {
    const uint32_t k = mix32(seed ^ (iteration * 0x27D4EB2Fu));
    return mix32(k ^ (image * 0x9E3779B9u));
}
__device__ __forceinline__ uint32_t rand_u32(uint32_t key, uint32_t ctr) {
    return mix32(key ^ (ctr * 0x85EBCA6Bu + 0xC2B2AE35u));
}
__device__ __forceinline__ int bounded(uint32_t u, int n) {   // floor(u / 2^32 * n)
    return (int)(((unsigned long long)u * (unsigned long long)(uint32_t)n) >> 32);
}

__device__ __forceinline__ bool centre_inside(const float* b, int p0, int p1, int p2, int p3) {
    const float cx = (b[0] + b[2]) / 2.0f, cy = (b[1] + b[3]) / 2.0f;    // transforms.py:1081
    return cx > (float)p0 && cy > (float)p1 && cx < (float)p2 && cy < (float)p3;
}

#define AUG_P_LEFT 0
#define AUG_P_TOP 1
#define AUG_P_CW 2
#define AUG_P_FLIP 3
#define AUG_P_KEPT 4
#define AUG_P_DRAWS 5
#define AUG_P_STATUS 6

__global__ __launch_bounds__(64) void aug_decide_kernel(
    const int32_t* __restrict__ src_hw, const float* __restrict__ boxes, const float* __restrict__ kps,
    const int32_t* __restrict__ gt_off, const YunetAugCfg cfg, uint32_t iteration, int32_t* __restrict__ params,
    float* __restrict__ out_boxes, float* __restrict__ out_kps, int32_t* __restrict__ out_count) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int h = src_hw[2 * n], w = src_hw[2 * n + 1];
    const int g0 = gt_off[n], G = gt_off[n + 1] - g0;
    const float* bx = boxes + (size_t)g0 * 4;
    const float* kp = kps + (size_t)g0 * 15;
    const uint32_t key = stream_key(cfg.seed, iteration, (uint32_t)n);
    uint32_t ctr = 0;
    const int S = cfg.out_size, gmax = cfg.gmax;
    float* ob = out_boxes + (size_t)n * gmax * 4;
    float* ok = out_kps + (size_t)n * gmax * 15;

    // ---- RandomSquareCrop: scale / window search (transforms.py:1032-1090) --------------------
    int left = 0, top = 0, cw = 0;
    bool found = false;
    const int short_side = w < h ? w : h;
    for (int retry = 0; retry < cfg.max_retries && !found && G > 0; ++retry) {
        const double scale = cfg.crop_choice[bounded(rand_u32(key, ctr++), cfg.n_choice)];
        cw = (int)(scale * (double)short_side);
        for (int attempt = 0; attempt < cfg.max_attempts && !found; ++attempt) {
            if (w == cw) left = 0;
            else if (w > cw) left = bounded(rand_u32(key, ctr++), w - cw);
            else left = (w - cw) + bounded(rand_u32(key, ctr++), cw - w);
            if (h == cw) top = 0;
            else if (h > cw) top = bounded(rand_u32(key, ctr++), h - cw);
            else top = (h - cw) + bounded(rand_u32(key, ctr++), cw - h);
            bool any = false;
            for (int g = lane; g < G; g += 64)
                any |= centre_inside(bx + 4 * g, left, top, left + cw, top + cw);
            found = __any(any);
        }
    }
    // ---- RandomFlip: one uniform against flip_ratio (transforms.py:514-521) -------------------
    const bool flip = found && ((double)rand_u32(key, ctr++) * (1.0 / 4294967296.0) < cfg.flip_ratio);

    // ---- kept boxes / keypoints: clip to the window, shift, scale, clip, flip; compact ---------
    int kept = 0;
    if (found) {
        const int p0 = left, p1 = top, p2 = left + cw, p3 = top + cw;
        const float sf = (float)((double)S / (double)cw);     // mmcv.imresize: w_scale = S / w
        const float fS = (float)S;
        for (int base = 0; base < G; base += 64) {
            const int g = base + lane;
            const bool keep = g < G && centre_inside(bx + 4 * g, p0, p1, p2, p3);
            const unsigned long long m = __ballot(keep);
            const int idx = kept + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && idx < gmax) {
                const float* b = bx + 4 * g;
                float x1 = fmaxf(b[0], (float)p0) - (float)p0, y1 = fmaxf(b[1], (float)p1) - (float)p1;
                float x2 = fminf(b[2], (float)p2) - (float)p0, y2 = fminf(b[3], (float)p3) - (float)p1;
                x1 = fminf(fmaxf(x1 * sf, 0.0f), fS); y1 = fminf(fmaxf(y1 * sf, 0.0f), fS);
                x2 = fminf(fmaxf(x2 * sf, 0.0f), fS); y2 = fminf(fmaxf(y2 * sf, 0.0f), fS);
                if (flip) { const float t = x1; x1 = fS - x2; x2 = fS - t; }
                float* o = ob + 4 * idx;
                o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
                const float* k = kp + 15 * g;
                float* q = ok + 15 * idx;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int js = flip ? (j == 0 ? 1 : j == 1 ? 0 : j == 3 ? 4 : j == 4 ? 3 : 2) : j;
                    float x = fmaxf(fminf(k[3 * js + 0], (float)p2), (float)p0) - (float)p0;
                    float y = fmaxf(fminf(k[3 * js + 1], (float)p3), (float)p1) - (float)p1;
                    x = fminf(fmaxf(x * sf, 0.0f), fS);
                    y = fminf(fmaxf(y * sf, 0.0f), fS);
                    if (flip) x = fS - x;
                    q[3 * j + 0] = x; q[3 * j + 1] = y; q[3 * j + 2] = k[3 * js + 2];
                }
            }
            kept += __popcll(m);
        }
    }
    const int count = kept < gmax ? kept : gmax;
    for (int i = count * 4 + lane; i < gmax * 4; i += 64) ob[i] = 0.0f;     // deterministic padding
    for (int i = count * 15 + lane; i < gmax * 15; i += 64) ok[i] = 0.0f;
    if (lane == 0) {
        int32_t* p = params + 8 * n;
        p[AUG_P_LEFT] = left; p[AUG_P_TOP] = top; p[AUG_P_CW] = found ? cw : 0;
        p[AUG_P_FLIP] = flip ? 1 : 0; p[AUG_P_KEPT] = kept; p[AUG_P_DRAWS] = (int32_t)ctr;
        p[AUG_P_STATUS] = found ? (kept > gmax ? 2 : 0) : 1; p[7] = 0;
        out_count[n] = count;
    }
}

// OpenCV INTER_LINEAR coefficient of one destination coordinate (float32 bilinear, half-pixel
// centres, edge clamp) -- restated in oracle/pipeline_oracle.py:linear_coeffs
__device__ __forceinline__ void lin_coef(int d, int dst, int src, int& s0, int& s1, float& w0, float& w1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= src - 1) { f = 0.0f; s = src - 1; }
    s0 = s;
    s1 = s + 1 < src ? s + 1 : src - 1;
    w0 = 1.0f - f;
    w1 = f;
}

__global__ __launch_bounds__(256) void aug_pixels_kernel(
    const uint8_t* __restrict__ src, const long long* __restrict__ src_off, const int32_t* __restrict__ src_hw,
    const int32_t* __restrict__ params, int S, float pad, float* __restrict__ out) {
    const int n = blockIdx.y;
    const int32_t* p = params + 8 * n;
    const int left = p[AUG_P_LEFT], top = p[AUG_P_TOP], cw = p[AUG_P_CW], flip = p[AUG_P_FLIP];
    const int h = src_hw[2 * n], w = src_hw[2 * n + 1];
    const uint8_t* im = src + src_off[n];
    float* o = out + (size_t)n * 3 * S * S;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
        const int dy = i / S, dx = i - dy * S;
        float v[3] = {pad, pad, pad};
        if (cw > 0) {
            const int dxs = flip ? S - 1 - dx : dx;
            int sx0, sx1, sy0, sy1;
            float a0, a1, b0, b1;
            lin_coef(dxs, S, cw, sx0, sx1, a0, a1);
            lin_coef(dy, S, cw, sy0, sy1, b0, b1);
            const int X0 = left + sx0, X1 = left + sx1, Y0 = top + sy0, Y1 = top + sy1;
            const bool x0in = X0 >= 0 && X0 < w, x1in = X1 >= 0 && X1 < w;
            const bool y0in = Y0 >= 0 && Y0 < h, y1in = Y1 >= 0 && Y1 < h;
            const uint8_t* r0 = im + ((size_t)Y0 * w) * 3;
            const uint8_t* r1 = im + ((size_t)Y1 * w) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v00 = (y0in && x0in) ? (float)r0[3 * X0 + c] : pad;
                const float v01 = (y0in && x1in) ? (float)r0[3 * X1 + c] : pad;
                const float v10 = (y1in && x0in) ? (float)r1[3 * X0 + c] : pad;
                const float v11 = (y1in && x1in) ? (float)r1[3 * X1 + c] : pad;
                const float t0 = v00 * a0 + v01 * a1;          // horizontal pass
                const float t1 = v10 * a0 + v11 * a1;
                v[c] = t0 * b0 + t1 * b1;                      // vertical pass
            }
        }
        o[i] = v[0];
        o[(size_t)S * S + i] = v[1];
        o[(size_t)2 * S * S + i] = v[2];
    }
}

}  // namespace

extern "C" int yunet_aug_decide(const int32_t* src_hw, const float* boxes, const float* kps,
                                const int32_t* gt_off, const YunetAugCfg* cfg, uint32_t iteration, int N,
                                int32_t* params, float* out_boxes, float* out_kps, int32_t* out_count,
                                void* stream) {
    if (!cfg || N < 1 || cfg->n_choice < 1 || cfg->n_choice > 8 || cfg->out_size < 1 || cfg->gmax < 1 ||
        cfg->max_attempts < 1 || cfg->max_retries < 1)
        return YUNET_EINVAL;
    hipLaunchKernelGGL(aug_decide_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, src_hw, boxes, kps,
                       gt_off, *cfg, iteration, params, out_boxes, out_kps, out_count);
    return hip_status();
}

extern "C" int yunet_aug_pixels(const uint8_t* src, const long long* src_off, const int32_t* src_hw,
                                const int32_t* params, const YunetAugCfg* cfg, int N, float* out_img,
                                void* stream) {
    if (!cfg || N < 1 || cfg->out_size < 1) return YUNET_EINVAL;
    const int S = cfg->out_size;
    int bx = (S * S + 255) / 256;
    if (bx > 64) bx = 64;                       // grid-stride over the pixels of one image
    hipLaunchKernelGGL(aug_pixels_kernel, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, src, src_off,
                       src_hw, params, S, cfg->pad_value, out_img);
    return hip_status();
}
