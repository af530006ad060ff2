// conv_stem.hip -- the stem (Conv_head.conv1: 3 x 3, stride 2, 3 -> 16 channels, bias; mmdet/models/utils/
// yunet_layer.py:51-52,58) forward and its weight gradient on the MATRIX CORES, as wave-streaming kernels.
//
// The VALU kernels they replace (conv_fwd.hip: stem_fwd_kernel, conv_bwd.hip: stem_bwd_kernel) run at 3.2 / 3.5 TB/s
// of algorithmic bytes with ~430 FMAs and ~100 LDS reads per output pixel; a version of the backward kernel that
// recomputed z on the VALU instead of reading it (-36 % bytes) was SLOWER (0.30 -> 0.46 ms,
// profiles/r04_stem_bwd_rz_ab.log): these kernels are instruction-bound.  As a matrix product the convolution is
// z[px][co] = sum_tap patch[px][tap] w[co][tap] with K = 27 taps (7 k-steps of v_mfma_f32_16x16x4_f32, exact fp32):
// 7 matrix instructions and 7 LDS gathers per 16 pixels instead of ~7 000 lane-FMAs.
//
// One WAVE owns a strip of 32 output columns and streams down a band of output rows.  The input rows live in a
// four-slot LDS ring per wave ([slot][channel][68 columns]: output row oy needs input rows 2 oy - 1 .. 2 oy + 1 and
// shares 2 oy + 1 with the next one); the two new rows of a step are in flight (registers) during the previous step.
// Lane (g, l15), tile t: output pixel 16 t + l15, channels 4 g .. 4 g + 3 (D layout of the instruction with the
// weights as the A operand) -- what a 16-byte store of the NHWC output wants.
//   forward : z + bias -> HBM; BatchNorm sums (fp32 per lane and band, fp64 across).
//   backward: z is RECOMPUTED (never read), dz = A dy + B z + D (bn_fold), dW[co][tap] += dz^T patch as a second
//             matrix product with K = the strip's 32 pixels (dz transposed through a 2.5 KB LDS slot, the patch
//             gathered in im2col order), db = sum dz.  Per output pixel it reads 48 B of image + 64 B of dy.
#include "common.h"
#ifndef YUNET_STEM_BWD_AUX      // cache-policy bits of the image / dy loads of the BACKWARD instance (round 6 measurement switch; 2 = non-temporal;
#define YUNET_STEM_BWD_AUX 0    //  the forward's loads on the default policy: non-temporal forward loads measured +1 % on the step)
#endif

namespace {
namespace stm {
constexpr int C = 16, PXO = 32, WAVES = 4, NTHR = 64 * WAVES;
constexpr int COLS = 2 * PXO + 4;                  // ring row: image columns 2 xo0 - 2 .. 2 xo0 + 65
constexpr int ROWF = 3 * COLS;                     // one ring slot: [channel][COLS]
constexpr int RING_F = 4 * ROWF;
constexpr int PST = 20;                            // dz slot: [pixel][16 + 4]
constexpr int WAVE_F = RING_F + PXO * PST;
constexpr int WROW = C * 27 + C;
static_assert(WAVES * WROW <= WAVES * WAVE_F, "flush area");
}  // namespace stm

template <bool BWD>
__global__ __launch_bounds__(stm::NTHR) void stem_mma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ z,
                                                             double* __restrict__ stats, const float* __restrict__ dy,
                                                             const YunetBN bn, float* __restrict__ partials, const int N,
                                                             const int H, const int W, const int R) {
    using namespace stm;
    __shared__ __attribute__((aligned(16))) float sm[WAVES * WAVE_F];
    __shared__ double s_stat[2 * C];
    __shared__ __attribute__((aligned(16))) float s_k[4][C];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int Ho = H / 2, Wo = W / 2;
    float* ring = sm + wid * WAVE_F;
    float* slot = ring + RING_F;

    if (tid < 2 * C) s_stat[tid] = 0.0;
    if (BWD && tid < C) {
        const BNFold f = bn_fold(bn_bwd_coef(bn, C, tid));       // dz = A dy + B z + D (common.h)
        s_k[0][tid] = f.a; s_k[1][tid] = f.b; s_k[2][tid] = f.dh; s_k[3][tid] = f.dl;
    }
    // A operand of the convolution: w[co = l15][tap = 4 s + g] (taps 27 is padding); its bias in the D layout
    float wa[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) wa[s] = (4 * s + g) < 27 ? w[l15 * 27 + 4 * s + g] : 0.0f;
    const float4 bq = *reinterpret_cast<const float4*>(b + 4 * g);
    __syncthreads();
    float fa[4], fb[4], fdh[4], fdl[4];
    if constexpr (BWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = s_k[0][4 * g + i]; fb[i] = s_k[1][4 * g + i]; fdh[i] = s_k[2][4 * g + i]; fdl[i] = s_k[3][4 * g + i];
        }
    }
    // ring offsets of this lane's taps.  Tap k = (ci, ky, kx) of output pixel j reads column 2 j + kx + 1 of ring row
    // (2 oy - 1 + ky) & 3: for even oy the rows sit in slots 3, 0, 1, for odd oy in slots 1, 2, 3
    auto tap_off = [&](int k, int par) {
        const int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
        return ((2 * par + 3 + ky) & 3) * ROWF + ci * COLS + kx + 1;
    };
    int koff[2][7];                     // convolution gather: tap 4 s + g
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + g < 27 ? 4 * s + g : 0;
        koff[0][s] = tap_off(k, 0);
        koff[1][s] = tap_off(k, 1);
    }
    int goff[2][2];                     // weight-gradient gather: taps l15 and 16 + l15
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = 16 * t + l15 < 27 ? 16 * t + l15 : 0;
        goff[0][t] = tap_off(k, 0);
        goff[1][t] = tap_off(k, 1);
    }
    const bool tap1_ok = 16 + l15 < 27;

    f32x4 gw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // dW[co = 4 g + i][tap = 16 t + l15]
    float gb[4] = {0.f, 0.f, 0.f, 0.f};                                          // db[co = 4 g + i] over the lane's pixels

    const int strips = (Wo + PXO - 1) / PXO, bands = (Ho + R - 1) / R;
    const int tasks_img = strips * bands, ntasks = N * tasks_img;
    // forward: z in the activation storage type of this build (common.h: act_t); backward: dy is fp32 in every build
    const unsigned img_bytes = (unsigned)(3 * H * W) * 4u, zbytes = (unsigned)(Ho * Wo * C) * (BWD ? 4u : ACT_B);
    const int total_waves = (int)gridDim.x * WAVES;

    for (int task = first_tile() * WAVES + wid; task < ntasks; task += total_waves) {
        const int n = task / tasks_img, rr = task - n * tasks_img;
        const int band = rr / strips, strip = rr - band * strips;
        const int y0 = band * R, y1 = (y0 + R < Ho) ? y0 + R : Ho;
        const int xo0 = strip * PXO;
        const auto r_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img) + (size_t)n * 3 * H * W, 0, img_bytes, 0x00020000);
        const auto r_z = BWD ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy) + (size_t)n * Ho * Wo * C, 0, zbytes, 0x00020000)
                             : __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(z) + (size_t)n * Ho * Wo * C, 0, zbytes, 0x00020000);
        // an image row of the strip = 34 aligned column pairs (lanes 0 .. 33); outside the image: offset past the end, reads 0
        const int colp = 2 * xo0 - 2 + 2 * lane;
        const bool colok = lane < COLS / 2 && colp >= 0 && colp < W;
        u32x2 lrow[2][3];                                       // the two new rows of the next step: [row][channel]
        auto issue_rows = [&](int oy) {                          // input rows 2 oy, 2 oy + 1
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int iy = 2 * oy + r;
                const bool ok = colok && (unsigned)iy < (unsigned)H;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
                    lrow[r][ci] = __builtin_amdgcn_raw_buffer_load_b64(r_img, ok ? (unsigned)((ci * H + iy) * W + colp) * 4u : img_bytes, 0, BWD ? YUNET_STEM_BWD_AUX : 0);
            }
        };
        u32x4 ldy[2];
        auto issue_dy = [&](int oy) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ox = xo0 + 16 * t + l15;
                ldy[t] = __builtin_amdgcn_raw_buffer_load_b128(
                    r_z, (oy < Ho && ox < Wo) ? (unsigned)((oy * Wo + ox) * C + 4 * g) * 4u : zbytes, 0, YUNET_STEM_BWD_AUX);
            }
        };
        // prologue: input row 2 y0 - 1 into its ring slot, rows 2 y0, 2 y0 + 1 in flight
        {
            const int iy = 2 * y0 - 1;
            const bool ok = colok && (unsigned)iy < (unsigned)H;
            u32x2 v[3];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                v[ci] = __builtin_amdgcn_raw_buffer_load_b64(r_img, ok ? (unsigned)((ci * H + iy) * W + colp) * 4u : img_bytes, 0, BWD ? YUNET_STEM_BWD_AUX : 0);
            issue_rows(y0);
            if constexpr (BWD) issue_dy(y0);
            if (lane < COLS / 2) {
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) *reinterpret_cast<u32x2*>(ring + (iy & 3) * ROWF + ci * COLS + 2 * lane) = v[ci];
            }
        }
        float ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // forward: BN sums of the band (sum | sum of squares, 4 channels)

#pragma unroll 1
        for (int oy = y0; oy < y1; ++oy) {
            const int par = oy & 1;
            // ---- the step's two new rows -> ring; the next step's rows go in flight -----------------------------------
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the previous step's gathers are done
            __builtin_amdgcn_wave_barrier();
            if (lane < COLS / 2) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci)
                        *reinterpret_cast<u32x2*>(ring + ((2 * oy + r) & 3) * ROWF + ci * COLS + 2 * lane) = lrow[r][ci];
            }
            u32x4 dyr[2];
            if constexpr (BWD) { dyr[0] = ldy[0]; dyr[1] = ldy[1]; }
            issue_rows(oy + 1);                      // (past the band: fetched, never used)
            if constexpr (BWD) issue_dy(oy + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- z = conv(patch) + bias: 7 k-steps per 16-pixel tile ------------------------------------------------------
            f32x4 zt[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = {bq.x, bq.y, bq.z, bq.w};
                const float* pj = ring + 2 * (16 * t + l15);
#pragma unroll
                for (int s = 0; s < 7; ++s) acc = mfma16(wa[s], pj[par ? koff[1][s] : koff[0][s]], acc);
                zt[t] = acc;
            }
            if constexpr (!BWD) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ox = xo0 + 16 * t + l15;
                    const bool ok = ox < Wo;
                    act_bufst4(r_z, ok ? (unsigned)((oy * Wo + ox) * C + 4 * g) * ACT_B : zbytes, make_float4(zt[t][0], zt[t][1], zt[t][2], zt[t][3]));
                    const float m = ok ? 1.0f : 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = zt[t][i] * m;
                        ts[i] += v;
                        ts[4 + i] = fmaf(v, v, ts[4 + i]);
                    }
                }
            } else {
                // ---- dz = A dy + B z + D (zero outside the map); dW += dz^T patch, db += dz -----------------------------------
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bool ok = xo0 + 16 * t + l15 < Wo;
                    float dzv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dzv[i] = ok ? bn_dz_folded(__uint_as_float(dyr[t][i]), zt[t][i], fa[i], fb[i], fdh[i], fdl[i]) : 0.0f;
                        gb[i] += dzv[i];
                    }
                    *reinterpret_cast<float4*>(slot + (16 * t + l15) * PST + 4 * g) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // K = pixels: k-step s covers pixels 4 s .. 4 s + 3; lane (g, l15) supplies dz[4 s + g][co = l15] and the
                // patch element of pixel 4 s + g for tap l15 (tile 0) / 16 + l15 (tile 1)
                const int g0 = par ? goff[1][0] : goff[0][0], g1 = par ? goff[1][1] : goff[0][1];
#pragma unroll
                for (int s = 0; s < PXO / 4; ++s) {
                    const float a = slot[(4 * s + g) * PST + l15];
                    const float* pj = ring + 2 * (4 * s + g);
                    const float p0 = pj[g0], p1 = tap1_ok ? pj[g1] : 0.0f;
                    gw[0] = mfma16(a, p0, gw[0]);
                    gw[1] = mfma16(a, p1, gw[1]);
                }
            }
        }
        if constexpr (!BWD) {
            // BN sums of the band: over the 16 pixels of a lane group, then fp64 in LDS
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = ts[i];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
                if (l15 == 0)
                    __hip_atomic_fetch_add(&s_stat[(i < 4 ? 0 : C) + 4 * g + (i & 3)], (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    if constexpr (!BWD) {
        if (tid < 2 * C) atomic_add_f64(&stats[tid], s_stat[tid]);
    } else {
        // ---- one partial row per workgroup: [co][27 taps] | [co] bias gradient -----------------------------------------
        float* red = sm + wid * WAVE_F;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (16 * t + l15 < 27) red[(4 * g + i) * 27 + 16 * t + l15] = gw[t][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = gb[i];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
            if (l15 == 0) red[C * 27 + 4 * g + i] = v;
        }
        __syncthreads();
        float* row = partials + (size_t)blockIdx.x * WROW;
        for (int i = tid; i < WROW; i += NTHR) {
            float v = 0.0f;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) v += sm[wv * WAVE_F + i];
            row[i] = v;
        }
    }
}

// rows per band: minimise (tasks per wave, rounded up) x (rows of a task + set-up)
int stem_rows(int N, int Ho, int Wo, int waves) {
    const long long strips = (Wo + stm::PXO - 1) / stm::PXO;
    int best = Ho;
    long long best_cost = -1;
    for (int R = 4; R <= Ho; ++R) {
        const long long tasks = (long long)N * strips * ((Ho + R - 1) / R);
        const long long cost = ((tasks + waves - 1) / waves) * (R + 3);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && R > best)) { best_cost = cost; best = R; }
    }
    return best;
}

}  // namespace

// conv_fwd.hip / conv_bwd.hip dispatch here (option stem_mma).  The forward is compiled once per activation storage type (fp32 |
// -DYUNET_ACT_BF16: z stored as bf16, BN sums of the unrounded values); the weight gradient with its recomputed z is fp32-storage
// only.  Grids: forward 256 CUs x 3 workgroups; backward = the rows of wgrad_partials (yunet_stem_bwd_blocks).
int ACT_SUFFIX(launch_stem_fwd_mma)(const float* img, const float* w, const float* b, float* z, double* stats, int N, int H, int W,
                                    hipStream_t stream) {
    if ((long long)3 * H * W * 4 >= (1ll << 31)) return YUNET_EINVAL;
    const int grid = 768;
    const int R = stem_rows(N, H / 2, W / 2, grid * stm::WAVES);
    hipLaunchKernelGGL(stem_mma_kernel<false>, dim3(grid), dim3(stm::NTHR), 0, stream, img, w, b, z, stats, (const float*)nullptr,
                       YunetBN{}, (float*)nullptr, N, H, W, R);
    return hip_status();
}
#ifndef YUNET_ACT_BF16
int launch_stem_bwd_mma(const float* img, const float* w, const float* b, const float* dy, const YunetBN* bn, float* partials,
                        int blocks, int N, int H, int W, hipStream_t stream) {
    if ((long long)3 * H * W * 4 >= (1ll << 31)) return YUNET_EINVAL;
    const int R = stem_rows(N, H / 2, W / 2, blocks * stm::WAVES);
    hipLaunchKernelGGL(stem_mma_kernel<true>, dim3(blocks), dim3(stm::NTHR), 0, stream, img, w, b, (float*)nullptr, (double*)nullptr, dy,
                       *bn, partials, N, H, W, R);
    return hip_status();
}
#endif
