// conv_fwd.hip -- forward kernels of the YuNet conv stack on gfx950 (NHWC fp32).
//
//   stem_fwd   Conv_head.conv1: 3x3 stride-2 3->16 (+bias)     yunet_layer.py:51-52,58
//   dp_fwd     ConvDPUnit: 1x1 pointwise (MFMA) -> 3x3 depthwise yunet_layer.py:30-36
//   pool_fwd   max_pool2d(relu(bn(z)), 2)                        yunet_backbone.py:39-40
//   upadd_fwd  relu(bn(a)) + nearest_up2(relu(bn(b)))            tfpn.py:39-40
//
// Train-mode BatchNorm is split across kernel boundaries: a producer writes its RAW conv
// output z plus fp64 per-channel sum / sum-of-squares; every consumer applies
// y = relu((z-mean)*scale+beta) while loading.  Each fused unit therefore reads its input
// once and writes its output once (SURVEY.md 8d "unit-boundary traffic").
#include <cstdlib>

#include <stdlib.h>

#include "common.h"

// 1: forward pointwise GEMM of the 32 / 64-channel units as a three-way bf16 split (fp32-accurate);
// 0: exact-fp32 matrix instruction (A/B builds, tools/ubench)
#ifndef YUNET_FWD_SPLIT3
#define YUNET_FWD_SPLIT3 1
#endif

namespace {

// ------------------------------------------------------------------------------- stem
#define STEM_TW 32
#define STEM_TH 16
#define STEM_PX 2
// 16x32 output tile per 256-thread workgroup; one thread = 2 consecutive output pixels of a
// row x 16 channels (32 accumulators), so every weight read from LDS feeds 2 pixels.  The
// input patch is staged from the NCHW image with 16-byte ALIGNED float4 loads: patch column j
// holds image column 2*x0 - 4 + j (x0 is a multiple of 32, so 2*x0 - 4 is 16-byte aligned).
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ img,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ b,
                                                       act_t* __restrict__ z,
                                                       double* __restrict__ stats, int N, int H,
                                                       int W) {
    constexpr int PH = 2 * STEM_TH + 1;                 // patch rows per channel
    constexpr int PW4 = (2 * STEM_TW + 8) / 4;          // float4 per patch row (aligned superset)
    constexpr int PWS = PW4 * 4 + 1;                    // odd LDS row stride
    constexpr int NLD = (3 * PH * PW4 + 255) / 256;     // float4 loads per thread per tile
    __shared__ float s_patch[3 * PH * PWS];
    __shared__ __attribute__((aligned(16))) float s_w[27][16];
    __shared__ double s_stat[32];      // sum | sum of squares per channel, fp64
    const int tid = threadIdx.x;
    const int Ho = H / 2, Wo = W / 2;
    for (int i = tid; i < 27 * 16; i += 256) {
        const int co = i & 15, t = i >> 4;  // t = ci*9 + ky*3 + kx ; w is [16][3][3][3]
        s_w[t][co] = w[co * 27 + t];
    }
    float bias[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) bias[c] = b[c];
    // BN statistics: fp32 per thread over at most STAT_FLUSH tiles (2*STAT_FLUSH values), then a
    // wave reduction and one fp64 LDS atomic per channel -- 32 fp64 accumulators per thread
    // would cost 64 VGPRs and a wave of occupancy
    constexpr int STAT_FLUSH = 8;
    float ts[16], tq[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) ts[c] = tq[c] = 0.0f;
    int since_flush = 0;
    if (tid < 32) s_stat[tid] = 0.0;

    const int tiles_x = (Wo + STEM_TW - 1) / STEM_TW, tiles_y = (Ho + STEM_TH - 1) / STEM_TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int cg = tid & 15, ty = tid >> 4;     // column group (STEM_PX pixels), row
    // register prefetch of the NEXT tile's patch (aligned float4 slots of the planar image)
    // through a per-image buffer descriptor: slots above / below / beside the image get an
    // out-of-range offset and read as zero (W is a multiple of 4, so a slot is never cut)
    float4 ld[NLD];
    const unsigned img_bytes = (unsigned)(3 * H * W) * 4u;
    auto issue = [&](int t) {
        const int n = t / (tiles_x * tiles_y);
        const int r = t - n * tiles_x * tiles_y;
        const int y0 = (r / tiles_x) * STEM_TH, x0 = (r % tiles_x) * STEM_TW;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img) + (size_t)n * 3 * H * W, 0,
                                                          img_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int rowi = i / PW4, c4 = i - rowi * PW4;
            const int ci = rowi / PH, py = rowi - ci * PH;
            const int iy = 2 * y0 - 1 + py, ix = 2 * x0 - 4 + 4 * c4;
            const bool ok = rowi < 3 * PH && (unsigned)iy < (unsigned)H && ix >= 0 && ix + 3 < W;
            const unsigned off = ok ? (unsigned)((ci * H + iy) * W + ix) * 4u : img_bytes;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            ld[k] = *reinterpret_cast<const float4*>(&v);
        }
    };
    int t = first_tile();
    if (t < ntiles) issue(t);
    for (; t < ntiles; t += gridDim.x) {
        const int n = t / (tiles_x * tiles_y);
        const int r = t - n * tiles_x * tiles_y;
        const int y0 = (r / tiles_x) * STEM_TH, x0 = (r % tiles_x) * STEM_TW;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            const int rowi = i / PW4, c4 = i - rowi * PW4;
            if (rowi < 3 * PH) {
                float* dst = s_patch + rowi * PWS + 4 * c4;
                dst[0] = ld[k].x; dst[1] = ld[k].y; dst[2] = ld[k].z; dst[3] = ld[k].w;
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) issue(t + gridDim.x);
        float acc[STEM_PX][16];
#pragma unroll
        for (int p = 0; p < STEM_PX; ++p)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[p][c] = bias[c];
#pragma unroll 1
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                const float* prow = s_patch + (ci * PH + 2 * ty + ky) * PWS + 2 * STEM_PX * cg + 3;
                float pv[2 * STEM_PX + 1];
#pragma unroll
                for (int j = 0; j < 2 * STEM_PX + 1; ++j) pv[j] = prow[j];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4* wr = reinterpret_cast<const float4*>(s_w[ci * 9 + ky * 3 + kx]);
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const float4 ww = wr[c4];
#pragma unroll
                        for (int p = 0; p < STEM_PX; ++p) {
                            const float v = pv[2 * p + kx];
                            acc[p][c4 * 4 + 0] = fmaf(v, ww.x, acc[p][c4 * 4 + 0]);
                            acc[p][c4 * 4 + 1] = fmaf(v, ww.y, acc[p][c4 * 4 + 1]);
                            acc[p][c4 * 4 + 2] = fmaf(v, ww.z, acc[p][c4 * 4 + 2]);
                            acc[p][c4 * 4 + 3] = fmaf(v, ww.w, acc[p][c4 * 4 + 3]);
                        }
                    }
                }
            }
        const int oy = y0 + ty;
#pragma unroll
        for (int p = 0; p < STEM_PX; ++p) {
            const int ox = x0 + STEM_PX * cg + p;
            if (oy < Ho && ox < Wo) {
                act_t* dst = z + (((size_t)n * Ho + oy) * Wo + ox) * 16;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
                    act_st4(dst + 4 * c4, make_float4(acc[p][c4 * 4], acc[p][c4 * 4 + 1], acc[p][c4 * 4 + 2],
                                                      acc[p][c4 * 4 + 3]));
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    ts[c] += acc[p][c];
                    tq[c] = fmaf(acc[p][c], acc[p][c], tq[c]);
                }
            }
        }
        if (++since_flush == STAT_FLUSH || t + (int)gridDim.x >= ntiles) {
            since_flush = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float a = ts[c], q = tq[c];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    a += __shfl_xor(a, o, 64);
                    q += __shfl_xor(q, o, 64);
                }
                if ((tid & 63) == 0) {
                    __hip_atomic_fetch_add(&s_stat[c], (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_stat[16 + c], (double)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ts[c] = tq[c] = 0.0f;
            }
        }
    }
    __syncthreads();
    if (tid < 32) atomic_add_f64(&stats[tid], s_stat[tid]);
}

// --------------------------------------------------------------------------- ConvDPUnit
// Per (TH x TW) output tile, 256 threads:
//   stage : the haloed raw input tile, PREFETCHED into registers during the previous tile's
//           compute phases, gets the input transform (BN+ReLU of the producer) and goes to LDS
//   pw    : [pixels x CIN] * [CIN x COUT] on the matrix cores (v_mfma_f32_16x16x4_f32, exact
//           fp32), result written back in place (a wave owns its 16-pixel M tiles)
//   dw    : depthwise 3x3 from LDS on the VALU, bias, coalesced store of the raw output,
//           BN statistics (fp32 per tile, fp64 across tiles)
template <int CIN, int COUT, int TH, int TW>
struct DpGeom {
    static constexpr int HW_ = TW + 2, HH_ = TH + 2, HP = HH_ * HW_;
    static constexpr int MT = (HP + 15) / 16, MP = MT * 16;
    static constexpr int CM = CIN > COUT ? CIN : COUT;
    static constexpr int LS = CM + 4;           // LDS row stride (floats), 16-byte aligned rows
    static constexpr int WS = CIN + 4;          // pointwise-weight row stride in LDS (16-B rows)
    static constexpr int C4I = CIN / 4, C4O = COUT / 4;
    static constexpr int NT = COUT / 16, KS = CIN / 4;
    static constexpr int PG = 256 / C4O;        // pixel groups in the depthwise phase
    static constexpr int RG = PG / TW;          // row groups
    static constexpr int RPT = TH / RG;         // output rows per thread
    static constexpr int NLD = (HP * C4I + 255) / 256;   // float4 loads per thread per tile
    static constexpr size_t BUF_BYTES = (size_t)MP * LS * 4;
    static constexpr size_t RED_BYTES = 256 * 8 * 8;
    static constexpr size_t BUFB = BUF_BYTES > RED_BYTES ? BUF_BYTES : RED_BYTES;
    // pointwise weights: fp32 [COUT][WS], or (split-bf16 forward, 32 / 64 input channels) three bf16
    // planes [COUT][CIN] with XOR-swizzled 16-byte chunks
    static constexpr size_t W1_FP32 = (size_t)COUT * WS * 4, W1_SPLIT = (size_t)3 * COUT * CIN * 2;
    static constexpr size_t W1_BYTES = (CIN % 32 == 0 && W1_SPLIT > W1_FP32) ? W1_SPLIT : W1_FP32;
    static constexpr size_t SMEM = BUFB + W1_BYTES + (size_t)(9 * COUT + COUT + 3 * CIN) * 4;
    static_assert(PG % TW == 0 && TH % RG == 0, "tile / thread mapping");
    static_assert(256 % C4I == 0, "load mapping");
};

// POOL (YunetDP.pool_out): the unit's output goes through BN + ReLU + max_pool2d(2).  BN + ReLU is monotone
// per channel (rising for gamma > 0, falling for gamma < 0), so the window element that wins AFTER the
// transform is the raw maximum / minimum: the depthwise phase writes, next to z, the raw winner of every 2x2
// window [N,H/2,W/2,COUT] and its window position 2*dy + dx (one byte per element).  The consumer reads the
// pooled tensor with the ordinary BN+ReLU input transform -- no pooling kernel, no second pass over z.
template <int CIN, int COUT, int TH, int TW, bool PACKED, bool POOL = false>
__global__ __launch_bounds__(256) void dp_fwd_kernel(const YunetDP d, const PackGeom pk) {
    static_assert(!(POOL && PACKED) && (!POOL || (TH % 2 == 0 && TW % 2 == 0)), "fused pooling: unpacked, even tiles");
    using G = DpGeom<CIN, COUT, TH, TW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* buf = reinterpret_cast<float*>(smem_raw);
    float* s_w1 = reinterpret_cast<float*>(smem_raw + G::BUFB);   // [COUT][WS]
    float* s_w2 = reinterpret_cast<float*>(smem_raw + G::BUFB + G::W1_BYTES);   // [9][COUT]
    float* s_b2 = s_w2 + 9 * COUT;                                // [COUT]
    float* s_coef = s_b2 + COUT;                                  // mean | scale | beta
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int H = d.H, W = d.W;
    const bool bn_in = d.in_transform == YUNET_T_BNRELU;

    const int lch4 = tid % G::C4I;  // input channel quad of this thread in the load phase
    // per-image tiling, or (PACKED) one tile grid over the packed canvas of all images
    const int tiles_x = ((PACKED ? pk.CW : W) + TW - 1) / TW, tiles_y = ((PACKED ? pk.CH : H) + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int ntiles = PACKED ? tiles_img : d.N * tiles_img;

    // prefetch registers: this thread's slice of the NEXT tile's raw halo input, loaded through
    // a per-image buffer descriptor: one 32-bit byte offset per slot, and a slot outside the
    // image gets an offset past the end (the hardware range check returns 0) -- no branches, no
    // 64-bit address arithmetic
    act_raw4 pre[G::NLD];
    // the fused heads write the fp32 [N,P,16] prediction tensor whatever the activation type is
    const bool z_f32 = YUNET_ACT_DTYPE == YUNET_F32 || (COUT == 16 && d.z_dtype == YUNET_F32);
    const unsigned ZB = z_f32 ? 4u : ACT_B;
    const unsigned xbytes = (unsigned)(H * W * CIN) * ACT_B, zbytes = (unsigned)(H * W * COUT) * ZB;
    constexpr int PSTEP = 256 / G::C4I;                   // halo pixels between a thread's slots
    constexpr int HSTEP_Y = PSTEP / G::HW_, HSTEP_X = PSTEP % G::HW_;
    auto issue = [&](int t) {
        const int n = PACKED ? 0 : t / tiles_img, rr = t - n * tiles_img;
        const int y0 = (rr / tiles_x) * TH, x0 = (rr % tiles_x) * TW;      // canvas coordinates if PACKED
        // packed: one descriptor over the whole tensor, the image index is part of the offset
        const unsigned xrange = PACKED ? (unsigned)d.N * (unsigned)d.x_img_stride * ACT_B : xbytes;
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + (PACKED ? (size_t)0 : (size_t)n * d.x_img_stride), 0,
            xrange, 0x00020000);
        int hp = tid / G::C4I;
        int hy = hp / G::HW_, hx = hp - hy * G::HW_;
#pragma unroll
        for (int i = 0; i < G::NLD; ++i) {
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            bool ok;
            unsigned off;
            if constexpr (PACKED) {
                int pn, py, px;
                ok = hp < G::HP && pk_locate(pk, y, x, pn, py, px);
                off = ok ? (unsigned)(pn * d.x_img_stride + (py * W + px) * CIN + lch4 * 4) * ACT_B : xrange;
            } else {
                ok = hp < G::HP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                off = ok ? (unsigned)((y * W + x) * CIN + lch4 * 4) * ACT_B : xbytes;
            }
            pre[i] = act_bufld4(r_x, off);
            hp += PSTEP; hy += HSTEP_Y; hx += HSTEP_X;
            if (hx >= G::HW_) { hx -= G::HW_; ++hy; }
        }
    };
    // the first tile's loads go out before the weights are staged (cold-start latency under the prologue)
    int t = first_tile();
    if (t < ntiles) issue(t);

    // bf16 build, 32 / 64 input channels: the pointwise GEMM runs on v_mfma_f32_16x16x32_bf16 (16x the
    // fp32 matrix rate) with a = bf16(relu(bn(x))) and W1 rounded to bf16 ("bf16 forward"); the 16-channel
    // units (K = 16, HBM-bound) keep the fp32 instruction
    constexpr bool BF16_MMA = YUNET_ACT_DTYPE == YUNET_BF16 && CIN % 32 == 0;
    constexpr int WSB = CIN + 8;                                      // bf16 weight row stride (elements)
    __bf16* s_w1b = reinterpret_cast<__bf16*>(s_w1);                  // [COUT][WSB], inside the fp32-sized block
    // fp32 build, 32 / 64 input channels: three-way bf16 split of BOTH operands, x = h + m + l EXACTLY
    // (3 x 8 significand bits), six bf16 MFMAs per product block (hh, hm, mh, mm, hl, lh; the dropped
    // ml, lm, ll terms are <= 2^-24 relative): fp32-accurate at 6/16 of the fp32 matrix time.
    constexpr bool SPLIT3 = YUNET_ACT_DTYPE == YUNET_F32 && CIN % 32 == 0 && YUNET_FWD_SPLIT3;
    constexpr int NCH = CIN / 8;                                      // 16-byte chunks per weight row
    __bf16* s_w1p = reinterpret_cast<__bf16*>(s_w1);                  // planes h | m | l, each [COUT][CIN], swizzled
    if constexpr (SPLIT3) {
        staged_table<COUT * CIN, 256>(d.w_pw, tid, [&](int i, float w) {       // (common.h: every load in flight first)
            const int co = i / CIN, ci = i % CIN;
            const __bf16 h = (__bf16)w;
            const float r1 = w - (float)h;
            const __bf16 m = (__bf16)r1;
            const __bf16 l = (__bf16)(r1 - (float)m);
            const int at = co * CIN + (((ci >> 3) ^ (co & (NCH - 1))) << 3) + (ci & 7);
            s_w1p[at] = h; s_w1p[COUT * CIN + at] = m; s_w1p[2 * COUT * CIN + at] = l;
        });
    } else if constexpr (BF16_MMA) {
        staged_table<COUT * CIN, 256>(d.w_pw, tid, [&](int i, float w) { s_w1b[(i / CIN) * WSB + (i % CIN)] = (__bf16)w; });
    } else {
        staged_table<COUT * CIN, 256>(d.w_pw, tid, [&](int i, float w) { s_w1[(i / CIN) * G::WS + (i % CIN)] = w; });
    }
    staged_table<COUT * 9, 256>(d.w_dw, tid, [&](int i, float w) { s_w2[(i % 9) * COUT + i / 9] = w; });
    for (int i = tid; i < COUT; i += 256) s_b2[i] = d.b_dw[i];
    for (int c = tid; c < CIN; c += 256) {
        BNCoef k{0.f, 1.f, 0.f, 1.f};
        if (bn_in) k = bn_coef(d.in_bn, CIN, c);
        s_coef[c] = k.mean;
        s_coef[CIN + c] = k.scale;
        s_coef[2 * CIN + c] = k.beta;
    }
    float bias_pw[G::NT];
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) bias_pw[nt] = d.b_pw[nt * 16 + l15];
    // depthwise: this thread owns channels cq*4..cq*4+3, column dtx, rows r0..r0+RPT-1
    const int cq = tid % G::C4O, pg = tid / G::C4O;
    const int dtx = pg % TW, r0 = (pg / TW) * G::RPT;
    double st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = 0.0;
    // fused pooling: +1 / -1 / 0 per channel = which raw value wins the window after BN + ReLU
    // (gamma == 0: every element ties, the first one is taken like F.max_pool2d does)
    float sg[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (POOL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gm = d.out_bn.gamma[cq * 4 + i];
            sg[i] = gm > 0.0f ? 1.0f : (gm < 0.0f ? -1.0f : 0.0f);
        }
    }
    __syncthreads();

    // identity units run the same straight-line transform: (x - 0) * 1 + 0 floored at -inf
    const float relu_floor = bn_in ? 0.0f : -__builtin_inff();

    unsigned long long pc[4] = {0, 0, 0, 0}, c0 = 0;
    // prof < 64 is a debug ablation mask (tools/kbench.py --ablate), not a pointer:
    // 1 skip pointwise MFMA, 2 skip depthwise phase, 4 skip global stores, 8 skip prefetch loads
    const bool prof = (unsigned long long)d.prof >= 64ull;
    const unsigned abl = (unsigned long long)d.prof < 64ull ? (unsigned)(unsigned long long)d.prof : 0u;
    for (; t < ntiles; t += gridDim.x) {
        if (prof) c0 = clock64();
        const int n = PACKED ? 0 : t / tiles_img, rr = t - n * tiles_img;
        const int y0 = (rr / tiles_x) * TH, x0 = (rr % tiles_x) * TW;      // canvas coordinates if PACKED
        // ---- stage: registers -> LDS with the input transform ----------------------------------
        {
            const float4 cm = *reinterpret_cast<const float4*>(s_coef + lch4 * 4);
            const float4 cs = *reinterpret_cast<const float4*>(s_coef + CIN + lch4 * 4);
            const float4 cb = *reinterpret_cast<const float4*>(s_coef + 2 * CIN + lch4 * 4);
            const int hp0 = tid / G::C4I;
#pragma unroll
            for (int i = 0; i < G::NLD; ++i) {
                const int hp = hp0 + PSTEP * i;
                if ((i + 1) * PSTEP <= G::MP || hp < G::MP) {
                    // (halo pixels outside the image hold T(0) here; their pointwise output is
                    // forced to zero in the pw epilogue, which is what the depthwise pads with)
                    float4 v = act_unpack(pre[i]);
                    v.x = fmaxf(fmaf(v.x - cm.x, cs.x, cb.x), relu_floor);
                    v.y = fmaxf(fmaf(v.y - cm.y, cs.y, cb.y), relu_floor);
                    v.z = fmaxf(fmaf(v.z - cm.z, cs.z, cb.z), relu_floor);
                    v.w = fmaxf(fmaf(v.w - cm.w, cs.w, cb.w), relu_floor);
                    *reinterpret_cast<float4*>(buf + hp * G::LS + lch4 * 4) = v;
                }
            }
        }
        __syncthreads();
        if (prof) { const unsigned long long c = clock64(); pc[0] += c - c0; c0 = c; }
        // the registers are free again: fetch the next tile while this one is computed
        if (t + (int)gridDim.x < ntiles && !(abl & 8)) issue(t + gridDim.x);
        // ---- pw: pointwise 1x1 on the matrix cores, in place ------------------------------------
        // epilogue of one 16-pixel M tile: bias, zero outside the image, back to LDS in place
        auto store_p = [&](int mt, const f32x4 (&acc)[G::NT]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hp = mt * 16 + 4 * g + r;
            const int hy = hp / G::HW_, hx = hp - hy * G::HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            // the depthwise conv zero-pads the POINTWISE OUTPUT: outside the image p = 0,
            // not pw(0)+bias (SURVEY.md 7 "zero-halo trap")
            int pn, py, px;
            const bool in = hp < G::HP && (PACKED ? pk_locate(pk, y, x, pn, py, px)
                                                 : (y >= 0 && y < H && x >= 0 && x < W));
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt)
                buf[hp * G::LS + nt * 16 + l15] = in ? acc[nt][r] + bias_pw[nt] : 0.0f;
        }
        };
        for (int mt = wid; mt < ((abl & 1) ? 0 : G::MT); mt += 4) {
            f32x4 acc[G::NT];
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT3) {
                // (measured: one M tile at a time, 0.240 ms at 80x80; the wave's three tiles together with
                //  shared weight fragments, 0.320 ms -- the long read / convert prologue no longer overlaps
                //  the matrix instructions of the previous tile)
                typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
                const float* arow = buf + (mt * 16 + l15) * G::LS + 8 * g;
#pragma unroll
                for (int kb = 0; kb < CIN / 32; ++kb) {
                    const float4 x0 = *reinterpret_cast<const float4*>(arow + 32 * kb);
                    const float4 x1 = *reinterpret_cast<const float4*>(arow + 32 * kb + 4);
                    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    u32x4 ah, am, al;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a0 = xs[2 * q], a1 = xs[2 * q + 1];
                        const unsigned hb = pack_bf16x2(a0, a1);
                        const float r0 = a0 - __uint_as_float(hb << 16), r1 = a1 - __uint_as_float(hb & 0xffff0000u);
                        const unsigned mb = pack_bf16x2(r0, r1);
                        const unsigned lb = pack_bf16x2(r0 - __uint_as_float(mb << 16), r1 - __uint_as_float(mb & 0xffff0000u));
                        ah[q] = hb; am[q] = mb; al[q] = lb;
                    }
                    const bf16x8_t Ah = __builtin_bit_cast(bf16x8_t, ah), Am = __builtin_bit_cast(bf16x8_t, am),
                                   Al = __builtin_bit_cast(bf16x8_t, al);
#pragma unroll
                    for (int nt = 0; nt < G::NT; ++nt) {
                        const int co = nt * 16 + l15;
                        const __bf16* bp = s_w1p + co * CIN + (((4 * kb + g) ^ (co & (NCH - 1))) << 3);
                        const bf16x8_t Bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(bp));
                        const bf16x8_t Bm = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(bp + COUT * CIN));
                        const bf16x8_t Bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(bp + 2 * COUT * CIN));
                        f32x4 c = acc[nt];                       // small terms first
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, c, 0, 0, 0);
                        acc[nt] = c;
                    }
                }
            } else if constexpr (BF16_MMA) {
                // lane group g supplies input channels 32*kb + 8g .. +7 on both sides
                typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
                const float* arow = buf + (mt * 16 + l15) * G::LS + 8 * g;
                const __bf16* brow = s_w1b + l15 * WSB + 8 * g;
#pragma unroll
                for (int kb = 0; kb < CIN / 32; ++kb) {
                    const float4 x0 = *reinterpret_cast<const float4*>(arow + 32 * kb);
                    const float4 x1 = *reinterpret_cast<const float4*>(arow + 32 * kb + 4);
                    const u32x4 ap = {pack_bf16x2(x0.x, x0.y), pack_bf16x2(x0.z, x0.w), pack_bf16x2(x1.x, x1.y),
                                      pack_bf16x2(x1.z, x1.w)};
                    const bf16x8_t av = __builtin_bit_cast(bf16x8_t, ap);
#pragma unroll
                    for (int nt = 0; nt < G::NT; ++nt) {
                        const u32x4 bp = *reinterpret_cast<const u32x4*>(brow + nt * 16 * WSB + 32 * kb);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, bp), acc[nt], 0, 0, 0);
                    }
                }
            } else {
            // MFMA operands are fetched four k-steps at a time with 16-byte LDS reads: k-step
            // (j, i) of lane group g multiplies input channel 16*j + 4*g + i (any permutation of
            // the reduction index is valid as long as A and B agree).
            const float* arow = buf + (mt * 16 + l15) * G::LS + 4 * g;
            const float* brow = s_w1 + l15 * G::WS + 4 * g;     // B[k = ci][n = co] = w_pw[co][ci]
            // software pipeline: the operands of k-group j+1 are in flight while the 4*NT MFMAs of
            // group j issue (hipcc otherwise sinks each LDS read next to its use and every MFMA
            // quartet eats a full LDS latency)
            constexpr int NJ = G::KS / 4;
            float4 a4[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) a4[j] = *reinterpret_cast<const float4*>(arow + 16 * j);
            float4 bq[2][G::NT];
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt)
                bq[0][nt] = *reinterpret_cast<const float4*>(brow + nt * 16 * G::WS);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) {
#pragma unroll
                    for (int nt = 0; nt < G::NT; ++nt)
                        bq[(j + 1) & 1][nt] =
                            *reinterpret_cast<const float4*>(brow + nt * 16 * G::WS + 16 * (j + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                // consecutive MFMAs go to DIFFERENT accumulators (a dependent MFMA costs 44 clk
                // instead of 32)
#pragma unroll
                for (int nt = 0; nt < G::NT; ++nt) acc[nt] = mfma16(a4[j].x, bq[j & 1][nt].x, acc[nt]);
#pragma unroll
                for (int nt = 0; nt < G::NT; ++nt) acc[nt] = mfma16(a4[j].y, bq[j & 1][nt].y, acc[nt]);
#pragma unroll
                for (int nt = 0; nt < G::NT; ++nt) acc[nt] = mfma16(a4[j].z, bq[j & 1][nt].z, acc[nt]);
#pragma unroll
                for (int nt = 0; nt < G::NT; ++nt) acc[nt] = mfma16(a4[j].w, bq[j & 1][nt].w, acc[nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            store_p(mt, acc);
        }
        __syncthreads();
        if (prof) { const unsigned long long c = clock64(); pc[1] += c - c0; c0 = c; }
        // ---- dw: depthwise 3x3 from LDS, bias, store raw z, BN statistics ------------------------
        if (!(abl & 2)) {
            const unsigned zrange = PACKED ? (unsigned)d.N * (unsigned)d.z_img_stride * ZB : zbytes;
            const size_t zimg = PACKED ? (size_t)0 : (size_t)n * d.z_img_stride;
            const auto r_z = __builtin_amdgcn_make_buffer_rsrc(
                z_f32 ? reinterpret_cast<void*>(d.z + zimg)
                      : reinterpret_cast<void*>(reinterpret_cast<act_t*>(d.z) + zimg), 0, zrange, 0x00020000);
            const float* pbase = buf + dtx * G::LS + cq * 4;
            float4 w2[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w2[k] = *reinterpret_cast<const float4*>(s_w2 + k * COUT + cq * 4);
            const float4 b2 = *reinterpret_cast<const float4*>(s_b2 + cq * 4);
            float ts[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ts[i] = 0.0f;
            const unsigned pobytes = (unsigned)((H >> 1) * (W >> 1) * COUT) * ACT_B;
            const auto r_po = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<act_t*>(d.pool_out) + (POOL ? (size_t)n * (pobytes / ACT_B) : (size_t)0), 0,
                POOL ? pobytes : 0u, 0x00020000);
            const auto r_pi = __builtin_amdgcn_make_buffer_rsrc(
                d.pool_idx + (POOL ? (size_t)n * (pobytes / ACT_B) : (size_t)0), 0, POOL ? pobytes / ACT_B : 0u, 0x00020000);
            float4 oprev = make_float4(0, 0, 0, 0);
            float4 rowA[3], rowB[3], rowC[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                rowA[j] = *reinterpret_cast<const float4*>(pbase + ((r0 + 0) * G::HW_ + j) * G::LS);
                rowB[j] = *reinterpret_cast<const float4*>(pbase + ((r0 + 1) * G::HW_ + j) * G::LS);
            }
#pragma unroll
            for (int r = 0; r < G::RPT; ++r) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    rowC[j] = *reinterpret_cast<const float4*>(pbase + ((r0 + r + 2) * G::HW_ + j) * G::LS);
                float4 o = b2;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    o.x = fmaf(rowA[j].x, w2[j].x, o.x); o.y = fmaf(rowA[j].y, w2[j].y, o.y);
                    o.z = fmaf(rowA[j].z, w2[j].z, o.z); o.w = fmaf(rowA[j].w, w2[j].w, o.w);
                    o.x = fmaf(rowB[j].x, w2[3 + j].x, o.x); o.y = fmaf(rowB[j].y, w2[3 + j].y, o.y);
                    o.z = fmaf(rowB[j].z, w2[3 + j].z, o.z); o.w = fmaf(rowB[j].w, w2[3 + j].w, o.w);
                    o.x = fmaf(rowC[j].x, w2[6 + j].x, o.x); o.y = fmaf(rowC[j].y, w2[6 + j].y, o.y);
                    o.z = fmaf(rowC[j].z, w2[6 + j].z, o.z); o.w = fmaf(rowC[j].w, w2[6 + j].w, o.w);
                }
                const int y = y0 + r0 + r, x = x0 + dtx;
                int pn = 0, py = y, px = x;
                const bool inside = PACKED ? pk_locate(pk, y, x, pn, py, px) : (y < H && x < W);
                if (inside) {
                    if (!(abl & 4)) {
                        const unsigned zoff = (unsigned)(pn * (PACKED ? d.z_img_stride : 0) + (py * W + px) * COUT + cq * 4);
                        if (z_f32)
                            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&o), r_z, zoff * 4u, 0, YUNET_ST_AUX);
                        else
                            act_bufst4(r_z, zoff * ACT_B, o);
                    }
                    ts[0] += o.x; ts[1] += o.y; ts[2] += o.z; ts[3] += o.w;
                    ts[4] = fmaf(o.x, o.x, ts[4]); ts[5] = fmaf(o.y, o.y, ts[5]);
                    ts[6] = fmaf(o.z, o.z, ts[6]); ts[7] = fmaf(o.w, o.w, ts[7]);
                }
                if constexpr (POOL) {
                    if ((r & 1) == 0) {
                        oprev = o;
                    } else {
                        // rows r-1 | r of this column, then the column to the right (lane + C4O); ties go to
                        // the smaller window position
                        const float a[4] = {oprev.x, oprev.y, oprev.z, oprev.w}, b[4] = {o.x, o.y, o.z, o.w};
                        float v[4];
                        unsigned jv = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool lower = b[i] * sg[i] > a[i] * sg[i];
                            v[i] = lower ? b[i] : a[i];
                            jv |= (lower ? 2u : 0u) << (8 * i);
                        }
                        float pv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) pv[i] = __shfl_down(v[i], G::C4O, 64);
                        const unsigned pj = __shfl_down(jv, G::C4O, 64) | 0x01010101u;
                        unsigned jw = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float kl = v[i] * sg[i], kr = pv[i] * sg[i];
                            const unsigned jl = (jv >> (8 * i)) & 3u, jr = (pj >> (8 * i)) & 3u;
                            const bool right = kr > kl || (kr == kl && jr < jl);
                            v[i] = right ? pv[i] : v[i];
                            jw |= (right ? jr : jl) << (8 * i);
                        }
                        if (inside && (dtx & 1) == 0) {
                            const unsigned eq = (unsigned)(((y >> 1) * (W >> 1) + (x >> 1)) * COUT + cq * 4);
                            act_bufst4(r_po, eq * ACT_B, make_float4(v[0], v[1], v[2], v[3]));
                            __builtin_amdgcn_raw_buffer_store_b32(jw, r_pi, eq, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    rowA[j] = rowB[j];
                    rowB[j] = rowC[j];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] += (double)ts[i];
        }
        if (prof) { const unsigned long long c = clock64(); pc[2] += c - c0; c0 = c; }
        __syncthreads();
        if (prof) { const unsigned long long c = clock64(); pc[3] += c - c0; c0 = c; }
    }
    if (prof && (tid & 63) == 0) {
        for (int i = 0; i < 4; ++i) d.prof[(blockIdx.x * 4 + wid) * 4 + i] = pc[i];
    }
    // ---- BN statistics of this unit's output: block reduce in LDS, one fp64 atomic per channel
    if (d.out_has_bn) {
        double* red = reinterpret_cast<double*>(smem_raw);  // [256][8]
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = st[i];
        __syncthreads();
        if (tid < 2 * COUT) {
            const int which = tid / COUT, c = tid % COUT;  // 0: sum, 1: sumsq
            const int q = c >> 2, k = (c & 3) + 4 * which;
            double v = 0.0;
            for (int p = 0; p < G::PG; ++p) v += red[(p * G::C4O + q) * 8 + k];
            atomic_add_f64(bn_slot(d.out_bn.stats, d.out_bn.slots, COUT) + which * COUT + c, v);
        }
    }
}

template <int CIN, int COUT, int TH, int TW, bool PACKED = false, bool POOL = false>
int launch_dp_fwd(const YunetDP* d, hipStream_t stream) {
    using G = DpGeom<CIN, COUT, TH, TW>;
    static PerDevice per_cu;        // resident workgroups per CU, per device (common.h)
    const int blocks_per_cu = per_device(per_cu, [] {
        const void* fn = reinterpret_cast<const void*>(dp_fwd_kernel<CIN, COUT, TH, TW, PACKED, POOL>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM) != hipSuccess) return -1;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, G::SMEM) != hipSuccess || nb < 1)
            nb = 1;
        nb = nb > 4 ? 4 : nb;
        const int forced = yunet_options().fwd_blocks_per_cu;       // occupancy experiments only
        if (forced >= 1 && forced <= nb) nb = forced;
        return nb;
    });
    if (blocks_per_cu < 1) return YUNET_EINVAL;
    // persistent grid: exactly the resident workgroups, so every one pipelines many tiles
    PackGeom pk = dp_pack_geom(d->N, d->H, d->W);
    pk.on = PACKED ? 1 : 0;
    if (!dp_pack_fits(pk, d->x_img_stride, d->z_img_stride)) return YUNET_EINVAL;
    const int tiles = PACKED ? ((pk.CW + TW - 1) / TW) * ((pk.CH + TH - 1) / TH)
                             : d->N * ((d->W + TW - 1) / TW) * ((d->H + TH - 1) / TH);
    const int resident = yunet_cu_count() * blocks_per_cu;
    const int grid = tiles < resident ? tiles : resident;
    hipLaunchKernelGGL((dp_fwd_kernel<CIN, COUT, TH, TW, PACKED, POOL>), dim3(grid), dim3(256), G::SMEM, stream, *d, pk);
    return hip_status();
}

// ------------------------------------------------------------------- pool / upsample-add
// one thread = one float4 of channels of one OUTPUT pixel
__global__ __launch_bounds__(256) void pool_fwd_kernel(const act_t* __restrict__ z, YunetBN bn,
                                                       act_t* __restrict__ out, int N, int H, int W,
                                                       int C) {
    const int C4 = C / 4, Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * C4;
    // 256 % C4 == 0, so the channel quad of a thread is loop-invariant
    const int c4 = threadIdx.x % C4;
    __shared__ float s_tab[5 * 64];
    bn_table_fill(s_tab, bn, C, threadIdx.x);
    __syncthreads();
    BNCoef k[4];
    bn_table_get(s_tab, C, c4 * 4, k);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long pix = e / C4;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho), n = (int)(pix / Ho);
        float4 m = make_float4(0, 0, 0, 0);  // relu output >= 0
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float4 v = act_ld4(z + (((size_t)n * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c4 * 4);
                m.x = fmaxf(m.x, bnrelu(v.x, k[0].mean, k[0].scale, k[0].beta));
                m.y = fmaxf(m.y, bnrelu(v.y, k[1].mean, k[1].scale, k[1].beta));
                m.z = fmaxf(m.z, bnrelu(v.z, k[2].mean, k[2].scale, k[2].beta));
                m.w = fmaxf(m.w, bnrelu(v.w, k[3].mean, k[3].scale, k[3].beta));
            }
        act_st4(out + e * 4, m);
    }
}

__global__ __launch_bounds__(256) void upadd_fwd_kernel(const act_t* __restrict__ za, YunetBN bna,
                                                        const act_t* __restrict__ zb, YunetBN bnb,
                                                        act_t* __restrict__ out, int N, int H, int W,
                                                        int C) {
    const int C4 = C / 4, Hb = H / 2, Wb = W / 2;
    const long long total = (long long)N * H * W * C4;
    const int c4 = threadIdx.x % C4;
    __shared__ float s_ta[5 * 64], s_tb[5 * 64];
    bn_table_fill(s_ta, bna, C, threadIdx.x);
    bn_table_fill(s_tb, bnb, C, threadIdx.x);
    __syncthreads();
    BNCoef ka[4], kb[4];
    bn_table_get(s_ta, C, c4 * 4, ka);
    bn_table_get(s_tb, C, c4 * 4, kb);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long pix = e / C4;
        const int x = (int)(pix % W);
        pix /= W;
        const int y = (int)(pix % H), n = (int)(pix / H);
        const float4 a = act_ld4(za + e * 4);
        const float4 b = act_ld4(zb + (((size_t)n * Hb + y / 2) * Wb + x / 2) * C + c4 * 4);
        float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = bnrelu(av[i], ka[i].mean, ka[i].scale, ka[i].beta) +
                   bnrelu(bv[i], kb[i].mean, kb[i].scale, kb[i].beta);
        act_st4(out + e * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

#ifndef YUNET_ACT_BF16
// running_mean/var update of one BN layer (nn.BatchNorm2d momentum 0.1, unbiased var)
__global__ void bn_running_kernel(const double* __restrict__ stats, float* __restrict__ rm,
                                  float* __restrict__ rv, int C, int count, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double inv = 1.0 / (double)count;
    const double mean = stats[c] * inv;
    double var = stats[C + c] * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double unb = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
    rm[c] = (1.0f - momentum) * rm[c] + momentum * (float)mean;
    rv[c] = (1.0f - momentum) * rv[c] + momentum * (float)unb;
}
#endif

// persistent grid of the element-wise forward kernels: 8 workgroups of 256 per CU, grid-stride loops
inline int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int ACT_SUFFIX(yunet_stem_fwd)(const float* img, const float* w, const float* b, float* z,
                                          double* stats, int N, int H, int W, int cmid, void* stream) {
    if (cmid != 16 || (H & 1) || (W & 1)) return YUNET_EINVAL;
    // the convolution as a matrix product on the matrix cores (conv_stem.hip)
    if (yunet_options().stem_mma) return ACT_SUFFIX(launch_stem_fwd_mma)(img, w, b, z, stats, N, H, W, (hipStream_t)stream);
    const int tiles = N * ((W / 2 + STEM_TW - 1) / STEM_TW) * ((H / 2 + STEM_TH - 1) / STEM_TH);
    // persistent grid = resident workgroups (158 VGPRs -> 3 waves/SIMD -> 3 per CU on 256 CUs)
    const int grid = tiles < 768 ? tiles : 768;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, w, b,
                       reinterpret_cast<act_t*>(z), stats, N, H, W);
    return hip_status();
}

extern "C" int ACT_SUFFIX(yunet_dp_fwd)(const YunetDP* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (d->x_dtype != YUNET_ACT_DTYPE) return YUNET_EINVAL;
    if (d->z_dtype != YUNET_ACT_DTYPE && !(d->cout == 16 && d->z_dtype == YUNET_F32)) return YUNET_EINVAL;
    if (d->in_transform != YUNET_T_IDENTITY && d->in_transform != YUNET_T_BNRELU) return YUNET_EINVAL;
    // units with 16 input channels: wave-streaming kernels (conv_fwd16.hip)
    if (d->cin == 16 && (d->cout == 16 || (d->cout == 64 && !d->pool_out)) && d->z_dtype == YUNET_ACT_DTYPE && !d->prof && yunet_options().fwd16s &&
        (!d->pool_out || (yunet_dp_pool_fusion_ok(d->N, d->H, d->W, 16, 16) && d->out_bn.gamma && d->pool_idx &&
                          !(reinterpret_cast<uintptr_t>(d->pool_idx) & 3))))
        return ACT_SUFFIX(launch_dp_fwd16s)(d, s);
    if (d->pool_out) {
        // also write the raw max_pool2d winners + their window positions (fused pooling)
        if (!yunet_dp_pool_fusion_ok(d->N, d->H, d->W, d->cin, d->cout) || !d->out_bn.gamma || !d->pool_idx ||   // (out_has_bn may be 0: eval())
            (reinterpret_cast<uintptr_t>(d->pool_idx) & 3))
            return YUNET_EINVAL;
        if (d->cin == 16) return launch_dp_fwd<16, 16, 16, 32, false, true>(d, s);
        if (d->cin == 32) return launch_dp_fwd<32, 64, 8, 16, false, true>(d, s);
        if ((unsigned long long)d->prof < 64ull && yunet_options().fwd64s && d->z_dtype == YUNET_ACT_DTYPE) return ACT_SUFFIX(launch_dp_fwd64s)(d, s);
        return launch_dp_fwd<64, 64, 8, 16, false, true>(d, s);
    }
#define DP_CASE(ci, co) \
    if (d->cin == ci && d->cout == co) return launch_dp_fwd<ci, co, 8, 16>(d, s);
    if (d->cin == 16 && d->cout == 16 && d->W >= 64 && d->H >= 32)
        return launch_dp_fwd<16, 16, 16, 32>(d, s);   // 160x160 / 80x80 levels: bigger tile
    // 20x20 / 10x10 levels: packed canvas -- except the fp32 64 -> 64 units, which run on the wave-streaming kernel
    // (one or two 14-column strips per image; option fwd64s = 1 puts them back here: 20 x 20 0.0324 -> 0.0266 ms,
    // 10 x 10 0.0152 -> 0.0132 ms, profiles/r04_fwd_small.log)
    if (dp_use_pack(d->N, d->H, d->W, d->cin, d->cout) && !(d->cout == 64 && d->z_dtype == YUNET_ACT_DTYPE && yunet_options().fwd64s >= 2))
        return d->cout == 64 ? launch_dp_fwd<64, 64, 8, 16, true>(d, s) : launch_dp_fwd<64, 16, 8, 16, true>(d, s);
    // the plain 64 -> 64 unit: wave-streaming kernel (conv_fwd64.hip); the phase-clock debug mode stays on the
    // tile kernel
#ifdef F64S_PROF
    if (d->cin == 64 && d->cout == 64 && yunet_options().fwd64s) return ACT_SUFFIX(launch_dp_fwd64s)(d, s);
#endif
    if (d->cin == 64 && d->cout == 64 && d->z_dtype == YUNET_ACT_DTYPE && (unsigned long long)d->prof < 64ull && yunet_options().fwd64s)
        return ACT_SUFFIX(launch_dp_fwd64s)(d, s);
    DP_CASE(16, 16)
    DP_CASE(16, 32)
    DP_CASE(16, 64)
    DP_CASE(32, 32)
    DP_CASE(32, 64)
    DP_CASE(64, 64)
    DP_CASE(64, 16)
#undef DP_CASE
    return YUNET_EINVAL;
}

// Independent units in one call (include/yunet_hip.h).  One launch when every unit is a plain 64 -> 64 unit that
// yunet_dp_fwd would put on the wave-streaming kernel (conv_fwd64.hip) -- the same tests as above, in the same order --
// else the units go out one after the other.
extern "C" int ACT_SUFFIX(yunet_dp_fwd_group)(const YunetDP* const* units, int n, void* stream) {
    if (!units || n < 1 || n > YUNET_DP_GROUP_MAX) return YUNET_EINVAL;
    bool one_grid = n >= 2 && yunet_options().fwd_group != 0;
    for (int i = 0; i < n && one_grid; ++i) {
        const YunetDP* d = units[i];
        one_grid = d && d->x_dtype == YUNET_ACT_DTYPE && d->z_dtype == YUNET_ACT_DTYPE && d->cin == 64 && d->cout == 64 &&
                   !d->pool_out && !d->prof && yunet_options().fwd64s != 0 &&
                   (d->in_transform == YUNET_T_IDENTITY || d->in_transform == YUNET_T_BNRELU) &&
                   !(dp_use_pack(d->N, d->H, d->W, d->cin, d->cout) && yunet_options().fwd64s < 2);
    }
    if (one_grid) return ACT_SUFFIX(launch_dp_fwd64s_group)(units, n, (hipStream_t)stream);
    for (int i = 0; i < n; ++i) {
        if (!units[i]) return YUNET_EINVAL;
        const int rc = ACT_SUFFIX(yunet_dp_fwd)(units[i], stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int ACT_SUFFIX(yunet_pool_fwd)(const float* z, const YunetBN* bn, float* out, int N, int H, int W,
                                          int C, void* stream) {
    if ((H & 1) || (W & 1) || (C & 3) || (256 % (C / 4)) || C > 64) return YUNET_EINVAL;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const act_t*>(z), *bn, reinterpret_cast<act_t*>(out), N, H, W, C);
    return hip_status();
}

extern "C" int ACT_SUFFIX(yunet_upadd_fwd)(const float* za, const YunetBN* bna, const float* zb,
                                           const YunetBN* bnb, float* out, int N, int H, int W, int C,
                                           void* stream) {
    if ((H & 1) || (W & 1) || (C & 3) || (256 % (C / 4)) || C > 64) return YUNET_EINVAL;
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(upadd_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const act_t*>(za), *bna, reinterpret_cast<const act_t*>(zb), *bnb,
                       reinterpret_cast<act_t*>(out), N, H, W, C);
    return hip_status();
}

#ifndef YUNET_ACT_BF16
extern "C" int yunet_bn_update_running(const double* stats, float* running_mean,
                                       float* running_var, int C, int count, float momentum,
                                       void* stream) {
    hipLaunchKernelGGL(bn_running_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       stats, running_mean, running_var, C, count, momentum);
    return hip_status();
}

extern "C" int yunet_conv_blocks(void) { return CONV_BLOCKS; }
extern "C" int yunet_abi_version(void) { return YUNET_ABI_VERSION; }
#endif
