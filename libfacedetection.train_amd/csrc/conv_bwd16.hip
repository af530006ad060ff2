// conv_bwd16.hip -- backward of the 16 -> 16 ConvDPUnit (yunet_layer.py:30-36; autograd in the reference) as a
// WAVE-STREAMING kernel that RECOMPUTES the unit's raw output z instead of reading it.
//
// Why (profiles/r03_pmc_traffic.json): the tile kernel dp_bwd_kernel<16,16,16,32,...> moves
// 1.40 GB per 160 x 160 launch for 0.97 GB of unit-boundary bytes -- the BatchNorm backward is two-pass, so next to
// dy it reads the whole output z a second time (x1.45) -- on one 512-thread workgroup per CU with seven barriers
// per tile.  For a 16-channel input the forward is cheap enough to repeat: p = W1 a + b1 is 16 x 16 per pixel on
// the matrix cores, the depthwise 3 x 3 is 9 taps x 16 channels.
//
// One WAVE owns a strip of 32 input columns (28 output columns: the recomputed z needs one more halo column than
// the gradient) and streams down a band of rows; there is no workgroup barrier in the main loop.  Two register
// layouts, converted through 2.5 KB per-wave LDS slots:
//   P (pixel-major): lane (g, l15), tile t holds pixel 16 t + l15, channels 4 g .. 4 g + 3 -- what a 16-byte global
//     access delivers and the B operand / D result layout of v_mfma_f32_16x16x4_f32 with the weights as A operand;
//   C (channel-major): lane (c, sg) holds channel c, pixels 8 sg .. 8 sg + 7 -- the depthwise layout: a pixel's
//     column neighbours sit in the same lane, the nine taps and the BatchNorm coefficients of the channel are nine +
//     four scalars per lane, and the depthwise weight gradient needs 9 accumulators per lane (36 in layout P).
// Per step X (one input row):
//     P  p(X)    = W1 a(X) + b1                   matrix cores, exact fp32              (a = relu(bn(x)))
//     C  z(X-1) += taps of p(X)                   accumulate form: the row is complete  (z = dw3x3(p) + b2)
//     C  dz(X-1) = A dy + B z + D                 folded BatchNorm backward (common.h: bn_fold)
//     C  dW2    += p(X-2 .. X) x dz(X-1),  db2;   dp(X-2) += taps of dz(X-1): complete;  db1
//     P  da(X-2) = W1^T dp(X-2)                   matrix cores; ReLU mask, producer's BN-backward sums, dx store
//     P  dW1    += a(X-2)^T dp(X-2)               matrix cores, K = the strip's 32 pixels
// The next row of x and of dy is in flight (registers, 18 per lane) while a row is processed.  (Fetching it straight into
// LDS -- buffer_load ... lds, no registers -- was built first: the compiler then makes EVERY LDS access that follows wait
// for the load, which serialises the prefetch.)  HBM traffic per pixel: x (read with a 2-pixel halo), dy (full size, or the pooled gradient + 1 position
// byte per element when the unit feeds max_pool2d: YunetDP.pool_idx) and dx -- z is not read.
#include "common.h"

// cache-policy bits of the streamed loads (measurement switches, round 6; 2 = non-temporal)
#ifndef YUNET_BWD16_X_AUX
#define YUNET_BWD16_X_AUX 0
#endif
#ifndef YUNET_BWD16_DY_AUX
#define YUNET_BWD16_DY_AUX 0
#endif

namespace {
namespace b16s {
constexpr int C = 16, PXW = 32, HALO = 2, OUTW = PXW - 2 * HALO;
constexpr int WAVES = 8, NTHR = 64 * WAVES;
constexpr int WROW = C * C + C + C * 9 + C;
constexpr int PST = 20;                            // floats per pixel of a row slot (16 + 4: 16-byte stores without bank conflicts)
constexpr int SLOT = PXW * PST;
constexpr int XRING = 3;                           // raw x rows X - 2 .. X (mask / BN sums two steps after the load)
constexpr int WAVE_F = 3 * SLOT + XRING * 2 * 256; // per wave: slot A (p row | a row), slot D (dp row), slot Y (dy row), x ring [XRING][2][lane][4]
constexpr int OFF_W2 = WAVES * WAVE_F;             // [9][C]
constexpr int OFF_B2 = OFF_W2 + 9 * C;
constexpr int OFF_B1 = OFF_B2 + C;
constexpr int OFF_FOLD = OFF_B1 + C;               // [4][C]  A | B | Dh | Dl of the unit's own BN
constexpr int OFF_IN = OFF_FOLD + 4 * C;           // [5][C]  mean | scale | beta | invstd | mean_lo of the producer's BN
constexpr int OFF_ST = OFF_IN + 5 * C;             // double [2][C]
constexpr int SMEM_F = OFF_ST + 4 * C;
static_assert(WROW <= WAVE_F, "flush area inside the wave's slots");
static_assert((OFF_ST * 4) % 8 == 0, "fp64 alignment");
}  // namespace b16s

__device__ __forceinline__ float lane_read(int src_lane, float v) {       // v of lane src_lane (no memory access)
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int opaque(int v) {        // stops the compiler from hoisting what is derived from v
    asm volatile("" : "+v"(v));
    return v;
}

template <bool POOLDY>
__global__ __launch_bounds__(b16s::NTHR, 1) void dp_bwd16s_kernel(const YunetDP d, const int R) {
    using namespace b16s;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sm = reinterpret_cast<float*>(smem);
    float* s_w2 = sm + OFF_W2;
    float* s_b2 = sm + OFF_B2;
    float* s_b1 = sm + OFF_B1;
    float* s_fold = sm + OFF_FOLD;
    float* s_in = sm + OFF_IN;
    double* s_st = reinterpret_cast<double*>(sm + OFF_ST);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;          // layout P: pixel (of a tile) | channel quad;  layout C: channel | pixel segment
    const int H = d.H, W = d.W;
    const bool bn_in = d.in_transform == YUNET_T_BNRELU;

    // ---- prologue: tables ---------------------------------------------------------------------------------------
    for (int i = tid; i < C * 9; i += NTHR) s_w2[(i % 9) * C + i / 9] = d.w_dw[i];
    if (tid < C) {
        s_b2[tid] = d.b_dw[tid];
        s_b1[tid] = d.b_pw[tid];
        const BNFold f = bn_fold(bn_bwd_coef(d.out_bn, C, tid));
        s_fold[tid] = f.a; s_fold[C + tid] = f.b; s_fold[2 * C + tid] = f.dh; s_fold[3 * C + tid] = f.dl;
        if (bn_in) {
            const BNCoef k = bn_coef(d.in_bn, C, tid);
            s_in[tid] = k.mean; s_in[C + tid] = k.scale; s_in[2 * C + tid] = k.beta; s_in[3 * C + tid] = k.invstd;
            s_in[4 * C + tid] = k.mean_lo;
        } else {
            s_in[tid] = 0.f; s_in[C + tid] = 1.f; s_in[2 * C + tid] = 0.f; s_in[3 * C + tid] = 1.f; s_in[4 * C + tid] = 0.f;
        }
    }
    if (tid < 2 * C) s_st[tid] = 0.0;
    // weight fragments: A operands of the p GEMM (W1[co = l15][ci = 4 g + s]) and of the da GEMM (W1[co = 4 g + s][ci = l15])
    float w1a[4], w1t[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w1a[s] = d.w_pw[l15 * C + 4 * g + s];
        w1t[s] = d.w_pw[(4 * g + s) * C + l15];
    }
    __syncthreads();
    // layout C constants of this lane's channel (c = l15): the nine taps, the depthwise bias, the folded BN backward
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = s_w2[k * C + l15];
    const float b2c = s_b2[l15];
    const float fA = s_fold[l15], fB = s_fold[C + l15], fDh = s_fold[2 * C + l15], fDl = s_fold[3 * C + l15];

    const float relu_floor = bn_in ? 0.0f : -__builtin_inff();
    float* slot_a = sm + wid * WAVE_F;             // p row (P write, C read), then the a row of the dW1 GEMM
    float* slot_d = slot_a + SLOT;                 // dp row (C write; P read for da, k-major read for dW1)
    float* ring_x = slot_d + SLOT;                 // [XRING][2][lane][4]
    float* pf_dy = ring_x + XRING * 2 * 256;       // dy row, layout P -> C
    auto ld4 = [&](const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    };

    // persistent accumulators.  Layout C: depthwise weight gradient of channel l15 (9 taps), db1, db2 over the lane's
    // pixels; layout P: dW1[ci = 4 g + i][co = l15] (matrix result)
    float gw2[9], gb1 = 0.0f, gb2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) gw2[k] = 0.0f;
    f32x4 gw1 = {0.f, 0.f, 0.f, 0.f};
    float p0[8], p1[8], p2[8], zacc0[8], zacc1[8], dpa0[8], dpa1[8];          // layout C rows
#pragma unroll
    for (int k = 0; k < 8; ++k) p0[k] = p1[k] = p2[k] = zacc0[k] = zacc1[k] = dpa0[k] = dpa1[k] = 0.0f;

    const int strips = (W + OUTW - 1) / OUTW, bands = (H + R - 1) / R;
    const int tasks_img = strips * bands, ntasks = d.N * tasks_img;
    // x in the activation storage type of this build (common.h: act_t); dy and dx are fp32 in every build
    const unsigned xbytes = (unsigned)(H * W * C) * ACT_B, gbytes = (unsigned)(H * W * C) * 4u;
    const int Wq = W >> 1;
    const unsigned pooledbytes = (unsigned)((H >> 1) * Wq * C) * 4u;
    const int total_waves = (int)gridDim.x * WAVES;

    for (int task = first_tile() * WAVES + wid; task < ntasks; task += total_waves) {
        const int n = task / tasks_img, rr = task - n * tasks_img;
        const int band = rr / strips, strip = rr - band * strips;
        const int y0 = band * R, y1 = (y0 + R < H) ? y0 + R : H;
        const int xs = strip * OUTW - HALO;                  // image column of the strip's pixel 0
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + (size_t)n * d.x_img_stride, 0, xbytes, 0x00020000);
        const auto r_dx = __builtin_amdgcn_make_buffer_rsrc(d.dx + (size_t)n * d.x_img_stride, 0, gbytes, 0x00020000);
        const auto r_dy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(d.dy) + (POOLDY ? (size_t)n * (pooledbytes / 4u) : (size_t)n * d.z_img_stride), 0,
            POOLDY ? pooledbytes : gbytes, 0x00020000);
        const auto r_id = __builtin_amdgcn_make_buffer_rsrc(d.pool_idx + (POOLDY ? (size_t)n * (pooledbytes / 4u) : (size_t)0), 0,
                                                            POOLDY ? pooledbytes / 4u : 0u, 0x00020000);
        // layout P geometry, re-derived where it is used: pixel 16 t + l15 is image column xs + 16 t + l15
        auto colP = [&](int t) { return xs + 16 * t + opaque(l15); };
        auto colvP = [&](int t) { return (unsigned)colP(t) < (unsigned)W; };
        auto ownP = [&](int t) {
            const int j = 16 * t + l15;
            return colvP(t) && j >= HALO && j < PXW - HALO;
        };
        // layout C geometry: bit k = pixel 8 g + k of the strip is inside the image | is an output pixel of this strip
        unsigned cmask = 0, omask = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = 8 * g + k;
            const bool v = (unsigned)(xs + j) < (unsigned)W;
            cmask |= v ? (1u << k) : 0u;
            omask |= (v && j >= HALO && j < PXW - HALO) ? (1u << k) : 0u;
        }
        act_raw4 lx[2];                    // the next row of x | dy in flight (layout P)
        u32x4 ldy[2];
        unsigned lid[2];                   // POOLDY: the four position bytes of the lane's channel quad
        auto issue_x = [&](int y) {
            const bool yin = (unsigned)y < (unsigned)H;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                lx[t] = act_bufld4_aux<YUNET_BWD16_X_AUX>(r_x, (yin && colvP(t)) ? (unsigned)((y * W + colP(t)) * C + 4 * g) * ACT_B : xbytes);
        };
        auto issue_dy = [&](int y) {
            const bool yin = (unsigned)y < (unsigned)H;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool ok = yin && colvP(t);
                if constexpr (POOLDY) {
                    const unsigned eq = (unsigned)(((y >> 1) * Wq + (colP(t) >> 1)) * C + 4 * g);
                    ldy[t] = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eq * 4u : pooledbytes, 0, YUNET_BWD16_DY_AUX);
                    lid[t] = __builtin_amdgcn_raw_buffer_load_b32(r_id, ok ? eq : pooledbytes, 0, 0);
                } else {
                    ldy[t] = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? (unsigned)((y * W + colP(t)) * C + 4 * g) * 4u : gbytes, 0, YUNET_BWD16_DY_AUX);
                }
            }
        };
        issue_x(y0 - 2);
        float ts0[4] = {0, 0, 0, 0}, ts1[4] = {0, 0, 0, 0};       // layout P: producer's BN-backward sums of the band
        int ring = 0;                                            // ring slot of x(X)

        float zdone[8], dpdone[8];
        auto pz_phase = [&](const int X) {
            const bool xin = (unsigned)X < (unsigned)H;
            // ---- [P] x(X) -> ring; a = T(x); p(X) = W1 a + b1 -> slot A -----------------------------------------------
            {
                const int gq = opaque(4 * g);      // (a fresh copy per phase: the table reads stay inside the loop, live only here)
                float im[4], isc[4], ibt[4], b1q[4];
                ld4(s_in + gq, im); ld4(s_in + C + gq, isc); ld4(s_in + 2 * C + gq, ibt); ld4(s_b1 + gq, b1q);
                float* rx = ring_x + ring * 512;
                const float4 xr[2] = {act_unpack(lx[0]), act_unpack(lx[1])};
                issue_x(X + 1);            // (the row past the band's last one is fetched and never used)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    *reinterpret_cast<float4*>(rx + t * 256 + lane * 4) = xr[t];
                    const float xv[4] = {xr[t].x, xr[t].y, xr[t].z, xr[t].w};
                    f32x4 acc = {b1q[0], b1q[1], b1q[2], b1q[3]};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float a = fmaxf(fmaf(xv[s] - im[s], isc[s], ibt[s]), relu_floor);
                        acc = mfma16(w1a[s], a, acc);
                    }
                    const bool ok = xin && colvP(t);          // zero padding of the depthwise input
                    const float4 pv = make_float4(ok ? acc[0] : 0.0f, ok ? acc[1] : 0.0f, ok ? acc[2] : 0.0f, ok ? acc[3] : 0.0f);
                    *reinterpret_cast<float4*>(slot_a + (16 * t + l15) * PST + gq) = pv;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- [C] p(X) of channel l15, pixels 8 g .. 8 g + 7 (+ the two neighbours outside the segment) --------------
            float pLe, pRe;
            {
                const int cq = opaque(l15), sq = opaque(g);
#pragma unroll
                for (int k = 0; k < 8; ++k) { p0[k] = p1[k]; p1[k] = p2[k]; }
                const float* pc = slot_a + (8 * sq) * PST + cq;
#pragma unroll
                for (int k = 0; k < 8; ++k) p2[k] = pc[k * PST];
                pLe = slot_a[(sq > 0 ? 8 * sq - 1 : 0) * PST + cq];
                pRe = slot_a[(sq < 3 ? 8 * sq + 8 : PXW - 1) * PST + cq];
            }
            // z in accumulate form: p(X) is the bottom tap row of z(X - 1), the middle of z(X), the top of z(X + 1)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float l = k > 0 ? p2[k - 1] : pLe, m = p2[k], r = k < 7 ? p2[k + 1] : pRe;
                zdone[k] = fmaf(r, wt[8], fmaf(m, wt[7], fmaf(l, wt[6], zacc0[k])));
                zacc0[k] = fmaf(r, wt[5], fmaf(m, wt[4], fmaf(l, wt[3], zacc1[k])));
                zacc1[k] = fmaf(r, wt[2], fmaf(m, wt[1], fmaf(l, wt[0], b2c)));
            }
            // from here on the row only feeds dW2, which counts a (dz, p) pair at the owner of the p pixel: keep the
            // pixels this wave owns
            {
                const bool rown = X >= y0 && X < y1;
#pragma unroll
                for (int k = 0; k < 8; ++k) p2[k] = (rown && ((omask >> k) & 1u)) ? p2[k] : 0.0f;
            }
        };
        auto grad_phase = [&](const int X) {
            // ---- [C] dz(q), q = X - 1: folded BatchNorm backward on the recomputed z -------------------------------------
            const int q = X - 1;
            const bool qin = (unsigned)q < (unsigned)H;
            float dz[8];
            {
                const int cq = opaque(l15), sq = opaque(g), gq = opaque(4 * g);
                // layout P -> C through the dy slot [px][PST].  Pooled dy (max_pool2d backward): the gradient reaches the
                // window maximum only -- decided here, where a lane holds the four position bytes of its channel quad
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 v = ldy[t];
                    if constexpr (POOLDY) {
                        const unsigned pos = ((unsigned)(q & 1) << 1) | (unsigned)(colP(t) & 1), id = lid[t];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = ((id >> (8 * i)) & 0xffu) == pos ? v[i] : 0u;
                    }
                    *reinterpret_cast<u32x4*>(pf_dy + (16 * t + l15) * PST + gq) = v;
                }
                issue_dy(X);               // dy(q + 1) for the next step
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float* dyp = pf_dy + (8 * sq) * PST + cq;
                float dyv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) dyv[k] = dyp[k * PST];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gy = dyv[k];
                    const bool ok = qin && ((cmask >> k) & 1u);
                    dz[k] = ok ? fmaf(fA, gy, fmaf(fB, zdone[k], fDh)) + fDl : 0.0f;
                }
            }
            // ---- [C] dW2, db2 over the p pixels this wave OWNS (each (dz, p) pair is counted by the owner of the p pixel);
            //      dp in accumulate form ------------------------------------------------------------------------------------
            {
                // the neighbours outside the segment: pixel 8 g - 1 is the last one of lane - 16, pixel 8 g + 8 the first of lane + 16
                const float dLe = lane_read(lane - 16, dz[7]), dRe = lane_read(lane + 16, dz[0]);
                const bool r1 = X - 1 >= y0 && X - 1 < y1;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float l = k > 0 ? dz[k - 1] : dLe, m = dz[k], r = k < 7 ? dz[k + 1] : dRe;
                    const bool mine = (omask >> k) & 1u;
                    // dW2[ky][kx] += p(q + ky - 1, c) dz(q, c - kx + 1): kx = 0 takes the RIGHT neighbour
                    const float q0 = p0[k], q1 = p1[k], q2 = p2[k];          // (rows already masked to the pixels this wave owns)
                    gw2[0] = fmaf(q0, r, gw2[0]); gw2[1] = fmaf(q0, m, gw2[1]); gw2[2] = fmaf(q0, l, gw2[2]);
                    gw2[3] = fmaf(q1, r, gw2[3]); gw2[4] = fmaf(q1, m, gw2[4]); gw2[5] = fmaf(q1, l, gw2[5]);
                    gw2[6] = fmaf(q2, r, gw2[6]); gw2[7] = fmaf(q2, m, gw2[7]); gw2[8] = fmaf(q2, l, gw2[8]);
                    gb2 += (mine && r1) ? m : 0.0f;
                    // dp(y', c) += w[ky][kx] dz(y' - ky + 1, c - kx + 1)
                    dpdone[k] = fmaf(l, wt[2], fmaf(m, wt[1], fmaf(r, wt[0], dpa0[k])));
                    dpa0[k] = fmaf(l, wt[5], fmaf(m, wt[4], fmaf(r, wt[3], dpa1[k])));
                    dpa1[k] = fmaf(l, wt[8], fmaf(m, wt[7], r * wt[6]));
                }
            }
        };
        auto row_phase = [&](const int X) {
            // ---- row r = X - 2: [C] dp -> slot D; [P] da = W1^T dp, ReLU mask + producer's BN-backward sums, dx; dW1 += a^T dp
            const int r = X - 2;
            {
                const int cq = opaque(l15), sq = opaque(g);
                float* dq = slot_d + (8 * sq) * PST + cq;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = ((omask >> k) & 1u) ? dpdone[k] : 0.0f;
                    gb1 += v;
                    dq[k * PST] = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const float* rx = ring_x + ring * 512;                  // x(X - 2)
#pragma unroll
                for (int t = 0; t < 2; ++t) {          // one tile at a time (fenced: its temporaries die before the next)
                    const int gq = opaque(4 * g);
                    const int j = 16 * t + l15;
                    float xr[4], dp[4], xc[4], tt[4];
                    ld4(rx + t * 256 + lane * 4, xr);
                    ld4(slot_d + j * PST + gq, dp);
                    f32x4 da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s) da = mfma16(w1t[s], dp[s], da);
                    {
                        float im[4], isc[4], ibt[4];
                        ld4(s_in + gq, im); ld4(s_in + C + gq, isc); ld4(s_in + 2 * C + gq, ibt);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            xc[i] = xr[i] - im[i];
                            tt[i] = fmaf(xc[i], isc[i], ibt[i]);
                        }
                    }
                    // the a row of the dW1 GEMM replaces the p row in slot A (read in full at the top of the step)
                    *reinterpret_cast<float4*>(slot_a + j * PST + gq) =
                        make_float4(fmaxf(tt[0], relu_floor), fmaxf(tt[1], relu_floor), fmaxf(tt[2], relu_floor), fmaxf(tt[3], relu_floor));
                    float4 o;
                    float* op = &o.x;
                    {
                        float iiv[4], ilo[4];
                        ld4(s_in + 3 * C + gq, iiv); ld4(s_in + 4 * C + gq, ilo);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = (!bn_in || tt[i] > 0.0f) ? da[i] : 0.0f;
                            ts0[i] += v;
                            ts1[i] = fmaf(v, (xc[i] - ilo[i]) * iiv[i], ts1[i]);      // v * xhat (the mean as hi + lo: bn_center)
                            op[i] = v;
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&o), r_dx,
                                                           ownP(t) ? (unsigned)((r * W + colP(t)) * C + gq) * 4u : gbytes, 0, YUNET_DX_AUX);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // K = pixels: k-step s covers pixels 4 s .. 4 s + 3; lane (g, l15) supplies a[4 s + g][ci = l15], dp[4 s + g][co = l15]
                f32x4 gwb = {0.f, 0.f, 0.f, 0.f};           // (two chains: a dependent matrix instruction waits out its predecessor)
#pragma unroll
                for (int s = 0; s < PXW / 4; s += 2) {
                    const int at = (4 * s + g) * PST + l15;
                    gw1 = mfma16(slot_a[at], slot_d[at], gw1);
                    gwb = mfma16(slot_a[at + 4 * PST], slot_d[at + 4 * PST], gwb);
                }
                gw1 += gwb;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // slot A is free for the next p row
            __builtin_amdgcn_wave_barrier();
        };
        // rows y0 - 2, y0 - 1: nothing complete yet (p and the first taps of z only); dy(y0 - 1) is needed next
        pz_phase(y0 - 2);
        ring = ring == XRING - 1 ? 0 : ring + 1;
        pz_phase(y0 - 1);
        ring = ring == XRING - 1 ? 0 : ring + 1;
        issue_dy(y0 - 1);
#pragma unroll 1
        for (int X = y0; X <= y1 + 1; ++X) {
            pz_phase(X);
            ring = ring == XRING - 1 ? 0 : ring + 1;          // now the slot of x(X - 2) (written two steps ago)
            grad_phase(X);
            if (X >= y0 + 2) row_phase(X);
        }
        // ---- producer's BN-backward sums of the band: over the 16 pixels of a lane group, then fp64 in LDS ---------
        if (bn_in) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = ts0[i], b = ts1[i];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
                if (l15 == 0) {
                    __hip_atomic_fetch_add(&s_st[4 * g + i], (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_st[C + 4 * g + i], (double)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }

    // ============ flush: one partial row per workgroup ============================================================
    __syncthreads();
    float* red = sm + wid * WAVE_F;         // (aliases the wave's row slots: it is past its last row)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[l15 * C + 4 * g + i] = gw1[i];          // dW1[co = l15][ci = 4 g + i]
    {
        // layout C: sum over the four pixel segments of a channel
        float b1v = gb1, b2v = gb2;
        b1v += __shfl_xor(b1v, 16, 64); b1v += __shfl_xor(b1v, 32, 64);
        b2v += __shfl_xor(b2v, 16, 64); b2v += __shfl_xor(b2v, 32, 64);
        if (g == 0) {
            red[C * C + l15] = b1v;
            red[C * C + C + C * 9 + l15] = b2v;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float v = gw2[k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) red[C * C + C + l15 * 9 + k] = v;
        }
    }
    __syncthreads();
    float* row = d.wgrad_partials + (size_t)blockIdx.x * WROW;
    for (int i = tid; i < WROW; i += NTHR) {
        float v = 0.0f;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) v += sm[wv * WAVE_F + i];
        row[i] = v;
    }
    if (bn_in && d.in_bn.bstats && tid < 2 * C) atomic_add_f64(bn_slot(d.in_bn.bstats, d.in_bn.slots, C) + tid, s_st[tid]);
}

// rows per band: minimise (tasks per wave, rounded up) x (rows of a task incl. the four halo rows + set-up)
int bwd16s_rows(int N, int H, int W, int waves) {
    const int forced = yunet_options().bwd16s_rows;
    if (forced > 0) return forced < H ? forced : H;
    const long long strips = (W + b16s::OUTW - 1) / b16s::OUTW;
    int best = H;
    long long best_cost = -1;
    for (int R = 4; R <= H; ++R) {
        const long long tasks = (long long)N * strips * ((H + R - 1) / R);
        const long long cost = ((tasks + waves - 1) / waves) * (R + 6);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && R > best)) { best_cost = cost; best = R; }
    }
    return best;
}

template <bool POOLDY>
int launch_bwd16s(const YunetDP* d, hipStream_t stream) {
    constexpr size_t smem = (size_t)b16s::SMEM_F * 4;
    static PerDevice attr_set;      // per device (common.h)
    if (per_device(attr_set, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(dp_bwd16s_kernel<POOLDY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess ? 1 : -1;
        }) < 0)
        return YUNET_EINVAL;
    if ((long long)d->H * d->W * 16 * 4 >= (1ll << 31)) return YUNET_EINVAL;      // 32-bit byte offsets per image
    const int grid = d->wgrad_blocks;                                             // every workgroup writes its partial row
    const int R = bwd16s_rows(d->N, d->H, d->W, grid * b16s::WAVES);
    hipLaunchKernelGGL(dp_bwd16s_kernel<POOLDY>, dim3(grid), dim3(b16s::NTHR), smem, stream, *d, R);
    return hip_status();
}

}  // namespace

// conv_bwd.hip's dispatcher: the 16 -> 16 unit on maps of the big-tile class, followed by BatchNorm, with a plain
// (non-accumulating) dx.  d->z is NOT read: it must be this unit's forward output for d->x and the weights.  Compiled once
// per activation storage type: with bf16 storage x is read as bf16 and z is recomputed from it in fp32 -- the UNROUNDED z
// the BatchNorm sums were taken from (the tile kernel reads the rounded one).
int ACT_SUFFIX(launch_dp_bwd16s)(const YunetDP* d, hipStream_t stream) {
    if (d->x_dtype != YUNET_ACT_DTYPE) return YUNET_EINVAL;
    return d->pool_idx ? launch_bwd16s<true>(d, stream) : launch_bwd16s<false>(d, stream);
}
