// conv_fwd64.hip -- forward of the plain 64 -> 64 ConvDPUnit (yunet_layer.py:30-36) as a WAVE-STREAMING kernel.
//
// Round-4 measurements of the tile kernel (dp_fwd_kernel<64,64,8,16>, conv_fwd.hip; profiles/r04_fwd_abl_before.log,
// r04_stall_fwd80.json): its three phases are ADDITIVE (load + stage 0.077 ms, pointwise GEMM +0.083, depthwise +0.070 of
// a 0.249 ms 80 x 80 launch), a wave is issuing only 42 % of its cycles (32 % parked at barriers / waitcnt, 25 %
// dependency stalls) and no pipe is busy more than a quarter: two waves per SIMD walking through barrier-separated
// phases cannot hide each other's waits.  That kernel also multiplies a haloed 10 x 18 tile padded to 192 pixels for
// 128 outputs (x1.5 of the GEMM, split and epilogue work).
//
// This kernel has NO workgroup barrier in its main loop and keeps 12 waves per CU resident:
//   * a WAVE owns a column strip of an image -- 14 output columns = 16 input columns with the halo = ONE 16-pixel
//     matrix tile per image row -- and streams down a band of rows.  Per input row it
//       - takes the row's 16 pixels x 64 channels straight from global memory INTO THE MATRIX OPERAND LAYOUT
//         (lane (g, l15): pixel l15, channels 32 kb + 8 g .. + 7): no staging of the input through LDS at all;
//       - applies the producer's BatchNorm + ReLU (one FMA + one median-of-three per element: the lane's
//         out-of-image columns get the clamp [0, 0]), splits the fp32 values EXACTLY into three bf16 pieces
//         and multiplies with the three bf16 planes of W1 (LDS, shared by the workgroup): six
//         v_mfma_f32_16x16x32_bf16 per product block, dropped terms <= 2^-24 -- the same arithmetic as the tile
//         kernel, with the weights as the A operand, so that a lane ends up with FOUR CONSECUTIVE output channels
//         of one pixel and the pointwise row goes to the wave's private LDS slot in 16-byte stores;
//       - runs the depthwise 3 x 3 in ACCUMULATE form: lane (cq, cgrp) owns channels 4 cq .. + 3 of the strip
//         columns cgrp, cgrp + 4, .., reads the three neighbours of each from the slot and adds the row's
//         contribution to the three output rows it touches; the oldest of them is complete and is stored.
//     Only one pointwise row exists at a time (4.3 KB of LDS per wave); the rolling window lives in 32 registers.
//   * the pointwise GEMM runs on (rows + 2) x 16 pixels per band of `rows` x 14 outputs: x1.26 at 20-row bands
//     instead of x1.5.
// BN statistics: fp32 per lane over a band, fp64 in LDS across bands, one fp64 atomic per channel and workgroup.
//
// Compiled twice (common.h: act_t).  -DYUNET_ACT_BF16 (BASELINE configs[2], "bf16 fwd / fp32 grads"): x, z and the pooled
// winners are stored as bf16 and the pointwise product is ONE bf16 matrix instruction per block on a = bf16(relu(bn(x)))
// and bf16(W1) (fp32 accumulation) instead of the exact three-way split -- the arithmetic of the bf16 tile kernel
// (conv_fwd.hip); depthwise, bias and BN sums (of the unrounded values) stay fp32.
#include "common.h"
#ifndef YUNET_FWD64_X_AUX       // cache-policy bits of the x loads (round 6 measurement switch; 2 = non-temporal)
#define YUNET_FWD64_X_AUX 0
#endif

namespace {
namespace f64s {
constexpr int C = 64, TW = 14, PX = 16, LS = 68, NCH = 8, WAVES = 4, NTHR = 64 * WAVES;
constexpr int SPX = PX + 2;                                  // slot pixels: the row + two zero pad pixels (columns 14, 15 of a lane's
                                                            // fourth depthwise column read them instead of a neighbour's slot)
constexpr size_t W1_BYTES = (size_t)3 * C * C * 2;          // bf16 planes h | m | l, [co][ci], 16-byte chunks XOR-swizzled
constexpr size_t SLOT_FLOATS = (size_t)SPX * LS;
constexpr size_t SMEM = W1_BYTES + (WAVES * SLOT_FLOATS + 9 * C + C + C + C + 2 * C) * 4 + 2 * C * 8;
}  // namespace f64s

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// -DF64S_PROF (tools/ubench/build_ab.sh): s_memtime stamps per image row, summed per wave into YunetDP.prof
// [(block * 4 + wave) * 4 + i]: i = 0 wait for the row's loads | 1 transform + split | 2 matrix phase + p store |
// 3 depthwise + z stores.  The stamps serialise the phases (each drains the LDS counter): read them as an upper bound.
#ifdef F64S_PROF
#define F64S_STAMP(i)                                                                    \
    {                                                                                    \
        if ((i) == 0) { pt = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
        const unsigned long long now_ = __builtin_readcyclecounter();                    \
        pc[i] += now_ - pt;                                                              \
        pt = now_;                                                                       \
    }
#else
#define F64S_STAMP(i)
#endif

__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// POOL (YunetDP.pool_out): besides z, the raw winner of every 2 x 2 window after BN + ReLU and its window position
// (conv_fwd.hip: dp_fwd_kernel<..., POOL> -- the same rule: maximum for gamma > 0, minimum for gamma < 0, first element
// for gamma == 0, ties to the smaller position).  A lane owns column PAIRS, so a window needs no cross-lane step: the
// even row of a pair waits in registers for the odd one (16 more registers: this instance runs two waves per SIMD).
// `first` / `nblk`: this workgroup's number inside the grid of ITS unit and the size of that grid (a launch may carry
// the grids of several independent units one after the other: dp_fwd64s_group_kernel below)
template <bool POOL>
__device__ __forceinline__ void dp_fwd64s_body(const YunetDP& d, const int R, const int first, const int nblk) {
    using namespace f64s;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* s_w1p = reinterpret_cast<__bf16*>(smem);
    float* s_p = reinterpret_cast<float*>(smem + W1_BYTES);          // [WAVES][PX][LS] pointwise row of each wave
    float* s_w2 = s_p + WAVES * SLOT_FLOATS;                          // [9][C]
    float* s_b2 = s_w2 + 9 * C;                                       // [C]
    float* s_sc = s_b2 + C;                                           // a = med3(x * sc + sh, floor, cap)
    float* s_sh = s_sc + C;
    float* s_b1 = s_sh + C;                                           // [2][C]: zeros | pointwise bias
    double* s_st = reinterpret_cast<double*>(s_b1 + 2 * C);           // [2 C] sum | sum of squares
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int H = d.H, W = d.W;
    const bool bn_in = d.in_transform == YUNET_T_BNRELU;

    // ---- prologue: tables ---------------------------------------------------------------------------------
    staged_table<C * C, NTHR>(d.w_pw, tid, [&](int i, float w) {
        const int co = i / C, ci = i % C;
        const __bf16 h = (__bf16)w;
        const int at = co * C + (((ci >> 3) ^ (co & (NCH - 1))) << 3) + (ci & 7);
        s_w1p[at] = h;
#ifndef YUNET_ACT_BF16
        const float r1 = w - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        s_w1p[C * C + at] = m; s_w1p[2 * C * C + at] = l;
#endif
    });
    staged_table<C * 9, NTHR>(d.w_dw, tid, [&](int i, float w) { s_w2[(i % 9) * C + i / 9] = w; });
    if (tid < C) {
        s_b2[tid] = d.b_dw[tid];
        s_b1[tid] = 0.0f;
        s_b1[C + tid] = d.b_pw[tid];
        float sc = 1.0f, sh = 0.0f;
        if (bn_in) {
            const BNCoef k = bn_coef(d.in_bn, C, tid);
            sc = k.scale;
            // beta - mean * scale with the mean carried as (hi, lo)
            sh = (float)((double)k.beta - ((double)k.mean + (double)k.mean_lo) * (double)k.scale);
        }
        s_sc[tid] = sc; s_sh[tid] = sh;
    }
    if (tid < 2 * C) s_st[tid] = 0.0;
    for (int i = tid; i < WAVES * 2 * LS; i += NTHR) s_p[(i / (2 * LS)) * SLOT_FLOATS + PX * LS + i % (2 * LS)] = 0.0f;
    __syncthreads();

    const float relu_floor = bn_in ? 0.0f : -__builtin_inff();
    float sg[4] = {1.f, 1.f, 1.f, 1.f};      // fused pooling: which raw value wins a window after BN + ReLU
    if constexpr (POOL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gm = d.out_bn.gamma[(lane & 15) * 4 + i];
            sg[i] = gm > 0.0f ? 1.0f : (gm < 0.0f ? -1.0f : 0.0f);
        }
    }
    // prof < 64 is a debug ablation mask (tools/ubench: ABL), not a pointer: 1 skip the matrix instructions,
    // 2 skip the depthwise phase, 4 skip the z stores, 8 skip the input loads (results are then wrong)
#ifdef F64S_PROF
    const unsigned abl = 0;
    unsigned long long pc[4] = {0, 0, 0, 0}, pt = 0;
#else
    const unsigned abl = (unsigned)(unsigned long long)d.prof;
#endif
    float* pslot = s_p + wid * SLOT_FLOATS;
    const int strips = (W + TW - 1) / TW, bands = (H + R - 1) / R;
    const int tasks_img = strips * bands, ntasks = d.N * tasks_img;
    const unsigned xbytes = (unsigned)(H * W * C) * ACT_B;        // x and z: [H][W][64] in the activation storage type
    const int cq = l15, cgrp = g;                       // depthwise role of the lane: channel quad, column group
    const int total_waves = nblk * WAVES;

    for (int t = first * WAVES + wid; t < ntasks; t += total_waves) {
        const int n = t / tasks_img, rr = t - n * tasks_img;
        const int band = rr / strips, strip = rr - band * strips;
        const int y0 = band * R, y1 = (y0 + R < H) ? y0 + R : H;
        const int xs = strip * TW;                                         // first output column of the strip
        const int col = xs - 1 + l15;                                      // matrix role: input column of this lane
        const bool colv = (unsigned)col < (unsigned)W;
        const float cap = colv ? __builtin_inff() : 0.0f;
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + (size_t)n * d.x_img_stride, 0, xbytes, 0x00020000);
        const auto r_z = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(d.z) + (size_t)n * d.z_img_stride, 0, xbytes, 0x00020000);
        const unsigned xlane = colv ? (unsigned)(col * C + 8 * g) * ACT_B : xbytes;       // out of the image: reads 0
        const float* biasp = s_b1 + (colv ? C : 0) + 4 * g;
        // depthwise columns of this lane: two adjacent pairs, strip-local c = 2 cgrp + 8 (j >> 1) + (j & 1); their store
        // offsets and masks (a pair shares two of its three input columns: 8 LDS reads per row instead of 12)
        unsigned zlane[4];
        float fm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 2 * cgrp + 8 * (j >> 1) + (j & 1), xc = xs + c;
            const bool v = c < TW && xc < W;
            zlane[j] = v ? (unsigned)(xc * C + cq * 4) * ACT_B : xbytes;     // dropped by the range check
            fm[j] = v ? 1.0f : 0.0f;
        }
        // fused pooling: window (row pair, column pair q) of this lane -> element offset in pool_out / pool_idx
        const unsigned poel = (unsigned)((H >> 1) * (W >> 1) * C), pobytes = poel * ACT_B;
        const auto r_po = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(d.pool_out) + (POOL ? (size_t)n * poel : (size_t)0), 0,
                                                            POOL ? pobytes : 0u, 0x00020000);
        const auto r_pi = __builtin_amdgcn_make_buffer_rsrc(d.pool_idx + (POOL ? (size_t)n * poel : (size_t)0), 0,
                                                            POOL ? poel : 0u, 0x00020000);
        unsigned plane[2] = {0, 0};
        if constexpr (POOL) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * cgrp + 8 * q, xc = xs + c;
                plane[q] = (c < TW && xc < W) ? (unsigned)((xc >> 1) * C + cq * 4) : poel;     // elements; dropped when out of range
            }
        }
        const int rs = y0 > 0 ? y0 - 1 : 0, re = y1 < H ? y1 : H - 1;
        u32x4 xr[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
        auto issue = [&](int y) {
            if (abl & 8) return;
            const unsigned o = xlane + (unsigned)(y * W * C) * ACT_B;
#ifdef YUNET_ACT_BF16
            xr[0] = __builtin_amdgcn_raw_buffer_load_b128(r_x, o, 0, YUNET_FWD64_X_AUX);          // 8 bf16 channels of block 0
            xr[2] = __builtin_amdgcn_raw_buffer_load_b128(r_x, o + 64, 0, YUNET_FWD64_X_AUX);     // ... of block 1
#else
            xr[0] = __builtin_amdgcn_raw_buffer_load_b128(r_x, o, 0, YUNET_FWD64_X_AUX);
            xr[1] = __builtin_amdgcn_raw_buffer_load_b128(r_x, o + 16, 0, YUNET_FWD64_X_AUX);
            xr[2] = __builtin_amdgcn_raw_buffer_load_b128(r_x, o + 128, 0, YUNET_FWD64_X_AUX);
            xr[3] = __builtin_amdgcn_raw_buffer_load_b128(r_x, o + 144, 0, YUNET_FWD64_X_AUX);
#endif
        };
        issue(rs);
        const float4 b2 = *reinterpret_cast<const float4*>(s_b2 + cq * 4);
        float4 oa[4], ob[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) oa[j] = ob[j] = b2;
        float ts[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ts[i] = 0.0f;

        // a finished output row: z to HBM, BN partial sums, fused pooling
        float4 prev[4];
        auto emit_row = [&](int yy, const float4 (&dn)[4]) {
            const unsigned zrow = (unsigned)(yy * W * C) * ACT_B;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!(abl & 4)) act_bufst4(r_z, zrow + zlane[j], dn[j]);
                const float m = fm[j];
                const float vx = dn[j].x * m, vy = dn[j].y * m, vz = dn[j].z * m, vw = dn[j].w * m;
                ts[0] += vx; ts[1] += vy; ts[2] += vz; ts[3] += vw;
                ts[4] = fmaf(vx, vx, ts[4]); ts[5] = fmaf(vy, vy, ts[5]);
                ts[6] = fmaf(vz, vz, ts[6]); ts[7] = fmaf(vw, vw, ts[7]);
            }
            if constexpr (POOL) {
                if ((yy & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) prev[j] = dn[j];
                } else {
                    const unsigned prow = (unsigned)((yy >> 1) * (W >> 1) * C);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float tl[4] = {prev[2 * q].x, prev[2 * q].y, prev[2 * q].z, prev[2 * q].w};
                        const float bl[4] = {dn[2 * q].x, dn[2 * q].y, dn[2 * q].z, dn[2 * q].w};
                        const float tr[4] = {prev[2 * q + 1].x, prev[2 * q + 1].y, prev[2 * q + 1].z, prev[2 * q + 1].w};
                        const float br[4] = {dn[2 * q + 1].x, dn[2 * q + 1].y, dn[2 * q + 1].z, dn[2 * q + 1].w};
                        float v[4];
                        unsigned jw = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool lowl = bl[i] * sg[i] > tl[i] * sg[i], lowr = br[i] * sg[i] > tr[i] * sg[i];
                            const float vl = lowl ? bl[i] : tl[i], vr = lowr ? br[i] : tr[i];
                            const unsigned jl = lowl ? 2u : 0u, jr = lowr ? 3u : 1u;
                            const float kl = vl * sg[i], kr = vr * sg[i];
                            const bool right = kr > kl || (kr == kl && jr < jl);
                            v[i] = right ? vr : vl;
                            jw |= (right ? jr : jl) << (8 * i);
                        }
                        const unsigned eq = plane[q] == poel ? poel : prow + plane[q];
                        act_bufst4(r_po, eq * ACT_B, make_float4(v[0], v[1], v[2], v[3]));
                        __builtin_amdgcn_raw_buffer_store_b32(jw, r_pi, eq, 0, 0);
                    }
                }
            }
        };

#pragma unroll 1
        for (int r = rs; r <= re; ++r) {
            F64S_STAMP(0)
            // ---- a = T(x), split into three bf16 pieces: the B operands (k = 8 channels of this lane, n = pixel)
            u32x4 bh[2], bm[2], bl[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float4 s0 = *reinterpret_cast<const float4*>(s_sc + 32 * kb + 8 * g);
                const float4 s1 = *reinterpret_cast<const float4*>(s_sc + 32 * kb + 8 * g + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(s_sh + 32 * kb + 8 * g);
                const float4 h1 = *reinterpret_cast<const float4*>(s_sh + 32 * kb + 8 * g + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                float a[8];
#ifdef YUNET_ACT_BF16
#pragma unroll
                for (int q = 0; q < 4; ++q) {          // dword q of the load: channels 2 q (low half), 2 q + 1 (high half)
                    const unsigned w2 = xr[2 * kb][q];
                    a[2 * q] = med3(fmaf(__uint_as_float(w2 << 16), sc[2 * q], sh[2 * q]), relu_floor, cap);
                    a[2 * q + 1] = med3(fmaf(__uint_as_float(w2 & 0xffff0000u), sc[2 * q + 1], sh[2 * q + 1]), relu_floor, cap);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bh[kb][q] = pack_bf16x2(a[2 * q], a[2 * q + 1]);        // a = bf16(relu(bn(x))): the one operand of this build
                    bm[kb][q] = bl[kb][q] = 0u;
                }
#else
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = med3(fmaf(__uint_as_float(xr[2 * kb][e]), sc[e], sh[e]), relu_floor, cap);
                    a[4 + e] = med3(fmaf(__uint_as_float(xr[2 * kb + 1][e]), sc[4 + e], sh[4 + e]), relu_floor, cap);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = a[2 * q], a1 = a[2 * q + 1];
                    const unsigned hb = pack_bf16x2(a0, a1);
                    const float r0 = a0 - __uint_as_float(hb << 16), r1 = a1 - __uint_as_float(hb & 0xffff0000u);
                    const unsigned mb = pack_bf16x2(r0, r1);
                    const unsigned lb = pack_bf16x2(r0 - __uint_as_float(mb << 16), r1 - __uint_as_float(mb & 0xffff0000u));
                    bh[kb][q] = hb; bm[kb][q] = mb; bl[kb][q] = lb;
                }
#endif
            }
            F64S_STAMP(1)
            // the registers are free again: the next row is in flight under this row's arithmetic
            if (r < re) issue(r + 1);

            // ---- pointwise: p^T[co][pixel] = W1 . a^T + b1 (zero in the lane's out-of-image columns)
            // eight groups (kb, nt) of six matrix instructions; the weight fragments of group i + 1 are read from LDS
            // while group i multiplies (as one block the compiler waited out an LDS latency in front of every group)
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 bv = *reinterpret_cast<const float4*>(biasp + 16 * nt);
                acc[nt] = f32x4{bv.x, bv.y, bv.z, bv.w};
            }
            auto wfrag = [&](int i, u32x4 (&f)[3]) {
                const int kb = i >> 2, co = (i & 3) * 16 + l15;
                const __bf16* wp = s_w1p + co * C + (((4 * kb + g) ^ (co & (NCH - 1))) << 3);
                f[0] = *reinterpret_cast<const u32x4*>(wp);
#ifdef YUNET_ACT_BF16
                f[1] = f[2] = f[0];
#else
                f[1] = *reinterpret_cast<const u32x4*>(wp + C * C);
                f[2] = *reinterpret_cast<const u32x4*>(wp + 2 * C * C);
#endif
            };
            if (!(abl & 1)) {
                u32x4 wf[2][3];
                wfrag(0, wf[0]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 7) wfrag(i + 1, wf[(i + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int kb = i >> 2, nt = i & 3;
                    const bf16x8_t Bh = __builtin_bit_cast(bf16x8_t, bh[kb]), Bm = __builtin_bit_cast(bf16x8_t, bm[kb]),
                                   Bl = __builtin_bit_cast(bf16x8_t, bl[kb]);
                    const bf16x8_t Wh = __builtin_bit_cast(bf16x8_t, wf[i & 1][0]), Wm = __builtin_bit_cast(bf16x8_t, wf[i & 1][1]),
                                   Wl = __builtin_bit_cast(bf16x8_t, wf[i & 1][2]);
                    f32x4 c = acc[nt];                                   // small terms first
#ifndef YUNET_ACT_BF16
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh, Bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wl, Bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wm, Bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh, Bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wm, Bh, c, 0, 0, 0);
#endif
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh, Bh, c, 0, 0, 0);
                    acc[nt] = c;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the slot's previous row has been read (LDS executes one wave's accesses in order; the fences keep the
            // compiler from moving accesses across)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)        // lane: pixel l15, channels 16 nt + 4 g .. + 3
                *reinterpret_cast<f32x4*>(pslot + l15 * LS + 16 * nt + 4 * g) = acc[nt];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            F64S_STAMP(2)
            // ---- depthwise, accumulate form: this row is the bottom tap row of output row r - 1 (now complete), the
            //      middle of r, the top of r + 1
            if (abl & 2) continue;
            float4 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const float4*>(s_w2 + k * C + cq * 4);
            float4 dn[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* pp = pslot + (2 * cgrp + 8 * q) * LS + cq * 4;
                float4 pv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const float4*>(pp + k * LS);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * q + h;
                    const float4 pl = pv[h], pm = pv[h + 1], pr = pv[h + 2];
                    float4 o = oa[j], mid = ob[j], top = b2;
#define TAP(o, k, v)                                                                        \
    o.x = fmaf(v.x, w[k].x, o.x); o.y = fmaf(v.y, w[k].y, o.y); o.z = fmaf(v.z, w[k].z, o.z); \
    o.w = fmaf(v.w, w[k].w, o.w);
                    TAP(o, 6, pl) TAP(o, 7, pm) TAP(o, 8, pr)
                    TAP(mid, 3, pl) TAP(mid, 4, pm) TAP(mid, 5, pr)
                    TAP(top, 0, pl) TAP(top, 1, pm) TAP(top, 2, pr)
#undef TAP
                    dn[j] = o; oa[j] = mid; ob[j] = top;
                }
            }
            if (r - 1 >= y0) emit_row(r - 1, dn);
            F64S_STAMP(3)
        }
        if (y1 == H) emit_row(H - 1, oa);      // the image's last row has no row below it: complete as it stands
        // ---- BN partial sums of the band: across the four column groups, then fp64 in LDS
        if (d.out_has_bn) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = ts[i];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                ts[i] = v;
            }
            if (cgrp == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __hip_atomic_fetch_add(&s_st[cq * 4 + i], (double)ts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_st[C + cq * 4 + i], (double)ts[4 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
#ifdef F64S_PROF
    if (d.prof && lane == 0)
        for (int i = 0; i < 4; ++i) d.prof[(blockIdx.x * 4 + wid) * 4 + i] = pc[i];
#endif
    if (d.out_has_bn) {
        __syncthreads();
        if (tid < 2 * C) atomic_add_f64(bn_slot(d.out_bn.stats, d.out_bn.slots, C) + tid, s_st[tid]);
    }
}

template <bool POOL>
__global__ __launch_bounds__(f64s::NTHR, POOL ? 2 : 3) void dp_fwd64s_kernel(const YunetDP d, const int R) {
    dp_fwd64s_body<POOL>(d, R, first_tile(), (int)gridDim.x);
}

// Several mutually independent plain units in ONE grid (the share convs of the three pyramid levels: yunet_head.py:175-247
// walks the levels in a Python loop, nothing connects them).  Workgroups start[i] .. start[i + 1] - 1 belong to unit i and
// do exactly what they would do in a launch of their own (same band height, same grid size, same order of the additions):
// what the group saves is the launch boundaries -- on the 20 x 20 / 10 x 10 levels a launch is a prologue, one or two
// bands per wave and a drain (27 / 13 us for 13 + 3 MB), and the chip idles in between.
struct Fwd64sGroup {
    YunetDP d[YUNET_DP_GROUP_MAX];
    int R[YUNET_DP_GROUP_MAX];
    int start[YUNET_DP_GROUP_MAX + 1];
};
__global__ __launch_bounds__(f64s::NTHR, 3) void dp_fwd64s_group_kernel(const Fwd64sGroup m) {
    const int b = (int)blockIdx.x;
    int u = 0;
#pragma unroll
    for (int i = 1; i < YUNET_DP_GROUP_MAX; ++i) u = b >= m.start[i] ? i : u;
    const int base = m.start[u], nblk = m.start[u + 1] - base;
    // the XCD-contiguous renumbering of first_tile(), inside the unit's own grid (hardware deals workgroup ids to the
    // 8 XCDs round-robin: valid where the grid starts at a multiple of 8 and has a multiple of 8 workgroups)
    const int l = b - base;
    const int first = (YUNET_XCD_REMAP && (nblk & 7) == 0 && (base & 7) == 0) ? (l & 7) * (nblk >> 3) + (l >> 3) : l;
    dp_fwd64s_body<false>(m.d[u], m.R[u], first, nblk);
}

}  // namespace

// rows per band: the tallest bands (least row-halo recompute) that still give every resident wave a band
// (measured, N = 256: 80 x 80 40 rows 0.194 ms / 20 rows 0.201 / 10 rows 0.211; 40 x 40 10 rows 0.058 / 20 rows 0.061 /
// 40 rows 0.071 -- half the waves idle)
static int fwd64s_rows(int N, int H, int W, int waves) {
    const int forced = yunet_options().fwd64s_rows;
    if (forced > 0) return forced < H ? forced : H;
    const int strips = (W + f64s::TW - 1) / f64s::TW;
    int R = H;
    while (R > 8 && (long long)N * strips * ((H + R - 1) / R) < (long long)waves) R = (R + 1) / 2;
    // the 20 x 20 / 10 x 10 levels (one or two strips per image): shorter bands while fewer than a wave per SIMD
    // would have work (20 x 20: 5 rows 0.0267 ms, 3 rows 0.0369; 10 x 10: 5 rows 0.0140, 3 rows 0.0132)
    while (R > 2 && (long long)N * strips * ((H + R - 1) / R) < (long long)waves / 4) R = (R + 1) / 2;
    return R;
}

// grid and band height of one unit (the same whether it is launched alone or inside a group)
static int fwd64s_geometry(const YunetDP* d, int blocks_per_cu, bool pool, int* R_out) {
    if ((long long)d->H * d->W * 64 * 4 >= (1ll << 31)) return -1;                // 32-bit byte offsets per image
    int grid = yunet_cu_count() * blocks_per_cu;
    int R = fwd64s_rows(d->N, d->H, d->W, grid * f64s::WAVES);
    if (pool && (R & 1)) ++R;                                                      // bands hold whole row pairs
    const long long tasks = (long long)d->N * ((d->W + f64s::TW - 1) / f64s::TW) * ((d->H + R - 1) / R);
    const long long need = (tasks + f64s::WAVES - 1) / f64s::WAVES;
    if (need < grid) grid = (int)need;
    *R_out = R;
    return grid;
}

template <bool POOL>
static int launch_fwd64s(const YunetDP* d, hipStream_t stream) {
    static PerDevice per_cu;        // resident workgroups per CU, per device (common.h)
    const int blocks_per_cu = per_device(per_cu, [] {
        const void* fn = reinterpret_cast<const void*>(dp_fwd64s_kernel<POOL>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f64s::SMEM) != hipSuccess) return -1;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, f64s::NTHR, f64s::SMEM) != hipSuccess || nb < 1) nb = 1;
        return nb > 3 ? 3 : nb;
    });
    if (blocks_per_cu < 1) return YUNET_EINVAL;
    int R = 0;
    const int grid = fwd64s_geometry(d, blocks_per_cu, POOL, &R);
    if (grid < 1) return YUNET_EINVAL;
    hipLaunchKernelGGL(dp_fwd64s_kernel<POOL>, dim3(grid), dim3(f64s::NTHR), f64s::SMEM, stream, *d, R);
    return hip_status();
}

// n <= YUNET_DP_GROUP_MAX plain (no fused pooling) 64 -> 64 units of this activation type in one grid; the caller
// (yunet_dp_fwd_group, conv_fwd.hip) has checked that every one of them would take this kernel on its own
int ACT_SUFFIX(launch_dp_fwd64s_group)(const YunetDP* const* ds, int n, hipStream_t stream) {
    static PerDevice per_cu;
    const int blocks_per_cu = per_device(per_cu, [] {
        const void* fn = reinterpret_cast<const void*>(dp_fwd64s_group_kernel);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f64s::SMEM) != hipSuccess) return -1;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, f64s::NTHR, f64s::SMEM) != hipSuccess || nb < 1) nb = 1;
        return nb > 3 ? 3 : nb;
    });
    if (blocks_per_cu < 1 || n < 1 || n > YUNET_DP_GROUP_MAX) return YUNET_EINVAL;
    Fwd64sGroup m;
    int at = 0;
    for (int i = 0; i < YUNET_DP_GROUP_MAX; ++i) {
        m.start[i] = at;
        if (i < n) {
            m.d[i] = *ds[i];
            const int g = fwd64s_geometry(ds[i], blocks_per_cu, false, &m.R[i]);
            if (g < 1) return YUNET_EINVAL;
            at += g;
        } else {
            m.d[i] = *ds[0];          // (no workgroup maps to it)
            m.R[i] = m.R[0];
        }
    }
    m.start[YUNET_DP_GROUP_MAX] = at;
    hipLaunchKernelGGL(dp_fwd64s_group_kernel, dim3(at), dim3(f64s::NTHR), f64s::SMEM, stream, m);
    return hip_status();
}

int ACT_SUFFIX(launch_dp_fwd64s)(const YunetDP* d, hipStream_t stream) {
    if (d->x_dtype != YUNET_ACT_DTYPE || d->z_dtype != YUNET_ACT_DTYPE) return YUNET_EINVAL;
    return d->pool_out ? launch_fwd64s<true>(d, stream) : launch_fwd64s<false>(d, stream);
}
