// conv_bwd.hip -- backward kernels of the YuNet conv stack on gfx950 (NHWC fp32).
//
// The reference gets these from autograd over F.conv2d / batch_norm / relu / max_pool2d /
// interpolate.  Gradient tensors exchanged between kernels are
// "grad w.r.t. the BatchNorm OUTPUT with the ReLU mask applied" (dy); the kernel that
// produces dy also accumulates the two BN-backward sums (sum dy, sum dy*xhat) in fp64, and
// the producer's backward turns dy into dz = k1*(dy - c1 - xhat*c2) while loading.
//
// dp_bwd per 8x16 tile (512 threads): dz halo -> LDS; a = T(x) -> LDS; p = a*W1^T+b1 (MFMA,
// recomputed, never stored in HBM); depthwise backward on the VALU (dp, dW2, db2, db1);
// dW1 += a^T * dp (MFMA, K = pixels, accumulators persistent across tiles);
// da = dp * W1 (MFMA); dx = da * relu-mask, BN-backward sums of the producer.
#include "common.h"

#include <type_traits>

namespace {

// cache-policy bits of the generic tile kernel's x loads on unpacked maps (x is read once: tile interior only; 2 = non-temporal)
#ifndef YUNET_BWDT_X_AUX
#define YUNET_BWDT_X_AUX 0
#endif
#define BWD_THREADS 512
#define BWD_WAVES 8
// Where the split-bf16 variant issues the next tile's global loads: 1 = right after the stage (the
// loads have the whole tile to land, 64 more live registers in the p GEMM and the VALU phase),
// 0 = after the VALU phase like the fp32 variant (whose two long GEMMs follow and cover the latency)
#ifndef DP_BWD_PF_EARLY
#define DP_BWD_PF_EARLY 0
#endif
#ifndef DP_BWD_PF_SPREAD
#define DP_BWD_PF_SPREAD 1
#endif
#ifndef DP_BWD_PF_LATE              // round 6: every piece one issue point later (the last one between the da GEMM and the mask)
#define DP_BWD_PF_LATE 0
#endif
#ifndef DP_BWD_PF_SPREAD_ALL        // also for the exact-fp32 variants (16 / 32-channel units)
#define DP_BWD_PF_SPREAD_ALL 0
#endif
// -DDP_BWD_PROF: per-workgroup phase cycle counters (tools/ubench/bwd_ab: PROF=1).  d.prof then points to
// [grid][8] uint64: cycles of wave 0 between the barriers that end stage | p GEMM | depthwise | dW1 + da |
// mask | store, summed over the workgroup's tiles.  Not in the product build (it costs registers).
#ifdef DP_BWD_PROF
#define DP_BWD_STAMP(k)                                                        \
    if (prof_on && threadIdx.x == 0) {                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();          \
        prof_acc[k] += now_ - prof_t;                                          \
        prof_t = now_;                                                         \
    }
#else
#define DP_BWD_STAMP(k)
#endif
#ifdef DP_BWD_PROF
#define DP_BWD64_STAMP(k)                                                      \
    if (prof_on && threadIdx.x == 0) {                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();          \
        s_prof[k] += now_ - s_prof[7];                                         \
        s_prof[7] = now_;                                                      \
    }
#else
#define DP_BWD64_STAMP(k)
#endif

// GEMM = 0: the three pointwise GEMMs on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32).
// GEMM = 1: split-bf16 -- every fp32 operand x is split on the fly into hi = bf16(x) and
//   lo = bf16(x - hi) (x = hi + lo to 2^-17 relative) and each product runs as the three bf16
//   MFMAs hi*hi + lo*hi + hi*lo with fp32 accumulation (v_mfma_f32_16x16x32_bf16: 16x the fp32
//   matrix rate, so 3/16 of the matrix time).  The dropped lo*lo term is <= 2^-18 relative: the
//   products carry ~1e-5 relative error instead of 6e-8.  Used for GRADIENTS only (backward of
//   the 64 -> 64 units); the forward pass stays on the exact instruction.
template <int CIN, int COUT, int TH, int TW, int GEMM = 0>
struct BwdGeom {
    static constexpr int HW_ = TW + 2, HH_ = TH + 2, HP = HH_ * HW_;
    static constexpr int IP = TH * TW, IMT = IP / 16;
    static constexpr int LSO = COUT + 4, LSI = CIN + 4, WS = CIN + 4, WST = COUT + 4;
    static constexpr int C4I = CIN / 4, C4O = COUT / 4;
    static constexpr int NTO = COUT / 16, NTI = CIN / 16;
    static constexpr int KSI = CIN / 4, KSO = COUT / 4;
    static constexpr int PG = BWD_THREADS / C4O;   // pixel groups of the VALU phase
    static constexpr int PPT = IP / PG;            // pixels (rows) per thread
    // dW1 (K = pixels): a wave owns MB x NB 16x16 tiles whose rows / columns are interleaved
    // (ci = MB*m + j, co = NB*n + i), so ONE MB-float and ONE NB-float LDS read feed MB*NB
    // MFMAs; the NGRP wave groups cover [CIN x COUT], the remaining waves split K
    static constexpr int MB = (NTI >= 2 && (NTI * NTO < 16 || GEMM == 1)) ? 2 : 1, NB = NTO >= 2 ? 2 : 1;
    static constexpr int NGRP = (NTI / MB) * (NTO / NB);
    static constexpr int KSPLIT = BWD_WAVES / NGRP;
    static constexpr int KSTEPS = (IP / 4) / KSPLIT;  // k-steps of 4 pixels per wave
    static constexpr int KPX = IP / KSPLIT;           // pixels of K per wave (dW1)
    static constexpr int NDZ = (HP * C4O + BWD_THREADS - 1) / BWD_THREADS;  // (dy,z) float4 pairs / thread
    static constexpr int NX = (IP * C4I) / BWD_THREADS;                       // x float4 / thread
    // LDS carve (floats)
    static constexpr int OFF_DZ = 0;
    static constexpr int OFF_A = OFF_DZ + HP * LSO;
    static constexpr int OFF_PB = OFF_A + IP * LSI;
    static constexpr int WORK_F = OFF_PB + IP * LSO;
    // GEMM = 1: W1 and W1^T as bf16 hi / lo planes, rows padded to WSB / WSTB elements
    static constexpr int WSB = CIN + 8, WSTB = COUT + 8;
    static constexpr int W1_F = GEMM ? (2 * COUT * WSB) / 2 + COUT : COUT * WS;     // (+ bias row)
    static constexpr int W1T_F = GEMM ? (2 * CIN * WSTB) / 2 : CIN * WST;
    static constexpr int PAR_F = W1_F + 9 * COUT + 7 * COUT + 5 * CIN + W1T_F + 4 * CIN + IP / 4;   // w1 | w2 | out-bn | in-bn | w1^T | fp64 sums | validity bytes
    static_assert(GEMM == 0 || (CIN % 32 == 0 && COUT % 32 == 0 && (IP / KSPLIT) % 32 == 0), "bf16 MFMA: K in blocks of 32");
    static constexpr int WROW = COUT * CIN + COUT + COUT * 9 + COUT;  // partial row width
    static constexpr size_t RED1 = ((size_t)BWD_WAVES / NGRP * COUT * CIN + (size_t)BWD_THREADS * 24) * 4;   // flush area: dW1 K-slice planes + dW2/db records
    static constexpr size_t WORK = (size_t)WORK_F * 4;
    static constexpr size_t WORKB = WORK > RED1 ? WORK : RED1;
    static constexpr size_t SMEM = WORKB + (size_t)PAR_F * 4;
    static constexpr int MPW = IMT / BWD_WAVES;     // 16-pixel M tiles per wave
    // the register-bound variants re-derive thread-invariant indices per tile (see opaque())
    static constexpr bool LAUNDER = COUT >= 64 || CIN * COUT >= 2048 || TH * TW > 128;
    static_assert(IP % 16 == 0 && IMT % BWD_WAVES == 0, "whole M tiles per wave");
    static_assert(PG % TW == 0 && IP % PG == 0, "VALU mapping");
    static_assert(BWD_THREADS % C4I == 0 && BWD_THREADS % C4O == 0, "load mapping");
    static_assert((IP * C4I) % BWD_THREADS == 0, "x load mapping");
    static_assert((IP / 4) % KSPLIT == 0, "k split");
    static_assert(BWD_WAVES % NGRP == 0 && NGRP <= BWD_WAVES, "dW1 wave groups");
};

// Opaque copy of a thread-invariant value: stops the compiler from hoisting everything derived
// from it (per-slot offsets, halo coordinates) out of the persistent tile loop, where those
// values would occupy dozens of VGPRs for the whole kernel.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ float tin(float x, float mean, float scale, float beta, float floor_) {
    return fmaxf(fmaf(x - mean, scale, beta), floor_);
}

// ---- split-bf16 helpers (GEMM = 1) --------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> packed bf16 pairs hi = rne(x), lo = rne(x - hi); element 0 in the low half
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x0, x1};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {x0 - __uint_as_float(hb << 16), x1 - __uint_as_float(hb & 0xffff0000u)};
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
struct Split8 {
    u32x4 hi, lo;     // 8 bf16 each: the 8 k-slots one lane feeds to v_mfma_f32_16x16x32_bf16
};
__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2(x[2 * i], x[2 * i + 1], h[i], l[i]);
    Split8 o;
    o.hi = u32x4{h[0], h[1], h[2], h[3]};
    o.lo = u32x4{l[0], l[1], l[2], l[3]};
    return o;
}
// D += A*B with A = ah + al, B = bh + bl (lo*lo dropped); small terms first
__device__ __forceinline__ f32x4 mfma3(const Split8& a, const u32x4 bh, const u32x4 bl, f32x4 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi), al = __builtin_bit_cast(bf16x8, a.lo);
    const bf16x8 vh = __builtin_bit_cast(bf16x8, bh), vl = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, vh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, vl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, vh, c, 0, 0, 0);
    return c;
}

// FULL: the map is an exact multiple of the tile (H % TH == 0, W % TW == 0: the 160 x 160 and 80 x 80 levels), so
// every interior tile pixel is a real pixel and the per-element validity tests -- hundreds of integer
// instructions per tile -- compile away.  The launch picks the instance.
template <int CIN, int COUT, int TH, int TW, bool PACKED, int GEMM = 0, bool POOLDY = false, bool FULL = false>
__global__ __launch_bounds__(BWD_THREADS) void dp_bwd_kernel(const YunetDP d, const PackGeom pk) {
    using G = BwdGeom<CIN, COUT, TH, TW, GEMM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* s_dz = sm + G::OFF_DZ;
    float* s_a = sm + G::OFF_A;
    float* s_pb = sm + G::OFF_PB;
    // the parameter/coefficient block sits after the (possibly larger) reduction work area
    float* s_w1 = reinterpret_cast<float*>(smem_raw + G::WORKB);   // [COUT][WS]  (GEMM 1: bf16 planes hi | lo [COUT][WSB], then b1[COUT])
    float* s_w2 = s_w1 + G::W1_F;                                  // [9][COUT]
    float* s_co = s_w2 + 9 * COUT;                                 // folded BN backward of the unit's own BN: A|B|Dh|Dl (+3 spare rows)
    float* s_ci = s_co + 7 * COUT;                                 // mean|scale|beta|invstd|mean_lo
    float* s_w1t = s_ci + 5 * CIN;                                 // [CIN][WST] (B operand of the da GEMM; GEMM 1: planes hi | lo [CIN][WSTB])
    double* s_bst = reinterpret_cast<double*>(s_w1t + G::W1T_F);   // [2][CIN] producer's BN-backward sums
    // GEMM 1 views of the weight block
    __bf16* s_w1h = reinterpret_cast<__bf16*>(s_w1);
    __bf16* s_w1l = s_w1h + COUT * G::WSB;
    float* s_b1 = reinterpret_cast<float*>(s_w1l + COUT * G::WSB);
    __bf16* s_w1th = reinterpret_cast<__bf16*>(s_w1t);
    __bf16* s_w1tl = s_w1th + CIN * G::WSTB;
    unsigned char* s_in = reinterpret_cast<unsigned char*>(s_bst + 2 * CIN);   // [IP] packed mode: pixel is real

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int H = d.H, W = d.W;
    const bool bn_in = d.in_transform == YUNET_T_BNRELU;
    const bool bn_out = d.out_has_bn != 0;
    // The input transform is applied branch-free where MFMA operands are read: with the identity
    // coefficients (mean 0, scale 1, beta 0) and a floor of -inf it returns its argument exactly,
    // so one straight-line GEMM body serves both kinds of unit and can be software-pipelined.
    const float relu_floor = bn_in ? 0.0f : -__builtin_inff();
    // debug ablation mask (tools/kbench.py --ablate): prof < 4096 is a bit mask, not a pointer
    const unsigned abl = (unsigned long long)d.prof < 4096ull ? (unsigned)(unsigned long long)d.prof : 0u;

#ifdef DP_BWD_PROF
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
    // per-image tiling, or (PACKED) one tile grid over the packed canvas of all images (common.h)
    const int tiles_x = ((PACKED ? pk.CW : W) + TW - 1) / TW, tiles_y = ((PACKED ? pk.CH : H) + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int ntiles = PACKED ? tiles_img : d.N * tiles_img;
    // inside(ip, Y, X): is interior tile pixel ip = (Y, X) a real pixel?  Packed tiles look it up in
    // a per-tile byte map written during the stage (one canvas -> image mapping per pixel and tile
    // instead of one per use).
    static_assert(!(FULL && PACKED), "FULL: unpacked maps only");
    auto inside = [&](int ip, int y, int x) {
        if constexpr (FULL) return true;
        else if constexpr (PACKED) return s_in[ip] != 0;
        else return y < H && x < W;
    };

    // ---- prefetch registers: raw dy / z_out (haloed) and x (interior) of the NEXT tile ------------
    // Loaded through per-image buffer descriptors: one 32-bit byte offset per slot, and a slot
    // outside the image (zero padding of the halo, ragged last tiles) simply gets an offset past
    // the end -- the hardware range check returns 0, so there are no branches and no 64-bit
    // address arithmetic.  okmask keeps one validity bit per halo slot for the stage.
    float4 pdy[G::NDZ];
    act_raw4 pz[G::NDZ], px[G::NX];          // saved activations: storage type of this build (fp32 | bf16)
    unsigned okmask = 0;
    // POOLDY (YunetDP.pool_idx): the unit's output feeds max_pool2d and nothing else.  A halo slot then
    // loads the gradient of ITS pooled element and that element's argmax bytes (4 channels = one dword);
    // the stage keeps the gradient where the slot is the window maximum.  posmask: window position
    // 2*(y&1) + (x&1) of every slot, two bits each.
    static_assert(!(POOLDY && PACKED), "pooled dy: unpacked levels only");
    static_assert(!POOLDY || G::NDZ <= 16, "posmask holds 16 slots");
    unsigned pid[POOLDY ? G::NDZ : 1];
    unsigned posmask = 0;
    const int Wq = W >> 1;
    const unsigned pooledbytes = (unsigned)((H >> 1) * Wq * COUT) * 4u;
    // dy / dx are fp32 in every build; z and x are activations
    const unsigned dybytes = (unsigned)(H * W * COUT) * 4u, zbytes = (unsigned)(H * W * COUT) * ACT_B;
    const unsigned xbytes = (unsigned)(H * W * CIN) * ACT_B, dxbytes = (unsigned)(H * W * CIN) * 4u;
    constexpr int PSTEP = BWD_THREADS / G::C4O;            // halo pixels between a thread's slots
    // issue(t, part): part -1 = everything at once; parts 0..3 = x | first | second | last third of the
    // (dy, z) slots.  A CU keeps far fewer bytes in flight than the 124 KB of a tile: issued in one go
    // the waves sit in the issue for the time the transfer takes (measured 5.8 k cycles per tile with
    // the per-phase counters); in four pieces between the phases the transfer runs under the compute.
    auto issue = [&](int t, auto part_c) {
        constexpr int PART = decltype(part_c)::value;
        const int tid = G::LAUNDER ? opaque((int)threadIdx.x) : (int)threadIdx.x;
        const int och4 = tid % G::C4O, ich4 = tid % G::C4I;
        const int n = PACKED ? 0 : t / tiles_img, rr = t - n * tiles_img;
        const int y0 = (rr / tiles_x) * TH, x0 = (rr % tiles_x) * TW;      // canvas coordinates if PACKED
        // packed: descriptors over the whole tensors, the image index is part of the offset
        const unsigned dyrange = PACKED ? (unsigned)d.N * (unsigned)d.z_img_stride * 4u : dybytes;
        const unsigned zrange = PACKED ? (unsigned)d.N * (unsigned)d.z_img_stride * ACT_B : zbytes;
        const unsigned xrange = PACKED ? (unsigned)d.N * (unsigned)d.x_img_stride * ACT_B : xbytes;
        const size_t zbase = PACKED ? (size_t)0 : (size_t)n * d.z_img_stride;
        const size_t xbase = PACKED ? (size_t)0 : (size_t)n * d.x_img_stride;
        const auto r_dy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(d.dy) + (POOLDY ? (size_t)n * (pooledbytes / 4u) : zbase), 0, POOLDY ? pooledbytes : dyrange, 0x00020000);
        const auto r_id = __builtin_amdgcn_make_buffer_rsrc(
            d.pool_idx + (POOLDY ? (size_t)n * (pooledbytes / 4u) : (size_t)0), 0,
            POOLDY ? pooledbytes / 4u : 0u, 0x00020000);
        const auto r_z = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.z)) + zbase, 0, zrange, 0x00020000);
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + xbase, 0, xrange, 0x00020000);
        if (PART <= 0) { okmask = 0; posmask = 0; }
#pragma unroll
        for (int i = 0; i < G::NDZ; ++i) {
            if (PART >= 0 && PART != 1 + (3 * i) / G::NDZ) continue;
            const int hp = tid / G::C4O + PSTEP * i;
            const int hy = hp / G::HW_, hx = hp - hy * G::HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            bool ok;
            unsigned eo;                              // element offset of the slot
            if constexpr (PACKED) {
                int pn, py, px;
                ok = hp < G::HP && pk_locate(pk, y, x, pn, py, px);
                eo = (unsigned)(pn * d.z_img_stride + (py * W + px) * COUT + och4 * 4);
            } else {
                ok = hp < G::HP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                eo = (unsigned)((y * W + x) * COUT + och4 * 4);
            }
            okmask |= ok ? (1u << i) : 0u;
            if constexpr (POOLDY) {
                const unsigned eq = (unsigned)(((y >> 1) * Wq + (x >> 1)) * COUT + och4 * 4);   // pooled element
                posmask |= (unsigned)(((y & 1) << 1) | (x & 1)) << (2 * i);
                const u32x4 vdy = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eq * 4u : pooledbytes, 0, 0);
                pdy[i] = *reinterpret_cast<const float4*>(&vdy);
                pid[i] = __builtin_amdgcn_raw_buffer_load_b32(r_id, ok ? eq : pooledbytes, 0, 0);
            } else {
                const u32x4 vdy = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eo * 4u : dyrange, 0, 0);
                pdy[i] = *reinterpret_cast<const float4*>(&vdy);
            }
            pz[i] = act_raw4{};
            if (bn_out) pz[i] = act_bufld4(r_z, ok ? eo * ACT_B : zrange);
            // packed: finish one slot's address arithmetic before the next one starts (otherwise
            // all 16 canvas -> image mappings are computed up front and spill)
            if constexpr (PACKED) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < G::NX; ++i) {
            if (PART > 0) continue;
            const int ip = (tid + BWD_THREADS * i) / G::C4I;
            const int y = y0 + ip / TW, x = x0 + ip % TW;
            unsigned off;
            if constexpr (PACKED) {
                int pn, py, px;
                off = pk_locate(pk, y, x, pn, py, px)
                          ? (unsigned)(pn * d.x_img_stride + (py * W + px) * CIN + ich4 * 4) * ACT_B : xrange;
            } else {
                off = (FULL || (y < H && x < W)) ? (unsigned)((y * W + x) * CIN + ich4 * 4) * ACT_B : xbytes;
            }
            px[i] = PACKED ? act_bufld4(r_x, off) : act_bufld4_aux<YUNET_BWDT_X_AUX>(r_x, off);
            if constexpr (PACKED) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the first tile's global loads go out BEFORE the weights / coefficients are staged: the HBM
    // latency of a cold start runs under the prologue
    using All = std::integral_constant<int, -1>;
    int t = first_tile();
    if (t < ntiles) issue(t, All{});

    if constexpr (GEMM == 1) {
        // W1 -> bf16 hi / lo planes in both orientations, 8 weights per thread and pass, every global
        // load of a pass in flight at once, 16-byte LDS stores
        // (32 -> 64: 256 groups for 512 threads -- the upper half repeats the lower half's groups, same values to the same
        // addresses, instead of a divergent branch)
        static_assert(COUT * CIN <= 8 * BWD_THREADS && (8 * BWD_THREADS) % (COUT * CIN) == 0, "8-weight groups per thread");
        const int t8 = tid % (COUT * CIN / 8);
        {
            const int co = t8 / (CIN / 8), c0 = (t8 % (CIN / 8)) * 8;
            const float4 a = *reinterpret_cast<const float4*>(d.w_pw + co * CIN + c0);
            const float4 b = *reinterpret_cast<const float4*>(d.w_pw + co * CIN + c0 + 4);
            const float w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const Split8 sp = split8(w8);
            *reinterpret_cast<u32x4*>(s_w1h + co * G::WSB + c0) = sp.hi;
            *reinterpret_cast<u32x4*>(s_w1l + co * G::WSB + c0) = sp.lo;
            if (c0 == 0) s_b1[co] = d.b_pw[co];
        }
        {
            const int ci = t8 / (COUT / 8), o0 = (t8 % (COUT / 8)) * 8;
            float w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = d.w_pw[(o0 + j) * CIN + ci];
            const Split8 sp = split8(w8);
            *reinterpret_cast<u32x4*>(s_w1th + ci * G::WSTB + o0) = sp.hi;
            *reinterpret_cast<u32x4*>(s_w1tl + ci * G::WSTB + o0) = sp.lo;
        }
    } else {
        staged_table<COUT * CIN, BWD_THREADS>(d.w_pw, tid, [&](int i, float w) {       // (common.h: every load in flight first)
            s_w1[(i / CIN) * G::WS + (i % CIN)] = w;
            if (i % CIN == 0) s_w1[(i / CIN) * G::WS + CIN] = d.b_pw[i / CIN];   // bias rides in the row padding
            s_w1t[(i % CIN) * G::WST + (i / CIN)] = w;
        });
    }
    staged_table<COUT * 9, BWD_THREADS>(d.w_dw, tid, [&](int i, float w) { s_w2[(i % 9) * COUT + i / 9] = w; });
    for (int c = tid; c < COUT; c += BWD_THREADS) {
        // dz = k1 * (dy - c1 - xhat * c2) folded into dz = A dy + B z + D (see bn_fold in common.h)
        if (bn_out) {
            const BNFold f = bn_fold(bn_bwd_coef(d.out_bn, COUT, c));
            s_co[c] = f.a; s_co[COUT + c] = f.b; s_co[2 * COUT + c] = f.dh; s_co[3 * COUT + c] = f.dl;
        } else {
            s_co[c] = d.dy_scale ? d.dy_scale[c] : 1.0f;
            s_co[COUT + c] = 0.f; s_co[2 * COUT + c] = 0.f; s_co[3 * COUT + c] = 0.f;
        }
    }
    for (int c = tid; c < CIN; c += BWD_THREADS) {
        if (bn_in) {
            const BNCoef k = bn_coef(d.in_bn, CIN, c);
            s_ci[c] = k.mean; s_ci[CIN + c] = k.scale; s_ci[2 * CIN + c] = k.beta;
            s_ci[3 * CIN + c] = k.invstd; s_ci[4 * CIN + c] = k.mean_lo;
        } else {
            s_ci[c] = 0.f; s_ci[CIN + c] = 1.f; s_ci[2 * CIN + c] = 0.f; s_ci[3 * CIN + c] = 1.f;
            s_ci[4 * CIN + c] = 0.f;
        }
    }
    __syncthreads();

    // ---- per-thread constants ---------------------------------------------------------------
    // VALU phase: channel quad cq, pixel column vtx, rows vr0..vr0+PPT-1
    const int cq = tid % G::C4O, pg = tid / G::C4O;
    const int vtx = pg % TW, vr0 = (pg / TW) * G::PPT;
    // persistent accumulators (flushed once per workgroup)
    float4 gw2[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) gw2[t] = make_float4(0, 0, 0, 0);
    float4 gb2 = make_float4(0, 0, 0, 0), gb1 = make_float4(0, 0, 0, 0);
    // per-lane partials of the producer's BN-backward sums, D layout: channel nt*16 + l15.
    // fp64: sum(dy) cancels heavily and feeds c1 = mean(dy) of EVERY dz of the producer --
    // fp32 partials here showed up as 0.3 % errors in depthwise weight gradients upstream.
    // (kept in LDS, one fp64 atomic per lane / channel block / tile: as registers they cost 16
    // VGPRs for the whole kernel and pushed the 64-channel variant into scratch)
    for (int i = tid; i < 2 * CIN; i += BWD_THREADS) s_bst[i] = 0.0;
    f32x4 gw1[G::MB * G::NB];   // dW1: this wave's interleaved 16x16 tiles
#pragma unroll
    for (int i = 0; i < G::MB * G::NB; ++i) gw1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int w1_grp = wid % G::NGRP, w1_kslice = wid / G::NGRP;
    const int w1_ci0 = (w1_grp / (G::NTO / G::NB)) * 16 * G::MB;   // first ci / co of the wave's block
    const int w1_co0 = (w1_grp % (G::NTO / G::NB)) * 16 * G::NB;

    // next tile's loads in four pieces (measured: -8 % on 16->16 at 160x160, -6 % on 16->64, +2 % on the
    // 64->16 heads, which keep the single issue)
    constexpr bool SPREAD = DP_BWD_PF_SPREAD && (GEMM == 1 || DP_BWD_PF_SPREAD_ALL || COUT >= 32 || CIN == 16);
    const bool pf_on = !(abl & 32);
#ifdef DP_BWD_PROF
    const bool prof_on = (unsigned long long)d.prof >= 4096ull;
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_readcyclecounter();
    const unsigned long long prof_pro = prof_t - prof_t0;       // prologue: weights / coefficients -> LDS
#endif
    for (; t < ntiles; t += gridDim.x) {
        const int n = PACKED ? 0 : t / tiles_img, rr = t - n * tiles_img;
        const int y0 = (rr / tiles_x) * TH, x0 = (rr % tiles_x) * TW;      // canvas coordinates if PACKED

        // ---- stage: dz (BN backward of this unit's own BN) and a = T(x) -> LDS ------------------
        {
            const int tid = G::LAUNDER ? opaque((int)threadIdx.x) : (int)threadIdx.x;
            const int och4 = tid % G::C4O, ich4 = tid % G::C4I;
            const float4 o_a = *reinterpret_cast<float4*>(s_co + och4 * 4);
            const float4 o_b = *reinterpret_cast<float4*>(s_co + COUT + och4 * 4);
            const float4 o_dh = *reinterpret_cast<float4*>(s_co + 2 * COUT + och4 * 4);
            const float4 o_dl = *reinterpret_cast<float4*>(s_co + 3 * COUT + och4 * 4);
            const int hp0 = tid / G::C4O;
#pragma unroll
            for (int i = 0; i < G::NDZ; ++i) {
                const int hp = hp0 + PSTEP * i;
                if ((i + 1) * PSTEP <= G::HP || hp < G::HP) {
                    float4 dy = pdy[i];
                    const float4 z = act_unpack(pz[i]);
                    if constexpr (POOLDY) {
                        // max_pool2d backward: the pooled gradient reaches the window maximum only
                        const unsigned id = pid[i], pos = (posmask >> (2 * i)) & 3u;
                        dy.x = (id & 0xffu) == pos ? dy.x : 0.0f;
                        dy.y = ((id >> 8) & 0xffu) == pos ? dy.y : 0.0f;
                        dy.z = ((id >> 16) & 0xffu) == pos ? dy.z : 0.0f;
                        dy.w = (id >> 24) == pos ? dy.w : 0.0f;
                    }
                    // zero padding of dz: a slot outside the image loaded dy = z = 0, which the BN backward
                    // would turn into D
                    const bool ok = (okmask >> i) & 1u;
                    float4 v;
                    v.x = ok ? fmaf(o_a.x, dy.x, fmaf(o_b.x, z.x, o_dh.x)) + o_dl.x : 0.0f;
                    v.y = ok ? fmaf(o_a.y, dy.y, fmaf(o_b.y, z.y, o_dh.y)) + o_dl.y : 0.0f;
                    v.z = ok ? fmaf(o_a.z, dy.z, fmaf(o_b.z, z.z, o_dh.z)) + o_dl.z : 0.0f;
                    v.w = ok ? fmaf(o_a.w, dy.w, fmaf(o_b.w, z.w, o_dh.w)) + o_dl.w : 0.0f;
                    *reinterpret_cast<float4*>(s_dz + hp * G::LSO + och4 * 4) = v;
                }
            }
            if constexpr (PACKED) {
                for (int ip = tid; ip < G::IP; ip += BWD_THREADS) {
                    int pn, py, px;
                    s_in[ip] = pk_locate(pk, y0 + ip / TW, x0 + ip % TW, pn, py, px) ? 1 : 0;
                }
            }
            // the interior input tile goes to LDS RAW; the input transform (BN+ReLU of the
            // producer) is applied where MFMA operands are read, so the raw values stay available
            // for the ReLU mask and the BN-backward sums of the producer
#pragma unroll
            for (int i = 0; i < G::NX; ++i) {
                const int ip = (tid + BWD_THREADS * i) / G::C4I;
                *reinterpret_cast<float4*>(s_a + ip * G::LSI + ich4 * 4) = act_unpack(px[i]);
            }
        }
        __syncthreads();
        DP_BWD_STAMP(0);
        const bool more = t + (int)gridDim.x < ntiles && pf_on;
        if (SPREAD) { if (more && !DP_BWD_PF_LATE) issue(t + gridDim.x, std::integral_constant<int, 0>{}); }
        else if (GEMM == 1 && DP_BWD_PF_EARLY && more) issue(t + gridDim.x, All{});

        // ---- p = a * W1^T + b1 on the interior pixels (one M tile per wave) ---------------------
        if constexpr (GEMM == 1) {
            if (!(abl & 1)) {
#pragma unroll 1
                for (int mi = 0; mi < G::MPW; ++mi) {
                    const int mt = wid * G::MPW + mi;
                    f32x4 acc[G::NTO];
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // lane group g supplies input channels 32*kb + 8g .. +7 of pixel row l15 (A) and
                    // of weight row nt*16 + l15 (B): the same k order on both sides
                    const float* arow = s_a + (mt * 16 + l15) * G::LSI + 8 * g;
                    const float* crow = s_ci + 8 * g;
                    const __bf16* bh = s_w1h + l15 * G::WSB + 8 * g;
                    const __bf16* bl = s_w1l + l15 * G::WSB + 8 * g;
#pragma unroll
                    for (int kb = 0; kb < CIN / 32; ++kb) {
                        float a8[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float4 xv = *reinterpret_cast<const float4*>(arow + 32 * kb + 4 * h);
                            const float4 mv = *reinterpret_cast<const float4*>(crow + 32 * kb + 4 * h);
                            const float4 sv = *reinterpret_cast<const float4*>(crow + CIN + 32 * kb + 4 * h);
                            const float4 tv = *reinterpret_cast<const float4*>(crow + 2 * CIN + 32 * kb + 4 * h);
                            a8[4 * h + 0] = tin(xv.x, mv.x, sv.x, tv.x, relu_floor);
                            a8[4 * h + 1] = tin(xv.y, mv.y, sv.y, tv.y, relu_floor);
                            a8[4 * h + 2] = tin(xv.z, mv.z, sv.z, tv.z, relu_floor);
                            a8[4 * h + 3] = tin(xv.w, mv.w, sv.w, tv.w, relu_floor);
                        }
                        const Split8 as = split8(a8);
#pragma unroll
                        for (int nt = 0; nt < G::NTO; ++nt) {
                            const u32x4 vh = *reinterpret_cast<const u32x4*>(bh + nt * 16 * G::WSB + 32 * kb);
                            const u32x4 vl = *reinterpret_cast<const u32x4*>(bl + nt * 16 * G::WSB + 32 * kb);
                            acc[nt] = mfma3(as, vh, vl, acc[nt]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ip = mt * 16 + 4 * g + r;
                        const bool in = inside(ip, y0 + ip / TW, x0 + ip % TW);
#pragma unroll
                        for (int nt = 0; nt < G::NTO; ++nt)
                            s_pb[ip * G::LSO + nt * 16 + l15] = in ? acc[nt][r] + s_b1[nt * 16 + l15] : 0.0f;
                    }
                }
            }
        } else if (!(abl & 1)) {
            // 16-channel inputs: one k block per pixel tile -- the wave's pixel tiles are unrolled so that their
            // read -> transform -> 4 dependent MFMAs -> write chains overlap (rolled, each tile pays the whole latency)
#pragma unroll (CIN == 16 ? G::MPW : 1)
            for (int mi = 0; mi < G::MPW; ++mi) {
                const int mt = wid * G::MPW + mi;
                f32x4 acc[G::NTO];
#pragma unroll
                for (int nt = 0; nt < G::NTO; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // k-permuted operands: within each block of 16 input channels lane group g supplies
                // channels 4g..4g+3 -- one 16-byte LDS read feeds four MFMA k-steps (A and B use
                // the same permutation, so the sum over k is unchanged)
                const float* arow = s_a + (mt * 16 + l15) * G::LSI + 4 * g;
                const float* brow = s_w1 + l15 * G::WS + 4 * g;
                const float* crow = s_ci + 4 * g;
                // rolled loop, operands of block q+1 loaded while block q is on the matrix cores
                // (a rolled loop bounds the live operand set to two blocks -- a fully unrolled one
                // lets the scheduler hoist every read and spill)
                constexpr int NQ = CIN / 16, UNR = (NQ > 2 && G::NTO >= 4) ? 1 : NQ;
                float4 a_c = *reinterpret_cast<const float4*>(arow);
                float4 m_c = *reinterpret_cast<const float4*>(crow);
                float4 s_c = *reinterpret_cast<const float4*>(crow + CIN);
                float4 t_c = *reinterpret_cast<const float4*>(crow + 2 * CIN);
                float4 b_c[G::NTO];
#pragma unroll
                for (int nt = 0; nt < G::NTO; ++nt)
                    b_c[nt] = *reinterpret_cast<const float4*>(brow + nt * 16 * G::WS);
#pragma unroll UNR
                for (int q = 0; q < NQ; ++q) {
                    const int qn = UNR == 1 ? 16 * ((q + 1) % NQ) : (q + 1 < NQ ? 16 * (q + 1) : 0);
                    const float4 a_n = *reinterpret_cast<const float4*>(arow + qn);
                    const float4 m_n = *reinterpret_cast<const float4*>(crow + qn);
                    const float4 s_n = *reinterpret_cast<const float4*>(crow + CIN + qn);
                    const float4 t_n = *reinterpret_cast<const float4*>(crow + 2 * CIN + qn);
                    float4 b_n[G::NTO];
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt)
                        b_n[nt] = *reinterpret_cast<const float4*>(brow + nt * 16 * G::WS + qn);
                    const float ax = tin(a_c.x, m_c.x, s_c.x, t_c.x, relu_floor);
                    const float ay = tin(a_c.y, m_c.y, s_c.y, t_c.y, relu_floor);
                    const float az = tin(a_c.z, m_c.z, s_c.z, t_c.z, relu_floor);
                    const float aw = tin(a_c.w, m_c.w, s_c.w, t_c.w, relu_floor);
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt) acc[nt] = mfma16(ax, b_c[nt].x, acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt) acc[nt] = mfma16(ay, b_c[nt].y, acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt) acc[nt] = mfma16(az, b_c[nt].z, acc[nt]);
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt) acc[nt] = mfma16(aw, b_c[nt].w, acc[nt]);
                    a_c = a_n; m_c = m_n; s_c = s_n; t_c = t_n;
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt) b_c[nt] = b_n[nt];
                }
                float bias_pw[G::NTO];
#pragma unroll
                for (int nt = 0; nt < G::NTO; ++nt) bias_pw[nt] = s_w1[(nt * 16 + (G::LAUNDER ? opaque(l15) : l15)) * G::WS + CIN];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ip = mt * 16 + 4 * g + r;
                    const bool in = inside(ip, y0 + ip / TW, x0 + ip % TW);
#pragma unroll
                    for (int nt = 0; nt < G::NTO; ++nt)
                        s_pb[ip * G::LSO + nt * 16 + l15] = in ? acc[nt][r] + bias_pw[nt] : 0.0f;
                }
            }
        }
        __syncthreads();
        DP_BWD_STAMP(1);
        if (SPREAD && more) issue(t + gridDim.x, std::integral_constant<int, DP_BWD_PF_LATE ? 0 : 1>{});

        // ---- depthwise backward on the VALU; dp overwrites p in place ----------------------------
        // A thread owns a channel quad and a column of PPT rows.  The dz column triple is walked
        // once with a sliding window: each dz value is read from LDS one time and feeds every
        // output row it touches (3x fewer LDS reads than tap-by-tap when PPT = 4).
        if (!(abl & 2)) {
            // thread coordinates re-derived from an opaque copy of tid: all LDS addresses of this phase
            // become ONE per-tile base register + compile-time offsets (hoisted out of the tile loop
            // they are ~30 separate address registers, which the allocator then spills)
            const int tv = G::LAUNDER ? opaque((int)threadIdx.x) : (int)threadIdx.x;
            const int cq = tv % G::C4O, pg = tv / G::C4O;
            const int vtx = pg % TW, vr0 = (pg / TW) * G::PPT;
            const float* zb = s_dz + (vr0 * G::HW_ + vtx) * G::LSO + cq * 4;
            float* pb = s_pb + (vr0 * TW + vtx) * G::LSO + cq * 4;
            const float* wb = s_w2 + cq * 4;
            float4 pv[G::PPT], dp[G::PPT];
#pragma unroll
            for (int r = 0; r < G::PPT; ++r) {
                pv[r] = *reinterpret_cast<const float4*>(pb + r * TW * G::LSO);
                dp[r] = make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float4 wk[3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    wk[a] = *reinterpret_cast<const float4*>(wb + (8 - (3 * a + b)) * COUT);
#pragma unroll
                for (int j = 0; j < G::PPT + 2; ++j) {
                    const float4 z4 = *reinterpret_cast<const float4*>(zb + (j * G::HW_ + b) * G::LSO);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int r = j - a;
                        if (r < 0 || r >= G::PPT) continue;
                        const int k = 8 - (3 * a + b);
                        dp[r].x = fmaf(z4.x, wk[a].x, dp[r].x); dp[r].y = fmaf(z4.y, wk[a].y, dp[r].y);
                        dp[r].z = fmaf(z4.z, wk[a].z, dp[r].z); dp[r].w = fmaf(z4.w, wk[a].w, dp[r].w);
                        gw2[k].x = fmaf(pv[r].x, z4.x, gw2[k].x); gw2[k].y = fmaf(pv[r].y, z4.y, gw2[k].y);
                        gw2[k].z = fmaf(pv[r].z, z4.z, gw2[k].z); gw2[k].w = fmaf(pv[r].w, z4.w, gw2[k].w);
                        if (a == 1 && b == 1) {
                            gb2.x += z4.x; gb2.y += z4.y; gb2.z += z4.z; gb2.w += z4.w;
                        }
                    }
                }
                // one column at a time: without this the scheduler hoists all 3*(PPT+2) reads
                if (G::PPT > 1) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < G::PPT; ++r) {
                const bool in = inside((vr0 + r) * TW + vtx, y0 + vr0 + r, x0 + vtx);
                if (!in) dp[r] = make_float4(0, 0, 0, 0);
                gb1.x += dp[r].x; gb1.y += dp[r].y; gb1.z += dp[r].z; gb1.w += dp[r].w;
                *reinterpret_cast<float4*>(pb + r * TW * G::LSO) = dp[r];
            }
        }
        __syncthreads();
        DP_BWD_STAMP(2);
        // prefetch the next tile's global data; issued here (not right after the stage) so that the
        // p GEMM and the VALU phase run without ~64 prefetch registers live -- the two GEMMs, the
        // mask phase and the store that follow are several microseconds, enough for HBM
        if (SPREAD) { if (more) issue(t + gridDim.x, std::integral_constant<int, DP_BWD_PF_LATE ? 1 : 2>{}); }
        else if (!(GEMM == 1 && DP_BWD_PF_EARLY) && more) issue(t + gridDim.x, All{});

        // ---- dW1 += a^T * dp (K = pixels) and da = dp * W1 on the matrix cores -------------------
        if constexpr (GEMM == 1) {
            if (!(abl & 4)) {
                static_assert(GEMM == 0 || (G::MB == 2 && G::NB == 2), "bf16 dW1: 2x2 interleaved tiles");
                // K = pixels.  Within a block of 32 pixels lane group g supplies rows 8g .. 8g+7 of BOTH
                // operands (8-byte reads of 2 interleaved channels, rows 8 apart between lane groups:
                // conflict-free for ds_read_b64 with the 68-float row stride)
                const float* ap = s_a + (w1_kslice * G::KPX + 8 * g) * G::LSI + w1_ci0 + 2 * l15;
                const float* bp = s_pb + (w1_kslice * G::KPX + 8 * g) * G::LSO + w1_co0 + 2 * l15;
                float am[2], as_[2], ab[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ca = w1_ci0 + 2 * (G::LAUNDER ? opaque(l15) : l15) + j;
                    am[j] = s_ci[ca]; as_[j] = s_ci[CIN + ca]; ab[j] = s_ci[2 * CIN + ca];
                }
#pragma unroll 1
                for (int kb = 0; kb < G::KPX / 32; ++kb) {
                    // dp first (split once, used by both ci tiles), then one ci tile at a time: at most
                    // 16 raw floats + 24 packed operand registers are live
                    Split8 sd0, sd1;
                    {
                        float d0[8], d1[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const float2 dv = *reinterpret_cast<const float2*>(bp + (32 * kb + t) * G::LSO);
                            d0[t] = dv.x; d1[t] = dv.y;
                        }
                        sd0 = split8(d0);
                        sd1 = split8(d1);
                    }
                    float2 av[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) av[t] = *reinterpret_cast<const float2*>(ap + (32 * kb + t) * G::LSI);
                    {
                        float a0[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) a0[t] = tin(av[t].x, am[0], as_[0], ab[0], relu_floor);
                        const Split8 sa0 = split8(a0);
                        gw1[0] = mfma3(sa0, sd0.hi, sd0.lo, gw1[0]);
                        gw1[1] = mfma3(sa0, sd1.hi, sd1.lo, gw1[1]);
                    }
                    {
                        float a1[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) a1[t] = tin(av[t].y, am[1], as_[1], ab[1], relu_floor);
                        const Split8 sa1 = split8(a1);
                        gw1[2] = mfma3(sa1, sd0.hi, sd0.lo, gw1[2]);
                        gw1[3] = mfma3(sa1, sd1.hi, sd1.lo, gw1[3]);
                    }
                }
            }
        } else
        if (!(abl & 4)) {
            const float* ap = s_a + (w1_kslice * G::KSTEPS * 4 + g) * G::LSI + w1_ci0 + G::MB * l15;
            const float* bp = s_pb + (w1_kslice * G::KSTEPS * 4 + g) * G::LSO + w1_co0 + G::NB * l15;
            float am[G::MB], as_[G::MB], ab[G::MB];
#pragma unroll
            for (int j = 0; j < G::MB; ++j) {
                const int ca = w1_ci0 + G::MB * (G::LAUNDER ? opaque(l15) : l15) + j;
                am[j] = s_ci[ca]; as_[j] = s_ci[CIN + ca]; ab[j] = s_ci[2 * CIN + ca];
            }
            // a single 16 x 16 tile per wave (16 -> 16 units) would be one chain of KSTEPS dependent MFMAs: four
            // partial accumulators, summed once per tile
            constexpr int NPART = (G::MB * G::NB == 1) ? 4 : 1;
            f32x4 part[NPART];
            if constexpr (NPART > 1) {
#pragma unroll
                for (int q = 0; q < NPART; ++q) part[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll 8
            for (int s = 0; s < G::KSTEPS; ++s) {
                float av[G::MB], bv[G::NB];
                if (G::MB == 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(ap + 4 * s * G::LSI);
                    av[0] = t2.x; av[G::MB - 1] = t2.y;
                } else {
                    av[0] = ap[4 * s * G::LSI];
                }
                if (G::NB == 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(bp + 4 * s * G::LSO);
                    bv[0] = t2.x; bv[G::NB - 1] = t2.y;
                } else {
                    bv[0] = bp[4 * s * G::LSO];
                }
#pragma unroll
                for (int j = 0; j < G::MB; ++j) {
                    const float a = tin(av[j], am[j], as_[j], ab[j], relu_floor);
                    if constexpr (NPART > 1) {
                        part[s % NPART] = mfma16(a, bv[0], part[s % NPART]);
                    } else {
#pragma unroll
                        for (int i = 0; i < G::NB; ++i)
                            gw1[j * G::NB + i] = mfma16(a, bv[i], gw1[j * G::NB + i]);
                    }
                }
            }
            if constexpr (NPART > 1) {
                static_assert(NPART == 1 || G::KSTEPS % 4 == 0, "partial accumulators: whole rounds of four k-steps");
#pragma unroll
                for (int r = 0; r < 4; ++r) gw1[0][r] += (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
            }
        }
        if (SPREAD && more) issue(t + gridDim.x, std::integral_constant<int, DP_BWD_PF_LATE ? 2 : 3>{});
        f32x4 da[G::MPW][G::NTI];
#pragma unroll
        for (int mi = 0; mi < G::MPW; ++mi)
#pragma unroll
            for (int nt = 0; nt < G::NTI; ++nt) da[mi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (GEMM == 1) {
            if (!(abl & 8)) {
#pragma unroll
                for (int mi = 0; mi < G::MPW; ++mi) {
                    // k = output channels: lane group g supplies channels 32*kb + 8g .. +7
                    const float* prow = s_pb + ((wid * G::MPW + mi) * 16 + l15) * G::LSO + 8 * g;
                    const __bf16* wh = s_w1th + l15 * G::WSTB + 8 * g;
                    const __bf16* wl = s_w1tl + l15 * G::WSTB + 8 * g;
#pragma unroll
                    for (int kb = 0; kb < COUT / 32; ++kb) {
                        float p8[8];
                        const float4 v0 = *reinterpret_cast<const float4*>(prow + 32 * kb);
                        const float4 v1 = *reinterpret_cast<const float4*>(prow + 32 * kb + 4);
                        p8[0] = v0.x; p8[1] = v0.y; p8[2] = v0.z; p8[3] = v0.w;
                        p8[4] = v1.x; p8[5] = v1.y; p8[6] = v1.z; p8[7] = v1.w;
                        const Split8 ps = split8(p8);
#pragma unroll
                        for (int nt = 0; nt < G::NTI; ++nt) {
                            const u32x4 vh = *reinterpret_cast<const u32x4*>(wh + nt * 16 * G::WSTB + 32 * kb);
                            const u32x4 vl = *reinterpret_cast<const u32x4*>(wl + nt * 16 * G::WSTB + 32 * kb);
                            da[mi][nt] = mfma3(ps, vh, vl, da[mi][nt]);
                        }
                    }
                }
            }
        } else if (!(abl & 8)) {
#pragma unroll
            for (int mi = 0; mi < G::MPW; ++mi) {
                const float* prow = s_pb + ((wid * G::MPW + mi) * 16 + l15) * G::LSO + 4 * g;
                const float* wrow = s_w1t + l15 * G::WST + 4 * g;
                constexpr int NQ = COUT / 16, UNR = (NQ > 2 && G::NTI >= 4) ? 1 : NQ;   // k = output channels, permuted as above
                float4 a_c = *reinterpret_cast<const float4*>(prow);
                float4 b_c[G::NTI];
#pragma unroll
                for (int nt = 0; nt < G::NTI; ++nt)
                    b_c[nt] = *reinterpret_cast<const float4*>(wrow + nt * 16 * G::WST);
#pragma unroll UNR
                for (int q = 0; q < NQ; ++q) {
                    const int qn = UNR == 1 ? 16 * ((q + 1) % NQ) : (q + 1 < NQ ? 16 * (q + 1) : 0);
                    const float4 a_n = *reinterpret_cast<const float4*>(prow + qn);
                    float4 b_n[G::NTI];
#pragma unroll
                    for (int nt = 0; nt < G::NTI; ++nt)
                        b_n[nt] = *reinterpret_cast<const float4*>(wrow + nt * 16 * G::WST + qn);
#pragma unroll
                    for (int nt = 0; nt < G::NTI; ++nt) da[mi][nt] = mfma16(a_c.x, b_c[nt].x, da[mi][nt]);
#pragma unroll
                    for (int nt = 0; nt < G::NTI; ++nt) da[mi][nt] = mfma16(a_c.y, b_c[nt].y, da[mi][nt]);
#pragma unroll
                    for (int nt = 0; nt < G::NTI; ++nt) da[mi][nt] = mfma16(a_c.z, b_c[nt].z, da[mi][nt]);
#pragma unroll
                    for (int nt = 0; nt < G::NTI; ++nt) da[mi][nt] = mfma16(a_c.w, b_c[nt].w, da[mi][nt]);
                    a_c = a_n;
#pragma unroll
                    for (int nt = 0; nt < G::NTI; ++nt) b_c[nt] = b_n[nt];
                }
            }
        }
        __syncthreads();  // every wave is done reading s_a for dW1
        DP_BWD_STAMP(3);
        if (SPREAD && DP_BWD_PF_LATE && more) issue(t + gridDim.x, std::integral_constant<int, 3>{});
        if (bn_in) {
#pragma unroll
            for (int nt = 0; nt < G::NTI; ++nt) {
                const int c = nt * 16 + (G::LAUNDER ? opaque(l15) : l15);   // (keeps the 5*NTI coefficients out of loop-invariant registers)
                const float cm = s_ci[c], cs = s_ci[CIN + c], cb = s_ci[2 * CIN + c], ci = s_ci[3 * CIN + c];
                const float cl = s_ci[4 * CIN + c];
                // Sums of this wave's 16 * MPW pixels in fp32 (a handful of terms: rounding 1e-7 of the
                // partial), folded over the four lane groups with two cross-lane adds; only then fp64 and
                // ONE LDS atomic per channel and wave (the long, heavily cancelling accumulation over the
                // whole tensor stays in fp64).  64 lanes x 2 fp64 atomics on 16 addresses per channel block
                // used to serialise in the LDS.
                float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                for (int mi = 0; mi < G::MPW; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ip = (wid * G::MPW + mi) * 16 + 4 * g + r;
                        float* ap = s_a + ip * G::LSI + c;
                        const float xr = *ap;
                        const bool in = inside(ip, y0 + ip / TW, x0 + ip % TW);
                        const float v = (in && fmaf(xr - cm, cs, cb) > 0.0f) ? da[mi][nt][r] : 0.0f;   // ReLU mask
                        t0 += v;
                        t1 = fmaf(v, bn_center(xr, cm, cl) * ci, t1);
                        *ap = v;
                    }
                t0 += __shfl_xor(t0, 16, 64); t1 += __shfl_xor(t1, 16, 64);
                t0 += __shfl_xor(t0, 32, 64); t1 += __shfl_xor(t1, 32, 64);
                if (g == 0) {
                    __hip_atomic_fetch_add(s_bst + c, (double)t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(s_bst + CIN + c, (double)t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < G::NTI; ++nt)
#pragma unroll
                for (int mi = 0; mi < G::MPW; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s_a[((wid * G::MPW + mi) * 16 + 4 * g + r) * G::LSI + nt * 16 + l15] = da[mi][nt][r];
        }
        __syncthreads();
        DP_BWD_STAMP(4);

        // ---- dx store (coalesced) + BN-backward sums of the producer ------------------------------
        if (d.dx && !(abl & 16)) {
            const int tid = G::LAUNDER ? opaque((int)threadIdx.x) : (int)threadIdx.x;
            const int ich4 = tid % G::C4I;
            const unsigned xrange = PACKED ? (unsigned)d.N * (unsigned)d.x_img_stride * 4u : dxbytes;
            const auto r_dx = __builtin_amdgcn_make_buffer_rsrc(
                d.dx + (PACKED ? (size_t)0 : (size_t)n * d.x_img_stride), 0, xrange, 0x00020000);
            unsigned off[G::NX];
#pragma unroll
            for (int i = 0; i < G::NX; ++i) {
                const int ip = (tid + BWD_THREADS * i) / G::C4I;
                const int y = y0 + ip / TW, x = x0 + ip % TW;
                if constexpr (PACKED) {
                    int pn, py, px;
                    off[i] = pk_locate(pk, y, x, pn, py, px)
                                 ? (unsigned)(pn * d.x_img_stride + (py * W + px) * CIN + ich4 * 4) * 4u : xrange;
                } else {
                    off[i] = (FULL || (y < H && x < W)) ? (unsigned)((y * W + x) * CIN + ich4 * 4) * 4u : dxbytes;
                }
            }
            // two separate paths: the plain store must not wait on the vector-memory counter (the next
            // tile's prefetch is in flight and the counter is in-order), only dx += reads memory
            if (d.accumulate_dx) {
                u32x4 old[G::NX];
#pragma unroll
                for (int i = 0; i < G::NX; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(r_dx, off[i], 0, 0);
#pragma unroll
                for (int i = 0; i < G::NX; ++i) {
                    const int ip = (tid + BWD_THREADS * i) / G::C4I;
                    float4 v = *reinterpret_cast<const float4*>(s_a + ip * G::LSI + ich4 * 4);
                    const float4 o = *reinterpret_cast<const float4*>(&old[i]);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), r_dx, off[i], 0, PACKED ? 0 : YUNET_DX_AUX);
                }
            } else {
#pragma unroll
                for (int i = 0; i < G::NX; ++i) {
                    const int ip = (tid + BWD_THREADS * i) / G::C4I;
                    const float4 v = *reinterpret_cast<const float4*>(s_a + ip * G::LSI + ich4 * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), r_dx, off[i], 0, PACKED ? 0 : YUNET_DX_AUX);
                }
            }
        }
        __syncthreads();
        DP_BWD_STAMP(5);
    }

    // ============ flush per-workgroup partial sums ==============================================
#ifdef DP_BWD_PROF
    if (prof_on && threadIdx.x == 0) {
        for (int k = 0; k < 6; ++k) d.prof[blockIdx.x * 8 + k] = prof_acc[k];
        d.prof[blockIdx.x * 8 + 6] = prof_pro;
    }
    const unsigned long long prof_t1 = __builtin_readcyclecounter();
#endif
    float* row = d.wgrad_partials + (size_t)blockIdx.x * G::WROW;
    // Every wave parks its dW1 tiles in the LDS plane of its K slice and every thread its
    // dW2 | db1 | db2 accumulators in a record (two passes of 6 / 5 float4: 11 at once do not fit next to
    // the planes with 512 threads); the sums over K slices / pixel groups are taken in a fixed order
    // while the row is written.  Three barriers in all.
    float* s_gw1 = sm;                                   // [KSPLIT][COUT][CIN]
    float* red = sm + G::KSPLIT * COUT * CIN;            // [BWD_THREADS][24]
    {
        float* pl = s_gw1 + w1_kslice * COUT * CIN;
#pragma unroll
        for (int j = 0; j < G::MB; ++j)
#pragma unroll
            for (int i = 0; i < G::NB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pl[(w1_co0 + G::NB * l15 + i) * CIN + w1_ci0 + G::MB * (4 * g + r) + j] = gw1[j * G::NB + i][r];
    }
    float4* my = reinterpret_cast<float4*>(red + tid * 24);
#pragma unroll
    for (int k = 0; k < 6; ++k) my[k] = gw2[k];
    __syncthreads();
    for (int i = tid; i < COUT * CIN; i += BWD_THREADS) {
        float v = 0.0f;
#pragma unroll
        for (int ks = 0; ks < G::KSPLIT; ++ks) v += s_gw1[ks * COUT * CIN + i];
        row[i] = v;
    }
    // record slot k of pass ps: dW2 tap 6*ps + k, then db1, db2
    auto reduce_pass = [&](int ps, int nslot) {
        for (int o = tid; o < COUT * nslot; o += BWD_THREADS) {
            const int c = o / nslot, k = o - c * nslot;
            const int q = c >> 2, e = c & 3;
            float v = 0.0f;
            for (int p = 0; p < G::PG; ++p) v += red[(p * G::C4O + q) * 24 + k * 4 + e];
            const int slot = ps * 6 + k;
            if (slot < 9) row[COUT * CIN + COUT + c * 9 + slot] = v;
            else if (slot == 9) row[COUT * CIN + c] = v;
            else row[COUT * CIN + COUT + COUT * 9 + c] = v;
        }
    };
    reduce_pass(0, 6);
    __syncthreads();
    my[0] = gw2[6]; my[1] = gw2[7]; my[2] = gw2[8]; my[3] = gb1; my[4] = gb2;
    __syncthreads();
    reduce_pass(1, 5);
    // (c) BN-backward sums of the producer: one global fp64 atomic per channel
    if (bn_in && d.dx && d.in_bn.bstats && tid < 2 * CIN) atomic_add_f64(bn_slot(d.in_bn.bstats, d.in_bn.slots, CIN) + tid, s_bst[tid]);
#ifdef DP_BWD_PROF
    if (prof_on && threadIdx.x == 0) d.prof[blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - prof_t1;
#endif
}

template <int CIN, int COUT, int TH, int TW, bool PACKED = false, int GEMM = 0, bool POOLDY = false, bool FULL = false>
int launch_dp_bwd(const YunetDP* d, hipStream_t stream) {
    using G = BwdGeom<CIN, COUT, TH, TW, GEMM>;
    if constexpr (FULL) {
        if (d->H % TH != 0 || d->W % TW != 0) return YUNET_EINVAL;
    }
    static PerDevice attr_set;      // per device (common.h)
    if (per_device(attr_set, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(dp_bwd_kernel<CIN, COUT, TH, TW, PACKED, GEMM, POOLDY, FULL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM) == hipSuccess ? 1 : -1;
        }) < 0)
        return YUNET_EINVAL;
    PackGeom pk = dp_pack_geom(d->N, d->H, d->W);
    pk.on = PACKED ? 1 : 0;
    if (!dp_pack_fits(pk, d->x_img_stride, d->z_img_stride)) return YUNET_EINVAL;
    const int tiles = PACKED ? ((pk.CW + TW - 1) / TW) * ((pk.CH + TH - 1) / TH)
                             : d->N * ((d->W + TW - 1) / TW) * ((d->H + TH - 1) / TH);
    int grid = tiles < CONV_BLOCKS ? tiles : CONV_BLOCKS;
    if (grid > d->wgrad_blocks) grid = d->wgrad_blocks;
    if (grid < 1) return YUNET_EINVAL;
    if (grid < d->wgrad_blocks) {
        // yunet_dp_bwd_blocks() sized the partial buffer for another kernel's grid (the 64 -> 64 units on 8 x 8
        // tiles, while this launch is their exact-fp32 A/B variant on 8 x 16 tiles): the reduction sums every row,
        // so the rows no workgroup of this grid writes are zeroed
        const size_t width = (size_t)COUT * CIN + COUT + (size_t)COUT * 9 + COUT;
        if (hipMemsetAsync(d->wgrad_partials + (size_t)grid * width, 0, (size_t)(d->wgrad_blocks - grid) * width * 4,
                           stream) != hipSuccess)
            return hip_status();
    }
    hipLaunchKernelGGL((dp_bwd_kernel<CIN, COUT, TH, TW, PACKED, GEMM, POOLDY, FULL>), dim3(grid), dim3(BWD_THREADS), G::SMEM,
                       stream, *d, pk);
    return hip_status();
}

// ================================================================================================
// dp_bwd64: the 64 -> 64 ConvDPUnit backward (split-bf16 matrix path), round 3.
//
// Same arithmetic as dp_bwd_kernel<64,64,8,16,PACKED,1,POOLDY> above, re-laid for the two units the round-2
// profile showed busy (VALU 44 %, LDS 38 %, matrix cores 9 %, seven barriers per tile):
//   * every operand is split into bf16 hi / lo ONCE, where it is produced (a = T(x) in the stage, dp at the end
//     of the depthwise phase) and lives in LDS as two XOR-swizzled bf16 planes [pixel][channel]; the three GEMMs
//     read ready-made matrix operands (the old kernel re-split `a` in two GEMMs and `dp` in two);
//   * the p and da GEMMs are split over OUTPUT channels: a wave owns one 16-channel tile and 4 of the 8 pixel
//     tiles, and keeps its W1 / W1^T fragments (hi + lo, 32 registers) for the whole launch -- no weight planes in
//     LDS (37 KB), no weight reads per tile (the old kernel re-read all of W1 from LDS for 16 pixels);
//   * dW1 (K = pixels) takes its operands from the same row-major planes with the gfx950 transposing LDS read
//     (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block and each lane receives one
//     channel's 4 pixels) -- no second, transposed copy and no in-register transposition;
//   * the ReLU mask + the producer's BN-backward sums run on the da accumulators in registers (a lane's channel
//     is fixed for the launch: fp64 partials per lane, reduced once at the end -- no LDS atomics);
//   * five barriers per tile.
// LDS: dz halo 45 KB | raw x 32 KB | p / dx staging 32 KB | a planes 32 KB | small tables; the dp planes alias the
// dz halo (dead after the depthwise phase).
// one activation element through a buffer descriptor, widened to fp32
template <typename R>
__device__ __forceinline__ float act_bufld1(R rsrc, unsigned byte_off) {
#ifdef YUNET_ACT_BF16
    return __uint_as_float(((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, byte_off, 0, 0)) << 16);
#else
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 0));
#endif
}
#ifndef YUNET_BWD64_ADDR2
#define YUNET_BWD64_ADDR2 1
#endif
// Round 6: the two fp32 tiles the matrix-layout phases touch with 4-byte accesses (p / dx writes, x reads of the mask) are
// XOR-swizzled: a wave's 32-lane group there is 16 channels x 2 pixel groups 4 pixels apart = 256 words = the SAME banks
// (every such access paid a 2-way conflict); element (pixel, channel) now lives at channel ^ 16 * ((pixel >> 2) & 1).  The
// row-wise 16-byte accesses of the stage / depthwise / store phases see a wave-uniform flip (a wave's four pixels share
// pixel bit 2), so nothing else changes.  -DYUNET_BWD64_SWZ=0 = the linear layout.
#ifndef YUNET_BWD64_SWZ
#define YUNET_BWD64_SWZ 1
#endif
#ifndef YUNET_BWD64_X_AUX        // cache-policy bits of the x loads of the unpacked instances: x is read exactly once (tile interior only),
                                 // non-temporal keeps it out of the L2 the dy / z halo re-reads live in (step -0.04 ms, profiles/r06_bench_ab_ntx.log)
#define YUNET_BWD64_X_AUX 2
#endif
#ifndef YUNET_BWD64_DY_AUX       // cache-policy bits of the dy loads (measurement switch; 0 = default policy; the dx stores: common.h YUNET_DX_AUX)
#define YUNET_BWD64_DY_AUX 0
#endif
#ifndef YUNET_BWD64_PFMODE      // where the next tile's loads are issued: 0 = four pieces from the p GEMM on (rounds 2-5), 1 = all at
#define YUNET_BWD64_PFMODE 5    // once after the p GEMM (measurement), 5 = four pieces, each one issue point later (round 6)
#endif
namespace bwd64 {
__device__ __forceinline__ int tile_swz(int pixel) { return YUNET_BWD64_SWZ ? ((pixel >> 2) & 1) << 4 : 0; }
}
namespace bwd64 {
constexpr int C = 64, C4 = 16;
constexpr int PLANE_PX = 128;      // (pixels of the largest tile: plane_off() only needs the row pitch)
// NW = waves per workgroup: 8 -> 8 x 16 pixel tiles, one workgroup (149 KB of LDS) per CU;
//                           4 -> 8 x 8 pixel tiles, TWO independent workgroups (80 KB each) per CU, whose phases
//                                interleave on the SIMDs instead of marching in lockstep through five barriers
template <int NW>
struct Geo {
    static constexpr int NT = NW * 64;
    static constexpr int TH = 8, TW = 2 * NW, HW_ = TW + 2, HH_ = TH + 2, HP = HH_ * HW_, IP = TH * TW;
    static constexpr int PSTEP = NT / C4;                              // halo pixels per pass of the workgroup
    // (dy, z) float4 pairs per thread.  NW = 8: 180 halo pixels = 5 full passes + a partial one;
    // NW = 4: 100 = 6 full passes + 4 pixels, which are loaded as ONE float per thread (REM)
    static constexpr int NDZ = NW == 8 ? (HP + PSTEP - 1) / PSTEP : HP / PSTEP;
    static constexpr bool REM = NW == 4;
    static constexpr int REM_HP0 = NDZ * PSTEP;                        // first halo pixel of the remainder
    static constexpr int NX = (IP * C4) / NT;                          // x float4 per thread
    static constexpr int PLANE = IP * C * 2;                           // one bf16 plane
    static constexpr int XP = C;                                       // floats per pixel of the fp32 tiles s_x / s_p
    static constexpr int OFF_DZ = 0;                                   // float [HP][64]; later dp planes hi | lo
    static constexpr int OFF_X = OFF_DZ + HP * C * 4;                  // float [IP][XP] raw x
    static constexpr int OFF_P = OFF_X + IP * XP * 4;                  // float [IP][XP] p, later the masked dx
    static constexpr int OFF_A = OFF_P + IP * XP * 4;                  // bf16 planes hi | lo of a = T(x)
    static constexpr int WORKB = OFF_A + 2 * PLANE;
    static constexpr int PAR_F = 9 * C + 7 * C + 5 * C + C;            // w2 | out-bn | in-bn | b1 (floats)
    static constexpr int MH = NW / 4;                                  // pixel halves (p / da GEMM: 4 pixel tiles per wave)
#ifdef DP_BWD_PROF
    static constexpr int SMEM = WORKB + PAR_F * 4 + MH * 2 * C * 8 + IP + 64;
#else
    static constexpr int SMEM = WORKB + PAR_F * 4 + MH * 2 * C * 8 + IP;   // + fp64 sums per pixel half + validity bytes
#endif
    static constexpr int KSPLIT = NW / 4;                              // dW1: pixels 64 ks .. 64 ks + 63 per wave quad
    static_assert(2 * PLANE <= HP * C * 4, "dp planes alias the dz halo");
    static_assert((size_t)KSPLIT * C * C * 4 + (size_t)NT * 24 * 4 <= (size_t)WORKB, "flush area");
    static_assert(!REM || (HP - REM_HP0) * C == NT, "remainder: one float per thread");
    static_assert(IP % 64 == 0 && NX * NT == IP * C4, "tile mapping");
};
constexpr int WROW = C * C + C + C * 9 + C;
// byte offset of channels 8*chunk .. 8*chunk+7 of pixel `pix` inside a plane (16-byte chunks, XOR-swizzled so
// that both the row-wise 16-byte operand reads and the transposing reads are bank-conflict free)
__device__ __forceinline__ int plane_off(int pix, int chunk) { return pix * (C * 2) + ((chunk ^ (pix & 7)) << 4); }
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// [4 pixels][16 channels] block, transposed: this lane's channel, 4 consecutive pixels (see tools/ubench/tr_probe.hip)
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(
            (__attribute__((address_space(3))) unsigned char*)p));
    return __builtin_bit_cast(u32x2, v);
}
// bf16 activation storage (YUNET_ACT_BF16): the forward of this mode multiplied
// bf16(a) with bf16(W1) (conv_fwd64.hip), so the backward that is consistent with it recomputes p as that ONE product,
// takes dW1 = bf16(a)^T dp as two (dp = hi + lo) and da = dp bf16(W1) as two: 5 matrix products per tile instead of 9,
// and the low plane of `a` is neither written nor read
// (round 5; -DYUNET_BWD64_BF16_LEAN=0 builds the earlier variant, which split the fp32 a and W1 as the fp32 build does:
// same-box A/B of the bf16 step 4.00 -> 3.89 ms, profiles/r05_bf16_lean_ab.log)
#ifndef YUNET_BWD64_BF16_LEAN
#define YUNET_BWD64_BF16_LEAN 1
#endif
#if defined(YUNET_ACT_BF16) && YUNET_BWD64_BF16_LEAN
#define BWD64_LEAN 1
#else
#define BWD64_LEAN 0
#endif
__device__ __forceinline__ f32x4 mfma1r(const u32x4 a, const u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma3r(const u32x4 ah, const u32x4 al, const u32x4 bh, const u32x4 bl, f32x4 c) {
    const bf16x8 xh = __builtin_bit_cast(bf16x8, ah), xl = __builtin_bit_cast(bf16x8, al);
    const bf16x8 yh = __builtin_bit_cast(bf16x8, bh), yl = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, yh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, yl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, yh, c, 0, 0, 0);
    return c;
}
}  // namespace bwd64

template <int NW, bool PACKED, bool POOLDY>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))     // 256 registers per lane: 8 waves per CU
void dp_bwd64_kernel(const YunetDP d, const PackGeom pk) {
    using namespace bwd64;
    using G = Geo<NW>;
    constexpr bool ADDR2 = !PACKED && (YUNET_BWD64_ADDR2 != 0);      // round 5: shift-only halo addressing (issue2)
    constexpr int NT = G::NT, TH = G::TH, TW = G::TW, HW_ = G::HW_, HP = G::HP, IP = G::IP, NDZ = G::NDZ, NX = G::NX;
    constexpr int PSTEP = G::PSTEP, PLANE = G::PLANE, OFF_DZ = G::OFF_DZ, OFF_X = G::OFF_X, OFF_P = G::OFF_P;
    constexpr int OFF_A = G::OFF_A, WORKB = G::WORKB, KSPLIT = G::KSPLIT, MH = G::MH, XP = G::XP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* s_dz = reinterpret_cast<float*>(smem_raw + OFF_DZ);
    float* s_x = reinterpret_cast<float*>(smem_raw + OFF_X);
    float* s_p = reinterpret_cast<float*>(smem_raw + OFF_P);
    unsigned char* s_a = smem_raw + OFF_A;                 // planes of a: hi at 0, lo at PLANE
    unsigned char* s_d = smem_raw + OFF_DZ;                // planes of dp (alias the dz halo)
    float* s_w2 = reinterpret_cast<float*>(smem_raw + WORKB);      // [9][64]
    float* s_co = s_w2 + 9 * C;                            // folded BN backward of the unit's own BN: A|B|Dh|Dl (+3 spare rows)
    float* s_ci = s_co + 7 * C;                            // mean|scale|beta|invstd|mean_lo
    float* s_b1 = s_ci + 5 * C;                            // [64] pointwise bias
    double* s_bst = reinterpret_cast<double*>(s_b1 + C);   // [MH pixel halves][2][64]
    unsigned char* s_in = reinterpret_cast<unsigned char*>(s_bst + MH * 2 * C);  // [IP] packed mode: pixel is real

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: everything derived from it lives in SGPRs
    const int l15 = lane & 15, g = lane >> 4;
    const int H = d.H, W = d.W;
    const bool bn_in = d.in_transform == YUNET_T_BNRELU;
    const bool bn_out = d.out_has_bn != 0;
    const float relu_floor = bn_in ? 0.0f : -__builtin_inff();
    // debug: prof < 4096 is an ablation bit mask; a -DDP_BWD_PROF build also takes the mask from the low 6 bits of
    // the (256-byte aligned) counter pointer, so that ablated runs can be profiled per phase
    const unsigned abl = (unsigned long long)d.prof < 4096ull ? (unsigned)(unsigned long long)d.prof
#ifdef DP_BWD_PROF
                                                              : (unsigned)((unsigned long long)d.prof & 63ull);
    unsigned long long* const prof_out = reinterpret_cast<unsigned long long*>((unsigned long long)d.prof & ~63ull);
#else
                                                              : 0u;
#endif
#ifdef DP_BWD_PROF
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
    const int tiles_x = ((PACKED ? pk.CW : W) + TW - 1) / TW, tiles_y = ((PACKED ? pk.CH : H) + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int ntiles = PACKED ? tiles_img : d.N * tiles_img;
    auto inside = [&](int ip, int y, int x) {
        if constexpr (PACKED) return s_in[ip] != 0;
        else return y < H && x < W;
    };

    // ---- prefetch registers (next tile): raw dy / z_out over the halo, x over the interior -- as in dp_bwd_kernel
    float4 pdy[NDZ];
    act_raw4 pz[NDZ], px[NX];
    unsigned okmask = 0;
    static_assert(!(POOLDY && PACKED), "pooled dy: unpacked levels only");
    unsigned pid[POOLDY ? NDZ : 1];
    unsigned posmask = 0;
    float rem_dy = 0.0f, rem_z = 0.0f;        // NW = 4: the last 4 halo pixels, one float per thread
    unsigned rem_id = 0;
    const int Wq = W >> 1;
    const unsigned pooledbytes = (unsigned)((H >> 1) * Wq * C) * 4u;
    const unsigned dybytes = (unsigned)(H * W * C) * 4u, zbytes = (unsigned)(H * W * C) * ACT_B;
    const unsigned xbytes = (unsigned)(H * W * C) * ACT_B, dxbytes = (unsigned)(H * W * C) * 4u;
    auto issue = [&](int t, auto part_c) {
        constexpr int PART = decltype(part_c)::value;
        const int tid = opaque((int)threadIdx.x);
        const int och4 = tid % C4;
        const int n = PACKED ? 0 : t / tiles_img, rr = t - n * tiles_img;
        const int y0 = (rr / tiles_x) * TH, x0 = (rr % tiles_x) * TW;
        const unsigned dyrange = PACKED ? (unsigned)d.N * (unsigned)d.z_img_stride * 4u : dybytes;
        const unsigned zrange = PACKED ? (unsigned)d.N * (unsigned)d.z_img_stride * ACT_B : zbytes;
        const unsigned xrange = PACKED ? (unsigned)d.N * (unsigned)d.x_img_stride * ACT_B : xbytes;
        const size_t zbase = PACKED ? (size_t)0 : (size_t)n * d.z_img_stride;
        const size_t xbase = PACKED ? (size_t)0 : (size_t)n * d.x_img_stride;
        const auto r_dy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(d.dy) + (POOLDY ? (size_t)n * (pooledbytes / 4u) : zbase), 0, POOLDY ? pooledbytes : dyrange, 0x00020000);
        const auto r_id = __builtin_amdgcn_make_buffer_rsrc(
            d.pool_idx + (POOLDY ? (size_t)n * (pooledbytes / 4u) : (size_t)0), 0,
            POOLDY ? pooledbytes / 4u : 0u, 0x00020000);
        const auto r_z = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.z)) + zbase, 0, zrange, 0x00020000);
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + xbase, 0, xrange, 0x00020000);
        if (PART <= 0) { okmask = 0; posmask = 0; }
#pragma unroll
        for (int i = 0; i < NDZ; ++i) {
            if (PART >= 0 && PART != 1 + (3 * i) / NDZ) continue;
            const int hp = tid / C4 + PSTEP * i;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            bool ok;
            unsigned eo;
            if constexpr (PACKED) {
                int pn, py, pxx;
                ok = hp < HP && pk_locate(pk, y, x, pn, py, pxx);
                eo = (unsigned)(pn * d.z_img_stride + (py * W + pxx) * C + och4 * 4);
            } else {
                ok = hp < HP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                eo = (unsigned)((y * W + x) * C + och4 * 4);
            }
            okmask |= ok ? (1u << i) : 0u;
            if constexpr (POOLDY) {
                const unsigned eq = (unsigned)(((y >> 1) * Wq + (x >> 1)) * C + och4 * 4);
                posmask |= (unsigned)(((y & 1) << 1) | (x & 1)) << (2 * i);
                const u32x4 vdy = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eq * 4u : pooledbytes, 0, 0);
                pdy[i] = *reinterpret_cast<const float4*>(&vdy);
                pid[i] = __builtin_amdgcn_raw_buffer_load_b32(r_id, ok ? eq : pooledbytes, 0, 0);
            } else {
                const u32x4 vdy = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eo * 4u : dyrange, 0, 0);
                pdy[i] = *reinterpret_cast<const float4*>(&vdy);
            }
            pz[i] = act_raw4{};
            if (bn_out) pz[i] = act_bufld4(r_z, ok ? eo * ACT_B : zrange);
            if constexpr (PACKED) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (G::REM) {
            if (PART < 0 || PART == 3) {
                const int hp = G::REM_HP0 + tid / C, ch = tid % C;
                const int hy = hp / HW_, hx = hp - hy * HW_;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                bool ok;
                unsigned eo;
                if constexpr (PACKED) {
                    int pn, py, pxx;
                    ok = pk_locate(pk, y, x, pn, py, pxx);
                    eo = (unsigned)(pn * d.z_img_stride + (py * W + pxx) * C + ch);
                } else {
                    ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                    eo = (unsigned)((y * W + x) * C + ch);
                }
                okmask |= ok ? (1u << NDZ) : 0u;
                if constexpr (POOLDY) {
                    const unsigned eq = (unsigned)(((y >> 1) * Wq + (x >> 1)) * C + ch);
                    posmask |= (unsigned)(((y & 1) << 1) | (x & 1)) << (2 * NDZ);
                    rem_dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_dy, ok ? eq * 4u : pooledbytes, 0, 0));
                    rem_id = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r_id, ok ? eq : pooledbytes, 0, 0) & 0xffu;
                } else {
                    rem_dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_dy, ok ? eo * 4u : dyrange, 0, 0));
                }
                rem_z = 0.0f;
                if (bn_out) rem_z = act_bufld1(r_z, ok ? eo * ACT_B : zrange);
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (PART > 0) continue;
            const int ip = tid / C4 + PSTEP * i;
            const int y = y0 + ip / TW, x = x0 + ip % TW;
            unsigned off;
            if constexpr (PACKED) {
                int pn, py, pxx;
                off = pk_locate(pk, y, x, pn, py, pxx)
                          ? (unsigned)(pn * d.x_img_stride + (py * W + pxx) * C + och4 * 4) * ACT_B : xrange;
            } else {
                off = (y < H && x < W) ? (unsigned)((y * W + x) * C + och4 * 4) * ACT_B : xbytes;
            }
            px[i] = act_bufld4(r_x, off);
            if constexpr (PACKED) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- round 5 (ADDR2, unpacked maps): the same loads with a shift-only thread -> halo-slot mapping ------------------
    // The old mapping (halo pixel hp = tid / 16 + 32 i, row hp / 18) cost a division by the halo width, ~15 vector and
    // ~15 scalar instructions per load -- 16 loads per thread and tile, a quarter of the tile's VALU work, and every one
    // of the four issue pieces decomposed the tile index again.  Here the halo [TH + 2][TW + 2] is split into its TW-wide
    // main part -- pixel slot ps = tid / 16 of pass i < 5 is halo row 2 i + (ps >> log2 TW), column ps & (TW - 1) -- and
    // the two extra columns TW, TW + 1 (pass 5: row ps >> 1, column TW + (ps & 1); with NW = 4 their rows 8, 9 are the
    // one-float-per-thread remainder): a thread's element offset is ONE tile-invariant value `tm` plus a scalar per
    // (tile, pass), the interior loads use the same `tm`, and a uniform branch drops every per-element validity test
    // when the whole halo lies inside the image (48 % of the 80 x 80 tiles, 36 % at 40 x 40).
    constexpr int XSH = NW == 8 ? 4 : 3;                    // log2(TW)
    constexpr int NEXTRA = NW == 8 ? 20 : 16;               // halo pixels of the extra pass (NW = 8: 20 of 32 slots)
    static_assert(!ADDR2 || (NDZ == 6 && NX == 4 && PSTEP == 2 * TW && (1 << XSH) == TW), "ADDR2 pass geometry");
    auto issue2 = [&](int t, auto part_c) {
        constexpr int PART = decltype(part_c)::value;
        const int tid = opaque((int)threadIdx.x);
        const int och4 = tid & 15, ps = tid >> 4;
        const int r = ps >> XSH, hxm = ps & (TW - 1);
        const int n = t / tiles_img, rr = t - n * tiles_img;
        const int ty = rr / tiles_x;
        const int y0 = ty * TH, x0 = (rr - ty * tiles_x) * TW;
        const size_t zbase = (size_t)n * d.z_img_stride, xbase = (size_t)n * d.x_img_stride;
        const auto r_dy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(d.dy) + (POOLDY ? (size_t)n * (pooledbytes / 4u) : zbase), 0, POOLDY ? pooledbytes : dybytes, 0x00020000);
        const auto r_id = __builtin_amdgcn_make_buffer_rsrc(
            d.pool_idx + (POOLDY ? (size_t)n * (pooledbytes / 4u) : (size_t)0), 0, POOLDY ? pooledbytes / 4u : 0u, 0x00020000);
        const auto r_z = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.z)) + zbase, 0, zbytes, 0x00020000);
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + xbase, 0, xbytes, 0x00020000);
        const bool inner = y0 > 0 && x0 > 0 && y0 + TH < H && x0 + TW < W;      // uniform: the halo is inside the image
        if (PART <= 0) okmask = 0;
        const int tm = (r * W + hxm) * C + och4 * 4;              // elements from slot (row 0, column 0) of a pass
        const int hbase = ((y0 - 1) * W + (x0 - 1)) * C;          // halo origin (negative on the top / left border: masked)
        // pooled dy: halo slot (hy, hx) reads the pooled element ((y0 - 1 + hy) >> 1, (x0 - 1 + hx) >> 1); y0, x0 even
        const int tq = (r * Wq + ((hxm - 1) >> 1)) * C + och4 * 4;
        const int qbase = (((y0 >> 1) - 1) * Wq + (x0 >> 1)) * C;
        auto ld = [&](int i, bool ok, unsigned eo, unsigned eq) {
            if constexpr (POOLDY) {
                const u32x4 vdy = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eq * 4u : pooledbytes, 0, 0);
                pdy[i] = *reinterpret_cast<const float4*>(&vdy);
                pid[i] = __builtin_amdgcn_raw_buffer_load_b32(r_id, ok ? eq : pooledbytes, 0, 0);
            } else {
                const u32x4 vdy = __builtin_amdgcn_raw_buffer_load_b128(r_dy, ok ? eo * 4u : dybytes, 0, YUNET_BWD64_DY_AUX);
                pdy[i] = *reinterpret_cast<const float4*>(&vdy);
            }
            pz[i] = act_raw4{};
            if (bn_out) pz[i] = act_bufld4(r_z, ok ? eo * ACT_B : zbytes);
        };
        auto body = [&](auto inner_c) {
            constexpr bool INNER = decltype(inner_c)::value;
            const bool xok_m = INNER || (unsigned)(x0 - 1 + hxm) < (unsigned)W;
#pragma unroll
            for (int i = 0; i < NDZ; ++i) {
                if (PART >= 0 && PART != 1 + (3 * i) / NDZ) continue;
                if (i < 5) {
                    const bool ok = INNER || (xok_m && (unsigned)(y0 - 1 + 2 * i + r) < (unsigned)H);
                    okmask |= ok ? (1u << i) : 0u;
                    ld(i, ok, (unsigned)(hbase + i * 2 * W * C + tm), (unsigned)(qbase + i * Wq * C + tq));
                } else {
                    const int hy = ps >> 1, hx = TW + (ps & 1);
                    const bool ok = ps < NEXTRA && (INNER || ((unsigned)(y0 - 1 + hy) < (unsigned)H &&
                                                              (unsigned)(x0 - 1 + hx) < (unsigned)W));
                    okmask |= ok ? (1u << i) : 0u;
                    ld(i, ok, (unsigned)(hbase + (hy * W + hx) * C + och4 * 4),
                       (unsigned)(qbase + ((((hy - 1) >> 1) + 1) * Wq + ((hx - 1) >> 1)) * C + och4 * 4));
                }
            }
            if constexpr (G::REM) {
                if (PART < 0 || PART == 3) {
                    const int j = tid >> 6, ch = tid & 63;
                    const int hy = TH + (j >> 1), hx = TW + (j & 1);
                    const bool ok = INNER || ((unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W);
                    const unsigned eo = (unsigned)(hbase + (hy * W + hx) * C + ch);
                    okmask |= ok ? (1u << NDZ) : 0u;
                    if constexpr (POOLDY) {
                        const unsigned eq = (unsigned)(qbase + ((((hy - 1) >> 1) + 1) * Wq + ((hx - 1) >> 1)) * C + ch);
                        rem_dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_dy, ok ? eq * 4u : pooledbytes, 0, 0));
                        rem_id = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r_id, ok ? eq : pooledbytes, 0, 0) & 0xffu;
                    } else {
                        rem_dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_dy, ok ? eo * 4u : dybytes, 0, 0));
                    }
                    rem_z = 0.0f;
                    if (bn_out) rem_z = act_bufld1(r_z, ok ? eo * ACT_B : zbytes);
                }
            }
        };
        if (PART != 0) {
            if (inner) body(std::true_type{});
            else body(std::false_type{});
        }
        if (PART <= 0) {
            // raw x over the tile: pass i covers tile rows 2 i, 2 i + 1 -- the same `tm`
            const bool tfull = y0 + TH <= H && x0 + TW <= W;
            const int xb = (y0 * W + x0) * C;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const bool ok = tfull || (y0 + 2 * i + r < H && x0 + hxm < W);
                px[i] = act_bufld4_aux<YUNET_BWD64_X_AUX>(r_x, ok ? (unsigned)(xb + i * 2 * W * C + tm) * ACT_B : xbytes);
            }
        }
    };
    // round 6: every piece one issue point LATER than in rounds 2-5 (x after the depthwise phase ... the last third of dy / z
    // between the da GEMM and the mask): the requests spend less time queued in a memory system that is already
    // oversubscribed by 256 CUs prefetching a whole tile each -- issuing EARLIER (inside the stage, into the registers it
    // frees: built, +23 %) or all at once (+3 %) is worse, later is neutral at 80 x 80 and -1 .. -3 % on the smaller maps
    // (profiles/r06_bwd64_pf5.log; -DYUNET_BWD64_PFMODE=0 = the earlier points)
    constexpr bool LATE = (YUNET_BWD64_PFMODE == 5);
    auto issue_any = [&](int t, auto part_c) {
        if constexpr (ADDR2) issue2(t, part_c);
        else issue(t, part_c);
    };
    using All = std::integral_constant<int, -1>;
    int t = first_tile();

    // ---- this wave's weight fragments, straight from global memory into registers ------------------------------
    // p = a * W1^T and da = dp * W1: wave `wid` owns output-channel tile nt = wid & 3 of both GEMMs and the pixel
    // tiles 4 * (wid >> 2) .. + 3.  B operand of v_mfma_f32_16x16x32_bf16: lane (l15, g) supplies column l15,
    // k = 32 kb + 8 g .. + 7.
    // The weight loads (L2 hits after the first workgroups) are issued BEFORE the first tile's 16 loads per thread
    // and consumed after: vector-memory returns are counted in order, so weights queued behind a cold-start tile
    // would wait for all of HBM's latency before the first split could run.
    const int nt = wid & 3, mh = wid >> 2;
    u32x4 w1h[2], w1l[2], wth[2], wtl[2];
    {
        const int co = nt * 16 + l15;             // p GEMM: column = output channel, k = input channel
        const int ci = nt * 16 + l15;             // da GEMM: column = input channel, k = output channel
        float4 ra[2], rb[2];
        float rt[2][8];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            ra[kb] = *reinterpret_cast<const float4*>(d.w_pw + co * C + 32 * kb + 8 * g);
            rb[kb] = *reinterpret_cast<const float4*>(d.w_pw + co * C + 32 * kb + 8 * g + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) rt[kb][j] = d.w_pw[(32 * kb + 8 * g + j) * C + ci];
        }
        if (t < ntiles) issue_any(t, All{});
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float w8[8] = {ra[kb].x, ra[kb].y, ra[kb].z, ra[kb].w, rb[kb].x, rb[kb].y, rb[kb].z, rb[kb].w};
            const Split8 sp = split8(w8);
            w1h[kb] = sp.hi; w1l[kb] = sp.lo;
            const Split8 st = split8(rt[kb]);
            wth[kb] = st.hi; wtl[kb] = st.lo;
        }
    }
    for (int c = tid; c < C; c += NT) s_b1[c] = d.b_pw[c];
    staged_table<C * 9, NT>(d.w_dw, tid, [&](int i, float w) { s_w2[(i % 9) * C + i / 9] = w; });
    for (int c = tid; c < C; c += NT) {
        // dz = k1 * (dy - c1 - xhat * c2) folded into dz = A dy + B z + D (bn_fold in common.h): two FMAs and an
        // add per element instead of nine operations
        if (bn_out) {
            const BNFold f = bn_fold(bn_bwd_coef(d.out_bn, C, c));
            s_co[c] = f.a; s_co[C + c] = f.b; s_co[2 * C + c] = f.dh; s_co[3 * C + c] = f.dl;
        } else {
            s_co[c] = d.dy_scale ? d.dy_scale[c] : 1.0f;
            s_co[C + c] = 0.f; s_co[2 * C + c] = 0.f; s_co[3 * C + c] = 0.f;
        }
        if (bn_in) {
            const BNCoef k = bn_coef(d.in_bn, C, c);
            s_ci[c] = k.mean; s_ci[C + c] = k.scale; s_ci[2 * C + c] = k.beta;
            s_ci[3 * C + c] = k.invstd; s_ci[4 * C + c] = k.mean_lo;
        } else {
            s_ci[c] = 0.f; s_ci[C + c] = 1.f; s_ci[2 * C + c] = 0.f; s_ci[3 * C + c] = 1.f; s_ci[4 * C + c] = 0.f;
        }
    }
    __syncthreads();

    // ---- persistent accumulators ----------------------------------------------------------------------------------
    float4 gw2[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gw2[k] = make_float4(0, 0, 0, 0);
    float4 gb2 = make_float4(0, 0, 0, 0), gb1 = make_float4(0, 0, 0, 0);
    // dW1 (K = pixels): wave quad ks = wid >> 2 takes pixels 64 ks .. 64 ks + 63; wave (wid & 3) of a quad owns the
    // 2 x 2 block of 16 x 16 tiles  ci tiles 2 * (grp >> 1) + {0, 1}  x  co tiles 2 * (grp & 1) + {0, 1}
    f32x4 gw1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gw1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int w1_ct = 2 * ((wid & 3) >> 1), w1_ot = 2 * (wid & 1), w1_ks = wid >> 2;
    // producer's BN-backward sums: fp64, one private LDS slot per (pixel half, channel) -- wave (mh, nt) owns
    // channels nt * 16 .. + 15 of half mh, so plain read-modify-write (as registers they cost 4 VGPRs)
    for (int i = tid; i < MH * 2 * C; i += NT) s_bst[i] = 0.0;

    const bool pf_on = !(abl & 32);
#ifdef DP_BWD_PROF
    // phase counters live in LDS (as registers they change the allocation of the kernel they measure)
    const bool prof_on = (unsigned long long)d.prof >= 4096ull;
    unsigned long long* s_prof = reinterpret_cast<unsigned long long*>(s_in + IP);
    if (tid == 0) {
        for (int q = 0; q < 6; ++q) s_prof[q] = 0;
        s_prof[7] = __builtin_readcyclecounter();
        s_prof[6] = s_prof[7] - prof_t0;
    }
#endif
    for (; t < ntiles; t += gridDim.x) {
        const int n = PACKED ? 0 : t / tiles_img, rr = t - n * tiles_img;
        const int y0 = (rr / tiles_x) * TH, x0 = (rr % tiles_x) * TW;
        // every pixel of the tile is a real pixel (all tiles of an 80 x 80 map, and of a 40 x 40 one with 8 x 8
        // tiles): the per-element validity tests -- hundreds of integer instructions per tile -- are skipped
        const bool tile_full = !PACKED && y0 + TH <= H && x0 + TW <= W;

        // ---- stage: dz = BN backward of this unit's own BN -> LDS; x raw -> LDS; a = T(x) split -> planes ---------
        {
            const int tid = opaque((int)threadIdx.x);
            const int och4 = tid % C4;
            const float4 o_a = *reinterpret_cast<float4*>(s_co + och4 * 4);
            const float4 o_b = *reinterpret_cast<float4*>(s_co + C + och4 * 4);
            const float4 o_dh = *reinterpret_cast<float4*>(s_co + 2 * C + och4 * 4);
            const float4 o_dl = *reinterpret_cast<float4*>(s_co + 3 * C + och4 * 4);
            const int hp0 = tid / C4;
            if constexpr (ADDR2) {
                // the slots of issue2: pass i < 5 -> halo row 2 i + r, column hxm; pass 5 -> the two extra columns
                const int ps = tid >> 4, r = ps >> XSH, hxm = ps & (TW - 1);
                const bool inner = y0 > 0 && x0 > 0 && y0 + TH < H && x0 + TW < W;
                float* const lm = s_dz + (r * HW_ + hxm) * C + och4 * 4;
                auto fold = [&](auto inner_c) {
                    constexpr bool INNER = decltype(inner_c)::value;
#pragma unroll
                    for (int i = 0; i < NDZ; ++i) {
                        const int hy = ps >> 1, hx = TW + (ps & 1);         // (pass 5)
                        if (i < 5 || ps < NEXTRA) {
                            float4 dy = pdy[i];
                            const float4 z = act_unpack(pz[i]);
                            if constexpr (POOLDY) {
                                // window position of the slot: y0, x0 are even, so the parities are the slot's own
                                const unsigned pos = i < 5 ? (unsigned)(((r ^ 1) << 1) | ((hxm + 1) & 1))
                                                           : (unsigned)((((hy + 1) & 1) << 1) | ((ps & 1) ^ 1));
                                const unsigned id = pid[i];
                                dy.x = (id & 0xffu) == pos ? dy.x : 0.0f;
                                dy.y = ((id >> 8) & 0xffu) == pos ? dy.y : 0.0f;
                                dy.z = ((id >> 16) & 0xffu) == pos ? dy.z : 0.0f;
                                dy.w = (id >> 24) == pos ? dy.w : 0.0f;
                            }
                            // zero padding of dz: a slot outside the image loaded dy = z = 0, which the BN backward
                            // would turn into D
                            const bool ok = INNER || ((okmask >> i) & 1u);
                            float4 v;
                            v.x = ok ? fmaf(o_a.x, dy.x, fmaf(o_b.x, z.x, o_dh.x)) + o_dl.x : 0.0f;
                            v.y = ok ? fmaf(o_a.y, dy.y, fmaf(o_b.y, z.y, o_dh.y)) + o_dl.y : 0.0f;
                            v.z = ok ? fmaf(o_a.z, dy.z, fmaf(o_b.z, z.z, o_dh.z)) + o_dl.z : 0.0f;
                            v.w = ok ? fmaf(o_a.w, dy.w, fmaf(o_b.w, z.w, o_dh.w)) + o_dl.w : 0.0f;
                            float* dst = i < 5 ? lm + i * 2 * HW_ * C : s_dz + (hy * HW_ + hx) * C + och4 * 4;
                            *reinterpret_cast<float4*>(dst) = v;
                        }
                    }
                    if constexpr (G::REM) {
                        const int j = tid >> 6, ch = tid & 63;
                        const int hp = (TH + (j >> 1)) * HW_ + TW + (j & 1);
                        float dyv = rem_dy;
                        if constexpr (POOLDY) dyv = rem_id == (unsigned)((((j >> 1) ^ 1) << 1) | ((j & 1) ^ 1)) ? dyv : 0.0f;
                        const bool ok = INNER || ((okmask >> NDZ) & 1u);
                        s_dz[hp * C + ch] = ok ? fmaf(s_co[ch], dyv, fmaf(s_co[C + ch], rem_z, s_co[2 * C + ch])) + s_co[3 * C + ch] : 0.0f;
                    }
                };
                if (inner) fold(std::true_type{});
                else fold(std::false_type{});
            } else {
#pragma unroll
            for (int i = 0; i < NDZ; ++i) {
                const int hp = hp0 + PSTEP * i;
                if ((i + 1) * PSTEP <= HP || hp < HP) {
                    float4 dy = pdy[i];
                    const float4 z = act_unpack(pz[i]);
                    if constexpr (POOLDY) {
                        const unsigned id = pid[i], pos = (posmask >> (2 * i)) & 3u;
                        dy.x = (id & 0xffu) == pos ? dy.x : 0.0f;
                        dy.y = ((id >> 8) & 0xffu) == pos ? dy.y : 0.0f;
                        dy.z = ((id >> 16) & 0xffu) == pos ? dy.z : 0.0f;
                        dy.w = (id >> 24) == pos ? dy.w : 0.0f;
                    }
                    // zero padding of dz: a slot outside the image loaded dy = z = 0, which the BN backward
                    // would turn into D
                    const bool ok = (okmask >> i) & 1u;
                    float4 v;
                    v.x = ok ? fmaf(o_a.x, dy.x, fmaf(o_b.x, z.x, o_dh.x)) + o_dl.x : 0.0f;
                    v.y = ok ? fmaf(o_a.y, dy.y, fmaf(o_b.y, z.y, o_dh.y)) + o_dl.y : 0.0f;
                    v.z = ok ? fmaf(o_a.z, dy.z, fmaf(o_b.z, z.z, o_dh.z)) + o_dl.z : 0.0f;
                    v.w = ok ? fmaf(o_a.w, dy.w, fmaf(o_b.w, z.w, o_dh.w)) + o_dl.w : 0.0f;
                    *reinterpret_cast<float4*>(s_dz + hp * C + och4 * 4) = v;
                }
            }
            if constexpr (G::REM) {
                const int hp = G::REM_HP0 + tid / C, ch = tid % C;
                float dyv = rem_dy;
                if constexpr (POOLDY) dyv = rem_id == ((posmask >> (2 * NDZ)) & 3u) ? dyv : 0.0f;
                const bool ok = (okmask >> NDZ) & 1u;
                s_dz[hp * C + ch] = ok ? fmaf(s_co[ch], dyv, fmaf(s_co[C + ch], rem_z, s_co[2 * C + ch])) + s_co[3 * C + ch] : 0.0f;
            }
            }
            if constexpr (PACKED) {
                for (int ip = tid; ip < IP; ip += NT) {
                    int pn, py, pxx;
                    s_in[ip] = pk_locate(pk, y0 + ip / TW, x0 + ip % TW, pn, py, pxx) ? 1 : 0;
                }
            }
            const float4 i_mean = *reinterpret_cast<float4*>(s_ci + och4 * 4);
            const float4 i_scale = *reinterpret_cast<float4*>(s_ci + C + och4 * 4);
            const float4 i_beta = *reinterpret_cast<float4*>(s_ci + 2 * C + och4 * 4);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int ip = hp0 + PSTEP * i;
                const float4 xv = act_unpack(px[i]);
                *reinterpret_cast<float4*>(s_x + ip * XP + ((och4 * 4) ^ tile_swz(hp0))) = xv;
                unsigned h0, l0, h1, l1;
                split2(tin(xv.x, i_mean.x, i_scale.x, i_beta.x, relu_floor),
                       tin(xv.y, i_mean.y, i_scale.y, i_beta.y, relu_floor), h0, l0);
                split2(tin(xv.z, i_mean.z, i_scale.z, i_beta.z, relu_floor),
                       tin(xv.w, i_mean.w, i_scale.w, i_beta.w, relu_floor), h1, l1);
                unsigned char* ap = s_a + plane_off(ip, och4 >> 1) + (och4 & 1) * 8;
                *reinterpret_cast<u32x2*>(ap) = u32x2{h0, h1};
                if (!BWD64_LEAN) *reinterpret_cast<u32x2*>(ap + PLANE) = u32x2{l0, l1};
            }
        }
        __syncthreads();
        DP_BWD64_STAMP(0);
        const bool more = t + (int)gridDim.x < ntiles && pf_on;

        // ---- p = a * W1^T + b1: this wave's 16 output channels on 4 pixel tiles -------------------------------------
        if (!(abl & 1)) {
            const int l15o = opaque(l15), go = opaque(g);
            const unsigned char* abase = s_a + (mh * 64 + l15o) * (C * 2);
            const int sw = l15o & 7;
            const float bias1 = s_b1[nt * 16 + l15o];
            f32x4 acc[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {           // four independent accumulator chains per k block
                u32x4 ah[4], al[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const unsigned char* ap = abase + mi * 16 * (C * 2) + (((4 * kb + go) ^ sw) << 4);
                    ah[mi] = *reinterpret_cast<const u32x4*>(ap);
                    if (!BWD64_LEAN) al[mi] = *reinterpret_cast<const u32x4*>(ap + PLANE);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[mi] = BWD64_LEAN ? mfma1r(ah[mi], w1h[kb], acc[mi]) : mfma3r(ah[mi], al[mi], w1h[kb], w1l[kb], acc[mi]);
                __builtin_amdgcn_sched_barrier(0);
            }
            float* pw = s_p + (mh * 64 + 4 * go) * XP + ((nt * 16 + l15o) ^ tile_swz(4 * go));
            if (tile_full) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pw[(mi * 16 + r) * XP] = acc[mi][r] + bias1;
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ip = (mh * 4 + mi) * 16 + 4 * go + r;
                        const bool in = inside(ip, y0 + ip / TW, x0 + ip % TW);
                        pw[(mi * 16 + r) * XP] = in ? acc[mi][r] + bias1 : 0.0f;
                    }
            }
        }
        __syncthreads();
        DP_BWD64_STAMP(1);
        // next tile's loads go out in four pieces from here on (the p GEMM above runs with no load in flight: a
        // CU cannot keep a whole tile's 124 KB in flight, and the in-order vector-memory queue would hold any
        // scratch access behind them)
#if YUNET_BWD64_PFMODE == 1     // experiment: the whole next tile at once
        if (more) issue_any(t + gridDim.x, All{});
#else
        if (more && !LATE) issue_any(t + gridDim.x, std::integral_constant<int, 0>{});
#endif

        // ---- depthwise backward on the VALU (sliding window over a 4-row column); dp stays in registers -------------
        float4 dp[4];
        int d_pix0;
        {
            const int tv = opaque((int)threadIdx.x);
            const int cq = tv % C4, pg = tv / C4;
            const int vtx = pg % TW, vr0 = (pg / TW) * 4;
            d_pix0 = vr0 * TW + vtx;
            const float* zb = s_dz + (vr0 * HW_ + vtx) * C + cq * 4;
            const float* pb = s_p + (vr0 * TW + vtx) * XP + ((cq * 4) ^ tile_swz(vtx));
            const float* wb = s_w2 + cq * 4;
            float4 pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = *reinterpret_cast<const float4*>(pb + r * TW * XP);
                dp[r] = make_float4(0, 0, 0, 0);
            }
            if (!(abl & 2)) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float4 wk[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        wk[a] = *reinterpret_cast<const float4*>(wb + (8 - (3 * a + b)) * C);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const float4 z4 = *reinterpret_cast<const float4*>(zb + (j * HW_ + b) * C);
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const int r = j - a;
                            if (r < 0 || r >= 4) continue;
                            const int k = 8 - (3 * a + b);
                            dp[r].x = fmaf(z4.x, wk[a].x, dp[r].x); dp[r].y = fmaf(z4.y, wk[a].y, dp[r].y);
                            dp[r].z = fmaf(z4.z, wk[a].z, dp[r].z); dp[r].w = fmaf(z4.w, wk[a].w, dp[r].w);
                            gw2[k].x = fmaf(pv[r].x, z4.x, gw2[k].x); gw2[k].y = fmaf(pv[r].y, z4.y, gw2[k].y);
                            gw2[k].z = fmaf(pv[r].z, z4.z, gw2[k].z); gw2[k].w = fmaf(pv[r].w, z4.w, gw2[k].w);
                            if (a == 1 && b == 1) {
                                gb2.x += z4.x; gb2.y += z4.y; gb2.z += z4.z; gb2.w += z4.w;
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!tile_full) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool in = inside((vr0 + r) * TW + vtx, y0 + vr0 + r, x0 + vtx);
                    if (!in) dp[r] = make_float4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gb1.x += dp[r].x; gb1.y += dp[r].y; gb1.z += dp[r].z; gb1.w += dp[r].w;
            }
        }
#if YUNET_BWD64_PFMODE != 1
        if (more) { if (LATE) issue_any(t + gridDim.x, std::integral_constant<int, 0>{}); else issue_any(t + gridDim.x, std::integral_constant<int, 1>{}); }
#endif
        __syncthreads();      // every dz read is done: the dp planes may overwrite the halo
        {
            const int cq = opaque((int)threadIdx.x) % C4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pix = d_pix0 + r * TW;
                unsigned h0, l0, h1, l1;
                split2(dp[r].x, dp[r].y, h0, l0);
                split2(dp[r].z, dp[r].w, h1, l1);
                unsigned char* q = s_d + plane_off(pix, cq >> 1) + (cq & 1) * 8;
                *reinterpret_cast<u32x2*>(q) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(q + PLANE) = u32x2{l0, l1};
            }
        }
        __syncthreads();
        DP_BWD64_STAMP(2);
#if YUNET_BWD64_PFMODE != 1
        if (more) { if (LATE) issue_any(t + gridDim.x, std::integral_constant<int, 1>{}); else issue_any(t + gridDim.x, std::integral_constant<int, 2>{}); }
#endif

        // ---- dW1 += a^T * dp (K = pixels): operands through the transposing LDS read --------------------------------
        // k index of lane group G, element e (0..7): pixel 32 kb + 4 * (4 (G >> 1) + 2 (e >> 2) + (G & 1)) + (e & 3) --
        // the two 4-pixel blocks a 32-lane half reads in one instruction then differ in pixel bit 2, which the
        // plane swizzle turns into different banks (any k order is valid as long as A and B agree)
        if (!(abl & 4)) {
            const int lo_ = opaque(lane);
            const int i16 = lo_ & 15, G = lo_ >> 4;
            const int prow = 4 * (4 * (G >> 1) + (G & 1)) + (i16 >> 2);          // + 8 for the second read
            const int sub = i16 & 3;                                              // 4-channel quad inside the 16-channel tile
#pragma unroll
            for (int kbi = 0; kbi < 2; ++kbi) {
                const int p0 = 64 * w1_ks + 32 * kbi + prow;
                u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int tI = 0; tI < 2; ++tI) {
                    const int chA = 2 * (w1_ct + tI) + (sub >> 1), chB = 2 * (w1_ot + tI) + (sub >> 1);
                    const unsigned char* a0 = s_a + plane_off(p0, chA) + (sub & 1) * 8;
                    const unsigned char* a1 = s_a + plane_off(p0 + 8, chA) + (sub & 1) * 8;
                    const unsigned char* b0 = s_d + plane_off(p0, chB) + (sub & 1) * 8;
                    const unsigned char* b1 = s_d + plane_off(p0 + 8, chB) + (sub & 1) * 8;
                    const u32x2 ah0 = tr_read(a0), ah1 = tr_read(a1);
                    u32x2 al0 = u32x2{0, 0}, al1 = u32x2{0, 0};
                    if (!BWD64_LEAN) { al0 = tr_read(a0 + PLANE); al1 = tr_read(a1 + PLANE); }
                    const u32x2 bh0 = tr_read(b0), bh1 = tr_read(b1), bl0 = tr_read(b0 + PLANE), bl1 = tr_read(b1 + PLANE);
                    ah[tI] = u32x4{ah0.x, ah0.y, ah1.x, ah1.y}; al[tI] = u32x4{al0.x, al0.y, al1.x, al1.y};
                    bh[tI] = u32x4{bh0.x, bh0.y, bh1.x, bh1.y}; bl[tI] = u32x4{bl0.x, bl0.y, bl1.x, bl1.y};
                }
                if (BWD64_LEAN) {      // a = its bf16 plane: a^T (dp_lo) then a^T (dp_hi)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        gw1[q] = mfma1r(ah[q >> 1], bh[q & 1], mfma1r(ah[q >> 1], bl[q & 1], gw1[q]));
                } else {
                gw1[0] = mfma3r(ah[0], al[0], bh[0], bl[0], gw1[0]);
                gw1[1] = mfma3r(ah[0], al[0], bh[1], bl[1], gw1[1]);
                gw1[2] = mfma3r(ah[1], al[1], bh[0], bl[0], gw1[2]);
                gw1[3] = mfma3r(ah[1], al[1], bh[1], bl[1], gw1[3]);
                }
            }
        }
        DP_BWD64_STAMP(3);
#if YUNET_BWD64_PFMODE != 1
        if (more) { if (LATE) issue_any(t + gridDim.x, std::integral_constant<int, 2>{}); else issue_any(t + gridDim.x, std::integral_constant<int, 3>{}); }
#endif

        // ---- da = dp * W1 (this wave's 16 input channels, 4 pixel tiles) + ReLU mask + BN-backward sums ----------------
        // Everything a step needs is requested before the step that consumes it (operands of both k blocks, then the
        // raw x of the mask): with two waves per SIMD a read -> wait -> use chain per pixel tile is pure LDS latency.
        {
            const int l15o = opaque(l15), go = opaque(g);
            const unsigned char* dbase = s_d + (mh * 64 + l15o) * (C * 2);
            const int sw = l15o & 7;
            const int c = nt * 16 + l15o;
            f32x4 da[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) da[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(abl & 8)) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {       // four independent accumulator chains per k block
                    u32x4 ph[4], pl[4];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const unsigned char* q = dbase + mi * 16 * (C * 2) + (((4 * kb + go) ^ sw) << 4);
                        ph[mi] = *reinterpret_cast<const u32x4*>(q);
                        pl[mi] = *reinterpret_cast<const u32x4*>(q + PLANE);
                    }
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        da[mi] = BWD64_LEAN ? mfma1r(ph[mi], wth[kb], mfma1r(pl[mi], wth[kb], da[mi]))
                                            : mfma3r(ph[mi], pl[mi], wth[kb], wtl[kb], da[mi]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (LATE && more) issue_any(t + gridDim.x, std::integral_constant<int, 3>{});
            if (bn_in) {
                const float m_mean = s_ci[c], m_scale = s_ci[C + c], m_beta = s_ci[2 * C + c], m_inv = s_ci[3 * C + c];
                const float m_lo = s_ci[4 * C + c];
                const float* xrd = s_x + (mh * 64 + 4 * go) * XP + (c ^ tile_swz(4 * go));
                float* pw = s_p + (mh * 64 + 4 * go) * XP + (c ^ tile_swz(4 * go));
                float xr[4][4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[mi][r] = xrd[(mi * 16 + r) * XP];
                if (!tile_full) {
                    // a pixel outside the image carries dp = 0, hence da = 0: only the mask of the BN sums is at
                    // stake, and da = 0 contributes nothing to them either -- but keep x finite and masked
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ip = (mh * 4 + mi) * 16 + 4 * go + r;
                            if (!inside(ip, y0 + ip / TW, x0 + ip % TW)) da[mi][r] = 0.0f;
                        }
                }
                float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = fmaf(xr[mi][r] - m_mean, m_scale, m_beta) > 0.0f ? da[mi][r] : 0.0f;   // ReLU mask
                        t0 += v;
                        t1 = fmaf(v, bn_center(xr[mi][r], m_mean, m_lo) * m_inv, t1);
                        pw[(mi * 16 + r) * XP] = v;
                    }
                // 64 pixels per lane in fp32 (a handful of terms), folded over the four lane groups; then fp64 for the
                // long, heavily cancelling accumulation over the whole tensor
                t0 += __shfl_xor(t0, 16, 64); t1 += __shfl_xor(t1, 16, 64);
                t0 += __shfl_xor(t0, 32, 64); t1 += __shfl_xor(t1, 32, 64);
                if (go == 0) {
                    double* bs = s_bst + (mh * 2) * C + c;
                    bs[0] += (double)t0;
                    bs[C] += (double)t1;
                }
            } else {
                float* pw = s_p + (mh * 64 + 4 * go) * XP + (c ^ tile_swz(4 * go));
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pw[(mi * 16 + r) * XP] = da[mi][r];
            }
        }
        __syncthreads();
        DP_BWD64_STAMP(4);

        // ---- dx store (coalesced rows of s_p) --------------------------------------------------------------------------
        if (d.dx && !(abl & 16)) {
            const int tid = opaque((int)threadIdx.x);
            const int ich4 = tid % C4;
            const unsigned xrange = PACKED ? (unsigned)d.N * (unsigned)d.x_img_stride * 4u : dxbytes;
            const auto r_dx = __builtin_amdgcn_make_buffer_rsrc(
                d.dx + (PACKED ? (size_t)0 : (size_t)n * d.x_img_stride), 0, xrange, 0x00020000);
            unsigned off[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int ip = tid / C4 + PSTEP * i;
                const int y = y0 + ip / TW, x = x0 + ip % TW;
                if constexpr (PACKED) {
                    int pn, py, pxx;
                    off[i] = pk_locate(pk, y, x, pn, py, pxx)
                                 ? (unsigned)(pn * d.x_img_stride + (py * W + pxx) * C + ich4 * 4) * 4u : xrange;
                } else {
                    off[i] = (y < H && x < W) ? (unsigned)((y * W + x) * C + ich4 * 4) * 4u : dxbytes;
                }
            }
            if (d.accumulate_dx) {
                u32x4 old[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(r_dx, off[i], 0, 0);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const int ip = tid / C4 + PSTEP * i;
                    float4 v = *reinterpret_cast<const float4*>(s_p + ip * XP + ((ich4 * 4) ^ tile_swz(ip)));
                    const float4 o = *reinterpret_cast<const float4*>(&old[i]);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), r_dx, off[i], 0, YUNET_DX_AUX);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const int ip = tid / C4 + PSTEP * i;
                    const float4 v = *reinterpret_cast<const float4*>(s_p + ip * XP + ((ich4 * 4) ^ tile_swz(ip)));
                    // (non-temporal on the big maps, default policy on the packed 20 x 20 / 10 x 10 levels: common.h)
                    if constexpr (PACKED) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), r_dx, off[i], 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), r_dx, off[i], 0, YUNET_DX_AUX);
                }
            }
        }
        DP_BWD64_STAMP(5);
        // no barrier here: the next stage writes the halo / x / a planes (all read before the barrier above) and
        // s_p is next written by the p GEMM, one barrier later
    }

    // ============ flush per-workgroup partial sums ================================================================
#ifdef DP_BWD_PROF
    if (prof_on && threadIdx.x == 0)
        for (int q = 0; q < 7; ++q) prof_out[blockIdx.x * 8 + q] = s_prof[q];
    const unsigned long long prof_t1 = __builtin_readcyclecounter();
#endif
    __syncthreads();                                     // the last tile's dx rows have been read
    float* row = d.wgrad_partials + (size_t)blockIdx.x * WROW;
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* s_gw1 = sm;                                   // [KSPLIT][COUT][CIN]
    float* red = sm + KSPLIT * C * C;                    // [NT][24]
    {
        float* pl = s_gw1 + w1_ks * C * C;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)      // D layout: column = co (B operand tile), row = ci (A operand tile)
                    pl[((w1_ot + i) * 16 + l15) * C + (w1_ct + j) * 16 + 4 * g + r] = gw1[j * 2 + i][r];
    }
    float4* my = reinterpret_cast<float4*>(red + tid * 24);
#pragma unroll
    for (int k = 0; k < 6; ++k) my[k] = gw2[k];
    __syncthreads();
    for (int i = tid; i < C * C; i += NT) {
        float v = s_gw1[i];
        if constexpr (KSPLIT == 2) v += s_gw1[C * C + i];
        row[i] = v;
    }
    constexpr int PG = NT / C4;
    auto reduce_pass = [&](int ps, int nslot) {
        for (int o = tid; o < C * nslot; o += NT) {
            const int c = o / nslot, k = o - c * nslot;
            const int q = c >> 2, e = c & 3;
            float v = 0.0f;
            for (int p = 0; p < PG; ++p) v += red[(p * C4 + q) * 24 + k * 4 + e];
            const int slot = ps * 6 + k;
            if (slot < 9) row[C * C + C + c * 9 + slot] = v;
            else if (slot == 9) row[C * C + c] = v;
            else row[C * C + C + C * 9 + c] = v;
        }
    };
    reduce_pass(0, 6);
    if (bn_in && d.dx && d.in_bn.bstats && tid < 2 * C) {
        double v = s_bst[tid];
        if constexpr (MH == 2) v += s_bst[2 * C + tid];
        atomic_add_f64(bn_slot(d.in_bn.bstats, d.in_bn.slots, C) + tid, v);
    }
    __syncthreads();
    my[0] = gw2[6]; my[1] = gw2[7]; my[2] = gw2[8]; my[3] = gb1; my[4] = gb2;
    __syncthreads();
    reduce_pass(1, 5);
#ifdef DP_BWD_PROF
    if (prof_on && threadIdx.x == 0) prof_out[blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - prof_t1;
#endif
}

template <int NW, bool PACKED, bool POOLDY>
int launch_dp_bwd64(const YunetDP* d, hipStream_t stream) {
    using G = bwd64::Geo<NW>;
    static PerDevice attr_set;      // per device (common.h)
    if (per_device(attr_set, [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(dp_bwd64_kernel<NW, PACKED, POOLDY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM) == hipSuccess ? 1 : -1;
        }) < 0)
        return YUNET_EINVAL;
    PackGeom pk = dp_pack_geom(d->N, d->H, d->W);
    pk.on = PACKED ? 1 : 0;
    if (!dp_pack_fits(pk, d->x_img_stride, d->z_img_stride)) return YUNET_EINVAL;
    const int tiles = PACKED ? ((pk.CW + G::TW - 1) / G::TW) * ((pk.CH + G::TH - 1) / G::TH)
                             : d->N * ((d->W + G::TW - 1) / G::TW) * ((d->H + G::TH - 1) / G::TH);
    int grid = tiles < CONV_BLOCKS ? tiles : CONV_BLOCKS;
    if (grid > d->wgrad_blocks) grid = d->wgrad_blocks;
    if (grid < 1) return YUNET_EINVAL;
    hipLaunchKernelGGL((dp_bwd64_kernel<NW, PACKED, POOLDY>), dim3(grid), dim3(G::NT), G::SMEM, stream, *d, pk);
    return hip_status();
}

// ----------------------------------------------------------------------------- stem wgrad
#define SB_TW 32
#define SB_TH 8
__global__ __launch_bounds__(256) void stem_bwd_kernel(const float* __restrict__ img,
                                                       const act_t* __restrict__ z,
                                                       const float* __restrict__ dy, YunetBN bn,
                                                       float* __restrict__ partials, int N, int H,
                                                       int W) {
    constexpr int PH = 2 * SB_TH + 1;
    constexpr int PW4 = (2 * SB_TW + 8) / 4;            // aligned float4 per patch row
    constexpr int PWS = PW4 * 4 + 1;                    // odd LDS row stride
    constexpr int NLD = (3 * PH * PW4 + 255) / 256;
    constexpr int DZS = 20;
    constexpr int PATCH_F = ((3 * PH * PWS + 3) / 4) * 4;
    constexpr int DZT_F = SB_TH * SB_TW * DZS;
    constexpr int ALL_F = (PATCH_F + DZT_F) > 256 * 33 ? (PATCH_F + DZT_F) : 256 * 33;
    __shared__ __attribute__((aligned(16))) float s_all[ALL_F];
    float* s_patch = s_all;
    float* s_dzt = s_all + PATCH_F;
    __shared__ float s_k[4][16];
    const int tid = threadIdx.x;
    const int Ho = H / 2, Wo = W / 2;
    if (tid < 16) {
        const BNFold f = bn_fold(bn_bwd_coef(bn, 16, tid));     // dz = A dy + B z + D (common.h)
        s_k[0][tid] = f.a; s_k[1][tid] = f.b; s_k[2][tid] = f.dh; s_k[3][tid] = f.dl;
    }
    __syncthreads();
    const int lc4 = tid & 3;  // channel quad in the dz load phase
    float fa[4], fb[4], fdh[4], fdl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = s_k[0][lc4 * 4 + i]; fb[i] = s_k[1][lc4 * 4 + i];
        fdh[i] = s_k[2][lc4 * 4 + i]; fdl[i] = s_k[3][lc4 * 4 + i];
    }
    // role: 4 output-channel quads x 4 tap groups of 7; 16 pixel slices of 16 pixels
    const int role = tid & 15, slice = tid >> 4;
    const int cog = role & 3, tg = role >> 2;
    int toff[7];
    bool tok[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int t = tg * 7 + k;
        tok[k] = t < 27;
        const int tt = tok[k] ? t : 0;
        const int ci = tt / 9, ky = (tt % 9) / 3, kx = tt % 3;
        toff[k] = ci * PH * PWS + ky * PWS + kx + 3;   // patch col 0 = image col 2*x0 - 4
    }
    float4 acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = make_float4(0, 0, 0, 0);
    float4 accb = make_float4(0, 0, 0, 0);

    const int tiles_x = (Wo + SB_TW - 1) / SB_TW, tiles_y = (Ho + SB_TH - 1) / SB_TH;
    const int ntiles = N * tiles_x * tiles_y;
    for (int t = first_tile(); t < ntiles; t += gridDim.x) {
        const int n = t / (tiles_x * tiles_y);
        const int r = t - n * tiles_x * tiles_y;
        const int y0 = (r / tiles_x) * SB_TH, x0 = (r % tiles_x) * SB_TW;
        __syncthreads();
        {
            float4 ld[NLD];
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = tid + 256 * k;
                const int rowi = i / PW4, c4 = i - rowi * PW4;
                const int ci = rowi / PH, py = rowi - ci * PH;
                const int iy = 2 * y0 - 1 + py, ix = 2 * x0 - 4 + 4 * c4;
                ld[k] = make_float4(0, 0, 0, 0);
                if (rowi < 3 * PH && iy >= 0 && iy < H && ix >= 0 && ix + 3 < W)
                    ld[k] = *reinterpret_cast<const float4*>(img + (((size_t)n * 3 + ci) * H + iy) * W + ix);
                else if (rowi < 3 * PH && iy >= 0 && iy < H) {
                    const float* src = img + (((size_t)n * 3 + ci) * H + iy) * W;
                    if (ix + 0 >= 0 && ix + 0 < W) ld[k].x = src[ix + 0];
                    if (ix + 1 >= 0 && ix + 1 < W) ld[k].y = src[ix + 1];
                    if (ix + 2 >= 0 && ix + 2 < W) ld[k].z = src[ix + 2];
                    if (ix + 3 >= 0 && ix + 3 < W) ld[k].w = src[ix + 3];
                }
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = tid + 256 * k;
                const int rowi = i / PW4, c4 = i - rowi * PW4;
                if (rowi < 3 * PH) {
                    float* dst = s_patch + rowi * PWS + 4 * c4;
                    dst[0] = ld[k].x; dst[1] = ld[k].y; dst[2] = ld[k].z; dst[3] = ld[k].w;
                }
            }
        }
        for (int q = tid; q < SB_TH * SB_TW * 4; q += 256) {
            const int pix = q >> 2;
            const int oy = y0 + pix / SB_TW, ox = x0 + pix % SB_TW;
            float4 v = make_float4(0, 0, 0, 0);
            if (oy < Ho && ox < Wo) {
                const size_t off = (((size_t)n * Ho + oy) * Wo + ox) * 16 + lc4 * 4;
                const float4 g4 = *reinterpret_cast<const float4*>(dy + off);
                const float4 z4 = act_ld4(z + off);
                v.x = bn_dz_folded(g4.x, z4.x, fa[0], fb[0], fdh[0], fdl[0]);
                v.y = bn_dz_folded(g4.y, z4.y, fa[1], fb[1], fdh[1], fdl[1]);
                v.z = bn_dz_folded(g4.z, z4.z, fa[2], fb[2], fdh[2], fdl[2]);
                v.w = bn_dz_folded(g4.w, z4.w, fa[3], fb[3], fdh[3], fdl[3]);
            }
            *reinterpret_cast<float4*>(s_dzt + pix * DZS + lc4 * 4) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int pix = slice * 16 + j;
            const int ty = pix / SB_TW, tx = pix % SB_TW;
            const float4 dz = *reinterpret_cast<const float4*>(s_dzt + pix * DZS + cog * 4);
            const float* pb = s_patch + 2 * ty * PWS + 2 * tx;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const float v = tok[k] ? pb[toff[k]] : 0.0f;
                acc[k].x = fmaf(v, dz.x, acc[k].x); acc[k].y = fmaf(v, dz.y, acc[k].y);
                acc[k].z = fmaf(v, dz.z, acc[k].z); acc[k].w = fmaf(v, dz.w, acc[k].w);
            }
            if (tg == 0) { accb.x += dz.x; accb.y += dz.y; accb.z += dz.z; accb.w += dz.w; }
        }
    }
    // reduce over the 16 pixel slices
    __syncthreads();
    float* red = s_all;  // [256][33], aliases the patch / dz tiles (all reads are done)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        red[tid * 33 + k * 4 + 0] = acc[k].x; red[tid * 33 + k * 4 + 1] = acc[k].y;
        red[tid * 33 + k * 4 + 2] = acc[k].z; red[tid * 33 + k * 4 + 3] = acc[k].w;
    }
    red[tid * 33 + 28] = accb.x; red[tid * 33 + 29] = accb.y;
    red[tid * 33 + 30] = accb.z; red[tid * 33 + 31] = accb.w;
    __syncthreads();
    float* row = partials + (size_t)blockIdx.x * (16 * 27 + 16);
    for (int o = tid; o < 16 * 27 + 16; o += 256) {
        float v = 0.0f;
        if (o < 16 * 27) {
            const int co = o / 27, tt = o - co * 27;
            const int tgi = tt / 7, k = tt - tgi * 7;
            const int rl = tgi * 4 + (co >> 2);
            for (int s = 0; s < 16; ++s) v += red[(s * 16 + rl) * 33 + k * 4 + (co & 3)];
        } else {
            const int co = o - 16 * 27;
            const int rl = (co >> 2);  // tg == 0
            for (int s = 0; s < 16; ++s) v += red[(s * 16 + rl) * 33 + 28 + (co & 3)];
        }
        row[o] = v;
    }
}

// ------------------------------------------------------------------- pool / upsample-add
// `extra` (may be null): a second, FULL-SIZE gradient of the same activation y = relu(bn(z)) -- the share the
// upsample-add of the neck sends to a pyramid tap (dsum, identity branch).  Both shares pass the same ReLU mask and
// feed the same BatchNorm-backward sums, so dx = mask (extra + route(dy_out)) is written once here instead of
// upadd_bwd writing mask extra and this kernel re-reading z and read-modify-writing dx (engine.py: _upadd / _pool).
__global__ __launch_bounds__(256) void pool_bwd_kernel(const act_t* __restrict__ z, YunetBN bn,
                                                       const float* __restrict__ dyo,
                                                       const float* __restrict__ extra,
                                                       float* __restrict__ dx, int accumulate, int N,
                                                       int H, int W, int C) {
    const int C4 = C / 4, Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * C4;
    const int c4 = threadIdx.x % C4;
    __shared__ float s_tab[5 * 64];
    bn_table_fill(s_tab, bn, C, threadIdx.x);
    __syncthreads();
    BNCoef k[4];
    bn_table_get(s_tab, C, c4 * 4, k);
    double bst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bst[i] = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long pix = e / C4;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho), n = (int)(pix / Ho);
        const float4 g4 = *reinterpret_cast<const float4*>(dyo + e * 4);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float zv[4][4], yv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = act_ld4(z + (((size_t)n * H + 2 * oy + (j >> 1)) * W + 2 * ox + (j & 1)) * C + c4 * 4);
            zv[j][0] = v.x; zv[j][1] = v.y; zv[j][2] = v.z; zv[j][3] = v.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) yv[j][i] = bnrelu(zv[j][i], k[i].mean, k[i].scale, k[i].beta);
        }
        float o[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int am = 0;
            float m = yv[0][i];
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (yv[j][i] > m) { m = yv[j][i]; am = j; }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j][i] = (j == am && m > 0.0f) ? gv[i] : 0.0f;
            if (m > 0.0f) {
                bst[i] += (double)gv[i];
                bst[4 + i] += (double)(gv[i] * (bn_center(zv[am][i], k[i].mean, k[i].mean_lo) * k[i].invstd));
            }
        }
        if (extra) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 e4 = *reinterpret_cast<const float4*>(
                    extra + (((size_t)n * H + 2 * oy + (j >> 1)) * W + 2 * ox + (j & 1)) * C + c4 * 4);
                const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (yv[j][i] > 0.0f) {
                        o[j][i] += ev[i];
                        bst[i] += (double)ev[i];
                        bst[4 + i] += (double)(ev[i] * (bn_center(zv[j][i], k[i].mean, k[i].mean_lo) * k[i].invstd));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4* dst = reinterpret_cast<float4*>(
                dx + (((size_t)n * H + 2 * oy + (j >> 1)) * W + 2 * ox + (j & 1)) * C + c4 * 4);
            float4 v = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
            if (accumulate) {
                const float4 p = *dst;
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            *dst = v;
        }
    }
    __shared__ double red[256 * 8];
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = bst[i];
    __syncthreads();
    if ((int)threadIdx.x < 2 * C && bn.bstats) {
        const int which = threadIdx.x / C, c = threadIdx.x % C;
        const int q = c >> 2, kk = (c & 3) + 4 * which;
        double v = 0.0;
        for (int p = 0; p < 256 / C4; ++p) v += red[(p * C4 + q) * 8 + kk];
        atomic_add_f64(bn_slot(bn.bstats, bn.slots, C) + which * C + c, v);
    }
}

__global__ __launch_bounds__(256) void upadd_bwd_kernel(const act_t* __restrict__ za, YunetBN bna,
                                                        const act_t* __restrict__ zb, YunetBN bnb,
                                                        const float* __restrict__ dout,
                                                        float* __restrict__ dxa, int acc_a,
                                                        float* __restrict__ dxb, int acc_b, int N,
                                                        int H, int W, int C) {
    // one thread = one float4 of one COARSE pixel (covers the 2x2 fine pixels)
    const int C4 = C / 4, Hb = H / 2, Wb = W / 2;
    const long long total = (long long)N * Hb * Wb * C4;
    const int c4 = threadIdx.x % C4;
    __shared__ float s_ta[5 * 64], s_tb[5 * 64];
    if (dxa) bn_table_fill(s_ta, bna, C, threadIdx.x);          // (the fine tensor's BatchNorm is not needed without its share)
    bn_table_fill(s_tb, bnb, C, threadIdx.x);
    __syncthreads();
    BNCoef ka[4], kb[4];
    bn_table_get(dxa ? s_ta : s_tb, C, c4 * 4, ka);
    bn_table_get(s_tb, C, c4 * 4, kb);
    double bsa[8], bsb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsa[i] = bsb[i] = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (long long)gridDim.x * 256) {
        long long pix = e / C4;
        const int bx = (int)(pix % Wb);
        pix /= Wb;
        const int by = (int)(pix % Hb), n = (int)(pix / Hb);
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t off = (((size_t)n * H + 2 * by + (j >> 1)) * W + 2 * bx + (j & 1)) * C + c4 * 4;
            const float4 g4 = *reinterpret_cast<const float4*>(dout + off);
            if (!dxa) {          // the fine tensor's share is applied by pool_bwd_kernel (extra): za is not read
                sum[0] += g4.x; sum[1] += g4.y; sum[2] += g4.z; sum[3] += g4.w;
                continue;
            }
            const float4 z4 = act_ld4(za + off);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sum[i] += gv[i];
                const bool on = bnrelu(zv[i], ka[i].mean, ka[i].scale, ka[i].beta) > 0.0f;
                o[i] = on ? gv[i] : 0.0f;
                if (on) {
                    bsa[i] += (double)gv[i];
                    bsa[4 + i] += (double)(gv[i] * (bn_center(zv[i], ka[i].mean, ka[i].mean_lo) * ka[i].invstd));
                }
            }
            float4* dst = reinterpret_cast<float4*>(dxa + off);
            float4 v = make_float4(o[0], o[1], o[2], o[3]);
            if (acc_a) {
                const float4 p = *dst;
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            *dst = v;
        }
        const size_t offb = (((size_t)n * Hb + by) * Wb + bx) * C + c4 * 4;
        const float4 zb4 = act_ld4(zb + offb);
        const float zbv[4] = {zb4.x, zb4.y, zb4.z, zb4.w};
        float ob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = bnrelu(zbv[i], kb[i].mean, kb[i].scale, kb[i].beta) > 0.0f;
            ob[i] = on ? sum[i] : 0.0f;
            if (on) {
                bsb[i] += (double)sum[i];
                bsb[4 + i] += (double)(sum[i] * (bn_center(zbv[i], kb[i].mean, kb[i].mean_lo) * kb[i].invstd));
            }
        }
        float4* dstb = reinterpret_cast<float4*>(dxb + offb);
        float4 v = make_float4(ob[0], ob[1], ob[2], ob[3]);
        if (acc_b) {
            const float4 p = *dstb;
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        *dstb = v;
    }
    __shared__ double red[256 * 8];
#define UPADD_FLUSH(SRC, DST, SLOTS)                                                  \
    __syncthreads();                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = SRC[i]; \
    __syncthreads();                                                                 \
    if ((int)threadIdx.x < 2 * C && DST) {                                           \
        const int which = threadIdx.x / C, c = threadIdx.x % C;                      \
        const int q = c >> 2, kk = (c & 3) + 4 * which;                              \
        double v = 0.0;                                                              \
        for (int p = 0; p < 256 / C4; ++p) v += red[(p * C4 + q) * 8 + kk];          \
        atomic_add_f64(bn_slot(DST, SLOTS, C) + which * C + c, v);                   \
    }
    UPADD_FLUSH(bsa, (dxa ? bna.bstats : nullptr), bna.slots)
    UPADD_FLUSH(bsb, bnb.bstats, bnb.slots)
#undef UPADD_FLUSH
}

// The coarse gradient alone (dxa == NULL: the fine tensor's share of a pyramid tap is applied by pool_bwd_kernel,
// DESIGN 3): dxb = mask_b (sum of the 2 x 2 fine gradients) + the coarse BatchNorm's backward sums.  Round 5: the
// general kernel above compiled `if (on) sum += ...` of its 16 channel lanes into exec-mask branches around fp64 adds
// and separated a thread's four fine loads by them (3.3 TB/s); here the four loads + the coarse z are issued together
// and the mask is a select (adding the +0.0 of a masked-out element leaves every sum unchanged: same values, same
// order of additions as the general kernel).
__global__ __launch_bounds__(256) void upadd_bwd_coarse_kernel(const act_t* __restrict__ zb, YunetBN bnb,
                                                               const float* __restrict__ dout, float* __restrict__ dxb,
                                                               int acc_b, int N, int H, int W, int C) {
    const int C4 = C / 4, Hb = H / 2, Wb = W / 2;
    const long long total = (long long)N * Hb * Wb * C4;
    const int c4 = threadIdx.x % C4;
    __shared__ float s_tb[5 * 64];
    bn_table_fill(s_tb, bnb, C, threadIdx.x);
    __syncthreads();
    BNCoef kb[4];
    bn_table_get(s_tb, C, c4 * 4, kb);
    double bsb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsb[i] = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        long long pix = e / C4;
        const int bx = (int)(pix % Wb);
        pix /= Wb;
        const int by = (int)(pix % Hb), n = (int)(pix / Hb);
        const size_t off0 = (((size_t)n * H + 2 * by) * W + 2 * bx) * C + c4 * 4;
        const size_t offb = (((size_t)n * Hb + by) * Wb + bx) * C + c4 * 4;
        const float4 g0 = *reinterpret_cast<const float4*>(dout + off0);
        const float4 g1 = *reinterpret_cast<const float4*>(dout + off0 + C);
        const float4 g2 = *reinterpret_cast<const float4*>(dout + off0 + (size_t)W * C);
        const float4 g3 = *reinterpret_cast<const float4*>(dout + off0 + (size_t)W * C + C);
        const float4 zb4 = act_ld4(zb + offb);
        float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
        if (acc_b) old = *reinterpret_cast<const float4*>(dxb + offb);
        const float sum[4] = {((0.f + g0.x) + g1.x) + g2.x + g3.x, ((0.f + g0.y) + g1.y) + g2.y + g3.y,
                              ((0.f + g0.z) + g1.z) + g2.z + g3.z, ((0.f + g0.w) + g1.w) + g2.w + g3.w};
        const float zbv[4] = {zb4.x, zb4.y, zb4.z, zb4.w};
        float ob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = bnrelu(zbv[i], kb[i].mean, kb[i].scale, kb[i].beta) > 0.0f;
            ob[i] = on ? sum[i] : 0.0f;
            const float xh = on ? bn_center(zbv[i], kb[i].mean, kb[i].mean_lo) * kb[i].invstd : 0.0f;
            bsb[i] += (double)ob[i];
            bsb[4 + i] += (double)(ob[i] * xh);
        }
        float4 v = make_float4(ob[0], ob[1], ob[2], ob[3]);
        if (acc_b) { v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
        *reinterpret_cast<float4*>(dxb + offb) = v;
    }
    __shared__ double red[256 * 8];
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = bsb[i];
    __syncthreads();
    if ((int)threadIdx.x < 2 * C && bnb.bstats) {
        const int which = threadIdx.x / C, c = threadIdx.x % C;
        const int q = c >> 2, kk = (c & 3) + 4 * which;
        double v = 0.0;
        for (int p = 0; p < 256 / C4; ++p) v += red[(p * C4 + q) * 8 + kk];
        atomic_add_f64(bn_slot(bnb.bstats, bnb.slots, C) + which * C + c, v);
    }
}

#ifndef YUNET_ACT_BF16
__global__ void bn_param_grad_kernel(const double* __restrict__ bstats, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float db = (float)bstats[c], dg = (float)bstats[C + c];
    dbeta[c] = accumulate ? dbeta[c] + db : db;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
}

// One row slice of a column: rows sl, sl + 16, ... added IN THAT ORDER.  Sixteen (then four) loads are issued before the
// first addition: written as `v += p[...]` in a plain loop the compiler waits out every load before the next one is issued
// (s_waitcnt vmcnt(0) per iteration), and a 768-row job -- 48 rows per slice, each an L2 / HBM round trip -- took 40 us
// at the END of the backward, where nothing overlaps it.  Same additions in the same order: bit-identical sums.
__device__ __forceinline__ float column_slice_sum(const float* __restrict__ p, int blocks, int width, int sl) {
    float v = 0.0f;
    int b = sl;
    for (; b + 16 * 15 < blocks; b += 16 * 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = p[(size_t)(b + 16 * u) * width];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += x[u];
    }
    for (; b + 16 * 3 < blocks; b += 16 * 4) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = p[(size_t)(b + 16 * u) * width];
#pragma unroll
        for (int u = 0; u < 4; ++u) v += x[u];
    }
    for (; b < blocks; b += 16) v += p[(size_t)b * width];
    return v;
}

// out[j] (+)= sum_b partials[b][j]: 64 columns x 16 row-slices per workgroup (coalesced 256-byte
// row segments, 16 x 16 loads in flight per column), combined in a fixed order -> deterministic.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials,
                                                               int blocks, int width,
                                                               float* __restrict__ out,
                                                               int accumulate) {
    __shared__ float s[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float v = 0.0f;
    if (col < width) v = column_slice_sum(partials + col, blocks, width, sl);
    s[sl][lane] = v;
    __syncthreads();
    if (sl == 0 && col < width) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[k][lane];
        out[col] = accumulate ? out[col] + t : t;
    }
}

// table-driven variant: workgroup -> (job, 64-column chunk); same arithmetic and order as above
__global__ __launch_bounds__(1024) void reduce_partials_batch_kernel(const YunetReduceJob* __restrict__ jobs,
                                                                     int njobs) {
    __shared__ float s[16][64];
    // the last job whose first chunk is <= this workgroup (chunk0 ascends): bisection -- 6 dependent scalar loads
    // instead of up to njobs
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].chunk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const YunetReduceJob job = jobs[lo];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = ((int)blockIdx.x - job.chunk0) * 64 + lane;
    float v = 0.0f;
    if (col < job.width) v = column_slice_sum(job.partials + col, job.blocks, job.width, sl);
    s[sl][lane] = v;
    __syncthreads();
    if (sl == 0 && col < job.width) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[k][lane];
        job.out[col] = job.accumulate ? job.out[col] + t : t;
    }
}

#endif

// Grid of the element-wise backward kernels (pool / upsample-add).  Every workgroup ends with 2 * C fp64 atomics on
// the producer's BN-backward sums -- the same 128 addresses for the whole launch: with 2048 workgroups those 262 k
// same-address atomics, not the 59 - 370 MB of traffic, set the time (pool_bwd + upadd_bwd 0.260 ms per step).
// Measured: cap 1024 0.223 ms, 768 0.213, 512 0.216, 384 0.242, 256 0.295 (too few waves in flight).
// With the sums in eight replicas (YunetBN::slots) the order is the same -- 768 0.209 ms, 1536 0.255, 2048 0.257,
// 4096 0.302: it is the NUMBER of fp64 atomics of a launch (2 * C per workgroup), not only their addresses.
inline int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    const long long cap = yunet_options().ew_grid;      // 768 unless a measurement changed it
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define DP_BWD_MAX_BLOCKS 256     // 512 threads, up to 138 KB LDS: one workgroup per CU
#define STEM_BWD_MAX_BLOCKS 768   // 256 threads, 35 KB LDS: three per CU
static bool dp_bwd_big_tile(int H, int W, int cin, int cout) {
    return cin == 16 && cout == 16 && W >= 64 && H >= 32;
}
// Waves per workgroup of dp_bwd64 (the 64 -> 64 units): 8 = one 512-thread workgroup per CU on 8 x 16 tiles,
// 4 = two independent 256-thread workgroups per CU on 8 x 8 tiles.  Measured (tools/ubench/bwd_ab, N = 256):
// 80 x 80 0.415 vs 0.429 ms, 40 x 40 0.139 vs 0.126 ms (a 40-wide map fills 8 x 8 tiles exactly, 17 % of every
// 8 x 16 tile row is padding), packed 20 x 20 / 10 x 10 canvases 0.052 / 0.023 vs 0.055 / 0.028 ms.  So: 8 x 8
// tiles where the width is a multiple of 8 but not of 16, 8 x 16 otherwise.  The option bwd64_nw = 4 | 8 forces one
// (A/B runs).  The choice fixes the persistent grid, i.e. the rows of wgrad_partials.
static int bwd64_nw(int N, int H, int W) {
    const int forced = yunet_options().bwd64_nw;
    if (forced == 4 || forced == 8) return forced;
    if (dp_pack_geom(N, H, W).on) return 8;
    return (W % 16 != 0 && W % 8 == 0) ? 4 : 8;
}
#ifdef YUNET_ACT_BF16
extern "C" int yunet_dp_bwd_blocks(int N, int H, int W, int cin, int cout);
extern "C" int yunet_dp_pool_fusion_ok(int N, int H, int W, int cin, int cout);
extern "C" int yunet_stem_bwd_blocks(int N, int H, int W);
#else
extern "C" int yunet_dp_bwd_blocks(int N, int H, int W, int cin, int cout) {
    const bool two_per_cu = cin == 64 && cout == 64 && bwd64_nw(N, H, W) == 4;      // dp_bwd64 on 8 x 8 tiles
    const int th = dp_bwd_big_tile(H, W, cin, cout) ? 16 : 8, tw = two_per_cu ? 8 : th * 2;
    const PackGeom pk = dp_pack_geom(N, H, W);       // small maps: one tile grid over the packed canvas
    const long long tiles = dp_use_pack_bwd(N, H, W, cin, cout)
                                ? (long long)((pk.CW + tw - 1) / tw) * ((pk.CH + th - 1) / th)
                                  : (long long)N * ((W + tw - 1) / tw) * ((H + th - 1) / th);
    const int cap = two_per_cu ? 2 * DP_BWD_MAX_BLOCKS : DP_BWD_MAX_BLOCKS;
    return (int)(tiles < cap ? tiles : cap);
}
extern "C" int yunet_dp_pool_fusion_ok(int N, int H, int W, int cin, int cout) {
    if ((H & 1) || (W & 1)) return 0;
    if (cin == 16 && cout == 16) return dp_bwd_big_tile(H, W, cin, cout) ? 1 : 0;
    if (cin == 64 && cout == 64) return dp_use_pack_bwd(N, H, W, cin, cout) ? 0 : 1;
    if (cin == 32 && cout == 64) return 1;       // YuNet_s: the unit in front of its 80x80 -> 40x40 pool
    return 0;
}
extern "C" int yunet_stem_bwd_blocks(int N, int H, int W) {
    const long long tiles = (long long)N * ((W / 2 + SB_TW - 1) / SB_TW) * ((H / 2 + SB_TH - 1) / SB_TH);
    return (int)(tiles < STEM_BWD_MAX_BLOCKS ? tiles : STEM_BWD_MAX_BLOCKS);
}
#endif

extern "C" int ACT_SUFFIX(yunet_dp_bwd)(const YunetDP* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (d->x_dtype != YUNET_ACT_DTYPE) return YUNET_EINVAL;
    if (!d->wgrad_partials ||
        d->wgrad_blocks != yunet_dp_bwd_blocks(d->N, d->H, d->W, d->cin, d->cout))
        return YUNET_EINVAL;   // the partial buffer must have exactly the rows the grid writes
    if (d->in_transform != YUNET_T_IDENTITY && d->in_transform != YUNET_T_BNRELU) return YUNET_EINVAL;
    // the 16 -> 16 unit on the 160 x 160 / 80 x 80 levels: wave-streaming kernel that recomputes z from x instead of
    // reading it (conv_bwd16.hip); same grid and partial rows as the tile kernel it replaces
    if (d->cin == 16 && d->cout == 16 && dp_bwd_big_tile(d->H, d->W, 16, 16) && d->out_has_bn && d->dx && !d->accumulate_dx &&
        !d->prof && yunet_options().bwd16s &&
        (!d->pool_idx || (yunet_dp_pool_fusion_ok(d->N, d->H, d->W, 16, 16) && !(reinterpret_cast<uintptr_t>(d->pool_idx) & 3))))
        return ACT_SUFFIX(launch_dp_bwd16s)(d, s);
    if (d->pool_idx) {
        // dy is the pooled gradient + argmax bytes (max_pool2d backward while staging)
        if (!yunet_dp_pool_fusion_ok(d->N, d->H, d->W, d->cin, d->cout) || !d->out_has_bn) return YUNET_EINVAL;
        const bool full816 = d->H % 8 == 0 && d->W % 16 == 0, full1632 = d->H % 16 == 0 && d->W % 32 == 0;
        if (d->cin == 16) return full1632 ? launch_dp_bwd<16, 16, 16, 32, false, 0, true, true>(d, s)
                                          : launch_dp_bwd<16, 16, 16, 32, false, 0, true>(d, s);
        // 32 -> 64 (YuNet_s, in front of its 80 x 80 -> 40 x 40 pool): on the exact-fp32 matrix instruction this unit is
        // MATRIX-bound (12.3 kFLOP per pixel at 157 TFLOP/s: 0.87 ms per 512-image launch = 0.16 of the HBM peak, the
        // slowest kernel of the YuNet_s step); round 5 puts it on the split-bf16 path of the 64 -> 64 units (GEMM = 1)
        if (d->cin == 32 && yunet_options().bwd_fp32mma == 0 && yunet_options().bwd32_split)
            return full816 ? launch_dp_bwd<32, 64, 8, 16, false, 1, true, true>(d, s)
                           : launch_dp_bwd<32, 64, 8, 16, false, 1, true>(d, s);
        if (d->cin == 32) return full816 ? launch_dp_bwd<32, 64, 8, 16, false, 0, true, true>(d, s)
                                         : launch_dp_bwd<32, 64, 8, 16, false, 0, true>(d, s);
        // (option bwd_fp32mma: the exact-fp32 matrix instruction for this instance too -- before round 5 the pooled-dy
        // 64 -> 64 unit stayed on the split-bf16 kernel even with the option set)
        if (yunet_options().bwd_fp32mma != 0) return launch_dp_bwd<64, 64, 8, 16, false, 0, true>(d, s);
        return bwd64_nw(d->N, d->H, d->W) == 4 ? launch_dp_bwd64<4, false, true>(d, s) : launch_dp_bwd64<8, false, true>(d, s);
    }
#define DP_CASE(ci, co) \
    if (d->cin == ci && d->cout == co) return launch_dp_bwd<ci, co, 8, 16>(d, s);
    if (dp_bwd_big_tile(d->H, d->W, d->cin, d->cout))    // 160x160 / 80x80 levels: bigger tile
        return (d->H % 16 == 0 && d->W % 32 == 0) ? launch_dp_bwd<16, 16, 16, 32, false, 0, false, true>(d, s)
                                                  : launch_dp_bwd<16, 16, 16, 32>(d, s);
    // 64 -> 64 units: split-bf16 GEMMs (gradients only); the option bwd_fp32mma keeps the exact-fp32
    // matrix instruction (bench.py's exact_fp32_bwd line, tools/kbench.py)
    const bool f32mma = yunet_options().bwd_fp32mma != 0;
    if (dp_use_pack_bwd(d->N, d->H, d->W, d->cin, d->cout)) {           // 20x20 / 10x10 levels: packed canvas
        if (d->cout == 64) {
            if (f32mma) return launch_dp_bwd<64, 64, 8, 16, true>(d, s);
            return bwd64_nw(d->N, d->H, d->W) == 4 ? launch_dp_bwd64<4, true, false>(d, s) : launch_dp_bwd64<8, true, false>(d, s);
        }
        return launch_dp_bwd<64, 16, 8, 16, true>(d, s);
    }
    if (d->cin == 64 && d->cout == 64 && !f32mma)
        return bwd64_nw(d->N, d->H, d->W) == 4 ? launch_dp_bwd64<4, false, false>(d, s) : launch_dp_bwd64<8, false, false>(d, s);
    if (d->cin == 32 && d->cout == 64 && !f32mma && yunet_options().bwd32_split)      // (plain 32 -> 64: split-bf16 as above)
        return (d->H % 8 == 0 && d->W % 16 == 0) ? launch_dp_bwd<32, 64, 8, 16, false, 1, false, true>(d, s)
                                                 : launch_dp_bwd<32, 64, 8, 16, false, 1>(d, s);
    if (d->H % 8 == 0 && d->W % 16 == 0) {       // whole-tile maps of the 16-channel stages (80 x 80 in the shipped nets)
        if (d->cin == 16 && d->cout == 64) return launch_dp_bwd<16, 64, 8, 16, false, 0, false, true>(d, s);
        if (d->cin == 16 && d->cout == 32) return launch_dp_bwd<16, 32, 8, 16, false, 0, false, true>(d, s);
        if (d->cin == 32 && d->cout == 32) return launch_dp_bwd<32, 32, 8, 16, false, 0, false, true>(d, s);
    }
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

extern "C" int ACT_SUFFIX(yunet_stem_bwd)(const float* img, const float* z, const float* dy, const YunetBN* bn,
                                          float* wgrad_partials, int wgrad_blocks, int N, int H, int W, int cmid,
                                          void* stream) {
    if (cmid != 16 || (H & 1) || (W & 1) || wgrad_blocks != yunet_stem_bwd_blocks(N, H, W))
        return YUNET_EINVAL;
    const int tiles = N * ((W / 2 + SB_TW - 1) / SB_TW) * ((H / 2 + SB_TH - 1) / SB_TH);
    int grid = tiles < CONV_BLOCKS ? tiles : CONV_BLOCKS;
    if (grid > wgrad_blocks) grid = wgrad_blocks;
    hipLaunchKernelGGL(stem_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img,
                       reinterpret_cast<const act_t*>(z), dy, *bn, wgrad_partials, N, H, W);
    return hip_status();
}

#ifndef YUNET_ACT_BF16
// the same weight gradient on the matrix cores with z RECOMPUTED from the image (w [16,3,3,3], b [16]: the stem's
// parameters) instead of read: 112 instead of 176 bytes per output pixel (conv_stem.hip)
extern "C" int yunet_stem_bwd_rz(const float* img, const float* w, const float* b, const float* dy, const YunetBN* bn,
                                 float* wgrad_partials, int wgrad_blocks, int N, int H, int W, int cmid, void* stream) {
    if (cmid != 16 || (H & 1) || (W & 1) || !w || !b || !bn->bstats || wgrad_blocks != yunet_stem_bwd_blocks(N, H, W)) return YUNET_EINVAL;
    return launch_stem_bwd_mma(img, w, b, dy, bn, wgrad_partials, wgrad_blocks, N, H, W, (hipStream_t)stream);
}
#endif

extern "C" int ACT_SUFFIX(yunet_pool_bwd_add)(const float* z, const YunetBN* bn, const float* dy_out, const float* extra,
                                              float* dx, int accumulate, int N, int H, int W, int C, void* stream) {
    if ((H & 1) || (W & 1) || (C & 3) || (256 % (C / 4)) || C > 64) return YUNET_EINVAL;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const act_t*>(z), *bn, dy_out, extra, dx, accumulate, N, H, W, C);
    return hip_status();
}

extern "C" int ACT_SUFFIX(yunet_pool_bwd)(const float* z, const YunetBN* bn, const float* dy_out, float* dx,
                                          int accumulate, int N, int H, int W, int C, void* stream) {
    return ACT_SUFFIX(yunet_pool_bwd_add)(z, bn, dy_out, nullptr, dx, accumulate, N, H, W, C, stream);
}

extern "C" int ACT_SUFFIX(yunet_upadd_bwd)(const float* za, const YunetBN* bna, const float* zb,
                                           const YunetBN* bnb, const float* dout, float* dxa, int accumulate_a,
                                           float* dxb, int accumulate_b, int N, int H, int W, int C,
                                           void* stream) {
    if ((H & 1) || (W & 1) || (C & 3) || (256 % (C / 4)) || C > 64) return YUNET_EINVAL;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    if (!dxa && yunet_options().upadd_coarse) {        // the coarse gradient alone: dedicated kernel (round 5)
        hipLaunchKernelGGL(upadd_bwd_coarse_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const act_t*>(zb), *bnb, dout, dxb, accumulate_b, N, H, W, C);
        return hip_status();
    }
    hipLaunchKernelGGL(upadd_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const act_t*>(za), *bna, reinterpret_cast<const act_t*>(zb), *bnb, dout, dxa,
                       accumulate_a, dxb, accumulate_b, N, H, W, C);
    return hip_status();
}

#ifndef YUNET_ACT_BF16
extern "C" int yunet_bn_param_grad(const double* bstats, float* dgamma, float* dbeta, int C,
                                   int accumulate, void* stream) {
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       bstats, dgamma, dbeta, C, accumulate);
    return hip_status();
}

extern "C" int yunet_reduce_partials(const float* partials, int blocks, int width, float* out,
                                     int accumulate, void* stream) {
    if (blocks < 1 || width < 1) return YUNET_EINVAL;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((width + 63) / 64), dim3(1024), 0,
                       (hipStream_t)stream, partials, blocks, width, out, accumulate);
    return hip_status();
}

extern "C" int yunet_reduce_partials_batch(const YunetReduceJob* jobs, int njobs, int total_chunks,
                                           void* stream) {
    if (!jobs || njobs < 1 || total_chunks < njobs) return YUNET_EINVAL;
    hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3(total_chunks), dim3(1024), 0,
                       (hipStream_t)stream, jobs, njobs);
    return hip_status();
}
#endif
