// conv_fwd16.hip -- forward of the ConvDPUnits with 16 input channels (16 -> 16, 16 -> 64; yunet_layer.py:30-36) as
// WAVE-STREAMING kernels, built from the pieces of conv_bwd16.hip / conv_stem.hip.
//
// The tile kernels they replace (dp_fwd_kernel<16,16,16,32>, <16,64,8,16>) run at 2.3 - 3.5 TB/s of algorithmic bytes;
// the round's measurements say why: these units are instruction-bound (VALU + LDS issue, barrier-separated phases), and the
// stem -- rebuilt with the convolution on the matrix cores and almost no VALU work -- went from 3.2 to 5.3 TB/s.
//
// One WAVE owns a strip of the image (32 columns for 16 output channels, 16 columns for 64) and streams down a band of
// rows; no workgroup barrier in the main loop.  Two register layouts, converted through per-wave LDS slots:
//   P (pixel-major): lane (g, l15): pixel l15 of a 16-pixel tile, 4 consecutive channels -- what a 16-byte global access
//     delivers and the B / D layout of v_mfma_f32_16x16x4_f32 with the weights as the A operand;
//   C (channel-major): lane = channel (x pixel segment): the depthwise layout -- a pixel's column neighbours sit in the same
//     lane, the nine taps are nine scalars, the BatchNorm sums two registers.
// Per step X:  P: a = relu(bn(x(X))), p(X) = W1 a + b1 (exact fp32 matrix instruction) -> slot;  C: p(X) is the bottom /
// middle / top tap row of z(X - 1), z(X), z(X + 1) (accumulate form): z(X - 1) is complete -> BN sums, -> slot -> P ->
// 16-byte stores.  POOL (16 -> 16 units that feed max_pool2d only, YunetDP.pool_out): the 2 x 2 winners after BN + ReLU
// and their window positions (dp_fwd_kernel<..., POOL>'s rule), from column pairs a lane owns and the row pair a band holds.
#include "common.h"
#ifndef YUNET_FWD16_X_AUX       // cache-policy bits of the x loads (round 6 measurement switch; 2 = non-temporal)
#define YUNET_FWD16_X_AUX 0
#endif

namespace {
namespace f16s {
constexpr int CIN = 16, WAVES = 4, NTHR = 64 * WAVES;
template <int COUT>
struct G {
    static constexpr int PXW = COUT == 16 ? 32 : 16;       // strip width (loaded pixels per row)
    static constexpr int NT = PXW / 16, MT = COUT / 16;
    static constexpr int PST = COUT + 4;                    // floats per pixel of a slot row
    static constexpr int SLOT = PXW * PST;
    static constexpr int WAVE_F = 2 * SLOT + 64;            // slot A (p row | pooled row), slot Z (z row), pooled position bytes [16][16]
};
}  // namespace f16s

template <int COUT, bool POOL>
__global__ __launch_bounds__(f16s::NTHR) __attribute__((amdgpu_waves_per_eu(3, 3)))      // <= 168 registers: three workgroups per CU
void dp_fwd16s_kernel(const YunetDP d, const int R) {
    using namespace f16s;
    using GG = G<COUT>;
    constexpr int PXW = GG::PXW, NT = GG::NT, MT = GG::MT, PST = GG::PST, SLOT = GG::SLOT, WAVE_F = GG::WAVE_F;
    constexpr int NSEG = PXW * COUT / 64;                   // pixels per lane in layout C: 8 (16 channels) | 16 (64 channels)
    constexpr int HALO = POOL ? 2 : 1, OUTW = PXW - 2 * HALO;
    static_assert(!POOL || COUT == 16, "fused pooling: the 16 -> 16 unit");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sm = reinterpret_cast<float*>(smem);
    float* s_w2 = sm + WAVES * WAVE_F;                      // [9][COUT]
    float* s_b2 = s_w2 + 9 * COUT;
    float* s_b1 = s_b2 + COUT;
    float* s_in = s_b1 + COUT;                              // [3][16] mean | scale | beta of the producer's BN
    double* s_st = reinterpret_cast<double*>(s_in + 3 * CIN);   // [2][COUT] (the float offset up to here is even: 8-byte aligned)
    static_assert(((WAVES * WAVE_F + 11 * COUT + 3 * CIN) & 1) == 0, "fp64 alignment");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;               // layout P
    const int cc = lane % COUT, sg = lane / COUT;           // layout C: channel, pixel segment
    const int H = d.H, W = d.W;
    const bool bn_in = d.in_transform == YUNET_T_BNRELU;

    staged_table<COUT * 9, NTHR>(d.w_dw, tid, [&](int i, float w) { s_w2[(i % 9) * COUT + i / 9] = w; });
    for (int i = tid; i < COUT; i += NTHR) { s_b2[i] = d.b_dw[i]; s_b1[i] = d.b_pw[i]; }
    if (tid < CIN) {
        if (bn_in) {
            const BNCoef k = bn_coef(d.in_bn, CIN, tid);
            s_in[tid] = k.mean; s_in[CIN + tid] = k.scale; s_in[2 * CIN + tid] = k.beta;
        } else {
            s_in[tid] = 0.f; s_in[CIN + tid] = 1.f; s_in[2 * CIN + tid] = 0.f;
        }
    }
    for (int i = tid; i < 2 * COUT; i += NTHR) s_st[i] = 0.0;
    // A operands of the pointwise product: W1[co = 16 mt + l15][ci = 4 g + s]
    float w1a[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 4; ++s) w1a[mt][s] = d.w_pw[(16 * mt + l15) * CIN + 4 * g + s];
    __syncthreads();
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = s_w2[k * COUT + cc];
    const float b2c = s_b2[cc];
    const float4 im = *reinterpret_cast<const float4*>(s_in + 4 * g), isc = *reinterpret_cast<const float4*>(s_in + CIN + 4 * g),
                 ibt = *reinterpret_cast<const float4*>(s_in + 2 * CIN + 4 * g);
    const float relu_floor = bn_in ? 0.0f : -__builtin_inff();
    float sgn = 1.0f;                                         // fused pooling: which raw value wins a window after BN + ReLU
    if constexpr (POOL) {
        const float gm = d.out_bn.gamma[cc];
        sgn = gm > 0.0f ? 1.0f : (gm < 0.0f ? -1.0f : 0.0f);
    }
    float* slot_a = sm + wid * WAVE_F;
    float* slot_z = slot_a + SLOT;
    unsigned char* slot_i = reinterpret_cast<unsigned char*>(slot_z + SLOT);     // [16 pooled pixels][16 channels]

    float zacc0[NSEG], zacc1[NSEG], zprev[POOL ? NSEG : 1];
#pragma unroll
    for (int k = 0; k < NSEG; ++k) zacc0[k] = zacc1[k] = 0.0f;

    const int strips = (W + OUTW - 1) / OUTW, bands = (H + R - 1) / R;
    const int tasks_img = strips * bands, ntasks = d.N * tasks_img;
    // activation storage of this build (common.h: act_t = float | bf16): x, z and the pooled winners
    const unsigned xbytes = (unsigned)(H * W * CIN) * ACT_B, zbytes = (unsigned)(H * W * COUT) * ACT_B;
    const unsigned poel = (unsigned)((H >> 1) * (W >> 1) * COUT), pobytes = poel * ACT_B;
    const int total_waves = (int)gridDim.x * WAVES;

    for (int task = first_tile() * WAVES + wid; task < ntasks; task += total_waves) {
        const int n = task / tasks_img, rr = task - n * tasks_img;
        const int band = rr / strips, strip = rr - band * strips;
        const int y0 = band * R, y1 = (y0 + R < H) ? y0 + R : H;
        const int xs = strip * OUTW - HALO;                   // image column of the strip's pixel 0
        const auto r_x = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(const_cast<float*>(d.x)) + (size_t)n * d.x_img_stride, 0, xbytes, 0x00020000);
        const auto r_z = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(d.z) + (size_t)n * d.z_img_stride, 0, zbytes, 0x00020000);
        const auto r_po = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<act_t*>(d.pool_out) + (POOL ? (size_t)n * poel : (size_t)0), 0, POOL ? pobytes : 0u, 0x00020000);
        const auto r_pi = __builtin_amdgcn_make_buffer_rsrc(d.pool_idx + (POOL ? (size_t)n * poel : (size_t)0), 0, POOL ? poel : 0u, 0x00020000);
        // layout C: bit k = pixel NSEG sg + k is an output pixel of this strip
        unsigned omask = 0;
#pragma unroll
        for (int k = 0; k < NSEG; ++k) {
            const int j = NSEG * sg + k;
            omask |= ((unsigned)(xs + j) < (unsigned)W && j >= HALO && j < PXW - HALO) ? (1u << k) : 0u;
        }
        auto colP = [&](int nt) { return xs + 16 * nt + l15; };
        act_raw4 lx[NT];
        auto issue_x = [&](int y) {
            const bool yin = (unsigned)y < (unsigned)H;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                lx[nt] = act_bufld4_aux<YUNET_FWD16_X_AUX>(r_x, (yin && (unsigned)colP(nt) < (unsigned)W) ? (unsigned)((y * W + colP(nt)) * CIN + 4 * g) * ACT_B : xbytes);
        };
        issue_x(y0 - 1);
        float ts0 = 0.0f, ts1 = 0.0f;                         // BN sums of the band: channel cc over the lane's output pixels

#pragma unroll 1
        for (int X = y0 - 1; X <= y1; ++X) {
            const bool xin = (unsigned)X < (unsigned)H;
            // ---- [P] a = T(x(X)); p(X) = W1 a + b1 -> slot A ------------------------------------------------------------
            {
                float4 xr[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) xr[nt] = act_unpack(lx[nt]);
                issue_x(X + 1);                // (the row past the band's last one is fetched and never used)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float a0 = fmaxf(fmaf(xr[nt].x - im.x, isc.x, ibt.x), relu_floor);
                    const float a1 = fmaxf(fmaf(xr[nt].y - im.y, isc.y, ibt.y), relu_floor);
                    const float a2 = fmaxf(fmaf(xr[nt].z - im.z, isc.z, ibt.z), relu_floor);
                    const float a3 = fmaxf(fmaf(xr[nt].w - im.w, isc.w, ibt.w), relu_floor);
                    const bool ok = xin && (unsigned)colP(nt) < (unsigned)W;       // zero padding of the depthwise input
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float4 bq = *reinterpret_cast<const float4*>(s_b1 + 16 * mt + 4 * g);
                        f32x4 acc = {bq.x, bq.y, bq.z, bq.w};
                        acc = mfma16(w1a[mt][0], a0, acc);
                        acc = mfma16(w1a[mt][1], a1, acc);
                        acc = mfma16(w1a[mt][2], a2, acc);
                        acc = mfma16(w1a[mt][3], a3, acc);
                        const float4 pv = make_float4(ok ? acc[0] : 0.0f, ok ? acc[1] : 0.0f, ok ? acc[2] : 0.0f, ok ? acc[3] : 0.0f);
                        *reinterpret_cast<float4*>(slot_a + (16 * nt + l15) * PST + 16 * mt + 4 * g) = pv;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- [C] p(X) of channel cc, pixels NSEG sg .. + NSEG - 1 (+ the two neighbours outside the segment); z ------
            float zdone[NSEG];
            {
                float pc[NSEG];
                const float* pp = slot_a + (NSEG * sg) * PST + cc;
#pragma unroll
                for (int k = 0; k < NSEG; ++k) pc[k] = pp[k * PST];
                const float pLe = slot_a[(NSEG * sg > 0 ? NSEG * sg - 1 : 0) * PST + cc];
                const float pRe = slot_a[(NSEG * sg + NSEG < PXW ? NSEG * sg + NSEG : PXW - 1) * PST + cc];
#pragma unroll
                for (int k = 0; k < NSEG; ++k) {
                    const float l = k > 0 ? pc[k - 1] : pLe, m = pc[k], r = k < NSEG - 1 ? pc[k + 1] : pRe;
                    zdone[k] = fmaf(r, wt[8], fmaf(m, wt[7], fmaf(l, wt[6], zacc0[k])));
                    zacc0[k] = fmaf(r, wt[5], fmaf(m, wt[4], fmaf(l, wt[3], zacc1[k])));
                    zacc1[k] = fmaf(r, wt[2], fmaf(m, wt[1], fmaf(l, wt[0], b2c)));
                }
            }
            if (X >= y0 + 1) {
                const int q = X - 1;                          // the completed output row
                // ---- [C] BN sums over the output pixels; z -> slot Z -> [P] 16-byte stores ---------------------------------
                float* zq = slot_z + (NSEG * sg) * PST + cc;
#pragma unroll
                for (int k = 0; k < NSEG; ++k) {
                    const float v = ((omask >> k) & 1u) ? zdone[k] : 0.0f;
                    ts0 += v;
                    ts1 = fmaf(v, v, ts1);
                    zq[k * PST] = zdone[k];
                }
                if constexpr (POOL) {
                    if ((q & 1) == 0) {
#pragma unroll
                        for (int k = 0; k < NSEG; ++k) zprev[k] = zdone[k];
                    } else {
                        // window (rows q - 1, q; pixels k, k + 1 with k even): pooled pixel (NSEG sg + k) / 2 of the strip
#pragma unroll
                        for (int k = 0; k < NSEG; k += 2) {
                            const float tl = zprev[k], tr = zprev[k + 1], bl = zdone[k], br = zdone[k + 1];
                            const bool lowl = bl * sgn > tl * sgn, lowr = br * sgn > tr * sgn;
                            const float vl = lowl ? bl : tl, vr = lowr ? br : tr;
                            const unsigned jl = lowl ? 2u : 0u, jr = lowr ? 3u : 1u;
                            const float kl = vl * sgn, kr = vr * sgn;
                            const bool right = kr > kl || (kr == kl && jr < jl);
                            const int pp = (NSEG * sg + k) >> 1;
                            slot_a[pp * PST + cc] = right ? vr : vl;             // (slot A: the p row has been read)
                            slot_i[pp * 16 + cc] = (unsigned char)(right ? jr : jl);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int j = 16 * nt + l15;
                    const bool mine = (unsigned)colP(nt) < (unsigned)W && j >= HALO && j < PXW - HALO;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float4 v = *reinterpret_cast<const float4*>(slot_z + j * PST + 16 * mt + 4 * g);
                        act_bufst4(r_z, mine ? (unsigned)((q * W + colP(nt)) * COUT + 16 * mt + 4 * g) * ACT_B : zbytes, v);
                    }
                }
                if constexpr (POOL) {
                    if (q & 1) {
                        // pooled pixel pp = l15 of the strip (16 per row pair): columns xs + 2 pp, + 1
                        const int j = 2 * l15, col = xs + j;
                        const bool mine = (unsigned)(col + 1) < (unsigned)W && col >= 0 && j >= HALO && j < PXW - HALO;
                        const unsigned eq = (unsigned)(((q >> 1) * (W >> 1) + (col >> 1)) * COUT + 4 * g);
                        const float4 v = *reinterpret_cast<const float4*>(slot_a + l15 * PST + 4 * g);
                        const unsigned id = *reinterpret_cast<const unsigned*>(slot_i + l15 * 16 + 4 * g);
                        act_bufst4(r_po, mine ? eq * ACT_B : pobytes, v);
                        __builtin_amdgcn_raw_buffer_store_b32(id, r_pi, mine ? eq : poel, 0, 0);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the slots are free for the next row
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- BN sums of the band: over the pixel segments of a channel, then fp64 in LDS ----------------------------------
        if (d.out_has_bn) {
            float a = ts0, b = ts1;
            if constexpr (COUT == 16) {
                a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
                a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
            }
            if (sg == 0) {
                __hip_atomic_fetch_add(&s_st[cc], (double)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&s_st[COUT + cc], (double)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (d.out_has_bn) {
        __syncthreads();
        for (int i = tid; i < 2 * COUT; i += NTHR) atomic_add_f64(bn_slot(d.out_bn.stats, d.out_bn.slots, COUT) + i, s_st[i]);
    }
}

// rows per band: minimise (tasks per wave, rounded up) x (rows of a task incl. the two halo rows + set-up)
int fwd16s_rows(int N, int H, int W, int outw, int waves, bool even) {
    const long long strips = (W + outw - 1) / outw;
    int best = H;
    long long best_cost = -1;
    for (int R = even ? 4 : 3; R <= H; R += even ? 2 : 1) {
        const long long tasks = (long long)N * strips * ((H + R - 1) / R);
        const long long cost = ((tasks + waves - 1) / waves) * (R + 4);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && R > best)) { best_cost = cost; best = R; }
    }
    return best;
}

template <int COUT, bool POOL>
int launch_fwd16s(const YunetDP* d, hipStream_t stream) {
    using GG = f16s::G<COUT>;
    constexpr size_t smem = ((size_t)f16s::WAVES * GG::WAVE_F + 11 * COUT + 3 * 16) * 4 + 2 * COUT * 8;
    static PerDevice per_cu;        // resident workgroups per CU, per device (common.h)
    const int blocks_per_cu = per_device(per_cu, [] {
        const void* fn = reinterpret_cast<const void*>(dp_fwd16s_kernel<COUT, POOL>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -1;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, f16s::NTHR, smem) != hipSuccess || nb < 1) nb = 1;
        return nb > 3 ? 3 : nb;
    });
    if (blocks_per_cu < 1) return YUNET_EINVAL;
    if ((long long)d->H * d->W * COUT * 4 >= (1ll << 31)) return YUNET_EINVAL;      // 32-bit byte offsets per image
    if (d->x_dtype != YUNET_ACT_DTYPE || d->z_dtype != YUNET_ACT_DTYPE) return YUNET_EINVAL;
    constexpr int HALO = POOL ? 2 : 1, OUTW = GG::PXW - 2 * HALO;
    int grid = yunet_cu_count() * blocks_per_cu;
    const int R = fwd16s_rows(d->N, d->H, d->W, OUTW, grid * f16s::WAVES, POOL);
    const long long tasks = (long long)d->N * ((d->W + OUTW - 1) / OUTW) * ((d->H + R - 1) / R);
    const long long need = (tasks + f16s::WAVES - 1) / f16s::WAVES;
    if (need < grid) grid = (int)need;
    hipLaunchKernelGGL((dp_fwd16s_kernel<COUT, POOL>), dim3(grid), dim3(f16s::NTHR), smem, stream, *d, R);
    return hip_status();
}

}  // namespace

// conv_fwd.hip's dispatcher: units with 16 input channels and 16 or 64 output channels (fused pooling: 16 -> 16); compiled
// once per activation storage type (fp32 | -DYUNET_ACT_BF16: x, z and the pooled winners as bf16, same arithmetic)
int ACT_SUFFIX(launch_dp_fwd16s)(const YunetDP* d, hipStream_t stream) {
    if (d->cout == 16) return d->pool_out ? launch_fwd16s<16, true>(d, stream) : launch_fwd16s<16, false>(d, stream);
    return launch_fwd16s<64, false>(d, stream);
}
