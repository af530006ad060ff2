// common.h -- shared device helpers for the YuNet gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/yunet_hip.h"

// Every conv-stack kernel is launched as a persistent grid of CONV_BLOCKS workgroups that
// stride over tiles; per-workgroup partial sums (BN statistics, weight gradients) are
// flushed once at the end, which bounds the number of same-address atomics / partial rows.
#define CONV_BLOCKS 1024

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---- activation storage type of this translation unit ------------------------------------------
// conv_fwd.hip / conv_bwd.hip are compiled twice: as is (fp32 activations, the headline path) and with
// -DYUNET_ACT_BF16 (BASELINE.json configs[2]: "bf16 fwd / fp32 grads").  In the second build every
// ACTIVATION tensor a forward kernel writes (raw conv outputs z, pool / upsample-add outputs) is stored
// as bf16 (round to nearest even) and every kernel that reads one widens it on load; gradients (dy, dx),
// the head output `flat`, BatchNorm sums (taken from the fp32 values before rounding), weights and the
// optimizer stay fp32.  The exported entry points of that build carry the suffix _bf16.
#ifdef YUNET_ACT_BF16
typedef unsigned short act_t;
typedef u32x2 act_raw4;                     // raw bits of 4 consecutive channels
#define ACT_SUFFIX(name) name##_bf16
#define YUNET_ACT_DTYPE YUNET_BF16
#else
typedef float act_t;
typedef u32x4 act_raw4;
#define ACT_SUFFIX(name) name
#define YUNET_ACT_DTYPE YUNET_F32
#endif
#define ACT_B ((unsigned)sizeof(act_t))

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {      // v_cvt_pk_bf16_f32 (RNE)
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float4 act_unpack(const act_raw4 r) {
#ifdef YUNET_ACT_BF16
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
#else
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
#endif
}
__device__ __forceinline__ act_raw4 act_pack(const float4 v) {
#ifdef YUNET_ACT_BF16
    return act_raw4{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
#else
    return act_raw4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#endif
}
// Cache policy of the big streaming stores: bit 1 = "nt" (non-temporal) on gfx950.  Round 6, same-box alternations of bench.py
// (profiles/r06_bench_ab_nt*.log): the dx stores of the BACKWARD units on the unpacked maps (>= 40 x 40: 105 MB - 1 GB per
// tensor, nothing of it survives in a cache until its consumer runs) are better kept out of the 4 MB L2 of their XCD, where
// they push out the halo rows the neighbouring tiles are about to re-read: dp_bwd64 -1.5 .. -3 %, step -0.5 %.  The forward
// units' z stores are NOT: on the 40 x 40 and smaller maps the next forward kernel finds them in the L2 / infinity cache, and
// with nt the ten plain 64 -> 64 forward launches got 4 % slower.  YUNET_ST_AUX: forward stores (default policy);
// YUNET_DX_AUX: backward dx stores on unpacked maps.
#ifndef YUNET_ST_AUX
#define YUNET_ST_AUX 0
#endif
#ifndef YUNET_DX_AUX
#define YUNET_DX_AUX 2
#endif
// 4 consecutive channels through a buffer descriptor (byte offset; out-of-range offsets read 0 / drop)
template <typename R>
__device__ __forceinline__ act_raw4 act_bufld4(R rsrc, unsigned byte_off) {
#ifdef YUNET_ACT_BF16
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, 0);
#else
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
#endif
}
// the same load with cache-policy bits (AUX = 2: non-temporal -- a tensor a kernel reads exactly once)
template <int AUX, typename R>
__device__ __forceinline__ act_raw4 act_bufld4_aux(R rsrc, unsigned byte_off) {
#ifdef YUNET_ACT_BF16
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, AUX);
#else
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, AUX);
#endif
}
template <typename R>
__device__ __forceinline__ void act_bufst4(R rsrc, unsigned byte_off, const float4 v) {
#ifdef YUNET_ACT_BF16
    __builtin_amdgcn_raw_buffer_store_b64(act_pack(v), rsrc, byte_off, 0, YUNET_ST_AUX);
#else
    __builtin_amdgcn_raw_buffer_store_b128(act_pack(v), rsrc, byte_off, 0, YUNET_ST_AUX);
#endif
}
// the same through plain pointers (element-wise kernels)
__device__ __forceinline__ float4 act_ld4(const act_t* p) {
    return act_unpack(*reinterpret_cast<const act_raw4*>(p));
}
__device__ __forceinline__ void act_st4(act_t* p, const float4 v) {
    *reinterpret_cast<act_raw4*>(p) = act_pack(v);
}

// Persistent tile kernels: hardware hands workgroup ids to the 8 XCDs round-robin (id % 8), so with
// "tile = blockIdx.x + k * gridDim.x" the spatial neighbours of a tile (which share its halo rows /
// columns) always run on OTHER XCDs and the halo is fetched once per L2.  first_tile() renumbers the
// workgroups so that each XCD walks a contiguous run of tiles: neighbours are in flight on the same
// XCD at the same time and the second reader of a halo line hits that XCD's L2.
#ifndef YUNET_XCD_REMAP
#define YUNET_XCD_REMAP 1
#endif
__device__ __forceinline__ int first_tile() {
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    if (YUNET_XCD_REMAP && (g & 7) == 0) return (b & 7) * (g >> 3) + (b >> 3);
    return b;
}

// D = A(16x4) * B(4x16) + C, exact fp32 (v_mfma_f32_16x16x4_f32).
// lane l supplies A[row = l&15][k = l>>4], B[k = l>>4][col = l&15];
// D/C: col = l&15, row = (l>>4)*4 + reg.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Per-channel constants of a train-mode BatchNorm, derived from the producer's fp64 sums.
//   y = (x - mean) * scale + beta,  scale = gamma * invstd,  invstd = 1/sqrt(var_biased + eps)
struct BNCoef {
    float mean, scale, beta, invstd;
    float mean_lo;   // mean = mean + mean_lo to ~fp64 accuracy (see bn_center)
};
// sum i of a [slots][2C] block of BN sums (YunetBN::slots), and the replica workgroup blockIdx.x adds into
// The first eight replicas are loaded unconditionally (replicas past `slots` re-read replica 0 and count as
// zero): eight independent loads in flight.  As a loop over `slots` the loads were issued one latency after the
// other at the start of every workgroup, and eight replicas made the 20 x 20 backward launch 9 us SLOWER.
__device__ __forceinline__ double bn_sum(const double* __restrict__ s, int slots, int C, int i) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = s[(k < slots ? k : 0) * 2 * C + i];
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = k < slots ? v[k] : 0.0;
    double r = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    for (int k = 8; k < slots; ++k) r += s[k * 2 * C + i];
    return r;
}
__device__ __forceinline__ double* bn_slot(const double* base, int slots, int C) {
    return const_cast<double*>(base) + (slots > 1 ? (int)(blockIdx.x % (unsigned)slots) * 2 * C : 0);
}
__device__ __forceinline__ BNCoef bn_coef(const YunetBN& bn, int C, int c) {
    const double inv = 1.0 / (double)bn.count;
    const double mean = bn_sum(bn.stats, bn.slots, C, c) * inv;
    double var = bn_sum(bn.stats, bn.slots, C, C + c) * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    BNCoef k;
    k.mean = (float)mean;
    k.mean_lo = (float)(mean - (double)k.mean);
    k.invstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    k.scale = bn.gamma[c] * k.invstd;
    k.beta = bn.beta[c];
    return k;
}
// The element-wise kernels (pool / upsample-add) derive the coefficients ONCE per workgroup: thread c < C computes
// channel c into an LDS table [5][C] (mean | scale | beta | invstd | mean_lo) and every thread picks up its four
// channels after the barrier.  (Per thread -- 4 or 8 bn_coef() calls of 16 loads + fp64 sqrt / divide each, in
// every one of up to 4096 workgroups -- the prologue cost more than the streaming: upadd_fwd ran at 1.4 TB/s.)
__device__ __forceinline__ void bn_table_fill(float* tab, const YunetBN& bn, int C, int tid) {
    if (tid < C) {
        const BNCoef k = bn_coef(bn, C, tid);
        tab[tid] = k.mean; tab[C + tid] = k.scale; tab[2 * C + tid] = k.beta; tab[3 * C + tid] = k.invstd;
        tab[4 * C + tid] = k.mean_lo;
    }
}
__device__ __forceinline__ void bn_table_get(const float* tab, int C, int c0, BNCoef (&k)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k[i].mean = tab[c0 + i]; k[i].scale = tab[C + c0 + i]; k[i].beta = tab[2 * C + c0 + i];
        k[i].invstd = tab[3 * C + c0 + i]; k[i].mean_lo = tab[4 * C + c0 + i];
    }
}

// backward constants: dz = k1 * (dy - c1 - xhat * c2), xhat = (z - mean) * invstd
struct BNBwd {
    float mean, invstd, k1, c1, c2;
    float mean_lo, c1_lo;
};
__device__ __forceinline__ BNBwd bn_bwd_coef(const YunetBN& bn, int C, int c) {
    const BNCoef f = bn_coef(bn, C, c);
    const double inv = 1.0 / (double)bn.count;
    BNBwd k;
    k.mean = f.mean;
    k.invstd = f.invstd;
    k.k1 = f.scale;
    const double c1 = bn_sum(bn.bstats, bn.slots, C, c) * inv;
    k.c1 = (float)c1;
    k.c1_lo = (float)(c1 - (double)k.c1);
    k.c2 = (float)(bn_sum(bn.bstats, bn.slots, C, C + c) * inv);
    k.mean_lo = f.mean_lo;
    return k;
}

// The same backward as  dz = A dy + B z + D:  A = k1, B = -k1 c2 invstd rounded to fp32, and
// D = -A c1 - B mean evaluated in fp64 WITH the rounded B and carried as a (hi, lo) pair.  The rounding of B
// then scales the centred term (z - mean) by 1 + 6e-8 instead of shifting every dz of the channel by
// 6e-8 |B mean| -- the systematic offset that breaks sum(dz) = 0 (see bn_center below).  Two FMAs and an add
// per element; measured on the 80 x 80 backward harness, |sum dz| is 30x smaller than with the nine-operation
// form (tools/ubench/bwd_ab: max |db2| 4.1e-4 vs 1.2e-2).
struct BNFold {
    float a, b, dh, dl;
};
__device__ __forceinline__ BNFold bn_fold(const BNBwd& k) {
    BNFold f;
    f.a = k.k1;
    f.b = (float)(-(double)k.k1 * (double)k.c2 * (double)k.invstd);
    const double D = -(double)f.a * ((double)k.c1 + (double)k.c1_lo) - (double)f.b * ((double)k.mean + (double)k.mean_lo);
    f.dh = (float)D;
    f.dl = (float)(D - (double)f.dh);
    return f;
}
__device__ __forceinline__ float bn_dz_folded(float dy, float z, float a, float b, float dh, float dl) {
    return fmaf(a, dy, fmaf(b, z, dh)) + dl;
}

// x - mean with the mean carried as a (hi, lo) float pair.  A plain float mean is off by up to
// 6e-8*|mean| for EVERY element of the channel, which makes sum(xhat) and sum(dz) of the BN
// backward non-zero by cnt*6e-8*|mean|/sigma; multiplied by mean(p) (depthwise weight gradient)
// or by the image mean 127 (stem weight gradient) that systematic term reaches 0.3-1 % -- the
// reference's fp32 path shows the same effect.  One extra subtraction removes it.
__device__ __forceinline__ float bn_center(float x, float mean_hi, float mean_lo) {
    return (x - mean_hi) - mean_lo;
}
// dz = k1 * (dy - c1 - xhat * c2),  xhat = (z - mean) * invstd, with (hi, lo) mean and c1
__device__ __forceinline__ float bn_dz(float dy, float z, float mean_hi, float mean_lo, float invstd,
                                       float k1, float c1_hi, float c1_lo, float c2) {
    return k1 * (((dy - c1_hi) - c1_lo) - bn_center(z, mean_hi, mean_lo) * invstd * c2);
}

__device__ __forceinline__ float bnrelu(float x, float mean, float scale, float beta) {
    return fmaxf(fmaf(x - mean, scale, beta), 0.0f);
}

// Measurement switches of the dispatchers (A/B runs of tools/kbench.py, bench.py's exact-fp32 backward line).
// One process-wide record, filled from the environment (YUNET_NO_PACK, YUNET_BWD_FP32MMA, YUNET_BWD64_NW,
// YUNET_EW_GRID, YUNET_DP_FWD_BLOCKS_PER_CU) the first time it is used and changed afterwards only through
// yunet_set_option() (api.hip): no launch reads the environment.
struct YunetOptions {
    int no_pack;             // 1: the 20 x 20 / 10 x 10 levels on per-image tiles instead of the packed canvas
    int bwd_fp32mma;         // 1: every backward GEMM on the exact-fp32 matrix instruction (64 -> 64 units too)
    int bwd64_nw;            // 0 = by shape, 4 | 8: waves per workgroup of dp_bwd64
    int ew_grid;             // workgroup cap of the element-wise backward kernels (default 768)
    int fwd_blocks_per_cu;   // 0 = occupancy API, 1..4: resident workgroups per CU of dp_fwd
    int fwd64s;              // >= 1: the plain 64 -> 64 forward unit on the wave-streaming kernel (conv_fwd64.hip);
                             // 2 (default): also on the 20 x 20 / 10 x 10 levels instead of the packed tile kernel
    int fwd64s_rows;         // 0 = by shape, else rows per band of that kernel
    int bwd16s;              // 1 (default): the fp32 16 -> 16 backward unit on big maps on the wave-streaming kernel that
                             // recomputes z (conv_bwd16.hip); 0: the tile kernel
    int bwd16s_rows;         // 0 = by shape, else rows per band of that kernel
    int fwd16s;              // 1 (default): the fp32 16 -> 16 / 16 -> 64 forward units on the wave-streaming kernel (conv_fwd16.hip)
    int stem_mma;            // 1 (default): the fp32 stem forward / weight gradient on the matrix cores (conv_stem.hip; the
                             // backward recomputes z from the image); 0: the VALU tile kernels
    int bwd32_split;         // 1 (default): the 32 -> 64 backward unit (YuNet_s) on the split-bf16 matrix path of the 64 -> 64 units
    int upadd_coarse;        // 1 (default): yunet_upadd_bwd without a fine-tensor share (dxa = NULL) on the dedicated kernel
    int fwd_group;           // 1 (default): yunet_dp_fwd_group launches independent plain 64 -> 64 units as one grid; 0: one by one
    int assign_v2;           // 1 (default): the SimOTA assignment on the chunk-balanced / candidate-pruned launches
                             // (loss_step.hip, round 5); 0: one workgroup per image + full evaluation of every pair
    int oneshot_timeout_ms;  // how long yunet_allreduce waits for a peer before it poisons the buffer and sets the status
                             // word (default 120 000 = 2 min; tests use 1 000)
};
YunetOptions& yunet_options();
int yunet_option_assign_v2();      // (for loss_step.hip, which does not include this header)

// ---- per-device launch set-up ------------------------------------------------------------------------------------
// hipFuncSetAttribute (the raised dynamic-LDS limit) and the occupancy query are PER DEVICE, and executor lanes are per
// (host thread, device): a process may drive several devices from several threads.  Launchers therefore keep what they
// computed in one slot per device (process-wide statics left a second device without the raised limit).  A
// slot holds 0 until its device has been set up; two threads racing on the same slot compute the same value.
#include <atomic>
constexpr int YUNET_MAX_DEVICES = 16;
struct PerDevice {
    std::atomic<int> v[YUNET_MAX_DEVICES];
};
inline int yunet_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= YUNET_MAX_DEVICES) return -1;
    return dev;
}
// value cached for the current device; `compute` returns > 0, or <= 0 for failure (not cached, passed through)
template <typename F>
inline int per_device(PerDevice& slots, F&& compute) {
    const int dev = yunet_device_index();
    if (dev < 0) return compute();
    int v = slots.v[dev].load(std::memory_order_acquire);
    if (v == 0) {
        v = compute();
        if (v > 0) slots.v[dev].store(v, std::memory_order_release);
    }
    return v;
}
// compute units of the current device (256 on MI355X); the persistent grids are a multiple of it
inline int yunet_cu_count() {
    static PerDevice cus;
    return per_device(cus, [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
            return 256;
        return n;
    });
}

#ifdef __HIPCC__
// Prologue copy of a weight table from global memory: thread `tid` takes elements tid, tid + NT, ...  ALL loads are issued
// before the first value is used.  Written as the obvious loop (`for (i = tid; i < COUNT; i += NT) lds[f(i)] = src[i]`)
// the compiler waits out every load before it issues the next one (s_waitcnt vmcnt(0) in front of each LDS store): the
// 64 x 64 pointwise weights of dp_fwd64s were 16 consecutive L2 / HBM round trips per workgroup, ~10 us in front of
// every launch -- half the duration of the 10 x 10 launches.
template <int COUNT, int NT, typename F>
__device__ __forceinline__ void staged_table(const float* __restrict__ src, int tid, F&& put) {
    constexpr int IT = (COUNT + NT - 1) / NT;
    constexpr bool EXACT = COUNT % NT == 0;
    float v[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = tid + k * NT;
        v[k] = (EXACT || i < COUNT) ? src[i] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = tid + k * NT;
        if (EXACT || i < COUNT) put(i, v[k]);
    }
}
#endif


// Small feature maps (the 20x20 / 10x10 pyramid levels) waste most of an 8x16 tile per image.
// For them the tile grid is laid over a virtual CANVAS on which the images of the batch sit side
// by side, R per row, one zero gap column / row between neighbours (pitch = size + 1): a tile then
// covers pieces of several images, the gap pixels play the role of the zero padding, and
// pk_locate maps a canvas pixel back to (image, y, x).  `on` = 0 keeps the per-image tiling.
struct PackGeom {
    int on, H, W, ph, pw, R, N, CH, CW;
    float inv_ph, inv_pw;
};
static inline PackGeom make_pack(int N, int H, int W) {
    PackGeom g;
    g.on = (H <= 20 && W <= 20 && N >= 4) ? 1 : 0;     // 40x40 measured slower packed (r2)
    g.H = H; g.W = W; g.N = N;
    g.ph = H + 1; g.pw = W + 1;
    g.R = N < 16 ? N : 16;
    g.CW = g.R * g.pw;
    g.CH = ((N + g.R - 1) / g.R) * g.ph;
    g.inv_ph = 1.0f / (float)g.ph;
    g.inv_pw = 1.0f / (float)g.pw;
    return g;
}
// The decision depends on (N, H, W) only, so that yunet_dp_bwd_blocks() and the launch agree on the
// grid.  Packed tensors are addressed with 32-bit byte offsets over the WHOLE tensor: the launch
// rejects the (absurd for <= 20x20 maps) case of more than 2^30 elements.
static inline PackGeom dp_pack_geom(int N, int H, int W) {
    PackGeom g = make_pack(N, H, W);
    if (yunet_options().no_pack) g.on = 0;    // A/B switch (tools/kbench.py)
    return g;
}
// the packed kernels are instantiated for the units that live on the small pyramid levels
static inline bool dp_use_pack(int N, int H, int W, int cin, int cout) {
    return cin == 64 && (cout == 64 || cout == 16) && dp_pack_geom(N, H, W).on;
}
// backward: the same decision as forward for both packed variants (64->16 and 64->64; the latter went
// scratch-free with the per-tile validity byte map)
static inline bool dp_use_pack_bwd(int N, int H, int W, int cin, int cout) {
    return dp_use_pack(N, H, W, cin, cout);
}
static inline bool dp_pack_fits(const PackGeom& g, long long x_img_stride, long long z_img_stride) {
    const long long lim = 1ll << 30;           // floats
    return !g.on || ((long long)g.N * x_img_stride < lim && (long long)g.N * z_img_stride < lim);
}
__device__ __forceinline__ bool pk_locate(const PackGeom& g, int Y, int X, int& n, int& y, int& x) {
    if ((unsigned)Y >= (unsigned)g.CH || (unsigned)X >= (unsigned)g.CW) return false;
    // (Y + 0.5) / pitch is at least 0.5 / pitch away from an integer: the float quotient is exact
    const int iy = (int)(((float)Y + 0.5f) * g.inv_ph), ix = (int)(((float)X + 0.5f) * g.inv_pw);
    y = Y - iy * g.ph;
    x = X - ix * g.pw;
    n = iy * g.R + ix;
    return y < g.H && x < g.W && n < g.N;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int hip_status() { return -(int)hipGetLastError(); }

// conv_fwd64.hip: the plain fp32 64 -> 64 forward unit (no packing, no fused pooling)
int ACT_SUFFIX(launch_dp_fwd64s)(const YunetDP* d, hipStream_t stream);
int ACT_SUFFIX(launch_dp_fwd64s_group)(const YunetDP* const* ds, int n, hipStream_t stream);     // independent plain units, one grid
// conv_bwd16.hip: backward of the fp32 16 -> 16 unit (plain or pooled dy), z recomputed from x
int ACT_SUFFIX(launch_dp_bwd16s)(const YunetDP* d, hipStream_t stream);
// conv_fwd16.hip: forward of the fp32 16 -> 16 (plain | fused pooling) and 16 -> 64 units
int ACT_SUFFIX(launch_dp_fwd16s)(const YunetDP* d, hipStream_t stream);
// conv_stem.hip: the fp32 stem on the matrix cores (forward; weight gradient with z recomputed from the image)
int ACT_SUFFIX(launch_stem_fwd_mma)(const float* img, const float* w, const float* b, float* z, double* stats, int N, int H, int W,
                                    hipStream_t stream);
int launch_stem_bwd_mma(const float* img, const float* w, const float* b, const float* dy, const YunetBN* bn, float* partials,
                        int blocks, int N, int H, int W, hipStream_t stream);
