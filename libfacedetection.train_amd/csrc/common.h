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
__device__ __forceinline__ BNCoef bn_coef(const YunetBN& bn, int C, int c) {
    const double inv = 1.0 / (double)bn.count;
    const double mean = bn.stats[c] * inv;
    double var = bn.stats[C + c] * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    BNCoef k;
    k.mean = (float)mean;
    k.mean_lo = (float)(mean - (double)k.mean);
    k.invstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    k.scale = bn.gamma[c] * k.invstd;
    k.beta = bn.beta[c];
    return k;
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
    const double c1 = bn.bstats[c] * inv;
    k.c1 = (float)c1;
    k.c1_lo = (float)(c1 - (double)k.c1);
    k.c2 = (float)(bn.bstats[C + c] * inv);
    k.mean_lo = f.mean_lo;
    return k;
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
    g.on = (H <= 20 && W <= 20 && N >= 4) ? 1 : 0;
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
    if (getenv("YUNET_NO_PACK")) g.on = 0;    // A/B switch (tools/kbench.py)
    return g;
}
// the packed kernels are instantiated for the units that live on the small pyramid levels
static inline bool dp_use_pack(int N, int H, int W, int cin, int cout) {
    return cin == 64 && (cout == 64 || cout == 16) && dp_pack_geom(N, H, W).on;
}
// backward: measured per-launch on MI355X the packed 64->16 unit wins (25.9 -> 21.5 us at 10x10),
// the packed 64->64 one does not yet (its canvas mapping pushes the kernel into scratch), so the
// latter keeps the per-image tiling
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
