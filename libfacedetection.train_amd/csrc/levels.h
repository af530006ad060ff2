// levels.h -- prior geometry shared by the loss step and the detection post-processing
// (MlvlPointGenerator, offset 0: mmdet/core/anchor/point_generator.py:80-175).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yunet_hip.h"

namespace {

struct Levels {
    int n;
    int h[YUNET_MAX_LEVELS], w[YUNET_MAX_LEVELS], s[YUNET_MAX_LEVELS], base[YUNET_MAX_LEVELS + 1];
};

static __host__ inline Levels make_levels(const YunetLevels* lv) {
    Levels L;
    L.n = lv->num_levels;
    int b = 0;
    for (int i = 0; i < YUNET_MAX_LEVELS; ++i) {
        L.h[i] = i < L.n ? lv->h[i] : 0;
        L.w[i] = i < L.n ? lv->w[i] : 0;
        L.s[i] = i < L.n ? lv->stride[i] : 1;
        L.base[i] = b;
        b += L.h[i] * L.w[i];
    }
    L.base[YUNET_MAX_LEVELS] = b;
    return L;
}

// prior p -> (x*s, y*s, s): MlvlPointGenerator, offset 0, y-major, levels concatenated.
__device__ __forceinline__ void prior_of(const Levels& L, int p, float& px, float& py, float& s) {
    int w = L.w[0], st = L.s[0], base = 0;
#pragma unroll
    for (int i = 1; i < YUNET_MAX_LEVELS; ++i) {
        const bool in = i < L.n && p >= L.base[i];
        w = in ? L.w[i] : w;
        st = in ? L.s[i] : st;
        base = in ? L.base[i] : base;
    }
    const int q = p - base;
    const int iy = q / w, ix = q - iy * w;
    s = (float)st;
    px = (float)ix * s;
    py = (float)iy * s;
}


}  // namespace
