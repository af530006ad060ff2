// bwd_ab.cpp -- torch-free A/B harness for the ConvDPUnit backward kernel (starts in milliseconds, so a
// GPU session spends its time on the kernel instead of on `import torch`).
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/bwd_ab.cpp -o tools/ubench/bwd_ab.bin -ldl
//   tools/ubench/bwd_ab.bin <libA.so> [<libB.so> ...]      (env per lib: see `variants` below)
//
// For every shape the harness runs yunet_dp_bwd of the FIRST variant as the yardstick, then each other
// variant on the same inputs, and prints (a) max |difference| / max |yardstick| for dx, the weight
// gradient sections and the producer's BN-backward sums, (b) the mean launch time over `reps` launches.
// A variant is "<lib path>[:ENV=VALUE]" -- the library is dlopen'ed privately and ENV is set while its
// launches run (e.g. YUNET_BWD_FP32MMA=1 selects the exact-fp32 matrix instruction inside one library).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/yunet_hip.h"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

typedef int (*dp_fn)(const YunetDP*, void*);
typedef int (*blocks_fn)(int, int, int, int, int);

struct Variant {
    std::string name, env_k, env_v;
    void* h;
    dp_fn bwd, fwd;
    blocks_fn blocks;
};

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand() {   // xorshift, uniform [-1, 1)
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) / 8388608.0 - 1.0);
}
static float grand() {   // zero mean, unit variance (sum of 2 uniforms)
    return (frand() + frand()) * 1.2247f;
}

template <typename T>
static T* dev(const std::vector<T>& v) {
    T* p;
    CK(hipMalloc(&p, v.size() * sizeof(T) + 256));
    CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

static double rel(const std::vector<double>& a, const std::vector<double>& b, size_t lo, size_t hi) {
    double md = 0, mr = 0;
    for (size_t i = lo; i < hi; ++i) {
        if (a[i] != a[i]) return INFINITY;           // a NaN (e.g. an element the kernel never wrote) must not vanish in fmax
        md = fmax(md, fabs(a[i] - b[i]));
        mr = fmax(mr, fabs(b[i]));
    }
    return md / (mr + 1e-300);
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <lib.so[:ENV=VAL]> ...\n", argv[0]);
        return 1;
    }
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
    const int N = getenv("BATCH") ? atoi(getenv("BATCH")) : 256;
    std::vector<Variant> vs;
    for (int i = 1; i < argc; ++i) {
        Variant v;
        std::string a = argv[i];
        size_t c = a.find(':');
        std::string lib = a.substr(0, c);
        if (c != std::string::npos) {
            std::string e = a.substr(c + 1);
            size_t q = e.find('=');
            v.env_k = e.substr(0, q);
            v.env_v = q == std::string::npos ? "1" : e.substr(q + 1);
        }
        v.name = a;
        v.h = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!v.h) {
            fprintf(stderr, "dlopen %s: %s\n", lib.c_str(), dlerror());
            return 2;
        }
        v.bwd = (dp_fn)dlsym(v.h, "yunet_dp_bwd");
        v.blocks = (blocks_fn)dlsym(v.h, "yunet_dp_bwd_blocks");
        v.fwd = (dp_fn)dlsym(v.h, "yunet_dp_fwd");
        if (!v.bwd || !v.blocks || !v.fwd) return 2;
        vs.push_back(v);
    }
    struct Shape { int ci, co, h, w; };
    std::vector<Shape> shapes = {{64, 64, 80, 80}, {64, 64, 40, 40}, {64, 64, 20, 20}, {64, 64, 10, 10}};
    if (getenv("SHAPES_ALL")) {
        shapes.push_back({16, 16, 160, 160});
        shapes.push_back({16, 64, 80, 80});
        shapes.push_back({64, 16, 40, 40});
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const int ci = sh.ci, co = sh.co, H = sh.h, W = sh.w;
        if (getenv("ONLY") && atoi(getenv("ONLY")) != H) continue;
        const size_t px = (size_t)N * H * W;
        std::vector<float> x(px * ci), z(px * co), dy(px * co), wp(co * ci), bp(co), wd(co * 9), bd(co);
        for (auto& v : x) v = grand() * 2.0f + 0.5f;
        for (auto& v : z) v = grand() * 1.5f + 0.2f;
        for (auto& v : dy) v = (frand() > 0.0f) ? grand() : 0.0f;   // ReLU-masked upstream gradient
        for (auto& v : wp) v = grand() * 0.12f;
        for (auto& v : bp) v = grand() * 0.1f;
        for (auto& v : wd) v = grand() * 0.3f;
        for (auto& v : bd) v = grand() * 0.1f;
        std::vector<float> gi(ci), bi(ci), go(co), bo(co);
        for (auto& v : gi) v = 1.0f + 0.4f * frand();
        for (auto& v : bi) v = 0.2f * frand();
        for (auto& v : go) v = 1.0f + 0.4f * frand();
        for (auto& v : bo) v = 0.2f * frand();
        // forward sums of both BNs (fp64, exact) and the output BN's backward sums
        std::vector<double> sin(2 * ci, 0.0), sout(2 * co, 0.0), bso(2 * co, 0.0);
        for (size_t p = 0; p < px; ++p)
            for (int c = 0; c < ci; ++c) {
                const double v = x[p * ci + c];
                sin[c] += v; sin[ci + c] += v * v;
            }
        for (size_t p = 0; p < px; ++p)
            for (int c = 0; c < co; ++c) {
                const double v = z[p * co + c];
                sout[c] += v; sout[co + c] += v * v;
            }
        for (int c = 0; c < co; ++c) {
            const double mean = sout[c] / px, var = sout[co + c] / px - mean * mean, inv = 1.0 / sqrt(var + 1e-5);
            double s0 = 0, s1 = 0;
            for (size_t p = 0; p < px; ++p) {
                const double g = dy[p * co + c];
                s0 += g; s1 += g * (z[p * co + c] - mean) * inv;
            }
            bso[c] = s0; bso[co + c] = s1;
        }
        float *dx_x = dev(x), *dz = dev(z), *ddy = dev(dy), *dwp = dev(wp), *dbp = dev(bp), *dwd = dev(wd),
              *dbd = dev(bd), *dgi = dev(gi), *dbi = dev(bi), *dgo = dev(go), *dbo = dev(bo);
        // SLOTS=<n>: every BN sum block as n replicas (YunetBN::slots); inputs carry their sums in replica 0
        const int slots = getenv("SLOTS") ? atoi(getenv("SLOTS")) : 1;
        auto padded = [&](std::vector<double> v) { v.resize(v.size() * slots, 0.0); return v; };
        auto fold = [&](std::vector<double>& v, size_t n) {
            for (int k = 1; k < slots; ++k) for (size_t i = 0; i < n; ++i) v[i] += v[k * n + i];
            v.resize(n);
        };
        double *dsin = dev(padded(sin)), *dsout = dev(padded(sout)), *dbso = dev(padded(bso));
        float* ddx;
        CK(hipMalloc(&ddx, px * ci * 4));
        double* dbsi;
        CK(hipMalloc(&dbsi, 2 * ci * 8 * slots));
        const int width = co * ci + co + co * 9 + co;
        std::vector<double> ref_dx, ref_w, ref_b;
        if (getenv("ZFWD") && !getenv("FWD")) {
            // ZFWD=1: z = the unit's FORWARD output for x and the weights (first variant's forward kernel) instead of
            // independent noise -- what backward kernels that recompute z from x need; the output BN's forward and
            // backward sums are re-derived from it
            double* dtmp;
            CK(hipMalloc(&dtmp, 2 * co * 8 * slots));
            CK(hipMemset(dtmp, 0, 2 * co * 8 * slots));
            YunetDP f;
            memset(&f, 0, sizeof(f));
            f.N = N; f.H = H; f.W = W; f.cin = ci; f.cout = co;
            f.in_transform = YUNET_T_BNRELU; f.out_has_bn = 1;
            f.x_img_stride = (int64_t)H * W * ci; f.z_img_stride = (int64_t)H * W * co;
            f.x = dx_x;
            f.in_bn = YunetBN{dsin, nullptr, dgi, dbi, (int32_t)px, 1e-5f, slots};
            f.w_pw = dwp; f.b_pw = dbp; f.w_dw = dwd; f.b_dw = dbd;
            f.z = dz;
            f.out_bn = YunetBN{dtmp, nullptr, dgo, dbo, (int32_t)px, 1e-5f, slots};
            // (inside the first variant's environment: a library reads its switches on first use)
            if (!vs[0].env_k.empty()) setenv(vs[0].env_k.c_str(), vs[0].env_v.c_str(), 1);
            const int rcf = vs[0].fwd(&f, st);
            CK(hipStreamSynchronize(st));
            if (!vs[0].env_k.empty()) unsetenv(vs[0].env_k.c_str());
            if (rcf != 0) { printf("ZFWD: forward rc=%d\n", rcf); exit(2); }
            CK(hipMemcpy(z.data(), dz, z.size() * 4, hipMemcpyDeviceToHost));
            std::fill(sout.begin(), sout.end(), 0.0);
            for (size_t p = 0; p < px; ++p)
                for (int c = 0; c < co; ++c) {
                    const double v = z[p * co + c];
                    sout[c] += v; sout[co + c] += v * v;
                }
            for (int c = 0; c < co; ++c) {
                const double mean = sout[c] / px, var = sout[co + c] / px - mean * mean, inv = 1.0 / sqrt(var + 1e-5);
                double s0 = 0, s1 = 0;
                for (size_t p = 0; p < px; ++p) {
                    const double g = dy[p * co + c];
                    s0 += g; s1 += g * (z[p * co + c] - mean) * inv;
                }
                bso[c] = s0; bso[co + c] = s1;
            }
            const std::vector<double> ps = padded(sout), pb = padded(bso);
            CK(hipMemcpy(dsout, ps.data(), ps.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dbso, pb.data(), pb.size() * 8, hipMemcpyHostToDevice));
            hipFree(dtmp);
        }
        if (getenv("FWD")) {
            // FWD=1: the FORWARD kernel instead (z and the output BN sums against the first variant)
            float* dzo;
            CK(hipMalloc(&dzo, px * co * 4));
            double* dst;
            CK(hipMalloc(&dst, 2 * co * 8 * slots));
            std::vector<double> rz, rs;
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                Variant& v = vs[vi];
                if (!v.env_k.empty()) setenv(v.env_k.c_str(), v.env_v.c_str(), 1);
                YunetDP d;
                memset(&d, 0, sizeof(d));
                d.N = N; d.H = H; d.W = W; d.cin = ci; d.cout = co;
                d.in_transform = YUNET_T_BNRELU; d.out_has_bn = 1;
                d.x_img_stride = (int64_t)H * W * ci; d.z_img_stride = (int64_t)H * W * co;
                d.x = dx_x;
                d.in_bn = YunetBN{dsin, nullptr, dgi, dbi, (int32_t)px, 1e-5f, slots};
                d.w_pw = dwp; d.b_pw = dbp; d.w_dw = dwd; d.b_dw = dbd;
                d.z = dzo;
                d.out_bn = YunetBN{dst, nullptr, dgo, dbo, (int32_t)px, 1e-5f, slots};
                // forward: ABL=<mask> (1 skip pointwise GEMM, 2 skip depthwise, 4 skip z stores, 8 skip next-tile loads),
                // PROF=1 per-phase clocks (stage | pw | dw | barrier) summed over a workgroup's tiles, per wave
                unsigned long long* fprof = nullptr;
                if (getenv("ABL")) d.prof = (unsigned long long*)(uintptr_t)atoll(getenv("ABL"));
                if (getenv("PROF")) {
                    CK(hipMalloc(&fprof, (size_t)2048 * 8 * 4 * 8));
                    CK(hipMemset(fprof, 0, (size_t)2048 * 8 * 4 * 8));
                    d.prof = fprof;
                }
                CK(hipMemsetAsync(dst, 0, 2 * co * 8 * slots, st));
                int rc = v.fwd(&d, st);
                CK(hipStreamSynchronize(st));
                if (rc != 0) { printf("%-40s fwd rc=%d\n", v.name.c_str(), rc); continue; }
                std::vector<float> hz(px * co);
                std::vector<double> hs(2 * co * slots);
                CK(hipMemcpy(hz.data(), dzo, hz.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
                fold(hs, 2 * co);
                std::vector<double> vz(hz.begin(), hz.end());
                for (int i = 0; i < 3; ++i) v.fwd(&d, st);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) v.fwd(&d, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                const double gbs = (double)px * (ci + co) * 4 / (ms * 1e-3) / 1e9;
                if (vi == 0) {
                    rz = vz; rs = hs;
                    printf("%dx%d %d->%d N=%d  fwd %-40s %8.4f ms %7.1f GB/s  (yardstick)\n", H, W, ci, co, N, v.name.c_str(), ms, gbs);
                } else {
                    printf("%dx%d %d->%d N=%d  fwd %-40s %8.4f ms %7.1f GB/s  err z %.2e stats %.2e\n", H, W, ci, co, N,
                           v.name.c_str(), ms, gbs, rel(vz, rz, 0, vz.size()), rel(hs, rs, 0, hs.size()));
                }
                if (fprof) {
                    CK(hipMemset(fprof, 0, (size_t)2048 * 8 * 4 * 8));
                    v.fwd(&d, st);
                    CK(hipStreamSynchronize(st));
                    std::vector<unsigned long long> hp2((size_t)2048 * 8 * 4);
                    CK(hipMemcpy(hp2.data(), fprof, hp2.size() * 8, hipMemcpyDeviceToHost));
                    double acc[4] = {0, 0, 0, 0};
                    int rows = 0;
                    for (size_t r = 0; r < hp2.size() / 4; ++r) {
                        if (!hp2[r * 4] && !hp2[r * 4 + 1]) continue;
                        ++rows;
                        for (int k = 0; k < 4; ++k) acc[k] += (double)hp2[r * 4 + k];
                    }
                    if (rows)
                        printf("    fwd clocks per wave (mean over %d waves): stage %.0f | pw %.0f | dw %.0f | barrier %.0f | total %.0f\n",
                               rows, acc[0] / rows, acc[1] / rows, acc[2] / rows, acc[3] / rows,
                               (acc[0] + acc[1] + acc[2] + acc[3]) / rows);
                    CK(hipFree(fprof));
                }
                if (!v.env_k.empty()) unsetenv(v.env_k.c_str());
            }
            fflush(stdout);
            hipFree(dzo); hipFree(dst);
        } else
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            Variant& v = vs[vi];
            if (!v.env_k.empty()) setenv(v.env_k.c_str(), v.env_v.c_str(), 1);
            const int blocks = v.blocks(N, H, W, ci, co);
            float* dpart;
            CK(hipMalloc(&dpart, (size_t)blocks * width * 4));
            YunetDP d;
            memset(&d, 0, sizeof(d));
            d.N = N; d.H = H; d.W = W; d.cin = ci; d.cout = co;
            d.in_transform = YUNET_T_BNRELU; d.out_has_bn = getenv("NOBN") ? 0 : 1; d.accumulate_dx = 0;      // NOBN: no z read
            d.x_img_stride = (int64_t)H * W * ci; d.z_img_stride = (int64_t)H * W * co;
            d.x = dx_x;
            d.in_bn = YunetBN{dsin, dbsi, dgi, dbi, (int32_t)px, 1e-5f, slots};
            d.w_pw = dwp; d.b_pw = dbp; d.w_dw = dwd; d.b_dw = dbd;
            d.z = dz;
            d.out_bn = YunetBN{dsout, dbso, dgo, dbo, (int32_t)px, 1e-5f, slots};
            d.dy = ddy; d.dy_scale = nullptr; d.dx = ddx;
            d.wgrad_partials = dpart; d.wgrad_blocks = blocks; d.prof = nullptr;
            // ABL=<mask>: the kernel's debug ablation mask (1 p GEMM, 2 depthwise, 4 dW1, 8 da, 16 dx
            // store, 32 next-tile prefetch are SKIPPED); results are then wrong, only the time matters
            if (getenv("ABL")) d.prof = (unsigned long long*)(uintptr_t)atoll(getenv("ABL"));
            unsigned long long* dprof = nullptr;
            if (getenv("PROF")) {     // libraries built with -DDP_BWD_PROF: per-phase cycle counters
                CK(hipMalloc(&dprof, (size_t)blocks * 64));
                CK(hipMemset(dprof, 0, (size_t)blocks * 64));
                d.prof = dprof;
                if (getenv("ABL"))     // dp_bwd64 profile builds: ablation mask in the low bits of the aligned pointer
                    d.prof = (unsigned long long*)((uintptr_t)dprof | ((uintptr_t)atoll(getenv("ABL")) & 63));
            }
            CK(hipMemsetAsync(dbsi, 0, 2 * ci * 8 * slots, st));
            CK(hipMemsetAsync(ddx, 0xff, px * ci * 4, st));       // NaN: a variant that skips a dx element must not inherit the previous one's
            int rc = v.bwd(&d, st);
            CK(hipStreamSynchronize(st));
            if (rc != 0) {
                printf("%-40s rc=%d\n", v.name.c_str(), rc);
                continue;
            }
            // results of this single launch
            std::vector<float> hdx(px * ci), hp((size_t)blocks * width);
            std::vector<double> hb(2 * ci * slots);
            CK(hipMemcpy(hdx.data(), ddx, hdx.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hp.data(), dpart, hp.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), dbsi, hb.size() * 8, hipMemcpyDeviceToHost));
            fold(hb, 2 * ci);
            std::vector<double> vdx(hdx.begin(), hdx.end()), vw(width, 0.0);
            for (int b = 0; b < blocks; ++b)
                for (int j = 0; j < width; ++j) vw[j] += hp[(size_t)b * width + j];
            // timing
            for (int i = 0; i < 3; ++i) v.bwd(&d, st);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) v.bwd(&d, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            const double gbs = (double)px * (2 * ci + co) * 4 / (ms * 1e-3) / 1e9;
            {   // db2 = sum(dz) is zero in exact arithmetic (BN backward): its magnitude against max |dW2| says how
                // well a variant keeps that cancellation
                const size_t o2 = (size_t)co * ci + co, o3 = o2 + (size_t)co * 9;
                double m2 = 0, mw = 0, m1 = 0;
                for (size_t i = o3; i < (size_t)width; ++i) m2 = fmax(m2, fabs(vw[i]));
                for (size_t i = o2; i < o3; ++i) mw = fmax(mw, fabs(vw[i]));
                for (size_t i = (size_t)co * ci; i < o2; ++i) m1 = fmax(m1, fabs(vw[i]));
                printf("    max|db2| %.3e  max|db1| %.3e  (max|dW2| %.3e)\n", m2, m1, mw);
            }
            if (vi == 0) {
                ref_dx = vdx; ref_w = vw; ref_b = hb;
                printf("%dx%d %d->%d N=%d  %-44s %8.4f ms %7.1f GB/s  (yardstick, %d rows)\n", H, W, ci, co, N,
                       v.name.c_str(), ms, gbs, blocks);
            } else {
                const size_t o1 = (size_t)co * ci, o2 = o1 + co, o3 = o2 + (size_t)co * 9;
                printf("%dx%d %d->%d N=%d  %-44s %8.4f ms %7.1f GB/s  err dx %.2e dW1 %.2e db1 %.2e dW2 %.2e db2 %.2e bn %.2e\n",
                       H, W, ci, co, N, v.name.c_str(), ms, gbs, rel(vdx, ref_dx, 0, vdx.size()),
                       rel(vw, ref_w, 0, o1), rel(vw, ref_w, o1, o2), rel(vw, ref_w, o2, o3),
                       rel(vw, ref_w, o3, (size_t)width), rel(hb, ref_b, 0, hb.size()));
            }
            if (dprof) {
                CK(hipMemset(dprof, 0, (size_t)blocks * 64));
                // d.prof may carry the ablation bits
                v.bwd(&d, st);
                CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> hp2((size_t)blocks * 8);
                CK(hipMemcpy(hp2.data(), dprof, hp2.size() * 8, hipMemcpyDeviceToHost));
                double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
                for (int b = 0; b < blocks; ++b)
                    for (int k = 0; k < 8; ++k) acc[k] += (double)hp2[(size_t)b * 8 + k] / blocks;
                for (int k = 0; k < 8; ++k) tot += acc[k];
                printf("    cycles per workgroup (mean): prologue %.0f | stage %.0f | p %.0f | dw %.0f | dW1+da %.0f | mask %.0f | store %.0f"
                       " | epilogue %.0f | total %.0f\n", acc[6], acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[7], tot);
                CK(hipFree(dprof));
            }
            if (!v.env_k.empty()) unsetenv(v.env_k.c_str());
            CK(hipFree(dpart));
        }
        fflush(stdout);
        hipFree(dx_x); hipFree(dz); hipFree(ddy); hipFree(dwp); hipFree(dbp); hipFree(dwd); hipFree(dbd);
        hipFree(dgi); hipFree(dbi); hipFree(dgo); hipFree(dbo); hipFree(dsin); hipFree(dsout); hipFree(dbso);
        hipFree(ddx); hipFree(dbsi);
    }
    return 0;
}
