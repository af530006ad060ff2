// micro-benchmark: issue rate of the exact-fp32 MFMAs on gfx950, per SIMD, as seen by one wave.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WITH_LDS>
__global__ void k16(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (WITH_LDS) { a = lds[(threadIdx.x * 4 + it * 64) & 4095]; b = lds[(threadIdx.x + it * 17) & 4095]; }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F> void run(const char* name, F launch, int nmfma_per_iter, int iters, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); launch(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    double per = (double)h[0] / ((double)iters * nmfma_per_iter);
    printf("%-44s %8.1f clk/MFMA (wave view)  kernel %.3f ms  => %.2f GHz shader clock\n", name, per, ms,
           (double)h[0] / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main() {
    const int it = 20000;
    run("16x16x4 f32, 1 wave/SIMD, 4 acc", [&](float* o, unsigned long long* c) { hipLaunchKernelGGL((k16<4, 0>), dim3(256), dim3(256), 0, 0, o, c, it); }, 4, it, 256);
    run("16x16x4 f32, 1 wave/SIMD, 1 acc (dependent)", [&](float* o, unsigned long long* c) { hipLaunchKernelGGL((k16<1, 0>), dim3(256), dim3(256), 0, 0, o, c, it); }, 1, it, 256);
    run("16x16x4 f32, 2 waves/SIMD, 4 acc", [&](float* o, unsigned long long* c) { hipLaunchKernelGGL((k16<4, 0>), dim3(256), dim3(512), 0, 0, o, c, it); }, 4, it, 256);
    run("16x16x4 f32, 1 wave/SIMD, 4 acc, LDS operands", [&](float* o, unsigned long long* c) { hipLaunchKernelGGL((k16<4, 1>), dim3(256), dim3(256), 0, 0, o, c, it); }, 4, it, 256);
    run("32x32x2 f32, 1 wave/SIMD, 2 acc", [&](float* o, unsigned long long* c) { hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, o, c, it); }, 2, it, 256);
    return 0;
}
