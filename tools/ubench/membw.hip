// membw.hip -- what the HBM path of this box delivers to plain streaming kernels (the practical ceiling the conv
// kernels are held against): read-only, write-only, copy, and "read 1.2 x + write 1 x" (the forward unit's mix).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/membw.hip -o tools/ubench/membw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, f4* __restrict__ out, size_t n) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
    if (acc.x == 1.2345f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ out, size_t n) {
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a[i];
}
// UNROLL independent loads in flight per thread
template <int U>
__global__ __launch_bounds__(256) void k_copyu(const f4* __restrict__ a, f4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) out[i + u * stride] = v[u];
    }
}
int main() {
    const size_t bytes = (size_t)800 << 20, n = bytes / 16;
    f4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {256, 512, 1024, 2048, 4096, 16384};
    for (int g : grids) {
        auto run = [&](const char* name, auto fn, double moved) {
            for (int i = 0; i < 3; ++i) fn();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) fn();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
            printf("grid %5d %-8s %7.3f ms %7.1f GB/s\n", g, name, ms, moved / (ms * 1e-3) / 1e9);
        };
        run("read", [&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, b, n); }, (double)bytes);
        run("write", [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); }, (double)bytes);
        run("copy", [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        run("copy x4", [&] { hipLaunchKernelGGL(k_copyu<4>, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
        run("copy x8", [&] { hipLaunchKernelGGL(k_copyu<8>, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    }
    return 0;
}
