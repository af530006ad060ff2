// tr_probe.hip -- empirical semantics of ds_read_b64_tr_b16 on gfx950: every lane supplies its own 8-byte
// address (lane l reads at byte 256*l), LDS element e (u16) holds its own index e.  Output element j of lane l
// then reveals (owner lane = value / 128, element within the owner's 8 bytes = value % 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    auto p = reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(
        (__attribute__((address_space(3))) unsigned short*)lds + 128 * l);
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (owner %2d, elem %d)", h[l * 4 + j] / 128, h[l * 4 + j] % 128);
        printf("\n");
    }
    return 0;
}
