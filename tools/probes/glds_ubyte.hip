// Probe: where does global_load_lds_ubyte put lane L's byte?  (M0 base + L, or M0 base + 4 L?)  Also ushort / dword.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SZ>
__global__ void probe(const unsigned char* src, unsigned char* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xEE;
    __syncthreads();
    const void __attribute__((address_space(1)))* g = (const void __attribute__((address_space(1)))*)(src + threadIdx.x * SZ);
    void __attribute__((address_space(3)))* l = (void __attribute__((address_space(3)))*)(lds + 16);
    if constexpr (SZ == 1) __builtin_amdgcn_global_load_lds(g, l, 1, 0, 0);
    if constexpr (SZ == 2) __builtin_amdgcn_global_load_lds(g, l, 2, 0, 0);
    if constexpr (SZ == 4) __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    unsigned char h[1024], *d, *o;
    for (int i = 0; i < 1024; ++i) h[i] = (unsigned char)(i & 0x7f);
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    for (int sz : {1, 2, 4}) {
        if (sz == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d, o);
        if (sz == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d, o);
        if (sz == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, d, o);
        unsigned char r[1024];
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("size %d:", sz);
        for (int i = 0; i < 96; ++i) printf(" %02x", r[i]);
        printf(" ... [272..288):");
        for (int i = 272; i < 288; ++i) printf(" %02x", r[i]);
        printf("\n");
    }
    return 0;
}
