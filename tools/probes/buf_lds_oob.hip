// Probe: does an out-of-range `buffer_load ... lds` (LDS-DMA through a buffer descriptor) write ZEROS to LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, int nbytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s = (float*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = -7.0f;   // poison
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 out of range
    unsigned voff = threadIdx.x < 32 ? threadIdx.x * 16 : 0x7ffffff0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    float *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, 4096, o);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    printf("in-range lane0: %g %g %g %g | lane31: %g\n", r[0], r[1], r[2], r[3], r[31 * 4]);
    printf("out-of-range lane32: %g %g %g %g | lane63: %g %g\n", r[128], r[129], r[130], r[131], r[252], r[255]);
    printf("untouched: %g\n", r[300]);
    return 0;
}
