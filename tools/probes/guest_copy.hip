// Probe guest: an HBM-bound copy kernel whose LDS footprint, workgroup size and register budget are parameters, to find
// out what can be co-resident on a CU with a workgroup of the dominant conv kernel (tools/probes/shadow_guest.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/guest_copy.hip -o tools/probes/libguest.so -L<torch/lib> -Wl,-rpath,<torch/lib>
#include <hip/hip_runtime.h>
#include <cstdint>

template <int REGS>
__global__ void __launch_bounds__(512) guest_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n,
                                                          int use_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* l = reinterpret_cast<uint4*>(smem);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // REGS/4 independent 16-byte loads in flight per thread: REGS VGPRs of payload
    constexpr int U = REGS / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = i + u * stride < n ? src[i + u * stride] : uint4{0, 0, 0, 0};
        if (use_lds) {                       // touch the LDS so the allocation is not optimised away
            l[threadIdx.x] = v[0];
            __syncthreads();
            v[0] = l[threadIdx.x ^ 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n) dst[i + u * stride] = v[u];
    }
}

extern "C" int guest_copy(const void* src, void* dst, size_t n16, int lds_bytes, int threads, int regs, int blocks,
                          void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto k = regs >= 64 ? guest_copy_kernel<64> : (regs >= 32 ? guest_copy_kernel<32> : guest_copy_kernel<8>);
    if (lds_bytes > 48 * 1024)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return 1;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), (size_t)lds_bytes, st, static_cast<const uint4*>(src),
                       static_cast<uint4*>(dst), n16, lds_bytes > 0 ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
