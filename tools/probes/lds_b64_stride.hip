// Probe: LDS cycles of ds_read_b64 for the lane -> address patterns of the fused stem's layer-2 operand reads (stride-2 pixels:
// 16 bytes between the 16 lanes of a k-group; the second k-group of a 32-lane half `X` bytes further).  One wave per CU-sized
// grid would add noise: ONE workgroup of 8 waves (the LDS array, not one wave's issue rate, is the limit), 16 384 reads per wave, s_memtime around them.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_b64_stride.hip -o tools/probes/lds_b64_stride && tools/probes/lds_b64_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const int* lane_off, unsigned long long* out, int reps) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)lane_off[lane];
    unsigned long long acc = 0;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:2048\n\tds_read_b64 %2, %8 offset:4096\n\tds_read_b64 %3, %8 offset:6144\n\t"
                     "ds_read_b64 %4, %8 offset:8192\n\tds_read_b64 %5, %8 offset:10240\n\tds_read_b64 %6, %8 offset:12288\n\tds_read_b64 %7, %8 offset:14336\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr));
        acc += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x + v7.x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (acc == 0x123456789ull) out[1] = acc;
}

int main() {
    int* d_off; unsigned long long* d_out;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    auto run = [&](const char* name, auto f) {
        std::vector<int> off(64);
        for (int l = 0; l < 64; ++l) off[l] = f(l & 15, l >> 4);
        hipMemcpy(d_off, off.data(), 256, hipMemcpyHostToDevice);
        unsigned long long h[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 65536, 0, d_off, d_out, 2048);
            hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        }
        printf("%-58s %6.2f LDS cycles per ds_read_b64 (8 waves issuing)\n", name, (double)h[0] / (2048.0 * 8 * 8));
    };
    run("contiguous: lane l -> 8 l", [](int c, int g) { return (g * 16 + c) * 8; });
    run("stride 16, groups 256 B apart (0 mod 256)", [](int c, int g) { return c * 16 + g * 256; });
    run("stride 16, groups 8 B apart", [](int c, int g) { return c * 16 + (g & 1) * 8 + (g >> 1) * 256; });
    run("stride 16, g&1 -> +904 (8 mod 16), g>>1 -> +3616", [](int c, int g) { return c * 16 + (g & 1) * 904 + (g >> 1) * 3616; });
    run("stride 16, g&1 -> +1792 (old layout), g>>1 -> +3584", [](int c, int g) { return c * 16 + (g & 1) * 1792 + (g >> 1) * 3584; });
    run("stride 16, g&1 -> +136, g>>1 -> +8 (taps dx, dx+1)", [](int c, int g) { return c * 16 + (g & 1) * 136 + (g >> 1) * 8; });
    run("stride 16, g&1 -> +904, g>>1 -> +8 (same row, dx+1)", [](int c, int g) { return c * 16 + (g & 1) * 904 + (g >> 1) * 8; });
    run("stride 16, g&1 -> +904, g>>1 -> +16", [](int c, int g) { return c * 16 + (g & 1) * 904 + (g >> 1) * 16; });
    run("stride 8 (layer-1 pattern), g&1 -> +1920, g>>1 -> +8", [](int c, int g) { return c * 8 + (g & 1) * 1920 + (g >> 1) * 8; });
    run("stride 16 within 16 lanes only, all groups same", [](int c, int g) { return c * 16; });
    run("stride 32", [](int c, int g) { return c * 32 + g * 8; });
    return 0;
}
