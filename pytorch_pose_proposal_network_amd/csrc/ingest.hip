// Frame ingest on the device (/root/reference/rt_test.py:150-157 grab_frame): camera frame BGR u8
// [B,Hs,Ws,3] -> cv2.resize(INTER_LINEAR) -> flip vertically and horizontally -> RGB u8 [B,Hd,Wd,3], written
// straight into the conv plan's input buffer (the stem kernel normalises on load).  One thread per output pixel;
// the interpolation coefficients are recomputed per thread with OpenCV's own float/double operations
// (-ffp-contract=off: no fused multiply-add may change a rounding), so there are no tables to upload.
// HBM-bound: reads <= 4 source pixels per output pixel (L2 serves the overlap), writes 3 bytes.
// Arithmetic: oracle/ingest_ref.py states the rule (OpenCV's 8-bit fixed-point bilinear; PARITY UNPINNED, cv2 is not
// available in this image).
#include "common.h"

namespace {

struct Axis {
    int s;          // first source index
    int a0, a1;     // 2^11 fixed-point weights of s and s + 1
};

__device__ __forceinline__ Axis axis_coeff(int d, int n_src, double scale) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
    Axis a;
    a.s = s;
    // saturate_cast<short>(cvRound(v * 2048)): round half to even, values are within [0, 2048]
    a.a0 = (int)rintf((1.f - f) * 2048.f);
    a.a1 = (int)rintf(f * 2048.f);
    return a;
}

__global__ void __launch_bounds__(256)
ingest_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int B, int Hs, int Ws, int Hd,
              int Wd, double scale_x, double scale_y, int flip, int swap_rb, int area2) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    const int b = blockIdx.z;
    if (ox >= Wd) return;
    // output (oy, ox) shows resized pixel (ry, rx): both flips of rt_test.py:154-155 = a rotation by 180 degrees
    const int ry = flip ? Hd - 1 - oy : oy, rx = flip ? Wd - 1 - ox : ox;
    const unsigned char* img = src + (size_t)b * Hs * Ws * 3;
    int v[3];
    if (area2) {
        const unsigned char* p0 = img + ((size_t)(2 * ry) * Ws + 2 * rx) * 3;
        const unsigned char* p1 = p0 + (size_t)Ws * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2;
    } else {
        const Axis ax = axis_coeff(rx, Ws, scale_x), ay = axis_coeff(ry, Hs, scale_y);
        const int x1 = min(ax.s + 1, Ws - 1), y1 = min(ay.s + 1, Hs - 1);
        const unsigned char* r0 = img + (size_t)ay.s * Ws * 3;
        const unsigned char* r1 = img + (size_t)y1 * Ws * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int t0 = r0[ax.s * 3 + c] * ax.a0 + r0[x1 * 3 + c] * ax.a1;
            const int t1 = r1[ax.s * 3 + c] * ax.a0 + r1[x1 * 3 + c] * ax.a1;
            v[c] = (((ay.a0 * (t0 >> 4)) >> 16) + ((ay.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
        }
    }
    unsigned char* o = dst + (((size_t)b * Hd + oy) * Wd + ox) * 3;
    o[0] = (unsigned char)min(max(v[swap_rb ? 2 : 0], 0), 255);
    o[1] = (unsigned char)min(max(v[1], 0), 255);
    o[2] = (unsigned char)min(max(v[swap_rb ? 0 : 2], 0), 255);
}

}  // namespace

extern "C" int ppn_ingest_frames(const void* src_bgr, int32_t batch, int32_t src_h, int32_t src_w, void* dst_rgb,
                                 int32_t dst_h, int32_t dst_w, int32_t flip, int32_t swap_rb, void* stream) {
    if (!src_bgr || !dst_rgb) return ppn::fail(PPN_E_INVALID, "ppn_ingest_frames: NULL buffer");
    if (batch < 1 || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || batch > 65535 || dst_h > 65535)
        return ppn::fail(PPN_E_INVALID, "ppn_ingest_frames: bad geometry %d x %dx%d -> %dx%d", batch, src_h, src_w, dst_h,
                         dst_w);
    const int area2 = (src_h == 2 * dst_h && src_w == 2 * dst_w) ? 1 : 0;
    const dim3 grid((dst_w + 255) / 256, dst_h, batch);
    hipLaunchKernelGGL(ingest_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(src_bgr), static_cast<unsigned char*>(dst_rgb), batch, src_h,
                       src_w, dst_h, dst_w, (double)src_w / (double)dst_w, (double)src_h / (double)dst_h, flip ? 1 : 0,
                       swap_rb ? 1 : 0, area2);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
