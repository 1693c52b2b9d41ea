// First layer of DRN-D: 7x7 conv 3->16 + BN + ReLU (drn.py:123-128), with the reference's input
// normalisation (rt_test.py:97-101 / aug.py:149-153: (u8 - mean_c)/std_c, constants in 0-1 units applied to
// 0-255 pixels, no /255) fused into the patch load.
//
// Cin = 3 does not fit the 16-byte-chunk gather of the generic kernel, so this layer has its own kernel:
// a 16x64 pixel tile per 256-thread workgroup, the (16+6)x(64+8) input patch staged in LDS as [y][x][4]
// (channel 3 = 0), weights held in registers as MFMA A-fragments for the whole kernel.
//   bf16 mode: one v_mfma_f32_16x16x32_bf16 per kernel row dy: K = 8 dx-slots x 4 channel-slots
//              (dx=7 and c=3 carry zero weights), 7 MFMAs per 16 output pixels.
//   f32 mode:  v_mfma_f32_16x16x4_f32 per (dy,dx): K = 4 channel-slots, 49 MFMAs per 16 pixels (exact f32).
// Output NHWC [B,H,W,16]; each lane owns 4 consecutive channels of one pixel so a wave store is one
// contiguous 512 B (bf16) / 1 KiB (f32) run.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int TH = 16, TW = 64, PH = TH + 6, PW = TW + 8, CO = 16;

struct StemArgs {
    const void* src;
    const float* weight;   // [16][3][7][7]
    const float* scale;    // [16]
    const float* shift;    // [16]
    void* out;
    int B, H, W, src_is_u8;
    float mean[3], inv_unused[1], stdv[3];
    int tiles_x, tiles_y;
};

template <typename T>
__device__ __forceinline__ void patch_store(T* p, float a, float b, float c);
template <>
__device__ __forceinline__ void patch_store<float>(float* p, float a, float b, float c) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, 0.f);
}
template <>
__device__ __forceinline__ void patch_store<__bf16>(__bf16* p, float a, float b, float c) {
    bf16x4 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)0.f;
    *reinterpret_cast<bf16x4*>(p) = v;
}

template <typename T>
__global__ void __launch_bounds__(256) stem7x7_kernel(StemArgs a) {
    __shared__ __attribute__((aligned(16))) T patch[PH * PW * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- weights as A fragments (rows = output channels) ---------------------------------------
    const int ch = lane & 15, g = lane >> 4;
    const float* wc = a.weight + (size_t)ch * 3 * 49;
    constexpr bool BF = sizeof(T) == 2;
    bf16x8 wa[7];        // bf16 mode: per dy, k = (dx = 2g + (i>>2), c = i&3)
    float wf[49];        // f32 mode: per (dy,dx), k = c = g
    if (BF) {
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int dx = 2 * g + (i >> 2), c = i & 3;
                const float w = (dx < 7 && c < 3) ? wc[(c * 7 + dy) * 7 + dx] : 0.f;
                wa[dy][i] = (__bf16)w;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 49; ++t) wf[t] = (g < 3) ? wc[g * 49 + t] : 0.f;
    }
    float sc[4], sh[4];
    const bool raw = a.scale == nullptr;          // train mode: plain convolution output, BN runs on batch statistics
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = raw ? 1.f : a.scale[4 * g + r]; sh[r] = raw ? 0.f : a.shift[4 * g + r]; }
    // ---- persistent loop over tiles: the weight-fragment set-up above is paid once per workgroup ----------
    const int col = lane & 15;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    lds_barrier();   // previous tile's readers are done with the patch (LDS only: its stores keep draining)
    // ---- stage the normalised patch ---------------------------------------------------------
    // (all global loads are issued before the first LDS store so that their latencies overlap)
    constexpr int NPIX = PH * PW, NIT = (NPIX + 255) / 256;
    float sv[NIT][3];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int py = i / PW, px = i - py * PW;
        const int gy = y0 + py - 3, gx = x0 + px - 3;
        sv[it][0] = sv[it][1] = sv[it][2] = 0.f;
        if (i < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            if (a.src_is_u8) {
                const unsigned char* s = static_cast<const unsigned char*>(a.src) + (((size_t)b * a.H + gy) * a.W + gx) * 3;
                // image.float().sub_(mean).div_(std)  (rt_test.py:99-101)
                sv[it][0] = ((float)s[0] - a.mean[0]) / a.stdv[0];
                sv[it][1] = ((float)s[1] - a.mean[1]) / a.stdv[1];
                sv[it][2] = ((float)s[2] - a.mean[2]) / a.stdv[2];
            } else {
                const float* s = static_cast<const float*>(a.src) + ((size_t)b * 3 * a.H + gy) * a.W + gx;
                const size_t plane = (size_t)a.H * a.W;
                sv[it][0] = s[0]; sv[it][1] = s[plane]; sv[it][2] = s[2 * plane];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        if (i < NPIX) patch_store<T>(patch + (size_t)i * 4, sv[it][0], sv[it][1], sv[it][2]);
    }

    __syncthreads();

    // ---- 16 row segments of 16 pixels per wave ----------------------------------------------
    for (int sgi = 0; sgi < 16; ++sgi) {
        const int ry = wave * 4 + (sgi >> 2), sx = (sgi & 3) * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (BF) {
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                const __bf16* p = reinterpret_cast<const __bf16*>(patch) + ((size_t)(ry + dy) * PW + sx + col + 2 * g) * 4;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 4);
                bf16x8 xb;
#pragma unroll
                for (int i = 0; i < 4; ++i) { xb[i] = lo[i]; xb[4 + i] = hi[i]; }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[dy], xb, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const float xv = reinterpret_cast<const float*>(patch)[((size_t)(ry + dy) * PW + sx + col + dx) * 4 + g];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[dy * 7 + dx], xv, acc, 0, 0, 0);
                }
        }
        const int gy = y0 + ry, gx = x0 + sx + col;
        if (gy < a.H && gx < a.W) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = acc[r] * sc[r] + sh[r];
                v[r] = (t > 0.f || raw) ? t : 0.f;                         // BN + ReLU (drn.py:126-127)
            }
            const size_t o = (((size_t)b * a.H + gy) * a.W + gx) * CO + 4 * g;
            if (BF) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (__bf16)v[r];
                *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(a.out) + o) = ov;
            } else {
                *reinterpret_cast<float4*>(static_cast<float*>(a.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    }   // persistent tile loop
}

}  // namespace

namespace ppn {
int stem_launch(int dtype, int src_is_u8, const void* src, int batch, int h, int w, const float* weight,
                const float* scale, const float* shift, const float* mean, const float* stdv, void* out,
                hipStream_t st) {
    if (dtype != PPN_F32 && dtype != PPN_BF16) return fail(PPN_E_INVALID, "bad dtype %d", dtype);
    if (!src || !weight || (scale == nullptr) != (shift == nullptr) || !out || batch < 1 || h < 1 || w < 1)
        return fail(PPN_E_INVALID, "ppn_stem7x7: bad arguments");
    if (src_is_u8 && (!mean || !stdv)) return fail(PPN_E_INVALID, "ppn_stem7x7: mean/std required for u8 input");
    StemArgs a;
    a.src = src; a.weight = weight; a.scale = scale; a.shift = shift; a.out = out;
    a.B = batch; a.H = h; a.W = w; a.src_is_u8 = src_is_u8;
    for (int i = 0; i < 3; ++i) { a.mean[i] = mean ? mean[i] : 0.f; a.stdv[i] = stdv ? stdv[i] : 1.f; }
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * batch;
    if (blocks > 0x7fffffffLL) return fail(PPN_E_UNSUPPORTED, "too many tiles");
    const unsigned grid = (unsigned)(blocks < 256 * 4 ? blocks : 256 * 4);      // persistent: <= 4 workgroups per CU
    if (dtype == PPN_F32)
        hipLaunchKernelGGL(stem7x7_kernel<float>, dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(stem7x7_kernel<__bf16>, dim3(grid), dim3(256), 0, st, a);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
}  // namespace ppn

extern "C" int ppn_stem7x7(int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w,
                           const float* weight, const float* scale, const float* shift, const float* mean,
                           const float* std_, void* out, void* stream) {
    return ppn::stem_launch(dtype, src_is_u8, src, batch, h, w, weight, scale, shift, mean, std_, out,
                            static_cast<hipStream_t>(stream));
}
