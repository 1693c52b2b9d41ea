// DRN-D layer0 + layer1 in ONE kernel: 7x7 conv 3->16 + BN + ReLU (drn.py:123-128) feeding 3x3 conv 16->16 + BN +
// ReLU (drn.py:130, 192-202), with the input normalisation of rt_test.py:97-101 fused into the patch load.
//
// Run separately, layer0 writes and layer1 re-reads a 16 x 384 x 384 activation (151 MB at batch 32 in bf16): both
// layers are HBM-bound.  Here a workgroup owns a 16 x 78 tile of layer1's output, computes the 18 x 80 tile of
// layer0 outputs it needs (1-pixel halo, 20 % redundant layer0 work) into LDS and consumes it from there:
//   LDS  in0  [24][88][4]   normalised input patch (3-pixel halo of the 7x7, +2 columns for the dx=7 slot)
//        mid  [18+][80][16] layer0 output after BN+ReLU, ZERO outside the image (layer1's zero padding applies to
//                           layer0's output, not to conv(zeros))
//   bf16: 7 + 5 x v_mfma_f32_16x16x32_bf16 per 16 pixels; f32: 49 + 36 x v_mfma_f32_16x16x4_f32 (exact, parity mode).
// Weights of both layers stay in registers as MFMA A fragments; each lane ends with 4 consecutive channels of one
// pixel, so wave stores are contiguous.
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int TH = 16, TW = 78;                  // layer1 output tile
constexpr int MH = TH + 2, MW = 80;              // layer0 output tile (1-pixel halo; 80 = 5 MFMA segments)
constexpr int MROWS = MH + 1;                    // +1 row so that masked lanes of the last segment stay in bounds
constexpr int IH = MH + 6, IW = MW + 8;          // input patch (3-pixel halo, +2 columns for the dx=7 slot)

struct Stem01Args {
    const void* src;
    const float *w0, *s0, *b0;                   // [16][3][7][7], [16], [16]
    const float *w1, *s1, *b1;                   // [16][16][3][3], [16], [16]
    void* out;                                   // NHWC [B,H,W,16]
    int B, H, W, src_is_u8;
    float mean[3], stdv[3];
    int tiles_x, tiles_y;
};

template <typename T>
__device__ __forceinline__ void store4ch(T* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4ch<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4ch<__bf16>(__bf16* p, float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    *reinterpret_cast<bf16x4*>(p) = v;
}

template <typename T>
__global__ void __launch_bounds__(256) stem01_kernel(Stem01Args a) {
    constexpr bool BF = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* in0 = reinterpret_cast<T*>(smem);                              // [IH][IW][4]
    T* mid = in0 + IH * IW * 4;                                       // [MROWS][MW][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;                             // layer1 output origin
    const int my0 = y0 - 1, mx0 = x0 - 1;                             // layer0 output origin (pad 1)
    const int iy0 = my0 - 3, ix0 = mx0 - 3;                           // input origin (pad 3)

    // ---- stage the normalised input patch (all loads first, then the LDS stores) -----------------
    constexpr int NPIX = IH * IW, NIT = (NPIX + 255) / 256;
    float sv[NIT][3];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int py = i / IW, px = i - py * IW;
        const int gy = iy0 + py, gx = ix0 + px;
        sv[it][0] = sv[it][1] = sv[it][2] = 0.f;
        if (i < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            if (a.src_is_u8) {
                const unsigned char* s = static_cast<const unsigned char*>(a.src) + (((size_t)b * a.H + gy) * a.W + gx) * 3;
                sv[it][0] = ((float)s[0] - a.mean[0]) / a.stdv[0];     // rt_test.py:99-101
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
        if (i < NPIX) store4ch<T>(in0 + (size_t)i * 4, sv[it][0], sv[it][1], sv[it][2], 0.f);
    }
    // the spare row of `mid` is only read by masked lanes, but must hold finite values
    for (int i = tid; i < MW * 16 / 4; i += 256) store4ch<T>(mid + (size_t)MH * MW * 16 + i * 4, 0.f, 0.f, 0.f, 0.f);

    // ---- weights of both layers as A fragments (rows = output channels) ---------------------------
    const int ch = lane & 15, g = lane >> 4, col = lane & 15;
    bf16x8 wa0[7], wa1[5];
    float wf0[49], wf1[36];
    {
        const float* wc = a.w0 + (size_t)ch * 3 * 49;
        const float* wd = a.w1 + (size_t)ch * 16 * 9;
        if constexpr (BF) {
#pragma unroll
            for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int dx = 2 * g + (i >> 2), c = i & 3;
                    wa0[dy][i] = (__bf16)((dx < 7 && c < 3) ? wc[(c * 7 + dy) * 7 + dx] : 0.f);
                }
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int tap = 2 * kk + (g >> 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) wa1[kk][i] = (__bf16)(tap < 9 ? wd[((g & 1) * 8 + i) * 9 + tap] : 0.f);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 49; ++t) wf0[t] = (g < 3) ? wc[g * 49 + t] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) wf1[t * 4 + cq] = wd[(4 * cq + g) * 9 + t];
        }
    }
    float sc0[4], sh0[4], sc1[4], sh1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc0[r] = a.s0[4 * g + r]; sh0[r] = a.b0[4 * g + r];
        sc1[r] = a.s1[4 * g + r]; sh1[r] = a.b1[4 * g + r];
    }
    __syncthreads();

    // ---- layer0: MH x 5 row segments of 16 pixels -> mid (BN + ReLU, zero outside the image) ---------
    for (int sgi = wave; sgi < MH * (MW / 16); sgi += 4) {
        const int ry = sgi / (MW / 16), sx = (sgi % (MW / 16)) * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                const __bf16* p = reinterpret_cast<const __bf16*>(in0) + ((size_t)(ry + dy) * IW + sx + col + 2 * g) * 4;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 4);
                bf16x8 xb;
#pragma unroll
                for (int i = 0; i < 4; ++i) { xb[i] = lo[i]; xb[4 + i] = hi[i]; }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa0[dy], xb, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int dy = 0; dy < 7; ++dy)
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const float xv = reinterpret_cast<const float*>(in0)[((size_t)(ry + dy) * IW + sx + col + dx) * 4 + g];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[dy * 7 + dx], xv, acc, 0, 0, 0);
                }
        }
        const int gy = my0 + ry, gx = mx0 + sx + col;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = acc[r] * sc0[r] + sh0[r];
            v[r] = (inside && t > 0.f) ? t : 0.f;                      // BN + ReLU (drn.py:126-127); layer1 pads with 0
        }
        store4ch<T>(mid + ((size_t)ry * MW + sx + col) * 16 + 4 * g, v[0], v[1], v[2], v[3]);
    }
    __syncthreads();

    // ---- layer1: TH x 5 row segments (the last one has 14 valid pixels) ---------------------------------
    constexpr int NSEG1 = (TW + 15) / 16;
    for (int sgi = wave; sgi < TH * NSEG1; sgi += 4) {
        const int ry = sgi / NSEG1, sx = (sgi % NSEG1) * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int tap = 2 * kk + (g >> 1);
                const int t = tap < 9 ? tap : 0;
                const int dy = t / 3, dx = t - dy * 3;
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(
                    reinterpret_cast<const __bf16*>(mid) + ((size_t)(ry + dy) * MW + sx + col + dx) * 16 + (g & 1) * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1[kk], xb, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t - dy * 3;
                const float* xp = reinterpret_cast<const float*>(mid) + ((size_t)(ry + dy) * MW + sx + col + dx) * 16 + g;
#pragma unroll
                for (int cq = 0; cq < 4; ++cq)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[t * 4 + cq], xp[4 * cq], acc, 0, 0, 0);
            }
        }
        const int oy = y0 + ry, ox = x0 + sx + col;
        if (sx + col < TW && oy < a.H && ox < a.W) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = acc[r] * sc1[r] + sh1[r];
                v[r] = t > 0.f ? t : 0.f;
            }
            store4ch<T>(static_cast<T*>(a.out) + (((size_t)b * a.H + oy) * a.W + ox) * 16 + 4 * g, v[0], v[1], v[2], v[3]);
        }
    }
}

template <typename T>
constexpr size_t stem01_lds() { return ((size_t)IH * IW * 4 + (size_t)MROWS * MW * 16) * sizeof(T); }

}  // namespace

namespace ppn {
int stem01_launch(int dtype, int src_is_u8, const void* src, int batch, int h, int w, const float* w0,
                  const float* s0, const float* b0, const float* mean, const float* stdv, const float* w1,
                  const float* s1, const float* b1, void* out, hipStream_t st) {
    if (dtype != PPN_F32 && dtype != PPN_BF16) return fail(PPN_E_INVALID, "bad dtype %d", dtype);
    if (!src || !w0 || !s0 || !b0 || !w1 || !s1 || !b1 || !out || batch < 1 || h < 1 || w < 1)
        return fail(PPN_E_INVALID, "ppn_stem01: bad arguments");
    if (src_is_u8 && (!mean || !stdv)) return fail(PPN_E_INVALID, "ppn_stem01: mean/std required for u8 input");
    Stem01Args a;
    a.src = src; a.w0 = w0; a.s0 = s0; a.b0 = b0; a.w1 = w1; a.s1 = s1; a.b1 = b1; a.out = out;
    a.B = batch; a.H = h; a.W = w; a.src_is_u8 = src_is_u8;
    for (int i = 0; i < 3; ++i) { a.mean[i] = mean ? mean[i] : 0.f; a.stdv[i] = stdv ? stdv[i] : 1.f; }
    a.tiles_x = (w + TW - 1) / TW; a.tiles_y = (h + TH - 1) / TH;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * batch;
    if (blocks > 0x7fffffffLL) return fail(PPN_E_UNSUPPORTED, "too many tiles");
    if (dtype == PPN_F32) {
        static int max_lds_set = 0;
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(stem01_kernel<float>),
                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem01_lds<float>());
        hipLaunchKernelGGL(stem01_kernel<float>, dim3((unsigned)blocks), dim3(256), stem01_lds<float>(), st, a);
    } else {
        static int max_lds_set = 0;
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(stem01_kernel<__bf16>),
                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem01_lds<__bf16>());
        hipLaunchKernelGGL(stem01_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), stem01_lds<__bf16>(), st, a);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
}  // namespace ppn

extern "C" int ppn_stem01(int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w,
                          const float* w0, const float* scale0, const float* shift0, const float* mean,
                          const float* std_, const float* w1, const float* scale1, const float* shift1, void* out,
                          void* stream) {
    return ppn::stem01_launch(dtype, src_is_u8, src, batch, h, w, w0, scale0, shift0, mean, std_, w1, scale1, shift1,
                              out, static_cast<hipStream_t>(stream));
}
