// DRN-D stem layers 1 and 2 (drn.py:130-133, _make_conv_layers drn.py:192-202): 3x3 conv 16->16 (stride 1) and
// 16->32 (stride 2), each followed by BN + ReLU, at 384x384 / 192x192.  With 16 input channels these layers
// are HBM-bound (151 MB in + 151 MB out at batch 32 for layer 1), so they do not go through the implicit-GEMM
// kernels: a workgroup stages the (TH*S+2) x (TW*S+2) x 16 input patch of its TH x TW output tile in LDS with
// 16-byte coalesced loads (zero padding applied while staging), keeps ALL weights in registers as MFMA A
// fragments for the whole kernel, and every wave sweeps 16-pixel row segments:
//   bf16:  K = 9 taps x 16 ci = 144 -> 5 x v_mfma_f32_16x16x32_bf16 per segment and 16-channel tile
//   f32:   36 x v_mfma_f32_16x16x4_f32 (exact f32, parity mode)
// Each lane ends with 4 consecutive channels of one pixel, so a wave store is one contiguous run.
// Epilogue: v = relu(acc*scale1+shift1) -> out_raw;  optional out_act = relu(v*scale2+shift2), the
// pre-activation of the first BasicBlock (drn.py:45-46).
#include <hip/hip_bf16.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int CIN = 16;

struct Stem3Args {
    const void* src;       // NHWC [B,H,W,16] T
    const float* weight;   // [COUT][16][3][3] f32 (reference layout)
    const float* scale1;
    const float* shift1;
    const float* scale2;   // may be NULL
    const float* shift2;
    void* out_raw;         // NHWC [B,Ho,Wo,COUT] T or NULL
    void* out_act;         // NHWC T or NULL
    int B, H, W, Ho, Wo;
    int tiles_x, tiles_y;
};

template <typename T, int COUT, int S>
struct Geo {
    // f32 tiles are halved (round 4): 8 x 64 x stride-2 f32 output pixels needed a 150 KB patch -- ONE workgroup per CU, single
    // buffered -- and the stride-1 f32 patch 76 KB (two).  Same arithmetic per pixel: results unchanged.  Measured: 336 ->
    // 315 us (16 -> 16) and 247 -> 245 us (16 -> 32 stride 2) at batch 32: occupancy was not what holds these kernels at ~40 %
    // of the f32 MFMA rate (36 dependent ds_read_b32 + v_mfma_f32_16x16x4_f32 pairs per segment); they are the prefix of the
    // float16x3 and exact-prefix modes (0.84 of 9.0 / 4.1 ms).
    static constexpr bool F32 = sizeof(T) == 4;
    static constexpr int TH = S == 1 ? (F32 ? 8 : 16) : 8, TW = (F32 && S == 2) ? 32 : 64;
    static constexpr int PH = TH * S + 2, PW = TW * S + 2;
    static constexpr int LDS_BYTES = PH * PW * CIN * (int)sizeof(T);
};

template <typename T, int COUT, int S>
__global__ void __launch_bounds__(256) stem3x3_kernel(Stem3Args a) {
    using G = Geo<T, COUT, S>;
    constexpr int TH = G::TH, TW = G::TW, PH = G::PH, PW = G::PW;
    constexpr int ES = sizeof(T);
    constexpr int CPP = CIN * ES / 16;                   // 16-byte chunks per pixel (2 bf16 / 4 f32)
    constexpr int NCT = COUT / 16;                       // 16-channel output tiles
    constexpr bool BF = ES == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- weights as A fragments (rows = output channels) ------------------------------------------
    const int ch = lane & 15, g = lane >> 4;
    bf16x8 wa[NCT][5];       // bf16: k-step kk covers taps 2kk, 2kk+1;  k = 8g+i -> tap 2kk + (g>>1), ci = (g&1)*8 + i
    float wf[NCT][36];       // f32:  MFMA (tap, cq) covers ci = 4cq .. 4cq+3;  k = g -> ci = 4cq + g
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const float* wc = a.weight + (size_t)(ct * 16 + ch) * CIN * 9;
        if constexpr (BF) {
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int tap = 2 * kk + (g >> 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ci = (g & 1) * 8 + i;
                    wa[ct][kk][i] = (__bf16)(tap < 9 ? wc[ci * 9 + tap] : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) wf[ct][t * 4 + cq] = wc[(4 * cq + g) * 9 + t];
        }
    }
    float s1[NCT][4], b1[NCT][4], s2[NCT][4], b2[NCT][4];
    const bool raw = a.scale1 == nullptr;         // train mode / input gradient: plain convolution output
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = ct * 16 + 4 * g + r;
            s1[ct][r] = a.scale1 ? a.scale1[c] : 1.f; b1[ct][r] = a.scale1 ? a.shift1[c] : 0.f;
            s2[ct][r] = a.scale2 ? a.scale2[c] : 1.f; b2[ct][r] = a.shift2 ? a.shift2[c] : 0.f;
        }
    // ---- persistent loop over output tiles: the fragment set-up above is paid once per workgroup ----------
    const int col = lane & 15;
    const int ntiles = a.tiles_x * a.tiles_y * a.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;     // pad 1
    lds_barrier();   // previous tile's readers are done with the patch (LDS only: its stores keep draining)
    // ---- stage the input patch (zero padded) ------------------------------------------------------
    const char* src = static_cast<const char*>(a.src);
    // (all loads are issued before the first LDS store so that their latencies overlap)
    constexpr int NCHUNK = PH * PW * CPP, NIT = (NCHUNK + 255) / 256;
    uint4 stage[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int py = i / (PW * CPP), r = i - py * (PW * CPP);
        const int px = r / CPP, ch = r - px * CPP;
        const int gy = iy0 + py, gx = ix0 + px;
        stage[it] = make_uint4(0, 0, 0, 0);
        if (i < NCHUNK && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            stage[it] = *reinterpret_cast<const uint4*>(src + ((((size_t)b * a.H + gy) * a.W + gx) * CPP + ch) * 16);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        if (i < NCHUNK) *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = stage[it];
    }

    __syncthreads();

    // ---- TH*TW/16 row segments of 16 output pixels, TH*4/4 per wave ---------------------------------
    constexpr int NSEG = TH * (TW / 16);
    for (int sgi = wave; sgi < NSEG; sgi += 4) {
        const int ry = sgi / (TW / 16), sx = (sgi % (TW / 16)) * 16;
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int tap = 2 * kk + (g >> 1);
                const int t = tap < 9 ? tap : 0;                    // dead half of the last k-step: weights are 0
                const int dy = t / 3, dx = t - dy * 3;
                const int py = ry * S + dy, px = (sx + col) * S + dx;
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(smem + ((size_t)(py * PW + px) * CIN + (g & 1) * 8) * 2);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][kk], xb, acc[ct], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t - dy * 3;
                const int py = ry * S + dy, px = (sx + col) * S + dx;
                const float* xp = reinterpret_cast<const float*>(smem) + (size_t)(py * PW + px) * CIN + g;
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) {
                    const float xv = xp[4 * cq];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][t * 4 + cq], xv, acc[ct], 0, 0, 0);
                }
            }
        }
        const int oy = oy0 + ry, ox = ox0 + sx + col;
        if (oy < a.Ho && ox < a.Wo) {
            const size_t pix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float v[4], u[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = acc[ct][r] * s1[ct][r] + b1[ct][r];
                    v[r] = (t > 0.f || raw) ? t : 0.f;               // BN + ReLU (drn.py:198-200)
                    const float w2 = v[r] * s2[ct][r] + b2[ct][r];
                    u[r] = w2 > 0.f ? w2 : 0.f;                      // next block's relu(bn1(x)) (drn.py:45-46)
                }
                const size_t o = pix * COUT + ct * 16 + 4 * g;
                if constexpr (BF) {
                    bf16x4 ov, ou;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ov[r] = (__bf16)v[r]; ou[r] = (__bf16)u[r]; }
                    if (a.out_raw) *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(a.out_raw) + o) = ov;
                    if (a.out_act) *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(a.out_act) + o) = ou;
                } else {
                    if (a.out_raw) *reinterpret_cast<float4*>(static_cast<float*>(a.out_raw) + o) = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.out_act) *reinterpret_cast<float4*>(static_cast<float*>(a.out_act) + o) = make_float4(u[0], u[1], u[2], u[3]);
                }
            }
        }
    }
    }   // persistent tile loop
}

template <typename T, int COUT, int S>
int launch(const Stem3Args& a, hipStream_t st) {
    using G = Geo<T, COUT, S>;
    auto k = stem3x3_kernel<T, COUT, S>;
    {
        static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      G::LDS_BYTES);
    }
    Stem3Args b = a;
    b.tiles_x = (a.Wo + G::TW - 1) / G::TW; b.tiles_y = (a.Ho + G::TH - 1) / G::TH;
    const int ntiles = b.tiles_x * b.tiles_y * a.B;
    const int per_cu = G::LDS_BYTES <= 40 * 1024 ? 4 : (G::LDS_BYTES <= 53 * 1024 ? 3 : (G::LDS_BYTES <= 80 * 1024 ? 2 : 1));
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), G::LDS_BYTES, st, b);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // namespace

namespace ppn {
// 3x3, pad 1, Cin 16, Cout in {16,32}, stride in {1,2}; weight in the reference layout (f32, device).
bool stem3x3_supported(int cin, int cout, int ksize, int stride, int dilation, int pad) {
    return cin == 16 && (cout == 16 || cout == 32) && ksize == 3 && (stride == 1 || stride == 2) && dilation == 1 &&
           pad == 1;
}

int stem3x3_launch(int dtype, const void* src, int batch, int h, int w, int cout, int stride, const float* weight,
                   const float* scale1, const float* shift1, const float* scale2, const float* shift2, void* out_raw,
                   void* out_act, hipStream_t st, const char** kname) {
    if (!src || !weight || (scale1 == nullptr) != (shift1 == nullptr) || (!out_raw && !out_act) || batch < 1 ||
        h < 1 || w < 1)
        return fail(PPN_E_INVALID, "stem3x3: bad arguments");
    Stem3Args a;
    a.src = src; a.weight = weight; a.scale1 = scale1; a.shift1 = shift1; a.scale2 = scale2; a.shift2 = shift2;
    a.out_raw = out_raw; a.out_act = out_act;
    a.B = batch; a.H = h; a.W = w;
    a.Ho = (h + 2 - 3) / stride + 1; a.Wo = (w + 2 - 3) / stride + 1;
    a.tiles_x = a.tiles_y = 0;                            // per instantiation: launch<>()
    static char name[64];
    snprintf(name, sizeof(name), "stem3x3_kernel<%s, %d, %d>", dtype == PPN_F32 ? "float" : "__bf16", cout, stride);
    if (kname) *kname = name;
    if (dtype == PPN_F32) {
        if (cout == 16 && stride == 1) return launch<float, 16, 1>(a, st);
        if (cout == 32 && stride == 2) return launch<float, 32, 2>(a, st);
        if (cout == 16 && stride == 2) return launch<float, 16, 2>(a, st);
        return launch<float, 32, 1>(a, st);
    }
    if (cout == 16 && stride == 1) return launch<__bf16, 16, 1>(a, st);
    if (cout == 32 && stride == 2) return launch<__bf16, 32, 2>(a, st);
    if (cout == 16 && stride == 2) return launch<__bf16, 16, 2>(a, st);
    return launch<__bf16, 32, 1>(a, st);
}
}  // namespace ppn
