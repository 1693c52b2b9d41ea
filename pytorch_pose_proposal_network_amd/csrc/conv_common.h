// Shared device helpers of the fused implicit-GEMM convolution kernels (conv.hip, conv_big.hip).
#pragma once
#include <hip/hip_bf16.h>

#include "common.h"

namespace ppnconv {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Division by a launch-invariant divisor (Granlund-Montgomery, round-up variant): exact for every 32-bit unsigned
// dividend, 4 VALU instructions instead of the ~35 of an emulated integer division.  The index arithmetic at the
// head of a workgroup sits on the critical path of its first loads.
struct FastDiv {
    unsigned mul, sh1, sh2;
};
inline FastDiv make_fastdiv(unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                                     // ceil(log2 d)
    const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    return FastDiv{(unsigned)m, l < 1 ? l : 1u, l > 0 ? l - 1 : 0u};
}
__device__ __forceinline__ int fast_div(int n, const FastDiv& f) {   // n >= 0
    const unsigned t = __umulhi(f.mul, (unsigned)n);
    return (int)((t + (((unsigned)n - t) >> f.sh1)) >> f.sh2);
}

struct ConvKArgs {
    const char* src;
    const char* wgt;
    const float* scale1;
    const float* shift1;
    const char* residual;
    char* out_raw;
    const float* scale2;
    const float* shift2;
    char* out_act;
    const char* zero;
    int B, H, W, Cin, Ho, Wo, Cout, ks, stride, dil, pad;
    int Ktot;      // padded GEMM depth (multiple of BK)
    int M;         // end of this launch's output-pixel range (B*Ho*Wo for a whole-tensor launch)
    int m_base;    // first output pixel of this launch (ppn_conv_desc.m_begin): tile p covers m_base + p*BP ...
    int HoWo;
    FastDiv div_howo, div_wo, div_nct;   // dividers by HoWo, Wo, n_ctiles
    int act1, act2, nchw;
    int log2Cin;
    int n_ctiles, n_ptiles;
    // fused head mode (ppn_conv_desc.argmax_keys): unary channels -> compact tensor, limb windows -> arg-max keys
    float* unary_out;                 // f32 [B][unary_ch][HoWo]
    unsigned long long* amax_keys;    // u64 [B][n_edges][HoWo], zeroed by the caller
    int unary_ch, window;             // 6K (108), sH*sW (441)
    // fused 1x1 projection shortcut (ppn_conv_desc.src2): extra K steps gathered from a second tensor
    const char* src2;                 // NHWC [B][H2][W2][Cin2]
    int H2, W2, Cin2, stride2;
    int nsteps_main;                  // K steps of the main convolution; the rest belong to the shortcut
    unsigned src2_bytes;
    int lo_off;                       // PPN_F16X3: bytes from a pixel's hi block to its lo' block in the SOURCE tensor
    int out_bf16;                     // PPN_F16 launch whose NHWC outputs are stored as bf16 (PPN_CONV_OUT_BF16)
    int out_plain;                    // PPN_F16X3 launch whose NHWC outputs are stored as plain half (PPN_CONV_X3_PLAIN_OUT)
    const char* pf_ptr;               // ppn_conv_desc.prefetch: the next launch's weights, touched line by line (large-tile kernel)
    unsigned pf_lines, pf_per_wg;     // 128-byte lines in all / per workgroup (at most 2 per thread)
    // ppn_conv_desc.stats_*: BatchNorm partial sums from the epilogue (large-tile kernel, ST instantiations)
    double* st_partial;
    int st_mode, st_act;
    const char* st_x;
    const float *st_gamma, *st_beta, *st_mean, *st_rstd;
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
};
template <>
struct Elem<__bf16> {
    static constexpr int EPC = 8;
};
template <>
struct Elem<_Float16> {
    static constexpr int EPC = 8;
};
// kernel-name spelling of the element type (as rocprofv3 demangles it) and the 16-bit test
template <typename T> constexpr const char* elem_name() { return sizeof(T) == 4 ? "float" : "__bf16"; }
template <> constexpr const char* elem_name<_Float16>() { return "_Float16"; }

// sigmoid as exp + reciprocal instructions (v_exp_f32, v_rcp_f32: ~1 ulp each, far inside the 1e-4 head tolerance);
// every head value -- materialised or folded into arg-max keys -- goes through this one function.
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case PPN_ACT_RELU: return v > 0.f ? v : 0.f;
        case PPN_ACT_LRELU: return v > 0.f ? v : v * 0.1f;      // nn.LeakyReLU(0.1), model.py:88
        case PPN_ACT_SIGMOID: return sigmoid_fast(v);
        default: return v;
    }
}

__device__ __forceinline__ void glds16(const char* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gptr,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// One K-substep (4 chunks = 128 B/4 of a row): acc += Wfrag x Xfrag
__device__ __forceinline__ void mma_step(f32x4& acc, const f32x4& wf, const f32x4& xf, float*) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, xf.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, xf.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, xf.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, xf.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(f32x4& acc, const f32x4& wf, const f32x4& xf, __bf16*) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf),
                                                  acc, 0, 0, 0);
}

__device__ __forceinline__ void mma_step(f32x4& acc, const f32x4& wf, const f32x4& xf, _Float16*) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, xf), acc, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void load8(const char* p, float* v);
template <>
__device__ __forceinline__ void load8<float>(const char* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 16);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<__bf16>(const char* p, float* v) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const unsigned u[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(u[i] << 16);
        v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void load8<_Float16>(const char* p, float* v) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
// IEEE half overflows to +-inf past 65504; the f16 mode stores RAW pre-activation residual streams, and one inf becomes
// a NaN in the next residual add / BN affine and a garbage decode.  Every f16 store path clamps to the largest finite
// half instead (one v_med3_f32; a NaN stays a NaN).
__device__ __forceinline__ float clamp_f16(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }

template <typename T>
__device__ __forceinline__ void store8(char* p, const float* v);
template <>
__device__ __forceinline__ void store8<float>(char* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 16) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<__bf16>(char* p, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];                 // RNE, v_cvt_pk_bf16_f32
    *reinterpret_cast<bf16x8*>(p) = o;
}

template <>
__device__ __forceinline__ void store8<_Float16>(char* p, const float* v) {
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)clamp_f16(v[i]);    // RNE (v_cvt_f16_f32) on a value held inside the finite range
    *reinterpret_cast<f16x8*>(p) = o;
}

// PPN_F16X3 storage: a value v as the half pair (hi, lo') = (half(v), half((v - hi) * 2^11)); lo' is kept at hi's
// magnitude so that it never falls into the half subnormals while hi is normal.  `lo_bytes`: distance hi -> lo' block.
constexpr float kX3LoScale = 2048.f, kX3LoInv = 1.f / 2048.f;
__device__ __forceinline__ void store8_x3(char* p, size_t lo_bytes, const float* v) {
    f16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (_Float16)clamp_f16(v[i]);
        l[i] = (_Float16)clamp_f16((v[i] - (float)h[i]) * kX3LoScale);   // the difference is exact in f32; |.| <= ulp(hi)/2 * 2^11
    }
    *reinterpret_cast<f16x8*>(p) = h;
    *reinterpret_cast<f16x8*>(p + lo_bytes) = l;
}
__device__ __forceinline__ void load8_x3(const char* p, size_t lo_bytes, float* v) {
    const f16x8 h = *reinterpret_cast<const f16x8*>(p), l = *reinterpret_cast<const f16x8*>(p + lo_bytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i] * kX3LoInv;
}

// conv_big.hip: 512-thread, (BP x BC) = ({128,192,256} x {128,256}) tiles for Cin % K-step == 0 layers
struct BigTile {
    int bp, bc;
};
// edge-aligned limb tile of the head conv (ppn_conv_desc.limb_edge_pad): 128 pixels x one edge's window padded to 448 rows
constexpr int kEdgeTileBP = 128, kEdgeTileBC = 448;
// ksteps: K steps of the launch (0 = unknown); shared_gpu: the launch shares the GPU with other streams' launches
// (ppn_conv_desc.flags & PPN_CONV_SHARED_GPU): tiles that only shorten a LONE launch are left out
bool big_tile_for(int cout, long long m, BigTile* out, int ksteps = 0, bool shared_gpu = false);
// Split point of a two-segment launch (0 = single launch): pixels [0, split) run whole rounds of the most efficient
// tile, the rest a smaller tile that fills one more round (conv_big.hip).
long long big_split_for(int cout, long long m);
int launch_big(const ConvKArgs& a, int dtype, BigTile t, hipStream_t st, const char** kname);
bool big_stats_ok(int dtype, BigTile t);   // the tile has the BatchNorm-statistics epilogue (ppn_conv_desc.stats_mode)

}  // namespace ppnconv
