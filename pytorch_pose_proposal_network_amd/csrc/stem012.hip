// DRN-D stem in ONE kernel (bf16 mode): layer0 7x7 conv 3->16 + BN + ReLU (drn.py:123-128), layer1 3x3 16->16 + BN +
// ReLU and layer2 3x3 stride 2 16->32 + BN + ReLU (drn.py:130-133, 192-202), with the input normalisation of
// rt_test.py:97-101 fused into the load and the pre-activation relu(bn1(x)) of the first BasicBlock (drn.py:45-46) as
// the second output.
//
// Run layer by layer the stem is HBM-bound: two 151 MB tensors (16 x 384 x 384 at batch 32) are written and read
// back between three kernels.  Here they never leave the CU.  A workgroup (4 waves) owns a band of 16 layer-2 rows x
// 48 layer-2 columns of one image and walks down it in chunks of 2 layer-2 rows, keeping ROLLING row buffers in LDS,
// so no row is computed twice inside a band:
//   raw bytes     10 rows x 368 B               the u8 frame rows as they lie in memory, filled by LDS-DMA
//   input patch   10 rows x 120 px x 4 ch      normalised (256-entry table per channel), zero outside the image
//   layer-0 ring   6 rows x 120 px x 16 ch     after BN + ReLU, ZERO outside the image (layer1 pads layer0's OUTPUT)
//   layer-1 ring   5 rows x 112 px x 16 ch     likewise for layer2's padding
// per chunk: 10 input rows -> 4 new layer-0 rows -> 4 new layer-1 rows -> 2 layer-2 rows (8-byte stores to HBM); one
// wave per row, its 7 segments of 16 pixels unrolled with immediate LDS offsets.  The u8 rows of chunk c+1 are requested
// (global_load_lds, aligned dwords: no registers) right after chunk c's patch has been converted, and the layer-2 rows
// of chunk c-1 are computed at the head of chunk c, so both the loads and the stores have a whole chunk of MFMA work
// to complete under.  First version of this kernel: 60 M VALU instructions and 68 M LDS-array cycles per batch-32
// launch (rocprofv3 SQ counters), i.e. instruction- and LDS-bound at 300+ us; this version computes addresses per ROW
// instead of per segment, normalises through a table instead of three divisions per pixel, keeps the BN constants in
// registers and reads the 7x7's operands with single ds_read_b64 (2 LDS cycles each; the compiler's merged
// ds_read2_b64 takes 8 for the same 16 bytes).
// 54.6 KB of LDS per workgroup, two workgroups per CU; HBM traffic = the u8 frames (re-read 2.5x through L2) + the two
// 32-channel outputs.  The three weight sets stay in registers as MFMA A fragments for the whole kernel
// (v_mfma_f32_16x16x32_bf16: 7 + 5 + 2x5 per 16 pixels); intermediate activations are rounded to bf16 exactly where
// the layer-by-layer kernels (stem.hip, stem3x3.hip) round them, and the MFMA sequences are the same, so the results
// are bit-identical to theirs (tests/test_conv_gpu.py::test_fused_stem_equals_layer_by_layer).
// Round 4 (tools/bench_stem.py, batch 32): 176 -> 155 us.  (1) the weights reach the lanes through LDS (the guarded per-lane
// global loads of the 7x7 fragments were 66 serialised L2 round trips at the head of every workgroup: -11 us); (2) layer 0 /
// layer 1 read the operands of segment s+1 under the MFMA chain of segment s (-6 us); layer 2's 32 BN constants wait in LDS.
// (3) convert_input issues the byte reads of a thread's five rows together (-5 us; with the compiler's merged, unaligned
// ds_read_u16 the same change cost +15 us).  Phase by phase (timing-only builds, -DPPN_S012_SKIP): convert 25,
// layer 0 47, layer 1 43, layer 2 31, input requests 21 us -- additive, no unit saturated (SQ counters of the 159 us
// kernel: MFMA pipe 25 %, VALU issue 39 %, LDS 45 % of which a fifth bank conflicts, 36 % of the wave cycles in s_waitcnt):
// round 4 read this as "latency-bound at two waves per SIMD (223 VGPRs: the three weight sets); the step change needs wave-specialised
// producers / consumers under 128 VGPRs each".  Round 5 built exactly that (8 waves = front role: requests, conversion, layer 0 | back
// role: layer 1, layer 2; 128 VGPRs, two workgroups = 16 waves per CU; bit-identical): 170 us against 168 -- and at equal wave count the
// role split is SLOWER than two independent workgroups (186 vs 168).  The phases are additive because the launch is bound by the
// throughput of the shared pipes (12.2 M LDS instructions = ~60 % of the LDS cycles, VALU 39 %, MFMA 25 %), not by waiting: what would
// shorten it is fewer LDS instructions per pixel, not another wave schedule (profiles/r05/stem_role_split.txt).
#include <hip/hip_bf16.h>

#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

// timing-only diagnostics (results wrong): -DPPN_S012_SKIP=mask, 1 convert, 2 layer0, 4 layer1, 8 layer2, 16 requests, 32 layer-2 epilogue
#ifndef PPN_S012_SKIP
#define PPN_S012_SKIP 0
#endif


namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int SW2 = 48;                 // layer-2 columns per strip
constexpr int BAND2 = 16;               // layer-2 rows per work unit
constexpr int NS = 7;                   // 16-pixel segments per layer-0 / layer-1 row
constexpr int W1 = NS * 16;             // layer-1 columns held: x1 = 2*C2 - 1 + i   (97 of them feed layer 2)
constexpr int W0 = W1 + 8;              // layer-0 columns held: x0 = 2*C2 - 2 + i   (112 computed; the spare ones are only read
                                        // by layer-1 columns nobody uses).  120: see the plane layout below
// Ring layout: [row][quarter q = 0..3][pixel] x 8 bytes (channels 4q .. 4q+3), not [row][pixel][16 channels]: an MFMA
// result lane (pixel n, quarter g) then stores 8 bytes next to its 15 neighbours of the same quarter -- 128 contiguous
// bytes per 16 lanes, conflict-free -- where the pixel-major layout put them 32 bytes apart (4-way bank conflict on
// every ds_write_b64: 12.9 M conflict cycles per launch).  Readers fetch a lane's 8 input channels as two ds_read_b64
// from planes 2h and 2h+1.  With W0 = 120 two planes are 1920 B = 128 (mod 256) apart, so the two lane groups of a
// ds_read_b64 (h = 0 / h = 1) fall on different halves of the 64 banks.
// Layer-1 ring (round 5): layer 2 reads it with STRIDE 2 (16 bytes between lanes), so the 16 lanes of one k-group g use every
// other 8-byte slot of a 256-byte bank row, and the two k-groups of a ds_read_b64 lane group (g = 0 / 1: planes 2(g&1)) must
// land on opposite slot parities: the planes are W1P = 113 pixels apart (904 B = 8 mod 16) and stored in the physical order
// q -> [0, 2, 1, 3][q], so that the planes the two k-groups read in the same instruction (q = 0 / 2, then 1 / 3) are
// NEIGHBOURS.  Before (pitch 112, natural order: 1792 B = 0 mod 256 between them) every layer-2 read was 2- to 4-way
// conflicted: 3.8 M of the kernel's 5.2 M conflict cycles (profiles/r05/stem_conflicts_by_phase.txt).
#ifdef PPN_S012_OLD_L1                   // A/B build of the round-4 layer-1 ring (tools/build_variant.py s012old stem012.hip -DPPN_S012_OLD_L1)
constexpr int W1P = W1;
#define PPN_S012_PQ(g) (g)
#define PPN_S012_LO(g) (2 * ((g) & 1))
constexpr int kHiPlanes = 1;
#else
constexpr int W1P = W1 + 1;
#define PPN_S012_PQ(g) ((((g) & 1) << 1) | ((g) >> 1))
#define PPN_S012_LO(g) ((g) & 1)
constexpr int kHiPlanes = 2;
#endif
constexpr int WI = W1 + 8;              // input columns held:   xi = 2*C2 - 5 + i   (column i+7 meets a zero weight)
constexpr int R0 = 6, R1 = 5, RI = 10;  // ring / patch rows
constexpr int RAWS = 368;               // raw u8 row: 120 px x 3 B = 360 B (+ alignment slack), a multiple of 16
constexpr int LDS_IN = RI * WI * 8, LDS_L0 = R0 * W0 * 32, LDS_L1 = R1 * 4 * W1P * 8, LDS_RAW = RI * RAWS, LDS_LUT = 3 * 256 * 2;
constexpr int LDS_CST = 4 * 32 * 4;      // layer-2 epilogue constants [scale2 | shift2 | scale3 | shift3][32] f32
constexpr int LDS_BYTES = LDS_IN + LDS_L0 + LDS_L1 + LDS_RAW + LDS_LUT + LDS_CST;

struct Stem012Args {
    const void* src;                    // u8 [B,H,W,3] or f32 [B,3,H,W]
    const float *w0, *s0, *b0;          // [16][3][7][7], folded BN scale / shift [16]
    const float *w1, *s1, *b1;          // [16][16][3][3], [16]
    const float *w2, *s2, *b2;          // [32][16][3][3], [32]
    const float *s3, *b3;               // second output: relu(v * s3 + b3)  (NULL: no second output)
    void* out_raw;                      // NHWC bf16 [B,Ho,Wo,32] or NULL
    void* out_act;                      // NHWC bf16 [B,Ho,Wo,32] or NULL
    int B, H, W, Ho, Wo, src_is_u8;
    int raw_s2;                         // PPN_STEM_RAW_S2: out_raw holds only the even (row, column) pixels, [B,(Ho+1)/2,(Wo+1)/2,32]
    float mean[3], stdv[3];
    int nstrips, nbands;
};

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// one 8-byte LDS read that the compiler cannot merge into ds_read2_b64 (which costs 8 LDS cycles instead of 2 x 2)
template <int OFF>
__device__ __forceinline__ u32x2 lds_read64(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c, int, int, int) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c, int, int, int) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// non-negative float -> storage type; the IEEE-half instantiation stays finite (65504) instead of overflowing to +inf
template <typename H>
__device__ __forceinline__ H to_store(float v) {
    if constexpr (sizeof(H) == 2 && !__is_same(H, __bf16)) return (H)fminf(v, 65504.f);
    else return (H)v;
}

// BN + ReLU of one accumulator tile -> 4 packed bf16 channels (zero when `inside` is false)
template <typename H>
__device__ __forceinline__ u32x2 bn_relu_pack(const f32x4& acc, const float (&sc)[4], const float (&sh)[4], bool inside) {
    typedef __attribute__((ext_vector_type(4))) H hx4;
    hx4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float t = acc[r] * sc[r] + sh[r];
        o[r] = to_store<H>(t > 0.f ? t : 0.f);
    }
    u32x2 p = __builtin_bit_cast(u32x2, o);
    p.x = inside ? p.x : 0u;
    p.y = inside ? p.y : 0u;
    return p;
}

// H: the 16-bit MFMA operand type of the stem and the storage type of everything that stays on chip (input patch, weights,
// layer-0 / layer-1 rings): __bf16, or _Float16 (same schedule, same rate).  HO: the storage type of the two OUTPUT tensors
// = the trunk's type.  Round 4: the bf16 mode runs H = _Float16, HO = __bf16 -- the stem is 1.8 % of the FLOPs but its
// rounding noise is amplified by all ~30 layers behind it; with IEEE-half internals (11 significant bits instead of 8) the
// bf16 pipeline reproduces 148 instead of 94 of the reference's 260 people (emulated, tests/precision_study_mixed.py) at
// the same speed.
template <bool U8, typename H, typename HO = H>
__global__ void __launch_bounds__(256, 2) stem012_kernel(Stem012Args a) {
    typedef __attribute__((ext_vector_type(8))) H hx8;
    typedef __attribute__((ext_vector_type(4))) H hx4;
    typedef __attribute__((ext_vector_type(4))) HO hox4;
    // EXACT INPUT (u8 frames, IEEE-half internals; round 4): the normalised pixel (x - mean_c) / std_c is not a half, and its
    // rounding is amplified by the whole network behind the stem (tests/precision_study_mixed.py: with the input exact the
    // f16 pipeline reproduces 233 instead of 225 of the reference's 260 people in emulation).  So the patch holds the INTEGER
    // x - 128 (exact in half) and the normalisation moves into layer 0's weights: conv(w, (x - mean) / std) =
    // conv(w / std, x - 128) + conv(w4, inside) with w4 = sum_c w_c (128 - mean_c) / std_c and `inside` = 1 where the tap
    // lies in the image, 0 in the zero padding -- the patch's fourth channel, which was a zero pad, carries it.
    constexpr bool kExactIn = U8 && std::is_same<H, _Float16>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_p = smem;                                  // [RI][WI][4] bf16
    char* l0_p = smem + LDS_IN;                         // [R0][W0][16] bf16
    char* l1_p = smem + LDS_IN + LDS_L0;                // [R1][W1][16] bf16
    char* raw_p = smem + LDS_IN + LDS_L0 + LDS_L1;      // [RI][RAWS] u8
    unsigned short* lut_p = reinterpret_cast<unsigned short*>(smem + LDS_IN + LDS_L0 + LDS_L1 + LDS_RAW);   // [3][256] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = lane & 15, g = lane >> 4, col = lane & 15;

    // ---- weights as MFMA A fragments (rows = output channels), same k layout as stem.hip / stem3x3.hip ----------
    hx8 wa0[7];           // per dy: k = (dx = 2g + (i>>2), c = i&3)
    hx8 wa1[5];           // k-step kk: k = 8g+i -> tap 2kk + (g>>1), ci = (g&1)*8 + i
    hx8 wa2[2][5];
    {
        // The three weight sets go through LDS once per workgroup: coalesced dword loads by all 256 threads (36 each, one
        // wait), then every lane picks its fragment elements with UNCONDITIONAL LDS reads (clamped index, value masked
        // afterwards).  Reading them per lane from global memory with guarded loads compiled to one branch +
        // global_load + s_waitcnt vmcnt(0) each -- 66 serialised L2 round trips at the head of every workgroup.
        float* wl = reinterpret_cast<float*>(smem);      // [16*147 | 16*144 | 32*144] f32 over the not-yet-used patch + rings
        constexpr int N0 = 16 * 147, N1 = 16 * 144, N2 = 32 * 144;
        static_assert((N0 + N1 + N2) * 4 <= LDS_IN + LDS_L0 + LDS_L1, "weight staging must fit below the raw rows / LUT");
        for (int i = tid; i < N0; i += 256) wl[i] = a.w0[i];
        for (int i = tid; i < N1; i += 256) wl[N0 + i] = a.w1[i];
        for (int i = tid; i < N2; i += 256) wl[N0 + N1 + i] = a.w2[i];
        __syncthreads();
        const float* wc = wl + ch * 3 * 49;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int dx = 2 * g + hf;
                const bool ok = dx < 7;
                const int dxc = ok ? dx : 6;
                float w3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) w3[c] = wc[(c * 7 + dy) * 7 + dxc];
                if constexpr (kExactIn) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        wa0[dy][4 * hf + c] = (H)(ok ? w3[c] / a.stdv[c] : 0.f);
                        v += w3[c] * ((128.f - a.mean[c]) / a.stdv[c]);
                    }
                    wa0[dy][4 * hf + 3] = (H)(ok ? v : 0.f);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) wa0[dy][4 * hf + c] = (H)(ok ? w3[c] : 0.f);
                    wa0[dy][4 * hf + 3] = (H)0.f;
                }
            }
        const float* wd = wl + N0 + ch * 16 * 9;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int tap = 2 * kk + (g >> 1), tapc = tap < 9 ? tap : 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = wd[((g & 1) * 8 + i) * 9 + tapc];
                wa1[kk][i] = (H)(tap < 9 ? v : 0.f);
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float* we = wl + N0 + N1 + (ct * 16 + ch) * 16 * 9;
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int tap = 2 * kk + (g >> 1), tapc = tap < 9 ? tap : 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = we[((g & 1) * 8 + i) * 9 + tapc];
                    wa2[ct][kk][i] = (H)(tap < 9 ? v : 0.f);
                }
            }
        }
    }
    // layer 0 / 1 constants in registers; layer 2's (used once per 16 output pixels, 32 registers) wait in LDS
    float sc0[4], sh0[4], sc1[4], sh1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sc0[r] = a.s0[4 * g + r]; sh0[r] = a.b0[4 * g + r];
        sc1[r] = a.s1[4 * g + r]; sh1[r] = a.b1[4 * g + r];
    }
    float* cst_p = reinterpret_cast<float*>(smem + LDS_IN + LDS_L0 + LDS_L1 + LDS_RAW + LDS_LUT);
    if (tid < 128) {
        const int c = tid & 31, which = tid >> 5;
        float v;
        if (which == 0) v = a.s2[c];
        else if (which == 1) v = a.b2[c];
        else if (which == 2) v = a.s3 ? a.s3[c] : 1.f;
        else v = a.b3 ? a.b3[c] : 0.f;
        cst_p[tid] = v;
    }
    if constexpr (U8 && !kExactIn) {
        // normalisation table: image.float().sub_(mean).div_(std) (rt_test.py:99-101) of every u8 value, rounded to
        // bf16 as the patch stores it -- the same expression stem.hip evaluates per pixel
        for (int i = tid; i < 3 * 256; i += 256) {
            const int c = i >> 8;
            const H v = (H)(((float)(i & 255) - a.mean[c]) / a.stdv[c]);
            lut_p[i] = __builtin_bit_cast(unsigned short, v);
        }
    }
    // per-lane tap geometry of the 3x3 k-steps: tap t = 2kk + (g>>1) -> (dy, dx); the dead half of k-step 4 reads tap 0
    int tdy[5], tdx[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int tap = 2 * kk + (g >> 1), t = tap < 9 ? tap : 0;
        tdy[kk] = t / 3; tdx[kk] = t - (t / 3) * 3;
    }
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address

    const int units = a.B * a.nbands * a.nstrips;
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        int u = unit;
        const int strip = u % a.nstrips; u /= a.nstrips;
        const int band = u % a.nbands;
        const int b = u / a.nbands;
        const int C2 = strip * SW2, R2 = band * BAND2;
        const int R2e = min(R2 + BAND2, a.Ho);
        const int x0b = 2 * C2 - 2;                    // image column of layer-0 ring column 0
        const int x1b = 2 * C2 - 1;                    // image column of layer-1 ring column 0
        const int xib = 2 * C2 - 5;                    // image column of input patch column 0

        // u8 source: request input rows yi0 .. yi0+nrows-1, image columns xs .. xib+119, into the raw buffer by LDS-DMA:
        // 4-byte-ALIGNED dwords from the aligned address below a row's first byte (global_load_lds_dword, lane L lands
        // at base + 4 L; the 1- and 2-byte forms also occupy a dword per lane, tools/probes/glds_ubyte.hip).  An aligned
        // dword never crosses a page, so the one that holds the tensor's last byte is safe to read.  Rows outside the
        // image are skipped and never looked at by convert_input.  f32 source: nothing to request.
        const int xs = xib < 0 ? 0 : xib;                              // first image column held by the raw rows
        const int nb_row = (min(xib + WI, a.W) - xs) * 3;              // bytes of a row that matter
        auto request_input = [&](int yi0, int nrows) {
            if constexpr (U8 && !(PPN_S012_SKIP & 16)) {
                const unsigned char* base = static_cast<const unsigned char*>(a.src);
                for (int q = wave; q < nrows * 2; q += 4) {
                    const int r = q >> 1, half = q & 1;
                    const int gy = yi0 + r;
                    if (gy < 0 || gy >= a.H) continue;                               // wave-uniform
                    const size_t S = (((size_t)b * a.H + gy) * a.W + xs) * 3;        // first byte that matters
                    const size_t A = S & ~(size_t)3;
                    const int k = half * 64 + lane;                                  // dword of the row
                    if (k < RAWS / 4 && A + 4 * (size_t)k < S + nb_row)
                        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + A + 4 * (size_t)k),
                                                         (void __attribute__((address_space(3)))*)(raw_p + r * RAWS + half * 256),
                                                         4, 0, 0);
                }
            }
        };
        // raw rows -> patch rows 0 .. nrows-1: normalised bf16 [px][4], zero outside the image (patch row 0 = input row yi0).
        // A thread keeps its column: 240 threads cover 2 rows x 120 columns per pass.
        auto convert_input = [&](int yi0, int nrows) {
            if (PPN_S012_SKIP & 1) return;
            if (tid >= 2 * WI) return;
            const int prow = tid >= WI ? 1 : 0, px = tid - prow * WI;
            const int gx = xib + px;
            const bool colok = gx >= 0 && gx < a.W;
            if constexpr (U8) {
                // The 15 byte reads of a thread's five rows are issued together, UNCONDITIONALLY (a row or column outside
                // the image reads stale bytes that are masked below), then -- unless the patch holds the exact integer --
                // the 15 table look-ups together: two LDS round trips per call instead of ten (-5 us per launch).  Single
                // ds_read_u8 on purpose: the compiler merges two of a pixel's bytes into an UNALIGNED ds_read_u16, which
                // cost 15 us per launch.
                const unsigned base = smem_base + (unsigned)(LDS_IN + LDS_L0 + LDS_L1) + prow * RAWS + (colok ? (gx - xs) * 3 : 0);
                const unsigned sh0 = (((unsigned)b * a.H + (unsigned)(yi0 + prow)) * a.W + xs) * 3u;   // low 2 bits: S & 3 of row `prow`
                const unsigned dsh = 2u * a.W * 3u;                                                     // ... and per two rows
                unsigned v[RI / 2][3];
#pragma unroll
                for (int k = 0; k < RI / 2; ++k) {
                    const unsigned ad = base + 2 * k * RAWS + ((sh0 + k * dsh) & 3u);
                    asm volatile("ds_read_u8 %0, %1" : "=v"(v[k][0]) : "v"(ad));
                    asm volatile("ds_read_u8 %0, %1 offset:1" : "=v"(v[k][1]) : "v"(ad));
                    asm volatile("ds_read_u8 %0, %1 offset:2" : "=v"(v[k][2]) : "v"(ad));
                }
                // The wait is TIED to the 15 values ("+v"): the reads are inline asm, so the compiler does not know that their
                // results arrive later, and an untied s_waitcnt lets it schedule the first USE of a value (the table address
                // below, the integer conversion) in front of the wait.  Round 4's batched reads did exactly that in the bf16
                // instantiation: ~0.06 % of the stem's outputs differed from run to run (stale registers as table indices), the
                // all-bf16 pipeline fell from 95 to 37-60 of the reference's 260 people and was no longer deterministic
                // (tests/test_16bit_floors_gpu.py::test_every_16bit_configuration_is_deterministic).
#define PPN_S012_WAIT15(v)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
                 : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[2][0]),   \
                   "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[3][0]), "+v"(v[3][1]), "+v"(v[3][2]), "+v"(v[4][0]), "+v"(v[4][1]),   \
                   "+v"(v[4][2])::"memory")
                static_assert(RI / 2 == 5, "PPN_S012_WAIT15 names 5 x 3 values");
                PPN_S012_WAIT15(v);
                if constexpr (!kExactIn) {
                    const unsigned lb = smem_base + (unsigned)(LDS_IN + LDS_L0 + LDS_L1 + LDS_RAW);
#pragma unroll
                    for (int k = 0; k < RI / 2; ++k)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const unsigned ad = lb + 512 * c + 2 * v[k][c];
                            asm volatile("ds_read_u16 %0, %1" : "=v"(v[k][c]) : "v"(ad));
                        }
                    PPN_S012_WAIT15(v);
                }
#pragma unroll
                for (int k = 0; k < RI / 2; ++k) {
                    const int py = prow + 2 * k;
                    if (py >= nrows) break;
                    const int gy = yi0 + py;
                    u32x2 o = {0u, 0u};
                    if (colok && gy >= 0 && gy < a.H) {
                        if constexpr (kExactIn) {
                            // the patch holds the integer x - 128, exact in half; fourth channel: half(1.0) = inside the image
                            hx4 t;
#pragma unroll
                            for (int c = 0; c < 3; ++c) t[c] = (H)(float)((int)v[k][c] - 128);
                            t[3] = (H)1.f;
                            o = __builtin_bit_cast(u32x2, t);
                        } else {
                            o.x = v[k][0] | (v[k][1] << 16);
                            o.y = v[k][2];
                        }
                    }
                    *reinterpret_cast<u32x2*>(in_p + ((size_t)py * WI + px) * 8) = o;
                }
            } else {
#pragma unroll
                for (int k = 0; k < RI / 2; ++k) {
                    const int py = prow + 2 * k;
                    if (py >= nrows) break;
                    const int gy = yi0 + py;
                    u32x2 o = {0u, 0u};
                    if (colok && gy >= 0 && gy < a.H) {
                        const float* sp = static_cast<const float*>(a.src) + ((size_t)b * 3 * a.H + gy) * a.W + gx;
                        const size_t plane = (size_t)a.H * a.W;
                        hx4 t;
                        t[0] = (H)sp[0]; t[1] = (H)sp[plane]; t[2] = (H)sp[2 * plane]; t[3] = (H)0.f;
                        o = __builtin_bit_cast(u32x2, t);
                    }
                    *reinterpret_cast<u32x2*>(in_p + ((size_t)py * WI + px) * 8) = o;
                }
            }
        };
        // every wave's requests have landed (and its earlier stores have been acknowledged), for all waves
        auto input_landed = [&]() {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // layer-0 rows y .. y+nrows-1 (patch row 0 = input row y - 3) -> ring; wave w computes row y + w
        auto layer0 = [&](int y, int nrows) {
            if (PPN_S012_SKIP & 2) return;
            if (wave >= nrows) return;
            const int gy = y + wave;
            const bool rowok = gy >= 0 && gy < a.H;
            const unsigned rd = smem_base + (unsigned)((wave * WI + col + 2 * g) * 8);            // patch (row, col + 2g)
            char* wr = l0_p + ((size_t)(((gy + 2 * R0) % R0) * 4 + g) * W0 + col) * 8;
            // software pipeline: the 14 reads of segment sg + 1 are in flight under the MFMA chain of segment sg (LDS ops
            // complete in order, so "at most 14 outstanding" = segment sg's operands and the previous ds_write are done)
            u32x2 lo[2][7], hi[2][7];
            auto fetch = [&](auto sgc, auto setc) {
                constexpr int sg = decltype(sgc)::value, st = decltype(setc)::value;
                sfor<7>([&](auto dyc) {                  // immediate offsets: no address arithmetic per read
                    constexpr int dy = decltype(dyc)::value;
                    lo[st][dy] = lds_read64<dy * WI * 8 + sg * 128>(rd);
                    hi[st][dy] = lds_read64<dy * WI * 8 + sg * 128 + 8>(rd);
                });
            };
            fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            sfor<NS>([&](auto sgc) {
                constexpr int sg = decltype(sgc)::value, cur = sg & 1;
                if constexpr (sg + 1 < NS) {
                    fetch(std::integral_constant<int, sg + 1>{}, std::integral_constant<int, cur ^ 1>{});
                    asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy < 7; ++dy) {
                    const u32x4 xb = {lo[cur][dy].x, lo[cur][dy].y, hi[cur][dy].x, hi[cur][dy].y};
                    acc = mfma16(wa0[dy], __builtin_bit_cast(hx8, xb), acc, 0, 0, 0);
                }
                const int gx = x0b + sg * 16 + col;
                // BN + ReLU (drn.py:126-127); zero outside the image: layer1 pads layer0's output with 0
                *reinterpret_cast<u32x2*>(wr + sg * 128) = bn_relu_pack<H>(acc, sc0, sh0, rowok && gx >= 0 && gx < a.W);
            });
        };
        // layer-1 rows y .. y+nrows-1 from layer-0 rows y-1 .. y+nrows -> ring; wave w computes row y + w
        auto layer1 = [&](int y, int nrows) {
            if (PPN_S012_SKIP & 4) return;
            if (wave >= nrows) return;
            const int gy = y + wave;
            const bool rowok = gy >= 0 && gy < a.H;
            const int s0 = (gy - 1 + 2 * R0) % R0;                        // ring slot of layer-0 row gy - 1
            unsigned rd[5];                                               // LDS byte address of (row, plane 2h, col + dx)
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                int s = s0 + tdy[kk];
                s = s >= R0 ? s - R0 : s;
                rd[kk] = smem_base + (unsigned)(LDS_IN + ((s * 4 + 2 * (g & 1)) * W0 + col + tdx[kk]) * 8);
            }
            char* wr = l1_p + ((size_t)(((gy + 2 * R1) % R1) * 4 + PPN_S012_PQ(g)) * W1P + col) * 8;   // physical plane [0,2,1,3][g]
            u32x2 lo[2][5], hi[2][5];
            auto fetch = [&](auto sgc, auto setc) {
                constexpr int sg = decltype(sgc)::value, st = decltype(setc)::value;
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    lo[st][kk] = lds_read64<sg * 128>(rd[kk]);
                    hi[st][kk] = lds_read64<sg * 128 + W0 * 8>(rd[kk]);
#ifdef PPN_S012_DUP_L1      // diagnostic: half of layer 1's operand reads issued twice (is the launch bound by LDS traffic?)
                    lo[st][kk] = lds_read64<sg * 128>(rd[kk]);
#endif
                }
            };
            fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            sfor<NS>([&](auto sgc) {
                constexpr int sg = decltype(sgc)::value, cur = sg & 1;
                if constexpr (sg + 1 < NS) {
                    fetch(std::integral_constant<int, sg + 1>{}, std::integral_constant<int, cur ^ 1>{});
#ifdef PPN_S012_DUP_L1
                    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
#else
                    asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
#endif
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    const u32x4 xb = {lo[cur][kk].x, lo[cur][kk].y, hi[cur][kk].x, hi[cur][kk].y};
                    acc = mfma16(wa1[kk], __builtin_bit_cast(hx8, xb), acc, 0, 0, 0);
                }
                const int gx = x1b + sg * 16 + col;
                *reinterpret_cast<u32x2*>(wr + sg * 128) = bn_relu_pack<H>(acc, sc1, sh1, rowok && gx >= 0 && gx < a.W);
            });
        };
        // layer-2 rows oy0 .. oy0+nrows-1 (stride 2) from layer-1 rows 2oy-1 .. -> HBM; waves 0,1 take row 0, waves 2,3
        // row 1: the even wave segments 0 and 1, the odd wave segment 2
        auto layer2 = [&](int oy0, int nrows) {
            if (PPN_S012_SKIP & 8) return;
            const int ry = wave >> 1;
            if (ry >= nrows) return;
            const int oy = oy0 + ry;
            const int s0 = (2 * oy - 1 + 2 * R1) % R1;
            const int sg0 = (wave & 1) * 2, nsg = (wave & 1) ? 1 : 2;
            unsigned rd[5];                                               // (row, plane 2h, 2 * pixel + dx) of the layer-1 ring
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                int s = s0 + tdy[kk];
                s = s >= R1 ? s - R1 : s;
                // channels 8(g&1) .. +3 = plane q = 2(g&1) at physical plane g&1; the next four (q + 1) two physical planes on
                rd[kk] = smem_base + (unsigned)(LDS_IN + LDS_L0 + ((s * 4 + PPN_S012_LO(g)) * W1P + 2 * (sg0 * 16 + col) + tdx[kk]) * 8);
            }
            for (int sg = 0; sg < nsg; ++sg) {
                const int ox = C2 + (sg0 + sg) * 16 + col;
                u32x2 lo[5], hi[5];
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    lo[kk] = lds_read64<0>(rd[kk] + sg * 256);
                    hi[kk] = lds_read64<kHiPlanes * W1P * 8>(rd[kk] + sg * 256);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    const u32x4 xw = {lo[kk].x, lo[kk].y, hi[kk].x, hi[kk].y};
                    const hx8 xb = __builtin_bit_cast(hx8, xw);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[ct] = mfma16(wa2[ct][kk], xb, acc[ct], 0, 0, 0);
                }
                if constexpr ((PPN_S012_SKIP & 32) != 0) {            // timing / counter builds: no layer-2 epilogue
                    asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
                    continue;
                }
                if (oy < a.Ho && ox < a.Wo) {
                    const size_t pix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const f32x4 sc2 = *reinterpret_cast<const f32x4*>(cst_p + ct * 16 + 4 * g);
                        const f32x4 sh2 = *reinterpret_cast<const f32x4*>(cst_p + 32 + ct * 16 + 4 * g);
                        const f32x4 sc3 = *reinterpret_cast<const f32x4*>(cst_p + 64 + ct * 16 + 4 * g);
                        const f32x4 sh3 = *reinterpret_cast<const f32x4*>(cst_p + 96 + ct * 16 + 4 * g);
                        hox4 ov, ou;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t = acc[ct][r] * sc2[r] + sh2[r];
                            const float v = t > 0.f ? t : 0.f;               // BN + ReLU (drn.py:198-200)
                            const float w2 = v * sc3[r] + sh3[r];
                            ov[r] = to_store<HO>(v);
                            ou[r] = to_store<HO>(w2 > 0.f ? w2 : 0.f);          // next block's relu(bn1(x)) (drn.py:45-46)
                        }
                        const size_t o = pix * 32 + ct * 16 + 4 * g;
                        if (a.out_raw) {
                            if (!a.raw_s2) *reinterpret_cast<hox4*>(static_cast<HO*>(a.out_raw) + o) = ov;
                            else if (((oy | ox) & 1) == 0)      // the raw tensor's only reader is a 1x1 stride-2 convolution
                                *reinterpret_cast<hox4*>(static_cast<HO*>(a.out_raw) + ((((size_t)b * ((a.Ho + 1) >> 1) + (oy >> 1)) *
                                                         ((a.Wo + 1) >> 1) + (ox >> 1)) * 32 + ct * 16 + 4 * g)) = ov;
                        }
                        if (a.out_act) *reinterpret_cast<hox4*>(static_cast<HO*>(a.out_act) + o) = ou;
                    }
                }
            }
        };

        // ---- warm-up of the band: layer-0 rows 2R2-2 .. 2R2, layer-1 row 2R2-1 ------------------------------------
        lds_barrier();                                  // the previous unit's readers are done with every buffer
        request_input(2 * R2 - 5, 9);
        input_landed();
        convert_input(2 * R2 - 5, 9);
        lds_barrier();
        request_input(2 * R2 - 2, RI);                  // rows of the first chunk, under the warm-up's MFMAs
        layer0(2 * R2 - 2, 3);
        lds_barrier();
        layer1(2 * R2 - 1, 1);
        // ---- chunks of 2 layer-2 rows: 4 new layer-0 rows, 4 new layer-1 rows; the layer-2 rows of a chunk are
        // computed at the head of the NEXT chunk (their stores then have a whole chunk to drain) ------------------
        for (int r2 = R2; r2 < R2e; r2 += 2) {
            input_landed();                             // input rows 2r2-2 .. 2r2+7; also: layer1 of the previous chunk done
            convert_input(2 * r2 - 2, RI);
            lds_barrier();
            if (r2 + 2 < R2e) request_input(2 * r2 + 2, RI);
            if (r2 > R2) layer2(r2 - 2, 2);
            layer0(2 * r2 + 1, 4);
            lds_barrier();
            layer1(2 * r2, 4);
        }
        lds_barrier();
        const int last = R2 + ((R2e - R2 - 1) / 2) * 2;
        layer2(last, min(2, R2e - last));
    }
}

}  // namespace

namespace ppn {
template <bool U8, typename H, typename HO = H>
static int stem012_launch_T(const Stem012Args& a, unsigned grid, hipStream_t st) {
    static int max_lds_set = 0;
    PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(stem012_kernel<U8, H, HO>),
                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((stem012_kernel<U8, H, HO>), dim3(grid), dim3(256), LDS_BYTES, st, a);
    return PPN_OK;
}

int stem012_launch(int dtype, int src_is_u8, const void* src, int batch, int h, int w, const float* w0, const float* s0,
                   const float* b0, const float* mean, const float* stdv, const float* w1, const float* s1,
                   const float* b1, const float* w2, const float* s2, const float* b2, const float* s3, const float* b3,
                   void* out_raw, void* out_act, hipStream_t st) {
    if (!src || !w0 || !s0 || !b0 || !w1 || !s1 || !b1 || !w2 || !s2 || !b2 || (!out_raw && !out_act) || batch < 1 ||
        h < 1 || w < 1 || (s3 == nullptr) != (b3 == nullptr))
        return fail(PPN_E_INVALID, "ppn_stem012: bad arguments");
    if (src_is_u8 && (!mean || !stdv)) return fail(PPN_E_INVALID, "ppn_stem012: mean/std required for u8 input");
    Stem012Args a;
    a.src = src; a.w0 = w0; a.s0 = s0; a.b0 = b0; a.w1 = w1; a.s1 = s1; a.b1 = b1; a.w2 = w2; a.s2 = s2; a.b2 = b2;
    a.s3 = s3; a.b3 = b3; a.out_raw = out_raw; a.out_act = out_act;
    a.B = batch; a.H = h; a.W = w; a.src_is_u8 = src_is_u8;
    a.Ho = (h + 2 - 3) / 2 + 1; a.Wo = (w + 2 - 3) / 2 + 1;
    for (int i = 0; i < 3; ++i) { a.mean[i] = mean ? mean[i] : 0.f; a.stdv[i] = stdv ? stdv[i] : 1.f; }
    a.nstrips = (a.Wo + SW2 - 1) / SW2; a.nbands = (a.Ho + BAND2 - 1) / BAND2;
    const long long units = (long long)batch * a.nstrips * a.nbands;
    if (units > 0x7fffffffLL) return fail(PPN_E_UNSUPPORTED, "too many tiles");
    if ((long long)batch * h * w * 3 > 0xffffffffLL) return fail(PPN_E_UNSUPPORTED, "frames too large for 32-bit byte offsets");
    static const int per_cu = getenv("PPN_S012_WGS") ? atoi(getenv("PPN_S012_WGS")) : 2;   // tuning knob
    const unsigned grid = (unsigned)(units < 256 * per_cu ? units : 256 * per_cu);   // persistent: 2 workgroups per CU
    // dtype: the stem's internal type in the low byte; PPN_STEM_IO(internal, out) adds a different OUTPUT storage type
    const int din = dtype & 0xff, dout = ((dtype >> 8) & 0xff) ? ((dtype >> 8) & 0xff) - 1 : din;
    a.raw_s2 = (dtype & PPN_STEM_RAW_S2) ? 1 : 0;
    if ((din != PPN_BF16 && din != PPN_F16) || (dout != PPN_BF16 && dout != PPN_F16) || (din == PPN_BF16 && dout != PPN_BF16))
        return fail(PPN_E_INVALID, "ppn_stem012: dtype must be PPN_BF16, PPN_F16 or PPN_STEM_IO(PPN_F16, PPN_BF16)");
    int rc;
    if (din == PPN_F16 && dout == PPN_BF16)
        rc = src_is_u8 ? stem012_launch_T<true, _Float16, __bf16>(a, grid, st) : stem012_launch_T<false, _Float16, __bf16>(a, grid, st);
    else if (din == PPN_F16)
        rc = src_is_u8 ? stem012_launch_T<true, _Float16>(a, grid, st) : stem012_launch_T<false, _Float16>(a, grid, st);
    else
        rc = src_is_u8 ? stem012_launch_T<true, __bf16>(a, grid, st) : stem012_launch_T<false, __bf16>(a, grid, st);
    if (rc != PPN_OK) return rc;
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
}  // namespace ppn

extern "C" int ppn_stem012(int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w, const float* w0,
                           const float* scale0, const float* shift0, const float* mean, const float* std_,
                           const float* w1, const float* scale1, const float* shift1, const float* w2,
                           const float* scale2, const float* shift2, const float* scale3, const float* shift3,
                           void* out_raw, void* out_act, void* stream) {
    return ppn::stem012_launch(PPN_BF16, src_is_u8, src, batch, h, w, w0, scale0, shift0, mean, std_, w1, scale1, shift1,
                               w2, scale2, shift2, scale3, shift3, out_raw, out_act, static_cast<hipStream_t>(stream));
}

extern "C" int ppn_stem012_dt(int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w,
                              const float* w0, const float* scale0, const float* shift0, const float* mean,
                              const float* std_, const float* w1, const float* scale1, const float* shift1,
                              const float* w2, const float* scale2, const float* shift2, const float* scale3,
                              const float* shift3, void* out_raw, void* out_act, void* stream) {
    return ppn::stem012_launch(dtype, src_is_u8, src, batch, h, w, w0, scale0, shift0, mean, std_, w1, scale1, shift1, w2,
                               scale2, shift2, scale3, shift3, out_raw, out_act, static_cast<hipStream_t>(stream));
}
