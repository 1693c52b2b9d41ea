// One launch per 64-channel stride-1 pre-activation BasicBlock (drn.py:25-57: bn1 -> relu -> conv1 -> bn2 -> relu -> conv2 -> += x;
// DRN-D's layer3, 96 x 96 for a 384 x 384 input), 16-bit modes.  The two 3x3 convolutions of the block run in ONE persistent
// kernel and the tensor between them never goes to HBM:
//
//   roles      a workgroup is 8 waves = two ROLES of 4 waves; every SIMD hosts one wave of each.  Role 0 computes conv1 (+ bn2 +
//              ReLU) of output tile t+1 into an LDS "mid" tile while role 1 computes conv2 (+ residual, second output) of tile
//              t from the mid tile role 0 wrote one phase earlier: a two-stage software pipeline with ONE workgroup barrier per
//              tile.  Each role keeps ITS convolution's whole 64 x 576 filter bank in registers as MFMA A fragments (32 channels
//              x 576 = 144 VGPRs per wave, as csrc/conv64.hip does for one convolution): no weight traffic in the loop.
//   tiles      8 x 16 output pixels.  conv2 needs the 10 x 18 mid pixels around them, conv1 the 12 x 20 input pixels around
//              those.  Input patch and mid tile share the row pitch 20 and are indexed LINEARLY (pixel (r, c) = row 20 r + c of
//              128 bytes = 64 channels): a filter tap is then a constant row shift 20 dy + dx for ANY 16 consecutive rows, so
//              conv1 walks the mid tile as 13 MFMA pixel tiles of 16 linear rows (208 rows cover the 10 x 18 region; the 28 rows
//              that are not mid pixels are computed and discarded) and conv2 as 8 row-aligned ones.
//   swizzle    row L keeps its eight 16-byte chunks XORed with (L >> 1) & 7 (source side for the LDS-DMA of the input patch,
//              store side for the mid tile): the key of row 16 n + r + 20 dy + dx is ((r + dx) >> 1) + 2 dy mod 8 whatever the
//              tile, so nine (conv1) / twelve (conv2) per-lane offsets plus immediates address every fragment read.
//   borders    input pixels outside the image arrive as zeros (out-of-range buffer offsets); mid pixels outside the image are
//              STORED as zeros (conv2's zero padding applies to relu(bn2(conv1)), drn.py:45-51), not computed from padding.
//   epilogue   conv2's epilogue works on the accumulators as they lie (4 consecutive channels of a pixel per lane and MFMA tile):
//              the residual rows were fetched into LDS by role 0 one phase ahead, outputs leave as 8-byte stores (no staging
//              tile, no barrier, no memory wait).  The arithmetic -- K order tap-major in 32-deep halves, v = act1(acc*s1+b1) (+res), out_act =
//              act2(v*s2+b2), one rounding per stored value -- is that of the two separate launches (conv.hip / conv_big.hip /
//              conv64.hip), so the block's outputs are BIT-IDENTICAL to them.
//
// Work per tile: conv1 13 x 4 x 18 + conv2 8 x 4 x 18 MFMAs (1.31x the separate launches' FLOPs: the halo is recomputed) against
// half their HBM traffic (x_act + x in, two tensors out; the mid tensor, 37.7 MB at batch 32, stays on chip) and one launch.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

using namespace ppnconv;

constexpr unsigned kOOB = 0x80000000u;

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)lds_wave_base, 16, voff, 0, 0, 0);
}

struct BlkArgs {
    const char* src;                 // relu(bn1(x)): NHWC [B][H][W][64]
    const char* res;                 // x (the block's residual) or NULL
    const char* w1;                  // packed [64][576], k = tap * 64 + ci
    const char* w2;
    const float *sm, *bm;            // bn2 folded (between the convolutions), act_mid
    const float *s1, *b1;            // conv2's own affine (NULL for a BasicBlock), act1
    const float *s2, *b2;            // second output: the next block's bn1 folded, act2
    char* out_raw;
    char* out_act;
    int B, H, W, act_mid, act1, act2;
    int tiles_x, tiles_y, n_tiles;   // 16-wide x 8-high output tiles per image
    FastDiv div_tx, div_tpi;
    unsigned long long* dbg;         // -DPPN_CLOCK builds only
};

constexpr int TH = 8, TW = 16;                    // output tile
constexpr int PW = 20;                            // row pitch of the input patch AND of the mid tile
constexpr int IN_ROWS = 256;                      // 12 x 20 = 240 rows are filled; conv1's last pixel tile reads up to row 249
constexpr int IN_DMA = 240 / 8;                   // 30 wave-instructions of 8 rows
constexpr int IN_BYTES = IN_ROWS * 128;
constexpr int MID_TILES = 13;                     // 16-row MFMA pixel tiles of conv1: linear mid rows 0 .. 207 (pixel half 0: tiles 0-6, half 1: 6-12)
constexpr int MID_BYTES = MID_TILES * 16 * 128;   // 26 624
constexpr int RES_BYTES = TH * TW * 128;           // the residual rows of one output tile (16 384)
constexpr int CST_BYTES = 6 * 64 * 4;
constexpr int OFF_MID = 2 * IN_BYTES, OFF_RES = OFF_MID + 2 * MID_BYTES, OFF_CST = OFF_RES + 2 * RES_BYTES;
constexpr int LDS_BYTES = OFF_CST + CST_BYTES;    // 153 088 of 163 840
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

template <typename T> struct Pack4;
template <> struct Pack4<_Float16> {
    typedef __attribute__((ext_vector_type(4))) _Float16 v4;
    static __device__ __forceinline__ uint2 pack(const float* v) {
        v4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (_Float16)clamp_f16(v[i]);
        return __builtin_bit_cast(uint2, o);
    }
};
template <> struct Pack4<__bf16> {
    typedef __attribute__((ext_vector_type(4))) __bf16 v4;
    static __device__ __forceinline__ uint2 pack(const float* v) {
        v4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
        return __builtin_bit_cast(uint2, o);
    }
};

template <typename T>
__global__ void __launch_bounds__(512, 2) block64_kernel(BlkArgs a, unsigned tensor_bytes) {
    static_assert(sizeof(T) == 2, "16-bit modes only");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, rw = wave & 3;    // role 0: conv1, role 1: conv2 (waves w and w + 4 share a SIMD)
    const int wc = rw >> 1, wp = rw & 1;          // 2 (channel halves) x 2 (pixel halves)
    float* cst = reinterpret_cast<float*>(smem + OFF_CST);           // [sm | bm | s1 | b1 | s2 | b2][64]

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, tensor_bytes, 0x00020000);

    // ---- this role's filter bank as MFMA A fragments: wf[tap][half][channel tile] ----------------------------------------
    f32x4 wf[9][2][2];
    {
        const char* wg = role ? a.w2 : a.w1;
        const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ch = wc * 32 + i * 16 + frow;
                    wf[t][h][i] = *reinterpret_cast<const f32x4*>(wg + ((size_t)ch * 576 + t * 64 + h * 32 + fq * 8) * 2);
                }
    }
    if (tid < 64) {
        cst[tid] = a.sm ? a.sm[tid] : 1.f;       cst[64 + tid] = a.bm ? a.bm[tid] : 0.f;
        cst[128 + tid] = a.s1 ? a.s1[tid] : 1.f; cst[192 + tid] = a.b1 ? a.b1[tid] : 0.f;
        cst[256 + tid] = a.s2 ? a.s2[tid] : 1.f; cst[320 + tid] = a.b2 ? a.b2[tid] : 0.f;
    }
    const float slope_m = a.act_mid == PPN_ACT_RELU ? 0.f : (a.act_mid == PPN_ACT_LRELU ? 0.1f : 1.f);
    const float slope1 = a.act1 == PPN_ACT_RELU ? 0.f : (a.act1 == PPN_ACT_LRELU ? 0.1f : 1.f);
    const float slope2 = a.act2 == PPN_ACT_RELU ? 0.f : (a.act2 == PPN_ACT_LRELU ? 0.1f : 1.f);

    const int G = gridDim.x, first = blockIdx.x;
    const int n = (a.n_tiles - first + G - 1) / G;               // tiles of this workgroup (>= 1: the grid never exceeds n_tiles)
    auto tile_pos = [&](int tile, int& img, int& ty, int& tx) {
        img = fast_div(tile, a.div_tpi);
        const int rem = tile - img * (a.tiles_x * a.tiles_y);
        ty = fast_div(rem, a.div_tx);
        tx = rem - ty * a.tiles_x;
    };

    // ---- role 0: the 12 x 20 input patch of a tile by LDS-DMA (30 instructions of 8 rows over the role's 4 waves) ----------
    // A wave's instructions g = rw, rw + 4, .. cover patch rows row0 + 32 i: the lane's (patch y, patch x) of instruction 0 is a
    // constant of the kernel and moves by (+1, +12) or (+2, -8) per instruction, its XOR chunk ((row >> 1) & 7) does not move at all,
    // so an instruction is a handful of adds and compares.  (The straightforward form -- row / 20, two 32-bit multiplies and the
    // swizzle per instruction, 227 cycles each by the stamps -- made role 0 the longer role: 2.7 k of its 11.3 k cycles per tile.)
    const int prow0 = rw * 8 + (lane >> 3);
    const int ppy0 = prow0 / PW, ppx0 = prow0 - ppy0 * PW;
    const int pchunk = ((lane & 7) ^ ((prow0 >> 1) & 7)) * 16;
    static_assert(PW == 20, "issue_patch steps 32 rows = one patch line + 12 pixels");
    auto issue_patch = [&](int tile, int buf) {
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        const int y0 = ty * TH - 2, x0 = tx * TW - 2;
        const int base = ((img * a.H + y0) * a.W + x0) * 128;          // wave-uniform; the sum with `rel` is only used where in range
        const int step = (a.W + 12) * 128, wrap = (a.W - PW) * 128;
        int py = ppy0, px = ppx0;
        int rel = (ppy0 * a.W + ppx0) * 128 + pchunk;
        for (int g = rw; g < IN_DMA; g += 4) {
            const int y = y0 + py, x = x0 + px;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            bufload_lds16(xrs, smem + buf * IN_BYTES + g * 1024, ok ? (unsigned)(base + rel) : kOOB);
            px += 12; py += 1; rel += step;
            if (px >= PW) { px -= PW; py += 1; rel += wrap; }
        }
    };

    // ---- role 0 also fetches the RESIDUAL rows of a tile (8 x 16 pixels, 16 instructions) for role 1, one phase ahead: role 1
    // then needs neither registers nor a memory wait for them (row r keeps its chunks XORed with r & 7)
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.src), 0, a.res ? tensor_bytes : 0u, 0x00020000);
    auto issue_res = [&](int tile, int buf) {                          // rows row0 + 32 i: two tile lines down, same column, same chunk
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        static_assert(TW == 16, "issue_res steps 32 rows = two tile lines");
        const int x = tx * TW + (prow0 & 15);
        int y = ty * TH + (prow0 >> 4);
        int off = ((img * a.H + y) * a.W + x) * 128 + ((lane & 7) ^ (prow0 & 7)) * 16;
        const int step = 2 * a.W * 128;
        const bool xok = x < a.W;
        for (int g = rw; g < TH * TW / 8; g += 4) {
            bufload_lds16(rrs, smem + OFF_RES + buf * RES_BYTES + g * 1024, (xok && y < a.H) ? (unsigned)off : kOOB);
            y += 2; off += step;
        }
    };

    // ---- role 0: conv1 + bn2 + ReLU of one tile: input patch `buf` -> mid tile `buf` ---------------------------------------
    // The role's 4 waves = 2 channel halves x 2 pixel halves; pixel half 0 computes the MFMA pixel tiles 0 .. 6 of the 13,
    // half 1 the tiles 6 .. 12 (tile 6 twice, same values: seven tiles per wave without a branch in the MFMA stream), each in
    // two passes of 4 + 3 tiles so that 32 accumulator registers are live beside the 144 of the filter bank.
    auto conv1_tile = [&](int tile, int buf) {
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        int frow = lane & 15, fq = lane >> 4;
        asm volatile("" : "+v"(frow), "+v"(fq));                     // (keeps the address expressions inside the tile loop)
        const int tb = wp * 6;                                       // first pixel tile of this wave
        const unsigned pbo = (unsigned)(buf * IN_BYTES + (tb * 16 + frow) * 128);
        unsigned a9[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a9[dy][dx] = pbo + (unsigned)((((((frow + dx) >> 1) + 2 * dy) & 7) ^ fq) << 4);
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const unsigned mbo = (unsigned)(OFF_MID + buf * MID_BYTES + (tb * 16 + frow) * 128 + (fq & 1) * 8);
        const unsigned ck0 = (unsigned)((wc * 4 + (fq >> 1)) ^ ((frow >> 1) & 7)) << 4;
        const unsigned ck1 = (unsigned)((wc * 4 + 2 + (fq >> 1)) ^ ((frow >> 1) & 7)) << 4;
        auto pass = [&](auto j0c, auto njc) {
            constexpr int J0 = decltype(j0c)::value, NJ = decltype(njc)::value;
            auto xread = [&](int s, int j) {                         // fragment of half-step s (tap s / 2, half s % 2), pixel tile j
                const int t = s >> 1, h = s & 1, dy = t / 3, dx = t % 3;
                return *reinterpret_cast<const f32x4*>(smem + (a9[dy][dx] ^ (unsigned)(h << 6)) + (16 * j + PW * dy + dx) * 128);
            };
            f32x4 acc[2][NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // 18 half-steps x NJ pixel tiles, fragments read three items ahead of their MFMAs (a ring of four)
            constexpr int NIT = 18 * NJ, AHEAD = 3;
            f32x4 ring[4];
            static_for<AHEAD>([&](auto nc) {
                constexpr int it = decltype(nc)::value;
                ring[it & 3] = xread(it / NJ, J0 + it % NJ);
            });
            static_for<NIT>([&](auto nc) {
                constexpr int it = decltype(nc)::value;
                constexpr int s = it / NJ, j = it % NJ, t = s / 2, h = s % 2;
                constexpr int nx = it + AHEAD;
                if constexpr (nx < NIT) ring[nx & 3] = xread(nx / NJ, J0 + nx % NJ);
                mma_step(acc[0][j], wf[t][h][0], ring[it & 3], (T*)nullptr);
                mma_step(acc[1][j], wf[t][h][1], ring[it & 3], (T*)nullptr);
                __builtin_amdgcn_sched_barrier(0);                   // the stream as written: one read, two MFMAs (no read clustering)
            });
            // epilogue: v = act(acc * sm + bm) as T, zero outside the image / outside the 10 x 18 mid region, into the mid tile
            f32x4 sm[2], bm[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sm[i] = *reinterpret_cast<const f32x4*>(cst + wc * 32 + i * 16 + 4 * fq);
                bm[i] = *reinterpret_cast<const f32x4*>(cst + 64 + wc * 32 + i * 16 + 4 * fq);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int L = (tb + J0 + j) * 16 + frow;
                const int my = L / PW, mx = L - my * PW;
                const bool ok = my < TH + 2 && mx < TW + 2 && (unsigned)(y0 + my) < (unsigned)a.H && (unsigned)(x0 + mx) < (unsigned)a.W;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t1 = acc[i][j][e] * sm[i][e] + bm[i][e];
                        v[e] = fmaxf(t1, t1 * slope_m);
                    }
                    uint2 o = Pack4<T>::pack(v);
                    if (!ok) o = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(smem + mbo + (i ? ck1 : ck0) + (J0 + j) * 16 * 128) = o;
                }
            }
        };
        pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        pass(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
    };

    // ---- role 1: conv2 + residual + second output of one tile: mid tile `buf` -> HBM ------------------------------------------
    // The epilogue works on the accumulators as they lie (a lane holds 4 consecutive channels of a pixel per MFMA tile): residual
    // from LDS (8 bytes per lane), two 8-byte stores per tile and output tensor -- no staging tile, no barrier.
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_raw ? a.out_raw : a.out_act), 0, a.out_raw ? tensor_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_act ? a.out_act : a.out_raw), 0, a.out_act ? tensor_bytes : 0u, 0x00020000);
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS address of smem (asm reads)
    auto conv2_tile = [&](int tile, int buf) {
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        int frow = lane & 15, fq = lane >> 4;
        asm volatile("" : "+v"(frow), "+v"(fq));
        const unsigned pbo = (unsigned)(OFF_MID + buf * MID_BYTES + (wp * 4 * PW + frow) * 128);
        unsigned a12[4][3];                                          // [(j + dy) & 3][dx]
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                a12[rr][dx] = pbo + (unsigned)(dx * 128) + (unsigned)((((((frow + dx) >> 1) + 2 * rr) & 7) ^ fq) << 4);
        auto xread = [&](int s, int j) {
            const int t = s >> 1, h = s & 1, dy = t / 3, dx = t % 3;
            return *reinterpret_cast<const f32x4*>(smem + (a12[(j + dy) & 3][dx] ^ (unsigned)(h << 6)) + (j + dy) * (PW * 128));
        };
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 18 half-steps x 4 pixel tiles, fragments read three items ahead (a ring of four)
        constexpr int NIT = 18 * 4, AHEAD = 3;
        f32x4 ring[4];
        static_for<AHEAD>([&](auto nc) {
            constexpr int it = decltype(nc)::value;
            ring[it & 3] = xread(it / 4, it % 4);
        });
        static_for<NIT>([&](auto nc) {
            constexpr int it = decltype(nc)::value;
            constexpr int s = it / 4, j = it % 4, t = s / 2, h = s % 2;
            constexpr int nx = it + AHEAD;
            if constexpr (nx < NIT) ring[nx & 3] = xread(nx / 4, nx % 4);
            mma_step(acc[0][j], wf[t][h][0], ring[it & 3], (T*)nullptr);
            mma_step(acc[1][j], wf[t][h][1], ring[it & 3], (T*)nullptr);
            __builtin_amdgcn_sched_barrier(0);
        });
        // epilogue: lane (frow, fq) holds channels wc*32 + i*16 + 4 fq .. +3 of pixel (tile row wp*4 + j, column frow)
        f32x4 s1[2], b1[2], s2[2], b2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = wc * 32 + i * 16 + 4 * fq;
            s1[i] = *reinterpret_cast<const f32x4*>(cst + 128 + c); b1[i] = *reinterpret_cast<const f32x4*>(cst + 192 + c);
            s2[i] = *reinterpret_cast<const f32x4*>(cst + 256 + c); b2[i] = *reinterpret_cast<const f32x4*>(cst + 320 + c);
        }
        const unsigned rbo = (unsigned)(OFF_RES + buf * RES_BYTES + (wp * 4 * 16 + frow) * 128 + (fq & 1) * 8);
        const unsigned rk0 = (unsigned)((wc * 4 + (fq >> 1)) ^ (frow & 7)) << 4, rk1 = (unsigned)((wc * 4 + 2 + (fq >> 1)) ^ (frow & 7)) << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = ty * TH + wp * 4 + j, x = tx * TW + frow;
            const unsigned obase = (y < a.H && x < a.W) ? (unsigned)((((img * a.H + y) * a.W + x) * 64 + wc * 32 + 4 * fq) * 2) : kOOB;
            // the residual of this pixel tile, as INLINE ASM: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it knows of
            // that may alias an LDS-DMA target (role 0 fills this buffer) -- here that would wait for the stores just issued
            u32x2 rres[2];
#ifdef PPN_B64_GLOBAL_RES
            rres[0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)obase, 0, 0));
            rres[1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrs, (int)(obase == kOOB ? kOOB : obase + 32u), 0, 0));
#elif defined(PPN_B64_PLAIN_RES)
            rres[0] = *reinterpret_cast<const u32x2*>(smem + rbo + rk0 + (unsigned)(j * 16 * 128));
            rres[1] = *reinterpret_cast<const u32x2*>(smem + rbo + rk1 + (unsigned)(j * 16 * 128));
#else
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(rres[0]), "=&v"(rres[1])
                         : "v"(lds_base + rbo + rk0 + (unsigned)(j * 16 * 128)), "v"(lds_base + rbo + rk1 + (unsigned)(j * 16 * 128))
                         : "memory");
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = acc[i][j][e] * s1[i][e] + b1[i][e];
                    v[e] = fmaxf(t1, t1 * slope1);
                }
                if (a.res) {
                    float r[8];
                    const uint4 r4 = make_uint4(rres[i].x, rres[i].y, 0u, 0u);
                    load8<T>(reinterpret_cast<const char*>(&r4), r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                }
                const unsigned off = obase == kOOB ? kOOB : obase + (unsigned)(i * 32);
                if (a.out_raw) {
                    const uint2 o = Pack4<T>::pack(v);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{o.x, o.y}, ors, (int)off, 0, 0);
                }
                if (a.out_act) {
                    float u[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t2 = v[e] * s2[i][e] + b2[i][e];
                        u[e] = fmaxf(t2, t2 * slope2);
                    }
                    const uint2 o = Pack4<T>::pack(u);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{o.x, o.y}, ars, (int)off, 0, 0);
                }
            }
        }
    };

    // ---- the pipeline: phase p = conv1 of tile p (role 0) beside conv2 of tile p - 1 (role 1); one barrier per phase --------
#ifdef PPN_CLOCK
    unsigned long long ck_work = 0, ck_dma = 0, ck_bar = 0, ck_issue = 0;
    const unsigned long long ck_begin = __builtin_amdgcn_s_memtime();
#define B64_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define B64_T(v) do { } while (0)
#endif
    if (role == 0) issue_patch(first, 0);
    for (int p = 0; p <= n; ++p) {
        // role 0's patch of tile p has landed (its only outstanding memory operations); everyone's LDS traffic of the previous
        // phase is complete: the mid tile p - 1 is written, mid tile p - 2 and patch p - 1 are no longer read
        B64_T(t0_);
        if (role == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        B64_T(t1_);
        lds_barrier();
        B64_T(t2_);
        if (role == 0) {
            if (p + 1 < n) issue_patch(first + (p + 1) * G, (p + 1) & 1);
            if (p < n && a.res) issue_res(first + p * G, p & 1);      // read by role 1 in phase p + 1
            B64_T(t3_);
            if (p < n) conv1_tile(first + p * G, p & 1);
#ifdef PPN_CLOCK
            ck_issue += t3_ - t2_; ck_work += __builtin_amdgcn_s_memtime() - t3_;
#endif
        } else if (p >= 1) {
            conv2_tile(first + (p - 1) * G, (p - 1) & 1);
#ifdef PPN_CLOCK
            ck_work += __builtin_amdgcn_s_memtime() - t2_;
#endif
        }
#ifdef PPN_CLOCK
        ck_dma += t1_ - t0_; ck_bar += t2_ - t1_;
#endif
    }
#ifdef PPN_CLOCK
    if (lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = ck_work; d[1] = ck_dma; d[2] = ck_bar; d[3] = ck_issue; d[4] = n; d[5] = __builtin_amdgcn_s_memtime() - ck_begin;
    }
#endif
}


// =====================================================================================================================================
// The FIRST block of layer3 as one launch (round 5): conv1 is a 3x3 STRIDE-2 convolution from 32 channels (the pre-activated stem
// output, 2H x 2W), and the block's shortcut is the 1x1 stride-2 projection + BN of the raw stem output (drn.py:53-54, 176-181) --
// which the fused stem now writes only at the pixels the projection reads (PPN_STEM_RAW_S2: a dense [B,H,W,32] tensor).  Same two
// roles and the same phase pipeline as block64_kernel; what differs:
//   tiles      4 x 16 output pixels (the 13 x 37-pixel input patch of an 8 x 16 tile would be 50 KB per buffer): mid tile 6 x 18 on
//              the row pitch 20 = 8 MFMA pixel tiles (4 per role-0 wave, one pass), conv2 2 row tiles per role-1 wave.
//   conv1      K = 9 taps x 32 channels: ONE 32-deep MFMA per tap, the 64 x 288 filter bank is 72 VGPRs per wave.  The input patch
//              lies in LDS as [row][column parity][column / 2] x 64 B, so that the stride-2 walk of a pixel tile is 16 consecutive
//              64-byte units and tap (dy, dx) is the constant unit shift 38 dy + 19 (dx & 1) + (dx >> 1): one per-lane base per
//              pixel tile + immediates address every fragment read (unswizzled: 2-way conflicts on a read stream that is far from
//              the LDS limit).
//   shortcut   role 0 also computes the projection of the tile's 64 pixels -- 2 x 2 more MFMAs per wave on operands fetched
//              straight from global memory at the head of the phase -- applies its BN, rounds to the 16-bit type and WRITES the
//              result where block64_kernel's LDS-DMA puts the residual rows: role 1 is the same code in both kernels.
// Arithmetic and K order are those of the three launches it replaces (downsample, conv1, conv2 + residual): bit-identical.
namespace s2 {
constexpr int TH = 4, TW = 16, PW = 20;
constexpr int MID_TILES = 8;                       // linear mid rows 0 .. 127 (valid: row < 6, column < 18)
constexpr int MID_BYTES = MID_TILES * 16 * 128;    // 16 384
constexpr int PR = 2 * (TH + 2) + 1, PC = 2 * (TW + 2) + 1;   // 13 x 37 input pixels
constexpr int PU = 19;                             // units of 64 B per column-parity plane of a patch row
constexpr int ROWU = 2 * PU;                       // 38 units per patch row
constexpr int IN_DMA = (PR * ROWU + 15) / 16;      // 31 wave-instructions of 16 pixels (64 B each)
constexpr int IN_BYTES = IN_DMA * 1024;
constexpr int RES_BYTES = TH * TW * 128;           // 8 192
constexpr int CST_BYTES = 8 * 64 * 4;              // [sm | bm | s1 | b1 | s2 | b2 | s_ds | b_ds][64]
constexpr int OFF_MID = 2 * IN_BYTES, OFF_RES = OFF_MID + 2 * MID_BYTES, OFF_CST = OFF_RES + 2 * RES_BYTES;
constexpr int LDS_BYTES = OFF_CST + CST_BYTES + 4096;   // + slack: conv1's discarded pixel tiles read past the last patch row
}  // namespace s2

struct BlkS2Args {
    const char* src;                 // relu(bn1(x)): NHWC [B][2H][2W][32]
    const char* xr;                  // x at the even pixels: NHWC [B][H][W][32]
    const char* w1;                  // packed [64][ld1], k = tap * 32 + ci
    const char* wd;                  // packed [64][ldd], k = ci
    const char* w2;                  // packed [64][576], k = tap * 64 + ci
    const float *sm, *bm, *sd, *bd, *s1, *b1, *s2, *b2;
    char* out_raw;
    char* out_act;
    int B, H, W, Hi, Wi, ld1, ldd, act_mid, act1, act2;
    int tiles_x, tiles_y, n_tiles;
    FastDiv div_tx, div_tpi;
};

template <typename T>
__global__ void __launch_bounds__(512, 2) block64s2_kernel(BlkS2Args a, unsigned in_bytes, unsigned xr_bytes, unsigned out_bytes) {
    // (local names shadow the stride-1 kernel's file-scope constants)
    constexpr int TH = s2::TH, TW = s2::TW, PW = s2::PW, MID_BYTES = s2::MID_BYTES, PR = s2::PR, PC = s2::PC, PU = s2::PU, ROWU = s2::ROWU;
    constexpr int IN_DMA = s2::IN_DMA, IN_BYTES = s2::IN_BYTES, RES_BYTES = s2::RES_BYTES, OFF_MID = s2::OFF_MID, OFF_RES = s2::OFF_RES;
    constexpr int OFF_CST = s2::OFF_CST;
    static_assert(sizeof(T) == 2, "16-bit modes only");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, rw = wave & 3;
    const int wc = rw >> 1, wp = rw & 1;
    float* cst = reinterpret_cast<float*>(smem + OFF_CST);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xr, 0, xr_bytes, 0x00020000);

    // ---- filter banks as MFMA A fragments: role 0 conv1 (9 taps x 32 channels) + the projection, role 1 conv2 ---------------
    f32x4 wf[9][2][2];               // role 1: [tap][half][channel tile]; role 0 uses [tap][0][channel tile] and wdf
    f32x4 wdf[2];
    {
        const int frow = lane & 15, fq = lane >> 4;
        if (role) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        wf[t][h][i] = *reinterpret_cast<const f32x4*>(a.w2 + ((size_t)(wc * 32 + i * 16 + frow) * 576 + t * 64 + h * 32 + fq * 8) * 2);
            wdf[0] = wdf[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wf[t][0][i] = *reinterpret_cast<const f32x4*>(a.w1 + ((size_t)(wc * 32 + i * 16 + frow) * a.ld1 + t * 32 + fq * 8) * 2);
                    wf[t][1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wdf[i] = *reinterpret_cast<const f32x4*>(a.wd + ((size_t)(wc * 32 + i * 16 + frow) * a.ldd + fq * 8) * 2);
        }
    }
    if (tid < 64) {
        cst[tid] = a.sm ? a.sm[tid] : 1.f;       cst[64 + tid] = a.bm ? a.bm[tid] : 0.f;
        cst[128 + tid] = a.s1 ? a.s1[tid] : 1.f; cst[192 + tid] = a.b1 ? a.b1[tid] : 0.f;
        cst[256 + tid] = a.s2 ? a.s2[tid] : 1.f; cst[320 + tid] = a.b2 ? a.b2[tid] : 0.f;
        cst[384 + tid] = a.sd ? a.sd[tid] : 1.f; cst[448 + tid] = a.bd ? a.bd[tid] : 0.f;
    }
    const float slope_m = a.act_mid == PPN_ACT_RELU ? 0.f : (a.act_mid == PPN_ACT_LRELU ? 0.1f : 1.f);
    const float slope1 = a.act1 == PPN_ACT_RELU ? 0.f : (a.act1 == PPN_ACT_LRELU ? 0.1f : 1.f);
    const float slope2 = a.act2 == PPN_ACT_RELU ? 0.f : (a.act2 == PPN_ACT_LRELU ? 0.1f : 1.f);

    const int G = gridDim.x, first = blockIdx.x;
    const int n = (a.n_tiles - first + G - 1) / G;
    auto tile_pos = [&](int tile, int& img, int& ty, int& tx) {
        img = fast_div(tile, a.div_tpi);
        const int rem = tile - img * (a.tiles_x * a.tiles_y);
        ty = fast_div(rem, a.div_tx);
        tx = rem - ty * a.tiles_x;
    };

    // ---- role 0: the 13 x 37-pixel input patch of a tile: unit U = row * 38 + parity * 19 + column / 2, 64 B each ----------------
    auto issue_patch = [&](int tile, int buf) {
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        const int y0 = 2 * (ty * TH - 1) - 1, x0 = 2 * (tx * TW - 1) - 1;
        for (int g = rw; g < IN_DMA; g += 4) {
            const int U = g * 16 + (lane >> 2);
            const int py = U / ROWU, r = U - py * ROWU;
            const int par = r >= PU ? 1 : 0, ci = r - par * PU;
            const int c = 2 * ci + par;
            const int y = y0 + py, x = x0 + c;
            const bool ok = py < PR && c < PC && (unsigned)y < (unsigned)a.Hi && (unsigned)x < (unsigned)a.Wi;
            const unsigned off = ok ? (unsigned)(((img * a.Hi + y) * a.Wi + x) * 64 + (lane & 3) * 16) : kOOB;
            bufload_lds16(xrs, smem + buf * IN_BYTES + g * 1024, off);
        }
    };

    // ---- role 0: conv1 (stride 2) + bn2 + ReLU -> mid tile; projection + BN -> residual rows ------------------------------------------
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    auto conv1_tile = [&](int tile, int buf) {
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        int frow = lane & 15, fq = lane >> 4;
        asm volatile("" : "+v"(frow), "+v"(fq));
        // the projection's operands: this wave's 2 x 16 output pixels, chunk fq of each 64-byte pixel, straight from global memory
        u32x4 xr[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int y = ty * TH + wp * 2 + jj, x = tx * TW + frow;
            const unsigned off = (y < a.H && x < a.W) ? (unsigned)(((img * a.H + y) * a.W + x) * 64 + fq * 16) : kOOB;
            xr[jj] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)off, 0, 0);
        }
        const int tb = wp * 4;                                       // pixel tiles tb .. tb + 3 of the 8
        unsigned ub[4];                                              // LDS byte offset of the lane's pixel (tap 0) per pixel tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int L = (tb + j) * 16 + frow;
            const int my = L / PW, mx = L - my * PW;
            ub[j] = (unsigned)(buf * IN_BYTES + (2 * my * ROWU + mx) * 64 + fq * 16);
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto xread = [&](int t, int j) {
            const int dy = t / 3, dx = t % 3;
            return *reinterpret_cast<const f32x4*>(smem + ub[j] + (dy * ROWU + (dx & 1) * PU + (dx >> 1)) * 64);
        };
        constexpr int NIT = 9 * 4, AHEAD = 3;
        f32x4 ring[4];
        static_for<AHEAD>([&](auto nc) {
            constexpr int it = decltype(nc)::value;
            ring[it & 3] = xread(it / 4, it % 4);
        });
        static_for<NIT>([&](auto nc) {
            constexpr int it = decltype(nc)::value;
            constexpr int t = it / 4, j = it % 4;
            constexpr int nx = it + AHEAD;
            if constexpr (nx < NIT) ring[nx & 3] = xread(nx / 4, nx % 4);
            mma_step(acc[0][j], wf[t][0][0], ring[it & 3], (T*)nullptr);
            mma_step(acc[1][j], wf[t][0][1], ring[it & 3], (T*)nullptr);
            __builtin_amdgcn_sched_barrier(0);
        });
        f32x4 sm[2], bm[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sm[i] = *reinterpret_cast<const f32x4*>(cst + wc * 32 + i * 16 + 4 * fq);
            bm[i] = *reinterpret_cast<const f32x4*>(cst + 64 + wc * 32 + i * 16 + 4 * fq);
        }
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const unsigned mbo = (unsigned)(OFF_MID + buf * MID_BYTES + (tb * 16 + frow) * 128 + (fq & 1) * 8);
        const unsigned ck0 = (unsigned)((wc * 4 + (fq >> 1)) ^ ((frow >> 1) & 7)) << 4;
        const unsigned ck1 = (unsigned)((wc * 4 + 2 + (fq >> 1)) ^ ((frow >> 1) & 7)) << 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int L = (tb + j) * 16 + frow;
            const int my = L / PW, mx = L - my * PW;
            const bool ok = my < TH + 2 && mx < TW + 2 && (unsigned)(y0 + my) < (unsigned)a.H && (unsigned)(x0 + mx) < (unsigned)a.W;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = acc[i][j][e] * sm[i][e] + bm[i][e];
                    v[e] = fmaxf(t1, t1 * slope_m);
                }
                uint2 o = Pack4<T>::pack(v);
                if (!ok) o = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(smem + mbo + (i ? ck1 : ck0) + j * 16 * 128) = o;
            }
        }
        // the shortcut: bn_ds(conv1x1(x at the even pixels)) of this wave's 2 x 16 pixels x 32 channels, rounded to T as the
        // stored residual tensor was, into the residual rows role 1 reads in the next phase
        f32x4 sd[2], bd[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sd[i] = *reinterpret_cast<const f32x4*>(cst + 384 + wc * 32 + i * 16 + 4 * fq);
            bd[i] = *reinterpret_cast<const f32x4*>(cst + 448 + wc * 32 + i * 16 + 4 * fq);
        }
        const unsigned rbo = (unsigned)(OFF_RES + buf * RES_BYTES + (wp * 2 * 16 + frow) * 128 + (fq & 1) * 8);
        const unsigned rk0 = (unsigned)((wc * 4 + (fq >> 1)) ^ (frow & 7)) << 4, rk1 = (unsigned)((wc * 4 + 2 + (fq >> 1)) ^ (frow & 7)) << 4;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 pa = f32x4{0.f, 0.f, 0.f, 0.f};
                mma_step(pa, wdf[i], __builtin_bit_cast(f32x4, xr[jj]), (T*)nullptr);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pa[e] * sd[i][e] + bd[i][e];
                *reinterpret_cast<uint2*>(smem + rbo + (i ? rk1 : rk0) + jj * 16 * 128) = Pack4<T>::pack(v);
            }
    };

    // ---- role 1: conv2 + residual (from LDS) + second output of one 4 x 16 tile -------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_raw ? a.out_raw : a.out_act), 0, a.out_raw ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_act ? a.out_act : a.out_raw), 0, a.out_act ? out_bytes : 0u, 0x00020000);
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto conv2_tile = [&](int tile, int buf) {
        int img, ty, tx;
        tile_pos(tile, img, ty, tx);
        int frow = lane & 15, fq = lane >> 4;
        asm volatile("" : "+v"(frow), "+v"(fq));
        const unsigned pbo = (unsigned)(OFF_MID + buf * MID_BYTES + (wp * 2 * PW + frow) * 128);
        unsigned a12[4][3];                                          // [(j + dy) & 3][dx]; row 20 (2 wp + j + dy) + frow + dx
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                a12[rr][dx] = pbo + (unsigned)(dx * 128) + (unsigned)((((((frow + dx) >> 1) + 2 * rr + 4 * wp) & 7) ^ fq) << 4);
        auto xread = [&](int s, int j) {
            const int t = s >> 1, h = s & 1, dy = t / 3, dx = t % 3;
            return *reinterpret_cast<const f32x4*>(smem + (a12[(j + dy) & 3][dx] ^ (unsigned)(h << 6)) + (j + dy) * (PW * 128));
        };
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NIT = 18 * 2, AHEAD = 3;
        f32x4 ring[4];
        static_for<AHEAD>([&](auto nc) {
            constexpr int it = decltype(nc)::value;
            ring[it & 3] = xread(it / 2, it % 2);
        });
        static_for<NIT>([&](auto nc) {
            constexpr int it = decltype(nc)::value;
            constexpr int s = it / 2, j = it % 2, t = s / 2, h = s % 2;
            constexpr int nx = it + AHEAD;
            if constexpr (nx < NIT) ring[nx & 3] = xread(nx / 2, nx % 2);
            mma_step(acc[0][j], wf[t][h][0], ring[it & 3], (T*)nullptr);
            mma_step(acc[1][j], wf[t][h][1], ring[it & 3], (T*)nullptr);
            __builtin_amdgcn_sched_barrier(0);
        });
        f32x4 s1[2], b1[2], s2v[2], b2v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = wc * 32 + i * 16 + 4 * fq;
            s1[i] = *reinterpret_cast<const f32x4*>(cst + 128 + c); b1[i] = *reinterpret_cast<const f32x4*>(cst + 192 + c);
            s2v[i] = *reinterpret_cast<const f32x4*>(cst + 256 + c); b2v[i] = *reinterpret_cast<const f32x4*>(cst + 320 + c);
        }
        const unsigned rbo = (unsigned)(OFF_RES + buf * RES_BYTES + (wp * 2 * 16 + frow) * 128 + (fq & 1) * 8);
        const unsigned rk0 = (unsigned)((wc * 4 + (fq >> 1)) ^ (frow & 7)) << 4, rk1 = (unsigned)((wc * 4 + 2 + (fq >> 1)) ^ (frow & 7)) << 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = ty * TH + wp * 2 + j, x = tx * TW + frow;
            const unsigned obase = (y < a.H && x < a.W) ? (unsigned)((((img * a.H + y) * a.W + x) * 64 + wc * 32 + 4 * fq) * 2) : kOOB;
            u32x2 rres[2];
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(rres[0]), "=&v"(rres[1])
                         : "v"(lds_base + rbo + rk0 + (unsigned)(j * 16 * 128)), "v"(lds_base + rbo + rk1 + (unsigned)(j * 16 * 128))
                         : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = acc[i][j][e] * s1[i][e] + b1[i][e];
                    v[e] = fmaxf(t1, t1 * slope1);
                }
                {
                    float r[8];
                    const uint4 r4 = make_uint4(rres[i].x, rres[i].y, 0u, 0u);
                    load8<T>(reinterpret_cast<const char*>(&r4), r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                }
                const unsigned off = obase == kOOB ? kOOB : obase + (unsigned)(i * 32);
                if (a.out_raw) {
                    const uint2 o = Pack4<T>::pack(v);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{o.x, o.y}, ors, (int)off, 0, 0);
                }
                if (a.out_act) {
                    float u[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t2 = v[e] * s2v[i][e] + b2v[i][e];
                        u[e] = fmaxf(t2, t2 * slope2);
                    }
                    const uint2 o = Pack4<T>::pack(u);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{o.x, o.y}, ars, (int)off, 0, 0);
                }
            }
        }
    };

    if (role == 0) issue_patch(first, 0);
    for (int p = 0; p <= n; ++p) {
        if (role == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (role == 0) {
            if (p + 1 < n) issue_patch(first + (p + 1) * G, (p + 1) & 1);
            if (p < n) conv1_tile(first + p * G, p & 1);
        } else if (p >= 1) {
            conv2_tile(first + (p - 1) * G, (p - 1) & 1);
        }
    }
}

}  // namespace

namespace ppn {

#ifdef PPN_CLOCK
static unsigned long long* g_block64_dbg = nullptr;   // tools/clock_block64.py: [workgroup][wave][8] u64
#endif
static int g_block64_on = -1;            // -1: not decided yet (PPN_BLOCK64=0 in the environment disables it)

bool block64_enabled() {
    if (g_block64_on < 0) g_block64_on = (getenv("PPN_BLOCK64") && atoi(getenv("PPN_BLOCK64")) == 0) ? 0 : 1;
    return g_block64_on != 0;
}

static int block64s2_launch(const ppn_block_desc* d, hipStream_t st, const char** kname) {
    if (d->dtype != PPN_BF16 && d->dtype != PPN_F16) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: 16-bit modes only");
    if (d->channels != 64) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: 64 channels (got %d)", d->channels);
    if (d->batch < 1 || d->h < 1 || d->w < 1 || d->in_h < 1 || d->in_w < 1 || d->h != (d->in_h - 1) / 2 + 1 || d->w != (d->in_w - 1) / 2 + 1)
        return ppn::fail(PPN_E_INVALID, "ppn_basicblock64 (stride 2): h / w must be (in_h - 1) / 2 + 1, (in_w - 1) / 2 + 1");
    if (!d->src || !d->weight1 || !d->weight2 || !d->proj_src || !d->proj_weight || (!d->out_raw && !d->out_act) || d->residual)
        return ppn::fail(PPN_E_INVALID, "ppn_basicblock64 (stride 2): NULL src / weight / projection / output, or a residual");
    if (d->w1_ld < 288 || d->proj_ld < 32 || (d->w1_ld & 7) || (d->proj_ld & 7))
        return ppn::fail(PPN_E_INVALID, "ppn_basicblock64 (stride 2): w1_ld >= 288, proj_ld >= 32, multiples of 8");
    for (int act : {d->act_mid, d->act1, d->act2})
        if (act != PPN_ACT_NONE && act != PPN_ACT_RELU && act != PPN_ACT_LRELU)
            return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: activations none / ReLU / LeakyReLU");
    const size_t in_bytes = (size_t)d->batch * d->in_h * d->in_w * 32 * 2, xr_bytes = (size_t)d->batch * d->h * d->w * 32 * 2;
    const size_t out_bytes = (size_t)d->batch * d->h * d->w * 64 * 2;
    if (in_bytes >= 0x7fffff00ull || out_bytes >= 0x7fffff00ull)
        return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for the buffer-addressed kernel");
    BlkS2Args a;
    a.src = static_cast<const char*>(d->src); a.xr = static_cast<const char*>(d->proj_src);
    a.w1 = static_cast<const char*>(d->weight1); a.wd = static_cast<const char*>(d->proj_weight);
    a.w2 = static_cast<const char*>(d->weight2);
    a.sm = d->scale_mid; a.bm = d->shift_mid; a.sd = d->proj_scale; a.bd = d->proj_shift;
    a.s1 = d->scale1; a.b1 = d->shift1; a.s2 = d->scale2; a.b2 = d->shift2;
    a.out_raw = static_cast<char*>(d->out_raw); a.out_act = static_cast<char*>(d->out_act);
    a.B = d->batch; a.H = d->h; a.W = d->w; a.Hi = d->in_h; a.Wi = d->in_w; a.ld1 = d->w1_ld; a.ldd = d->proj_ld;
    a.act_mid = d->act_mid; a.act1 = d->act1; a.act2 = d->act2;
    a.tiles_x = (d->w + s2::TW - 1) / s2::TW; a.tiles_y = (d->h + s2::TH - 1) / s2::TH;
    const long long nt = (long long)d->batch * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffffLL) return ppn::fail(PPN_E_UNSUPPORTED, "too many tiles");
    a.n_tiles = (int)nt;
    a.div_tx = make_fastdiv((unsigned)a.tiles_x);
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        PPN_HIP_CHECK(hipGetDevice(&dev));
        PPN_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int grid = (int)std::min<long long>(nt, n_cu);
    if (d->dtype == PPN_F16) {
        if (kname) *kname = "block64s2_kernel<_Float16>";
        static int set16 = 0;
        PPN_LDS_ONCE(set16, reinterpret_cast<const void*>(block64s2_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, s2::LDS_BYTES);
        hipLaunchKernelGGL(block64s2_kernel<_Float16>, dim3(grid), dim3(512), s2::LDS_BYTES, st, a, (unsigned)in_bytes, (unsigned)xr_bytes, (unsigned)out_bytes);
    } else {
        if (kname) *kname = "block64s2_kernel<__bf16>";
        static int setb = 0;
        PPN_LDS_ONCE(setb, reinterpret_cast<const void*>(block64s2_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, s2::LDS_BYTES);
        hipLaunchKernelGGL(block64s2_kernel<__bf16>, dim3(grid), dim3(512), s2::LDS_BYTES, st, a, (unsigned)in_bytes, (unsigned)xr_bytes, (unsigned)out_bytes);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

int block64_launch(const ppn_block_desc* d, hipStream_t st, const char** kname) {
    if (!d) return ppn::fail(PPN_E_INVALID, "block desc is NULL");
    if (d->stride == 2) return block64s2_launch(d, st, kname);
    if (d->stride != 0 && d->stride != 1) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: stride 1 or 2");
    if (d->dtype != PPN_BF16 && d->dtype != PPN_F16) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: 16-bit modes only");
    if (d->channels != 64) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: 64 channels (got %d)", d->channels);
    if (d->batch < 1 || d->h < 1 || d->w < 1) return ppn::fail(PPN_E_INVALID, "ppn_basicblock64: bad geometry");
    if (!d->src || !d->weight1 || !d->weight2 || (!d->out_raw && !d->out_act))
        return ppn::fail(PPN_E_INVALID, "ppn_basicblock64: NULL src/weight/output");
    for (int act : {d->act_mid, d->act1, d->act2})
        if (act != PPN_ACT_NONE && act != PPN_ACT_RELU && act != PPN_ACT_LRELU)
            return ppn::fail(PPN_E_UNSUPPORTED, "ppn_basicblock64: activations none / ReLU / LeakyReLU");
    const size_t bytes = (size_t)d->batch * d->h * d->w * 64 * 2;
    if (bytes >= 0x7fffff00ull) return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for the buffer-addressed kernel");
    BlkArgs a;
    a.src = static_cast<const char*>(d->src);
    a.res = static_cast<const char*>(d->residual);
    a.w1 = static_cast<const char*>(d->weight1);
    a.w2 = static_cast<const char*>(d->weight2);
    a.sm = d->scale_mid; a.bm = d->shift_mid;
    a.s1 = d->scale1; a.b1 = d->shift1; a.s2 = d->scale2; a.b2 = d->shift2;
    a.out_raw = static_cast<char*>(d->out_raw);
    a.out_act = static_cast<char*>(d->out_act);
    a.B = d->batch; a.H = d->h; a.W = d->w; a.act_mid = d->act_mid; a.act1 = d->act1; a.act2 = d->act2;
    a.tiles_x = (d->w + TW - 1) / TW; a.tiles_y = (d->h + TH - 1) / TH;
    const long long nt = (long long)d->batch * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffffLL) return ppn::fail(PPN_E_UNSUPPORTED, "too many tiles");
    a.n_tiles = (int)nt;
    a.div_tx = make_fastdiv((unsigned)a.tiles_x);
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
#ifdef PPN_CLOCK
    a.dbg = g_block64_dbg;
#else
    a.dbg = nullptr;
#endif
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        PPN_HIP_CHECK(hipGetDevice(&dev));
        PPN_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int grid = (int)std::min<long long>(nt, n_cu);            // persistent: one workgroup per CU
    if (d->dtype == PPN_F16) {
        if (kname) *kname = "block64_kernel<_Float16>";
        static int set16 = 0;
        PPN_LDS_ONCE(set16, reinterpret_cast<const void*>(block64_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipLaunchKernelGGL(block64_kernel<_Float16>, dim3(grid), dim3(512), LDS_BYTES, st, a, (unsigned)bytes);
    } else {
        if (kname) *kname = "block64_kernel<__bf16>";
        static int setb = 0;
        PPN_LDS_ONCE(setb, reinterpret_cast<const void*>(block64_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipLaunchKernelGGL(block64_kernel<__bf16>, dim3(grid), dim3(512), LDS_BYTES, st, a, (unsigned)bytes);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // namespace ppn

#ifdef PPN_CLOCK
extern "C" int ppn_block64_set_debug(void* p) { ppn::g_block64_dbg = static_cast<unsigned long long*>(p); return PPN_OK; }
#endif
extern "C" int ppn_basicblock64_fused(const ppn_block_desc* d, void* stream) {
    return ppn::block64_launch(d, static_cast<hipStream_t>(stream), nullptr);
}
