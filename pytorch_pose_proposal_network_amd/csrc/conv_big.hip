// Large-tile variant of the fused implicit-GEMM convolution (see conv.hip for the GEMM view and the
// epilogue algebra).  Used for every layer whose Cin is a multiple of the K step and Cout >= 128, i.e.
// 93 % of the FLOPs of DRN-D-22.
//
// Why a second kernel: on gfx950 a CU's global->LDS fill path moves ~64 B/clk while its four SIMDs retire
// 4 x 1024 bf16 FLOP/clk, so a 128x128x64 tile (64 FLOP per staged byte) is fill-bound at ~1/3 of the MFMA
// peak, and two waves per SIMD running the same barrier-paced program contend for the VALU/MFMA issue in
// lockstep.  This kernel therefore gives every SIMD exactly ONE wave with the whole 512-register file:
//   workgroup  256 threads = 4 waves as 2 (channels) x 2 (pixels), one workgroup per CU
//   tile       BP in {128,192,256} pixels x BC in {128,256} channels x 64 (bf16) / 32 (f32) deep
//   wave       (BC/2) x (BP/2) outputs = up to 8x8 MFMA 16x16 tiles (256 accumulator registers)
//   LDS        2 stages x (BP+BC) x 128 B (<= 128 KiB), rows XOR-swizzled through the SOURCE address
//   loads      buffer_load ... lds (LDS-DMA) through raw buffer descriptors: per-lane 32-bit offsets, the
//              K-step offset of the weight stream in SGPRs, and padded taps expressed as an OUT-OF-RANGE
//              offset, which the hardware range check turns into zeros in LDS (tools/probes/buf_lds_oob.hip)
// BP is picked per layer so that the number of workgroups is close to a multiple of the 256 CUs.
// K loop: two named fragment sets; the MFMAs of one set cover the LDS reads of the other and the DMA issue
// of the next stage, so the matrix pipe restarts immediately after the per-step barrier.
// Epilogue through LDS in 64-pixel (NHWC) / 64-channel (NCHW head) chunks: 16-byte coalesced stores.  The head
// conv can additionally (or instead) run the decode's limb arg-max in its epilogue (ppn_conv_desc.argmax_keys).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

using namespace ppnconv;

// the tiles whose single-output 16-bit epilogue exists (the whole tile as 16-bit rows in the staging LDS) AND that carry the
// BatchNorm-statistics instantiation: kFastFits of the kernel, restated for the host
constexpr bool stats_tile_ok(int bp, int bc) {
    return bc >= 128 && (size_t)bp * (size_t)(bc * 2 + 16) <= 2 * (size_t)(bp + bc) * 128;
}

// Build-time diagnostics (tools/build_variant.py NAME conv_big.hip -DPPN_DIAG=n): TIMING ONLY, results are wrong.
//   1 = no wait for the DMA, 2 = no DMA in the K loop, 3 = 32x32x16 MFMAs (half the MFMA issue slots) on the same reads,
//   4 = 2 and 3 together, 5 = 2 without the per-step barrier, 6 = 2 without the LDS fragment reads,
//   7 = no global stores in the NHWC epilogue, 8 = no arg-max pass in the head epilogue, 9 = the pass without its atomics
#ifndef PPN_DIAG
#define PPN_DIAG 0
#endif
constexpr unsigned kOOB = 0x80000000u;   // byte offset beyond any tensor this kernel accepts (< 2 GiB)

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <typename T>
__device__ __forceinline__ void load4(const char* p, float* v);
template <>
__device__ __forceinline__ void load4<float>(const char* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <>
__device__ __forceinline__ void load4<__bf16>(const char* p, float* v) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
template <>
__device__ __forceinline__ void load4<_Float16>(const char* p, float* v) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    const f16x4 a = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
template <typename T>
__device__ __forceinline__ void store4(char* p, const float* v);
template <>
__device__ __forceinline__ void store4<float>(char* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<__bf16>(char* p, const float* v) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = o;
}

template <>
__device__ __forceinline__ void store4<_Float16>(char* p, const float* v) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
    f16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (_Float16)clamp_f16(v[i]);
    *reinterpret_cast<f16x4*>(p) = o;
}

__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff,
                                              unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)lds_wave_base, 16, voff,
                                             soff, 0, 0);
}

// NW waves per workgroup as 2 (channels) x NW/2 (pixels): NW = 4 -> one wave per SIMD with 8x8 MFMA tiles,
// NW = 8 -> two waves per SIMD (each hides the other's LDS-DMA issue stalls) with 8 x TP<=4 tiles.
// SC: the launch carries a fused 1x1 projection shortcut (second activation source, ConvKArgs.src2); compiled out
// of the plain instantiation -- its extra loader state costs the main loop a few per cent.
// X3 (PPN_F16X3, T = _Float16): split-precision operands.  Activation tensors hold every value as an IEEE-half pair,
// NHWC [pixel][hi(C) | lo'(C)] with hi = half(v), lo' = half((v - hi) * 2^11); the packed weight rows hold, per 64-channel
// slab, three copies [half(ws) | half(ws - half(ws)) | half(ws * 2^-11)] of the layer's weights ws = w * 2^s (s chosen on
// the host so that max |ws| <= 32768; 2^-s is folded into scale1).  The K loop walks three virtual slabs per real slab --
// (a_hi, w_hi), (a_hi, w_lo), (a_lo', w_hi * 2^-11) -- i.e. acc += a_hi w_hi + a_hi w_lo + a_lo w_hi in f32: products of
// 22-bit operands minus the lo x lo term (2^-22 relative), at a third of the f16 MFMA rate = ~5x the exact-f32 MFMA rate.
// Same loop, same staging: only the slab bookkeeping of advance() and the epilogue's loads / stores differ.
// ST: the single-output 16-bit epilogue additionally folds train-mode BatchNorm partial sums per pixel tile and channel
// (ppn_conv_desc.stats_mode); its own instantiations, so the inference kernels are the code they were.
// (the 192 x 128 tile runs two workgroups per CU = four waves per SIMD at 126 VGPRs; its ST twin came out at 129, i.e. ONE
// workgroup per CU, so that instantiation states the four waves it needs)
template <typename T, int BP, int BC, int NW, bool SC, bool X3 = false, bool ST = false>
__global__ void __launch_bounds__(64 * NW, (ST && BP == 192 && BC == 128) ? 4 : NW / 4)
conv_igemm_big_kernel(ConvKArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    static_assert(!X3 || (std::is_same<T, _Float16>::value && !SC), "X3 is the split-f16 mode without a fused shortcut");
    static_assert(!ST || (sizeof(T) == 2 && !SC && !X3 && NW == 8 && BC >= 128), "ST: plain 16-bit launches of the 8-wave tiles");
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int ES = sizeof(T);
    constexpr int NT = 64 * NW;
    // Wave grid WC (channels) x WP (pixels): 2 x NW/2 wherever the pixel tile splits into NW/2 multiples of 16; else (the
    // 144-pixel tile: 9 x 16) all NW waves side by side along the channels, each owning every pixel of the tile.
    constexpr int WC = (BP % (16 * (NW / 2)) == 0) ? 2 : NW;
    constexpr int WP = NW / WC;                                     // waves along the pixel dimension
    // load instructions per thread and K step: an activation piece is 8 rows x 128 B; a tile whose row count is not a
    // multiple of 8 * NW (144 = 18 pieces over 8 waves) gives its last round of pieces to the first waves only (xpiece())
    constexpr int NXI = (BP + 8 * NW - 1) / (8 * NW), NWI = BC / (8 * NW);
    constexpr bool XRAGGED = BP % (8 * NW) != 0;
    constexpr int TP = BP / WP / 16, TC = BC / WC / 16;            // 16x16 tiles per wave
    constexpr int STAGE = (BP + BC) * 128;
    static_assert(BP % 8 == 0 && BC % (8 * NW) == 0 && (BP / WP) % 16 == 0 && BC % (16 * WC) == 0, "tile shape");
    static_assert(TP <= 9 && TC <= 8 && TC * TP % 2 == 0, "accumulators must fit the register file");

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PPN_CLOCK
    const unsigned long long ck_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long rk_entry = __builtin_amdgcn_s_memrealtime();   // absolute 100 MHz time (-DPPN_CLOCK=2: reported)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;

    int ptile, ctile;
    {
        const int nb = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, loc = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        ptile = fast_div(logical, a.div_nct);
        ctile = logical - ptile * a.n_ctiles;
    }
    const int m0 = a.m_base + ptile * BP, c0 = ctile * BC;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, wgt_bytes, 0x00020000);
    // second activation source: the block input of a fused 1x1 projection shortcut (zero records when unused)
    const __amdgpu_buffer_rsrc_t xrs2 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(SC ? a.src2 : a.src), 0, SC ? a.src2_bytes : 0u, 0x00020000);

    // ---- per-lane loader state: 32-bit byte offsets ----------------------------------------------
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2));
    const int ntaps = a.ks * a.ks;
    int xbase[NXI];          // byte offset of (pixel row, tap 0, ci 0, this lane's chunk); may be negative
    int xbase2[NXI];         // same for the shortcut source (1x1, stride2, no padding)
    unsigned xmask[NXI];     // bit t: tap t in bounds;  bit 31: the output pixel itself exists (m < M)
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int row = (j * NW + wave) * 8 + lrow;
        const int m = m0 + row;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int b = fast_div(mm, a.div_howo), rem = mm - b * a.HoWo;
        const int oy = fast_div(rem, a.div_wo), ox = rem - oy * a.Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        xbase[j] = (((b * a.H + iy0) * a.W + ix0) * a.Cin + chunk * EPC) * ES;
        // tap t = dy * ks + dx is in bounds iff its row and its column are: ks column bits, replicated per valid row
        // (no per-tap division: this runs once per workgroup but on the critical path of the first DMA)
        unsigned mk = 0, cx = 0;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)                               // ksize <= 5 (checked by ppn_conv2d_fused)
            cx |= (dx < a.ks && (unsigned)(ix0 + dx * a.dil) < (unsigned)a.W) ? (1u << dx) : 0u;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
            mk |= (dy < a.ks && (unsigned)(iy0 + dy * a.dil) < (unsigned)a.H) ? (cx << (dy * a.ks)) : 0u;
        mk = vm ? mk : 0u;
        xmask[j] = mk | (vm ? 0x80000000u : 0u);
        if constexpr (SC)
            xbase2[j] = (((b * a.H2 + oy * a.stride2) * a.W2 + ox * a.stride2) * a.Cin2 + chunk * EPC) * ES;
        else
            xbase2[j] = 0;
    }
    unsigned woff[NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = (j * NW + wave) * 8 + lrow;
        woff[j] = (unsigned)(((size_t)(c0 + row) * a.Ktot + chunk * EPC) * ES);
    }

    const int nsteps = a.Ktot / BK;
    // Wave-uniform K iteration state, advanced incrementally so the loop keeps only a handful of SGPRs live
    // (depth order 1 of ppn_conv_tiling: taps innermost, then the next BK-channel slab).
    const int dx_bytes = a.dil * a.Cin * ES;                        // one tap to the right
    const int dy_bytes = a.dil * a.W * a.Cin * ES - a.ks * dx_bytes; // next tap row, back to dx = 0
    const int ksz = a.ks;
    int u_tap = 0, u_dx = 0, u_slab = 0, tapoff = 0, u_ph = 0;
    unsigned tapbit = 1u;                                           // 0 once past the last K step
    unsigned ksoff = 0;
    bool live = true;
    bool phase2 = a.nsteps_main == 0;                             // (never true at start: nsteps_main >= 1)
    int ld_step = 0;
    auto advance = [&]() {
        ++ld_step;
        live = ld_step < nsteps;
        ksoff = live ? ksoff + BK * ES : 0u;
        const bool now2 = SC && ld_step >= a.nsteps_main;         // past the main convolution: shortcut slabs
        if (!now2) {
            ++u_tap; ++u_dx;
            tapoff += dx_bytes;
            const bool wrapx = u_dx == ksz;
            u_dx = wrapx ? 0 : u_dx;
            tapoff += wrapx ? dy_bytes : 0;
            const bool wrapt = u_tap == ntaps;
            u_tap = wrapt ? 0 : u_tap;
            if constexpr (X3) {
                // three virtual slabs per real one: hi, hi, lo' (a.lo_off bytes further in the pixel)
                u_ph += wrapt ? 1 : 0;
                const bool wrapp = u_ph == 3;
                u_ph = wrapp ? 0 : u_ph;
                u_slab += wrapp ? BK * ES : 0;
                tapoff = wrapt ? u_slab + (u_ph == 2 ? a.lo_off : 0) : tapoff;
            } else {
                u_slab += wrapt ? BK * ES : 0;
                tapoff = wrapt ? u_slab : tapoff;
            }
            tapbit = live ? (1u << u_tap) : 0u;
        } else {
            tapoff = phase2 ? tapoff + BK * ES : 0;               // plain channel slabs of the second source
            tapbit = live ? 0x80000000u : 0u;                     // valid wherever the output pixel exists
        }
        phase2 = now2;
    };
    // one LDS-DMA instruction of the stage being loaded: g < NXI activation rows, else weight rows.
    // Past the last K step every offset is out of range (zero fill, no memory traffic); the SGPR offset is
    // not part of the hardware range check, so "dead" goes through the VGPR offset.
    auto issue_one = [&](auto gc, int buf) {
        constexpr int g = decltype(gc)::value;
        char* xs = smem + buf * STAGE;
        if constexpr (g < NXI) {
            if constexpr (XRAGGED && g == NXI - 1)
                if ((g * NW + wave) * 8 >= BP) return;               // wave-uniform: this piece lies past the tile
            if (!SC || !phase2) {
                const unsigned voff = (xmask[g] & tapbit) ? (unsigned)(xbase[g] + tapoff) : kOOB;
                bufload_lds16(xrs, xs + (g * NW + wave) * 1024, voff, 0);
            } else {
                const unsigned voff = (xmask[g] & tapbit) ? (unsigned)(xbase2[g] + tapoff) : kOOB;
                bufload_lds16(xrs2, xs + (g * NW + wave) * 1024, voff, 0);
            }
        } else {
            constexpr int j = g - NXI;
            bufload_lds16(wrs, xs + BP * 128 + (j * NW + wave) * 1024, live ? woff[j] : kOOB, ksoff);
        }
    };
    constexpr int NL = NXI + NWI;                                    // LDS-DMA instructions per stage and thread

    const int frow = lane & 15, fq = lane >> 4;
    const int fswz = (frow >> 1) & 7;
    int foff[2];
    foff[0] = frow * 128 + (((0 + fq) ^ fswz) << 4);
    foff[1] = frow * 128 + (((4 + fq) ^ fswz) << 4);
    const int x_tile_off = wp * (BP / WP) * 128;
    const int w_tile_off = BP * 128 + wc * (BC / WC) * 128;

    f32x4 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Two named fragment sets (A, B): while the MFMAs of one set issue, the other set is being read.
    f32x4 wA[TC], xA[TP], wB[TC], xB[TP];
    constexpr int NRD = TC + TP;                                     // ds_read_b128 per fragment set
    auto read_one = [&](auto rc, f32x4 (&wf)[TC], f32x4 (&xf)[TP], int buf, int ks) {
        constexpr int r = decltype(rc)::value;
#if PPN_DIAG == 6
        if (buf >= 0 && ks >= 0 && nsteps > 0) { asm volatile("" : "+v"(xf[0]), "+v"(wf[0])); return; }
#endif
        if constexpr (r < TP)
            xf[r] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + x_tile_off + foff[ks] + r * 16 * 128);
        else
            wf[r - TP] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + w_tile_off + foff[ks] + (r - TP) * 16 * 128);
    };
    // MFMA group: GS consecutive output tiles of the wave (flat index = i*TP + j); 4 wherever the tile count allows
    constexpr int GS = (TC * TP % 4 == 0) ? 4 : ((TC * TP % 3 == 0) ? 3 : 2);
    constexpr int NG = TC * TP / GS;
#if PPN_DIAG == 3 || PPN_DIAG == 4
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc32[TC * TP / 4];
#pragma unroll
    for (int i = 0; i < TC * TP / 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
#endif
    auto mma_group = [&](auto gc, const f32x4 (&wf)[TC], const f32x4 (&xf)[TP]) {
        constexpr int g = decltype(gc)::value;
#if PPN_DIAG == 3 || PPN_DIAG == 4
        if constexpr (std::is_same<T, __bf16>::value) {
            constexpr int i0 = g * 4, i1 = g * 4 + 3;
            acc32[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[i0 / TP]),
                                                               __builtin_bit_cast(bf16x8, xf[i0 % TP]), acc32[g], 0, 0, 0);
            acc32[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[i1 / TP]),
                                                               __builtin_bit_cast(bf16x8, xf[i1 % TP]), acc32[g], 0, 0, 0);
            return;
        }
#endif
        static_for<GS>([&](auto tc) {
            constexpr int idx = g * GS + decltype(tc)::value;
            mma_step(acc[idx / TP][idx % TP], wf[idx / TP], xf[idx % TP], (T*)nullptr);
        });
    };
    constexpr int RPG = (NRD + NG - 1) / NG;                         // LDS reads per MFMA group
    constexpr int LPG = (NL + NG - 1) / NG;                          // DMA issues per MFMA group

    // ---- prologue: stage 0 -> LDS, both fragment sets of stage 0, stage 1 in flight --------------
#ifdef PPN_CLOCK
    const unsigned long long ck_state = __builtin_amdgcn_s_memtime();
#endif
    // Both stages are requested before the first wait, so their (cold) fill latencies overlap; LDS-DMA completes
    // in issue order, so "all but the NL youngest" means stage 0 has landed.
    static_for<NL>([&](auto gc) { issue_one(gc, 0); });
    advance();
    static_for<NL>([&](auto gc) { issue_one(gc, 1); });
    advance();
    if constexpr (XRAGGED) {
        // waves whose last activation piece lies past the tile issued one DMA fewer per stage
        if ((NXI - 1) * NW * 8 + wave * 8 < BP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL - 1) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    }
    __syncthreads();
#ifdef PPN_CLOCK
    const unsigned long long ck_bar = __builtin_amdgcn_s_memtime();
#endif
    static_for<NRD>([&](auto rc) { read_one(rc, wA, xA, 0, 0); });
    static_for<NRD>([&](auto rc) { read_one(rc, wB, xB, 0, 1); });
    static_for<NG>([&](auto gc) { mma_group(gc, wA, xA); });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- steady state -----------------------------------------------------------------------------
    // entering step s: set B holds (s-1, second half); stage s is complete in buffer s&1; the other buffer
    // is free.  The MFMAs of set B start right after the barrier; between groups of 4 MFMAs the wave issues
    // the DMA of stage s+1 and the LDS reads of set A (order pinned with sched_barrier so neither clusters);
    // then the MFMAs of set A cover the reads of the new set B and the scalar bookkeeping of the next step.
#ifdef PPN_STAMP
    unsigned long long st_b1 = 0, st_b2 = 0, st_wait = 0, st_bar = 0, tq0, tq1;
#define PPN_T(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PPN_T(var) do { } while (0)
#endif
#ifdef PPN_CLOCK
    // in-kernel clock: shader cycles (s_memtime) over the 100 MHz constant clock (s_memrealtime) around the K loop
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rk0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int s = 1; s < nsteps; ++s) {
        const int buf = s & 1;
#ifdef PPN_STAMP
        PPN_T(tq0);
#endif
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            mma_group(gc, wB, xB);
#if PPN_DIAG != 2 && PPN_DIAG != 4 && PPN_DIAG != 5 && PPN_DIAG != 6
            static_for<LPG>([&](auto lc) {
                constexpr int l = g * LPG + decltype(lc)::value;
                if constexpr (l < NL) issue_one(std::integral_constant<int, l>{}, buf ^ 1);
            });
#endif
            static_for<RPG>([&](auto rc) {
                constexpr int r = g * RPG + decltype(rc)::value;
                if constexpr (r < NRD) read_one(std::integral_constant<int, r>{}, wA, xA, buf, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        advance();
#ifdef PPN_STAMP
        PPN_T(tq1); st_b1 += tq1 - tq0;
#endif
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            mma_group(gc, wA, xA);
            static_for<RPG>([&](auto rc) {
                constexpr int r = g * RPG + decltype(rc)::value;
                if constexpr (r < NRD) read_one(std::integral_constant<int, r>{}, wB, xB, buf, 1);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef PPN_STAMP
        PPN_T(tq0); st_b2 += tq0 - tq1;
#endif
#if PPN_DIAG == 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifdef PPN_STAMP
        PPN_T(tq1); st_wait += tq1 - tq0;
#endif
#if PPN_DIAG != 5
        __syncthreads();
#endif
#ifdef PPN_STAMP
        PPN_T(tq0); st_bar += tq0 - tq1;
#endif
    }
#ifdef PPN_CLOCK
    {
        const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), rk1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.scale2 == nullptr && a.shift2 != nullptr) {   // diagnostic channel: shift2 = u64 buffer
            unsigned long long* dbg = (unsigned long long*)a.shift2 + ((size_t)blockIdx.x * NW + wave) * 8;
            dbg[0] = ck1 - ck0; dbg[1] = rk1 - rk0; dbg[2] = ck0 - ck_entry; dbg[3] = ck1; dbg[5] = ck_state - ck_entry; dbg[6] = ck_bar - ck_entry;
        }
    }
#endif
#ifdef PPN_STAMP
    unsigned long long t_epi0;
    PPN_T(t_epi0);
    if (lane == 0 && a.scale2 == nullptr && a.shift2 != nullptr) {   // diagnostic channel: shift2 = u64 buffer
        unsigned long long* dbg = (unsigned long long*)a.shift2 + ((size_t)blockIdx.x * NW + wave) * 8;
        dbg[0] = st_b1; dbg[1] = st_b2; dbg[2] = st_wait; dbg[3] = st_bar;
    }
#endif
    static_for<NG>([&](auto gc) { mma_group(gc, wB, xB); });
#if PPN_DIAG == 3 || PPN_DIAG == 4
#pragma unroll
    for (int i = 0; i < TC * TP / 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[(i * 4 + j / 4) / TP][(i * 4 + j / 4) % TP][j % 4] += acc32[i][j];
#endif
    __syncthreads();                                                 // LDS is reused by the epilogue

    // ---- prefetch hint (ppn_conv_desc.prefetch): touch this workgroup's share of the NEXT launch's weights, one dword per
    // 128-byte line, so that they sit in the Infinity Cache when that launch's first round starts (they were last read a whole
    // forward pass ago).  The values are kept "live" up to the end of the kernel (empty asm below): the loads' latency then
    // passes under the epilogue instead of in front of whatever instruction would reuse their registers.
    int pf_keep0 = 0, pf_keep1 = 0;
    if (a.pf_per_wg) {
        const unsigned l0 = blockIdx.x * a.pf_per_wg + tid, lend = min((blockIdx.x + 1) * a.pf_per_wg, a.pf_lines);
        // ordinary loads (a non-temporal load does not allocate in the Infinity Cache: measured, no effect)
        if (l0 < lend) pf_keep0 = *reinterpret_cast<const int*>(a.pf_ptr + (size_t)l0 * 128);
        if (l0 + 64 * NW < lend) pf_keep1 = *reinterpret_cast<const int*>(a.pf_ptr + (size_t)(l0 + 64 * NW) * 128);
        asm volatile("" : "+v"(pf_keep0), "+v"(pf_keep1));            // issued HERE (not sunk to the use at the kernel's end)
    }

    // ---- epilogue through LDS chunks of 64 pixels (NHWC) / 64 channels (NCHW head) -------------------
    float* ct = reinterpret_cast<float*>(smem);
    // Single-output bf16 launches (every conv1 of a block, the train-mode and input-gradient convolutions): scale,
    // shift and activation are applied on the accumulators, the WHOLE tile goes to LDS once as bf16 ([pixel][channel],
    // rows padded by 16 B) and leaves as 16-byte stores: one barrier and ~40 % fewer instructions than the chunked
    // f32 path below, same arithmetic (f32 math, one rounding at the bf16 conversion).
    constexpr int RS16 = BC * 2 + 16;                                // bf16 tile row stride in bytes
#ifdef PPN_NO_FAST_EPI
    constexpr bool kFastFits = false;
#else
    constexpr bool kFastFits = !X3 && sizeof(T) == 2 && BC >= 128 && (size_t)BP * RS16 <= 2 * (size_t)STAGE;
#endif
#ifndef PPN_NO_FAST_EPI
    static_assert(!ST || kFastFits, "ST lives in the single-output 16-bit epilogue");
#endif
    bool fast = false;
    if constexpr (kFastFits) fast = !a.nchw && !a.residual && !a.out_act && a.out_raw && (a.Cout & 7) == 0 && !a.out_bf16;
    if (fast) {
        if constexpr (kFastFits) {
            const float slope1 = a.act1 == PPN_ACT_RELU ? 0.f : (a.act1 == PPN_ACT_LRELU ? 0.1f : 1.f);
            const bool al = (!a.scale1 || (reinterpret_cast<size_t>(a.scale1) & 15) == 0) &&
                            (!a.shift1 || (reinterpret_cast<size_t>(a.shift1) & 15) == 0);
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int chl = wc * (BC / WC) + i * 16 + 4 * fq;    // this lane's 4 channels of channel tile i
                const int c = c0 + chl;
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
                if (al && c + 4 <= a.Cout) {
                    if (a.scale1) { const float4 t = *reinterpret_cast<const float4*>(a.scale1 + c); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
                    if (a.shift1) { const float4 t = *reinterpret_cast<const float4*>(a.shift1 + c); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (a.scale1 && c + r < a.Cout) sc[r] = a.scale1[c + r];
                        if (a.shift1 && c + r < a.Cout) sh[r] = a.shift1[c + r];
                    }
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int px = wp * (BP / WP) + j * 16 + frow;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t1 = acc[i][j][r] * sc[r] + sh[r];
                        v[r] = fmaxf(t1, t1 * slope1);
                    }
                    store4<T>(smem + px * RS16 + chl * 2, v);
                }
            }
            constexpr int CPR = BC / 8;                              // 16-byte chunks per tile row
            constexpr int NQ = BP * CPR / NT;                        // chunks per thread
            static_assert(BP * CPR % NT == 0, "tile must split into whole 16-byte chunks per thread");
            const size_t row_bytes = (size_t)a.Cout * 2;
            // ---- ST: BatchNorm partial sums of this tile (ppn_conv_desc.stats_mode) ----------------------------------
            // A thread stores the SAME 8 channels (chunk tid % CPR) of NQ pixels, so it folds them as it goes: mode 1
            // {sum v, sum v^2}, mode 2 {sum g, sum g * xhat} with g = v * act'(x * sc + sh), xhat = (x - mean) * rstd over the
            // BatchNorm input x at the same positions -- of the ROUNDED values v the tile stores, i.e. of what the separate
            // reduction pass (train.hip bn_reduce_kernel) would read back.  Then: the 64 / CPR lanes of a wave that share a
            // chunk (butterfly), the NW waves through LDS in wave order, one f64 pair per channel and tile -- a fixed order.
            static_assert(!ST || NT % CPR == 0, "ST: a thread keeps its chunk column");
            uint4 sx[ST ? NQ : 1];
            float st_sc[8], st_sh[8], st_mu[8], st_rs[8], st_s0[8], st_s1[8];
            float st_neg = 1.f;
            const int st_mode = ST ? a.st_mode : 0;
            // (every loop over the NQ chunks below is a static_for: a `#pragma unroll` loop with the sums inside stayed a loop and
            // indexed sx[] through a branch ladder -- +9..26 us per launch)
            if constexpr (ST) {
#pragma unroll
                for (int j = 0; j < 8; ++j) st_s0[j] = st_s1[j] = 0.f;
                if (st_mode == 2) {
                    const int cc = tid % CPR, c = c0 + cc * 8;
                    const bool vc = c < a.Cout;                      // Cout % 8 == 0: the whole chunk or nothing
                    static_for<NQ>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        const int m = m0 + q * (NT / CPR) + tid / CPR;
                        sx[q] = (vc && m < a.M) ? *reinterpret_cast<const uint4*>(a.st_x + (size_t)m * row_bytes + (size_t)c * 2)
                                                : make_uint4(0u, 0u, 0u, 0u);
                    });
                    const int cb = vc ? c : 0;                       // 8 consecutive floats of each per-channel vector: 2 x 16 bytes
                    float ga[8], be[8];
                    auto vec8 = [&](const float* p, float* o) {
                        if ((reinterpret_cast<size_t>(p) & 15) == 0) {
                            const float4 lo = *reinterpret_cast<const float4*>(p + cb), hi = *reinterpret_cast<const float4*>(p + cb + 4);
                            o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] = p[cb + j];
                        }
                    };
                    vec8(a.st_mean, st_mu); vec8(a.st_rstd, st_rs); vec8(a.st_gamma, ga); vec8(a.st_beta, be);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        st_sc[j] = ga[j] * st_rs[j];
                        st_sh[j] = be[j] - st_mu[j] * st_sc[j];
                    }
                    st_neg = a.st_act == PPN_ACT_RELU ? 0.f : (a.st_act == PPN_ACT_LRELU ? 0.1f : 1.f);
                }
            }
            lds_barrier();
            auto store_chunk = [&](auto qc, auto modec) {
                constexpr int q = decltype(qc)::value;
                constexpr int MODE = decltype(modec)::value;
                const int id = q * NT + tid;
                const int px = id / CPR, cc = id % CPR;
                const int m = m0 + px, c = c0 + cc * 8;
                if (m < a.M && c < a.Cout) {
                    const uint4 o = *reinterpret_cast<const uint4*>(smem + px * RS16 + cc * 16);
                    *reinterpret_cast<uint4*>(a.out_raw + (size_t)m * row_bytes + (size_t)c * 2) = o;
                    if constexpr (MODE != 0) {
                        float v[8];
                        load8<T>(reinterpret_cast<const char*>(&o), v);
                        if constexpr (MODE == 1) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                st_s0[j] += v[j];
                                st_s1[j] = fmaf(v[j], v[j], st_s1[j]);
                            }
                        } else {
                            float x[8];
                            load8<T>(reinterpret_cast<const char*>(&sx[q]), x);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float z = fmaf(x[j], st_sc[j], st_sh[j]);
                                const float g = v[j] * (z > 0.f ? 1.f : st_neg);
                                const float xh = (x[j] - st_mu[j]) * st_rs[j];
                                st_s0[j] += g;
                                st_s1[j] = fmaf(g, xh, st_s1[j]);
                            }
                        }
                    }
                }
            };
            if constexpr (!ST) {                                     // the inference kernels: the loop they always had
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int id = q * NT + tid;
                    const int px = id / CPR, cc = id % CPR;
                    const int m = m0 + px, c = c0 + cc * 8;
                    if (m < a.M && c < a.Cout) {
                        const uint4 o = *reinterpret_cast<const uint4*>(smem + px * RS16 + cc * 16);
                        *reinterpret_cast<uint4*>(a.out_raw + (size_t)m * row_bytes + (size_t)c * 2) = o;
                    }
                }
            } else {
                if (st_mode == 0) static_for<NQ>([&](auto qc) { store_chunk(qc, std::integral_constant<int, 0>{}); });
                if (st_mode == 1) static_for<NQ>([&](auto qc) { store_chunk(qc, std::integral_constant<int, 1>{}); });
                if (st_mode == 2) static_for<NQ>([&](auto qc) { store_chunk(qc, std::integral_constant<int, 2>{}); });
            }
            if constexpr (ST) {
                if (st_mode != 0) {
#pragma unroll
                    for (int off = 32; off >= CPR; off >>= 1)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            st_s0[j] += __shfl_xor(st_s0[j], off, 64);
                            st_s1[j] += __shfl_xor(st_s1[j], off, 64);
                        }
                    lds_barrier();                                   // every wave has read its chunks of the tile: LDS is free
                    float* red = reinterpret_cast<float*>(smem);     // [NW][CPR][16]
                    if (lane < CPR) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            red[(wave * CPR + lane) * 16 + j] = st_s0[j];
                            red[(wave * CPR + lane) * 16 + 8 + j] = st_s1[j];
                        }
                    }
                    lds_barrier();
                    if (tid < CPR * 16) {
                        const int ccx = tid >> 4, j = tid & 15;
                        double acc = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) acc += (double)red[(w * CPR + ccx) * 16 + j];
                        const int ch = c0 + ccx * 8 + (j & 7);
                        if (ch < a.Cout) a.st_partial[((size_t)ptile * a.Cout + ch) * 2 + (j >> 3)] = acc;
                    }
                }
            }
        }
    } else if (!a.nchw) {
        constexpr int LD = BC + 4;                                   // [pixel][channel] f32
        constexpr int JC = (TP % 2 == 0 && WP == 2) ? 2 : 1;         // pixel tiles per wave and chunk
        constexpr int CPX = WP * JC * 16;                            // pixels per chunk (64)
        constexpr int TPP = BC / 8, PPP = NT / TPP;                  // threads per pixel, pixels per pass
        static_assert(TP % JC == 0 && CPX % PPP == 0, "epilogue chunking");
        static_assert((size_t)CPX * LD * 4 <= 2 * (size_t)STAGE, "epilogue chunk must fit the staging LDS");
        const int cg = tid % TPP, prow = tid / TPP;
        const int c = c0 + cg * 8;
        // NHWC layers only use none / ReLU / LeakyReLU(0.1): all three are  t > 0 ? t : t * slope  =  max(t, t * slope)
        // for slope in [0, 1] (same values, signed zeros included; one v_max instead of compare + select)
        const float slope1 = a.act1 == PPN_ACT_RELU ? 0.f : (a.act1 == PPN_ACT_LRELU ? 0.1f : 1.f);
        const float slope2 = a.act2 == PPN_ACT_RELU ? 0.f : (a.act2 == PPN_ACT_LRELU ? 0.1f : 1.f);
        float s1[8], b1[8], s2[8], b2[8];
        // 8 consecutive channels per thread: two 16-byte loads where the run is whole and aligned
        auto affine8 = [&](const float* p, float dflt, float* o) {
            if (p && c + 8 <= a.Cout && (reinterpret_cast<size_t>(p) & 15) == 0) {
                const float4 lo = *reinterpret_cast<const float4*>(p + c), hi = *reinterpret_cast<const float4*>(p + c + 4);
                o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (p && c + i < a.Cout) ? p[c + i] : dflt;
            }
        };
        affine8(a.scale1, 1.f, s1); affine8(a.shift1, 0.f, b1);
        affine8(a.scale2, 1.f, s2); affine8(a.shift2, 0.f, b2);
#ifdef PPN_CLOCK
        unsigned long long ep_w = 0, ep_s = 0;
#endif
        auto chunk = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
#ifdef PPN_CLOCK
            const unsigned long long e0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
            for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    const int px = (wp * JC + jj) * 16 + frow;
                    const int ch = wc * (BC / WC) + i * 16 + 4 * fq;
                    *reinterpret_cast<f32x4*>(ct + px * LD + ch) = acc[i][q * JC + jj];
                }
            lds_barrier();
#ifdef PPN_CLOCK
            const unsigned long long e1 = __builtin_amdgcn_s_memtime();
#endif
            if (c < a.Cout) {
                constexpr int NPASS = CPX / PPP;
                int mrow[NPASS];
                size_t moff[NPASS];                                  // byte offset of (row, c) in the NHWC tensors
                float res[NPASS][8];
                const size_t row_bytes = (size_t)a.Cout * ES * (X3 ? 2 : 1);       // X3: [hi(Cout) | lo'(Cout)] per pixel
                const size_t lo_bytes = (size_t)a.Cout * ES;
                // residual rows of ALL passes are requested first: one memory latency per chunk, not per pass
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int px = pass * PPP + prow;                // 0 .. CPX-1
                    const int pw = px / (JC * 16), pj = (px / 16) % JC, pr = px % 16;
                    const int m = m0 + pw * (BP / WP) + (q * JC + pj) * 16 + pr;
                    mrow[pass] = m < a.M ? m : -1;
                    if constexpr (JC == 1 && PPP % 16 == 0) {
                        // rows of successive passes are a fixed distance apart: one 64-bit multiply per chunk
                        constexpr int DM = (PPP / 16) * (BP / WP);
                        moff[pass] = pass == 0 ? (size_t)m * row_bytes + (size_t)c * ES : moff[0] + pass * (DM * row_bytes);
                    } else {
                        moff[pass] = (size_t)m * row_bytes + (size_t)c * ES;
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) res[pass][i] = 0.f;
                    if (a.residual && m < a.M) {
                        if constexpr (X3) load8_x3(a.residual + moff[pass], lo_bytes, res[pass]);
                        else load8<T>(a.residual + moff[pass], res[pass]);
                    }
                }
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    const int px = pass * PPP + prow;
                    if (mrow[pass] >= 0) {
                        float v[8];
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + px * LD + cg * 8);
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + px * LD + cg * 8 + 4);
                        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
                        v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                        // X3 launch that feeds a plain-half trunk (PPN_CONV_X3_PLAIN_OUT): the residual it READS is a half
                        // pair tensor (row = 2 Cout halves), what it WRITES plain half rows (Cout halves)
                        size_t off = moff[pass];
                        if constexpr (X3) {
                            if (a.out_plain) off = (size_t)mrow[pass] * lo_bytes + (size_t)c * ES;
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float t1 = v[i] * s1[i] + b1[i];
                            v[i] = fmaxf(t1, t1 * slope1) + res[pass][i];
                        }
#if PPN_DIAG == 7
                        if (a.out_raw && v[0] == 12345.678f) store8<T>(a.out_raw + off, v);
#else
                        if (a.out_raw) {
                            if constexpr (X3) {
                                if (a.out_plain) store8<_Float16>(a.out_raw + off, v);
                                else store8_x3(a.out_raw + off, lo_bytes, v);
                            }
                            else if constexpr (std::is_same<T, _Float16>::value) {
                                if (a.out_bf16) store8<__bf16>(a.out_raw + off, v);      // f16 launch feeding a bf16 trunk
                                else store8<T>(a.out_raw + off, v);
                            } else store8<T>(a.out_raw + off, v);
                        }
#endif
                        if (a.out_act) {
                            float u[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float t2 = v[i] * s2[i] + b2[i];
                                u[i] = fmaxf(t2, t2 * slope2);
                            }
                            if constexpr (X3) {
                                if (a.out_plain) store8<_Float16>(a.out_act + off, u);
                                else store8_x3(a.out_act + off, lo_bytes, u);
                            }
                            else if constexpr (std::is_same<T, _Float16>::value) {
                                if (a.out_bf16) store8<__bf16>(a.out_act + off, u);
                                else store8<T>(a.out_act + off, u);
                            } else store8<T>(a.out_act + off, u);
                        }
                    }
                }
            }
            lds_barrier();
#ifdef PPN_CLOCK
            const unsigned long long e2 = __builtin_amdgcn_s_memtime();
            ep_w += e1 - e0; ep_s += e2 - e1;
#endif
        };
        static_for<TP / JC>(chunk);
#ifdef PPN_CLOCK
        if (lane == 0 && a.scale2 == nullptr && a.shift2 != nullptr) {
            unsigned long long* dbg = (unsigned long long*)a.shift2 + ((size_t)blockIdx.x * NW + wave) * 8;
            dbg[7] = (ep_w << 32) | ep_s;
        }
#endif
    } else {
        // head: f32 NCHW [B, Cout, Ho*Wo] (model.py:136): 64 channels per chunk, pixel-contiguous rows
        constexpr int LD = BP + 4;                                   // [channel][pixel] f32
        constexpr int IC = WC == 2 ? 2 : 1;                          // channel tiles per wave and chunk
        constexpr int TPC = BP / 4;                                  // threads per channel row (4 pixels each)
        constexpr int NITEM = WC * IC * 16 * TPC;                    // (channel row, pixel quad) items per chunk
        static_assert(TC % IC == 0, "channel tiles per wave must be even");
        static_assert((size_t)(WC * IC * 16 * LD + 4 * IC * 16) * 4 <= 2 * (size_t)STAGE, "epilogue chunk must fit the staging LDS");
        float* out = reinterpret_cast<float*>(a.out_raw);
        const bool vec = (a.HoWo & 3) == 0;
        auto chunk = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            // The chunk goes to LDS as LOGITS t = acc * scale1 + shift1 (the affine is applied by the lane that holds the
            // accumulator: 4 consecutive channels, two 16-byte loads of constants), so both consumers below read one
            // value per element.
#pragma unroll
            for (int ii = 0; ii < IC; ++ii) {
                const int chl = (wc * IC + ii) * 16 + 4 * fq;            // row of this lane's first channel in the chunk
                const int c = c0 + wc * (BC / WC) + (q * IC + ii) * 16 + 4 * fq;
                float sc[4], sh[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[r] = (a.scale1 && c + r < a.Cout) ? a.scale1[c + r] : 1.f;
                    sh[r] = (a.shift1 && c + r < a.Cout) ? a.shift1[c + r] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int px = wp * (BP / WP) + j * 16 + frow;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct[(chl + r) * LD + px] = acc[q * IC + ii][j][r] * sc[r] + sh[r];
                }
            }
            lds_barrier();
#if PPN_DIAG == 8
            if (a.amax_keys && a.M < 0) {
#else
            if (a.amax_keys) {
#endif
                // Fused decode front end: one thread per (pixel, 32-channel run).  Channels < unary_ch (resp, conf,
                // x, y, w, h) go to the compact tensor; every limb channel competes in its (image, edge, cell)
                // arg-max through one 64-bit atomicMax per run and edge:  key = value bits << 32 | ~s  (sigmoid
                // outputs are >= 0 so the bit pattern is monotonic; ~s makes the LOWEST window index win ties,
                // np.argmax semantics of datatest.py:113).  A 32-channel run crosses at most one edge boundary
                // (window >= 32), so it is processed as two branch-free segments.
                //
                // The key must hold max_k sigmoid(t_k) with the FIRST k that reaches it -- exactly what decoding the
                // materialised head gives -- but evaluating 32 sigmoids per run (v_exp + v_rcp each) made this
                // epilogue 40 % of the head conv.  The sigmoid is monotone up to its rounding, so the winner is the
                // first maximum of the LOGITS unless another logit is so close to the maximum that both sigmoids may
                // round to the same float (or swap by an ulp of the approximations).  One pass over the logits finds
                // the maximum m, its first index and the runner-up m2; the preimage of one float of sigmoid around m
                // is 2^-23 (1 + e^m) wide, and with a factor 8 for the <= 3 ulp of v_exp/v_rcp error the segment is
                // decided by its logits alone when m2 < m - 2^-20 (1 + e^m): ONE sigmoid.  Otherwise (saturated
                // heads: every logit above ~16.6 gives 1.0f, the lowest index must win) the lane falls back to the
                // sigmoid of every element, as before.
                constexpr int RUN = IC * 16;
                const int nedges = (a.Cout - a.unary_ch) / a.window;
                for (int item = tid; item < WC * BP; item += NT) {
                    const int cw = item / BP, px = item - cw * BP;
                    const int m = m0 + px;
                    if (m >= a.M) continue;
                    const int nb = fast_div(m, a.div_howo), np = m - nb * a.HoWo;
                    const int cb = c0 + cw * (BC / WC) + q * RUN;    // first channel of this run
                    const float* row = ct + (cw * RUN) * LD + px;
                    int r = 0;
                    // unary part of the run (only the first channel tile of the layer has one)
                    for (; r < RUN && cb + r < a.unary_ch; ++r)
                        a.unary_out[((size_t)nb * a.unary_ch + cb + r) * a.HoWo + np] = sigmoid_fast(row[r * LD]);
                    const int rend = min(RUN, a.Cout - cb);           // padded channels past Cout do not compete
                    if (r < rend) {
                        const int l0 = cb + r - a.unary_ch;           // limb channel index of element r
                        int e = l0 / a.window, sidx = l0 - e * a.window;
                        while (r < rend) {
                            const int seg = min(rend - r, a.window - sidx);   // elements left in this edge's window
                            float tm = -3.0e38f, tm2 = -3.0e38f;
                            int km = 0;
#pragma unroll 8
                            for (int k = 0; k < seg; ++k) {
                                const float t = row[(r + k) * LD];
                                tm2 = fmaxf(tm2, fminf(tm, t));       // runner-up (equal maxima count: tm2 == tm)
                                const bool gt = t > tm;                // strict: first maximum of the segment
                                tm = gt ? t : tm;
                                km = gt ? k : km;
                            }
                            float best = sigmoid_fast(tm);
                            int best_k = km;
                            if (!(tm2 < tm - 9.5367431640625e-7f * (1.0f + __expf(tm)))) {
                                best = -1.f;
                                best_k = 0;
                                for (int k = 0; k < seg; ++k) {       // ambiguous: the sigmoid values themselves decide
                                    const float v = sigmoid_fast(row[(r + k) * LD]);
                                    const bool gt = v > best;
                                    best = gt ? v : best;
                                    best_k = gt ? k : best_k;
                                }
                            }
#if PPN_DIAG == 9
                            if (e < nedges && best == 12345.678f) {
#else
                            if (e < nedges) {
#endif
                                const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) |
                                                               (unsigned)(0xFFFFFFFFu - (unsigned)(sidx + best_k));
                                atomicMax(a.amax_keys + ((size_t)nb * nedges + e) * a.HoWo + np, key);
                            }
                            r += seg; sidx = 0; ++e;
                        }
                    }
                }
            }
            if (out)
            for (int item = tid; item < NITEM; item += NT) {
                const int chl = item / TPC, pq = item - chl * TPC;   // 0 .. 2*IC*16-1
                const int cw = chl / (IC * 16), ci = (chl / 16) % IC, cr = chl % 16;
                const int c = c0 + cw * (BC / WC) + (q * IC + ci) * 16 + cr;
                const int m = m0 + 4 * pq;
                if (c < a.Cout && m < a.M) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(ct + chl * LD + 4 * pq);
                    float v[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], a.act1);
                    if (vec) {
                        const int nb = fast_div(m, a.div_howo), np = m - nb * a.HoWo;
                        *reinterpret_cast<float4*>(out + ((size_t)nb * a.Cout + c) * a.HoWo + np) =
                            make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int mi = m + i;
                            if (mi < a.M) {
                                const int bi = fast_div(mi, a.div_howo), pi = mi - bi * a.HoWo;
                                out[((size_t)bi * a.Cout + c) * a.HoWo + pi] = v[i];
                            }
                        }
                    }
                }
            }
            lds_barrier();
        };
        static_for<TC / IC>(chunk);
    }
    asm volatile("" ::"v"(pf_keep0), "v"(pf_keep1));              // the prefetch loads are waited for here, not earlier
#ifdef PPN_CLOCK
    if (lane == 0 && a.scale2 == nullptr && a.shift2 != nullptr) {
        unsigned long long* dbg = (unsigned long long*)a.shift2 + ((size_t)blockIdx.x * NW + wave) * 8;
        dbg[4] = __builtin_amdgcn_s_memtime() - dbg[3];           // epilogue cycles
#if PPN_CLOCK + 0 >= 2
        // tools/clock_conv_seq.py: when the workgroup entered and left, in 10 ns ticks of the chip-wide constant clock
        // (instead of the two prologue marks), so that rounds, launch ramp and tail can be laid on one time axis
        dbg[5] = rk_entry; dbg[6] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    (void)rk_entry;
#endif
#ifdef PPN_STAMP
    {
        unsigned long long t_epi1;
        PPN_T(t_epi1);
        if (lane == 0 && a.scale2 == nullptr && a.shift2 != nullptr) {
            unsigned long long* dbg = (unsigned long long*)a.shift2 + ((size_t)blockIdx.x * NW + wave) * 8;
            dbg[4] = t_epi1 - t_epi0;
        }
    }
#endif
}

template <typename T, int BP, int BC, int NW, bool SC, bool X3 = false, bool ST = false>
int launch_sc(const ConvKArgs& a, hipStream_t st, const char** kname) {
    constexpr size_t lds = 2 * (size_t)(BP + BC) * 128;
    static char name[96];
    if (!name[0])
        snprintf(name, sizeof(name), "conv_igemm_big_kernel<%s, %d, %d, %d, %s%s>", elem_name<T>(),
                 BP, BC, NW, SC ? "true" : "false", X3 ? ", true" : (ST ? ", false, true" : ""));
    if (kname) *kname = name;
    const size_t src_bytes = (size_t)a.B * a.H * a.W * a.Cin * sizeof(T);
    const size_t wgt_bytes = (size_t)a.n_ctiles * BC * a.Ktot * sizeof(T);
    auto k = conv_igemm_big_kernel<T, BP, BC, NW, SC, X3, ST>;
    {
        static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
    }
    ConvKArgs b = a;
    if (b.pf_lines) {                                  // prefetch hint: lines per workgroup, at most two per thread
        const unsigned nwg = (unsigned)(a.n_ctiles * a.n_ptiles);
        b.pf_per_wg = std::min<unsigned>((b.pf_lines + nwg - 1) / nwg, 2u * 64 * NW);
    }
    hipLaunchKernelGGL(k, dim3(a.n_ctiles * a.n_ptiles), dim3(64 * NW), lds, st, b, (unsigned)src_bytes,
                       (unsigned)wgt_bytes);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

template <typename T, int BP, int BC, int NW>
int launch_one(const ConvKArgs& a, hipStream_t st, const char** kname) {
    if constexpr (NW == 8 && BC >= 128) {
        if (a.src2) return launch_sc<T, BP, BC, NW, true>(a, st, kname);
    } else {
        if (a.src2) return ppn::fail(PPN_E_UNSUPPORTED, "fused shortcut needs the 8-wave 128/256-channel tiles");
    }
    if (a.st_mode != 0) {                              // conv_launch only asks where stats_tile_ok() said yes
        if constexpr (std::is_same<T, __bf16>::value && NW == 8 && stats_tile_ok(BP, BC))
            return launch_sc<T, BP, BC, NW, false, false, true>(a, st, kname);
        else
            return ppn::fail(PPN_E_UNSUPPORTED, "BatchNorm statistics: no such instantiation of the large-tile kernel");
    }
    return launch_sc<T, BP, BC, NW, false>(a, st, kname);
}

int conv_waves() {
    static const int nw = getenv("PPN_CONV_WAVES") ? atoi(getenv("PPN_CONV_WAVES")) : 8;   // tuning knob
    return nw;
}

template <typename T>
int launch_T(const ConvKArgs& a, BigTile t, hipStream_t st, const char** kname) {
    const int nw = conv_waves();
    if (nw == 4 && t.bp != 144) {
        if (t.bc == 256) {
            if (t.bp == 256) return launch_one<T, 256, 256, 4>(a, st, kname);
            if (t.bp == 192) return launch_one<T, 192, 256, 4>(a, st, kname);
            return launch_one<T, 128, 256, 4>(a, st, kname);
        }
        if (t.bp == 256) return launch_one<T, 256, 128, 4>(a, st, kname);
        if (t.bp == 192) return launch_one<T, 192, 128, 4>(a, st, kname);
        return launch_one<T, 128, 128, 4>(a, st, kname);
    }
    if (t.bc == 256) {
        if (t.bp == 256) return launch_one<T, 256, 256, 8>(a, st, kname);
        if (t.bp == 192) return launch_one<T, 192, 256, 8>(a, st, kname);
        if (t.bp == 144) return launch_one<T, 144, 256, 8>(a, st, kname);
        return launch_one<T, 128, 256, 8>(a, st, kname);
    }
    if (t.bc == 128) {
        if (t.bp == 256) return launch_one<T, 256, 128, 8>(a, st, kname);
        if (t.bp == 192) return launch_one<T, 192, 128, 8>(a, st, kname);
        return launch_one<T, 128, 128, 8>(a, st, kname);
    }
    if (t.bp == 256) return launch_one<T, 256, 64, 8>(a, st, kname);
    return launch_one<T, 128, 64, 8>(a, st, kname);
}

// split-f16 mode (PPN_F16X3): the 8-wave tiles, no fused shortcut
int launch_X3(const ConvKArgs& a, BigTile t, hipStream_t st, const char** kname) {
    if (a.src2) return ppn::fail(PPN_E_UNSUPPORTED, "PPN_F16X3: no fused shortcut");
    if (t.bc == 256) {
        if (t.bp == 256) return launch_sc<_Float16, 256, 256, 8, false, true>(a, st, kname);
        if (t.bp == 192) return launch_sc<_Float16, 192, 256, 8, false, true>(a, st, kname);
        if (t.bp == 144) return launch_sc<_Float16, 144, 256, 8, false, true>(a, st, kname);
        return launch_sc<_Float16, 128, 256, 8, false, true>(a, st, kname);
    }
    if (t.bc == 128) {
        if (t.bp == 256) return launch_sc<_Float16, 256, 128, 8, false, true>(a, st, kname);
        if (t.bp == 192) return launch_sc<_Float16, 192, 128, 8, false, true>(a, st, kname);
        return launch_sc<_Float16, 128, 128, 8, false, true>(a, st, kname);
    }
    if (t.bp == 256) return launch_sc<_Float16, 256, 64, 8, false, true>(a, st, kname);
    return launch_sc<_Float16, 128, 64, 8, false, true>(a, st, kname);
}

}  // namespace

namespace ppnconv {

// Pick the tile: channels 256 / 128 / 64 by Cout, pixel height so that the fewest CU-rounds are wasted
// (256 CUs; one workgroup per CU, two for the 64-channel tile whose LDS footprint is 80 KB).
// PPN_CONV_TILE="bp,bc" overrides the choice for every eligible layer (tuning knob).
static int g_tile_policy = getenv("PPN_CONV_CONT") ? 1 : 0;
// forced tile (ppn_set_conv_tile_override; initial value from PPN_CONV_TILE="bp,bc"): 0,0 = automatic choice
static int g_ov_bp = -1, g_ov_bc = 0;

static bool tile_shape_ok(int bp, int bc) {
    if (bp == 144) return bc == 256;                     // the 8 x 1 wave grid exists for the 256-channel tile only
    return (bp == 128 || bp == 192 || bp == 256) && (bc == 64 || bc == 128 || bc == 256) && !(bc == 64 && bp == 192);
}

// relative efficiency of the 144 x 256 tile in the cost model below (PPN_EFF144 overrides: 0 takes the tile out).
// Measured (round 4, same box, in sequence): the five 24 x 24 512 -> 512 layers 106.5 -> 94.9 us (= 0.93 on this scale), but
// layer5's 256 -> 256 launches 101.5 -> 118 / 87.8 -> 96.1 us against 192 x 128 at three rounds (= 0.82-0.87): 0.90 picks it
// for the former only.  It shortens a LONE launch by filling all 256 CUs with smaller, less efficient workgroups (CU-time
// per launch +19 %): with three lanes in flight the same switch cost 3.5 % of the images/s (10.69 k vs 11.07 k, two
// interleaved pairs of runs), so plans that share the GPU (PPN_CONV_SHARED_GPU) never take it.
static const double kEff144 = getenv("PPN_EFF144") ? atof(getenv("PPN_EFF144")) : 0.90;

bool big_tile_for(int cout, long long m, BigTile* out, int ksteps, bool shared_gpu) {
    if (cout < 64) return false;
    if (g_ov_bp < 0) {                                   // first call: the environment knob
        g_ov_bp = g_ov_bc = 0;
        int bp = 0, bc = 0;
        const char* ov = getenv("PPN_CONV_TILE");
        if (ov && sscanf(ov, "%d,%d", &bp, &bc) == 2 && tile_shape_ok(bp, bc)) { g_ov_bp = bp; g_ov_bc = bc; }
    }
    // a forced tile applies to every layer whose padded Cout it divides (the packed weights are padded to the
    // LARGEST channel tile of the layer's Cout class, ppn_conv_tiling, so any smaller power-of-two tile fits)
    if (g_ov_bp > 0 && g_ov_bc <= ((cout + 63) / 64) * 64 && g_ov_bc <= (cout >= 256 ? 256 : (cout >= 128 ? 128 : 64))) {
        out->bp = g_ov_bp; out->bc = g_ov_bc;
        return true;
    }
    if (cout >= 4096) {
        // the head conv (512 -> 7605, K = 512) writes 17.5 MB/image and is store-bound: a 192x128 tile keeps
        // two workgroups per CU so one's epilogue stores overlap the other's main loop (measured 322 vs 399 us)
        out->bp = 192; out->bc = 128;
        return true;
    }
    // time ~ ceil(tiles / 256 CUs) * tile area / eff(tile); eff = relative throughput of a tile shape measured on
    // the 48x48x512 3x3 layers at batch 32 (tools/bench_conv.py with PPN_CONV_TILE), i.e. how well its FLOPs per
    // staged byte feed the MFMA pipe.  For Cout = 256 the 128-channel tile wins through quantisation
    // (768 workgroups = 3 full rounds instead of 384 = 1.5).
    struct Cand { int bp, bc; double eff; };
    // 144 x 256 (round 4; 8 waves side by side along the channels, each 32 channels x all 144 pixels): 18 432 pixels x 512
    // channels (the five 24 x 24 layers of DRN-D-22 at batch 32) = exactly 256 workgroups = one full round, where 192 x 256
    // leaves 64 of the 256 CUs idle; and 73 728 x 256 (layer5) = 512 = two rounds.  Same K order: results unchanged.
    // (round 4, measured and removed: a 288 x 256 tile -- 4 x 2 wave grid, 36 MFMA tiles per wave, 256 VGPRs with 6 spilled --
    // makes layer5's 73 728 x 256 launches ONE round of 256 workgroups: 93.9 -> 77.8 us alone, but -1.7 % images/s with three
    // lanes (a 136 KB workgroup holds its CU alone for 78 us; two 80 KB workgroups of 192 x 128 share one), nothing on a
    // single lane or the training step, and on the 512-channel layers its two rounds take what 192 x 256's three do.)
    static const Cand cands[] = {{256, 256, 1.27}, {192, 256, 1.10}, {144, 256, kEff144}, {128, 256, 0.92}, {256, 128, 0.90},
                                 {192, 128, 0.95}, {128, 128, 0.87}, {256, 64, 0.60},  {128, 64, 0.55}};
    int bc_max = cout >= 256 ? 256 : (cout >= 128 ? 128 : 64);
    const int bc_min = cout >= 256 ? 128 : bc_max;
    // 1x1 projections (<= 8 K steps) are prologue/epilogue-bound: the 128-channel tile's shorter epilogue wins over the
    // 256-channel tile's staging efficiency (tools/ab_tiles.py: 256->512 at 48x48 32.5 vs 38.4 us, 512->512 stride 2
    // 15.9 vs 17.1, 128->512 at 24x24 10.7 vs 11.6)
    if (ksteps > 0 && ksteps <= 8 && bc_max == 256) bc_max = 128;
    double best = 1e30;
    for (const Cand& cd : cands) {
        if (cd.bc > bc_max || cd.bc < bc_min || cd.eff <= 0.0 || (shared_gpu && cd.bp == 144)) continue;
        const long long tiles = ((m + cd.bp - 1) / cd.bp) * ((cout + cd.bc - 1) / cd.bc);
        // policy 1 (several launches in flight on different streams, ppn_set_conv_tile_policy): another stream's
        // workgroups fill a partial last round, so only the tile's efficiency counts
        // (round 4, measured and dropped: under PPN_CONV_SHARED_GPU, launches with fewer tiles than CUs -- the 24 x 24 layers --
        // priced by efficiency alone, i.e. 256 x 256 tiles on 144 of the CUs: 10.79 / 10.79 k images/s against 10.83 / 10.84 k
        // with the whole-round rule, interleaved runs on one box, and 3.25 vs 3.16 ms for the conv stack in sequence)
        // PPN_POLICY1_MASK (experiment knob, with policy 1): bit 0 = 512-wide layers at >= 65 536 pixels, bit 1 = 512-wide below that,
        // bit 2 = the narrower layers: which classes are priced by fractional rounds
        static const int p1mask = getenv("PPN_POLICY1_MASK") ? atoi(getenv("PPN_POLICY1_MASK")) : 7;
        const int cls = cout >= 512 ? (m >= 65536 ? 1 : 2) : 4;
        const bool frac = g_tile_policy == 1 && (p1mask & cls);
        const double rounds = frac ? (double)tiles / 256.0 : (double)((tiles + 255) / 256);
        const double cost = rounds * cd.bp * cd.bc / cd.eff;
        if (cost < best) { best = cost; out->bp = cd.bp; out->bc = cd.bc; }
    }
    return true;
}

// Two-segment launch (tile policy 2, opt-in): the most efficient tile (256x256: 1.27 vs 1.10 for 192x256) wastes most
// of a round when its workgroup count is not a multiple of the 256 CUs (73 728 pixels x 512 channels = 576 tiles =
// 2.25 rounds), which is why the single-tile choice settles for 192x256 there (768 = 3 rounds).  Cutting the pixel
// range instead -- whole rounds of the big tile, then ONE round of a small tile over the remaining pixels -- should cost
// 2 x 51.6 k + 18.8 k = 122 k units instead of 134 k.  MEASURED (round 2, tools/ab_split.sh): the 256x256 segments do
// run at 1.28-1.46 PFLOP/s (0.51-0.58 of peak) instead of 1.14-1.28, but the lone 128x128 round takes 55-58 us instead of
// the ~35 the model assumes (one small workgroup per CU stages 2x the bytes per FLOP of the big tile through the same
// L2->LDS path; a four-stage pipeline changed nothing), so the layer is not faster (277 us either way) and the extra
// launches cost 1.8 % (three lanes) to 3.6 % (one lane) of throughput.  Not the default; kept for the pixel-range
// interface it exercises.  Returns the cut (0: single launch).  Only for Cout >= 256, where efficiencies were measured.
long long big_split_for(int cout, long long m) {
    BigTile single;
    if (g_tile_policy != 2 || cout < 256 || cout >= 4096 || !big_tile_for(cout, m, &single) || g_ov_bp > 0) return 0;
    struct Cand { int bp, bc; double eff; };
    static const Cand cands[] = {{256, 256, 1.27}, {192, 256, 1.10}, {128, 256, 0.92}, {256, 128, 0.90},
                                 {192, 128, 0.95}, {128, 128, 0.87}};
    auto rounds_cost = [&](const Cand& cd, long long mm) {
        const long long tiles = ((mm + cd.bp - 1) / cd.bp) * ((cout + cd.bc - 1) / cd.bc);
        return (double)((tiles + 255) / 256) * cd.bp * cd.bc / cd.eff;
    };
    double best_single = 1e30;
    for (const Cand& cd : cands) best_single = std::min(best_single, rounds_cost(cd, m));
    double best = best_single * 0.96;                    // the second launch must pay for its own start-up
    long long cut = 0;
    for (const Cand& p : cands) {
        const long long ct = (cout + p.bc - 1) / p.bc;
        if (256 % ct != 0) continue;
        const long long rounds = (m / p.bp) * ct / 256;  // whole rounds of whole tiles
        if (rounds < 1) continue;
        const long long m1 = rounds * 256 / ct * p.bp;
        if (m1 >= m) continue;
        double tail = 1e30;
        for (const Cand& q : cands) tail = std::min(tail, rounds_cost(q, m - m1));
        const double cost = (double)rounds * p.bp * p.bc / p.eff + tail;
        if (cost < best) { best = cost; cut = m1; }
    }
    return cut;
}

// Does a launch of this tile carry the BatchNorm-statistics epilogue (ppn_conv_desc.stats_mode)?  The caller (conv_launch) adds
// the per-launch conditions of the single-output 16-bit epilogue.
bool big_stats_ok(int dtype, BigTile t) {
    return dtype == PPN_BF16 && conv_waves() == 8 && stats_tile_ok(t.bp, t.bc);
}

int launch_big(const ConvKArgs& a, int dtype, BigTile t, hipStream_t st, const char** kname) {
    // buffer descriptors address up to 2 GiB with the out-of-range marker used for padding
    const size_t es = dtype == PPN_F32 ? 4 : 2;
    if ((size_t)a.B * a.H * a.W * a.Cin * es >= 0x7fffff00ull || (size_t)a.n_ctiles * t.bc * a.Ktot * es >= 0x7fffff00ull)
        return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for the buffer-addressed conv kernel");
    if (dtype == PPN_F32) return launch_T<float>(a, t, st, kname);
    if (dtype == PPN_F16X3) return launch_X3(a, t, st, kname);
    if (dtype == PPN_F16) return launch_T<_Float16>(a, t, st, kname);
    return launch_T<__bf16>(a, t, st, kname);
}

}  // namespace ppnconv

extern "C" int ppn_set_conv_tile_override(int32_t bp, int32_t bc) {
    if (bp == 0 && bc == 0) { ppnconv::g_ov_bp = ppnconv::g_ov_bc = 0; return PPN_OK; }
    if (!ppnconv::tile_shape_ok(bp, bc))
        return ppn::fail(PPN_E_INVALID, "ppn_set_conv_tile_override: bp in {128,192,256}, bc in {64,128,256}, not 192x64; or 144x256");
    ppnconv::g_ov_bp = bp; ppnconv::g_ov_bc = bc;
    return PPN_OK;
}

extern "C" int ppn_set_conv_tile_policy(int32_t policy) {
    if (policy < 0 || policy > 2) return ppn::fail(PPN_E_INVALID, "ppn_set_conv_tile_policy: 0, 1 or 2");
    ppnconv::g_tile_policy = policy;
    return PPN_OK;
}
