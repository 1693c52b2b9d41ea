// 3x3 stride-1 convolution 64 -> 64 channels (16-bit modes) with the full fused epilogue of conv.hip -- the three 64-wide
// BasicBlock convolutions of DRN-D's layer3 (drn.py:42-57 at 96x96 for a 384x384 input) and, in training, their raw and
// input-gradient forms.  With the generic implicit-GEMM kernels these launches are bound by fixed costs, not by MFMA or
// HBM work: K = 576 is nine 64-deep steps, so a 128x64 tile spends more cycles on its prologue and epilogue than in its K
// loop (46-60 us per launch at batch 32 where the MFMA work is ~11 us and the HBM traffic ~20 us).  This kernel removes
// the per-tile fixed costs instead of tuning them:
//
//   weights    the WHOLE 64 x 576 filter bank lives in registers as MFMA A fragments (each wave: 32 channels x 576 = 144
//              VGPRs), loaded once per workgroup: no weight traffic, no weight fragment reads in the loop
//   tiles      PERSISTENT workgroups (4 waves, two workgroups per CU) walk 8 x 16-pixel output tiles; the 10 x 18 x 64
//              input patch of a tile is fetched ONCE by LDS-DMA (23 KB instead of nine shifted 16 KB tiles) into one of
//              two patch buffers -- the patch of tile t+1 is requested before tile t's MFMAs start
//   taps       a pixel is one 128-byte LDS row (64 channels = one K step), so tap (dy, dx) of an MFMA pixel tile is the
//              same 16 consecutive patch rows shifted by dy * 18 + dx rows: the nine taps are nine fragment reads of
//              the resident patch (rows XOR-swizzled by their patch column on the source side)
//   epilogue   through a dedicated f32 LDS tile (not the patch buffers: the next patch is in flight), then 16-byte
//              coalesced residual loads / stores: v = act1(acc*s1+b1) (+res); out_raw = v; out_act = act2(v*s2+b2) --
//              the same arithmetic, in the same order, as conv.hip / conv_big.hip; K is accumulated in the same order
//              (tap-major, two 32-deep halves), so results are bit-identical to the generic kernels
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

struct C64Args {
    const char* src;                 // NHWC [B][H][W][64]
    const char* wgt;                 // packed [64][576], k = tap * 64 + ci
    const float *scale1, *shift1, *scale2, *shift2;
    const char* residual;
    char* out_raw;
    char* out_act;
    int B, H, W, act1, act2;
    int tiles_x, tiles_y, n_tiles;   // 16-wide x 8-high output tiles per image
    FastDiv div_tx, div_tpi;         // by tiles_x, by tiles per image
    unsigned long long* dbg;         // -DPPN_CLOCK builds only (tools/clock_conv64.py)
};

constexpr int TH = 8, TW = 16;                    // output tile
constexpr int PH = TH + 2, PW = TW + 2;           // input patch (halo 1)
constexpr int PROWS = PH * PW;                    // 180 pixel rows of 128 B
constexpr int PATCH = ((PROWS + 7) / 8) * 8 * 128;   // 23 552 B: whole 8-row DMA instructions
constexpr int NDMA = PATCH / 1024;                // 23 wave-instructions per patch
constexpr int EPI_LD = 64 + 4;                    // f32 tile row stride
constexpr int EPI = TH * TW * EPI_LD * 4;         // 34 816 B
constexpr int LDS_BYTES = 2 * PATCH + EPI;        // 81 920 B: two workgroups per CU (the affine constants sit in the
                                                  // f32 tile's row padding)

template <typename T>
__global__ void __launch_bounds__(256, 2) conv64_kernel(C64Args a, unsigned src_bytes) {
    static_assert(sizeof(T) == 2, "16-bit modes only (the f32 filter bank does not fit the register file)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;      // 2 (channels) x 2 (pixel halves: tile rows 0-3 / 4-7)
    float* ct = reinterpret_cast<float*>(smem + 2 * PATCH);

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, src_bytes, 0x00020000);

    // ---- the filter bank as MFMA A fragments: wf[tap][half][channel tile] ------------------------------------------------
    f32x4 wf[9][2][2];
    {
        const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ch = wc * 32 + i * 16 + frow;
                wf[t][h][i] = *reinterpret_cast<const f32x4*>(a.wgt + ((size_t)ch * 576 + t * 64 + h * 32 + fq * 8) * 2);
            }
    }

    // ---- patch loader: instruction g of a patch covers LDS rows 8g .. 8g+7; this lane's row and (swizzled) chunk ------
    auto issue_patch = [&](int tile, int buf) {
        const int img = fast_div(tile, a.div_tpi), rem = tile - img * (a.tiles_x * a.tiles_y);
        const int ty = fast_div(rem, a.div_tx), tx = rem - ty * a.tiles_x;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        for (int g = wave; g < NDMA; g += 4) {
            const int row = g * 8 + (lane >> 3);
            const int py = row / PW, px = row - py * PW;             // (constant divisor: multiply-shift)
            const int y = y0 + py, x = x0 + px;
            const int chunk = (lane & 7) ^ ((px >> 1) & 7);         // swizzle by the patch COLUMN (see xaddr below)
            const bool ok = row < PROWS && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + y) * a.W + x) * 128 + chunk * 16) : kOOB;
            bufload_lds16(xrs, smem + buf * PATCH + g * 1024, off);
        }
    };

    const int G = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= a.n_tiles) return;
    issue_patch(tile, 0);
#ifdef PPN_CLOCK
    const unsigned long long ck_start = __builtin_amdgcn_s_memtime();
    unsigned long long ck_w = 0, ck_wait = 0, ck_m = 0, ck_e = 0, ck_n = 0;
    { f32x4 keep = wf[8][1][1]; asm volatile("" :: "v"(keep)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ck_w = __builtin_amdgcn_s_memtime() - ck_start;
#define C64_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define C64_T(v) do { } while (0)
#endif

    const float slope1 = a.act1 == PPN_ACT_RELU ? 0.f : (a.act1 == PPN_ACT_LRELU ? 0.1f : 1.f);
    const float slope2 = a.act2 == PPN_ACT_RELU ? 0.f : (a.act2 == PPN_ACT_LRELU ? 0.1f : 1.f);
    // the epilogue's affine constants live in LDS (a global load per tile would put a memory round trip on every tile)
    // constant k of [scale1 | shift1 | scale2 | shift2][64] sits in the 4-float padding of staging-tile row k / 4
    auto cst = [&](int k) { return ct + (k >> 2) * EPI_LD + 64 + (k & 3); };
    if (tid < 64) {
        *cst(tid) = a.scale1 ? a.scale1[tid] : 1.f;
        *cst(64 + tid) = a.shift1 ? a.shift1[tid] : 0.f;
        *cst(128 + tid) = a.scale2 ? a.scale2[tid] : 1.f;
        *cst(192 + tid) = a.shift2 ? a.shift2[tid] : 0.f;
    }
    // Residual loads and output stores go through buffer descriptors with an out-of-range offset for pixels outside the
    // image: EVERY wave then issues exactly 4 loads and n_st stores per tile, which is what lets the top of the tile loop
    // wait with a COUNTED vmcnt for the patch (older than the previous tile's stores) instead of draining the stores.
    const unsigned tensor_bytes = src_bytes;                        // residual / outputs have the input's shape
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual : a.src), 0,
                                                                         a.residual ? tensor_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_raw ? a.out_raw : a.out_act), 0,
                                                                         a.out_raw ? tensor_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out_act ? a.out_act : a.out_raw), 0,
                                                                         a.out_act ? tensor_bytes : 0u, 0x00020000);
    const bool two_out = a.out_raw && a.out_act;                     // wave-uniform: 8 stores per tile, else 4
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

    for (int it = 0;; ++it) {
        const int buf = it & 1;
        const int next = tile + G;
        const bool more = next < a.n_tiles;
        // The patch of this tile was requested one tile ago (before the previous tile's 4 + n_st epilogue operations): wait
        // for everything but the previous tile's stores.  Everyone has left the previous tile (barrier) before the other
        // patch buffer is refilled and before the f32 staging tile is rewritten.
        C64_T(t0_);
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (two_out) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        C64_T(t1_);
        if (more) issue_patch(next, buf ^ 1);

        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* pb = smem + buf * PATCH;
        // lane constants re-derived per tile from laundered copies: hipcc otherwise hoists the 72 fragment addresses out of
        // the tile loop and spills them (the filter bank leaves no register to spare)
        int frow = lane & 15, fq = lane >> 4, cg = tid & 7, prow = tid >> 3;
        asm volatile("" : "+v"(frow), "+v"(fq), "+v"(cg), "+v"(prow));
        // this thread's 4 output rows (pixel q*32 + prow of the tile, 8 channels from cg*8): byte offsets, out of range
        // outside the image
        const int img = fast_div(tile, a.div_tpi), rem = tile - img * (a.tiles_x * a.tiles_y);
        const int ty = fast_div(rem, a.div_tx), tx = rem - ty * a.tiles_x;
        unsigned ooff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int px = q * 32 + prow;
            const int y = ty * TH + (px >> 4), x = tx * TW + (px & 15);
            ooff[q] = (y < a.H && x < a.W) ? (unsigned)((((img * a.H + y) * a.W + x) * 64 + cg * 8) * 2) : kOOB;
        }
        // x fragment of pixel tile j (tile row wp*4 + j), tap (dy, dx), half h: 16 consecutive patch rows.  Rows are
        // XOR-swizzled by their patch COLUMN ((col >> 1) & 7; 16 consecutive columns still cover every (parity, key) pair,
        // so the reads stay conflict-free), which makes the swizzle independent of the patch row: six per-lane addresses
        // (dx, h) cover all 72 reads of a tile, the (j + dy) row offset is an immediate -- no address arithmetic in the loop
        const char* xaddr[3][2];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = frow + dx;
                xaddr[dx][h] = pb + (wp * 4 * PW + col) * 128 + (((4 * h + fq) ^ ((col >> 1) & 7)) << 4);
            }
        auto xread = [&](int j, int dy, int dx, int h) {
            return *reinterpret_cast<const f32x4*>(xaddr[dx][h] + (j + dy) * (PW * 128));
        };
        f32x4 xa[4], xb[4];
        u32x4 rres[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = xread(j, 0, 0, 0);
        // 18 half-steps (tap-major, halves 0 then 1: the generic kernels' K order); the fragments of half-step s+1 are
        // read while the MFMAs of half-step s issue; the residual rows are requested two thirds of the way through, so
        // their latency passes under the remaining MFMAs
        static_for<18>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int t = s / 2, h = s % 2;
            constexpr int s1_ = s + 1, t1 = s1_ / 2, h1 = s1_ % 2;
            f32x4 (&cur)[4] = (s % 2 == 0) ? xa : xb;
            f32x4 (&nxt)[4] = (s % 2 == 0) ? xb : xa;
            if constexpr (s1_ < 18) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = xread(j, t1 / 3, t1 % 3, h1);
            }
            if constexpr (s == 11) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rres[q] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)ooff[q], 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma_step(acc[i][j], wf[t][h][i], cur[j], (T*)nullptr);
        });

        C64_T(t2_);
        // ---- epilogue: accumulators -> f32 LDS tile -> 16-byte rows -----------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int px = (wp * 4 + j) * 16 + frow;
                *reinterpret_cast<f32x4*>(ct + px * EPI_LD + wc * 32 + i * 16 + 4 * fq) = acc[i][j];
            }
        lds_barrier();
        {
            float s1[8], b1[8], s2[8], b2[8];
            auto ld8 = [&](const float* lo_p, const float* hi_p, float* o) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(lo_p), hi = *reinterpret_cast<const f32x4*>(hi_p);
                o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
            };
            ld8(cst(cg * 8), cst(cg * 8 + 4), s1); ld8(cst(64 + cg * 8), cst(68 + cg * 8), b1);
            ld8(cst(128 + cg * 8), cst(132 + cg * 8), s2); ld8(cst(192 + cg * 8), cst(196 + cg * 8), b2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = q * 32 + prow;
                float v[8];
                ld8(ct + px * EPI_LD + cg * 8, ct + px * EPI_LD + cg * 8 + 4, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t1 = v[i] * s1[i] + b1[i];
                    v[i] = fmaxf(t1, t1 * slope1);
                }
                if (a.residual) {                                    // (workgroup-uniform; a NULL residual loaded zeros)
                    float r[8];
                    load8<T>(reinterpret_cast<const char*>(&rres[q]), r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += r[i];
                }
                u32x4 o;
                if (a.out_raw) {
                    store8<T>(reinterpret_cast<char*>(&o), v);
                    __builtin_amdgcn_raw_buffer_store_b128(o, ors, (int)ooff[q], 0, 0);
                }
                if (a.out_act) {
                    float u[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float t2 = v[i] * s2[i] + b2[i];
                        u[i] = fmaxf(t2, t2 * slope2);
                    }
                    store8<T>(reinterpret_cast<char*>(&o), u);
                    __builtin_amdgcn_raw_buffer_store_b128(o, ars, (int)ooff[q], 0, 0);
                }
            }
        }
#ifdef PPN_CLOCK
        { const unsigned long long t3_ = __builtin_amdgcn_s_memtime(); ck_wait += t1_ - t0_; ck_m += t2_ - t1_; ck_e += t3_ - t2_; ++ck_n; }
#endif
        if (!more) break;
        tile = next;
    }
#ifdef PPN_CLOCK
    if (lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        d[0] = ck_w; d[1] = ck_wait; d[2] = ck_m; d[3] = ck_e; d[4] = ck_n; d[5] = __builtin_amdgcn_s_memtime() - ck_start;
    }
#endif
}

}  // namespace

namespace ppn {

static int g_conv64_on = -1;             // -1: not decided yet (PPN_CONV64=0 in the environment disables it)

bool conv64_supported(const ppn_conv_desc* d) {
    if (g_conv64_on < 0) g_conv64_on = (getenv("PPN_CONV64") && atoi(getenv("PPN_CONV64")) == 0) ? 0 : 1;
    if (!g_conv64_on || (d->flags & PPN_CONV_NO_FILTER_BANK)) return false;
    return (d->dtype == PPN_BF16 || d->dtype == PPN_F16) && d->cin == 64 && d->cout == 64 && d->ksize == 3 && d->stride == 1 &&
           d->dilation == 1 && d->pad == 1 && !d->src2 && !d->out_nchw_f32 && !d->argmax_keys && d->k_total == 576 &&
           d->cout_pad == 64 && d->m_count == 0 && d->act1 != PPN_ACT_SIGMOID && d->act2 != PPN_ACT_SIGMOID;
}

int conv64_launch(const ppn_conv_desc* d, hipStream_t st, const char** kname) {
    if (!d->src || !d->weight || (!d->out_raw && !d->out_act)) return ppn::fail(PPN_E_INVALID, "conv64: NULL src/weight/output");
    const size_t src_bytes = (size_t)d->batch * d->in_h * d->in_w * 64 * 2;
    if (src_bytes >= 0x7fffff00ull) return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for the buffer-addressed conv kernel");
    C64Args a;
    a.src = static_cast<const char*>(d->src);
    a.wgt = static_cast<const char*>(d->weight);
    a.scale1 = d->scale1; a.shift1 = d->shift1; a.scale2 = d->scale2; a.shift2 = d->shift2;
    a.residual = static_cast<const char*>(d->residual);
    a.out_raw = static_cast<char*>(d->out_raw);
    a.out_act = static_cast<char*>(d->out_act);
    a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.act1 = d->act1; a.act2 = d->act2;
    a.tiles_x = (d->in_w + TW - 1) / TW; a.tiles_y = (d->in_h + TH - 1) / TH;
    const long long nt = (long long)d->batch * a.tiles_x * a.tiles_y;
    if (nt > 0x7fffffffLL) return ppn::fail(PPN_E_UNSUPPORTED, "too many tiles");
    a.n_tiles = (int)nt;
    a.div_tx = make_fastdiv((unsigned)a.tiles_x);
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
#ifdef PPN_CLOCK
    a.dbg = (unsigned long long*)d->zero_page;   // diagnostic channel of the stamped build
#else
    a.dbg = nullptr;
#endif
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        PPN_HIP_CHECK(hipGetDevice(&dev));
        PPN_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int grid = (int)std::min<long long>(nt, 2LL * n_cu);      // persistent: two workgroups per CU
    static char names[2][48];
    if (!names[0][0]) {
        snprintf(names[0], sizeof(names[0]), "conv64_kernel<__bf16>");
        snprintf(names[1], sizeof(names[1]), "conv64_kernel<_Float16>");
    }
    if (d->dtype == PPN_F16) {
        if (kname) *kname = names[1];
        static int set16 = 0;
        PPN_LDS_ONCE(set16, reinterpret_cast<const void*>(conv64_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipLaunchKernelGGL(conv64_kernel<_Float16>, dim3(grid), dim3(256), LDS_BYTES, st, a, (unsigned)src_bytes);
    } else {
        if (kname) *kname = names[0];
        static int setb = 0;
        PPN_LDS_ONCE(setb, reinterpret_cast<const void*>(conv64_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipLaunchKernelGGL(conv64_kernel<__bf16>, dim3(grid), dim3(256), LDS_BYTES, st, a, (unsigned)src_bytes);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // namespace ppn

extern "C" int ppn_set_conv64_enabled(int32_t on) {
    ppn::g_conv64_on = on ? 1 : 0;
    return PPN_OK;
}
