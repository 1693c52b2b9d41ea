// Convolution weight gradient for gfx950:  dW[co][ci][dy][dx] = sum_p dY[p][co] * X[p @ (dy,dx)][ci]
// (what autograd computes for every nn.Conv2d of drn.py / model.py in loss.backward(), main.py:677-683).
//
// GEMM view per filter tap: M = cout, N = cin, depth = output pixels p = (b, oy, ox).  Both operands are NHWC, so the
// depth index is the SLOW one in memory for both -- the opposite of what an MFMA operand wants (8 consecutive depth
// values per lane).  The tiles are therefore staged exactly as they lie in memory, [pixel rows][channels], by LDS-DMA
// (buffer_load ... lds, out-of-image taps and ragged ends read as zeros through the buffer range check), and the
// operands are fetched with gfx950's transposing LDS read ds_read_b64_tr_b16 (bf16) -- no shuffle, no second image.
// The f32 parity mode uses v_mfma_f32_16x16x4_f32, whose operands are one scalar per lane (plain ds_read_b32).
//
//   workgroup tile (cout x cin, one tap): 256 x 256 with 8 waves (4 x 2, each 64 x 128) for >= 256-wide layers,
//              128 x 128 with 4 waves (2 x 2, each 64 x 64) otherwise
//   depth step 64 pixels (bf16) / 32 pixels (f32), 2 LDS stages (128 KB / 64 KB)
//   LDS image: pixel rows of 256 or 512 bytes, 16-byte chunk c of row r at slot c ^ (((r&3)<<2) | ((r>>2)&3)) inside
//              its 256-byte window, applied on the SOURCE side of the DMA (the DMA destination is lane-linear);
//              conflict-free for the transposed reads
//   split over pixels: grid.y slices; every slice writes its own f32 partial, a second launch folds them in a fixed
//              order (bitwise reproducible, no atomics) into the reference layout [cout][cin][k][k].
//
// Algorithmic work: 2*P*cout*cin*k*k flops; unique bytes: x + dy once each.
#include "common.h"
#include "conv_common.h"

namespace ppn {   // stem_wgrad.hip: dedicated kernel for the small-channel stem layers (bf16)
bool stem_wgrad_supported(const ppn_wgrad_desc* d);
size_t stem_wgrad_workspace_bytes(const ppn_wgrad_desc* d);
int stem_wgrad_launch(const ppn_wgrad_desc* d, hipStream_t st);
}  // namespace ppn

namespace {

using namespace ppnconv;

constexpr unsigned kOOB = 0x80000000u;

struct WgArgs {
    const char* x;
    const char* dy;
    float* partial;
    int B, H, W, Cin, Ho, Wo, Cout, ks, stride, dil, pad;
    int P, HoWo;
    float inv_wo, inv_howo;
    int n_mt, n_nt, ntaps;
    int steps_per_split, total_steps;
    unsigned x_bytes, dy_bytes;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)lds_wave_base, 16, voff,
                                             0, 0, 0);
}

// q / d for 0 <= q < 2^24 with a float reciprocal and a one-step fix-up
__device__ __forceinline__ int fdiv(int q, int d, float inv, int* rem) {
    int t = (int)((float)q * inv);
    int r = q - t * d;
    if (r < 0) { --t; r += d; }
    else if (r >= d) { ++t; r -= d; }
    *rem = r;
    return t;
}

// LDS image of a [pixel rows][channels] tile: 16-byte chunk `ch` of row `row` sits at chunk slot
// (ch & ~15) | ((ch & 15) ^ (((row&3)<<2) | ((row>>2)&3))) -- rows are a multiple of 256 B apart, i.e. they alias
// to the same banks; the XOR permutes the chunks of every 256-byte window per row so that the transposed reads
// (4 rows x 32 B per 16 lanes) are conflict-free.  f32 tiles are read with scalar ds_read_b32 and stay linear.
__device__ __forceinline__ int swz(int row, int ch) { return (ch & ~15) | ((ch & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// WM x WN waves, each TM x TN MFMA tiles of 16 x 16:  workgroup tile BM = WM*TM*16 (cout) x BN = WN*TN*16 (cin).
//   <2,2,4,4>: 128 x 128, 4 waves, 64 KB LDS (two workgroups per CU)  -- narrow layers
//   <4,2,4,8>: 256 x 256, 8 waves, 128 KB LDS (two waves per SIMD)    -- >= 256-wide layers: twice the FLOP per
//              byte staged through LDS, half the L2 re-reads of x and dy
template <typename T, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(64 * WM * WN) wgrad_kernel(WgArgs a) {
    constexpr int ES = sizeof(T), EPC = Elem<T>::EPC;
    constexpr bool BF = ES == 2;
    constexpr int NWAVE = WM * WN, NTHR = 64 * NWAVE;
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int BKP = BF ? 64 : 32;                                 // pixels per depth step
    constexpr int CPRA = BM / EPC, CPRB = BN / EPC;                   // 16-byte chunks per pixel row
    constexpr int ROWA = BM * ES, ROWB = BN * ES;
    constexpr int TILEA = BKP * ROWA, TILEB = BKP * ROWB;
    constexpr int NPA = TILEA / (NTHR * 16), NPB = TILEB / (NTHR * 16);   // DMA instructions per operand and step
    static_assert(TILEA % (NTHR * 16) == 0 && TILEB % (NTHR * 16) == 0, "tile must split into whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // tile -> (tap, cin tile, cout tile); taps fastest so that concurrent workgroups share the same pixel rows in L2
    int tix = blockIdx.x;
    const int tap = tix % a.ntaps; tix /= a.ntaps;
    const int nt_ = tix % a.n_nt;
    const int mt_ = tix / a.n_nt;
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int tdy = tap / a.ks, tdx = tap % a.ks;
    const int oy_off = tdy * a.dil - a.pad, ox_off = tdx * a.dil - a.pad;

    const int step0 = blockIdx.y * a.steps_per_split;
    int nsteps = a.total_steps - step0;
    nsteps = nsteps < a.steps_per_split ? nsteps : a.steps_per_split;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

    // per-lane DMA geometry: piece i covers the lane-linear LDS range [(i*NWAVE + wave)*1024, +1024)
    int arow[NPA], brow[NPB];
    unsigned a_choff[NPA], b_choff[NPB];              // channel byte offset of the lane's chunk, or kOOB when past C
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int q = (i * NWAVE + wave) * 64 + lane;
        const int row = q / CPRA, slot = q % CPRA;
        const int ch = BF ? swz(row, slot) : slot;     // the XOR is an involution: slot -> logical chunk
        arow[i] = row;
        a_choff[i] = (m0 + ch * EPC) < a.Cout ? (unsigned)((m0 + ch * EPC) * ES) : kOOB;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int q = (i * NWAVE + wave) * 64 + lane;
        const int row = q / CPRB, slot = q % CPRB;
        const int ch = BF ? swz(row, slot) : slot;
        brow[i] = row;
        b_choff[i] = (n0 + ch * EPC) < a.Cin ? (unsigned)((n0 + ch * EPC) * ES) : kOOB;
    }

    auto issue = [&](int step, int stage) {
        char* sa = smem + stage * (TILEA + TILEB);
        char* sb = sa + TILEA;
        const int pbase = (step0 + step) * BKP;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int p = pbase + arow[i];
            unsigned va = kOOB;
            if (p < a.P && a_choff[i] != kOOB) va = (unsigned)p * (unsigned)(a.Cout * ES) + a_choff[i];
            bufload_lds16(yrs, sa + (i * NWAVE + wave) * 1024, va);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int p = pbase + brow[i];
            unsigned vb = kOOB;
            int rem, rx;
            const int b = fdiv(p, a.HoWo, a.inv_howo, &rem);
            const int oy = fdiv(rem, a.Wo, a.inv_wo, &rx);
            const int iy = oy * a.stride + oy_off, ix = rx * a.stride + ox_off;
            if (p < a.P && b_choff[i] != kOOB && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                vb = (unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)(a.Cin * ES) + b_choff[i];
            bufload_lds16(xrs, sb + (i * NWAVE + wave) * 1024, vb);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addresses (relative to the operand tile of a stage)
    const int g = lane >> 4, li = lane & 15;
    unsigned ra[TM][2], rb[TN][2];                    // bf16: [tile][half of the 8-pixel group]
    unsigned fa[TM], fb[TN];                          // f32 : [tile]
    if (BF) {
        const int q = li >> 2, p = li & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * g + 4 * h + q;                            // + 32*substep (the XOR is invariant)
#pragma unroll
            for (int t = 0; t < TM; ++t)
                ra[t][h] = row * ROWA + swz(row, (wm * TM + t) * 2 + (p >> 1)) * 16 + 8 * (p & 1);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                rb[t][h] = row * ROWB + swz(row, (wn * TN + t) * 2 + (p >> 1)) * 16 + 8 * (p & 1);
        }
    } else {
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = g * ROWA + ((wm * TM + t) * 16 + li) * 4;   // + 4 rows per substep
#pragma unroll
        for (int t = 0; t < TN; ++t) fb[t] = g * ROWB + ((wn * TN + t) * 16 + li) * 4;
    }

    if (nsteps > 0) issue(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nsteps) issue(s + 1, (s + 1) & 1);
        const char* sa = smem + (s & 1) * (TILEA + TILEB);
        const char* sb = sa + TILEA;
        if (BF) {
#pragma unroll
            for (int sub = 0; sub < BKP / 32; ++sub) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sa + ra[t][0] + sub * 32 * ROWA));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sa + ra[t][1] + sub * 32 * ROWA));
                    af[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sb + rb[t][0] + sub * 32 * ROWB));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sb + rb[t][1] + sub * 32 * ROWB));
                    bfr[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll 2
            for (int sub = 0; sub < BKP / 4; ++sub) {
                float af[TM], bfr[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const float*>(sa + fa[t] + sub * 4 * ROWA);
#pragma unroll
                for (int t = 0; t < TN; ++t) bfr[t] = *reinterpret_cast<const float*>(sb + fb[t] + sub * 4 * ROWB);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // partial[split][tap][co][ci]
    float* out = a.partial + ((size_t)blockIdx.y * a.ntaps + tap) * (size_t)a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = n0 + (wn * TN + j) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + (wm * TM + i) * 16 + 4 * g + r;
                if (co < a.Cout && ci < a.Cin) out[(size_t)co * a.Cin + ci] = acc[i][j][r];
            }
        }
}

// dw[co][ci][tap] = beta*dw + sum_s partial[s][tap][co][ci]   (fixed order; one thread per output element, the
// loads of four splits in flight at a time -- a serial chain of nsplit*taps loads per thread was latency-bound)
__global__ void __launch_bounds__(256) wgrad_fold_kernel(const float* __restrict__ partial, int nsplit, int ntaps,
                                                         int Cout, int Cin, float beta, float* __restrict__ dw) {
    const long long n = (long long)Cout * Cin;
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= n * ntaps) return;
    const int t = (int)(w / n);
    const long long o = w - (long long)t * n;
    const float* p = partial + (size_t)t * n + o;
    const size_t stride = (size_t)ntaps * n;
    float acc = 0.f;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        const float v0 = p[(size_t)s * stride], v1 = p[(size_t)(s + 1) * stride];
        const float v2 = p[(size_t)(s + 2) * stride], v3 = p[(size_t)(s + 3) * stride];
        acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; s < nsplit; ++s) acc += p[(size_t)s * stride];
    float* d = dw + o * ntaps + t;
    *d = beta != 0.f ? beta * *d + acc : acc;
}

struct Geom {
    int bkp, total_steps, nsplit, steps_per_split, n_mt, n_nt, ntaps;
    int big;     // 256 x 256 tile (8 waves) instead of 128 x 128 (4 waves)
};

int geometry(const ppn_wgrad_desc* d, Geom* g) {
    if (!d) return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: NULL descriptor");
    if (d->dtype != PPN_F32 && d->dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    const int epc = d->dtype == PPN_F32 ? 4 : 8;
    if (d->batch <= 0 || d->cin <= 0 || d->cout <= 0 || d->cin % epc || d->cout % epc)
        return ppn::fail(PPN_E_UNSUPPORTED, "ppn_conv_wgrad: cin (%d) and cout (%d) must be multiples of %d", d->cin,
                         d->cout, epc);
    if (d->ksize < 1 || d->ksize > 7 || d->stride < 1 || d->dilation < 1 || d->pad < 0)
        return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: bad filter geometry");
    const int eff = d->dilation * (d->ksize - 1) + 1;
    if ((d->in_h + 2 * d->pad - eff) / d->stride + 1 != d->out_h ||
        (d->in_w + 2 * d->pad - eff) / d->stride + 1 != d->out_w)
        return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: out size does not match in size / stride / pad");
    const long long P = (long long)d->batch * d->out_h * d->out_w;
    const long long es = d->dtype == PPN_F32 ? 4 : 2;
    if (P >= (1 << 24) || P * d->cout * es >= 0x80000000LL ||
        (long long)d->batch * d->in_h * d->in_w * d->cin * es >= 0x80000000LL)
        return ppn::fail(PPN_E_UNSUPPORTED, "ppn_conv_wgrad: tensor too large (pixels < 2^24, bytes < 2 GiB)");
    g->bkp = d->dtype == PPN_F32 ? 32 : 64;
    g->total_steps = (int)((P + g->bkp - 1) / g->bkp);
    static const char* force = getenv("PPN_WGRAD_TILE");                 // tuning knob: "128" / "256"
    g->big = force ? atoi(force) == 256 : (d->cout >= 256 && d->cin >= 256);
    const int bm = g->big ? 256 : 128;
    g->n_mt = (d->cout + bm - 1) / bm;
    g->n_nt = (d->cin + bm - 1) / bm;
    g->ntaps = d->ksize * d->ksize;
    const int tiles = g->n_mt * g->n_nt * g->ntaps;
    // 512 workgroups fill the GPU for two rounds (256-tile: one resident workgroup per CU) or one round (128-tile:
    // two per CU); round DOWN so that the count never spills a nearly empty extra round (540 workgroups measured
    // 1.3x slower than 504 on the 512x512x9 layers).
    int ns = 512 / tiles;
    ns = ns < 1 ? 1 : ns;
    const int max_ns = g->total_steps / 8 > 0 ? g->total_steps / 8 : 1;   // ... of at least 8 depth steps
    ns = ns > max_ns ? max_ns : ns;
    ns = ns > 64 ? 64 : ns;
    g->steps_per_split = (g->total_steps + ns - 1) / ns;
    g->nsplit = (g->total_steps + g->steps_per_split - 1) / g->steps_per_split;
    return PPN_OK;
}

template <typename T, int WM, int WN, int TM, int TN>
int launch(const WgArgs& a, dim3 grid, hipStream_t st) {
    static int lds_set = 0;
    constexpr int bkp = sizeof(T) == 2 ? 64 : 32;
    constexpr int lds = 2 * bkp * (WM * TM + WN * TN) * 16 * (int)sizeof(T);
    auto k = wgrad_kernel<T, WM, WN, TM, TN>;
    PPN_LDS_ONCE(lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<<<grid, 64 * WM * WN, lds, st>>>(a);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // namespace

extern "C" {

size_t ppn_conv_wgrad_workspace_bytes(const ppn_wgrad_desc* d) {
    Geom g;
    if (geometry(d, &g) != PPN_OK) return 0;
    if (ppn::stem_wgrad_supported(d)) return ppn::stem_wgrad_workspace_bytes(d);
    return (size_t)g.nsplit * g.ntaps * d->cout * d->cin * sizeof(float);
}

int ppn_conv_wgrad(const ppn_wgrad_desc* d, void* stream) {
    Geom g;
    if (int rc = geometry(d, &g)) return rc;
    if (!d->x || !d->dy || !d->dw || !d->workspace) return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: NULL pointer");
    if (ppn::stem_wgrad_supported(d)) {
        if (d->workspace_bytes < ppn::stem_wgrad_workspace_bytes(d))
            return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: workspace too small");
        return ppn::stem_wgrad_launch(d, (hipStream_t)stream);
    }
    const size_t need = (size_t)g.nsplit * g.ntaps * d->cout * d->cin * sizeof(float);
    if (d->workspace_bytes < need)
        return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: workspace %zu < %zu bytes", (size_t)d->workspace_bytes, need);
    const int es = d->dtype == PPN_F32 ? 4 : 2;
    WgArgs a{};
    a.x = (const char*)d->x;
    a.dy = (const char*)d->dy;
    a.partial = (float*)d->workspace;
    a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = d->cin;
    a.Ho = d->out_h; a.Wo = d->out_w; a.Cout = d->cout;
    a.ks = d->ksize; a.stride = d->stride; a.dil = d->dilation; a.pad = d->pad;
    a.P = d->batch * d->out_h * d->out_w;
    a.HoWo = d->out_h * d->out_w;
    a.inv_wo = 1.0f / (float)d->out_w;
    a.inv_howo = 1.0f / (float)a.HoWo;
    a.n_mt = g.n_mt; a.n_nt = g.n_nt; a.ntaps = g.ntaps;
    a.steps_per_split = g.steps_per_split;
    a.total_steps = g.total_steps;
    a.x_bytes = (unsigned)((size_t)d->batch * d->in_h * d->in_w * d->cin * es);
    a.dy_bytes = (unsigned)((size_t)a.P * d->cout * es);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.n_mt * g.n_nt * g.ntaps, g.nsplit);
    int rc;
    if (d->dtype == PPN_F32) rc = g.big ? launch<float, 4, 2, 4, 8>(a, grid, st) : launch<float, 2, 2, 4, 4>(a, grid, st);
    else rc = g.big ? launch<__bf16, 4, 2, 4, 8>(a, grid, st) : launch<__bf16, 2, 2, 4, 4>(a, grid, st);
    if (rc) return rc;
    const long long n = (long long)d->cout * d->cin * g.ntaps;
    wgrad_fold_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(a.partial, g.nsplit, g.ntaps, d->cout, d->cin, d->beta,
                                                              d->dw);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // extern "C"
