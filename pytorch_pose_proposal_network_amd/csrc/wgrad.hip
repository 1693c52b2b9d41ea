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
//   workgroup tile 128 (cout) x 128 (cin) for one tap, 4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles
//   depth step 64 pixels (bf16) / 32 pixels (f32): 2 x 16 KB per stage, 2 stages
//   LDS image: 256-byte pixel rows, 16-byte chunk c of row r at chunk slot c ^ (((r&3)<<2) | ((r>>2)&3)), applied on
//              the SOURCE side of the DMA (the DMA destination is lane-linear); conflict-free for the transposed reads
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
constexpr int BM = 128, BN = 128;
constexpr int kTileBytes = 16384;     // one operand tile of one stage
constexpr int kThreads = 256;

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

template <typename T>
__global__ void __launch_bounds__(kThreads) wgrad_kernel(WgArgs a) {
    constexpr int ES = sizeof(T), EPC = Elem<T>::EPC;
    constexpr bool BF = ES == 2;
    constexpr int CPR = BM / EPC;                    // 16-byte chunks per pixel row: 16 (bf16) / 32 (f32)
    constexpr int ROWB = BM * ES;                    // 256 / 512
    constexpr int BKP = kTileBytes / ROWB;           // pixels per depth step: 64 / 32
    constexpr int NPIECE = kTileBytes / (kThreads * 16);   // 4 DMA instructions per operand per step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

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

    // per-lane DMA geometry: piece i fills rows [i*RPP, +RPP) of the tile; this lane owns row prow[i], chunk pch[i]
    int prow[NPIECE];
    unsigned a_choff[NPIECE], b_choff[NPIECE];       // channel byte offset of the chunk, or kOOB when past C
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int q = (i * 4 + wave) * 64 + lane;
        const int row = q / CPR, slot = q % CPR;
        const int ch = BF ? (slot ^ (((row & 3) << 2) | ((row >> 2) & 3))) : slot;
        prow[i] = row;
        a_choff[i] = (m0 + ch * EPC) < a.Cout ? (unsigned)((m0 + ch * EPC) * ES) : kOOB;
        b_choff[i] = (n0 + ch * EPC) < a.Cin ? (unsigned)((n0 + ch * EPC) * ES) : kOOB;
    }

    auto issue = [&](int step, int stage) {
        char* sa = smem + stage * 2 * kTileBytes;
        char* sb = sa + kTileBytes;
        const int pbase = (step0 + step) * BKP;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int p = pbase + prow[i];
            const bool pv = p < a.P;
            unsigned va = kOOB, vb = kOOB;
            if (pv && a_choff[i] != kOOB) va = (unsigned)p * (unsigned)(a.Cout * ES) + a_choff[i];
            int rem, rx;
            const int b = fdiv(p, a.HoWo, a.inv_howo, &rem);
            const int oy = fdiv(rem, a.Wo, a.inv_wo, &rx);
            const int iy = oy * a.stride + oy_off, ix = rx * a.stride + ox_off;
            if (pv && b_choff[i] != kOOB && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                vb = (unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)(a.Cin * ES) + b_choff[i];
            bufload_lds16(yrs, sa + (i * 4 + wave) * 1024, va);
            bufload_lds16(xrs, sb + (i * 4 + wave) * 1024, vb);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addresses (relative to the operand tile of a stage)
    const int g = lane >> 4, li = lane & 15;
    unsigned ra[4][2], rb[4][2];                      // bf16: [tile][half of the 8-pixel group]
    unsigned fa[4], fb[4];                            // f32 : [tile]
    if (BF) {
        const int q = li >> 2, p = li & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * g + 4 * h + q;                       // + 32*substep
                const int x = ((row & 3) << 2) | ((row >> 2) & 3);       // invariant under row += 32
                const int ca = (wm * 8 + t * 2 + (p >> 1)) ^ x, cb = (wn * 8 + t * 2 + (p >> 1)) ^ x;
                ra[t][h] = row * ROWB + ca * 16 + 8 * (p & 1);
                rb[t][h] = row * ROWB + cb * 16 + 8 * (p & 1);
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t] = g * ROWB + (wm * 64 + t * 16 + li) * 4;              // + 4 rows per substep
            fb[t] = g * ROWB + (wn * 64 + t * 16 + li) * 4;
        }
    }

    if (nsteps > 0) issue(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nsteps) issue(s + 1, (s + 1) & 1);
        const char* sa = smem + (s & 1) * 2 * kTileBytes;
        const char* sb = sa + kTileBytes;
        if (BF) {
#pragma unroll
            for (int sub = 0; sub < BKP / 32; ++sub) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sa + ra[t][0] + sub * 32 * ROWB));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sa + ra[t][1] + sub * 32 * ROWB));
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sb + rb[t][0] + sub * 32 * ROWB));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(sb + rb[t][1] + sub * 32 * ROWB));
                    const s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    af[t] = __builtin_bit_cast(bf16x8, av);
                    bfr[t] = __builtin_bit_cast(bf16x8, bv);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll 2
            for (int sub = 0; sub < BKP / 4; ++sub) {
                float af[4], bfr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    af[t] = *reinterpret_cast<const float*>(sa + fa[t] + sub * 4 * ROWB);
                    bfr[t] = *reinterpret_cast<const float*>(sb + fb[t] + sub * 4 * ROWB);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // partial[split][tap][co][ci]
    float* out = a.partial + ((size_t)blockIdx.y * a.ntaps + tap) * (size_t)a.Cout * a.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = n0 + wn * 64 + j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wm * 64 + i * 16 + 4 * g + r;
                if (co < a.Cout && ci < a.Cin) out[(size_t)co * a.Cin + ci] = acc[i][j][r];
            }
        }
}

// dw[co][ci][tap] = beta*dw + sum_s partial[s][tap][co][ci]   (fixed order)
__global__ void __launch_bounds__(256) wgrad_fold_kernel(const float* __restrict__ partial, int nsplit, int ntaps,
                                                         int Cout, int Cin, float beta, float* __restrict__ dw) {
    const long long n = (long long)Cout * Cin;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    for (int t = 0; t < ntaps; ++t) {
        float acc = 0.f;
        for (int s = 0; s < nsplit; ++s) acc += partial[((size_t)s * ntaps + t) * n + o];
        float* d = dw + o * ntaps + t;
        *d = beta != 0.f ? beta * *d + acc : acc;
    }
}

struct Geom {
    int bkp, total_steps, nsplit, steps_per_split, n_mt, n_nt, ntaps;
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
    g->n_mt = (d->cout + BM - 1) / BM;
    g->n_nt = (d->cin + BN - 1) / BN;
    g->ntaps = d->ksize * d->ksize;
    const int tiles = g->n_mt * g->n_nt * g->ntaps;
    int ns = (1024 + tiles - 1) / tiles;                 // aim at ~1024 workgroups (2 resident per CU x 2 rounds) ...
    const int max_ns = g->total_steps / 8 > 0 ? g->total_steps / 8 : 1;   // ... of at least 8 depth steps
    ns = ns > max_ns ? max_ns : ns;
    ns = ns > 64 ? 64 : ns;
    g->steps_per_split = (g->total_steps + ns - 1) / ns;
    g->nsplit = (g->total_steps + g->steps_per_split - 1) / g->steps_per_split;
    return PPN_OK;
}

int lds_limit_f32 = 0, lds_limit_bf16 = 0;

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
    const int lds = 4 * kTileBytes;
    if (d->dtype == PPN_F32) {
        PPN_LDS_ONCE(lds_limit_f32, reinterpret_cast<const void*>(&wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        wgrad_kernel<float><<<grid, kThreads, lds, st>>>(a);
    } else {
        PPN_LDS_ONCE(lds_limit_bf16, reinterpret_cast<const void*>(&wgrad_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        wgrad_kernel<__bf16><<<grid, kThreads, lds, st>>>(a);
    }
    PPN_LAUNCH_CHECK();
    const long long n = (long long)d->cout * d->cin;
    wgrad_fold_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(a.partial, g.nsplit, g.ntaps, d->cout, d->cin, d->beta,
                                                              d->dw);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // extern "C"
