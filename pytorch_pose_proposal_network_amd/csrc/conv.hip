// Fused implicit-GEMM convolution for gfx950 (CDNA4, wave64, MFMA) -- the conv/BN/activation stack of
// PoseProposalNet.forward (model.py:104-136; drn.py:42-57, 77-97, 192-202).
//
// GEMM view (per launch):   D[channel][pixel] = sum_k  Wp[channel][k] * X[pixel][k]
//   k = (ky*ksize + kx)*Cin + ci,   pixel = (b*Ho + oy)*Wo + ox,   activations NHWC, weights packed
//   [cout_pad][k_total] so BOTH operands are K-contiguous 128-byte rows.
//
// Data movement: every K step a workgroup stages a BC x 128 B weight tile and a BP x 128 B activation
// tile into LDS with direct-to-LDS loads (global_load_lds_dwordx4; the per-lane SOURCE address does the
// im2col gather; padded taps read a device zero page), double buffered, one barrier per step.  LDS rows
// are XOR-swizzled on the source side (chunk ^= (row>>1)&7) so the ds_read_b128 fragment reads are
// bank-conflict free.  MFMA: v_mfma_f32_16x16x32_bf16 (bf16 mode) or v_mfma_f32_16x16x4_f32 (exact
// f32 mode used for the 1e-4 parity gate).  A = weights (rows = channels), B = activations (cols =
// pixels), so each lane ends with 4 consecutive channels of one pixel.
//
// Epilogue (through an LDS f32 tile so that global stores are 16 B per lane and coalesced):
//   v = act1(acc*scale1 + shift1) (+ residual);  out_raw = v;  out_act = act2(v*scale2 + shift2)
// which covers conv->BN->ReLU (drn.py:192-202), the pre-activation BasicBlock (drn.py:42-57: the
// second output is the NEXT block's relu(bn1(x)), so zero padding happens after BN/ReLU as in the
// reference), Bottleneck (drn.py:77-97) and the PPN neck/head incl. bias + sigmoid (model.py:113-134).
#include "conv_common.h"

namespace {

using namespace ppnconv;

// BP x BC output tile per 256-thread workgroup; WP x WC waves; SMALLC: Cin < BK (several taps per K step).
template <typename T, int BP, int BC, int WP, int WC, bool SMALLC>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvKArgs a) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;                    // 128-byte rows
    constexpr int ES = sizeof(T);
    constexpr int NXI = BP / 32;                   // activation-tile load instructions per thread
    constexpr int NWI = (BC + 31) / 32;            // weight-tile load instructions per thread (some waves idle if BC<32)
    constexpr int TP = BP / WP / 16, TC = BC / WC / 16;
    constexpr int STAGE = (BP + BC) * 128;         // bytes per stage buffer
    static_assert(WP * WC == 4, "4 waves");
    static_assert(TP >= 1 && TC >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / WC, wc = wave % WC;

    // XCD-aware tile order: consecutive logical tiles (same pixel tile, neighbouring pixel tiles) share an L2
    int ptile, ctile;
    {
        const int nb = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, loc = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        ptile = fast_div(logical, a.div_nct);
        ctile = logical - ptile * a.n_ctiles;
    }
    const int m0 = a.m_base + ptile * BP, c0 = ctile * BC;

    // ---- per-lane loader state -------------------------------------------------------------
    // LDS slot (row, s) holds global chunk s ^ ((row>>1)&7); for this lane the xor term is constant
    const int lrow = lane >> 3;                                      // row inside an 8-row wave-instruction
    const int chunk = (lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2));
    const int ntaps = a.ks * a.ks;
    int xbase[NXI];
    unsigned xmask[NXI];
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int row = (j * 4 + wave) * 8 + lrow;
        const int m = m0 + row;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int b = fast_div(mm, a.div_howo), rem = mm - b * a.HoWo;
        const int oy = fast_div(rem, a.div_wo), ox = rem - oy * a.Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        xbase[j] = ((b * a.H + iy0) * a.W + ix0) * a.Cin;
        // tap t = dy * ks + dx is in bounds iff its row and its column are (no per-tap division; ksize <= 5)
        unsigned mk = 0, cx = 0;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx)
            cx |= (dx < a.ks && (unsigned)(ix0 + dx * a.dil) < (unsigned)a.W) ? (1u << dx) : 0u;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
            mk |= (dy < a.ks && (unsigned)(iy0 + dy * a.dil) < (unsigned)a.H) ? (cx << (dy * a.ks)) : 0u;
        xmask[j] = vm ? mk : 0u;
    }
    const char* wptr[NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = (j * 4 + wave) * 8 + lrow;
        wptr[j] = a.wgt + ((size_t)(c0 + row) * a.Ktot + chunk * EPC) * ES;
    }
    int* s_tapoff = reinterpret_cast<int*>(smem + 2 * STAGE);        // SMALLC only: [32]
    if (SMALLC) {
        if (tid < 32) {
            const int dy = tid / a.ks, dx = tid - dy * a.ks;
            s_tapoff[tid] = (dy * a.dil * a.W + dx * a.dil) * a.Cin;
        }
        __syncthreads();
    }

    const int nsteps = a.Ktot / BK;
    // uniform-tap iteration state (Cin % BK == 0): the whole K step lies inside one tap
    int u_tap = 0, u_ci0 = 0, u_dy = 0, u_dx = 0;

    auto issue_loads = [&](int step, int buf) {
        char* xs = smem + buf * STAGE;
        char* ws = xs + BP * 128;
        if (!SMALLC) {
            const int tapoff = (u_dy * a.dil * a.W + u_dx * a.dil) * a.Cin + u_ci0 + chunk * EPC;
#pragma unroll
            for (int j = 0; j < NXI; ++j) {
                const bool ok = (xmask[j] >> u_tap) & 1u;
                const char* g = ok ? a.src + (ptrdiff_t)(xbase[j] + tapoff) * ES : a.zero;
                glds16(g, xs + (j * 4 + wave) * 1024);
            }
            u_ci0 += BK;
            if (u_ci0 >= a.Cin) {
                u_ci0 = 0; ++u_tap; ++u_dx;
                if (u_dx == a.ks) { u_dx = 0; ++u_dy; }
            }
        } else {
            const int k = step * BK + chunk * EPC;
            const int tap = k >> a.log2Cin, ci = k & (a.Cin - 1);
            const int toff = s_tapoff[tap & 31] + ci;
#pragma unroll
            for (int j = 0; j < NXI; ++j) {
                const bool ok = (tap < ntaps) && ((xmask[j] >> (tap & 31)) & 1u);
                const char* g = ok ? a.src + (ptrdiff_t)(xbase[j] + toff) * ES : a.zero;
                glds16(g, xs + (j * 4 + wave) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            if ((j * 4 + wave) * 8 < BC) glds16(wptr[j] + (size_t)step * BK * ES, ws + (j * 4 + wave) * 1024);
        }
    };

    // ---- per-lane fragment read offsets ------------------------------------------------------
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

    issue_loads(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) issue_loads(s + 1, buf ^ 1);
        const char* xs = smem + buf * STAGE + x_tile_off;
        const char* ws = smem + buf * STAGE + w_tile_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 wf[TC], xf[TP];
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const f32x4*>(ws + i * 16 * 128 + foff[ks]);
#pragma unroll
            for (int j = 0; j < TP; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xs + j * 16 * 128 + foff[ks]);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_step(acc[i][j], wf[i], xf[j], (T*)nullptr);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    float* ct = reinterpret_cast<float*>(smem);
    if (!a.nchw) {
        constexpr int LD = BC + 4;                                   // [pixel][channel] f32
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int px = wp * (BP / WP) + j * 16 + frow;
                const int ch = wc * (BC / WC) + i * 16 + 4 * fq;
                *reinterpret_cast<f32x4*>(ct + px * LD + ch) = acc[i][j];
            }
        __syncthreads();
        constexpr int TPP = BC / 8;                                  // threads per pixel (8 channels each)
        constexpr int PPP = 256 / TPP;                               // pixels per pass
        const int cg = tid % TPP, prow = tid / TPP;
        const int c = c0 + cg * 8;
        // none / ReLU / LeakyReLU(0.1) are all  max(t, t * slope)  with slope 1 / 0 / 0.1 (same values as the
        // compare + select form, signed zeros included); sigmoid keeps the general path
        const bool simple = a.act1 != PPN_ACT_SIGMOID && a.act2 != PPN_ACT_SIGMOID;
        const float slope1 = a.act1 == PPN_ACT_RELU ? 0.f : (a.act1 == PPN_ACT_LRELU ? 0.1f : 1.f);
        const float slope2 = a.act2 == PPN_ACT_RELU ? 0.f : (a.act2 == PPN_ACT_LRELU ? 0.1f : 1.f);
        if (c < a.Cout) {
            float s1[8], b1[8], s2[8], b2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s1[i] = a.scale1 ? a.scale1[c + i] : 1.f;
                b1[i] = a.shift1 ? a.shift1[c + i] : 0.f;
                s2[i] = a.scale2 ? a.scale2[c + i] : 1.f;
                b2[i] = a.shift2 ? a.shift2[c + i] : 0.f;
            }
            for (int pass = 0; pass < BP / PPP; ++pass) {
                const int px = pass * PPP + prow;
                const int m = m0 + px;
                if (m >= a.M) break;
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(ct + px * LD + cg * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(ct + px * LD + cg * 8 + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                const size_t off = ((size_t)m * a.Cout + c) * ES;
                if (simple) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float t1 = v[i] * s1[i] + b1[i];
                        v[i] = fmaxf(t1, t1 * slope1);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = apply_act(v[i] * s1[i] + b1[i], a.act1);
                }
                if (a.residual) {
                    float r[8];
                    load8<T>(a.residual + off, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += r[i];
                }
                if (a.out_raw) store8<T>(a.out_raw + off, v);
                if (a.out_act) {
                    float u[8];
                    if (simple) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float t2 = v[i] * s2[i] + b2[i];
                            u[i] = fmaxf(t2, t2 * slope2);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) u[i] = apply_act(v[i] * s2[i] + b2[i], a.act2);
                    }
                    store8<T>(a.out_act + off, u);
                }
            }
        }
    } else {
        // head: f32 NCHW [B, Cout, Ho*Wo] (model.py:136), pixel-contiguous rows
        constexpr int LD = BP + 4;                                   // [channel][pixel] f32
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int px = wp * (BP / WP) + j * 16 + frow;
                const int ch = wc * (BC / WC) + i * 16 + 4 * fq;
#pragma unroll
                for (int r = 0; r < 4; ++r) ct[(ch + r) * LD + px] = acc[i][j][r];
            }
        __syncthreads();
        constexpr int TPC = BP / 4;                                  // threads per channel row (4 pixels each)
        constexpr int CPP = 256 / TPC;                               // channels per pass
        const int pq = tid % TPC, crow = tid / TPC;
        const int m = m0 + 4 * pq;
        float* out = reinterpret_cast<float*>(a.out_raw);
        const bool vec = (a.HoWo & 3) == 0;
        for (int pass = 0; pass < BC / CPP; ++pass) {
            const int chl = pass * CPP + crow;
            const int c = c0 + chl;
            if (c >= a.Cout || m >= a.M) continue;
            const float s1 = a.scale1 ? a.scale1[c] : 1.f, b1 = a.shift1 ? a.shift1[c] : 0.f;
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(ct + chl * LD + 4 * pq);
            float v[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i] * s1 + b1, a.act1);
            if (vec) {
                const int b = fast_div(m, a.div_howo), p = m - b * a.HoWo;
                *reinterpret_cast<float4*>(out + ((size_t)b * a.Cout + c) * a.HoWo + p) =
                    make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mi = m + i;
                    if (mi < a.M) {
                        const int b = fast_div(mi, a.div_howo), p = mi - b * a.HoWo;
                        out[((size_t)b * a.Cout + c) * a.HoWo + p] = v[i];
                    }
                }
            }
        }
    }
}

template <typename T, int BP, int BC>
constexpr size_t conv_lds_bytes() {
    size_t stage = 2 * (size_t)(BP + BC) * 128 + 128;               // + tap table
    size_t epi_a = (size_t)BP * (BC + 4) * 4, epi_b = (size_t)BC * (BP + 4) * 4;
    size_t e = epi_a > epi_b ? epi_a : epi_b;
    return stage > e ? stage : e;
}

struct TileChoice {
    int bp, bc;
};

TileChoice choose_tile(int cout) {
    if (cout <= 16) return {256, 16};
    if (cout <= 32) return {256, 32};
    if (cout <= 64) return {128, 64};
    return {128, 128};
}

template <typename T, int BP, int BC, int WP, int WC>
int launch_tile(const ConvKArgs& a, bool smallc, hipStream_t st, const char** kname) {
    const size_t lds = conv_lds_bytes<T, BP, BC>();
    const int grid = a.n_ctiles * a.n_ptiles;
    // the name as rocprofv3 --kernel-trace prints it (a substring of the demangled symbol)
    static char names[2][96];
    if (!names[0][0]) {
        for (int i = 0; i < 2; ++i)
            snprintf(names[i], sizeof(names[i]), "conv_igemm_kernel<%s, %d, %d, %d, %d, %s>",
                     elem_name<T>(), BP, BC, WP, WC, i ? "true" : "false");
    }
    if (smallc) {
        auto k = conv_igemm_kernel<T, BP, BC, WP, WC, true>;
        if (kname) *kname = names[1];
        {
            static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
            PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds);
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    } else {
        auto k = conv_igemm_kernel<T, BP, BC, WP, WC, false>;
        if (kname) *kname = names[0];
        {
            static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
            PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds);
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

template <typename T>
int launch_dtype(const ConvKArgs& a, bool smallc, TileChoice tc, hipStream_t st, const char** kname) {
    if (tc.bc == 16) return launch_tile<T, 256, 16, 4, 1>(a, smallc, st, kname);
    if (tc.bc == 32) return launch_tile<T, 256, 32, 4, 1>(a, smallc, st, kname);
    if (tc.bc == 64) return launch_tile<T, 128, 64, 2, 2>(a, smallc, st, kname);
    return launch_tile<T, 128, 128, 2, 2>(a, smallc, st, kname);
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// weight packing: [cout,cin,k,k] f32 -> [cout_pad][k_total] T with k = (ky*ks+kx)*cin + ci
__device__ __forceinline__ float pack_value(const float* __restrict__ w, size_t i, int cout, int cin, int ntaps, int ktot,
                                            int korder, int kstep, int transposed) {
    const int co = (int)(i / ktot), k = (int)(i % ktot);
    if (co >= cout || k >= ntaps * cin) return 0.f;
    int tap, ci;
    if (korder == 0) {
        tap = k / cin; ci = k % cin;
    } else if (korder == 2) {                              // reference layout [cout][cin][k][k]
        ci = k / ntaps; tap = k % ntaps;
    } else {
        const int blk = k / kstep, within = k % kstep;     // blk = cchunk*ntaps + tap
        tap = blk % ntaps; ci = (blk / ntaps) * kstep + within;
    }
    // transposed: the input-gradient convolution's weight w'[co][ci][tap] = w[ci][co][last - tap] of the
    // forward weight w [cin][cout][k][k] (channel roles swapped, filter rotated by 180 degrees)
    return transposed ? w[((size_t)ci * cout + co) * ntaps + (ntaps - 1 - tap)] : w[((size_t)co * cin + ci) * ntaps + tap];
}

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin, int ks,
                                   int cout_pad, int ktot, int korder, int kstep, int transposed) {
    const size_t n = (size_t)cout_pad * ktot;
    const int ntaps = ks * ks;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (T)pack_value(w, i, cout, cin, ntaps, ktot, korder, kstep, transposed);
}

// ppn_pack_table_run: every pack of a training iteration in one launch.  A workgroup owns kPackChunk consecutive
// elements of ONE entry; it finds the entry by bisection over the entries' first workgroup (uniform, scalar loads).
struct PackItemDev {
    const float* w;
    void* out;
    unsigned long long n;                 // cout_pad * k_total
    int cout, cin, ntaps, ktot, korder, kstep, transposed, otype;   // otype: 0 f32, 1 bf16, 2 f16
    int first_block, tiled;               // tiled: pack_tile() units of 8 rows x one 64-channel chunk (see there)
};
static_assert(sizeof(PackItemDev) == PPN_PACK_ITEM_BYTES, "ppn.h: PPN_PACK_ITEM_BYTES");
constexpr int kPackChunk = 2048;

// The MFMA layers' 16-bit packs (k_order 1, 64-channel chunks, no K padding, <= 9 taps): a workgroup owns 8 packed rows x one
// 64-channel chunk = 8 x 64 x ntaps elements, which are CONTIGUOUS runs on both sides -- forward layout: 8 runs of 64 * ntaps
// floats in, 8 runs of ntaps * 64 halves out; input-gradient layout (channel roles swapped, filter rotated): 64 runs of 8 * ntaps
// floats in -- and differ by a [channel][tap] <-> [tap][channel] transposition, done in LDS.  The element-wise form below gathers
// 8 floats at a stride of ntaps (or cout * ntaps) floats per thread: ~1.4 TB/s of useful traffic, 165 us per training iteration
// for DRN-D-22's two layouts; same values, same rounding.
constexpr int kTileRows = 8, kTileMaxTaps = 9;
// NT: the tap count as a compile-time constant (1 and 9: the index arithmetic is divisions by nt -- by run-time values it costs
// more than the memory traffic), 0 = any
template <int NT>
__device__ __forceinline__ void pack_tile(const PackItemDev& it, int unit, float* tile) {
    const int t = threadIdx.x, nt = NT > 0 ? NT : it.ntaps;
    const int nchunks = it.cin / 64;
    const int co0 = (unit / nchunks) * kTileRows, ci0 = (unit % nchunks) * 64;
    const int E = kTileRows * 64 * nt;
    if (!it.transposed) {
        for (int e = t; e < E; e += 256) {
            const int r = e / (64 * nt), off = e - r * (64 * nt);
            tile[e] = co0 + r < it.cout ? it.w[((size_t)(co0 + r) * it.cin + ci0) * nt + off] : 0.f;
        }
    } else {
        for (int e = t; e < E; e += 256) {
            const int ci = e / (kTileRows * nt), off = e - ci * (kTileRows * nt);
            const int r = off / nt, tp = off - r * nt;
            const float v = co0 + r < it.cout ? it.w[((size_t)(ci0 + ci) * it.cout + co0) * nt + off] : 0.f;
            tile[(r * 64 + ci) * nt + (nt - 1 - tp)] = v;
        }
    }
    __syncthreads();
    for (int g = t; g < kTileRows * nt * 8; g += 256) {
        const int r = g / (nt * 8), rem = g - r * (nt * 8), tap = rem >> 3, c8 = rem & 7;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[(r * 64 + c8 * 8 + j) * nt + tap];
        const size_t o = (size_t)(co0 + r) * it.ktot + (size_t)((ci0 / 64) * nt + tap) * 64 + c8 * 8;
        if (it.otype == 1) {
            ppnconv::store8<__bf16>(reinterpret_cast<char*>(static_cast<__bf16*>(it.out) + o), v);
        } else {
            typedef __attribute__((ext_vector_type(8))) _Float16 h8;
            h8 q;
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = (_Float16)v[j];     // as ppn_pack_weight converts (no clamp)
            *reinterpret_cast<h8*>(static_cast<_Float16*>(it.out) + o) = q;
        }
    }
}

__global__ void __launch_bounds__(256) pack_table_kernel(const PackItemDev* __restrict__ tab, int n_items) {
    __shared__ float tile[kTileRows * 64 * kTileMaxTaps];
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackItemDev it = tab[lo];
    if (it.tiled) {
        const int unit = (int)blockIdx.x - it.first_block;
        if (it.ntaps == 9) pack_tile<9>(it, unit, tile);
        else if (it.ntaps == 1) pack_tile<1>(it, unit, tile);
        else pack_tile<0>(it, unit, tile);
        return;
    }
    const size_t base = (size_t)((int)blockIdx.x - it.first_block) * kPackChunk;
    // 8 consecutive k of one output row share their tap when the channel index runs fastest over a multiple of 8 (every
    // MFMA layer: k_order 1 with 64-channel chunks, k_order 0 with cin % 8 == 0): ONE index decomposition (six integer
    // divisions by run-time values -- the per-element form is instruction-bound at ~0.2 ms for DRN-D-22's 33 M packed
    // elements) and one 16- / 32-byte store per 8 elements.
    const bool wide = it.ktot % 8 == 0 && ((it.korder == 1 && it.kstep % 8 == 0) || (it.korder == 0 && it.cin % 8 == 0));
    if (wide) {
        const size_t i = base + (size_t)threadIdx.x * 8;
        if (i >= it.n) return;
        const int co = (int)(i / it.ktot), k = (int)(i % it.ktot);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (co < it.cout && k < it.ntaps * it.cin) {
            int tap, ci;
            if (it.korder == 0) {
                tap = k / it.cin; ci = k % it.cin;
            } else {
                const int blk = k / it.kstep, within = k % it.kstep;
                tap = blk % it.ntaps; ci = (blk / it.ntaps) * it.kstep + within;
            }
            const float* src = it.transposed ? it.w + ((size_t)ci * it.cout + co) * it.ntaps + (it.ntaps - 1 - tap)
                                             : it.w + ((size_t)co * it.cin + ci) * it.ntaps + tap;
            const size_t stride = it.transposed ? (size_t)it.cout * it.ntaps : (size_t)it.ntaps;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[j * stride];
        }
        if (it.otype == 0) {
            float4* o = reinterpret_cast<float4*>(static_cast<float*>(it.out) + i);
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            o[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else if (it.otype == 1) {
            ppnconv::store8<__bf16>(reinterpret_cast<char*>(static_cast<__bf16*>(it.out) + i), v);
        } else {
            typedef __attribute__((ext_vector_type(8))) _Float16 h8;
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)v[j];     // as ppn_pack_weight converts (no clamp)
            *reinterpret_cast<h8*>(static_cast<_Float16*>(it.out) + i) = o;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kPackChunk / 256; ++j) {
        const size_t i = base + j * 256 + threadIdx.x;
        if (i >= it.n) break;
        const float v = pack_value(it.w, i, it.cout, it.cin, it.ntaps, it.ktot, it.korder, it.kstep, it.transposed);
        if (it.otype == 0) static_cast<float*>(it.out)[i] = v;
        else if (it.otype == 1) static_cast<__bf16*>(it.out)[i] = (__bf16)v;
        else static_cast<_Float16*>(it.out)[i] = (_Float16)v;
    }
}

// PPN_F16X3 weight packing (ppn_pack_weight_x3): row = [slab][copy 0..2][tap][64 channels], ws = w * 2^s
__global__ void pack_weight_x3_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int cout, int cin, int ks,
                                      int cout_pad, float scale) {
    const int ntaps = ks * ks, kreal = ntaps * cin;
    const size_t n = (size_t)cout_pad * 3 * kreal;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / (3 * (size_t)kreal)), k = (int)(i % (3 * (size_t)kreal));
        const int blk = k / 64, within = k % 64;                      // blk = (slab * 3 + copy) * ntaps + tap
        const int tap = blk % ntaps, sc = blk / ntaps, copy = sc % 3, ci = (sc / 3) * 64 + within;
        float v = 0.f;
        if (co < cout) {
            const float ws = w[((size_t)co * cin + ci) * ntaps + tap] * scale;
            const float hi = (float)(_Float16)ws;
            v = copy == 0 ? hi : (copy == 1 ? ws - hi : ws * ppnconv::kX3LoInv);
        }
        out[i] = (_Float16)v;
    }
}

// f32 NHWC -> PPN_F16X3 pairs (ppn_split_f16x3): 8 channels per thread
__global__ void __launch_bounds__(256) split_x3_kernel(const float* __restrict__ src, char* __restrict__ dst, size_t pixels,
                                                       int channels) {
    const int cpr = channels / 8;
    const size_t n = pixels * cpr;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i / cpr;
        const int c = (int)(i % cpr) * 8;
        float v[8];
        ppnconv::load8<float>(reinterpret_cast<const char*>(src + px * channels + c), v);
        ppnconv::store8_x3(dst + (px * 2 * channels + c) * 2, (size_t)channels * 2, v);
    }
}

}  // namespace

namespace ppn {
int split_x3_launch(const float* src, long long pixels, int channels, void* dst, hipStream_t st) {
    if (!src || !dst || pixels < 1 || channels < 8 || channels % 8 != 0)
        return ppn::fail(PPN_E_INVALID, "ppn_split_f16x3: NULL pointer or channels %% 8 != 0");
    const size_t n = (size_t)pixels * (channels / 8);
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(split_x3_kernel, dim3(blocks), dim3(256), 0, st, src, static_cast<char*>(dst), (size_t)pixels, channels);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
}  // namespace ppn

extern "C" int ppn_split_f16x3(const float* src, int64_t pixels, int32_t channels, void* dst, void* stream) {
    return ppn::split_x3_launch(src, pixels, channels, dst, static_cast<hipStream_t>(stream));
}

extern "C" int ppn_pack_weight_x3(const float* w, int32_t cout, int32_t cin, int32_t ksize, int32_t cout_pad,
                                  int32_t scale_log2, void* out, void* stream) {
    if (!w || !out || cout < 1 || cin < 64 || cin % 64 != 0 || ksize < 1 || cout_pad < cout || scale_log2 < -60 || scale_log2 > 60)
        return ppn::fail(PPN_E_INVALID, "ppn_pack_weight_x3: bad arguments (cin %% 64 == 0)");
    const size_t n = (size_t)cout_pad * 3 * ksize * ksize * cin;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_weight_x3_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       static_cast<_Float16*>(out), cout, cin, ksize, cout_pad, ldexpf(1.f, scale_log2));
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

namespace ppn {
int conv_launch(const ppn_conv_desc* d, hipStream_t st, const char** kname);
int head_limb_launch(const ppn_conv_desc* d, long long m_lo, long long m_hi, hipStream_t st, const char** kname);
bool conv64_supported(const ppn_conv_desc* d);
int conv64_launch(const ppn_conv_desc* d, hipStream_t st, const char** kname);
bool stem3x3_supported(int cin, int cout, int ksize, int stride, int dilation, int pad);
int stem3x3_launch(int dtype, const void* src, int batch, int h, int w, int cout, int stride, const float* weight,
                   const float* scale1, const float* shift1, const float* scale2, const float* shift2, void* out_raw,
                   void* out_act, hipStream_t st, const char** kname);
}

extern "C" int ppn_conv_tiling(int32_t dtype, int32_t cin, int32_t cout, int32_t ksize, int32_t* k_step,
                               int32_t* cout_tile, int32_t* k_order) {
    if (dtype != PPN_F32 && dtype != PPN_BF16 && dtype != PPN_F16 && dtype != PPN_F16X3)
        return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    if (cin < 1 || cout < 1 || ksize < 1) return ppn::fail(PPN_E_INVALID, "bad conv shape");
    const int bk = dtype == PPN_F32 ? 32 : 64;
    if (dtype == PPN_F16X3 && (cin % bk != 0 || cout < 64))
        return ppn::fail(PPN_E_UNSUPPORTED, "PPN_F16X3 needs cin %% 64 == 0 and cout >= 64 (run the layer as PPN_F32)");
    if (cin == 16 && (cout == 16 || cout == 32) && ksize == 3) {
        // direct small-channel kernel (stem3x3.hip): weights stay in the reference layout, f32, unpadded
        if (k_step) *k_step = 144;
        if (cout_tile) *cout_tile = cout;
        if (k_order) *k_order = 2;
        return PPN_OK;
    }
    if (k_step) *k_step = bk;
    BigTile bt;
    const bool big = (cin % bk == 0) && big_tile_for(cout, 1, &bt);
    // pad granularity = the LARGEST channel tile the launcher may pick for this Cout (it chooses per problem size)
    if (cout_tile) *cout_tile = big ? (cout >= 256 ? 256 : (cout >= 128 ? 128 : 64)) : choose_tile(cout).bc;
    if (k_order) *k_order = big ? 1 : 0;
    return PPN_OK;
}

int ppn::conv_launch(const ppn_conv_desc* d, hipStream_t st, const char** kname) {
    if (!d) return ppn::fail(PPN_E_INVALID, "conv desc is NULL");
    if (d->stats_mode != 0) {
        // BatchNorm statistics from the epilogue: a request, not a demand -- *stats_tiles says whether this launch delivered
        if (d->stats_mode < 1 || d->stats_mode > 2 || !d->stats_partial || !d->stats_tiles)
            return ppn::fail(PPN_E_INVALID, "stats_mode 1 or 2 needs stats_partial and stats_tiles");
        if (d->stats_mode == 2 && (!d->stats_x || !d->stats_gamma || !d->stats_beta || !d->stats_mean || !d->stats_rstd ||
                                   d->stats_act < PPN_ACT_NONE || d->stats_act > PPN_ACT_LRELU))
            return ppn::fail(PPN_E_INVALID, "stats_mode 2 needs stats_x, gamma, beta, mean, rstd and act none / relu / lrelu");
        *d->stats_tiles = 0;
    }
    if (d->dtype != PPN_F32 && d->dtype != PPN_BF16 && d->dtype != PPN_F16 && d->dtype != PPN_F16X3)
        return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    const bool x3 = d->dtype == PPN_F16X3;
    const int bk = d->dtype == PPN_F32 ? 32 : 64, epc = d->dtype == PPN_F32 ? 4 : 8;
    if (d->batch < 1 || d->in_h < 1 || d->in_w < 1 || d->cin < 1 || d->cout < 1)
        return ppn::fail(PPN_E_INVALID, "bad conv geometry");
    if (d->ksize < 1 || d->ksize > 5) return ppn::fail(PPN_E_UNSUPPORTED, "ksize %d not supported by the generic kernel", d->ksize);
    if (d->cin % epc != 0) return ppn::fail(PPN_E_UNSUPPORTED, "cin %d must be a multiple of %d", d->cin, epc);
    const int eff = d->dilation * (d->ksize - 1) + 1;
    const int oh = (d->in_h + 2 * d->pad - eff) / d->stride + 1, ow = (d->in_w + 2 * d->pad - eff) / d->stride + 1;
    if (oh != d->out_h || ow != d->out_w)
        return ppn::fail(PPN_E_INVALID, "out size %dx%d inconsistent with %dx%d", d->out_h, d->out_w, oh, ow);
    if (ppn::stem3x3_supported(d->cin, d->cout, d->ksize, d->stride, d->dilation, d->pad) && d->k_total == 144 &&
        d->cout_pad == d->cout && !d->residual && !d->out_nchw_f32 &&
        ((d->act1 == PPN_ACT_RELU && d->scale1 && d->shift1) ||
         (d->act1 == PPN_ACT_NONE && !d->scale1 && !d->shift1 && !d->out_act)) &&
        (!d->out_act || d->act2 == PPN_ACT_RELU)) {
        if (!d->src || !d->weight) return ppn::fail(PPN_E_INVALID, "NULL src/weight");
        if (oh != d->out_h || ow != d->out_w) return ppn::fail(PPN_E_INVALID, "inconsistent output size");
        return ppn::stem3x3_launch(d->dtype, d->src, d->batch, d->in_h, d->in_w, d->cout, d->stride,
                                   static_cast<const float*>(d->weight), d->scale1, d->shift1, d->scale2, d->shift2,
                                   d->out_raw, d->out_act, st, kname);
    }
    const bool smallc = (d->cin % bk) != 0;
    const int log2c = ilog2_exact(d->cin);
    if (smallc && log2c < 0) return ppn::fail(PPN_E_UNSUPPORTED, "cin %d < K step must be a power of two", d->cin);
    const int kreal = d->ksize * d->ksize * d->cin;
    int kpad = ((kreal + bk - 1) / bk) * bk;
    if (x3) {
        if (smallc || d->cout < 64 || d->src2 || d->limb_edge_pad)
            return ppn::fail(PPN_E_UNSUPPORTED, "PPN_F16X3: cin %% 64 == 0, cout >= 64, no fused shortcut, no edge tile");
        kpad *= 3;                                                    // three virtual slabs per real one
    }
    const int kmain = kpad;
    if (d->src2) {
        if (d->cin2 < bk || d->cin2 % bk != 0 || d->stride2 < 1 || d->in2_h < 1 || d->in2_w < 1 || d->scale1 ||
            d->out_nchw_f32 || smallc || (d->out_h - 1) * d->stride2 >= d->in2_h || (d->out_w - 1) * d->stride2 >= d->in2_w)
            return ppn::fail(PPN_E_INVALID, "fused shortcut: cin2 must be a multiple of the K step, scale1 NULL, "
                                            "and (out-1)*stride2 inside the second source");
        kpad += d->cin2;
    }
    if (d->k_total != kpad) return ppn::fail(PPN_E_INVALID, "k_total %d != %d", d->k_total, kpad);
    const long long m_all = (long long)d->batch * d->out_h * d->out_w;
    if (m_all > 0x7fffffffLL) return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for 32-bit indexing");
    long long m_lo = 0, m_hi = m_all;
    if (d->m_count != 0) {
        if (d->m_begin < 0 || d->m_count < 0 || (long long)d->m_begin + d->m_count > m_all)
            return ppn::fail(PPN_E_INVALID, "pixel range [%d, +%d) outside the %lld output pixels", d->m_begin, d->m_count, m_all);
        m_lo = d->m_begin; m_hi = m_lo + d->m_count;
        // The NCHW f32 epilogues store four consecutive pixels of one channel plane as a float4 when HoWo % 4 == 0:
        // a cut that is not a multiple of 4 would store misaligned, write up to 3 pixels of the neighbouring range
        // (a race with the launch that owns it) or run into the next channel plane.
        if (d->out_nchw_f32 && ((d->out_h * d->out_w) & 3) == 0 && ((m_lo & 3) != 0 || ((m_hi & 3) != 0 && m_hi != m_all)))
            return ppn::fail(PPN_E_INVALID, "NCHW output: pixel range [%d, +%d) must begin on a multiple of 4 and end on one "
                                            "(or at the last pixel)", d->m_begin, d->m_count);
    } else if (!smallc && d->stats_mode == 0) {
        // whole tensor: two launches with different tiles where that saves a round of workgroups (ppn_conv_split)
        const long long cut = big_split_for(d->cout, m_all);
        if (cut > 0 && cut < m_all) {
            ppn_conv_desc part = *d;
            part.m_begin = 0; part.m_count = (int32_t)cut;
            const int rc = ppn::conv_launch(&part, st, kname);
            if (rc != PPN_OK) return rc;
            part.m_begin = (int32_t)cut; part.m_count = (int32_t)(m_all - cut);
            return ppn::conv_launch(&part, st, nullptr);
        }
    }
    // 64 -> 64 3x3 stride 1 in the 16-bit modes: the register-resident filter bank kernel (conv64.hip), same results
    if (ppn::conv64_supported(d) && d->in_h == d->out_h && d->in_w == d->out_w && !(d->flags & PPN_CONV_OUT_BF16))
        return ppn::conv64_launch(d, st, kname);
    const long long m = m_hi - m_lo;                                  // pixels of THIS launch: the tile is chosen for them
    TileChoice tc = choose_tile(d->cout);
    BigTile bt{0, 0};
    const bool big = !smallc && big_tile_for(d->cout, m, &bt, kpad / bk, (d->flags & PPN_CONV_SHARED_GPU) != 0);
    if (d->limb_edge_pad != 0) {
        // edge-aligned limb tile (conv_head.hip): rows [e * limb_edge_pad, +limb_window) of the packed weight / shift1
        // hold edge e's window, the rest of each edge's rows are padding
        if (d->limb_edge_pad != kEdgeTileBC || smallc || d->src2 || !d->argmax_keys || !d->out_nchw_f32 || d->out_raw ||
            d->unary_out || d->unary_channels != 0 || d->residual || d->out_act || d->limb_window < 1 ||
            d->limb_window > d->limb_edge_pad || d->cout % d->limb_window != 0 ||
            d->cout_pad != d->cout / d->limb_window * d->limb_edge_pad || d->act1 != PPN_ACT_SIGMOID)
            return ppn::fail(PPN_E_INVALID, "limb_edge_pad: needs %d rows per edge (window <= that), cout = E * window, "
                                            "cout_pad = E * limb_edge_pad, sigmoid, argmax_keys only (no out_raw / unary_out)",
                             kEdgeTileBC);
        if (!d->src || !d->weight) return ppn::fail(PPN_E_INVALID, "NULL src/weight");
        return ppn::head_limb_launch(d, m_lo, m_hi, st, kname);
    }
    const int bc = big ? bt.bc : tc.bc, bp = big ? bt.bp : tc.bp;
    if (d->cout_pad % bc != 0 || d->cout_pad < d->cout)
        return ppn::fail(PPN_E_INVALID, "cout_pad %d must be a multiple of %d and >= cout", d->cout_pad, bc);
    if (!d->src || !d->weight || !d->zero_page) return ppn::fail(PPN_E_INVALID, "NULL src/weight/zero_page");
    if (!d->out_raw && !d->out_act && !d->argmax_keys) return ppn::fail(PPN_E_INVALID, "conv has no output");
    if (d->argmax_keys && !d->limb_edge_pad &&
        (!d->out_nchw_f32 || !d->unary_out || d->unary_channels < 0 || d->limb_window < 1 ||
         d->unary_channels > d->cout || (d->cout - d->unary_channels) % d->limb_window != 0))
        return ppn::fail(PPN_E_INVALID, "fused arg-max needs the NCHW head mode, unary_out and cout = unary + E*window");
    if (d->argmax_keys && d->act1 != PPN_ACT_SIGMOID)
        return ppn::fail(PPN_E_INVALID, "fused arg-max keys assume non-negative (sigmoid) outputs");
    if (!d->out_nchw_f32 && (d->act1 == PPN_ACT_SIGMOID || d->act2 == PPN_ACT_SIGMOID))
        return ppn::fail(PPN_E_UNSUPPORTED, "sigmoid is only implemented for the NCHW head output");
    if (d->out_nchw_f32) {
        if (d->residual || d->out_act || (!d->out_raw && !d->argmax_keys))
            return ppn::fail(PPN_E_UNSUPPORTED, "NCHW head output supports out_raw / fused arg-max only");
    } else if (d->cout % 8 != 0) {
        return ppn::fail(PPN_E_UNSUPPORTED, "NHWC output needs cout %% 8 == 0 (got %d)", d->cout);
    }
    const long long in_elems = (long long)d->batch * d->in_h * d->in_w * d->cin * (x3 ? 2 : 1);
    if (in_elems > 0x7fffffffLL) return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for 32-bit indexing");
    ConvKArgs a;
    a.src = static_cast<const char*>(d->src);
    a.wgt = static_cast<const char*>(d->weight);
    a.scale1 = d->scale1; a.shift1 = d->shift1;
    a.residual = static_cast<const char*>(d->residual);
    a.out_raw = static_cast<char*>(d->out_raw);
    a.scale2 = d->scale2; a.shift2 = d->shift2;
    a.out_act = static_cast<char*>(d->out_act);
    a.zero = static_cast<const char*>(d->zero_page);
    a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = x3 ? 2 * d->cin : d->cin; a.Ho = d->out_h; a.Wo = d->out_w; a.Cout = d->cout;
    a.ks = d->ksize; a.stride = d->stride; a.dil = d->dilation; a.pad = d->pad;
    a.Ktot = d->k_total; a.M = (int)m_hi; a.m_base = (int)m_lo; a.HoWo = d->out_h * d->out_w;
    a.div_howo = make_fastdiv((unsigned)a.HoWo); a.div_wo = make_fastdiv((unsigned)d->out_w);
    a.act1 = d->act1; a.act2 = d->act2; a.nchw = d->out_nchw_f32;
    a.log2Cin = log2c < 0 ? 0 : log2c;
    a.n_ctiles = d->cout_pad / bc;
    a.div_nct = make_fastdiv((unsigned)a.n_ctiles);
    a.n_ptiles = (int)((m + bp - 1) / bp);
    a.src2 = static_cast<const char*>(d->src2);
    a.H2 = d->src2 ? d->in2_h : 1; a.W2 = d->src2 ? d->in2_w : 1; a.Cin2 = d->src2 ? d->cin2 : 0;
    a.stride2 = d->src2 ? d->stride2 : 1;
    a.nsteps_main = kmain / bk;
    a.src2_bytes = d->src2 ? (unsigned)((size_t)d->batch * d->in2_h * d->in2_w * d->cin2 * (d->dtype == PPN_F32 ? 4 : 2)) : 0u;
    if (d->src2 && (size_t)d->batch * d->in2_h * d->in2_w * d->cin2 * 4 >= 0x7fffff00ull)
        return ppn::fail(PPN_E_UNSUPPORTED, "second source too large for the buffer-addressed conv kernel");
    a.unary_out = d->unary_out;
    a.amax_keys = reinterpret_cast<unsigned long long*>(d->argmax_keys);
    a.unary_ch = d->unary_channels;
    a.window = d->limb_window;
    a.lo_off = x3 ? d->cin * 2 : 0;                                   // source pixel = [hi(cin) | lo'(cin)] halves
    a.out_plain = (d->flags & PPN_CONV_X3_PLAIN_OUT) ? 1 : 0;
    if (a.out_plain && (!x3 || d->out_nchw_f32))
        return ppn::fail(PPN_E_UNSUPPORTED, "PPN_CONV_X3_PLAIN_OUT: a PPN_F16X3 launch with NHWC outputs");
    a.pf_ptr = static_cast<const char*>(d->prefetch);
    a.pf_lines = (d->prefetch && d->prefetch_bytes > 0) ? (unsigned)std::min<long long>(d->prefetch_bytes / 128, 1 << 24) : 0u;
    a.pf_per_wg = 0;                                                  // set by launch_big (it knows the grid)
    a.out_bf16 = (d->flags & PPN_CONV_OUT_BF16) ? 1 : 0;
    a.st_partial = nullptr; a.st_mode = 0; a.st_act = 0; a.st_x = nullptr;
    a.st_gamma = a.st_beta = a.st_mean = a.st_rstd = nullptr;
    if (d->stats_mode != 0 && big && big_stats_ok(d->dtype, bt) && !d->out_nchw_f32 && !d->residual && !d->out_act && d->out_raw &&
        d->cout % 8 == 0 && !a.out_bf16 && !d->src2 && !d->argmax_keys && d->m_count == 0 && a.n_ptiles <= 1024) {
        // the conditions of the single-output 16-bit epilogue (conv_big.hip `fast`) + one launch over the whole tensor +
        // no more pixel tiles than a BatchNorm workspace has partial blocks (train.hip kMaxBlocks)
        a.st_partial = static_cast<double*>(d->stats_partial);
        a.st_mode = d->stats_mode; a.st_act = d->stats_act;
        a.st_x = static_cast<const char*>(d->stats_x);
        a.st_gamma = d->stats_gamma; a.st_beta = d->stats_beta; a.st_mean = d->stats_mean; a.st_rstd = d->stats_rstd;
        *d->stats_tiles = a.n_ptiles;
    }
    if (a.out_bf16 && (d->dtype != PPN_F16 || !big || d->out_nchw_f32))
        return ppn::fail(PPN_E_UNSUPPORTED, "PPN_CONV_OUT_BF16: a PPN_F16 launch of the large-tile kernel with NHWC outputs");
    if (x3 && !big) return ppn::fail(PPN_E_UNSUPPORTED, "PPN_F16X3 is implemented by the large-tile kernel only");
    if (big) return launch_big(a, d->dtype, bt, st, kname);
    if (d->argmax_keys) return ppn::fail(PPN_E_UNSUPPORTED, "fused arg-max is implemented by the large-tile kernel only");
    if (d->src2) return ppn::fail(PPN_E_UNSUPPORTED, "the fused shortcut is implemented by the large-tile kernel only");
    if (d->dtype == PPN_F32) return launch_dtype<float>(a, smallc, tc, st, kname);
    if (d->dtype == PPN_F16) return launch_dtype<_Float16>(a, smallc, tc, st, kname);
    return launch_dtype<__bf16>(a, smallc, tc, st, kname);
}

extern "C" int ppn_conv_split(int32_t dtype, int32_t cin, int32_t cout, int64_t m, int64_t* m_split) {
    if (dtype != PPN_F32 && dtype != PPN_BF16 && dtype != PPN_F16 && dtype != PPN_F16X3)
        return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    if (cin < 1 || cout < 1 || m < 1 || !m_split) return ppn::fail(PPN_E_INVALID, "ppn_conv_split: bad arguments");
    const int bk = dtype == PPN_F32 ? 32 : 64;
    const long long cut = (cin % bk == 0) ? big_split_for(cout, m) : 0;
    *m_split = (cut > 0 && cut < m) ? cut : 0;
    return PPN_OK;
}

static thread_local const char* g_last_conv_kernel = "";

extern "C" int ppn_conv2d_fused(const ppn_conv_desc* d, void* stream) {
    const char* kn = nullptr;
    const int rc = ppn::conv_launch(d, static_cast<hipStream_t>(stream), &kn);
    if (rc == PPN_OK && kn) g_last_conv_kernel = kn;
    return rc;
}

extern "C" const char* ppn_last_conv_kernel(void) { return g_last_conv_kernel; }

static int pack_weight_impl(int32_t dtype, const float* w, int32_t cout, int32_t cin, int32_t ksize,
                            int32_t cout_pad, int32_t k_total, int32_t k_order, int32_t k_step, void* out,
                            void* stream, int transposed) {
    if (!w || !out || cout < 1 || cin < 1 || ksize < 1 || cout_pad < cout || k_total < ksize * ksize * cin)
        return ppn::fail(PPN_E_INVALID, "ppn_pack_weight: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k_order == 2 && !transposed) {   // reference layout, f32, as is
        PPN_HIP_CHECK(hipMemcpyAsync(out, w, sizeof(float) * (size_t)cout * cin * ksize * ksize,
                                     hipMemcpyDeviceToDevice, st));
        return PPN_OK;
    }
    if (k_order == 2) {                  // reference layout f32 of the transposed filter: [cout][cin*k*k] rows
        if (cout_pad != cout || k_total != cin * ksize * ksize)
            return ppn::fail(PPN_E_INVALID, "ppn_pack_weight: k_order 2 is unpadded");
        const size_t n = (size_t)cout * k_total;
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3((int)((n + 255) / 256)), dim3(256), 0, st, w,
                           static_cast<float*>(out), cout, cin, ksize, cout_pad, k_total, 2, 1, 1);
        PPN_LAUNCH_CHECK();
        return PPN_OK;
    }
    if (k_order != 0 && (k_order != 1 || k_step < 1 || cin % k_step != 0))
        return ppn::fail(PPN_E_INVALID, "ppn_pack_weight: k_order %d needs cin %% k_step == 0", k_order);
    const size_t n = (size_t)cout_pad * k_total;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == PPN_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, st, w, static_cast<float*>(out), cout,
                           cin, ksize, cout_pad, k_total, k_order, k_step, transposed);
    else if (dtype == PPN_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, w, static_cast<__bf16*>(out),
                           cout, cin, ksize, cout_pad, k_total, k_order, k_step, transposed);
    else if (dtype == PPN_F16)
        hipLaunchKernelGGL(pack_weight_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, w, static_cast<_Float16*>(out),
                           cout, cin, ksize, cout_pad, k_total, k_order, k_step, transposed);
    else
        return ppn::fail(PPN_E_INVALID, "bad dtype %d", dtype);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

extern "C" int ppn_pack_weight(int32_t dtype, const float* w, int32_t cout, int32_t cin, int32_t ksize,
                               int32_t cout_pad, int32_t k_total, int32_t k_order, int32_t k_step, void* out,
                               void* stream) {
    return pack_weight_impl(dtype, w, cout, cin, ksize, cout_pad, k_total, k_order, k_step, out, stream, 0);
}

extern "C" int ppn_pack_weight_dgrad(int32_t dtype, const float* w, int32_t cout, int32_t cin, int32_t ksize,
                                     int32_t cout_pad, int32_t k_total, int32_t k_order, int32_t k_step, void* out,
                                     void* stream) {
    return pack_weight_impl(dtype, w, cout, cin, ksize, cout_pad, k_total, k_order, k_step, out, stream, 1);
}

extern "C" int ppn_pack_table_build(const ppn_pack_item* items, int32_t n, void* table, int32_t* total_blocks) {
    if (!items || !table || !total_blocks || n < 1) return ppn::fail(PPN_E_INVALID, "ppn_pack_table_build: bad arguments");
    PackItemDev* t = static_cast<PackItemDev*>(table);
    long long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ppn_pack_item& s = items[i];
        if (!s.w || !s.out || s.cout < 1 || s.cin < 1 || s.ksize < 1 || s.cout_pad < s.cout ||
            s.k_total < s.ksize * s.ksize * s.cin)
            return ppn::fail(PPN_E_INVALID, "ppn_pack_table_build: entry %d: bad arguments", i);
        if (s.k_order == 2) {
            if (s.cout_pad != s.cout || s.k_total != s.cin * s.ksize * s.ksize)
                return ppn::fail(PPN_E_INVALID, "ppn_pack_table_build: entry %d: k_order 2 is unpadded", i);
        } else if (s.k_order != 0 && (s.k_order != 1 || s.k_step < 1 || s.cin % s.k_step != 0)) {
            return ppn::fail(PPN_E_INVALID, "ppn_pack_table_build: entry %d: k_order %d needs cin %% k_step == 0", i, s.k_order);
        }
        if (s.dtype != PPN_F32 && s.dtype != PPN_BF16 && s.dtype != PPN_F16)
            return ppn::fail(PPN_E_INVALID, "ppn_pack_table_build: entry %d: bad dtype %d", i, s.dtype);
        PackItemDev d;
        d.w = s.w; d.out = s.out;
        d.n = (unsigned long long)s.cout_pad * (unsigned long long)s.k_total;
        d.cout = s.cout; d.cin = s.cin; d.ntaps = s.ksize * s.ksize; d.ktot = s.k_total; d.korder = s.k_order;
        d.kstep = s.k_step > 0 ? s.k_step : 1; d.transposed = s.transposed ? 1 : 0;
        d.otype = (s.k_order == 2 || s.dtype == PPN_F32) ? 0 : (s.dtype == PPN_BF16 ? 1 : 2);   // k_order 2 stays f32
        d.first_block = (int)blocks;
        static const bool tiles_on = !(getenv("PPN_PACK_TILED") && atoi(getenv("PPN_PACK_TILED")) == 0);   // A/B switch
        d.tiled = (tiles_on && d.korder == 1 && d.kstep == 64 && d.cin % 64 == 0 && d.ktot == d.ntaps * d.cin &&
                   d.ntaps <= kTileMaxTaps && d.otype != 0 && s.cout_pad % kTileRows == 0) ? 1 : 0;
        blocks += d.tiled ? (long long)(s.cout_pad / kTileRows) * (d.cin / 64)
                          : (long long)((d.n + kPackChunk - 1) / kPackChunk);
        if (blocks > 0x7fffffffLL) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_pack_table_build: too many elements");
        t[i] = d;
    }
    *total_blocks = (int32_t)blocks;
    return PPN_OK;
}

extern "C" int ppn_pack_table_run(const void* table_dev, int32_t n, int32_t total_blocks, void* stream) {
    if (!table_dev || n < 1 || total_blocks < 1) return ppn::fail(PPN_E_INVALID, "ppn_pack_table_run: bad arguments");
    hipLaunchKernelGGL(pack_table_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PackItemDev*>(table_dev), n);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
