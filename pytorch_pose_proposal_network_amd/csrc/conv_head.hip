// The limb part of the PPN head conv (model.py:133-136: conv3 1x1 512 -> 6K + E*sH*sW, bias, sigmoid) fused with the
// decode's limb arg-max (datatest.py:113 np.argmax over the sH*sW window of every (edge, cell)) -- the fast form of the
// fused-decode path (ppn_conv_desc.limb_edge_pad).  The 17.5 MB/image head tensor is never written.
//
// GEMM view: D[channel][pixel] = sum_k W[channel][k] * X[pixel][k], K = Cin (1x1 conv), M = B*H*W pixels.
// One workgroup tile = 128 pixels x ONE EDGE's whole window (441 channels padded to 448 rows), so the window's arg-max
// is reduced on the accumulators and the key of every (image, edge, cell) is STORED: no atomics, no pre-zeroed key
// buffer, no staging of the tile through LDS (the chunked epilogue of conv_big.hip spends more cycles on that than
// on the K loop: 15.2 k vs 13.5 k).
//
//   workgroup  512 threads = 8 waves as 4 (channels) x 2 (pixels); wave tile 112 channels x 64 pixels = 7 x 4 MFMA
//              16x16 tiles (112 accumulator registers); two waves per SIMD
//   LDS        2 stages x (128 + 448) rows x 128 B = 144 KB + 6 KB of arg-max partials -> one workgroup per CU
//   loads      buffer_load ... lds (LDS-DMA), source-side XOR swizzle, rows past M as out-of-range offsets (zero fill)
//   grid       PERSISTENT: one workgroup per CU walks tiles it*G + xcd_slot; K is only 512 (8 steps of 64), so a
//              tile's fixed costs matter: the two first stages of the NEXT tile are requested before the epilogue of
//              this one (the stages are free then, the epilogue only uses the partials' 6 KB), i.e. the cold fill of
//              a tile hides behind the previous tile's arg-max instead of heading the tile.
// Tiles that run at the same time on one XCD are consecutive (same pixel tile, neighbouring edges): they share the
// activation rows in that XCD's L2.
#include <cstdio>
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

using namespace ppnconv;

constexpr unsigned kOOB = 0x80000000u;   // byte offset beyond any tensor this kernel accepts (< 2 GiB)

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff,
                                              unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)lds_wave_base, 16, voff,
                                             soff, 0, 0);
}

struct HeadArgs {
    const char* src;                 // NHWC [M][Cin] (1x1 conv: input pixel == output pixel)
    const char* wgt;                 // packed [E * 448][Ktot], edge e's window in rows e*448 .. e*448 + window - 1
    const float* shift;              // [E * 448] conv bias in the same padded layout, or NULL
    unsigned long long* keys;        // u64 [B][E][HoWo]
    int M, m_base, HoWo, Cin, Ktot, window, n_edges, n_tiles;
    int pq, pr;                      // pixel tiles per XCD slot: n_ptiles = 8 * pq + pr (slots 0 .. pr-1 take pq + 1)
    FastDiv div_howo, div_pq, div_pq1;   // by HoWo, by max(pq, 1), by pq + 1
    unsigned long long* dbg;         // -DPPN_CLOCK builds only (tools/clock_head.py): per-workgroup cycle stamps
};

constexpr int BP = kEdgeTileBP, BC = kEdgeTileBC, NW = 8, WC = 4, WP = NW / WC;
constexpr int TP = BP / WP / 16, TC = BC / WC / 16;                  // 4 x 7 MFMA tiles per wave
constexpr int STAGE = (BP + BC) * 128;
constexpr int NXI = BP / (8 * NW), NWI = BC / (8 * NW), NL = NXI + NWI;
constexpr int PARTS = 3 * WC * BP * 4 + 16;                          // arg-max partials + the ambiguity flag
constexpr int SHIFTS = 2 * 512 * 4;                                  // the bias rows of this and of the next tile's edge
constexpr int SCRATCH = PARTS + SHIFTS;
static_assert(BP % (8 * NW) == 0 && BC % (8 * NW) == 0 && TC * TP % 4 == 0, "tile shape");
static_assert(2 * STAGE + SCRATCH <= 160 * 1024, "stages + partials must fit the CU's LDS");

template <typename T>
__global__ void __launch_bounds__(64 * NW, 2) head_limb_argmax_kernel(HeadArgs a, unsigned src_bytes, unsigned wgt_bytes) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int ES = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, wgt_bytes, 0x00020000);
    // the bias rows travel by LDS-DMA too (no VGPR-destination global load anywhere in the tile loop: hipcc would wait
    // vmcnt(0) at its first use and drain the next tile's stages with it); a NULL bias reads zeros (empty record range)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.shift ? a.shift : (const float*)a.wgt), 0, a.shift ? (unsigned)(a.n_edges * BC * 4) : 0u, 0x00020000);

    // Tile order.  Workgroups with equal blockIdx % 8 share an XCD (round-robin dealing: speed only, nothing depends on
    // it) and therefore an L2.  Each such group owns a contiguous EIGHTH of the pixel tiles for ALL edges and walks it
    // edge-major: its activation rows (18 tiles x 128 KB at batch 32) stay in that L2 for the whole launch and every
    // edge's 458 KB of weights is fetched once per XCD.  (Dealing consecutive tiles -- one pixel tile, all 17 edges --
    // to an XCD per round made every XCD stream all 7.8 MB of weights every round: 635 MB of fabric reads per launch
    // by the FETCH_SIZE counter against 27 MB of operands.)
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = G >> 3;
    const int psize = a.pq + (xcd < a.pr ? 1 : 0);                    // pixel tiles of this XCD slot
    const int pstart = xcd * a.pq + min(xcd, a.pr);
    const int ntl = psize * a.n_edges;                               // tiles of this XCD slot
    const FastDiv div_ps = xcd < a.pr ? a.div_pq1 : a.div_pq;
    auto tile_of = [&](int it) { return it * per + slot; };          // LOCAL tile index: edge = li / psize
    auto edge_of = [&](int li) { return fast_div(li, div_ps); };

    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2));
    const int nsteps = a.Ktot / BK;

    // ---- loader state of the tile being loaded (32-bit byte offsets) -------------------------------------------------
    unsigned xoff[NXI], woff[NWI];
    unsigned ksoff = 0;
    int ld_step = 0;
    bool live = true;
    auto setup_tile = [&](int tile) {
        const int e = edge_of(tile), pt = pstart + tile - e * psize;
        const int m0 = a.m_base + pt * BP;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int m = m0 + (j * NW + wave) * 8 + lrow;
            xoff[j] = m < a.M ? (unsigned)(m * a.Cin + chunk * EPC) * (unsigned)ES : kOOB;   // (< 2 GiB: checked by the launcher)
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            const int row = e * BC + (j * NW + wave) * 8 + lrow;
            woff[j] = (unsigned)(row * a.Ktot + chunk * EPC) * (unsigned)ES;
        }
        ksoff = 0; ld_step = 0; live = true;
    };
    auto advance = [&]() {
        ++ld_step;
        live = ld_step < nsteps;
        ksoff = live ? ksoff + BK * ES : 0u;
    };
    // one LDS-DMA instruction of the stage being loaded: g < NXI activation rows, else weight rows; past the last K step
    // every offset is out of range (zero fill, no memory traffic)
    auto issue_one = [&](auto gc, int buf) {
        constexpr int g = decltype(gc)::value;
        char* xs = smem + buf * STAGE;
        if constexpr (g < NXI)
            bufload_lds16(xrs, xs + (g * NW + wave) * 1024, live ? xoff[g] : kOOB, ksoff);
        else
            bufload_lds16(wrs, xs + BP * 128 + ((g - NXI) * NW + wave) * 1024, live ? woff[g - NXI] : kOOB, ksoff);
    };

    const int frow = lane & 15, fq = lane >> 4;
    const int fswz = (frow >> 1) & 7;
    int foff[2];
    foff[0] = frow * 128 + (((0 + fq) ^ fswz) << 4);
    foff[1] = frow * 128 + (((4 + fq) ^ fswz) << 4);
    const int x_tile_off = wp * (BP / WP) * 128;
    const int w_tile_off = BP * 128 + wc * (BC / WC) * 128;

    f32x4 acc[TC][TP];
    f32x4 wA[TC], xA[TP], wB[TC], xB[TP];
    constexpr int NRD = TC + TP;
    auto read_one = [&](auto rc, f32x4 (&wf)[TC], f32x4 (&xf)[TP], int buf, int ks) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < TP)
            xf[r] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + x_tile_off + foff[ks] + r * 16 * 128);
        else
            wf[r - TP] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + w_tile_off + foff[ks] + (r - TP) * 16 * 128);
    };
    constexpr int NG = TC * TP / 4;
    auto mma_group = [&](auto gc, const f32x4 (&wf)[TC], const f32x4 (&xf)[TP]) {
        constexpr int g = decltype(gc)::value;
        static_for<4>([&](auto tc) {
            constexpr int idx = g * 4 + decltype(tc)::value;
            mma_step(acc[idx / TP][idx % TP], wf[idx / TP], xf[idx % TP], (T*)nullptr);
        });
    };
    constexpr int RPG = (NRD + NG - 1) / NG, LPG = (NL + NG - 1) / NG;

    // arg-max partials live BEHIND the two stages: the next tile's stages are in flight during the epilogue
    float* pm = reinterpret_cast<float*>(smem + 2 * STAGE);          // [WC][BP] maxima
    int* pk = reinterpret_cast<int*>(smem + 2 * STAGE) + WC * BP;    // [WC][BP] first indices
    float* pm2 = reinterpret_cast<float*>(smem + 2 * STAGE) + 2 * WC * BP;   // [WC][BP] runner-ups
    int* flag = reinterpret_cast<int*>(smem + 2 * STAGE) + 3 * WC * BP;      // some pixel of the tile is ambiguous
    char* shifts = smem + 2 * STAGE + PARTS;                                   // [2][512] f32: edge bias rows by tile parity
    // 512 floats (the 448 of the edge + 64 of the next edge / zeros past the end) by two wave-instructions of 1 KiB
    auto issue_shift = [&](int tile_, int par) {
        if (wave < 2) {
            const int e_ = edge_of(tile_);
            bufload_lds16(srs, shifts + par * 2048 + wave * 1024, (unsigned)((e_ * BC + wave * 256 + lane * 4) * 4), 0);
        }
    };

    int tile = tile_of(0);
    if (tile >= ntl) return;                                         // (workgroup-uniform)
    setup_tile(tile);
    issue_shift(tile, 0);
    static_for<NL>([&](auto gc) { issue_one(gc, 0); });
    advance();
    static_for<NL>([&](auto gc) { issue_one(gc, 1); });
    advance();
    // LDS-DMA completes in issue order: "all but the NL youngest" covers the bias row and stage 0 of the first tile
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    __builtin_amdgcn_s_barrier();

#ifdef PPN_CLOCK
    unsigned long long ck_wait = 0, ck_k = 0, ck_issue = 0, ck_epi = 0, ck_tiles = 0;
    const unsigned long long ck_start = __builtin_amdgcn_s_memtime(), rk_start = __builtin_amdgcn_s_memrealtime();
#define HEAD_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define HEAD_T(v) do { } while (0)
#endif
    for (int it = 0;; ++it) {
        HEAD_T(t0_);
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- stage 0 of this tile is in buffer 0 (requested during the previous tile's LAST K step, waited for and
        // fenced by that step's barrier; first tile: above); stage 1 is in flight -----------------------------------------
        const int next = tile_of(it + 1);
        const bool more = next < ntl;                                 // workgroup-uniform
        const int edge = edge_of(tile), pt_e = pstart + tile - edge * psize;
        const int m0 = a.m_base + pt_e * BP;
        HEAD_T(t1_);
        static_for<NRD>([&](auto rc) { read_one(rc, wA, xA, 0, 0); });
        static_for<NRD>([&](auto rc) { read_one(rc, wB, xB, 0, 1); });
        static_for<NG>([&](auto gc) { mma_group(gc, wA, xA); });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- steady state: as conv_big.hip (set B = second half of step s-1; DMA of stage s+1 and the reads of set A
        // between groups of 4 MFMAs, order pinned) --------------------------------------------------------------------------
        for (int s = 1; s < nsteps; ++s) {
            const int buf = s & 1;
            // the loads issued during the LAST step have no stage of this tile left to fetch: they fetch stage 0 of the
            // NEXT tile into buffer 0 (nsteps is even), interleaved with this step's MFMAs like any other stage -- the
            // cold fill of a tile costs no issue slots and no latency of its own (dead, zero-filling loads after the
            // workgroup's last tile)
            if (s == nsteps - 1 && more) setup_tile(next);
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                mma_group(gc, wB, xB);
                static_for<LPG>([&](auto lc) {
                    constexpr int l = g * LPG + decltype(lc)::value;
                    if constexpr (l < NL) issue_one(std::integral_constant<int, l>{}, buf ^ 1);
                });
                static_for<RPG>([&](auto rc) {
                    constexpr int r = g * RPG + decltype(rc)::value;
                    if constexpr (r < NRD) read_one(std::integral_constant<int, r>{}, wA, xA, buf, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            advance();
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                mma_group(gc, wA, xA);
                static_for<RPG>([&](auto rc) {
                    constexpr int r = g * RPG + decltype(rc)::value;
                    if constexpr (r < NRD) read_one(std::integral_constant<int, r>{}, wB, xB, buf, 1);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        static_for<NG>([&](auto gc) { mma_group(gc, wB, xB); });
        // every wave has finished its LDS reads (they were waited for before the last barrier): both stages are free

        HEAD_T(t2_);
        // ---- stage 1 and the bias row of the NEXT tile go out now (buffer 1 was being read until the barrier above) and
        // fly under this tile's arg-max ---------------------------------------------------------------------------------------
        if (more) {
            issue_shift(next, (it + 1) & 1);
            static_for<NL>([&](auto gc) { issue_one(gc, 1); });
            advance();
        }

        // ---- epilogue: this workgroup holds ALL `window` channels of `edge` for its BP pixels ------------------------------
        // Semantics (identical to decoding the materialised head, datatest.py:113 np.argmax): key = bits of
        // max_s sigmoid(t_s) << 32 | ~(first s attaining it).  The window is decided from its LOGITS (maximum m, first
        // index, runner-up m2) when m2 < m - 2^-20 (1 + e^m) -- the preimage of one float of the sigmoid around m is
        // 2^-23 (1 + e^m) wide, times 8 for the <= 3 ulp of v_exp / v_rcp -- and then only the winner's sigmoid is
        // evaluated; otherwise (ties, saturated heads: every logit above ~16.6 gives 1.0f, the lowest index must win) the
        // sigmoid VALUES of all elements decide.  Reduction order: per lane over its TC x 4 channels of a pixel (ascending
        // channel), over the 4 lane groups that share the pixel (xor 16, 32), over the WC waves through LDS.
        HEAD_T(t3_);
        // Lane constants of the epilogue are re-derived per tile from laundered copies: left to itself hipcc hoists every
        // address / index expression below out of the tile loop, keeps ~20 of them live across the K loop (which has
        // no register to spare), spills them, and each reload in the epilogue is a scratch load whose vmcnt(0) wait
        // drains the next tile's stage requested just above.
        int fq = lane >> 4, frow = lane & 15, tid = threadIdx.x;
        asm volatile("" : "+v"(fq), "+v"(frow), "+v"(tid));
        // (plain scalars, no struct: hipcc turns  takeB ? B.x : A.x  on structs into a pointer select + loads, which
        // forces the triples into scratch memory)
        auto merge = [](float& m, int& k, float& m2, float om, int ok, float om2) {
            const float lo = fminf(m, om);
            const bool takeB = om > m || (om == m && ok < k);
            m2 = fmaxf(fmaxf(m2, om2), lo);
            m = takeB ? om : m;
            k = takeB ? ok : k;
        };
        constexpr float NEG = -3.0e38f;
        const int window = a.window;
        if (tid == 0) *flag = 0;
        {
            // channel-tile outer, pixel-tile inner: per lane TP running (m, k, m2) triples; 5 VALU per value:
            // add, med3 (the runner-up of (m, m2, t) when m >= m2), compare, max, select
            float qm[TP], qm2[TP];
            int qk[TP];
            f32x4 sh[TC];
#pragma unroll
            for (int i = 0; i < TC; ++i)
                sh[i] = *reinterpret_cast<const f32x4*>(shifts + (it & 1) * 2048 + (wc * (BC / WC) + i * 16 + 4 * fq) * 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) { qm[j] = NEG; qk[j] = 0; qm2[j] = NEG; }
#pragma unroll
            for (int i = 0; i < TC; ++i) {
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = wc * (BC / WC) + i * 16 + 4 * fq + r;
                        float t = acc[i][j][r] + sh[i][r];
                        if (i == TC - 1) t = s < window ? t : NEG;    // pad rows (only in the last channel tile) do not compete
                        qm2[j] = __builtin_amdgcn_fmed3f(qm[j], qm2[j], t);
                        const bool gt = t > qm[j];                    // strict: first maximum in ascending s
                        qm[j] = fmaxf(qm[j], t);
                        qk[j] = gt ? s : qk[j];
                    }
                __builtin_amdgcn_sched_barrier(0);   // one channel tile at a time: hoisting all 112 adds spills registers
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
#pragma unroll
                for (int d = 16; d <= 32; d <<= 1) {
                    const float om = __shfl_xor(qm[j], d), om2 = __shfl_xor(qm2[j], d);
                    const int ok = __shfl_xor(qk[j], d);
                    merge(qm[j], qk[j], qm2[j], om, ok, om2);
                }
                if (fq == 0) {
                    const int px = wp * (BP / WP) + j * 16 + frow;
                    pm[wc * BP + px] = qm[j]; pk[wc * BP + px] = qk[j]; pm2[wc * BP + px] = qm2[j];
                }
            }
        }
        lds_barrier();
        float fm = NEG, fm2 = NEG;
        int fk = 0;
        bool amb = false;
        if (tid < BP) {
            fm = pm[tid]; fk = pk[tid]; fm2 = pm2[tid];
#pragma unroll
            for (int w = 1; w < WC; ++w) merge(fm, fk, fm2, pm[w * BP + tid], pk[w * BP + tid], pm2[w * BP + tid]);
            amb = !(fm2 < fm - 9.5367431640625e-7f * (1.0f + __expf(fm)));
            if (amb && m0 + tid < a.M) *flag = 1;                     // benign race: every writer stores 1
        }
        lds_barrier();
        const bool any_amb = *flag != 0;    // workgroup-uniform; pass 1's partials were all read before this barrier
        float best = sigmoid_fast(fm);
        int best_k = fk;
        if (any_amb) {
            // value pass (rare): the sigmoid values themselves decide, lowest index among equal values
            auto vmerge = [](float& v, int& k, float ov, int ok) {
                const bool take = ov > v || (ov == v && ok < k);
                v = take ? ov : v; k = take ? ok : k;
            };
            float v[TP]; int k[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) { v[j] = -1.f; k[j] = 0; }
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shifts + (it & 1) * 2048 + (wc * (BC / WC) + i * 16 + 4 * fq) * 4);
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = wc * (BC / WC) + i * 16 + 4 * fq + r;
                        const float u = (i < TC - 1 || s < window) ? sigmoid_fast(acc[i][j][r] + sh[r]) : -1.f;
                        const bool gt = u > v[j];
                        v[j] = gt ? u : v[j]; k[j] = gt ? s : k[j];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
#pragma unroll
                for (int d = 16; d <= 32; d <<= 1) vmerge(v[j], k[j], __shfl_xor(v[j], d), __shfl_xor(k[j], d));
                if (fq == 0) {
                    const int px = wp * (BP / WP) + j * 16 + frow;
                    pm[wc * BP + px] = v[j]; pk[wc * BP + px] = k[j];
                }
            }
            lds_barrier();
            if (tid < BP) {
                float v = pm[tid]; int k = pk[tid];
#pragma unroll
                for (int w = 1; w < WC; ++w) vmerge(v, k, pm[w * BP + tid], pk[w * BP + tid]);
                best = amb ? v : best;
                best_k = amb ? k : best_k;
            }
        }
        if (tid < BP) {
            const int m = m0 + tid;
            if (m < a.M) {
                const int nb = fast_div(m, a.div_howo), np = m - nb * a.HoWo;
                const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) |
                                               (unsigned)(0xFFFFFFFFu - (unsigned)best_k);
                a.keys[((size_t)nb * a.n_edges + edge) * a.HoWo + np] = key;
            }
        }
#ifdef PPN_CLOCK
        {
            const unsigned long long t4_ = __builtin_amdgcn_s_memtime();
            ck_wait += t1_ - t0_; ck_k += t2_ - t1_; ck_issue += t3_ - t2_; ck_epi += t4_ - t3_; ++ck_tiles;
        }
#endif
        if (!more) break;
        tile = next;
        // the loader state of the new tile is RECOMPUTED here rather than kept live through the epilogue (9 registers
        // the epilogue would otherwise spill): two stages are already requested.  No barrier: the next tile's K loop
        // holds >= 2 of them between this epilogue's LDS reads and the next epilogue's writes of the partials.
        setup_tile(tile);
        advance();
        advance();
    }
#ifdef PPN_CLOCK
    if (lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
        d[0] = ck_wait; d[1] = ck_k; d[2] = ck_issue; d[3] = ck_epi; d[4] = ck_tiles;
        d[5] = __builtin_amdgcn_s_memtime() - ck_start; d[6] = __builtin_amdgcn_s_memrealtime() - rk_start;
    }
#endif
}

template <typename T>
int launch_T(const HeadArgs& a, int batch, hipStream_t st, const char** kname) {
    constexpr size_t lds = 2 * (size_t)STAGE + SCRATCH;
    static char name[80];
    if (!name[0]) snprintf(name, sizeof(name), "head_limb_argmax_kernel<%s>", elem_name<T>());
    if (kname) *kname = name;
    auto k = head_limb_argmax_kernel<T>;
    {
        static int max_lds_set = 0;
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        PPN_HIP_CHECK(hipGetDevice(&dev));
        PPN_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (n_cu < 8) n_cu = 8;
    }
    // one persistent workgroup per CU (144 KB of LDS each), a multiple of 8 so that the XCD slots of a round are equal
    int grid = (n_cu / 8) * 8;
    const int need = 8 * (a.pq + (a.pr ? 1 : 0)) * a.n_edges;       // 8 x the tiles of the largest XCD slot
    if (grid > need) grid = need;
    const size_t src_bytes = (size_t)a.M * a.Cin * sizeof(T);
    const size_t wgt_bytes = (size_t)a.n_edges * BC * a.Ktot * sizeof(T);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), lds, st, a, (unsigned)src_bytes, (unsigned)wgt_bytes);
    PPN_LAUNCH_CHECK();
    (void)batch;
    return PPN_OK;
}

}  // namespace

namespace ppn {

// called by conv_launch (conv.hip) for descriptors with limb_edge_pad != 0, after validation
int head_limb_launch(const ppn_conv_desc* d, long long m_lo, long long m_hi, hipStream_t st, const char** kname) {
    if (d->ksize != 1 || d->stride != 1 || d->pad != 0 || d->scale1)
        return ppn::fail(PPN_E_UNSUPPORTED, "limb_edge_pad: the head conv is 1x1, stride 1, no padding, bias only (scale1 NULL)");
    const int bk = d->dtype == PPN_F32 ? 32 : 64;
    if (d->k_total % (2 * bk) != 0)
        return ppn::fail(PPN_E_UNSUPPORTED, "limb_edge_pad: k_total must be a multiple of %d (an even number of K steps)", 2 * bk);
    const size_t es = d->dtype == PPN_F32 ? 4 : 2;
    const long long m_all = (long long)d->batch * d->out_h * d->out_w;
    if ((size_t)m_all * d->cin * es >= 0x7fffff00ull || (size_t)d->cout_pad * d->k_total * es >= 0x7fffff00ull)
        return ppn::fail(PPN_E_UNSUPPORTED, "tensor too large for the buffer-addressed head kernel");
    HeadArgs a;
    a.src = static_cast<const char*>(d->src);
    a.wgt = static_cast<const char*>(d->weight);
    a.shift = d->shift1;
    a.keys = reinterpret_cast<unsigned long long*>(d->argmax_keys);
    a.M = (int)m_hi; a.m_base = (int)m_lo; a.HoWo = d->out_h * d->out_w; a.Cin = d->cin; a.Ktot = d->k_total;
    a.window = d->limb_window;
    a.n_edges = d->cout / d->limb_window;
    const int n_ptiles = (int)((m_hi - m_lo + BP - 1) / BP);
    a.n_tiles = n_ptiles * a.n_edges;
    a.pq = n_ptiles / 8; a.pr = n_ptiles % 8;
    a.div_howo = make_fastdiv((unsigned)a.HoWo);
    a.div_pq = make_fastdiv((unsigned)(a.pq > 0 ? a.pq : 1));
    a.div_pq1 = make_fastdiv((unsigned)(a.pq + 1));
#ifdef PPN_CLOCK
    a.dbg = (unsigned long long*)d->shift2;      // diagnostic channel of the stamped build
#else
    a.dbg = nullptr;
#endif
    // M is the END of the range in the whole tensor; the source descriptor must cover every row below it
    if (d->dtype == PPN_F32) return launch_T<float>(a, d->batch, st, kname);
    if (d->dtype == PPN_F16) return launch_T<_Float16>(a, d->batch, st, kname);
    return launch_T<__bf16>(a, d->batch, st, kname);
}

}  // namespace ppn
