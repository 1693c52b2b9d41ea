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
//   workgroup tile (cout x cin, one tap): 256 x 256 with 8 waves (4 x 2, each 64 x 128) for >= 256-wide bf16 layers,
//              128 x 128 with 4 waves (2 x 2, each 64 x 64) otherwise (and every f32 layer: 256 x 256, 8 waves, old loop)
//   bf16 256 x 256: ping-pong loop -- the two wave groups (0-3 / 4-7, one wave of each per SIMD) run half a step out of
//              phase, one reading fragments while the other issues MFMAs -- over a four-stage ring of 32-pixel steps
//              (4 x 32 KB); bf16 128 x 128: 64-pixel steps, two stages, DMA issued between the MFMAs of the first substep;
//              f32: 32-pixel steps, two stages
//   LDS image: pixel rows of 256 or 512 bytes, 16-byte chunk c of row r at slot c ^ (((r&3)<<2) | ((r>>2)&3)) inside
//              its 256-byte window, applied on the SOURCE side of the DMA (the DMA destination is lane-linear);
//              conflict-free for the transposed reads (SQ_LDS_BANK_CONFLICT = 0)
//   bf16 reads: inline asm (the compiler would drain every pending LDS-DMA in front of a `ds_read_tr` builtin);
//   bf16 DMA offsets: dy rows linear, x rows from a per-workgroup table in LDS (built once from the two divisions and
//              four bounds tests a row needs) -- see the comments at the loops
//   work items: (pixel split, tile) pairs, the k-th contiguous eighth on XCD k (L2 locality)
//   split over pixels: every split writes its own f32 partial, a second launch folds them in a fixed
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
constexpr int kWgTableEntries = 8192;   // ping-pong kernel: x row offsets of (steps per split + 3) * 32 pixels, 32 KB of LDS
constexpr int kWgTableSmall = 4096;     // 4-wave bf16 kernel: (steps per split + 1) * 64 pixels, 16 KB (two workgroups per CU)

struct WgArgs {
    const char* x;
    const char* dy;
    float* partial;
    int B, H, W, Cin, Ho, Wo, Cout, ks, stride, dil, pad;
    int P, HoWo;
    float inv_wo, inv_howo;
    int n_mt, n_nt, ntaps;
    int n_tiles, n_items, per_xcd;   // tiles per split, tiles x splits, ceil(items / 8)
    int steps_per_split, total_steps;
    unsigned x_bytes, dy_bytes;
    unsigned long long* dbg;         // -DPPN_CLOCK builds only (tools/clock_wgrad.py): the tail of the workspace
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Transposing LDS read as inline assembly, on purpose: the compiler orders every LDS access it knows about behind ALL
// outstanding LDS-DMA (s_waitcnt vmcnt(0) in front of the first ds_read after a buffer_load ... lds -- it cannot tell
// the two stages of the ring apart), which serialises the next stage's fetch with this stage's MFMAs.  Reads it does not
// see leave the ordering to the kernel: vmcnt(0) + barrier at the top of a step, lgkmcnt(0) before the MFMAs.
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

__device__ __forceinline__ void bufload_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)lds_wave_base, 16, voff,
                                             0, 0, 0);
}

// q / d for 0 <= q < 2^24 with a float reciprocal and a one-step fix-up
__device__ __forceinline__ int fdiv(int q, int d, float inv, int* rem) {
    const int t = (int)((float)q * inv);
    const int r = q - t * d;
    const int adj = r < 0 ? -1 : (r >= d ? 1 : 0);        // selects, no branches: this runs between MFMAs
    *rem = r - adj * d;
    return t + adj;
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
    // PP: the 8-wave bf16 kernel runs its two wave groups (waves 0-3 / 4-7, one of each per SIMD) half a step out of
    // phase -- one group reads fragments while the other issues MFMAs -- over a four-stage ring of 32-pixel steps.
    constexpr bool PP = BF && NWAVE == 8;
    constexpr int BKP = BF ? (PP ? 32 : 64) : 32;                     // pixels per depth step
    constexpr int CPRA = BM / EPC, CPRB = BN / EPC;                   // 16-byte chunks per pixel row
    constexpr int ROWA = BM * ES, ROWB = BN * ES;
    constexpr int TILEA = BKP * ROWA, TILEB = BKP * ROWB;
    constexpr int NPA = TILEA / (NTHR * 16), NPB = TILEB / (NTHR * 16);   // DMA instructions per operand and step
    static_assert(TILEA % (NTHR * 16) == 0 && TILEB % (NTHR * 16) == 0, "tile must split into whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // Work item = (pixel split, tile), split-major; XCD k (workgroups k, k+8, ... in dispatch order) takes the k-th
    // CONTIGUOUS eighth of the items, so the ~32 workgroups an XCD runs at a time stream through the same pixel rows
    // and its L2 serves all but the first reader (with tiles spread round-robin every XCD read every pixel of x and
    // dy: 64 % of the L2 requests missed, 1.4 GB of HBM/MALL traffic per launch -- tools/pmc_wgrad.sh).
    const int item = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= a.per_xcd || item >= a.n_items) return;
    const int split = item / a.n_tiles;
    // tile -> (tap, cin tile, cout tile); taps fastest so that concurrent workgroups share the same pixel rows in L2
    int tix = item - split * a.n_tiles;
    const int tap = tix % a.ntaps; tix /= a.ntaps;
    const int nt_ = tix % a.n_nt;
    const int mt_ = tix / a.n_nt;
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int tdy = tap / a.ks, tdx = tap % a.ks;
    const int oy_off = tdy * a.dil - a.pad, ox_off = tdx * a.dil - a.pad;

    const int step0 = split * a.steps_per_split;
    int nsteps = a.total_steps - step0;
    nsteps = nsteps < a.steps_per_split ? nsteps : a.steps_per_split;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);

    // per-lane DMA geometry: piece i covers the lane-linear LDS range [(i*NWAVE + wave)*1024, +1024).  Consecutive
    // pieces are RSTEP rows apart (a multiple of 16, so the swizzle key and with it the lane's channel chunk are
    // the same for every piece): one row and one channel offset per operand instead of one per piece.
    constexpr int RSTEPA = NTHR / CPRA, RSTEPB = NTHR / CPRB;
    static_assert(NTHR % CPRA == 0 && NTHR % CPRB == 0 && (!BF || (RSTEPA % 16 == 0 && RSTEPB % 16 == 0)),
                  "pieces must keep the swizzle key");
    int arow0, brow0;
    unsigned a_choff, b_choff;                        // channel byte offset of the lane's chunk, or kOOB when past C
    {
        const int row = tid / CPRA, slot = tid % CPRA;
        const int ch = BF ? swz(row, slot) : slot;     // the XOR is an involution: slot -> logical chunk
        arow0 = row;
        a_choff = (m0 + ch * EPC) < a.Cout ? (unsigned)((m0 + ch * EPC) * ES) : kOOB;
    }
    {
        const int row = tid / CPRB, slot = tid % CPRB;
        const int ch = BF ? swz(row, slot) : slot;
        brow0 = row;
        b_choff = (n0 + ch * EPC) < a.Cin ? (unsigned)((n0 + ch * EPC) * ES) : kOOB;
    }

    // one DMA instruction (1 KB per wave) of the stage: pieces 0..NPA-1 are dy rows, NPA.. are x rows
    auto piece_voff = [&](int step, int piece, bool live) -> unsigned {
        const int pbase = (step0 + step) * BKP;
        if (piece < NPA) {
            const int p = pbase + arow0 + piece * RSTEPA;
            const bool ok = live & (p < a.P) & (a_choff != kOOB);
            return ok ? (unsigned)p * (unsigned)(a.Cout * ES) + a_choff : kOOB;
        }
        const int p = pbase + brow0 + (piece - NPA) * RSTEPB;
        int rem, rx;
        const int b = fdiv(p, a.HoWo, a.inv_howo, &rem);
        const int oy = fdiv(rem, a.Wo, a.inv_wo, &rx);
        const int iy = oy * a.stride + oy_off, ix = rx * a.stride + ox_off;
        const bool ok = live & (p < a.P) & (b_choff != kOOB) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        return ok ? (unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)(a.Cin * ES) + b_choff : kOOB;
    };
    auto piece_dma = [&](int stage, int piece, unsigned voff) {
        char* sa = smem + stage * (TILEA + TILEB);
        if (piece < NPA) bufload_lds16(yrs, sa + (piece * NWAVE + wave) * 1024, voff);
        else bufload_lds16(xrs, sa + TILEA + ((piece - NPA) * NWAVE + wave) * 1024, voff);
    };
    auto issue_piece = [&](int step, int stage, int piece, bool live) { piece_dma(stage, piece, piece_voff(step, piece, live)); };
    auto issue = [&](int step, int stage) {
#pragma unroll
        for (int i = 0; i < NPA + NPB; ++i) issue_piece(step, stage, i, true);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addresses (relative to the operand tile of a stage)
    const int g = lane >> 4, li = lane & 15;
    // bf16: MFMA tile t, half h (rows 8g + 4h + q of every 32-pixel substep) reads chunk (w*T + t)*2 + (p>>1) of its
    // row.  Inside the swizzle, t only sets bits 1..3 of the chunk slot and h only flips bit 0 of the key, so every
    // address is the (t = 0, h = 0) one XOR a compile-time constant, plus 4 rows for h = 1: two base registers.
    static_assert(!BF || (TM == 4 && (TN == 4 || TN == 8)), "the fragment wait lists its operands");
    static_assert(TM <= 8 && TN <= 8 && (TM & (TM - 1)) == 0 && (TN & (TN - 1)) == 0, "tile index must stay inside the key");
    unsigned ra0 = 0, rb0 = 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(char __attribute__((address_space(3)))*)smem;
    unsigned fa[TM], fb[TN];                          // f32 : [tile]
    if (BF) {
        const int q = li >> 2, p = li & 3;
        const int row = 8 * g + q;                                        // + 4*h + 32*substep
        ra0 = row * ROWA + swz(row, (wm * TM) * 2 + (p >> 1)) * 16 + 8 * (p & 1);
        rb0 = row * ROWB + swz(row, (wn * TN) * 2 + (p >> 1)) * 16 + 8 * (p & 1);
    } else {
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = g * ROWA + ((wm * TM + t) * 16 + li) * 4;   // + 4 rows per substep
#pragma unroll
        for (int t = 0; t < TN; ++t) fb[t] = g * ROWB + ((wn * TN + t) * 16 + li) * 4;
    }

#ifdef PPN_CLOCK
    unsigned long long ck_dma = 0, ck_bar = 0, ck_issue = 0, ck_mma = 0;
    const unsigned long long ck_start = __builtin_amdgcn_s_memtime(), rk_start = __builtin_amdgcn_s_memrealtime();
#define WG_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define WG_T(v) do { } while (0)
#endif
    // bf16: tbl[e] = byte offset of the x row that pixel step0*BKP + e reads through this tile's tap, or out of range
    // (entries past the split's last step are out of range: the ring's run-ahead requests are dead there)
    auto build_table = [&](unsigned* tbl, int n_tbl) {
        for (int e = tid; e < n_tbl; e += NTHR) {
            const int pp = step0 * BKP + e;
            int rem, rx;
            const int b = fdiv(pp, a.HoWo, a.inv_howo, &rem);
            const int oy = fdiv(rem, a.Wo, a.inv_wo, &rx);
            const int iy = oy * a.stride + oy_off, ix = rx * a.stride + ox_off;
            const bool ok = (e < nsteps * BKP) & (pp < a.P) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            tbl[e] = ok ? (unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)(a.Cin * ES) : kOOB;
        }
    };
    if constexpr (PP) {
        // ---- ping-pong loop -----------------------------------------------------------------------------------------
        // Every wave runs  R(s): 24 transposed reads of stage s  |barrier|  M(s): 32 MFMAs  |barrier|  and group 1 is
        // one barrier behind group 0, so in every interval between two barriers one wave of each SIMD reads while the
        // other one multiplies: the reads (48 KB per interval) hide behind the matrix pipe instead of in front of it
        // (with all eight waves in lock step a step cost 3 900 cycles against 2 048 of MFMA issue).
        // Ring of four 32 KB stages; stage s+3 is requested during step s (its buffer held stage s-1, whose last
        // readers passed the barrier before this wave's R(s)), RP pieces in R and the rest between the MFMAs.  A
        // stage is consumed three steps after its request: before the barrier that ends R(s) / M(s) a wave lets at most
        // PPS + RP / 2*PPS DMA instructions stay in flight (stage s+2 and what it has issued of s+3), i.e. its share
        // of stage s+1 has landed when the other group starts reading it.
        //
        // A wave issues one instruction every ~4-5 cycles, so a 500-cycle phase holds ~100 of them and the reads take
        // 50: the DMA offsets must cost a handful.  dy rows are linear (one add per piece and step).  The x row of a
        // pixel needs two divisions and four bounds tests (~45 VALU instructions per piece when recomputed, ~40 SALU
        // when carried as scalar state: either made the read phase 1 000-1 200 cycles, tools/clock_wgrad.py), so the
        // workgroup tabulates them once: tbl[e] = byte offset of the x row that pixel step0*32 + e reads through this
        // tile's tap, or out of range; per step a lane fetches its two entries with ds_read_b32.
#ifndef PPN_WG_RP
#define PPN_WG_RP 2
#endif
        constexpr int PPS = NPA + NPB, STAGE = TILEA + TILEB, RP = PPN_WG_RP;
        static_assert(PPS == 4 && NPA == 2 && TM == 4 && TN == 8 && RP <= NPA, "ping-pong schedule is written for the 256 x 256 tile");
        static_assert(RSTEPA == 16 && RSTEPB == 16, "piece rows");
        const int grp = wave >> 2;
        auto phase_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        unsigned* tbl = reinterpret_cast<unsigned*>(smem + 4 * STAGE);
        build_table(tbl, (nsteps + 3) * BKP);                           // geometry(): <= kWgTableEntries
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int i = 0; i < PPS; ++i) issue_piece(st, st, i, st < nsteps);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPS) : "memory");
        phase_barrier();
        if (grp) phase_barrier();
        // per-lane DMA state of stage 3: dy offsets (running), table address (running), channel offset / range bit of x
        const unsigned a_oob = a_choff == kOOB ? kOOB : 0u, b_oob = b_choff == kOOB ? kOOB : 0u;
        const unsigned b_ch = b_choff == kOOB ? 0u : b_choff;
        const unsigned a_inc = (unsigned)BKP * (unsigned)(a.Cout * ES);
        unsigned va0 = ((unsigned)((step0 + 3) * BKP + arow0) * (unsigned)(a.Cout * ES) + (a_choff == kOOB ? 0u : a_choff)) | a_oob;
        unsigned va1 = va0 + (unsigned)RSTEPA * (unsigned)(a.Cout * ES);          // rows past P: beyond num_records, zeros
        unsigned tba = lds0 + 4 * STAGE + (3 * BKP + brow0) * 4;
        for (int s = 0; s < nsteps; ++s) {
            WG_T(t0_);
            const unsigned la = lds0 + (s & 3) * STAGE + ra0, lb = lds0 + (s & 3) * STAGE + TILEA + rb0;
            bf16x8 af[TM], bfr[TN];
            static_for<TN>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const s16x4 b0 = lds_read_tr16<0>(lb ^ (t << 5));
                const s16x4 b1 = lds_read_tr16<4 * ROWB>(lb ^ (t << 5 | 16));
                bfr[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            });
            static_for<TM>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const s16x4 a0 = lds_read_tr16<0>(la ^ (t << 5));
                const s16x4 a1 = lds_read_tr16<4 * ROWA>(la ^ (t << 5 | 16));
                af[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            });
            unsigned t0, t1;                                             // x row offsets of the lane's two pixels, stage s+3
            asm volatile("ds_read_b32 %0, %1" : "=v"(t0) : "v"(tba));
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t1) : "v"(tba), "n"(RSTEPB * 4));
            if constexpr (RP > 0) piece_dma((s + 3) & 3, 0, va0);
            if constexpr (RP > 1) piece_dma((s + 3) & 3, 1, va1);
            // every fragment is an operand of the wait, so no MFMA can be scheduled in front of it
            asm volatile("s_waitcnt vmcnt(%14) lgkmcnt(0)"
                         : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]),
                           "+v"(bfr[3]), "+v"(bfr[4]), "+v"(bfr[5]), "+v"(bfr[6]), "+v"(bfr[7]), "+v"(t0), "+v"(t1)
                         : "n"(PPS + RP)
                         : "memory");
            WG_T(t1_);
            phase_barrier();
            WG_T(t2_);
            const unsigned voff[PPS] = {va0, va1, (t0 + b_ch) | b_oob, (t1 + b_ch) | b_oob};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    constexpr int GAP = TM * TN / (PPS - RP + 1);          // MFMAs between two pieces
                    const int n = i * TN + j;
                    if (n % GAP == 0 && n > 0 && n / GAP <= PPS - RP) {
                        piece_dma((s + 3) & 3, RP + n / GAP - 1, voff[RP + n / GAP - 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            va0 += a_inc; va1 += a_inc; tba += BKP * 4;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPS) : "memory");
            WG_T(t3_);
            phase_barrier();
#ifdef PPN_CLOCK
            {
                const unsigned long long t4_ = __builtin_amdgcn_s_memtime();
                ck_dma += t1_ - t0_; ck_bar += t2_ - t1_; ck_issue += t3_ - t2_; ck_mma += t4_ - t3_;
            }
#endif
        }
        if (!grp) phase_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // 4-wave bf16 kernel: the same cheap DMA offsets (dy linear, x from the table) -- recomputed per lane they were ~230 VALU
    // instructions per wave and step, more SIMD time than the step's 32 MFMAs
    constexpr int TBL_OFS = 2 * (TILEA + TILEB);
    static_assert(!BF || (NPA == 4 && NPB == 4 && RSTEPA == 16 && RSTEPB == 16), "piece rows of the 4-wave bf16 kernel");
    unsigned va0 = 0, tba = 0, b_ch = 0, b_oob = 0;
    const unsigned a_rstep = (unsigned)RSTEPA * (unsigned)(a.Cout * ES);
    if (BF) {
        build_table(reinterpret_cast<unsigned*>(smem + TBL_OFS), (nsteps + 1) * BKP);   // geometry(): <= kWgTableSmall
        const unsigned a_oob = a_choff == kOOB ? kOOB : 0u;
        va0 = ((unsigned)((step0 + 1) * BKP + arow0) * (unsigned)(a.Cout * ES) + (a_choff == kOOB ? 0u : a_choff)) | a_oob;
        tba = lds0 + TBL_OFS + (BKP + brow0) * 4;
        b_ch = b_choff == kOOB ? 0u : b_choff;
        b_oob = b_choff == kOOB ? kOOB : 0u;
    }
    if (nsteps > 0) issue(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        WG_T(t0_);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WG_T(t1_);
        __syncthreads();
        WG_T(t2_);
        // bf16: the next stage's DMA instructions are issued BETWEEN the MFMAs of the first 32-pixel substep (one per
        // PER MFMAs); the second substep leaves the loads a substep's worth of time to land before the vmcnt(0) of
        // the next step.  (On the last step the pieces are dead: table entries out of range, dy rows of the next split
        // into the stage nobody reads.)
        const bool more = s + 1 < nsteps;
        if (!BF && more) issue(s + 1, (s + 1) & 1);
        WG_T(t3_);
        const char* sa = smem + (s & 1) * (TILEA + TILEB);
        const char* sb = sa + TILEA;
        if (BF) {
            constexpr int NPT = NPA + NPB;
            constexpr int PER = (TM * TN) / NPT > 0 ? (TM * TN) / NPT : 1;
            const unsigned la = lds0 + (s & 1) * (TILEA + TILEB) + ra0, lb = lds0 + (s & 1) * (TILEA + TILEB) + TILEA + rb0;
            static_for<BKP / 32>([&](auto subc) {
                constexpr int sub = decltype(subc)::value;
                bf16x8 af[TM], bfr[TN];
                static_for<TM>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    const s16x4 a0 = lds_read_tr16<sub * 32 * ROWA>(la ^ (t << 5));
                    const s16x4 a1 = lds_read_tr16<(sub * 32 + 4) * ROWA>(la ^ (t << 5 | 16));
                    af[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                });
                static_for<TN>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    const s16x4 b0 = lds_read_tr16<sub * 32 * ROWB>(lb ^ (t << 5));
                    const s16x4 b1 = lds_read_tr16<(sub * 32 + 4) * ROWB>(lb ^ (t << 5 | 16));
                    bfr[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                });
                unsigned tx[4] = {0, 0, 0, 0};                       // x row offsets of the lane's four pixels, stage s+1
                if constexpr (sub == 0) {
                    asm volatile("ds_read_b32 %0, %1" : "=v"(tx[0]) : "v"(tba));
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tx[1]) : "v"(tba), "n"(RSTEPB * 4));
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tx[2]) : "v"(tba), "n"(2 * RSTEPB * 4));
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(tx[3]) : "v"(tba), "n"(3 * RSTEPB * 4));
                }
                // every fragment is an operand of the wait, so no MFMA can be scheduled in front of it
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bfr[0]), "+v"(bfr[1]),
                               "+v"(bfr[2]), "+v"(bfr[3]), "+v"(tx[0]), "+v"(tx[1]), "+v"(tx[2]), "+v"(tx[3]));
                if (sub == 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = i * TN + j;
                            if (n % PER == 0 && n / PER < NPT) {
                                const int pc = n / PER;
                                piece_dma((s + 1) & 1, pc, pc < NPA ? va0 + pc * a_rstep : (tx[pc < NPA ? 0 : pc - NPA] + b_ch) | b_oob);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                        }
                    va0 += (unsigned)BKP * (unsigned)(a.Cout * ES);
                    tba += BKP * 4;
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            });
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
#ifdef PPN_CLOCK
        {
            const unsigned long long t4_ = __builtin_amdgcn_s_memtime();
            ck_dma += t1_ - t0_; ck_bar += t2_ - t1_; ck_issue += t3_ - t2_; ck_mma += t4_ - t3_;
        }
#endif
    }
    }
#ifdef PPN_CLOCK
    if (lane == 0 && a.dbg) {
        unsigned long long* d = a.dbg + ((size_t)item * NWAVE + wave) * 8;
        d[0] = ck_dma; d[1] = ck_bar; d[2] = ck_issue; d[3] = ck_mma; d[4] = (unsigned long long)nsteps;
        d[5] = __builtin_amdgcn_s_memtime() - ck_start; d[6] = __builtin_amdgcn_s_memrealtime() - rk_start;
    }
#endif

    // partial[split][tap][co][ci]
    float* out = a.partial + ((size_t)split * a.ntaps + tap) * (size_t)a.Cout * a.Cin;
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

// dw[co][ci][tap] = beta*dw + sum_s partial[s][tap][co][ci]   (fixed order; one thread per FOUR consecutive output elements
// of a tap -- 16-byte loads, 1 KB per wave and load instead of 256 B -- with the loads of four splits in flight at a time: a
// serial chain of nsplit*taps loads per thread was latency-bound)
__global__ void __launch_bounds__(256) wgrad_fold_kernel(const float* __restrict__ partial, int nsplit, int ntaps,
                                                         int Cout, int Cin, float beta, float* __restrict__ dw) {
    const long long n = (long long)Cout * Cin, n4 = n / 4;              // Cin % 4 == 0 (geometry())
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= n4 * ntaps) return;
    const int t = (int)(w / n4);
    const long long o = (w - (long long)t * n4) * 4;
    const float* p = partial + (size_t)t * n + o;
    const size_t stride = (size_t)ntaps * n;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4& v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)s * stride);
        const float4 v1 = *reinterpret_cast<const float4*>(p + (size_t)(s + 1) * stride);
        const float4 v2 = *reinterpret_cast<const float4*>(p + (size_t)(s + 2) * stride);
        const float4 v3 = *reinterpret_cast<const float4*>(p + (size_t)(s + 3) * stride);
        add(v0); add(v1); add(v2); add(v3);
    }
    for (; s < nsplit; ++s) add(*reinterpret_cast<const float4*>(p + (size_t)s * stride));
    float* d = dw + o * ntaps + t;
    const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) d[(size_t)j * ntaps] = beta != 0.f ? beta * d[(size_t)j * ntaps] + r[j] : r[j];
}

struct Geom {
    int bkp, total_steps, nsplit, steps_per_split, n_mt, n_nt, ntaps;
    int big;     // 256 x 256 tile (8 waves) instead of 128 x 128 (4 waves)
};

int geometry(const ppn_wgrad_desc* d, Geom* g) {
    if (!d) return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: NULL descriptor");
    if (d->dtype != PPN_F32 && d->dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "bad dtype %d", d->dtype);
    const int epc = d->dtype == PPN_F32 ? 4 : 8;
    // (the dedicated stem kernels bring their own shapes: layer 0 on a 4-channel input)
    if (d->batch <= 0 || d->cin <= 0 || d->cout <= 0 || ((d->cin % epc || d->cout % epc) && !ppn::stem_wgrad_supported(d)))
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
    static const char* force = getenv("PPN_WGRAD_TILE");                 // tuning knob: "128" / "256"
    g->big = force ? atoi(force) == 256 : (d->cout >= 256 && d->cin >= 256);
    g->bkp = d->dtype == PPN_F32 || g->big ? 32 : 64;                    // wgrad_kernel's BKP
    g->total_steps = (int)((P + g->bkp - 1) / g->bkp);
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
    if (d->dtype != PPN_F32) {                  // the x-row offset table in LDS bounds a split
        const int cap = g->big ? kWgTableEntries / 32 - 3 : kWgTableSmall / 64 - 1;
        if (g->steps_per_split > cap) g->steps_per_split = cap;
    }
    g->nsplit = (g->total_steps + g->steps_per_split - 1) / g->steps_per_split;
    return PPN_OK;
}

template <typename T, int WM, int WN, int TM, int TN>
int launch(const WgArgs& a, dim3 grid, hipStream_t st) {
    static int lds_set = 0;
    constexpr bool pp = sizeof(T) == 2 && WM * WN == 8;
    constexpr int bkp = sizeof(T) == 2 && !pp ? 64 : 32;
    constexpr int lds = (pp ? 4 : 2) * bkp * (WM * TM + WN * TN) * 16 * (int)sizeof(T) +
                        (pp ? kWgTableEntries * 4 : sizeof(T) == 2 ? kWgTableSmall * 4 : 0);
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
    size_t need = (size_t)g.nsplit * g.ntaps * d->cout * d->cin * sizeof(float);
#ifdef PPN_CLOCK
    need += (size_t)g.n_mt * g.n_nt * g.ntaps * g.nsplit * 8 * 64;     // cycle stamps behind the partials
#endif
    return need;
}

int ppn_conv_wgrad(const ppn_wgrad_desc* d, void* stream) {
    Geom g;
    if (int rc = geometry(d, &g)) return rc;
    if (!d->x || !d->dy || !d->dw || !d->workspace) return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: NULL pointer");
    if (reinterpret_cast<uintptr_t>(d->workspace) & 15)
        return ppn::fail(PPN_E_INVALID, "ppn_conv_wgrad: workspace must be 16-byte aligned");
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
    a.n_tiles = g.n_mt * g.n_nt * g.ntaps;
    a.n_items = a.n_tiles * g.nsplit;
    a.per_xcd = (a.n_items + 7) / 8;
    a.steps_per_split = g.steps_per_split;
    a.total_steps = g.total_steps;
    a.x_bytes = (unsigned)((size_t)d->batch * d->in_h * d->in_w * d->cin * es);
    a.dy_bytes = (unsigned)((size_t)a.P * d->cout * es);
#ifdef PPN_CLOCK
    a.dbg = d->workspace_bytes >= need + (size_t)g.n_mt * g.n_nt * g.ntaps * g.nsplit * 8 * 64
                ? (unsigned long long*)((char*)d->workspace + need) : nullptr;
#else
    a.dbg = nullptr;
#endif
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(8 * a.per_xcd);
    int rc;
    if (d->dtype == PPN_F32) rc = g.big ? launch<float, 4, 2, 4, 8>(a, grid, st) : launch<float, 2, 2, 4, 4>(a, grid, st);
    else rc = g.big ? launch<__bf16, 4, 2, 4, 8>(a, grid, st) : launch<__bf16, 2, 2, 4, 4>(a, grid, st);
    if (rc) return rc;
    const long long n = (long long)d->cout * d->cin * g.ntaps;
    wgrad_fold_kernel<<<(int)((n / 4 + 255) / 256), 256, 0, st>>>(a.partial, g.nsplit, g.ntaps, d->cout, d->cin, d->beta,
                                                                  d->dw);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // extern "C"
