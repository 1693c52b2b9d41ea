// Decode + root-box NMS + greedy limb parse on the device head tensor (gfx950, wave64).
//
// Replaces the NumPy/Python of the reference:
//   head slicing, delta = resp*conf ............ rt_test.py:106-130
//   restore_xy / restore_size / bbox ........... datatest.py:63-86
//   candidates, non_maximum_suppression ........ datatest.py:87-95, 134-160
//   limb parse over DIRECTED_GRAPHS ............ datatest.py:98-132, config.py:67-80
//
// Two kernels per batch:
//   limb_argmax_kernel  HBM-bound: streams the whole e block (E*sH*sW channels x H*W cells) of
//                       every image exactly once with 16-byte coalesced loads and keeps the
//                       FIRST maximum per (edge, cell) -- np.argmax semantics (datatest.py:113).
//   parse_kernel        one workgroup per image: ballot-compaction of root candidates, rank sort,
//                       pairwise-IoU bit matrix, a one-wave greedy resolve, and one lane per
//                       surviving root walking the skeleton tree through the arg-max map.
//
// All floating-point arithmetic follows the reference's fp32 operation order; this file is
// compiled with -ffp-contract=off so no multiply-add is fused (the IoU >= 0.3 and
// delta < 0.15 tests are knife edges).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

#ifdef PPN_STAMP
#define PPN_DT(i) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    reinterpret_cast<unsigned long long*>(out_bbox + ((size_t)blockIdx.x * c.max_humans + c.max_humans - 1) * c.K * 4)[i] = t_; } } while (0)
#else
#define PPN_DT(i) do { } while (0)
#endif

constexpr int kWave = 64;

struct DecodeParams {
    ppn_decode_cfg c;
    int C;       // channels per image of `head` = 6K + E*sH*sW (6K for the compact unary tensor of the fused path)
    int ncell;   // H*W
    int S;       // sH*sW
    const unsigned long long* keys;   // fused path: arg-max keys u64 [B][E][ncell] instead of the int arg-max map
    const int* early;                 // root NMS done inside the arg-max launch (early_root_nms) or nullptr
    // fused path, round 4: the sorted root candidates and their pairwise-suppression bit matrix, computed by the
    // root_mask_kernel launch in front of the parse (G workgroups per image) -- or nullptr
    const int* root_hdr;              // i32 [B][root_hdr_stride]: n (-1: not computed, the parse kernel does it), sorted cells
    const unsigned long long* root_mask;   // u64 [B][ncell][nwords]: row i, word w = [iou(i, j) >= thr] for j = 64 w .. < i
};

// Root candidates + NMS of an image do not depend on the limb arg-max: in the stand-alone decode the FIRST workgroup of
// every image of the arg-max launch (edge 0, cell group 0: dispatched first, so the extra work disappears in the
// launch's dynamic schedule) runs them BEFORE its arg-max share and leaves the survivors for the parse kernel:
// early[b][0] = number kept, early[b][1..] = their cells in kept order.  Any number of candidates (round 3; round 2
// bounded it to 128 and left dense heads -- ~490 candidates -- to the parse kernel, whose NMS then ran alone for ~100 us).
// Same device functions, same order of operations: results are identical either way.
//
// Round 3 also built and measured the single-launch form the review asked for -- every arg-max workgroup counts itself
// done on a per-image counter (write-through stores, vmcnt(0), barrier, one agent-scope atomic add), the LAST one
// acquires, stages the arg-max map (u16) and a hop-acceptance bitmap in LDS and walks the tree in place, no parse launch.
// Bit-exact, but not faster on the crowd heads: 116.5-117.3 us against 111.6-115.1 us for the two launches (same box).
// The walk's tail still costs ~11 us after the image's last workgroup (returning atomic, agent-scope acquire, one HBM
// round trip for the tables -- the write-through stores drop their lines from L2 --, the 17-hop LDS chain, one more
// round trip for the boxes) and the completion protocol adds ~2 us of exit latency to each of the 1 632 workgroups,
// which the parse kernel's 15 us + 1.5 us launch boundary do not exceed.  Removed again; what stayed from it is the
// unbounded early NMS (dense heads: 208 -> ~120 us) and the explicitly batched loads of the arg-max loop.
// per-image stride of the early lists: whole 128-byte lines
__host__ __device__ inline int early_stride(int ncell) { return ((1 + ncell + 31) / 32) * 32; }                 // ints

// ------------------------------------------------------------------------------------------
// Kernel 1: dense limb arg-max.  grid = (E, batch); thread t -> (column group q, row slice r).
// ------------------------------------------------------------------------------------------
template <int V>
struct VecT;
template <>
struct VecT<4> {
    using type = float4;
};
template <>
struct VecT<1> {
    using type = float;
};

template <int V>
__device__ __forceinline__ void vec_to_arr(const typename VecT<V>::type& v, float* a);
template <>
__device__ __forceinline__ void vec_to_arr<4>(const float4& v, float* a) {
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
}
template <>
__device__ __forceinline__ void vec_to_arr<1>(const float& v, float* a) {
    a[0] = v;
}

// ------------------------------------------------------------------------------------------
// Shared pieces of the per-image parse and the standalone NMS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_area(const float4& b) {       // (ymin,xmin,ymax,xmax)
    return (b.z - b.x) * (b.w - b.y);                               // datatest.py:141
}

// iou(b_i, b_j) >= thr with the reference's operation order (datatest.py:145-150).
__device__ __forceinline__ bool iou_ge(const float4& bi, float ai, const float4& bj, float aj, float thr) {
    float tl0 = fmaxf(bi.x, bj.x), tl1 = fmaxf(bi.y, bj.y);
    float br0 = fminf(bi.z, bj.z), br1 = fminf(bi.w, bj.w);
    float inter = ((br0 - tl0) * (br1 - tl1)) * ((tl0 < br0 && tl1 < br1) ? 1.0f : 0.0f);
    float iou = inter / ((ai + aj) - inter);
    return iou >= thr;
}

// Block-wide exclusive prefix of a per-thread flag (threads in row-major order). Returns the
// position of this thread among the flagged ones and writes the total to *total (LDS).
__device__ __forceinline__ int block_compact(bool flag, int* s_wave_cnt, int* total) {
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = blockDim.x / kWave;
    unsigned long long m = __ballot(flag);
    int prefix = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        int c = s_wave_cnt[w];
        if (w < wid) off += c;
        tot += c;
    }
    if (threadIdx.x == 0) *total = tot;
    __syncthreads();
    return off + prefix;
}

// Steps shared by parse and nms, operating on n boxes already in sorted (priority) order.
//   s_box/s_area [n], s_mask [n][nw] scratch, s_sel [n] out (sorted positions kept, in order), *s_nsel out.
// One 64-bit word of the pairwise-suppression matrix: row i, candidates j = 64 w .. min(64 w + 64, i) - 1.
__device__ __forceinline__ unsigned long long iou_word(int i, int w, const float4* s_box, const float* s_area, float thr) {
    const int j0 = w * 64;
    const int j1 = min(j0 + 64, i);
    unsigned long long bits = 0ull;
    if (j0 < j1) {
        const float4 bi = s_box[i];
        const float ai = s_area[i];
        // four independent IoU chains in flight: one evaluation is a ~300-cycle dependent chain
        // (LDS read, min/max, IEEE division) and only ~2 waves share a SIMD here
        int j = j0;
        for (; j + 4 <= j1; j += 4) {
            bool r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = iou_ge(bi, ai, s_box[j + u], s_area[j + u], thr);
#pragma unroll
            for (int u = 0; u < 4; ++u) bits |= r[u] ? (1ull << (j + u - j0)) : 0ull;
        }
        for (; j < j1; ++j) {
            if (iou_ge(bi, ai, s_box[j], s_area[j], thr)) bits |= (1ull << (j - j0));
        }
    }
    return bits;
}

__device__ __forceinline__ void greedy_resolve(int n, int nwords, const unsigned long long* s_mask, int* s_sel, int* s_nsel,
                                               int limit);

__device__ __forceinline__ void greedy_nms(int n, int nwords, const float4* s_box, const float* s_area,
                                           unsigned long long* s_mask, int* s_sel, int* s_nsel, float thr,
                                           int limit) {
    const int t = threadIdx.x;
    // pairwise suppression bits: row i holds, for every higher-priority j < i, [iou(i,j) >= thr]
    // Work item = one 64-bit word w of one row i.  Consecutive lanes take consecutive rows of the SAME word, so
    // every lane of a wave reads the same s_box[j] (an LDS broadcast; the transposed assignment made 9 lanes hit
    // one bank with 9 addresses) and whole waves of the empty upper triangle exit at once.
    for (int q = t; q < n * nwords; q += blockDim.x) {
        const int w = q / n, i = q - w * n;
        s_mask[(size_t)i * nwords + w] = iou_word(i, w, s_box, s_area, thr);
    }
    __syncthreads();
    greedy_resolve(n, nwords, s_mask, s_sel, s_nsel, limit);
}

// The greedy order on a finished bit matrix (rows in LDS); ends with a workgroup barrier.
__device__ __forceinline__ void greedy_resolve(int n, int nwords, const unsigned long long* s_mask, int* s_sel, int* s_nsel,
                                               int limit) {
    const int t = threadIdx.x;
    // One wave resolves the greedy order, 64 candidates (one per lane) at a time:
    //   (a) in parallel, a lane drops out if any already-kept box of an EARLIER chunk suppresses it;
    //   (b) inside the chunk the dependence is sequential, but it runs on scalar registers only: the chunk's
    //       own 64x64 bit block is walked with readlane, ~6 SALU instructions per candidate.
    // Lane w keeps the keep-word of chunk w.  Identical to the one-by-one loop of datatest.py:143-156.
    if (t < kWave) {
        unsigned long long keep_w = 0ull;                            // lane w: kept bits of chunk w
        int nsel = 0;
        const int nchunks = (n + 63) >> 6;
        for (int cidx = 0; cidx < nchunks; ++cidx) {
            const int i = (cidx << 6) + t;
            const bool valid = i < n;
            bool alive = valid;
            const unsigned long long* rowp = s_mask + (size_t)(valid ? i : 0) * nwords;
            const unsigned long long intra = valid ? rowp[cidx] : 0ull;
            for (int w = 0; w < cidx; ++w) {
                const unsigned long long kw = __shfl(keep_w, w);     // executed by all 64 lanes (no divergence)
                if (valid && (rowp[w] & kw)) alive = false;
            }
            const unsigned long long cand = __ballot(alive);
            unsigned long long keepc = 0ull;
            const unsigned lo = (unsigned)intra, hi = (unsigned)(intra >> 32);
            for (int bpos = 0; bpos < 64; ++bpos) {
                if ((cand >> bpos) & 1ull) {
                    const unsigned long long row = ((unsigned long long)__builtin_amdgcn_readlane(hi, bpos) << 32) |
                                                   (unsigned)__builtin_amdgcn_readlane(lo, bpos);
                    if ((row & keepc) == 0ull) keepc |= (1ull << bpos);
                }
            }
            if (limit > 0 && nsel + __popcll(keepc) > limit) {        // datatest.py:157-158: stop at `limit`
                int room = limit - nsel;
                unsigned long long trimmed = 0ull, rest = keepc;
                while (room-- > 0 && rest) { const unsigned long long low = rest & (0ull - rest); trimmed |= low; rest ^= low; }
                keepc = trimmed;
            }
            if ((keepc >> t) & 1ull) s_sel[nsel + __popcll(keepc & ((1ull << t) - 1ull))] = i;
            if (t == cidx) keep_w = keepc;
            nsel += __popcll(keepc);
            if (limit > 0 && nsel >= limit) break;
        }
        if (t == 0) *s_nsel = nsel;
    }
    __syncthreads();
}

// Phases 1-2 of the parse for one image: root candidates (delta[0] > thr, datatest.py:89), ranked by descending score
// (ties: ascending cell), boxes / areas / cells stored in priority order.  Whole workgroup, blockDim.x >= ncell; returns n.
__device__ __forceinline__ int sort_roots(const ppn_decode_cfg& c, const float* __restrict__ img, int ncell, float4* s_box,
                                          unsigned long long* s_key, float* s_area, int* s_cell, int* s_misc) {
    const int t = threadIdx.x, K = c.K, W = c.W, H = c.H;
    float d0 = 0.0f;
    bool is_c = false;
    if (t < ncell) {
        d0 = img[t] * img[(size_t)K * ncell + t];                     // delta of the root keypoint (rt_test.py:130)
        is_c = d0 > c.det_thr;
    }
    const int pos = block_compact(is_c, s_misc, s_misc + 32);
    const int n = s_misc[32];
    if (is_c) s_key[pos] = ((unsigned long long)(~__float_as_uint(d0)) << 32) | (unsigned)t;
    __syncthreads();
    if (is_c) {
        const unsigned long long my = s_key[pos];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (s_key[j] < my) ? 1 : 0;
        const float gridW = (float)(c.inW / W), gridH = (float)(c.inH / H);
        const float x = img[(size_t)(2 * K) * ncell + t], y = img[(size_t)(3 * K) * ncell + t];
        const float w = img[(size_t)(4 * K) * ncell + t], h = img[(size_t)(5 * K) * ncell + t];
        const float X = (float)(t % W), Y = (float)(t / W);
        const float rx = (x + X) * gridW, ry = (y + Y) * gridH;       // datatest.py:63-84, as parse_kernel::bbox_at
        const float rw = (float)c.inW * w, rh = (float)c.inH * h;
        float4 bb;
        bb.x = ry - rh / 2.0f; bb.y = rx - rw / 2.0f; bb.z = ry + rh / 2.0f; bb.w = rx + rw / 2.0f;
        s_box[rank] = bb;
        s_area[rank] = box_area(bb);
        s_cell[rank] = t;
    }
    __syncthreads();
    return n;
}

// Phases 1-4 of the parse (candidates, rank sort, greedy NMS) for one image.
// Called by a whole workgroup with blockDim.x >= ncell; LDS: early_lds_bytes(ncell).
__device__ __forceinline__ void early_root_nms(const ppn_decode_cfg& c, const float* __restrict__ img, int ncell,
                                               char* smem, int* __restrict__ out) {
    const int t = threadIdx.x;
    const int nwords = (ncell + 63) >> 6;
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* ptr = smem + off; off += (bytes + 15) & ~size_t(15); return ptr; };
    float4* s_box = reinterpret_cast<float4*>(carve(sizeof(float4) * ncell));
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(carve(8 * ncell));
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(carve(8 * (size_t)ncell * nwords));
    float* s_area = reinterpret_cast<float*>(carve(4 * ncell));
    int* s_cell = reinterpret_cast<int*>(carve(4 * ncell));
    int* s_sel = reinterpret_cast<int*>(carve(4 * ncell));
    int* s_misc = reinterpret_cast<int*>(carve(4 * 40));
    const int n = sort_roots(c, img, ncell, s_box, s_key, s_area, s_cell, s_misc);
    greedy_nms(n, nwords, s_box, s_area, s_mask, s_sel, s_misc + 33, c.nms_thr, 0);
    const int nsel = s_misc[33];
    if (t == 0) out[0] = nsel;
    if (t < nsel) out[1 + t] = s_cell[s_sel[t]];
}

// ------------------------------------------------------------------------------------------
// Fused path, dense heads (round 4): the O(n^2) pairwise-IoU bit matrix of an image's root candidates spread over G
// workgroups.  One workgroup per image took ~60 of the parse kernel's 109 us for it on the benchmark's heads (~490 of 576
// cells are root candidates there: 120 k IoU evaluations, each an IEEE division); here every workgroup of an image
// repeats the cheap phases (candidates, rank sort: identical results) and computes every G-th block of (row, word) items.
// grid = (G, batch), blockDim >= ncell.  Output per image: hdr[0] = n, hdr[1..n] = cells in priority order, mask rows
// [n][nwords] (word w of row i written iff 64 w <= i; the rest is never read).  Below `min_n` candidates the spread is not
// worth a global round trip: hdr[0] = -1 and the parse kernel runs its own NMS as before.  Same device functions, same
// operation order as the parse kernel's own phases: bit-identical people.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
root_mask_kernel(ppn_decode_cfg c, const float* __restrict__ unary, int C, int ncell, int hdr_stride, int min_n,
                 int* __restrict__ hdr, unsigned long long* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, g = blockIdx.x, G = gridDim.x, b = blockIdx.y;
    const int nwords = (ncell + 63) >> 6;
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* ptr = smem + off; off += (bytes + 15) & ~size_t(15); return ptr; };
    float4* s_box = reinterpret_cast<float4*>(carve(sizeof(float4) * ncell));
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(carve(8 * ncell));
    float* s_area = reinterpret_cast<float*>(carve(4 * ncell));
    int* s_cell = reinterpret_cast<int*>(carve(4 * ncell));
    int* s_misc = reinterpret_cast<int*>(carve(4 * 40));
    const int n = sort_roots(c, unary + (size_t)b * C * ncell, ncell, s_box, s_key, s_area, s_cell, s_misc);
    int* h = hdr + (size_t)b * hdr_stride;
    if (n < min_n) {
        if (g == 0 && t == 0) h[0] = -1;
        return;
    }
    if (g == 0) {
        if (t == 0) h[0] = n;
        if (t < n) h[1 + t] = s_cell[t];
    }
    // Work item = one QUARTER (16 candidates j) of one 64-bit word of one row: 4x the items of the per-word split, so that
    // every thread of the G workgroups has one and the dependent chain per thread is 16 IoU evaluations, not 64.
    unsigned short* m16 = reinterpret_cast<unsigned short*>(mask + (size_t)b * ncell * nwords);
    const int nq_n = 4 * ((n + 63) >> 6), total = n * nq_n;     // whole words: every quarter of a word that is read is written
    for (int q = g * (int)blockDim.x + t; q < total; q += G * (int)blockDim.x) {
        const int qw = q / n, i = q - qw * n;                      // consecutive lanes: consecutive rows, same s_box[j]
        const int w = qw >> 2, j0 = qw * 16;
        if (64 * w > i) continue;                                 // word past the diagonal: never read
        const int j1 = min(j0 + 16, i);
        unsigned bits = 0u;
        if (j0 < j1) {
            const float4 bi = s_box[i];
            const float ai = s_area[i];
            int j = j0;
            for (; j + 4 <= j1; j += 4) {
                bool r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = iou_ge(bi, ai, s_box[j + u], s_area[j + u], c.nms_thr);
#pragma unroll
                for (int u = 0; u < 4; ++u) bits |= r[u] ? (1u << (j + u - j0)) : 0u;
            }
            for (; j < j1; ++j)
                if (iou_ge(bi, ai, s_box[j], s_area[j], c.nms_thr)) bits |= (1u << (j - j0));
        }
        m16[((size_t)i * nwords + w) * 4 + (qw & 3)] = (unsigned short)bits;      // little endian: quarter k = bits 16k..
    }
}

size_t root_mask_lds_bytes(int ncell) {
    auto r16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    return r16(16 * (size_t)ncell) + r16(8 * (size_t)ncell) + 2 * r16(4 * (size_t)ncell) + r16(4 * 40);
}

size_t early_lds_bytes(int ncell) {
    const size_t nwords = (ncell + 63) / 64;
    auto r16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    return r16(16 * (size_t)ncell) + r16(8 * (size_t)ncell) + r16(8 * (size_t)ncell * nwords) + 3 * r16(4 * (size_t)ncell) +
           r16(4 * 40);
}

// ------------------------------------------------------------------------------------------
// Kernel 1 (defined here, after the helpers its early-NMS role calls)
// ------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(1024)
limb_argmax_kernel(const float* __restrict__ head, int* __restrict__ out_arg, int C, int e_chan0, int S,
                   int ncell, int ncl, int Q, int NS, int E, ppn_decode_cfg cfg, int* __restrict__ early) {
    // grid = (E, batch, cell groups): a workgroup owns `ncl` consecutive cells of one (edge, image) and streams all S
    // channels of them; cell groups only exist to cut the work finer than E*batch workgroups (smooth tail)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_val = reinterpret_cast<float*>(smem);                 // [NS][ncl]
    int* s_idx = reinterpret_cast<int*>(smem + sizeof(float) * NS * ncl);

    const int edge = blockIdx.x, b = blockIdx.y, cell0 = blockIdx.z * ncl;
    const int t = threadIdx.x;
    if (early && edge == 0 && blockIdx.z == 0) {                   // workgroup-uniform
        early_root_nms(cfg, head + (size_t)b * C * ncell, ncell, smem, early + (size_t)b * early_stride(ncell));
        __syncthreads();                                           // the arg-max phase reuses the LDS
    }
    const int q = t % Q, r = t / Q;
    const float* base = head + ((size_t)b * C + e_chan0 + (size_t)edge * S) * ncell + cell0;
    using vec = typename VecT<V>::type;

    if (r < NS) {
        float best[V];
        int bidx[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { best[i] = -INFINITY; bidx[i] = 0x7fffffff; }
        if (r < S) {
            float a[V];
            vec_to_arr<V>(*reinterpret_cast<const vec*>(base + (size_t)r * ncell + V * q), a);
#pragma unroll
            for (int i = 0; i < V; ++i) { best[i] = a[i]; bidx[i] = r; }
        }
        // eight rows per trip, all eight loads issued before the first compare (a row index past the window re-reads
        // this thread's first row and is ignored: no branch between the loads, so they stay batched whatever else
        // the kernel contains -- with the in-launch decode tail compiled in, hipcc serialised the plain unrolled loop)
        for (int s0 = r + NS; s0 < S; s0 += 8 * NS) {
            float a[8][V];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * NS;
                vec_to_arr<V>(*reinterpret_cast<const vec*>(base + (size_t)(s < S ? s : r) * ncell + V * q), a[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * NS;
                const bool in = s < S;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    if (in && a[u][i] > best[i]) { best[i] = a[u][i]; bidx[i] = s; }   // strict: first max within a slice
                }
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            s_val[r * ncl + V * q + i] = best[i];
            s_idx[r * ncl + V * q + i] = bidx[i];
        }
    }
    __syncthreads();
    for (int cell = t; cell < ncl; cell += blockDim.x) {
        float bv = s_val[cell];
        int bi = s_idx[cell];
        for (int rr = 1; rr < NS; ++rr) {
            float v = s_val[rr * ncl + cell];
            int i = s_idx[rr * ncl + cell];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }     // lowest s wins ties
        }
        out_arg[((size_t)b * E + edge) * ncell + cell0 + cell] = bi;
    }
}

// ------------------------------------------------------------------------------------------
// Kernel 2: per-image parse.  One workgroup per image, blockDim = 64*ceil(ncell/64).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
parse_kernel(DecodeParams p, const float* __restrict__ head, const int* __restrict__ argmap,
             int* __restrict__ out_count, int* __restrict__ out_kp_cell, int* __restrict__ out_limb_arg,
             float* __restrict__ out_bbox, float* __restrict__ out_score) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ppn_decode_cfg& c = p.c;
    const int ncell = p.ncell, K = c.K, E = c.E, W = c.W, H = c.H;
    const int nwords = blockDim.x / kWave;
    const int t = threadIdx.x, b = blockIdx.x;

    // LDS carve (all offsets multiples of 16)
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* ptr = smem + off; off += (bytes + 15) & ~size_t(15); return ptr; };
    float4* s_box = reinterpret_cast<float4*>(carve(sizeof(float4) * ncell));
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(carve(8 * ncell));
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(carve(8 * (size_t)ncell * nwords));
    {   // the first region doubles as the delta map [K][ncell] f32 after the NMS: make sure it is large enough
        const size_t need = ((size_t)4 * K * ncell + 15) & ~size_t(15);
        if (off < need) off = need;
    }
    float* s_area = reinterpret_cast<float*>(carve(4 * ncell));
    int* s_cell = reinterpret_cast<int*>(carve(4 * ncell));          // sorted position -> cell
    int* s_sel = reinterpret_cast<int*>(carve(4 * ncell));
    unsigned short* s_kp = reinterpret_cast<unsigned short*>(carve(2 * (size_t)ncell * K));
    unsigned short* s_la = reinterpret_cast<unsigned short*>(carve(2 * (size_t)ncell * E));
    unsigned short* s_am = reinterpret_cast<unsigned short*>(carve(2 * (size_t)ncell * E));   // arg-max map of this image
    int* s_misc = reinterpret_cast<int*>(carve(4 * (36 + PPN_MAX_EDGES)));   // wave counts [32], n, nsel, nkept, -, edge table
    // after the NMS the box/key/mask region is dead: it is reused for the delta map [K][ncell] f32
    float* s_delta = reinterpret_cast<float*>(smem);

    const float* img = head + (size_t)b * p.C * ncell;
    const float gridW = (float)(c.inW / W), gridH = (float)(c.inH / H);
    const float inW = (float)c.inW, inH = (float)c.inH;

    auto delta_at = [&](int k, int cell) -> float {                   // rt_test.py:130
        return img[(size_t)k * ncell + cell] * img[(size_t)(K + k) * ncell + cell];
    };
    auto bbox_at = [&](int k, int cell) -> float4 {                   // datatest.py:63-84
        const float x = img[(size_t)(2 * K + k) * ncell + cell], y = img[(size_t)(3 * K + k) * ncell + cell];
        const float w = img[(size_t)(4 * K + k) * ncell + cell], h = img[(size_t)(5 * K + k) * ncell + cell];
        const float X = (float)(cell % W), Y = (float)(cell / W);
        const float rx = (x + X) * gridW, ry = (y + Y) * gridH;
        const float rw = inW * w, rh = inH * h;
        float4 r;
        r.x = ry - rh / 2.0f;   // ymin
        r.y = rx - rw / 2.0f;   // xmin
        r.z = ry + rh / 2.0f;   // ymax
        r.w = rx + rw / 2.0f;   // xmax
        return r;
    };

    // The tree walk at the end needs delta = resp*conf for every (keypoint, cell) and this image's arg-max map in
    // LDS.  Their global loads are issued HERE, into registers, so that their latency passes under the candidate /
    // sort / NMS phases (which own the LDS region the tables will overlay); the LDS writes follow the NMS.
    constexpr int PF = 20;                                            // table entries per thread kept in registers
    const int nd = K * ncell, na = E * ncell, pstride = blockDim.x;
    const int* am_img = argmap + (size_t)b * E * ncell;
    const unsigned long long* key_img = p.keys + (size_t)b * E * ncell;
    float pf_d[PF];
    int pf_a[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = t + u * pstride;
        const float r_ = i < nd ? img[i] : 0.f, c_ = i < nd ? img[(size_t)nd + i] : 0.f;
        pf_d[u] = r_ * c_;                                            // rt_test.py:130
        if (p.keys) pf_a[u] = i < na ? (int)(0xFFFFFFFFu - (unsigned)key_img[i]) : 0;   // key = value<<32 | ~s
        else pf_a[u] = i < na ? am_img[i] : 0;
    }

    PPN_DT(0);
    // phases 1-4 may already have run inside the arg-max launch (early_root_nms): workgroup-uniform
    const int* early_b = p.early ? p.early + (size_t)b * early_stride(ncell) : nullptr;
    const int early_n = early_b ? early_b[0] : -1;
    // ... or phases 1-3 in the root_mask_kernel launch in front of this one (fused path): sorted cells + bit matrix
    const int* root_b = p.root_hdr ? p.root_hdr + (size_t)b * early_stride(ncell) : nullptr;
    const int root_n = root_b ? root_b[0] : -1;
    int nsel;
    if (early_n < 0 && root_n >= 0) {
        PPN_DT(1);
        const int n = root_n;
        if (t < n) s_cell[t] = root_b[1 + t];
        const unsigned long long* gm = p.root_mask + (size_t)b * ncell * nwords;
        // rows of the bit matrix, eight loads in flight per thread (words past the diagonal were never written and are
        // never read by the resolve: they may hold anything)
        for (int q0 = t; q0 < n * nwords; q0 += 8 * (int)blockDim.x) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * (int)blockDim.x;
                v[u] = q < n * nwords ? gm[q] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + u * (int)blockDim.x;
                if (q < n * nwords) s_mask[q] = v[u];
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = t + u * pstride;
            if (i < na) s_am[i] = (unsigned short)pf_a[u];
        }
        __syncthreads();
        PPN_DT(2);
        greedy_resolve(n, nwords, s_mask, s_sel, s_misc + 33, 0);
        nsel = s_misc[33];
    } else if (early_n < 0) {
        // 1. candidates: delta[0] > thr, row-major (datatest.py:89)
        float d0 = 0.0f;
        bool is_c = false;
        if (t < ncell) {
            d0 = delta_at(0, t);
            is_c = d0 > c.det_thr;
        }
        const int pos = block_compact(is_c, s_misc, s_misc + 32);
        const int n = s_misc[32];
        if (is_c) {
            // ascending key == descending score, ties by ascending cell (documented tie rule)
            s_key[pos] = ((unsigned long long)(~__float_as_uint(d0)) << 32) | (unsigned)t;
        }
        __syncthreads();
        PPN_DT(1);
        // 2. rank sort; boxes/areas stored in priority order
        if (is_c) {
            const unsigned long long my = s_key[pos];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += (s_key[j] < my) ? 1 : 0;
            const float4 bb = bbox_at(0, t);
            s_box[rank] = bb;
            s_area[rank] = box_area(bb);
            s_cell[rank] = t;
        }
        // the arg-max table has its own LDS region: park the prefetched entries there before the NMS needs registers
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = t + u * pstride;
            if (i < na) s_am[i] = (unsigned short)pf_a[u];
        }
        __syncthreads();
        PPN_DT(2);
        // 3./4. greedy NMS on the root boxes (datatest.py:134-160)
        greedy_nms(n, nwords, s_box, s_area, s_mask, s_sel, s_misc + 33, c.nms_thr, 0);
        nsel = s_misc[33];
    } else {
        PPN_DT(1);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = t + u * pstride;
            if (i < na) s_am[i] = (unsigned short)pf_a[u];
        }
        PPN_DT(2);
        nsel = early_n;
    }
    PPN_DT(3);

    // 5. one lane per surviving root: tree walk through the arg-max map (datatest.py:103-127).
    // The walk is a chain of dependent look-ups, so the two tables it touches -- delta = resp*conf for every
    // (keypoint, cell) and this image's arg-max map -- are first staged in LDS with coalesced loads.
    const unsigned short NONE = 0xFFFFu;
    int root_cell = 0;
    if (t < nsel) root_cell = early_n < 0 ? s_cell[s_sel[t]] : early_b[1 + t];
    __syncthreads();                                                  // everyone is done with box/key/mask
    if (nsel > 0) {
        // the first PF entries per thread are already in registers; larger grids finish with batched loads
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = t + u * pstride;
            if (i < nd) s_delta[i] = pf_d[u];
        }
        const int stride = blockDim.x;
        for (int i0 = t + PF * stride; i0 < nd; i0 += 8 * stride) {
            float r[8], cf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * stride;
                r[u] = i < nd ? img[i] : 0.f;
                cf[u] = i < nd ? img[(size_t)nd + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * stride;
                if (i < nd) s_delta[i] = r[u] * cf[u];
            }
        }
        for (int i0 = t + PF * stride; i0 < na; i0 += 8 * stride) {
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * stride;
                if (p.keys) v[u] = i < na ? (int)(0xFFFFFFFFu - (unsigned)key_img[i]) : 0;
                else v[u] = i < na ? am_img[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * stride;
                if (i < na) s_am[i] = (unsigned short)v[u];
            }
        }
    }
    __syncthreads();
    PPN_DT(4);
    // packed skeleton table in LDS (the kernel-argument arrays would cost a scalar-memory round trip per hop)
    int* s_edge = s_misc + 36;
    if (t < E) {
        const int e = c.edge_order[t];
        s_edge[t] = e | (c.edge_src[e] << 8) | (c.edge_dst[e] << 16);
    }
    __syncthreads();
    bool keep_h = false;
    const float rcpW = 1.0f / (float)W, rcpS = 1.0f / (float)c.sW;
    // one hop of the walk (datatest.py:110-127): edge e from keypoint s to d for the human whose rows are kp / la
    auto hop = [&](int pk, unsigned short* kp, unsigned short* la) -> int {
        const int e = pk & 0xff, s_ = (pk >> 8) & 0xff, d = (pk >> 16) & 0xff;
        const unsigned short cs = kp[s_];
        if (cs == NONE) return 0;                                     // parent chain broke earlier
        const int am = s_am[e * ncell + cs];
        la[e] = (unsigned short)am;
        // exact small-integer division through a float reciprocal ((x+0.5)/d is >= 0.5/d from an integer)
        const int ch_ = (int)(((float)cs + 0.5f) * rcpW), cw_ = (int)cs - ch_ * W;
        const int ah_ = (int)(((float)am + 0.5f) * rcpS), aw_ = am - ah_ * c.sW;
        const int jh = ch_ + ah_ - c.sH / 2;
        const int jw = cw_ + aw_ - c.sW / 2;
        if (jh < 0 || jw < 0 || jh >= H || jw >= W) return 0;         // datatest.py:118
        const int cd = jh * W + jw;
        if (s_delta[d * ncell + cd] < c.det_thr) return 0;            // datatest.py:121 (== passes)
        kp[d] = (unsigned short)cd;
        return 1;
    };
    if (t < nsel) {
        unsigned short* kp = s_kp + (size_t)t * K;
        unsigned short* la = s_la + (size_t)t * E;
        for (int k = 0; k < K; ++k) kp[k] = NONE;
        for (int e = 0; e < E; ++e) la[e] = NONE;
        kp[0] = (unsigned short)root_cell;
        int found = 0;
        for (int oi = 0; oi < E; ++oi) found += hop(s_edge[oi], kp, la);
        keep_h = c.min_kp <= found;                                   // datatest.py:129
    }
    PPN_DT(5);
    const int opos = block_compact(keep_h, s_misc, s_misc + 34);
    const int nkept = s_misc[34];
    if (t == 0) out_count[b] = nkept;
    // 6. compact output rows.  The surviving humans' walk results stay in LDS; the (human, keypoint) and
    // (human, edge) items are dealt over the whole workgroup so each thread has one short load chain.
    if (keep_h && opos < c.max_humans) s_sel[opos] = t;             // output row -> walker thread
    __syncthreads();
    const int nout = min(nkept, c.max_humans);
    for (int item = t; item < nout * K; item += blockDim.x) {
        const int o = item / K, k = item - o * K;
        const unsigned short cell = s_kp[(size_t)s_sel[o] * K + k];
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        if (cell != NONE) {
            bb = bbox_at(k, cell);
            sc = s_delta[k * ncell + cell];
        }
        const size_t row = (size_t)b * c.max_humans + o;
        out_kp_cell[row * K + k] = (cell == NONE) ? -1 : (int)cell;
        reinterpret_cast<float4*>(out_bbox)[row * K + k] = bb;
        out_score[row * K + k] = sc;
    }
    for (int item = t; item < nout * E; item += blockDim.x) {
        const int o = item / E, e = item - o * E;
        const unsigned short v = s_la[(size_t)s_sel[o] * E + e];
        out_limb_arg[((size_t)b * c.max_humans + o) * E + e] = (v == NONE) ? -1 : (int)v;
    }
    __syncthreads();
    PPN_DT(6);
}

size_t parse_lds_bytes(int ncell, int K, int E) {
    const int nwords = (ncell + 63) / 64;
    auto r16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    size_t first = r16(16 * (size_t)ncell) + r16(8 * (size_t)ncell) + r16(8 * (size_t)ncell * nwords);
    if (first < r16(4 * (size_t)K * ncell)) first = r16(4 * (size_t)K * ncell);   // reused as the delta map
    return first + r16(4 * (size_t)ncell) * 3 + r16(2 * (size_t)ncell * K) + 2 * r16(2 * (size_t)ncell * E) +
           r16(4 * (36 + PPN_MAX_EDGES));
}

// ------------------------------------------------------------------------------------------
// Standalone NMS (datatest.py:134-160) -- one workgroup.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
nms_kernel(const float* __restrict__ bbox, const float* __restrict__ score, int n, float thr, int limit,
           int* __restrict__ out_sel, int* __restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwords = (n + 63) / 64;
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* ptr = smem + off; off += (bytes + 15) & ~size_t(15); return ptr; };
    float4* s_box = reinterpret_cast<float4*>(carve(sizeof(float4) * n));
    unsigned long long* s_key = reinterpret_cast<unsigned long long*>(carve(8 * n));
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(carve(8 * (size_t)n * nwords));
    float* s_area = reinterpret_cast<float*>(carve(4 * n));
    int* s_orig = reinterpret_cast<int*>(carve(4 * n));
    int* s_sel = reinterpret_cast<int*>(carve(4 * n));
    int* s_misc = reinterpret_cast<int*>(carve(16));
    const int t = threadIdx.x;
    if (t < n) {
        // order = score.argsort()[::-1] (descending; ties by ascending index); input order without score
        // monotonic float->uint map so that an ascending key means a descending float value
        unsigned k = 0u;
        if (score) {
            const unsigned u = __float_as_uint(score[t]);
            const unsigned mono = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotonic float->uint
            k = ~mono;
        }
        s_key[t] = ((unsigned long long)k << 32) | (unsigned)t;
    }
    __syncthreads();
    if (t < n) {
        const unsigned long long my = s_key[t];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (s_key[j] < my) ? 1 : 0;
        const float4 bb = reinterpret_cast<const float4*>(bbox)[t];
        s_box[rank] = bb;
        s_area[rank] = box_area(bb);
        s_orig[rank] = t;
    }
    __syncthreads();
    greedy_nms(n, nwords, s_box, s_area, s_mask, s_sel, s_misc, thr, limit);
    const int nsel = s_misc[0];
    if (t == 0) *out_count = nsel;
    if (t < nsel) out_sel[t] = s_orig[s_sel[t]];
}

size_t nms_lds_bytes(int n) {
    const int nwords = (n + 63) / 64;
    auto r16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    return r16(16 * (size_t)n) + r16(8 * (size_t)n) + r16(8 * (size_t)n * nwords) + r16(4 * (size_t)n) * 3 + 16;
}

int check_cfg(const ppn_decode_cfg* c) {
    if (!c) return ppn::fail(PPN_E_INVALID, "decode cfg is NULL");
    if (c->K < 1 || c->K > PPN_MAX_KP || c->E < 0 || c->E > PPN_MAX_EDGES)
        return ppn::fail(PPN_E_INVALID, "K=%d / E=%d out of range", c->K, c->E);
    if (c->H < 1 || c->W < 1 || c->sH < 1 || c->sW < 1 || c->inH < c->H || c->inW < c->W)
        return ppn::fail(PPN_E_INVALID, "bad grid geometry H=%d W=%d sH=%d sW=%d", c->H, c->W, c->sH, c->sW);
    if (c->H * c->W > 0xFFFE) return ppn::fail(PPN_E_UNSUPPORTED, "grid of %d cells is too large", c->H * c->W);
    for (int e = 0; e < c->E; ++e) {
        if (c->edge_src[e] < 0 || c->edge_src[e] >= c->K || c->edge_dst[e] < 0 || c->edge_dst[e] >= c->K ||
            c->edge_order[e] < 0 || c->edge_order[e] >= c->E)
            return ppn::fail(PPN_E_INVALID, "skeleton table entry %d out of range", e);
    }
    return PPN_OK;
}

}  // namespace

// workspace layout: arg-max map i32 [B][E][ncell] (padded to a 128-byte multiple) | early lists i32 [B][early_stride]
static size_t ws_argmap_bytes(const ppn_decode_cfg* c, int batch) {
    return (((size_t)batch * c->E * c->H * c->W * 4) + 127) & ~size_t(127);
}
extern "C" size_t ppn_decode_workspace_bytes(const ppn_decode_cfg* cfg, int32_t batch) {
    if (!cfg || batch < 0) return 0;
    return ws_argmap_bytes(cfg, batch) + (size_t)batch * early_stride(cfg->H * cfg->W) * 4;
}

static int launch_limb_argmax(const ppn_decode_cfg* cfg, const float* head, int32_t batch, int32_t* out_arg,
                              int32_t* early, void* stream, bool* early_used) {
    if (early_used) *early_used = false;
    if (int rc = check_cfg(cfg)) return rc;
    if (batch == 0 || cfg->E == 0) return PPN_OK;
    if (!head || !out_arg || batch < 0) return ppn::fail(PPN_E_INVALID, "ppn_limb_argmax: NULL pointer or batch<0");
    const int ncell = cfg->H * cfg->W, S = cfg->sH * cfg->sW;
    const int C = 6 * cfg->K + cfg->E * S;
    const int V = (ncell % 4 == 0 && (reinterpret_cast<uintptr_t>(head) % 16 == 0)) ? 4 : 1;
    // cell groups: cut every (edge, image) into CS pieces of >= 64 cells so that the grid is several times the
    // 256 CUs x 3 resident workgroups (17 x 32 = 544 workgroups left a tail of half-empty CUs)
    static const int cs_env = getenv("PPN_ARGMAX_SPLIT") ? atoi(getenv("PPN_ARGMAX_SPLIT")) : 0;
    int CS = cs_env > 0 ? cs_env : 3;          // measured at 24x24, batch 32: 105 us (1 group) -> 97 us (3 groups)
    while (CS > 1 && (ncell % (CS * V) != 0 || ncell / CS < 64)) --CS;
    const int ncl = ncell / CS;
    const int Q = ncl / V;
    if (Q > 1024) return ppn::fail(PPN_E_UNSUPPORTED, "grid of %d cells is too large for ppn_limb_argmax", ncell);
    int NS = 576 / Q;
    if (NS < 1) NS = 1;
    if (NS > 32) NS = 32;
    if (NS > S) NS = S;
    int threads = ((NS * Q + 63) / 64) * 64;
    if (threads > 1024) { NS = 1024 / Q; threads = ((NS * Q + 63) / 64) * 64; }
    // the early root-NMS role needs one thread per cell and its own LDS carve (62 KB at 24x24: two workgroups per CU
    // instead of three -- measured harmless for this HBM-bound launch: 96.8-97.3 us vs 97.4-97.6 us)
    static const bool early_off = getenv("PPN_DECODE_EARLY_NMS") && atoi(getenv("PPN_DECODE_EARLY_NMS")) == 0;
    size_t lds = (size_t)NS * ncl * 8;
    if (early_off || threads < ncell || std::max(lds, early_lds_bytes(ncell)) > 96 * 1024) early = nullptr;
    if (early) lds = std::max(lds, early_lds_bytes(ncell));
    if (early_used) *early_used = early != nullptr;
    if (lds > 160 * 1024) return ppn::fail(PPN_E_UNSUPPORTED, "limb_argmax LDS %zu too large", lds);
    dim3 grid(cfg->E, batch, CS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (V == 4) {
        {
            static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
            PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(limb_argmax_kernel<4>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        hipLaunchKernelGGL(limb_argmax_kernel<4>, grid, dim3(threads), lds, st, head, out_arg, C, 6 * cfg->K, S,
                           ncell, ncl, Q, NS, cfg->E, *cfg, early);
    } else {
        {
            static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
            PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(limb_argmax_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        hipLaunchKernelGGL(limb_argmax_kernel<1>, grid, dim3(threads), lds, st, head, out_arg, C, 6 * cfg->K, S,
                           ncell, ncl, Q, NS, cfg->E, *cfg, early);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

extern "C" int ppn_limb_argmax(const ppn_decode_cfg* cfg, const float* head, int32_t batch, int32_t* out_arg,
                               void* stream) {
    return launch_limb_argmax(cfg, head, batch, out_arg, nullptr, stream, nullptr);
}

extern "C" int ppn_decode(const ppn_decode_cfg* cfg, const float* head, int32_t batch, void* workspace,
                          int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg, float* out_bbox,
                          float* out_score, void* stream) {
    if (int rc = check_cfg(cfg)) return rc;
    if (batch < 0) return ppn::fail(PPN_E_INVALID, "ppn_decode: batch < 0");
    if (batch == 0) return PPN_OK;
    if (!head || !workspace || !out_count || !out_kp_cell || !out_limb_arg || !out_bbox || !out_score)
        return ppn::fail(PPN_E_INVALID, "ppn_decode: NULL pointer");
    if (cfg->max_humans < 1) return ppn::fail(PPN_E_INVALID, "ppn_decode: max_humans < 1");
    if (reinterpret_cast<uintptr_t>(out_bbox) % 16 != 0 || reinterpret_cast<uintptr_t>(workspace) % 4 != 0)
        return ppn::fail(PPN_E_INVALID, "ppn_decode: out_bbox must be 16-byte aligned");
    const int ncell = cfg->H * cfg->W;
    const size_t lds = parse_lds_bytes(ncell, cfg->K, cfg->E);
    if (ncell > 1024 || lds > 160 * 1024)
        return ppn::fail(PPN_E_UNSUPPORTED, "grid of %d cells needs %zu B of LDS (max 163840)", ncell, lds);
    int32_t* argmap = static_cast<int32_t*>(workspace);
    int32_t* early = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + ws_argmap_bytes(cfg, batch));
    bool early_used = false;
    if (int rc = launch_limb_argmax(cfg, head, batch, argmap, cfg->E > 0 ? early : nullptr, stream, &early_used)) return rc;
    DecodeParams p;
    p.c = *cfg;
    p.ncell = ncell;
    p.S = cfg->sH * cfg->sW;
    p.C = 6 * cfg->K + cfg->E * p.S;
    p.keys = nullptr;
    p.early = early_used ? early : nullptr;
    p.root_hdr = nullptr;
    p.root_mask = nullptr;
    const int threads = ((ncell + 63) / 64) * 64;
    {
        static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(parse_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(parse_kernel, dim3(batch), dim3(threads), lds, static_cast<hipStream_t>(stream), p, head,
                       argmap, out_count, out_kp_cell, out_limb_arg, out_bbox, out_score);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

// workspace of the fused decode: root headers i32 [B][early_stride] (whole 128-byte lines) | bit matrices u64 [B][ncell][nwords]
static size_t ws_root_hdr_bytes(const ppn_decode_cfg* c, int batch) {
    return (((size_t)batch * early_stride(c->H * c->W) * 4) + 127) & ~size_t(127);
}
extern "C" size_t ppn_decode_fused_workspace_bytes(const ppn_decode_cfg* cfg, int32_t batch) {
    if (!cfg || batch < 0) return 0;
    const size_t ncell = (size_t)cfg->H * cfg->W;
    return ws_root_hdr_bytes(cfg, batch) + (size_t)batch * ncell * ((ncell + 63) / 64) * 8;
}

static int decode_fused_impl(const ppn_decode_cfg* cfg, const float* unary, const uint64_t* keys, int32_t batch,
                             void* workspace, int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg,
                             float* out_bbox, float* out_score, void* stream) {
    if (int rc = check_cfg(cfg)) return rc;
    if (batch < 0) return ppn::fail(PPN_E_INVALID, "ppn_decode_fused: batch < 0");
    if (batch == 0) return PPN_OK;
    if (!unary || !keys || !out_count || !out_kp_cell || !out_limb_arg || !out_bbox || !out_score)
        return ppn::fail(PPN_E_INVALID, "ppn_decode_fused: NULL pointer");
    if (cfg->max_humans < 1) return ppn::fail(PPN_E_INVALID, "ppn_decode_fused: max_humans < 1");
    if (reinterpret_cast<uintptr_t>(out_bbox) % 16 != 0)
        return ppn::fail(PPN_E_INVALID, "ppn_decode_fused: out_bbox must be 16-byte aligned");
    if (workspace && reinterpret_cast<uintptr_t>(workspace) % 16 != 0)
        return ppn::fail(PPN_E_INVALID, "ppn_decode_fused_ws: workspace must be 16-byte aligned");
    const int ncell = cfg->H * cfg->W;
    const size_t lds = parse_lds_bytes(ncell, cfg->K, cfg->E);
    if (ncell > 1024 || lds > 160 * 1024)
        return ppn::fail(PPN_E_UNSUPPORTED, "grid of %d cells needs %zu B of LDS (max 163840)", ncell, lds);
    DecodeParams p;
    p.c = *cfg;
    p.ncell = ncell;
    p.S = cfg->sH * cfg->sW;
    p.C = 6 * cfg->K;                                                 // compact unary tensor
    p.keys = reinterpret_cast<const unsigned long long*>(keys);
    p.early = nullptr;
    p.root_hdr = nullptr;
    p.root_mask = nullptr;
    const int threads = ((ncell + 63) / 64) * 64;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Spread root NMS (root_mask_kernel): G workgroups per image compute the pairwise-IoU bit matrix in front of the parse.
    // PPN_DECODE_SPREAD=0 keeps the single-kernel form; PPN_DECODE_SPREAD_MIN: candidates below which an image leaves
    // the matrix to its parse workgroup (default 128: ~8 k IoU evaluations, a few microseconds on one CU).
    static const int spread_g = getenv("PPN_DECODE_SPREAD") ? atoi(getenv("PPN_DECODE_SPREAD")) : 8;
    static const int spread_min = getenv("PPN_DECODE_SPREAD_MIN") ? atoi(getenv("PPN_DECODE_SPREAD_MIN")) : 128;
    if (workspace && spread_g > 0 && threads <= 1024) {
        int* hdr = static_cast<int*>(workspace);
        unsigned long long* mask = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + ws_root_hdr_bytes(cfg, batch));
        const size_t rlds = root_mask_lds_bytes(ncell);
        {
            static int max_lds_set = 0;
            PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(root_mask_kernel),
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
        }
        hipLaunchKernelGGL(root_mask_kernel, dim3(spread_g, batch), dim3(threads), rlds, st, *cfg, unary, p.C, ncell,
                           early_stride(ncell), spread_min, hdr, mask);
        PPN_LAUNCH_CHECK();
        p.root_hdr = hdr;
        p.root_mask = mask;
    }
    {
        static int max_lds_set = 0;
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(parse_kernel),
                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(parse_kernel, dim3(batch), dim3(threads), lds, st, p, unary,
                       static_cast<const int*>(nullptr), out_count, out_kp_cell, out_limb_arg, out_bbox, out_score);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

extern "C" int ppn_decode_fused(const ppn_decode_cfg* cfg, const float* unary, const uint64_t* keys, int32_t batch,
                                int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg, float* out_bbox,
                                float* out_score, void* stream) {
    return decode_fused_impl(cfg, unary, keys, batch, nullptr, out_count, out_kp_cell, out_limb_arg, out_bbox, out_score,
                             stream);
}

extern "C" int ppn_decode_fused_ws(const ppn_decode_cfg* cfg, const float* unary, const uint64_t* keys, int32_t batch,
                                   void* workspace, int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg,
                                   float* out_bbox, float* out_score, void* stream) {
    if (!workspace) return ppn::fail(PPN_E_INVALID, "ppn_decode_fused_ws: NULL workspace");
    return decode_fused_impl(cfg, unary, keys, batch, workspace, out_count, out_kp_cell, out_limb_arg, out_bbox, out_score,
                             stream);
}

extern "C" int ppn_nms(const float* bbox, const float* score, int32_t n, float thresh, int32_t limit,
                       int32_t* out_sel, int32_t* out_count, void* stream) {
    if (n < 0 || !out_count) return ppn::fail(PPN_E_INVALID, "ppn_nms: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) {                                                     // datatest.py:135-136
        PPN_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
        return PPN_OK;
    }
    if (!bbox || !out_sel) return ppn::fail(PPN_E_INVALID, "ppn_nms: NULL pointer");
    if (reinterpret_cast<uintptr_t>(bbox) % 16 != 0) return ppn::fail(PPN_E_INVALID, "ppn_nms: bbox must be 16-byte aligned");
    const size_t lds = nms_lds_bytes(n);
    if (n > 1024 || lds > 160 * 1024)
        return ppn::fail(PPN_E_UNSUPPORTED, "ppn_nms: n=%d needs %zu B of LDS (max 163840)", n, lds);
    const int threads = ((n + 63) / 64) * 64;
    {
        static int max_lds_set = 0;   // the attribute sticks to the function: set it when it grows
        PPN_LDS_ONCE(max_lds_set, reinterpret_cast<const void*>(nms_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(threads), lds, st, bbox, score, n, thresh, limit, out_sel,
                       out_count);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}
