// Weight gradient of the small-channel stem convolutions (drn.py:123-133 layer0 7x7 3->16, layer1 3x3 16->16,
// layer2 3x3 stride 2 16->32), bf16.  These are tall-skinny reductions -- a few thousand outputs summed over
// millions of pixels -- and HBM-bound: x and dy must each be read once (2 x 151 MB at batch 32 for layer1).  The
// generic 128 x 128 GEMM tile of wgrad.hip wastes > 98 % of its MFMA work on them.
//
//   workgroup: persistent loop over 8 x 64 output-pixel tiles; dy tile and the x tile with its halo are staged in
//              LDS exactly as they lie in memory ([pixel][channel], 16-byte chunks, zero-filled outside the image)
//   wave:      runs of 32 consecutive output pixels = one MFMA depth step; for every filter tap one
//              v_mfma_f32_16x16x32_bf16 per 16 output channels: D[co][ci] += dy[p][co] * x[p @ tap][ci].  Both
//              operands are pixel-major in LDS, so both are fetched with the transposing read ds_read_b64_tr_b16;
//              a tap only shifts the row addresses of the x operand.  Input channels beyond the tensor's (8 of the
//              16 columns for layer0) point at a zeroed LDS line.
//   result:    every wave keeps all taps in registers (<= 49 x 4 VGPRs), waves are folded through LDS, every
//              workgroup writes one partial, wgrad_fold sums the partials in a fixed order (deterministic).
#include "common.h"
#include "conv_common.h"

namespace {

using namespace ppnconv;

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int TW = 64, kThreads = 256;
// output rows per tile: 8, except the stride-2 layer (its 17 x 129 x 16-channel input tile alone is 70 KB: 4 rows -> 53 KB
// of LDS in all, three workgroups per CU hide each other's staging instead of one)
template <int S> constexpr int tile_rows() { return S == 2 ? 4 : 8; }

struct SwArgs {
    const __bf16* x;      // NHWC [B][H][W][CIP]
    const __bf16* dy;     // NHWC [B][Ho][Wo][CO]
    float* partial;       // [gridDim.x][CO][CIP][KS*KS]
    int B, H, W, Ho, Wo;
    int tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

template <int CO, int CIP, int KS, int S>
__global__ void __launch_bounds__(kThreads, 2) stem_wgrad_kernel(SwArgs a) {
    constexpr int TH = tile_rows<S>();
    constexpr int PAD = KS / 2;
    constexpr int XH = (TH - 1) * S + KS, XW = (TW - 1) * S + KS;     // x tile incl. halo
    constexpr int XPB = CIP * 2, YPB = CO * 2;                         // bytes per pixel
    constexpr int XBYTES = XH * XW * XPB, YBYTES = TH * TW * YPB;
    constexpr int NCB = CO / 16;                                       // 16-channel blocks of cout
    constexpr int NT = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                       // [XH][XW][CIP]
    char* ys = smem + XBYTES;              // [TH][TW][CO]
    char* zs = ys + YBYTES;                // 16 zero bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    if (tid < 4) reinterpret_cast<unsigned*>(zs)[tid] = 0u;

    f32x4 acc[NT][NCB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t_ = tile;
        const int tx = t_ % a.tiles_x; t_ /= a.tiles_x;
        const int ty = t_ % a.tiles_y;
        const int b = t_ / a.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
        __syncthreads();                                               // previous tile fully consumed
        // ---- stage x (with halo) and dy, 16 bytes per item, zeros outside the tensors --------------------------------
        constexpr int XCH = XPB / 16, YCH = YPB / 16;
        for (int it = tid; it < XH * XW * XCH; it += kThreads) {
            const int ch = it % XCH, px = it / XCH;
            const int yy = px / XW, xx = px % XW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.x) +
                                                    ((((size_t)b * a.H + iy) * a.W + ix) * XPB + ch * 16));
            *reinterpret_cast<uint4*>(xs + (size_t)px * XPB + ch * 16) = v;
        }
        for (int it = tid; it < TH * TW * YCH; it += kThreads) {
            const int ch = it % YCH, px = it / YCH;
            const int yy = px / TW, xx = px % TW;
            const int oy = oy0 + yy, ox = ox0 + xx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (oy < a.Ho && ox < a.Wo)
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.dy) +
                                                    ((((size_t)b * a.Ho + oy) * a.Wo + ox) * YPB + ch * 16));
            *reinterpret_cast<uint4*>(ys + (size_t)px * YPB + ch * 16) = v;
        }
        __syncthreads();
        // ---- 16 runs of 32 output pixels per tile, 4 per wave ------------------------------------------------------------
#pragma unroll 1
        for (int run = wave; run < TH * (TW / 32); run += 4) {
            const int ry = run >> 1, cx0 = (run & 1) * 32;
            // this lane's pixel inside the run for the two transposed reads of an operand
            const int j0 = 8 * g + q, j1 = 8 * g + 4 + q;
            bf16x8 af[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const s16x4 a0 = tr_read(ys + (size_t)(ry * TW + cx0 + j0) * YPB + c * 32 + 8 * p);
                const s16x4 a1 = tr_read(ys + (size_t)(ry * TW + cx0 + j1) * YPB + c * 32 + 8 * p);
                af[c] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            const bool real = 4 * p < CIP;                             // columns 4p..4p+3 exist in memory
            const char* xb0 = real ? xs + ((size_t)(ry * S) * XW + (cx0 + j0) * S) * XPB + 8 * p : zs;
            const char* xb1 = real ? xs + ((size_t)(ry * S) * XW + (cx0 + j1) * S) * XPB + 8 * p : zs;
            const int tap_stride = real ? 1 : 0;
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const int off = (dy * XW + dx) * XPB * tap_stride;
                    const s16x4 b0 = tr_read(xb0 + off);
                    const s16x4 b1 = tr_read(xb1 + off);
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        acc[dy * KS + dx][c] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bf, acc[dy * KS + dx][c], 0, 0, 0);
                }
        }
    }
    // ---- fold the 4 waves through LDS, one filter ROW of taps at a time (a [4][NT][CO][CIP] buffer was 100 KB for the 7x7
    // layer and alone held the kernel to one workgroup per CU; a row is <= 14 KB), write this workgroup's partial
    // [co][ci][tap].  The sum runs over the waves in the order 0, 1, 2, 3 as before: same bits.
    constexpr int NOUT = CO * CIP * NT, ROW = KS * CO * CIP;
    float* red = reinterpret_cast<float*>(smem);                       // [4][KS][CO][CIP]
    float* out = a.partial + (size_t)blockIdx.x * NOUT;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
        __syncthreads();
        if (li < CIP) {
#pragma unroll
            for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[((wave * KS + dx) * CO + c * 16 + 4 * g + r) * CIP + li] = acc[dy * KS + dx][c][r];
        }
        __syncthreads();
        for (int o = tid; o < ROW; o += kThreads) {
            const int dx = o % KS, ci = (o / KS) % CIP, co = o / (KS * CIP);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((w * KS + dx) * CO + co) * CIP + ci];
            out[(co * CIP + ci) * NT + dy * KS + dx] = v;
        }
    }
}

// Layer 0 (7x7) on a FOUR-channel input (3 image channels + a zero: what PPNTrainer.forward keeps for this purpose).  The
// generic kernel above, on the 8-channel padded input, spends one MFMA per TAP with 16 N columns of which 3 carry data --
// 49 MFMAs and 98 transposing reads per 32 pixels, 196 accumulator registers (two workgroups per CU): 270 us at batch 32
// for 22 GFLOP, the last thing the side stream does in a training step.  Here the N dimension runs over (dx, channel): the
// four channels of four CONSECUTIVE pixels are the 4 x 8 bytes one transposing read gathers into 16 columns
// n = 4 (dx mod 4) + c, so a filter ROW is two MFMAs (dx 0..3 and 4..7; column dx = 7 is discarded): 14 MFMAs / 28 reads
// per 32 pixels, 56 accumulator registers, four workgroups per CU, half the x bytes.  Same staging, same fold order;
// partial layout [co][ci 0..3][tap].
__global__ void __launch_bounds__(kThreads, 4) stem_wgrad7_kernel(SwArgs a) {
    constexpr int CO = 16, CIP = 4, KS = 7, TH = 8, PAD = 3, NB = 2;
    constexpr int XH = TH - 1 + KS, XW = TW - 1 + KS;                  // x tile incl. halo
    constexpr int XPB = CIP * 2, YPB = CO * 2;                         // bytes per pixel
    constexpr int XBYTES = XH * XW * XPB;
    constexpr int NT = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                       // [XH][XW][CIP]  (+ 16 bytes: the dx = 7 column of the last pixel reads past the tile)
    char* ys = smem + XBYTES + 16;         // [TH][TW][CO]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;

    f32x4 acc[KS][NB];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int c = 0; c < NB; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int t_ = tile;
        const int tx = t_ % a.tiles_x; t_ /= a.tiles_x;
        const int ty = t_ % a.tiles_y;
        const int b = t_ / a.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 - PAD, ix0 = ox0 - PAD;
        __syncthreads();                                               // previous tile fully consumed
        for (int it = tid; it < XH * XW; it += kThreads) {             // one 8-byte pixel per item
            const int yy = it / XW, xx = it % XW;
            const int iy = iy0 + yy, ix = ix0 + xx;
            uint2 v = make_uint2(0, 0);
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.x) + (((size_t)b * a.H + iy) * a.W + ix) * XPB);
            *reinterpret_cast<uint2*>(xs + (size_t)it * XPB) = v;
        }
        if (tid == 0) *reinterpret_cast<uint4*>(xs + XBYTES) = make_uint4(0, 0, 0, 0);
        constexpr int YCH = YPB / 16;
        for (int it = tid; it < TH * TW * YCH; it += kThreads) {
            const int ch = it % YCH, px = it / YCH;
            const int yy = px / TW, xx = px % TW;
            const int oy = oy0 + yy, ox = ox0 + xx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (oy < a.Ho && ox < a.Wo)
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.dy) +
                                                    ((((size_t)b * a.Ho + oy) * a.Wo + ox) * YPB + ch * 16));
            *reinterpret_cast<uint4*>(ys + (size_t)px * YPB + ch * 16) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int run = wave; run < TH * (TW / 32); run += 4) {
            const int ry = run >> 1, cx0 = (run & 1) * 32;
            const int j0 = 8 * g + q, j1 = 8 * g + 4 + q;
            const s16x4 a0 = tr_read(ys + (size_t)(ry * TW + cx0 + j0) * YPB + 8 * p);
            const s16x4 a1 = tr_read(ys + (size_t)(ry * TW + cx0 + j1) * YPB + 8 * p);
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            // lane (q, p): the four channels of pixel (run pixel j) + p -> after the transpose, column n = 4 p + channel
            const char* xb0 = xs + ((size_t)ry * XW + cx0 + j0 + p) * XPB;
            const char* xb1 = xs + ((size_t)ry * XW + cx0 + j1 + p) * XPB;
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int off = (dy * XW + 4 * nb) * XPB;
                    const s16x4 b0 = tr_read(xb0 + off);
                    const s16x4 b1 = tr_read(xb1 + off);
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[dy][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[dy][nb], 0, 0, 0);
                }
        }
    }
    // ---- fold the 4 waves through LDS one filter row at a time, in the order 0, 1, 2, 3; write [co][ci][tap] ------------
    constexpr int NOUT = CO * CIP * NT, ROW = KS * CO * CIP;
    float* red = reinterpret_cast<float*>(smem);                       // [4][NB][CO][16]
    float* out = a.partial + (size_t)blockIdx.x * NOUT;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * NB + nb) * CO + 4 * g + r) * 16 + li] = acc[dy][nb][r];
        __syncthreads();
        for (int o = tid; o < ROW; o += kThreads) {
            const int dx = o % KS, ci = (o / KS) % CIP, co = o / (KS * CIP);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((w * NB + (dx >> 2)) * CO + co) * 16 + 4 * (dx & 3) + ci];
            out[(co * CIP + ci) * NT + dy * KS + dx] = v;
        }
    }
}

// dw[o] = beta*dw[o] + sum_wg partial[wg][o].  64 outputs x 16 partial groups per workgroup: group j sums the partials
// j, j + 16, ... (eight 256-byte-coalesced loads in flight), the 16 group sums are folded through LDS in the order 0..15 --
// a fixed order, so run-to-run reproducible.  (One thread per output walking all <= 1024 partials was a chain of 128
// dependent load rounds: 72 us per launch, as long as the reduction it completes.)
constexpr int kFoldOut = 64, kFoldGroups = 16;
__global__ void __launch_bounds__(kFoldOut * kFoldGroups) stem_wgrad_fold_kernel(const float* __restrict__ partial, int nwg,
                                                                                int n, float beta, float* __restrict__ dw) {
    __shared__ float red[kFoldGroups][kFoldOut];
    const int ol = threadIdx.x % kFoldOut, grp = threadIdx.x / kFoldOut;
    const int o = blockIdx.x * kFoldOut + ol;
    float acc = 0.f;
    if (o < n) {
        int w = grp;
        for (; w + 7 * kFoldGroups < nwg; w += 8 * kFoldGroups) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(w + j * kFoldGroups) * n + o];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; w < nwg; w += kFoldGroups) acc += partial[(size_t)w * n + o];
    }
    red[grp][ol] = acc;
    __syncthreads();
    if (grp == 0 && o < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < kFoldGroups; ++j) t += red[j][ol];
        dw[o] = beta != 0.f ? beta * dw[o] + t : t;
    }
}

template <int CO, int CIP, int KS, int S>
constexpr int lds_bytes() {
    constexpr int TH = tile_rows<S>();
    constexpr int XH = (TH - 1) * S + KS, XW = (TW - 1) * S + KS;
    constexpr int tiles = XH * XW * CIP * 2 + TH * TW * CO * 2 + 16;
    constexpr int red = 4 * KS * CO * CIP * 4;
    return tiles > red ? tiles : red;
}

template <int CO, int CIP, int KS, int S>
int launch(const SwArgs& a, int nwg, float beta, float* dw, hipStream_t st) {
    static int lds_set = 0;
    constexpr int lds = lds_bytes<CO, CIP, KS, S>();
    auto k = stem_wgrad_kernel<CO, CIP, KS, S>;
    PPN_LDS_ONCE(lds_set, reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<<<nwg, kThreads, lds, st>>>(a);
    PPN_LAUNCH_CHECK();
    constexpr int n = CO * CIP * KS * KS;
    stem_wgrad_fold_kernel<<<(n + kFoldOut - 1) / kFoldOut, kFoldOut * kFoldGroups, 0, st>>>(a.partial, nwg, n, beta, dw);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

}  // namespace

namespace ppn {

bool stem_wgrad_supported(const ppn_wgrad_desc* d) {
    if (d->dtype != PPN_BF16 || d->dilation != 1 || d->pad != d->ksize / 2) return false;
    if (d->ksize == 7 && d->stride == 1 && (d->cin == 8 || d->cin == 4) && d->cout == 16) return true;
    if (d->ksize == 3 && d->stride == 1 && d->cin == 16 && d->cout == 16) return true;
    if (d->ksize == 3 && d->stride == 2 && d->cin == 16 && d->cout == 32) return true;
    return false;
}

static int stem_wgrad_grid(const ppn_wgrad_desc* d) {
    const int TH = d->stride == 2 ? tile_rows<2>() : tile_rows<1>();
    const long long tiles = (long long)d->batch * ((d->out_h + TH - 1) / TH) * ((d->out_w + TW - 1) / TW);
    // persistent workgroups: four per CU for the 3x3 layers (37-53 KB of LDS: three or four co-reside and hide each
    // other's staging latency; 152 -> 124 us on layer1), two for the 7x7 layer (32 KB of LDS, 256 VGPRs: two resident)
    const long long cap = (d->ksize == 7 && d->cin == 8) ? 512 : 1024;
    return (int)(tiles < cap ? tiles : cap);
}

size_t stem_wgrad_workspace_bytes(const ppn_wgrad_desc* d) {
    return (size_t)stem_wgrad_grid(d) * d->cout * d->cin * d->ksize * d->ksize * sizeof(float);
}

int stem_wgrad_launch(const ppn_wgrad_desc* d, hipStream_t st) {
    SwArgs a{};
    a.x = (const __bf16*)d->x;
    a.dy = (const __bf16*)d->dy;
    a.partial = (float*)d->workspace;
    a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Ho = d->out_h; a.Wo = d->out_w;
    a.tiles_x = (d->out_w + TW - 1) / TW;
    const int TH = d->stride == 2 ? tile_rows<2>() : tile_rows<1>();
    a.tiles_y = (d->out_h + TH - 1) / TH;
    a.ntiles = a.tiles_x * a.tiles_y * d->batch;
    const int nwg = stem_wgrad_grid(d);
    if (d->ksize == 7 && d->cin == 4) {
        static int lds_set = 0;
        constexpr int lds = (7 + 7) * (TW - 1 + 7) * 8 + 16 + 8 * TW * 32;             // x tile + slack + dy tile (>= the fold's 8 KB)
        PPN_LDS_ONCE(lds_set, reinterpret_cast<const void*>(stem_wgrad7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        stem_wgrad7_kernel<<<nwg, kThreads, lds, st>>>(a);
        PPN_LAUNCH_CHECK();
        constexpr int n = 16 * 4 * 49;
        stem_wgrad_fold_kernel<<<(n + kFoldOut - 1) / kFoldOut, kFoldOut * kFoldGroups, 0, st>>>(a.partial, nwg, n, d->beta, d->dw);
        PPN_LAUNCH_CHECK();
        return PPN_OK;
    }
    if (d->ksize == 7) return launch<16, 8, 7, 1>(a, nwg, d->beta, d->dw, st);
    if (d->stride == 1) return launch<16, 16, 3, 1>(a, nwg, d->beta, d->dw, st);
    return launch<32, 16, 3, 2>(a, nwg, d->beta, d->dw, st);
}

}  // namespace ppn
