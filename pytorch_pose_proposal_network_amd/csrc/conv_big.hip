// Large-tile variant of the fused implicit-GEMM convolution (see conv.hip for the GEMM view and the
// epilogue algebra).  Used for every layer whose Cin is a multiple of the K step and Cout >= 128, i.e.
// 93 % of the FLOPs of DRN-D-22.
//
// Why a second kernel: on gfx950 a CU's global->LDS fill path moves ~64 B/clk while its four SIMDs retire
// 4 x 1024 bf16 FLOP/clk, so a 128x128x64 tile (64 FLOP per staged byte) is fill-bound at ~1/3 of the MFMA
// peak.  A (BP x 256) tile with 8 waves doubles the FLOPs per staged byte:
//   workgroup  512 threads = 8 waves as WC x WP (channels x pixels), one workgroup per CU
//   tile       BP in {128,192,256} pixels x BC in {128,256} channels x 64 (bf16) / 32 (f32) deep
//   LDS        2 stages x (BP+BC) x 128 B (<= 128 KiB), rows XOR-swizzled through the SOURCE address
//   wave       (BC/WC) x (BP/WP) outputs in 16x16 MFMA tiles, weights = A operand, pixels = B operand
// BP is picked per layer so that the number of workgroups is close to a multiple of the 256 CUs
// (e.g. 192 px x 256 ch on the 48x48x512 layers at batch 32: 768 workgroups = 3 full rounds).
// Epilogue straight from the accumulators: each lane owns 4 consecutive channels of one pixel.
#include <type_traits>

#include "conv_common.h"

namespace {

using namespace ppnconv;

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

template <typename T, int BP, int BC, int WP, int WC>
__global__ void __launch_bounds__(512, 2) conv_igemm_big_kernel(ConvKArgs a) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int BK = 8 * EPC;
    constexpr int ES = sizeof(T);
    constexpr int NXI = BP / 64, NWI = BC / 64;                    // load instructions per thread and K step
    constexpr int TP = BP / WP / 16, TC = BC / WC / 16;
    constexpr int STAGE = (BP + BC) * 128;
    static_assert(WP * WC == 8, "8 waves");
    static_assert(BP % 64 == 0 && BC % 64 == 0 && (BP / WP) % 16 == 0 && (BC / WC) % 16 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;

    int ptile, ctile;
    {
        const int nb = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, loc = id >> 3, q = nb >> 3, r = nb & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        ptile = logical / a.n_ctiles;
        ctile = logical - ptile * a.n_ctiles;
    }
    const int m0 = ptile * BP, c0 = ctile * BC;

    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2));
    const int ntaps = a.ks * a.ks;
    int xbase[NXI];
    unsigned xmask[NXI];
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int row = (j * 8 + wave) * 8 + lrow;
        const int m = m0 + row;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int b = mm / a.HoWo, rem = mm - b * a.HoWo;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        xbase[j] = ((b * a.H + iy0) * a.W + ix0) * a.Cin + chunk * EPC;
        unsigned mk = 0;
        for (int t = 0; t < ntaps; ++t) {
            const int dy = t / a.ks, dx = t - dy * a.ks;
            const int iy = iy0 + dy * a.dil, ix = ix0 + dx * a.dil;
            if (vm && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) mk |= (1u << t);
        }
        xmask[j] = mk;
    }
    const char* wptr[NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = (j * 8 + wave) * 8 + lrow;
        wptr[j] = a.wgt + ((size_t)(c0 + row) * a.Ktot + chunk * EPC) * ES;
    }

    const int nsteps = a.Ktot / BK;
    int u_tap = 0, u_ci0 = 0, u_dy = 0, u_dx = 0;
    auto issue_loads = [&](int step, int buf) {
        char* xs = smem + buf * STAGE;
        char* ws = xs + BP * 128;
        const int tapoff = (u_dy * a.dil * a.W + u_dx * a.dil) * a.Cin + u_ci0;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const bool ok = (xmask[j] >> u_tap) & 1u;
            const char* g = ok ? a.src + (ptrdiff_t)(xbase[j] + tapoff) * ES : a.zero;
            glds16(g, xs + (j * 8 + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) glds16(wptr[j] + (size_t)step * BK * ES, ws + (j * 8 + wave) * 1024);
        u_ci0 += BK;
        if (u_ci0 >= a.Cin) {
            u_ci0 = 0; ++u_tap; ++u_dx;
            if (u_dx == a.ks) { u_dx = 0; ++u_dy; }
        }
    };

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
            for (int j = 0; j < TP; ++j) xf[j] = *reinterpret_cast<const f32x4*>(xs + j * 16 * 128 + foff[ks]);
#pragma unroll
            for (int i = 0; i < TC; ++i) wf[i] = *reinterpret_cast<const f32x4*>(ws + i * 16 * 128 + foff[ks]);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma_step(acc[i][j], wf[i], xf[j], (T*)nullptr);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue straight from the accumulators -------------------------------------------------
    // lane: pixel = tile column frow, channels 4*fq .. 4*fq+3 of channel tile i
    // (generic lambda over a compile-time pixel-tile index keeps every acc[][] access static: a runtime j
    //  would push the accumulators through scratch memory)
    auto epilogue_col = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m = m0 + wp * (BP / WP) + j * 16 + frow;
        const bool mvalid = m < a.M;
        int nb = 0, np = 0;
        if (a.nchw) { nb = m / a.HoWo; np = m - nb * a.HoWo; }
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int c = c0 + wc * (BC / WC) + i * 16 + 4 * fq;
            float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
            if (!mvalid || c >= a.Cout) {
                // nothing to store for this fragment
            } else if (!a.nchw) {
                // NHWC: cout % 8 == 0, so the 4 channels are all valid
                if (a.scale1) {
                    const float4 s = *reinterpret_cast<const float4*>(a.scale1 + c);
                    v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
                }
                if (a.shift1) {
                    const float4 s = *reinterpret_cast<const float4*>(a.shift1 + c);
                    v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], a.act1);
                const size_t off = ((size_t)m * a.Cout + c) * ES;
                if (a.residual) {
                    float rr[4];
                    load4<T>(a.residual + off, rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rr[r];
                }
                if (a.out_raw) store4<T>(a.out_raw + off, v);
                if (a.out_act) {
                    float u[4] = {v[0], v[1], v[2], v[3]};
                    if (a.scale2) {
                        const float4 s = *reinterpret_cast<const float4*>(a.scale2 + c);
                        const float4 t = *reinterpret_cast<const float4*>(a.shift2 + c);
                        u[0] = u[0] * s.x + t.x; u[1] = u[1] * s.y + t.y; u[2] = u[2] * s.z + t.z; u[3] = u[3] * s.w + t.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) u[r] = apply_act(u[r], a.act2);
                    store4<T>(a.out_act + off, u);
                }
            } else {
                // head: f32 NCHW, per channel 16 lanes write 16 consecutive pixels (64 B)
                float* out = reinterpret_cast<float*>(a.out_raw);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = c + r;
                    if (cc < a.Cout) {
                        const float s1 = a.scale1 ? a.scale1[cc] : 1.f, b1 = a.shift1 ? a.shift1[cc] : 0.f;
                        out[((size_t)nb * a.Cout + cc) * a.HoWo + np] = apply_act(v[r] * s1 + b1, a.act1);
                    }
                }
            }
        }
    };
    epilogue_col(std::integral_constant<int, 0>{});
    if constexpr (TP > 1) epilogue_col(std::integral_constant<int, 1>{});
    if constexpr (TP > 2) epilogue_col(std::integral_constant<int, 2>{});
    if constexpr (TP > 3) epilogue_col(std::integral_constant<int, 3>{});
    static_assert(TP <= 4, "epilogue handles up to 4 pixel tiles per wave");
}

template <typename T, int BP, int BC, int WP, int WC>
int launch_one(const ConvKArgs& a, hipStream_t st, const char** kname) {
    constexpr size_t lds = 2 * (size_t)(BP + BC) * 128;
    static char name[96];
    if (!name[0])
        snprintf(name, sizeof(name), "conv_igemm_big_kernel<%s, %d, %d, %d, %d>", sizeof(T) == 4 ? "float" : "__bf16",
                 BP, BC, WP, WC);
    if (kname) *kname = name;
    auto k = conv_igemm_big_kernel<T, BP, BC, WP, WC>;
    PPN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    hipLaunchKernelGGL(k, dim3(a.n_ctiles * a.n_ptiles), dim3(512), lds, st, a);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

template <typename T>
int launch_T(const ConvKArgs& a, BigTile t, hipStream_t st, const char** kname) {
    if (t.bc == 256) {
        if (t.bp == 256) return launch_one<T, 256, 256, 4, 2>(a, st, kname);
        if (t.bp == 192) return launch_one<T, 192, 256, 4, 2>(a, st, kname);
        return launch_one<T, 128, 256, 4, 2>(a, st, kname);
    }
    if (t.bp == 256) return launch_one<T, 256, 128, 4, 2>(a, st, kname);
    if (t.bp == 192) return launch_one<T, 192, 128, 4, 2>(a, st, kname);
    return launch_one<T, 128, 128, 4, 2>(a, st, kname);
}

}  // namespace

namespace ppnconv {

// Pick the pixel-tile height that wastes the fewest CU-rounds (one workgroup per CU, 256 CUs).
bool big_tile_for(int cout, long long m, BigTile* out) {
    if (cout < 128) return false;
    const int bc = cout >= 256 ? 256 : 128;
    const long long nct = (cout + bc - 1) / bc;
    double best = 1e30;
    int best_bp = 256;
    for (int bp : {256, 192, 128}) {
        const long long tiles = ((m + bp - 1) / bp) * nct;
        const long long rounds = (tiles + 255) / 256;
        // time ~ rounds * (work per tile + fixed cost per K loop pass); smaller tiles stage more bytes per FLOP
        const double eff = bp == 256 ? 1.0 : (bp == 192 ? 0.95 : 0.85);
        const double cost = (double)rounds * bp / eff;
        if (cost < best) { best = cost; best_bp = bp; }
    }
    out->bp = best_bp;
    out->bc = bc;
    return true;
}

int launch_big(const ConvKArgs& a, int dtype, BigTile t, hipStream_t st, const char** kname) {
    if (dtype == PPN_F32) return launch_T<float>(a, t, st, kname);
    return launch_T<__bf16>(a, t, st, kname);
}

}  // namespace ppnconv
