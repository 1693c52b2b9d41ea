// Training-target encoder on the device (dataset.py:96-185): person lists -> delta, tx, ty, tw, th, tx_half, ty_half,
// weight [B,K,H,W] and te, weight_ij [B,E,sH,sW,H,W].  The reference builds these per sample on the host and ships
// 2 x 17.3 MB per sample over PCIe every step (main.py:649-661); here the batch arrives as a few hundred bytes per
// person and the two dense limb tensors are written once at HBM speed (2 x 553 MB at batch 32).
//
//   encode_unary_kernel  one workgroup per image: people are applied IN ORDER (a later person overwrites a cell an
//                        earlier one claimed, as the Python loop does), then weight / tx_half / ty_half
//   encode_limb_kernel   every (edge, sh, sw, h, w): te = 0 and weight_ij from the delta maps (dataset.py:155-173)
//   encode_te_kernel     one lane per (image, edge): te[ei, j - i + s/2, i] = 1 for every labeled limb (idempotent)
// All arithmetic is the reference's f32 arithmetic, operation for operation: outputs are bit-exact.
#include "common.h"

namespace {

struct EncArgs {
    int K, E, sH, sW, H, W, inH, inW, B, pmax;
    int esrc[PPN_MAX_EDGES], edst[PPN_MAX_EDGES];
    const float* people;     // [B][pmax][5 + 2*(K-1)]: cx, cy, w, h, size, then (x, y) of keypoints 1..K-1
    const int* visible;      // [B][pmax]: bit k-1 set = keypoint k labeled
    const int* count;        // [B]
    float *delta, *weight, *weight_ij, *tx_half, *ty_half, *tx, *ty, *tw, *th, *te;
    unsigned char* limb_c;   // optional: te and weight_ij in two bits per element (bit 0: te = 1, bit 1: weight_ij = 1)
};

__device__ __forceinline__ bool labeled(const EncArgs& a, const float* P, int vis, int k) {
    return k == 0 ? (P[2] > 0.f && P[3] > 0.f) : ((vis >> (k - 1)) & 1);      // dataset.py:113-117
}
__device__ __forceinline__ void point(const float* P, int k, float* x, float* y) {
    if (k == 0) { *x = P[0]; *y = P[1]; }                                     // the instance "keypoint" is the bbox centre
    else { *x = P[5 + 2 * (k - 1)]; *y = P[6 + 2 * (k - 1)]; }
}

__global__ void __launch_bounds__(256) encode_unary_kernel(EncArgs a) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int HW = a.H * a.W, n = a.K * HW;
    const size_t base = (size_t)b * n;
    for (int i = t; i < n; i += 256) {
        a.delta[base + i] = 0.f; a.tx[base + i] = 0.f; a.ty[base + i] = 0.f; a.tw[base + i] = 0.f; a.th[base + i] = 0.f;
    }
    __syncthreads();
    const int stride = 5 + 2 * (a.K - 1);
    const float gridW = (float)(a.inW / a.W), gridH = (float)(a.inH / a.H);
    const int np = min(a.count[b], a.pmax);
    if (t < a.K) {
        const int k = t;
        for (int p = 0; p < np; ++p) {                                        // dataset.py:108-134, in order
            const float* P = a.people + ((size_t)b * a.pmax + p) * stride;
            const int vis = a.visible[b * a.pmax + p];
            if (!labeled(a, P, vis, k)) continue;
            float x, y;
            point(P, k, &x, &y);
            const float gx = x / gridW, gy = y / gridH;
            const int ix = (int)gx, iy = (int)gy;                             // int(): truncation toward zero
            const float sw_ = k == 0 ? P[2] : P[4], sh_ = k == 0 ? P[3] : P[4];
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                const size_t o = base + (size_t)k * HW + iy * a.W + ix;
                a.delta[o] = 1.f;
                a.tx[o] = gx - (float)ix;
                a.ty[o] = gy - (float)iy;
                a.tw[o] = sw_ / (float)a.inW;
                a.th[o] = sh_ / (float)a.inH;
            }
        }
    }
    __syncthreads();
    for (int i = t; i < n; i += 256) {                                        // dataset.py:172-185
        const float d = a.delta[base + i];
        const float lo = d < 0.5f ? 1.f : 0.f;
        a.weight[base + i] = fminf(d + lo * 0.0005f, 1.f);
        a.tx_half[base + i] = a.tx[base + i] + lo * 0.5f;
        a.ty_half[base + i] = a.ty[base + i] + lo * 0.5f;
    }
}

// one thread per V consecutive w of (b, ei, sh, sw, h): 16-byte stores when W is a multiple of 4
template <int V>
__global__ void __launch_bounds__(256) encode_limb_kernel(EncArgs a) {
    const int HW = a.H * a.W, W4 = a.W / V;
    const long long total = (long long)a.B * a.E * a.sH * a.sW * a.H * W4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        long long r = i;
        const int w0 = (int)(r % W4) * V; r /= W4;
        const int h = (int)(r % a.H); r /= a.H;
        const int sw = (int)(r % a.sW); r /= a.sW;
        const int sh = (int)(r % a.sH); r /= a.sH;
        const int ei = (int)(r % a.E);
        const int b = (int)(r / a.E);
        const float* ds = a.delta + ((size_t)b * a.K + a.esrc[ei]) * HW;
        const float* dt = a.delta + ((size_t)b * a.K + a.edst[ei]) * HW;
        const int hh = h + sh - a.sH / 2;
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int w = w0 + j, ww = w + sw - a.sW / 2;
            float m;
            if (ds[h * a.W + w] != 0.f) m = 1.f;
            else m = (hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) ? dt[hh * a.W + ww] : 0.f;
            v[j] = fminf(m + (m < 0.5f ? 0.0005f : 0.f), 1.f);
        }
        const size_t o = ((((size_t)b * a.E + ei) * a.sH + sh) * a.sW + sw) * HW + h * a.W + w0;
        if constexpr (V == 4) {
            *reinterpret_cast<float4*>(a.weight_ij + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(a.te + o) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            a.weight_ij[o] = v[0];
            a.te[o] = 0.f;
        }
        if (a.limb_c) {                    // delta is 0 / 1 here, so weight_ij is 1 or 0.0005: one bit
#pragma unroll
            for (int j = 0; j < V; ++j) a.limb_c[o + j] = v[j] == 1.f ? 2 : 0;
        }
    }
}

__global__ void __launch_bounds__(64) encode_te_kernel(EncArgs a) {
    const int b = blockIdx.x, ei = threadIdx.x;
    if (ei >= a.E) return;
    const int HW = a.H * a.W;
    const int stride = 5 + 2 * (a.K - 1);
    const float gridW = (float)(a.inW / a.W), gridH = (float)(a.inH / a.H);
    const int np = min(a.count[b], a.pmax);
    const int s = a.esrc[ei], t = a.edst[ei];
    for (int p = 0; p < np; ++p) {                                            // dataset.py:136-152
        const float* P = a.people + ((size_t)b * a.pmax + p) * stride;
        const int vis = a.visible[b * a.pmax + p];
        if (!labeled(a, P, vis, s) || !labeled(a, P, vis, t)) continue;
        float sx, sy, tx_, ty_;
        point(P, s, &sx, &sy);
        point(P, t, &tx_, &ty_);
        const int iy = (int)(sy / gridH), ix = (int)(sx / gridW);
        const int jy = (int)(ty_ / gridH) - iy + a.sH / 2, jx = (int)(tx_ / gridW) - ix + a.sW / 2;
        if (iy < 0 || ix < 0 || iy >= a.H || ix >= a.W) continue;
        if (jy < 0 || jx < 0 || jy >= a.sH || jx >= a.sW) continue;
        const size_t o = ((((size_t)b * a.E + ei) * a.sH + jy) * a.sW + jx) * HW + iy * a.W + ix;
        a.te[o] = 1.f;
        if (a.limb_c) a.limb_c[o] |= 1;    // this (image, edge) thread owns every byte it touches
    }
}

}  // namespace

static int encode_targets_impl(const ppn_loss_cfg* cfg, const int32_t* edges, const float* people,
                               const int32_t* visible, const int32_t* count, int32_t batch, int32_t pmax,
                               float* delta, float* weight, float* weight_ij, float* tx_half, float* ty_half,
                               float* tx, float* ty, float* tw, float* th, float* te, unsigned char* limb_c, void* stream) {
    if (!cfg || !edges || !people || !visible || !count || !delta || !weight || !weight_ij || !tx_half || !ty_half ||
        !tx || !ty || !tw || !th || !te)
        return ppn::fail(PPN_E_INVALID, "ppn_encode_targets: NULL pointer");
    if (batch < 1 || pmax < 1 || cfg->K < 1 || cfg->K > PPN_MAX_KP || cfg->E < 0 || cfg->E > PPN_MAX_EDGES ||
        cfg->H < 1 || cfg->W < 1 || cfg->sH < 1 || cfg->sW < 1 || cfg->inH % cfg->H || cfg->inW % cfg->W)
        return ppn::fail(PPN_E_INVALID, "ppn_encode_targets: bad geometry (insize must be a multiple of outsize)");
    EncArgs a{};
    a.K = cfg->K; a.E = cfg->E; a.sH = cfg->sH; a.sW = cfg->sW; a.H = cfg->H; a.W = cfg->W; a.inH = cfg->inH;
    a.inW = cfg->inW; a.B = batch; a.pmax = pmax;
    for (int e = 0; e < cfg->E; ++e) {
        a.esrc[e] = edges[2 * e]; a.edst[e] = edges[2 * e + 1];
        if (a.esrc[e] < 0 || a.esrc[e] >= cfg->K || a.edst[e] < 0 || a.edst[e] >= cfg->K)
            return ppn::fail(PPN_E_INVALID, "ppn_encode_targets: edge %d out of range", e);
    }
    a.people = people; a.visible = visible; a.count = count;
    a.delta = delta; a.weight = weight; a.weight_ij = weight_ij; a.tx_half = tx_half; a.ty_half = ty_half;
    a.tx = tx; a.ty = ty; a.tw = tw; a.th = th; a.te = te;
    a.limb_c = limb_c;
    hipStream_t st = (hipStream_t)stream;
    encode_unary_kernel<<<batch, 256, 0, st>>>(a);
    PPN_LAUNCH_CHECK();
    if (cfg->E > 0) {
        const bool vec = cfg->W % 4 == 0 && reinterpret_cast<uintptr_t>(weight_ij) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(te) % 16 == 0;
        const long long total = (long long)batch * cfg->E * cfg->sH * cfg->sW * cfg->H * (cfg->W / (vec ? 4 : 1));
        long long blocks = (total + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (vec) encode_limb_kernel<4><<<(int)blocks, 256, 0, st>>>(a);
        else encode_limb_kernel<1><<<(int)blocks, 256, 0, st>>>(a);
        PPN_LAUNCH_CHECK();
        encode_te_kernel<<<batch, 64, 0, st>>>(a);
        PPN_LAUNCH_CHECK();
    }
    return PPN_OK;
}

extern "C" int ppn_encode_targets(const ppn_loss_cfg* cfg, const int32_t* edges, const float* people,
                                  const int32_t* visible, const int32_t* count, int32_t batch, int32_t pmax,
                                  float* delta, float* weight, float* weight_ij, float* tx_half, float* ty_half,
                                  float* tx, float* ty, float* tw, float* th, float* te, void* stream) {
    return encode_targets_impl(cfg, edges, people, visible, count, batch, pmax, delta, weight, weight_ij, tx_half, ty_half,
                               tx, ty, tw, th, te, nullptr, stream);
}

extern "C" int ppn_encode_targets_c(const ppn_loss_cfg* cfg, const int32_t* edges, const float* people,
                                    const int32_t* visible, const int32_t* count, int32_t batch, int32_t pmax,
                                    float* delta, float* weight, float* weight_ij, float* tx_half, float* ty_half,
                                    float* tx, float* ty, float* tw, float* th, float* te, uint8_t* limb_compact,
                                    void* stream) {
    if (!limb_compact) return ppn::fail(PPN_E_INVALID, "ppn_encode_targets_c: limb_compact is NULL");
    return encode_targets_impl(cfg, edges, people, visible, count, batch, pmax, delta, weight, weight_ij, tx_half, ty_half,
                               tx, ty, tw, th, te, limb_compact, stream);
}
