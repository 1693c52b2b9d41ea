// PPN training loss, forward + backward in one pass (reference main.py:125-216).
//
//   L_resp = mean_b sum (resp - delta)^2                         L_coor = mean_b sum weight ((x-tx_half)^2 + (y-ty_half)^2)
//   L_iou  = mean_b sum delta (conf - iou(pred, target))^2       L_size = mean_b sum weight ((sqrt(w+e)-sqrt(tw+e))^2 + (..h..)^2)
//   L_limb = mean_b sum weight_ij (e - te)^2                     iou on centre-format boxes with ReLU-clamped overlaps,
//                                                                inter / (a0 + a1 - inter + 1e-6)  (main.py:125-144)
// and, when grad_head != NULL, d(sum_i coeff_i L_i)/d(head) for the whole head tensor -- the gradient also flows
// through iou into x, y, w, h as in the reference (ious is not detached).
//
// The limb term is the memory-bound part: per sample it streams e (in the head), te and weight_ij (17.3 MB each)
// and writes 17.3 MB of gradient.  limb_kernel does that with 16-byte coalesced accesses on a fixed grid with a
// grid-stride loop; unary_kernel handles the 6K unary channels (one thread per (image, keypoint, cell));
// per-workgroup partial sums are combined by finalize_kernel in a fixed order, so the five losses are bitwise
// reproducible from run to run (no float atomics).
#include "common.h"
#include "conv_common.h"

namespace {

constexpr float kEps = 1e-6f;     // config.py:82 EPSILON

struct LossArgs {
    const float* head;
    const float *delta, *weight, *weight_ij, *tx_half, *ty_half, *tx, *ty, *tw, *th, *te;
    const unsigned char* limb_c;   // optional (the two fused training kernels): te | weight_ij in two bits per element, as
                                   // ppn_encode_targets_c writes them -- 1 byte read instead of 8 per limb element
    float* grad;          // may be NULL
    float* partial;       // workspace: [nblk_unary][4] then [nblk_limb]
    float* losses;        // [5]
    float coeff[5];
    const float* coeff_dev;   // ppn_loss_fwd_bwd_dev: c_i = coeff_dev[i] / coeff_div, read on the device
    float coeff_div;
    int B, K, E, S, H, W, C, inW, inH;
    int nblk_unary, nblk_limb;
    int Cg;               // channels of the tensor `grad` points at: C (head layout) or 6K (compact: unary channels only)
};

// coefficient i: by value (host call) or from device memory (a uniform scalar load)
__device__ __forceinline__ float coef(const LossArgs& a, int i) {
    return a.coeff_dev ? a.coeff_dev[i] / a.coeff_div : a.coeff[i];
}

// limb targets of element li: from the compact byte when there is one (weight_ij = 1 or 0.0005, te = 0 or 1: exactly the
// f32 values encode_limb_kernel / encode_te_kernel write), else from the f32 tensors
__device__ __forceinline__ void limb_targets(const LossArgs& a, size_t li, float* wj, float* te) {
    if (a.limb_c) {
        const unsigned v = a.limb_c[li];
        *wj = (v & 2u) ? 1.f : 0.0005f;
        *te = (v & 1u) ? 1.f : 0.f;
    } else {
        *wj = a.weight_ij[li];
        *te = a.te[li];
    }
}

// ... of V consecutive elements starting at li (V = 4: li % 4 == 0 and 16-byte aligned f32 tensors, checked on the host)
template <int V>
__device__ __forceinline__ void limb_targets_v(const LossArgs& a, size_t li, float (&wj)[V], float (&te)[V]) {
    if constexpr (V == 4) {
        if (a.limb_c) {
            const unsigned v = *reinterpret_cast<const unsigned*>(a.limb_c + li);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned bte = (v >> (8 * j)) & 0xffu;
                wj[j] = (bte & 2u) ? 1.f : 0.0005f;
                te[j] = (bte & 1u) ? 1.f : 0.f;
            }
        } else {
            const float4 w4 = *reinterpret_cast<const float4*>(a.weight_ij + li), t4 = *reinterpret_cast<const float4*>(a.te + li);
            wj[0] = w4.x; wj[1] = w4.y; wj[2] = w4.z; wj[3] = w4.w;
            te[0] = t4.x; te[1] = t4.y; te[2] = t4.z; te[3] = t4.w;
        }
    } else {
        limb_targets(a, li, &wj[0], &te[0]);
    }
}
template <int V>
__device__ __forceinline__ void load_v(const float* p, float (&o)[V]) {
    if constexpr (V == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
    } else {
        o[0] = p[0];
    }
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if (lane == 0) s_red[wid] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) t += s_red[w];
    return t;   // valid in thread 0
}

// d min(a,c)/da and d max(b,d)/db with PyTorch's tie rule (half the gradient on equality)
__device__ __forceinline__ float pick_lt(float a, float c) { return a < c ? 1.f : (a == c ? 0.5f : 0.f); }
__device__ __forceinline__ float pick_gt(float b, float d) { return b > d ? 1.f : (b == d ? 0.5f : 0.f); }

__global__ void __launch_bounds__(256) unary_kernel(LossArgs a) {
    __shared__ float s_red[4];
    const int HW = a.H * a.W;
    const long long n = (long long)a.B * a.K * HW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float l_resp = 0.f, l_iou = 0.f, l_coor = 0.f, l_size = 0.f;
    if (i < n) {
        const int cell = (int)(i % HW);
        const int k = (int)((i / HW) % a.K);
        const int b = (int)(i / ((long long)HW * a.K));
        const size_t hb = (size_t)b * a.C * HW + (size_t)k * HW + cell;       // resp channel of this (b,k,cell)
        const size_t KS = (size_t)a.K * HW;
        const float resp = a.head[hb], conf = a.head[hb + KS];
        const float x = a.head[hb + 2 * KS], y = a.head[hb + 3 * KS], w = a.head[hb + 4 * KS], h = a.head[hb + 5 * KS];
        const size_t t = (size_t)i;
        const float dl = a.delta[t], wt = a.weight[t];
        const float txh = a.tx_half[t], tyh = a.ty_half[t], tx = a.tx[t], ty = a.ty[t], tw = a.tw[t], th = a.th[t];
        const float invB = 1.0f / (float)a.B;
        const float gW = (float)(a.inW / a.W), gH = (float)(a.inH / a.H), inW = (float)a.inW, inH = (float)a.inH;
        const float X = (float)(cell % a.W), Y = (float)(cell / a.W);
        // restore_xy / restore_size (main.py:169-178)
        const float rx = (x + X) * gW, ry = (y + Y) * gH, rw = inW * w, rh = inH * h;
        const float rtx = (tx + X) * gW, rty = (ty + Y) * gH, rtw = inW * tw, rth = inH * th;
        // iou (main.py:125-144)
        const float a1 = rx + rw / 2, c1 = rtx + rtw / 2, b1 = rx - rw / 2, d1 = rtx - rtw / 2;
        const float a2 = ry + rh / 2, c2 = rty + rth / 2, b2 = ry - rh / 2, d2 = rty - rth / 2;
        const float wr = fminf(a1, c1) - fmaxf(b1, d1), hr = fminf(a2, c2) - fmaxf(b2, d2);
        const float wI = fmaxf(wr, 0.f), hI = fmaxf(hr, 0.f);
        const float I = wI * hI;
        const float U = rw * rh + rtw * rth - I + kEps;
        const float iou = I / U;
        // losses (main.py:199-208)
        const float dr = resp - dl, dc = conf - iou, dx = x - txh, dy = y - tyh;
        const float sw = sqrtf(w + kEps), sh = sqrtf(h + kEps);
        const float dsw = sw - sqrtf(tw + kEps), dsh = sh - sqrtf(th + kEps);
        l_resp = dr * dr;
        l_iou = dl * dc * dc;
        l_coor = wt * (dx * dx + dy * dy);
        l_size = wt * (dsw * dsw + dsh * dsh);
        if (a.grad) {
            const float c0 = coef(a, 0) * invB, c1c = coef(a, 1) * invB, c2c = coef(a, 2) * invB, c3c = coef(a, 3) * invB;
            const float g_iou = -2.f * dl * dc * c1c;                        // dL/d(iou)
            const float gI = g_iou * (U + I) / (U * U), gA0 = -g_iou * I / (U * U);
            const float g_wr = wr > 0.f ? gI * hI : 0.f, g_hr = hr > 0.f ? gI * wI : 0.f;
            const float da1 = pick_lt(a1, c1), db1 = pick_gt(b1, d1), da2 = pick_lt(a2, c2), db2 = pick_gt(b2, d2);
            const float g_rx = g_wr * (da1 - db1), g_ry = g_hr * (da2 - db2);
            const float g_rw = g_wr * 0.5f * (da1 + db1) + gA0 * rh, g_rh = g_hr * 0.5f * (da2 + db2) + gA0 * rw;
            const size_t gb = (size_t)b * a.Cg * HW + (size_t)k * HW + cell;
            a.grad[gb] = 2.f * dr * c0;
            a.grad[gb + KS] = 2.f * dl * dc * c1c;
            a.grad[gb + 2 * KS] = g_rx * gW + 2.f * wt * dx * c2c;
            a.grad[gb + 3 * KS] = g_ry * gH + 2.f * wt * dy * c2c;
            a.grad[gb + 4 * KS] = g_rw * inW + wt * dsw / sw * c3c;
            a.grad[gb + 5 * KS] = g_rh * inH + wt * dsh / sh * c3c;
        }
    }
    float* out = a.partial + (size_t)blockIdx.x * 4;
    float s;
    s = block_sum(l_resp, s_red); if (threadIdx.x == 0) out[0] = s;
    s = block_sum(l_iou, s_red);  if (threadIdx.x == 0) out[1] = s;
    s = block_sum(l_coor, s_red); if (threadIdx.x == 0) out[2] = s;
    s = block_sum(l_size, s_red); if (threadIdx.x == 0) out[3] = s;
}


// ---- second order: gradient AND Hessian-vector product of the losses w.r.t. the head -----------------------------------
// GradNorm's Lgrad.backward() (main.py:759) differentiates <v, dL_i/dW> again; in head space that needs
//     sdot_bar = c_i * dL_i/ds          and          s_bar = c_i * (d2L_i/ds2) sdot
// for a tangent sdot = s(1-s) * tz of the sigmoid outputs.  The Hessian-vector product is obtained by evaluating the
// SAME analytic gradient code on dual numbers (value, derivative along sdot): forward-over-reverse, no hand-derived
// second derivatives (the IoU term couples conf, x, y, w, h through min / max / ReLU).
struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual mk(float v, float d = 0.f) { return Dual{v, d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { return {a.v / b.v, (a.d * b.v - a.v * b.d) / (b.v * b.v)}; }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float b, Dual a) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return {a.v / b, a.d / b}; }
__device__ __forceinline__ Dual dsqrt(Dual a) { const float r = sqrtf(a.v); return {r, a.d * 0.5f / r}; }
// min / max with PyTorch's tie rule for the derivative (half on equality), second operand constant
__device__ __forceinline__ Dual dmin_c(Dual a, float c) { return {fminf(a.v, c), a.d * pick_lt(a.v, c)}; }
__device__ __forceinline__ Dual dmax_c(Dual b, float d) { return {fmaxf(b.v, d), b.d * pick_gt(b.v, d)}; }
__device__ __forceinline__ Dual drelu(Dual a) { return {fmaxf(a.v, 0.f), a.v > 0.f ? a.d : 0.f}; }

// g[0..5] = d(sum_i c_i L_i)/d(resp, conf, x, y, w, h) for one (image, keypoint, cell), on dual numbers.
// The arithmetic mirrors unary_kernel line by line (same operation order, same tie rules).
__device__ __forceinline__ void unary_grad_dual(Dual resp, Dual conf, Dual x, Dual y, Dual w, Dual h, float dl, float wt,
                                                float txh, float tyh, float tx, float ty, float tw, float th, float X,
                                                float Y, float gW, float gH, float inW, float inH, const float* c,
                                                Dual* g) {
    const Dual rx = (x + X) * gW, ry = (y + Y) * gH, rw = inW * w, rh = inH * h;
    const float rtx = (tx + X) * gW, rty = (ty + Y) * gH, rtw = inW * tw, rth = inH * th;
    const Dual a1 = rx + rw / 2.f, b1 = rx - rw / 2.f, a2 = ry + rh / 2.f, b2 = ry - rh / 2.f;
    const float c1 = rtx + rtw / 2, d1 = rtx - rtw / 2, c2 = rty + rth / 2, d2 = rty - rth / 2;
    const Dual wr = dmin_c(a1, c1) - dmax_c(b1, d1), hr = dmin_c(a2, c2) - dmax_c(b2, d2);
    const Dual wI = drelu(wr), hI = drelu(hr);
    const Dual I = wI * hI;
    const Dual U = rw * rh + (rtw * rth) - I + kEps;
    const Dual iou = I / U;
    const Dual dr = resp - dl, dc = conf - iou, dx = x - txh, dy = y - tyh;
    const Dual sw = dsqrt(w + kEps), sh = dsqrt(h + kEps);
    const Dual dsw = sw - sqrtf(tw + kEps), dsh = sh - sqrtf(th + kEps);
    const Dual g_iou = -2.f * dl * dc * c[1];
    const Dual gI = g_iou * (U + I) / (U * U), gA0 = -(g_iou * I / (U * U));
    // the selector functions are piecewise constant: no derivative through them
    const Dual g_wr = wr.v > 0.f ? gI * hI : mk(0.f), g_hr = hr.v > 0.f ? gI * wI : mk(0.f);
    const float da1 = pick_lt(a1.v, c1), db1 = pick_gt(b1.v, d1), da2 = pick_lt(a2.v, c2), db2 = pick_gt(b2.v, d2);
    const Dual g_rx = g_wr * (da1 - db1), g_ry = g_hr * (da2 - db2);
    const Dual g_rw = g_wr * (0.5f * (da1 + db1)) + gA0 * rh, g_rh = g_hr * (0.5f * (da2 + db2)) + gA0 * rw;
    g[0] = 2.f * dr * c[0];
    g[1] = 2.f * dl * dc * c[1];
    g[2] = g_rx * gW + 2.f * wt * dx * c[2];
    g[3] = g_ry * gH + 2.f * wt * dy * c[2];
    g[4] = g_rw * inW + wt * dsw / sw * c[3];
    g[5] = g_rh * inH + wt * dsh / sh * c[3];
}

// zbar = s_bar*sig' + sdot_bar*sig''*tz,  tzbar = sdot_bar*sig'   (sig' = s(1-s), sig'' = sig'(1-2s)); written in head layout
__device__ __forceinline__ void sigmoid_dual_adjoint(float s, float tz, float s_bar, float sdot_bar, float* zbar,
                                                     float* tzbar) {
    // the contractions are written out: this is inlined into kernels of different shapes (scalar and four-cell loops) whose
    // results the tests compare bit for bit, and the compiler's own choice of which product to fuse differs between them
    const float s1 = s * (1.f - s), s2 = s1 * __builtin_fmaf(-2.f, s, 1.f);
    *zbar = __builtin_fmaf(s_bar, s1, (sdot_bar * s2) * tz);
    *tzbar = sdot_bar * s1;
}

struct DualArgs {
    LossArgs a;
    const float* tz;      // tangent of the logits, [B][Cd][H*W]
    float* zbar;          // adjoint of the logits (primal stream), [B][Cd][H*W]
    float* tzbar;         // adjoint of the logit tangents, [B][Cd][H*W]
    int Cd;               // channels of the three dual tensors: C (head layout) or 6K (compact, unary passes)
};

__global__ void __launch_bounds__(256) unary_dual_kernel(DualArgs p) {
    const LossArgs& a = p.a;
    const int HW = a.H * a.W;
    const long long n = (long long)a.B * a.K * HW;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cell = (int)(i % HW);
    const int k = (int)((i / HW) % a.K);
    const int b = (int)(i / ((long long)HW * a.K));
    const size_t hb = (size_t)b * a.C * HW + (size_t)k * HW + cell;
    const size_t db = (size_t)b * p.Cd * HW + (size_t)k * HW + cell;
    const size_t KS = (size_t)a.K * HW;
    float s[6], tz[6];
    Dual in[6], g[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        s[j] = a.head[hb + j * KS];
        tz[j] = p.tz[db + j * KS];
        in[j] = mk(s[j], s[j] * (1.f - s[j]) * tz[j]);                    // sdot = sig'(z) * tz
    }
    const size_t t = (size_t)i;
    const float invB = 1.0f / (float)a.B;
    const float c[4] = {a.coeff[0] * invB, a.coeff[1] * invB, a.coeff[2] * invB, a.coeff[3] * invB};
    unary_grad_dual(in[0], in[1], in[2], in[3], in[4], in[5], a.delta[t], a.weight[t], a.tx_half[t], a.ty_half[t],
                    a.tx[t], a.ty[t], a.tw[t], a.th[t], (float)(cell % a.W), (float)(cell / a.W), (float)(a.inW / a.W),
                    (float)(a.inH / a.H), (float)a.inW, (float)a.inH, c, g);
#pragma unroll
    for (int j = 0; j < 6; ++j)
        sigmoid_dual_adjoint(s[j], tz[j], /*s_bar=*/g[j].d, /*sdot_bar=*/g[j].v, &p.zbar[db + j * KS],
                             &p.tzbar[db + j * KS]);
}

// limb loss: g = 2 c4/B * w_ij * (e - te),  H sdot = 2 c4/B * w_ij * sdot
__global__ void __launch_bounds__(256) limb_dual_kernel(DualArgs p) {
    const LossArgs& a = p.a;
    const int HW = a.H * a.W;
    const size_t per_img = (size_t)a.E * a.S * HW;
    const size_t head_img = (size_t)a.C * HW, head_off = (size_t)6 * a.K * HW;
    const size_t total = (size_t)a.B * per_img;
    const float g2 = 2.f * a.coeff[4] / (float)a.B;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / per_img, r = i - b * per_img;
        const size_t ho = b * head_img + head_off + r;
        const float s = a.head[ho], tz = p.tz[ho], wj = a.weight_ij[i];
        const float sdot = s * (1.f - s) * tz;
        sigmoid_dual_adjoint(s, tz, g2 * wj * sdot, g2 * wj * (s - a.te[i]), &p.zbar[ho], &p.tzbar[ho]);
    }
}

// The limb stream of the second-order pass, fused with the relayout the convolutions want: for the limb channels the same
// arithmetic as limb_dual_kernel, but zbar / tzbar go straight to NHWC `T` tensors [B][HW][Cpad] (what conv3's weight
// and input gradients read) through a 64 x 64 LDS transpose instead of to two f32 head-layout tensors that a second pass
// re-reads (2 x 541 MB written + read at batch 32) -- and the per-channel sums of zbar over a block's pixels (conv3.bias'
// second-order gradient) leave as partials [B][ceil(HW/64)][Cpad].  Channels outside [6K, C) are zero.
// V = 4 (H*W a multiple of 4, 16-byte aligned tensors): a thread takes FOUR consecutive cells of a channel -- 16-byte loads of
// the head / tangent planes (1 KB per wave and load instead of 256 B) and one 4-byte load of the compact targets.
template <typename T, int V>
__global__ void __launch_bounds__(256) limb_dual_nhwc_kernel(LossArgs a, const float* __restrict__ tzp, float c4, int Cpad,
                                                             T* __restrict__ zb, T* __restrict__ tzb,
                                                             float* __restrict__ zsum) {
    __shared__ float tz_t[64][65], z_t[64][65];
    const int HW = a.H * a.W, C6 = 6 * a.K;
    const int c0 = blockIdx.x * 64, p0 = blockIdx.y * 64, b = blockIdx.z;
    const int t = threadIdx.x;
    const size_t per_img = (size_t)a.E * a.S * HW;
    const float g2 = 2.f * c4 / (float)a.B;
    constexpr int PXT = 64 / V;                                          // threads along the cells
    constexpr int CPP = 256 / PXT;                                       // channels per pass
    const int px0 = (t % PXT) * V;
#pragma unroll 4
    for (int i = 0; i < 64 / CPP; ++i) {
        const int cl = t / PXT + CPP * i;
        const int c = c0 + cl, p = p0 + px0;
        float zv[V], tv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) zv[j] = tv[j] = 0.f;
        if (c >= C6 && c < a.C && p < HW) {                              // (V = 4: HW % 4 == 0, so the four cells exist together)
            const size_t ho = ((size_t)b * a.C + c) * HW + p;
            const size_t li = (size_t)b * per_img + (size_t)(c - C6) * HW + p;
            float s[V], tz[V], wj[V], te[V];
            load_v<V>(a.head + ho, s);
            load_v<V>(tzp + ho, tz);
            limb_targets_v<V>(a, li, wj, te);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float sdot = s[j] * (1.f - s[j]) * tz[j];
                sigmoid_dual_adjoint(s[j], tz[j], g2 * wj[j] * sdot, g2 * wj[j] * (s[j] - te[j]), &zv[j], &tv[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            z_t[px0 + j][cl] = zv[j];
            tz_t[px0 + j][cl] = tv[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = t + 256 * i;
        const int pr = item >> 3, ch = item & 7;
        const int p = p0 + pr;
        if (p < HW && c0 + ch * 8 < Cpad) {
            float v[8], w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = z_t[pr][ch * 8 + j]; w[j] = tz_t[pr][ch * 8 + j]; }
            const size_t o = (((size_t)b * HW + p) * Cpad + c0 + ch * 8) * sizeof(T);
            ppnconv::store8<T>(reinterpret_cast<char*>(zb) + o, v);
            ppnconv::store8<T>(reinterpret_cast<char*>(tzb) + o, w);
        }
    }
    if (t < 64 && c0 + t < Cpad) {                                       // fixed order: reproducible
        float acc = 0.f;
        for (int q = 0; q < 64; ++q) acc += z_t[q][t];
        zsum[((size_t)b * gridDim.y + blockIdx.y) * Cpad + c0 + t] = acc;
    }
}

// Limb loss forward + backward fused with what the head's backward does first (ppn_head_grad): for a 64-channel x 64-cell
// block the loss terms w_ij (e - te)^2 are summed (one partial per block), the gradient through the sigmoid
// dz = g * s(1 - s) goes straight to the NHWC `T` tensor conv3's weight / input gradients read (the unary channels' g
// comes from the compact tensor unary_kernel wrote), and the per-channel sums of dz over the block's cells leave as
// partials (conv3.bias' gradient).  Saves the f32 head-layout gradient (17 MB per image written, then re-read twice).
template <typename T, int V>
__global__ void __launch_bounds__(256) limb_loss_dz_kernel(LossArgs a, const float* __restrict__ ugrad, int Cpad,
                                                           T* __restrict__ dz, float* __restrict__ dbsum) {
    __shared__ float d_t[64][65];
    __shared__ float s_red[4];
    const int HW = a.H * a.W, C6 = 6 * a.K;
    const int c0 = blockIdx.x * 64, p0 = blockIdx.y * 64, b = blockIdx.z;
    const int t = threadIdx.x;
    const size_t per_img = (size_t)a.E * a.S * HW;
    const float g2 = 2.f * coef(a, 4) / (float)a.B;
    float lsum = 0.f;
    constexpr int PXT = 64 / V, CPP = 256 / PXT;                          // as limb_dual_nhwc_kernel
    const int px0 = (t % PXT) * V;
#pragma unroll 4
    for (int i = 0; i < 64 / CPP; ++i) {
        const int cl = t / PXT + CPP * i;
        const int c = c0 + cl, p = p0 + px0;
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = 0.f;
        if (c < a.C && p < HW) {
            float s[V], g[V];
            load_v<V>(a.head + ((size_t)b * a.C + c) * HW + p, s);
            if (c >= C6) {
                const size_t li = (size_t)b * per_img + (size_t)(c - C6) * HW + p;
                float wj[V], te[V];
                limb_targets_v<V>(a, li, wj, te);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = s[j] - te[j];
                    lsum += wj[j] * d * d;
                    g[j] = g2 * wj[j] * d;
                }
            } else {
                load_v<V>(ugrad + ((size_t)b * C6 + c) * HW + p, g);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = g[j] * (s[j] * (1.f - s[j]));
        }
#pragma unroll
        for (int j = 0; j < V; ++j) d_t[px0 + j][cl] = v[j];
    }
    const float bs = block_sum(lsum, s_red);                             // (contains the barrier the tile needs)
    if (t == 0)
        a.partial[(size_t)a.nblk_unary * 4 + ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = bs;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = t + 256 * i;
        const int pr = item >> 3, ch = item & 7;
        const int p = p0 + pr;
        if (p < HW && c0 + ch * 8 < Cpad) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = d_t[pr][ch * 8 + j];
            ppnconv::store8<T>(reinterpret_cast<char*>(dz) + (((size_t)b * HW + p) * Cpad + c0 + ch * 8) * sizeof(T), v);
        }
    }
    if (t < 64 && c0 + t < Cpad) {                                       // fixed order: reproducible
        float acc = 0.f;
        for (int q = 0; q < 64; ++q) acc += d_t[q][t];
        dbsum[((size_t)b * gridDim.y + blockIdx.y) * Cpad + c0 + t] = acc;
    }
}

template <int V>
__global__ void __launch_bounds__(256) limb_kernel(LossArgs a) {
    __shared__ float s_red[4];
    const int HW = a.H * a.W;
    const size_t per_img = (size_t)a.E * a.S * HW;                      // limb elements per image
    const size_t head_img = (size_t)a.C * HW, head_off = (size_t)6 * a.K * HW;
    const size_t nvec = per_img / V;                                   // V | per_img is checked on the host
    const float g = 2.f * coef(a, 4) / (float)a.B;
    float acc = 0.f;
    const size_t total = (size_t)a.B * nvec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / nvec, r = (i - b * nvec) * V;
        const float* e = a.head + b * head_img + head_off + r;
        const float* te = a.te + b * per_img + r;
        const float* wj = a.weight_ij + b * per_img + r;
        float ev[V], tv[V], wv[V], gv[V];
        if (V == 4) {
            const float4 e4 = *reinterpret_cast<const float4*>(e), t4 = *reinterpret_cast<const float4*>(te),
                         w4 = *reinterpret_cast<const float4*>(wj);
            ev[0] = e4.x; ev[1] = e4.y; ev[2] = e4.z; ev[3] = e4.w;
            tv[0] = t4.x; tv[1] = t4.y; tv[2] = t4.z; tv[3] = t4.w;
            wv[0] = w4.x; wv[1] = w4.y; wv[2] = w4.z; wv[3] = w4.w;
        } else {
            ev[0] = e[0]; tv[0] = te[0]; wv[0] = wj[0];
        }
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const float d = ev[u] - tv[u];
            acc += wv[u] * d * d;                                       // main.py:208
            gv[u] = g * wv[u] * d;
        }
        if (a.grad) {
            float* go = a.grad + b * head_img + head_off + r;
            if (V == 4) *reinterpret_cast<float4*>(go) = make_float4(gv[0], gv[1], gv[2], gv[3]);
            else go[0] = gv[0];
        }
    }
    const float s = block_sum(acc, s_red);
    if (threadIdx.x == 0) a.partial[(size_t)a.nblk_unary * 4 + blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) finalize_kernel(LossArgs a) {
    // fixed-order, double-precision combination of the per-workgroup partial sums; mean over the batch.  One workgroup per
    // loss.  (The ~50 us rocprofv3 shows for this launch in the training step are not its own work -- one workgroup walking
    // the five losses in turn, or eight loads in flight per thread, measure the same: it starts behind the limb kernel's
    // 280 MB of output still draining from the L2s.)
    __shared__ double s_acc[256];
    const int t = threadIdx.x, q = blockIdx.x;
    double v = 0.0;
    if (q < 4) {
        for (int i = t; i < a.nblk_unary; i += 256) v += (double)a.partial[(size_t)i * 4 + q];
    } else {
        for (int i = t; i < a.nblk_limb; i += 256) v += (double)a.partial[(size_t)a.nblk_unary * 4 + i];
    }
    s_acc[t] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) s_acc[t] += s_acc[t + o];
        __syncthreads();
    }
    if (t == 0) a.losses[q] = (float)(s_acc[0] / (double)a.B);
}

int fill(LossArgs& a, const ppn_loss_cfg* cfg, int batch) {
    if (!cfg) return ppn::fail(PPN_E_INVALID, "loss cfg is NULL");
    if (cfg->K < 1 || cfg->E < 0 || cfg->sH < 1 || cfg->sW < 1 || cfg->H < 1 || cfg->W < 1 || cfg->inH < cfg->H ||
        cfg->inW < cfg->W || batch < 1)
        return ppn::fail(PPN_E_INVALID, "bad loss geometry");
    a.B = batch; a.K = cfg->K; a.E = cfg->E; a.S = cfg->sH * cfg->sW; a.H = cfg->H; a.W = cfg->W;
    a.C = 6 * cfg->K + cfg->E * a.S; a.inW = cfg->inW; a.inH = cfg->inH;
    a.Cg = a.C;
    a.limb_c = nullptr;
    const long long n_unary = (long long)batch * a.K * a.H * a.W;
    a.nblk_unary = (int)((n_unary + 255) / 256);
    a.nblk_limb = 2048;
    return PPN_OK;
}

}  // namespace

extern "C" size_t ppn_loss_workspace_bytes(const ppn_loss_cfg* cfg, int32_t batch) {
    LossArgs a;
    if (fill(a, cfg, batch)) return 0;
    return ((size_t)a.nblk_unary * 4 + a.nblk_limb) * sizeof(float);
}

static int loss_fwd_bwd(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                        const float* weight, const float* weight_ij, const float* tx_half, const float* ty_half,
                        const float* tx, const float* ty, const float* tw, const float* th, const float* te,
                        const float* coeff, const float* coeff_dev, float coeff_div, float* losses, float* grad_head,
                        void* workspace, void* stream) {
    LossArgs a;
    if (int rc = fill(a, cfg, batch)) return rc;
    a.coeff_dev = coeff_dev; a.coeff_div = coeff_div;
    if (!head || !delta || !weight || !weight_ij || !tx_half || !ty_half || !tx || !ty || !tw || !th || !te || !losses ||
        !workspace)
        return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd: NULL pointer");
    if (grad_head && !coeff && !coeff_dev) return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd: coeff is required with grad_head");
    a.head = head; a.delta = delta; a.weight = weight; a.weight_ij = weight_ij; a.tx_half = tx_half; a.ty_half = ty_half;
    a.tx = tx; a.ty = ty; a.tw = tw; a.th = th; a.te = te;
    a.grad = grad_head; a.partial = static_cast<float*>(workspace); a.losses = losses;
    for (int i = 0; i < 5; ++i) a.coeff[i] = coeff ? coeff[i] : 0.f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(unary_kernel, dim3(a.nblk_unary), dim3(256), 0, st, a);
    PPN_LAUNCH_CHECK();
    const size_t per_img = (size_t)a.E * a.S * a.H * a.W;
    const bool vec = per_img % 4 == 0 && ((size_t)a.C * a.H * a.W) % 4 == 0 && ((size_t)6 * a.K * a.H * a.W) % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(head) % 16 == 0 && reinterpret_cast<uintptr_t>(te) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(weight_ij) % 16 == 0 &&
                     (!grad_head || reinterpret_cast<uintptr_t>(grad_head) % 16 == 0);
    if (a.E > 0) {
        if (vec) hipLaunchKernelGGL(limb_kernel<4>, dim3(a.nblk_limb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(limb_kernel<1>, dim3(a.nblk_limb), dim3(256), 0, st, a);
        PPN_LAUNCH_CHECK();
    } else {
        PPN_HIP_CHECK(hipMemsetAsync(a.partial + (size_t)a.nblk_unary * 4, 0, sizeof(float) * a.nblk_limb, st));
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(5), dim3(256), 0, st, a);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

extern "C" int ppn_loss_fwd_bwd(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                                const float* weight, const float* weight_ij, const float* tx_half, const float* ty_half,
                                const float* tx, const float* ty, const float* tw, const float* th, const float* te,
                                const float* coeff, float* losses, float* grad_head, void* workspace, void* stream) {
    return loss_fwd_bwd(cfg, head, batch, delta, weight, weight_ij, tx_half, ty_half, tx, ty, tw, th, te, coeff, nullptr,
                        1.f, losses, grad_head, workspace, stream);
}

// The same with the coefficients read ON THE DEVICE at kernel time: c_i = coeff_dev[i] / coeff_div (the task weights
// of the previous iteration's optimizerR.step(), main.py:668 `loss = sum(w_i * l_i) / 5`): the host needs no value, so
// it can enqueue the next iteration without waiting for the previous one.
extern "C" int ppn_loss_fwd_bwd_dev(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                                    const float* weight, const float* weight_ij, const float* tx_half,
                                    const float* ty_half, const float* tx, const float* ty, const float* tw,
                                    const float* th, const float* te, const float* coeff_dev, float coeff_div,
                                    float* losses, float* grad_head, void* workspace, void* stream) {
    if (!coeff_dev || !(coeff_div != 0.f)) return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd_dev: coeff_dev / coeff_div");
    return loss_fwd_bwd(cfg, head, batch, delta, weight, weight_ij, tx_half, ty_half, tx, ty, tw, th, te, nullptr,
                        coeff_dev, coeff_div, losses, grad_head, workspace, stream);
}

// Gradient of the four unary losses only (resp, iou, coor, size): d(sum_i coeff_i L_i)/d(head[:, 0:6K]) written
// into grad_head (head layout; the limb channels are NOT touched).  This is what a GradNorm probe pass needs for
// losses 0..3 (main.py:704-707) -- their gradients live in the first 6K of 7605 channels.
extern "C" int ppn_loss_unary_bwd(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                                  const float* weight, const float* tx_half, const float* ty_half, const float* tx,
                                  const float* ty, const float* tw, const float* th, const float* coeff4,
                                  float* grad_head, void* workspace, void* stream) {
    LossArgs a;
    if (int rc = fill(a, cfg, batch)) return rc;
    if (!head || !delta || !weight || !tx_half || !ty_half || !tx || !ty || !tw || !th || !coeff4 || !grad_head ||
        !workspace)
        return ppn::fail(PPN_E_INVALID, "ppn_loss_unary_bwd: NULL pointer");
    a.head = head; a.delta = delta; a.weight = weight; a.weight_ij = nullptr; a.tx_half = tx_half; a.ty_half = ty_half;
    a.tx = tx; a.ty = ty; a.tw = tw; a.th = th; a.te = nullptr;
    a.grad = grad_head; a.partial = static_cast<float*>(workspace); a.losses = nullptr;
    for (int i = 0; i < 4; ++i) a.coeff[i] = coeff4[i];
    a.coeff[4] = 0.f;
    a.coeff_dev = nullptr; a.coeff_div = 1.f;
    hipLaunchKernelGGL(unary_kernel, dim3(a.nblk_unary), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

// Second-order seeds of GradNorm's Lgrad.backward() (main.py:759) in head space, for ONE coefficient vector:
//   tz      f32 head layout: forward-mode tangent of the LOGITS (conv3 output before the sigmoid)
//   zbar    f32 head layout (out): adjoint of the logits            = s_bar*sig' + sdot_bar*sig''*tz
//   tzbar   f32 head layout (out): adjoint of the logit tangents    = sdot_bar*sig'
// with sdot = sig'*tz, sdot_bar = d(sum c_i L_i)/ds, s_bar = (d2(sum c_i L_i)/ds2) sdot.  unary_only: tz / zbar / tzbar are
// COMPACT tensors [B][6K][H*W] holding only the unary channels (coeff[4] must be 0); otherwise they have the head layout.
extern "C" int ppn_loss_dual(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch,
                             const float* delta, const float* weight, const float* weight_ij, const float* tx_half,
                             const float* ty_half, const float* tx, const float* ty, const float* tw, const float* th,
                             const float* te, const float* coeff, int32_t unary_only, float* zbar, float* tzbar,
                             void* stream) {
    DualArgs p;
    if (int rc = fill(p.a, cfg, batch)) return rc;
    if (!head || !tz || !delta || !weight || !tx_half || !ty_half || !tx || !ty || !tw || !th || !coeff || !zbar || !tzbar)
        return ppn::fail(PPN_E_INVALID, "ppn_loss_dual: NULL pointer");
    if (!unary_only && (!weight_ij || !te)) return ppn::fail(PPN_E_INVALID, "ppn_loss_dual: limb targets required");
    LossArgs& a = p.a;
    a.head = head; a.delta = delta; a.weight = weight; a.weight_ij = weight_ij; a.tx_half = tx_half; a.ty_half = ty_half;
    a.tx = tx; a.ty = ty; a.tw = tw; a.th = th; a.te = te; a.grad = nullptr; a.partial = nullptr; a.losses = nullptr;
    for (int i = 0; i < 5; ++i) a.coeff[i] = coeff[i];
    a.coeff_dev = nullptr; a.coeff_div = 1.f;
    p.tz = tz; p.zbar = zbar; p.tzbar = tzbar;
    p.Cd = unary_only ? 6 * a.K : a.C;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(unary_dual_kernel, dim3(a.nblk_unary), dim3(256), 0, st, p);
    PPN_LAUNCH_CHECK();
    if (!unary_only && a.E > 0) {
        hipLaunchKernelGGL(limb_dual_kernel, dim3(256 * 16), dim3(256), 0, st, p);
        PPN_LAUNCH_CHECK();
    }
    return PPN_OK;
}

// The limb loss's stream of ppn_loss_dual (coefficient vector (0,0,0,0,c4)) with NHWC outputs: zb, tzb `dtype`
// [B][H*W][cpad] (cpad a multiple of 64 >= 6K + E*S; channels outside the limb range are zero) and
// zsum f32 [B][ceil(H*W/64)][cpad], the per-block pixel sums of zbar.  Same arithmetic as ppn_loss_dual followed by
// ppn_nchw_to_nhwc, without the two f32 head-layout intermediates.
static int limb_dual_nhwc_impl(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch,
                              const float* weight_ij, const float* te, const unsigned char* limb_c, float c4, int32_t dtype,
                              int32_t cpad, void* zb, void* tzb, float* zsum, void* stream) {
    LossArgs a;
    if (int rc = fill(a, cfg, batch)) return rc;
    if (!head || !tz || (!limb_c && (!weight_ij || !te)) || !zb || !tzb || !zsum)
        return ppn::fail(PPN_E_INVALID, "ppn_loss_limb_dual_nhwc: NULL pointer");
    a.limb_c = limb_c;
    if (cpad % 64 || cpad < a.C) return ppn::fail(PPN_E_INVALID, "ppn_loss_limb_dual_nhwc: cpad %d for %d channels", cpad, a.C);
    if (dtype != PPN_F32 && dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "ppn_loss_limb_dual_nhwc: bad dtype %d", dtype);
    a.head = head; a.weight_ij = weight_ij; a.te = te;
    a.delta = a.weight = a.tx_half = a.ty_half = a.tx = a.ty = a.tw = a.th = nullptr;
    a.grad = nullptr; a.partial = nullptr; a.losses = nullptr; a.coeff_dev = nullptr; a.coeff_div = 1.f;
    for (int i = 0; i < 5; ++i) a.coeff[i] = 0.f;
    const int HW = a.H * a.W;
    const dim3 grid(cpad / 64, (HW + 63) / 64, batch);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = HW % 4 == 0 && al16(head) && al16(tz) &&
                     (limb_c ? (reinterpret_cast<uintptr_t>(limb_c) & 3) == 0 : (al16(weight_ij) && al16(te)));
    if (dtype == PPN_F32) {
        if (vec) hipLaunchKernelGGL((limb_dual_nhwc_kernel<float, 4>), grid, dim3(256), 0, st, a, tz, c4, cpad, (float*)zb, (float*)tzb, zsum);
        else hipLaunchKernelGGL((limb_dual_nhwc_kernel<float, 1>), grid, dim3(256), 0, st, a, tz, c4, cpad, (float*)zb, (float*)tzb, zsum);
    } else {
        if (vec) hipLaunchKernelGGL((limb_dual_nhwc_kernel<__bf16, 4>), grid, dim3(256), 0, st, a, tz, c4, cpad, (__bf16*)zb, (__bf16*)tzb, zsum);
        else hipLaunchKernelGGL((limb_dual_nhwc_kernel<__bf16, 1>), grid, dim3(256), 0, st, a, tz, c4, cpad, (__bf16*)zb, (__bf16*)tzb, zsum);
    }
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

extern "C" int ppn_loss_limb_dual_nhwc(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch,
                                       const float* weight_ij, const float* te, float c4, int32_t dtype, int32_t cpad,
                                       void* zb, void* tzb, float* zsum, void* stream) {
    return limb_dual_nhwc_impl(cfg, head, tz, batch, weight_ij, te, nullptr, c4, dtype, cpad, zb, tzb, zsum, stream);
}
extern "C" int ppn_loss_limb_dual_nhwc_c(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch,
                                         const uint8_t* limb_compact, float c4, int32_t dtype, int32_t cpad, void* zb,
                                         void* tzb, float* zsum, void* stream) {
    if (!limb_compact) return ppn::fail(PPN_E_INVALID, "ppn_loss_limb_dual_nhwc_c: limb_compact is NULL");
    return limb_dual_nhwc_impl(cfg, head, tz, batch, nullptr, nullptr, limb_compact, c4, dtype, cpad, zb, tzb, zsum, stream);
}

// ppn_loss_fwd_bwd_dev + ppn_head_grad in one: the five losses, and instead of d loss / d head in the head layout the
// gradient w.r.t. conv3's LOGITS in the layout its backward reads -- dz `dtype` [B][H*W][cpad] NHWC (channels >= C zero)
// -- plus dbsum f32 [B][ceil(H*W/64)][cpad], per-block cell sums of dz (summed over the first two axes: d loss / d
// conv3.bias).  grad_unary: f32 scratch [B][6K][H*W] (the unary channels' d loss / d head, compact).
extern "C" size_t ppn_loss_dz_workspace_bytes(const ppn_loss_cfg* cfg, int32_t batch, int32_t cpad) {
    LossArgs a;
    if (fill(a, cfg, batch) || cpad < 64) return 0;
    const size_t nb = (size_t)(cpad / 64) * ((a.H * a.W + 63) / 64) * batch;
    return ((size_t)a.nblk_unary * 4 + nb) * sizeof(float);
}

static int loss_fwd_bwd_dz_impl(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                               const float* weight, const float* weight_ij, const float* tx_half,
                               const float* ty_half, const float* tx, const float* ty, const float* tw,
                               const float* th, const float* te, const unsigned char* limb_c, const float* coeff_dev,
                               float coeff_div, float* losses, float* grad_unary, int32_t dtype, int32_t cpad, void* dz,
                               float* dbsum, void* workspace, void* stream) {
    LossArgs a;
    if (int rc = fill(a, cfg, batch)) return rc;
    if (!head || !delta || !weight || (!limb_c && (!weight_ij || !te)) || !tx_half || !ty_half || !tx || !ty || !tw || !th ||
        !coeff_dev || !losses || !grad_unary || !dz || !dbsum || !workspace)
        return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd_dz: NULL pointer");
    a.limb_c = limb_c;
    if (!(coeff_div != 0.f)) return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd_dz: coeff_div");
    if (cpad % 64 || cpad < a.C) return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd_dz: cpad %d for %d channels", cpad, a.C);
    if (dtype != PPN_F32 && dtype != PPN_BF16) return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd_dz: bad dtype %d", dtype);
    if (a.E < 1) return ppn::fail(PPN_E_UNSUPPORTED, "ppn_loss_fwd_bwd_dz: needs limb channels");
    a.head = head; a.delta = delta; a.weight = weight; a.weight_ij = weight_ij; a.tx_half = tx_half; a.ty_half = ty_half;
    a.tx = tx; a.ty = ty; a.tw = tw; a.th = th; a.te = te;
    a.grad = grad_unary; a.Cg = 6 * a.K;
    a.partial = static_cast<float*>(workspace); a.losses = losses;
    for (int i = 0; i < 5; ++i) a.coeff[i] = 0.f;
    a.coeff_dev = coeff_dev; a.coeff_div = coeff_div;
    const int HW = a.H * a.W;
    const dim3 grid(cpad / 64, (HW + 63) / 64, batch);
    a.nblk_limb = (int)(grid.x * grid.y * grid.z);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(unary_kernel, dim3(a.nblk_unary), dim3(256), 0, st, a);
    PPN_LAUNCH_CHECK();
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = HW % 4 == 0 && al16(head) && al16(grad_unary) &&
                     (limb_c ? (reinterpret_cast<uintptr_t>(limb_c) & 3) == 0 : (al16(weight_ij) && al16(te)));
    if (dtype == PPN_F32) {
        if (vec) hipLaunchKernelGGL((limb_loss_dz_kernel<float, 4>), grid, dim3(256), 0, st, a, grad_unary, cpad, (float*)dz, dbsum);
        else hipLaunchKernelGGL((limb_loss_dz_kernel<float, 1>), grid, dim3(256), 0, st, a, grad_unary, cpad, (float*)dz, dbsum);
    } else {
        if (vec) hipLaunchKernelGGL((limb_loss_dz_kernel<__bf16, 4>), grid, dim3(256), 0, st, a, grad_unary, cpad, (__bf16*)dz, dbsum);
        else hipLaunchKernelGGL((limb_loss_dz_kernel<__bf16, 1>), grid, dim3(256), 0, st, a, grad_unary, cpad, (__bf16*)dz, dbsum);
    }
    PPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_kernel, dim3(5), dim3(256), 0, st, a);
    PPN_LAUNCH_CHECK();
    return PPN_OK;
}

extern "C" int ppn_loss_fwd_bwd_dz(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                                   const float* weight, const float* weight_ij, const float* tx_half,
                                   const float* ty_half, const float* tx, const float* ty, const float* tw,
                                   const float* th, const float* te, const float* coeff_dev, float coeff_div,
                                   float* losses, float* grad_unary, int32_t dtype, int32_t cpad, void* dz,
                                   float* dbsum, void* workspace, void* stream) {
    return loss_fwd_bwd_dz_impl(cfg, head, batch, delta, weight, weight_ij, tx_half, ty_half, tx, ty, tw, th, te, nullptr,
                                coeff_dev, coeff_div, losses, grad_unary, dtype, cpad, dz, dbsum, workspace, stream);
}

extern "C" int ppn_loss_fwd_bwd_dz_c(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                                     const float* weight, const uint8_t* limb_compact, const float* tx_half,
                                     const float* ty_half, const float* tx, const float* ty, const float* tw,
                                     const float* th, const float* coeff_dev, float coeff_div, float* losses,
                                     float* grad_unary, int32_t dtype, int32_t cpad, void* dz, float* dbsum,
                                     void* workspace, void* stream) {
    if (!limb_compact) return ppn::fail(PPN_E_INVALID, "ppn_loss_fwd_bwd_dz_c: limb_compact is NULL");
    return loss_fwd_bwd_dz_impl(cfg, head, batch, delta, weight, nullptr, tx_half, ty_half, tx, ty, tw, th, nullptr,
                                limb_compact, coeff_dev, coeff_div, losses, grad_unary, dtype, cpad, dz, dbsum, workspace,
                                stream);
}
