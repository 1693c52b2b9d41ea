// A recorded forward pass: the ordered list of fused launches of PoseProposalNet.forward
// (model.py:104-136), replayed on a stream.  ppn_plan_run_timed brackets every launch with HIP events on
// the launch stream so bench.py can report per-kernel durations next to rocprofv3.
#include <string>
#include <vector>

#include "common.h"

namespace ppn {
int conv_launch(const ppn_conv_desc* d, hipStream_t st, const char** kname);
int split_x3_launch(const float* src, long long pixels, int channels, void* dst, hipStream_t st);
int block64_launch(const ppn_block_desc* d, hipStream_t st, const char** kname);
int stem_launch(int dtype, int src_is_u8, const void* src, int batch, int h, int w, const float* weight,
                const float* scale, const float* shift, const float* mean, const float* stdv, void* out,
                hipStream_t st);
int stem01_launch(int dtype, int src_is_u8, const void* src, int batch, int h, int w, const float* w0,
                  const float* s0, const float* b0, const float* mean, const float* stdv, const float* w1,
                  const float* s1, const float* b1, void* out, hipStream_t st);
int stem012_launch(int dtype, int src_is_u8, const void* src, int batch, int h, int w, const float* w0, const float* s0,
                   const float* b0, const float* mean, const float* stdv, const float* w1, const float* s1,
                   const float* b1, const float* w2, const float* s2, const float* b2, const float* s3, const float* b3,
                   void* out_raw, void* out_act, hipStream_t st);
}  // namespace ppn

struct ppn_plan {
    struct Op {
        int kind;  // 0 conv, 1 stem, 2 memset, 3 stem01 (layer0 + layer1), 4 stem012 (layer0 + layer1 + layer2, bf16),
                   // 5 f32 -> PPN_F16X3 pair conversion (src = f32 tensor, ms_ptr = destination, ms_bytes = pixels, batch = channels)
        const float *w1 = nullptr, *scale1 = nullptr, *shift1 = nullptr;
        const float *w2 = nullptr, *scale2 = nullptr, *shift2 = nullptr, *scale3 = nullptr, *shift3 = nullptr;
        void* out2 = nullptr;
        void* ms_ptr = nullptr;
        size_t ms_bytes = 0;
        ppn_conv_desc conv;
        ppn_block_desc block;      // kind 6: a whole 64-channel BasicBlock (csrc/block64.hip)
        int dtype, src_is_u8, batch, h, w;
        const void* src;
        const float *weight, *scale, *shift;
        float mean[3], stdv[3];
        void* out;
        std::string kname;
    };
    std::vector<Op> ops;
    std::vector<hipEvent_t> events;
    // the launch sequence captured once per (input pointer, stream) into a hipGraph: one graph launch per forward
    // instead of ~35 kernel launches (smaller inter-kernel gaps on the GPU, no per-launch host work)
    hipGraphExec_t graph_exec = nullptr;
    const void* graph_src = nullptr;
    hipStream_t graph_stream = nullptr;
    int direct_runs = 0;
    int captures = 0;          // how many times the launch sequence was captured + instantiated
    bool graph_off = false;
};

static const void* plan_src(const ppn_plan* p) {
    for (auto& op : p->ops)
        if (op.kind == 1 || op.kind == 3 || op.kind == 4) return op.src;
    return nullptr;
}

// Zero fill as a kernel, not hipMemsetAsync: inside a captured hipGraph the memset NODE was observed to run out of order
// with respect to the kernel nodes around it (the arg-max keys were cleared after the head conv had written them --
// wrong people lists on replay, bench.py's `verified` check), a kernel node keeps its place in the chain.
__global__ void __launch_bounds__(256) zero_fill_kernel(uint4* p, size_t n16, unsigned char* tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

static int run_op(ppn_plan::Op& op, hipStream_t st) {
    if (op.kind == 2) {
        if (op.kname.empty()) op.kname = "zero_fill_kernel";
        const size_t n16 = op.ms_bytes / 16;
        const int ntail = (int)(op.ms_bytes - n16 * 16);
        const size_t want = (n16 + 255) / 256;
        const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
        hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint4*>(op.ms_ptr), n16,
                           static_cast<unsigned char*>(op.ms_ptr) + n16 * 16, ntail);
        PPN_LAUNCH_CHECK();
        return PPN_OK;
    }
    if (op.kind == 5) {
        if (op.kname.empty()) op.kname = "split_x3_kernel";
        return ppn::split_x3_launch(static_cast<const float*>(op.src), (long long)op.ms_bytes, op.batch, op.ms_ptr, st);
    }
    if (op.kind == 6) {
        const char* kn = nullptr;
        int rc = ppn::block64_launch(&op.block, st, &kn);
        if (rc == PPN_OK && kn && op.kname.empty()) op.kname = kn;
        return rc;
    }
    if (op.kind == 0) {
        const char* kn = nullptr;
        int rc = ppn::conv_launch(&op.conv, st, &kn);
        if (rc == PPN_OK && kn && op.kname.empty()) op.kname = kn;
        return rc;
    }
    if (op.kind == 4) {
        if (op.kname.empty()) op.kname = "stem012_kernel";
        return ppn::stem012_launch(op.dtype, op.src_is_u8, op.src, op.batch, op.h, op.w, op.weight, op.scale, op.shift, op.mean,
                                   op.stdv, op.w1, op.scale1, op.shift1, op.w2, op.scale2, op.shift2, op.scale3,
                                   op.shift3, op.out, op.out2, st);
    }
    if (op.kind == 3) {
        if (op.kname.empty()) op.kname = op.dtype == PPN_F32 ? "stem01_kernel<float>" : "stem01_kernel<__bf16>";
        return ppn::stem01_launch(op.dtype, op.src_is_u8, op.src, op.batch, op.h, op.w, op.weight, op.scale, op.shift,
                                  op.mean, op.stdv, op.w1, op.scale1, op.shift1, op.out, st);
    }
    if (op.kname.empty()) op.kname = op.dtype == PPN_F32 ? "stem7x7_kernel<float>" : "stem7x7_kernel<__bf16>";
    return ppn::stem_launch(op.dtype, op.src_is_u8, op.src, op.batch, op.h, op.w, op.weight, op.scale, op.shift,
                            op.mean, op.stdv, op.out, st);
}

extern "C" int ppn_plan_create(ppn_plan** out) {
    if (!out) return ppn::fail(PPN_E_INVALID, "ppn_plan_create: NULL out");
    *out = new (std::nothrow) ppn_plan();
    return *out ? PPN_OK : ppn::fail(PPN_E_HIP, "out of host memory");
}

extern "C" int ppn_plan_add_conv(ppn_plan* p, const ppn_conv_desc* d) {
    if (!p || !d) return ppn::fail(PPN_E_INVALID, "ppn_plan_add_conv: NULL argument");
    ppn_plan::Op op{};
    op.kind = 0;
    op.conv = *d;
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_block(ppn_plan* p, const ppn_block_desc* d) {
    if (!p || !d) return ppn::fail(PPN_E_INVALID, "ppn_plan_add_block: NULL argument");
    ppn_plan::Op op{};
    op.kind = 6;
    op.block = *d;
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_split(ppn_plan* p, const float* src, int64_t pixels, int32_t channels, void* dst) {
    if (!p || !src || !dst || pixels < 1 || channels < 8 || channels % 8 != 0)
        return ppn::fail(PPN_E_INVALID, "ppn_plan_add_split: NULL argument or channels %% 8 != 0");
    ppn_plan::Op op{};
    op.kind = 5;
    op.src = src;
    op.ms_ptr = dst;
    op.ms_bytes = (size_t)pixels;
    op.batch = channels;
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_memset(ppn_plan* p, void* ptr, size_t bytes) {
    if (!p || !ptr) return ppn::fail(PPN_E_INVALID, "ppn_plan_add_memset: NULL argument");
    // the fill runs as a kernel with 16-byte stores (never a memset node: see zero_fill_kernel), so the destination
    // must be 16-byte aligned -- every torch allocation is
    if ((reinterpret_cast<size_t>(ptr) & 15) != 0)
        return ppn::fail(PPN_E_INVALID, "ppn_plan_add_memset: destination must be 16-byte aligned");
    ppn_plan::Op op{};
    op.kind = 2;
    op.ms_ptr = ptr;
    op.ms_bytes = bytes;
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_stem(ppn_plan* p, int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch,
                                 int32_t h, int32_t w, const float* weight, const float* scale, const float* shift,
                                 const float* mean, const float* std_, void* out) {
    if (!p) return ppn::fail(PPN_E_INVALID, "ppn_plan_add_stem: NULL plan");
    ppn_plan::Op op{};
    op.kind = 1;
    op.dtype = dtype; op.src_is_u8 = src_is_u8; op.src = src; op.batch = batch; op.h = h; op.w = w;
    op.weight = weight; op.scale = scale; op.shift = shift; op.out = out;
    for (int i = 0; i < 3; ++i) { op.mean[i] = mean ? mean[i] : 0.f; op.stdv[i] = std_ ? std_[i] : 1.f; }
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_stem01(ppn_plan* p, int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch,
                                   int32_t h, int32_t w, const float* w0, const float* scale0, const float* shift0,
                                   const float* mean, const float* std_, const float* w1, const float* scale1,
                                   const float* shift1, void* out) {
    if (!p) return ppn::fail(PPN_E_INVALID, "ppn_plan_add_stem01: NULL plan");
    ppn_plan::Op op{};
    op.kind = 3;
    op.dtype = dtype; op.src_is_u8 = src_is_u8; op.src = src; op.batch = batch; op.h = h; op.w = w;
    op.weight = w0; op.scale = scale0; op.shift = shift0; op.w1 = w1; op.scale1 = scale1; op.shift1 = shift1;
    op.out = out;
    for (int i = 0; i < 3; ++i) { op.mean[i] = mean ? mean[i] : 0.f; op.stdv[i] = std_ ? std_[i] : 1.f; }
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_stem012_dt(ppn_plan* p, int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h,
                                    int32_t w, const float* w0, const float* scale0, const float* shift0,
                                    const float* mean, const float* std_, const float* w1, const float* scale1,
                                    const float* shift1, const float* w2, const float* scale2, const float* shift2,
                                    const float* scale3, const float* shift3, void* out_raw, void* out_act) {
    if (!p) return ppn::fail(PPN_E_INVALID, "ppn_plan_add_stem012: NULL plan");
    if ((dtype & 0xff) != PPN_BF16 && (dtype & 0xff) != PPN_F16)
        return ppn::fail(PPN_E_INVALID, "ppn_plan_add_stem012: dtype must be PPN_BF16, PPN_F16 or PPN_STEM_IO(PPN_F16, PPN_BF16)");
    ppn_plan::Op op{};
    op.kind = 4;
    op.dtype = dtype; op.src_is_u8 = src_is_u8; op.src = src; op.batch = batch; op.h = h; op.w = w;
    op.weight = w0; op.scale = scale0; op.shift = shift0; op.w1 = w1; op.scale1 = scale1; op.shift1 = shift1;
    op.w2 = w2; op.scale2 = scale2; op.shift2 = shift2; op.scale3 = scale3; op.shift3 = shift3;
    op.out = out_raw; op.out2 = out_act;
    for (int i = 0; i < 3; ++i) { op.mean[i] = mean ? mean[i] : 0.f; op.stdv[i] = std_ ? std_[i] : 1.f; }
    p->ops.push_back(op);
    return PPN_OK;
}

extern "C" int ppn_plan_add_stem012(ppn_plan* p, int32_t src_is_u8, const void* src, int32_t batch, int32_t h,
                                    int32_t w, const float* w0, const float* scale0, const float* shift0,
                                    const float* mean, const float* std_, const float* w1, const float* scale1,
                                    const float* shift1, const float* w2, const float* scale2, const float* shift2,
                                    const float* scale3, const float* shift3, void* out_raw, void* out_act) {
    return ppn_plan_add_stem012_dt(p, PPN_BF16, src_is_u8, src, batch, h, w, w0, scale0, shift0, mean, std_, w1, scale1,
                                   shift1, w2, scale2, shift2, scale3, shift3, out_raw, out_act);
}

extern "C" int ppn_plan_set_input(ppn_plan* p, const void* src) {
    if (!p || !src) return ppn::fail(PPN_E_INVALID, "ppn_plan_set_input: NULL argument");
    for (auto& op : p->ops)
        if (op.kind == 1 || op.kind == 3 || op.kind == 4) { op.src = src; return PPN_OK; }
    return ppn::fail(PPN_E_INVALID, "ppn_plan_set_input: plan has no input layer");
}

extern "C" int ppn_plan_run(ppn_plan* p, void* stream) {
    if (!p) return ppn::fail(PPN_E_INVALID, "ppn_plan_run: NULL plan");
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const bool graphs = !(getenv("PPN_PLAN_GRAPH") && atoi(getenv("PPN_PLAN_GRAPH")) == 0);
    // the first runs go launch by launch (they set kernel attributes, which must not happen inside a capture)
    // (the legacy default stream cannot be captured: launches on it stay direct, without giving up on graphs)
    if (graphs && !p->graph_off && st != nullptr && p->direct_runs >= 2) {
        const void* src = plan_src(p);
        if (!p->graph_exec || p->graph_src != src || p->graph_stream != st) {
            if (p->graph_exec) {
                // the old executable graph may still have a launch queued on its stream: retire it only once that
                // stream has drained (rare path: the callers above keep one input buffer and one stream per plan)
                (void)hipStreamSynchronize(p->graph_stream);
                (void)hipGraphExecDestroy(p->graph_exec);
                p->graph_exec = nullptr;
            }
            hipGraph_t g = nullptr;
            bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                int rc = PPN_OK;
                for (auto& op : p->ops)
                    if ((rc = run_op(op, st))) break;
                ok = hipStreamEndCapture(st, &g) == hipSuccess && rc == PPN_OK && g;
            }
            if (ok) ok = hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
            if (!ok) {                                   // keep launching directly (same kernels, same results)
                (void)hipGetLastError();
                p->graph_exec = nullptr;
                p->graph_off = true;
            } else {
                p->graph_src = src;
                p->graph_stream = st;
                ++p->captures;
            }
        }
        if (p->graph_exec) {
            PPN_HIP_CHECK(hipGraphLaunch(p->graph_exec, st));
            return PPN_OK;
        }
    }
    for (auto& op : p->ops)
        if (int rc = run_op(op, st)) return rc;
    ++p->direct_runs;
    return PPN_OK;
}

extern "C" int ppn_plan_run_timed(ppn_plan* p, void* stream, float* ms, int32_t n_ms, int32_t repeats) {
    if (!p || !ms) return ppn::fail(PPN_E_INVALID, "ppn_plan_run_timed: NULL argument");
    const size_t n = p->ops.size();
    if ((size_t)n_ms < n) return ppn::fail(PPN_E_INVALID, "ms buffer too small (%d < %zu)", n_ms, n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    while (p->events.size() < n + 1) {
        hipEvent_t e;
        PPN_HIP_CHECK(hipEventCreate(&e));
        p->events.push_back(e);
    }
    PPN_HIP_CHECK(hipEventRecord(p->events[0], st));
    if (repeats < 1) repeats = 1;
    for (size_t i = 0; i < n; ++i) {
        for (int r = 0; r < repeats; ++r)
            if (int rc = run_op(p->ops[i], st)) return rc;
        PPN_HIP_CHECK(hipEventRecord(p->events[i + 1], st));
    }
    PPN_HIP_CHECK(hipEventSynchronize(p->events[n]));
    for (size_t i = 0; i < n; ++i) {
        PPN_HIP_CHECK(hipEventElapsedTime(&ms[i], p->events[i], p->events[i + 1]));
        ms[i] /= (float)repeats;
    }
    return PPN_OK;
}

extern "C" int ppn_plan_graph_captures(const ppn_plan* p) { return p ? p->captures : 0; }

extern "C" int ppn_plan_size(const ppn_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" const char* ppn_plan_kernel_name(const ppn_plan* p, int32_t i) {
    if (!p || i < 0 || (size_t)i >= p->ops.size()) return "";
    return p->ops[i].kname.c_str();
}

extern "C" int ppn_plan_destroy(ppn_plan* p) {
    if (!p) return PPN_OK;
    for (auto e : p->events) (void)hipEventDestroy(e);
    if (p->graph_exec) {
        (void)hipStreamSynchronize(p->graph_stream);
        (void)hipGraphExecDestroy(p->graph_exec);
    }
    delete p;
    return PPN_OK;
}
