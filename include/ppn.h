/*
 * libppn -- MI355X (gfx950) native Pose Proposal Network hot path, C ABI.
 *
 * The reference (noirmist/Pytorch_Pose_Proposal_Network) is 100 % Python and has no FFI;
 * its call surface for this path is
 *     PoseProposalNet.forward            model.py:104-136   (conv/BN/act stack)
 *     rt_test.inference                  rt_test.py:87-147  (normalise, forward, slice, decode)
 *     datatest.get_humans_by_feature     datatest.py:74-132 (decode + root NMS + limb parse)
 *     datatest.non_maximum_suppression   datatest.py:134-160
 *     PPNLoss.forward + loss.backward()  main.py:125-216, 664-683 (fused loss forward + gradient)
 *     train()                            main.py:623-777   (train-mode BN, conv gradients, GradNorm task weights,
 *                                                           Adam; see "Training building blocks" below)
 *     KeypointsDataset target encoding   dataset.py:96-185
 * The entry points below are what a ctypes binding for those functions binds
 * (INTEGRATION.md shows the stub).  Conventions:
 *   - every function returns 0 on success, a negative PPN_E_* code on error;
 *     ppn_last_error() returns a thread-local message.  No C++ exception crosses the ABI.
 *   - all tensor arguments are BORROWED raw device pointers (PyTorch-ROCm owns the memory);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every launch is asynchronous on that stream; nothing here synchronises or allocates,
 *     except ppn_plan_create/ppn_plan_destroy (host memory, optional hipGraph).
 *   - not thread-safe per handle (the reference is single-threaded).
 */
#ifndef PPN_H
#define PPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPN_OK 0
#define PPN_E_INVALID (-1)   /* bad argument / unsupported shape */
#define PPN_E_HIP (-2)       /* a HIP runtime call failed */
#define PPN_E_UNSUPPORTED (-3)

#define PPN_MAX_EDGES 32
#define PPN_MAX_KP 32

const char* ppn_last_error(void);
int ppn_version(void);

/* ------------------------------------------------------------------------------------------
 * Decode: grid-cell decode + root-box NMS + greedy limb parse on the DEVICE head tensor.
 * Replaces datatest.py:74-132 (get_humans_by_feature) incl. the head slicing and
 * delta = resp*conf of rt_test.py:106-130; bit-exact on every index it returns.
 * ---------------------------------------------------------------------------------------- */
typedef struct ppn_decode_cfg {
    int32_t K, E;                 /* keypoints (18), edges (17)                config.py:64-65   */
    int32_t sH, sW;               /* local limb window (21,21)                 rt_test.py:58     */
    int32_t H, W;                 /* output grid (24,24)                       datatest.py:54    */
    int32_t inH, inW;             /* network input size (384,384)              datatest.py:53    */
    float det_thr;                /* 0.15  rt_test.py:133                                        */
    float nms_thr;                /* 0.3   datatest.py:94                                        */
    int32_t min_kp;               /* 1     datatest.py:74                                        */
    int32_t max_humans;           /* rows per image in the output buffers (<= H*W)               */
    int32_t edge_src[PPN_MAX_EDGES];   /* EDGES[e][0]                          config.py:65      */
    int32_t edge_dst[PPN_MAX_EDGES];   /* EDGES[e][1]                                            */
    int32_t edge_order[PPN_MAX_EDGES]; /* edges parent-before-child (tree form of DIRECTED_GRAPHS,
                                          config.py:67-80)                                      */
} ppn_decode_cfg;

/* Bytes of device scratch ppn_decode needs for `batch` images (limb arg-max map + the root-NMS survivor lists that
 * the first workgroup per image of the arg-max launch leaves for the parse kernel). */
size_t ppn_decode_workspace_bytes(const ppn_decode_cfg* cfg, int32_t batch);

/*
 * head       f32 [batch, 6K + E*sH*sW, H, W]  sigmoid outputs, NCHW contiguous (model.py:136)
 * workspace  >= ppn_decode_workspace_bytes()
 * out_count  i32 [batch]                          humans kept per image (before max_humans clamp)
 * out_kp_cell  i32 [batch, max_humans, K]         row-major cell h*W+w of each accepted keypoint, -1 absent
 * out_limb_arg i32 [batch, max_humans, E]         arg-max index sh*sW+sw of each evaluated limb, -1 otherwise
 * out_bbox   f32 [batch, max_humans, K, 4]        (ymin,xmin,ymax,xmax)  datatest.py:80-86
 * out_score  f32 [batch, max_humans, K]           delta = resp*conf of accepted keypoints
 * Humans are ordered by descending root score (datatest.py:103,139); equal scores by ascending cell.
 */
int ppn_decode(const ppn_decode_cfg* cfg, const float* head, int32_t batch, void* workspace,
               int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg, float* out_bbox,
               float* out_score, void* stream);

/* ppn_decode for the fused head conv: `unary` f32 [batch, 6K, H, W] and `keys` u64 [batch, E, H, W] as written
 * by ppn_conv2d_fused with argmax_keys set.  Same outputs, bit-identical to ppn_decode on the full head. */
int ppn_decode_fused(const ppn_decode_cfg* cfg, const float* unary, const uint64_t* keys, int32_t batch,
                     int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg, float* out_bbox,
                     float* out_score, void* stream);

/* ppn_decode_fused with a device workspace (>= ppn_decode_fused_workspace_bytes, 16-byte aligned): the O(n^2) pairwise-IoU
 * bit matrix of every image's root candidates (datatest.py:134-160 inside get_humans_by_feature, :87-95) is then computed
 * by a launch of 8 workgroups per image in front of the per-image parse workgroup, which only runs the greedy order on it
 * (dense heads: ~490 of 576 cells are candidates on the benchmark's synthetic checkpoint).  Images with fewer than 128
 * candidates keep the single-workgroup form.  Same results, bit for bit. */
size_t ppn_decode_fused_workspace_bytes(const ppn_decode_cfg* cfg, int32_t batch);
int ppn_decode_fused_ws(const ppn_decode_cfg* cfg, const float* unary, const uint64_t* keys, int32_t batch, void* workspace,
                        int32_t* out_count, int32_t* out_kp_cell, int32_t* out_limb_arg, float* out_bbox,
                        float* out_score, void* stream);

/* First half of ppn_decode on its own (the HBM-bound kernel): dense first-index arg-max over the
 * sH*sW limb window for every (image, edge, cell).  out_arg i32 [batch, E, H, W]. */
int ppn_limb_argmax(const ppn_decode_cfg* cfg, const float* head, int32_t batch, int32_t* out_arg,
                    void* stream);

/*
 * datatest.py:134-160 non_maximum_suppression.  bbox f32 [n,4] (ymin,xmin,ymax,xmax) on device,
 * score f32 [n] or NULL, limit <= 0 means None.  out_sel i32 [n] receives the selected indices in the
 * reference's order (descending score when score is given), out_count i32 [1].  n <= 1024.
 */
int ppn_nms(const float* bbox, const float* score, int32_t n, float thresh, int32_t limit, int32_t* out_sel,
            int32_t* out_count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training loss (forward + backward), replaces PPNLoss.forward + autograd through it (main.py:125-216, 664-683).
 * ---------------------------------------------------------------------------------------- */
typedef struct ppn_loss_cfg {
    int32_t K, E, sH, sW, H, W, inH, inW;   /* as in ppn_decode_cfg */
} ppn_loss_cfg;

size_t ppn_loss_workspace_bytes(const ppn_loss_cfg* cfg, int32_t batch);

/*
 * head        f32 [B, 6K+E*sH*sW, H, W]   feature_map (sigmoid outputs)
 * delta, weight, tx_half, ty_half, tx, ty, tw, th   f32 [B,K,H,W];  weight_ij, te  f32 [B,E,sH,sW,H,W]
 *             (the CustomBatch fields of dataset.py:233-248, argument order of main.py:180)
 * coeff       HOST pointer to 5 floats c_i (read at call time); may be NULL when grad_head is NULL
 * losses      f32 [5] device: loss_resp, loss_iou, loss_coor, loss_size, loss_limb (each summed over the
 *             non-batch dims, then mean over the batch -- main.py:199-214); bitwise reproducible
 * grad_head   f32 like head or NULL: d(sum_i c_i L_i)/d(head); with c_i = w_i/5 this is the backward of
 *             main.py:668-683, with a one-hot c the per-loss gradient the GradNorm step needs (main.py:704-708)
 */
int ppn_loss_fwd_bwd(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                     const float* weight, const float* weight_ij, const float* tx_half, const float* ty_half,
                     const float* tx, const float* ty, const float* tw, const float* th, const float* te,
                     const float* coeff, float* losses, float* grad_head, void* workspace, void* stream);

/*
 * ppn_loss_fwd_bwd with the five coefficients read on the DEVICE when the kernels run: c_i = coeff_dev[i] / coeff_div
 * (main.py:668 `loss = sum_i w_i l_i / 5` with the task weights optimizerR.step() left on the device, main.py:761-777).
 * The host passes no value, so it can enqueue an iteration before the previous one has finished.
 */
int ppn_loss_fwd_bwd_dev(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                         const float* weight, const float* weight_ij, const float* tx_half, const float* ty_half,
                         const float* tx, const float* ty, const float* tw, const float* th, const float* te,
                         const float* coeff_dev, float coeff_div, float* losses, float* grad_head, void* workspace,
                         void* stream);

/*
 * ppn_loss_fwd_bwd_dev and ppn_head_grad in one pass over the head: the five losses and, instead of d loss / d head in
 * the head layout, the gradient w.r.t. conv3's LOGITS in the layout the convolutions' backward reads:
 *   dz     `dtype` (PPN_F32 / PPN_BF16) [B][H*W][cpad] NHWC = g * s(1-s), channels >= 6K + E*sH*sW zero; cpad % 64 == 0
 *   dbsum  f32 [B][ceil(H*W/64)][cpad]: per-block cell sums of dz; summed over the first two axes = d loss / d conv3.bias
 *   grad_unary  f32 scratch [B][6K][H*W] (the unary channels' d loss / d head)
 * workspace >= ppn_loss_dz_workspace_bytes(cfg, batch, cpad).  Saves the f32 head-layout gradient (17 MB per image
 * written once and read twice).  loss_limb is summed per 64 x 64 block and then in a fixed order (reproducible; it
 * differs from ppn_loss_fwd_bwd's value in the last bits).
 */
size_t ppn_loss_dz_workspace_bytes(const ppn_loss_cfg* cfg, int32_t batch, int32_t cpad);
int ppn_loss_fwd_bwd_dz(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                        const float* weight, const float* weight_ij, const float* tx_half, const float* ty_half,
                        const float* tx, const float* ty, const float* tw, const float* th, const float* te,
                        const float* coeff_dev, float coeff_div, float* losses, float* grad_unary, int32_t dtype,
                        int32_t cpad, void* dz, float* dbsum, void* workspace, void* stream);
/* ... with the limb targets as ppn_encode_targets_c's compact bytes instead of weight_ij / te (bit-identical results) */
int ppn_loss_fwd_bwd_dz_c(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                          const float* weight, const uint8_t* limb_compact, const float* tx_half, const float* ty_half,
                          const float* tx, const float* ty, const float* tw, const float* th, const float* coeff_dev,
                          float coeff_div, float* losses, float* grad_unary, int32_t dtype, int32_t cpad, void* dz,
                          float* dbsum, void* workspace, void* stream);

/*
 * Training-target encoder (dataset.py:96-185) on the device: person lists -> the ten target tensors of
 * ppn_loss_fwd_bwd, bit-exact with the host encoder.  Replaces the per-sample host encoding + 2 x 17.3 MB/sample
 * H2D of main.py:649-661.
 *   edges    HOST i32 [E][2]  (src, dst keypoint of every limb, config.py:44-62)
 *   people   f32 [B][pmax][5 + 2*(K-1)] device: cx, cy, w, h (instance box, centre format), size (side of a part box,
 *            pixels), then x, y of keypoints 1..K-1 in input pixels
 *   visible  i32 [B][pmax] device: bit k-1 set = keypoint k is labeled;  count i32 [B]: people per image (<= pmax)
 * People are applied in list order (a later person overwrites a grid cell an earlier one claimed).
 */
int ppn_encode_targets(const ppn_loss_cfg* cfg, const int32_t* edges, const float* people, const int32_t* visible,
                       const int32_t* count, int32_t batch, int32_t pmax, float* delta, float* weight,
                       float* weight_ij, float* tx_half, float* ty_half, float* tx, float* ty, float* tw, float* th,
                       float* te, void* stream);
/* ... and, besides the ten f32 tensors, limb_compact u8 [B][E][sH][sW][H][W]: te and weight_ij in two bits per element
 * (bit 0: te = 1, bit 1: weight_ij = 1; weight_ij is 1 or 0.0005 for delta maps of 0 / 1, which is what this encoder
 * writes).  ppn_loss_fwd_bwd_dz_c and ppn_loss_limb_dual_nhwc_c -- the two kernels of a training iteration that stream the
 * limb targets -- read it instead of the two f32 tensors (1 byte instead of 8 per element: 1.1 GB less HBM traffic per
 * kernel at batch 32), with bit-identical results. */
int ppn_encode_targets_c(const ppn_loss_cfg* cfg, const int32_t* edges, const float* people, const int32_t* visible,
                         const int32_t* count, int32_t batch, int32_t pmax, float* delta, float* weight,
                         float* weight_ij, float* tx_half, float* ty_half, float* tx, float* ty, float* tw, float* th,
                         float* te, uint8_t* limb_compact, void* stream);

/*
 * Gradient of the four unary losses only: d(sum_{i<4} coeff4_i L_i)/d(head[:, 0:6K]) into grad_head (head layout;
 * the limb channels are not touched, no loss values are produced).  The GradNorm probe passes for losses 0..3
 * (main.py:704-707) need nothing else: their gradients live in the first 6K of the 7605 channels.  coeff4 is a
 * HOST pointer to 4 floats.
 */
int ppn_loss_unary_bwd(const ppn_loss_cfg* cfg, const float* head, int32_t batch, const float* delta,
                       const float* weight, const float* tx_half, const float* ty_half, const float* tx,
                       const float* ty, const float* tw, const float* th, const float* coeff4, float* grad_head,
                       void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution stack: one fused launch per convolution of PoseProposalNet.forward
 * (model.py:104-136, drn.py:42-57,77-97,192-202).
 *   acc  = conv(src, weight)                        implicit GEMM on MFMA, NHWC activations
 *   v    = act1(acc * scale1 + shift1)              folded conv-bias / eval-mode BN + ReLU/LReLU/sigmoid
 *   v   += residual                                 (optional)
 *   out_raw = v                                     (optional)
 *   out_act = act2(v * scale2 + shift2)             (optional) pre-activation of the consumer block
 * ---------------------------------------------------------------------------------------- */
enum { PPN_ACT_NONE = 0, PPN_ACT_RELU = 1, PPN_ACT_LRELU = 2, PPN_ACT_SIGMOID = 3 };
/* PPN_F16: IEEE half operands (v_mfma_f32_16x16x32_f16: the bf16 MFMA rate, 3 more mantissa bits), f32 accumulation,
 * f32 head -- inference only (the conv stack, ppn_plan_add_stem012, ppn_pack_weight); the training entry points take
 * PPN_F32 / PPN_BF16. */
/* PPN_F16X3 (round 4): split-precision inference mode for the convolutions with cin % 64 == 0.  Activations are NHWC
 * half PAIRS [pixel][hi(C) | lo'(C)] (hi = half(v), lo' = half((v - hi) * 2^11)), weights are packed by
 * ppn_pack_weight_x3 as three half copies per 64-channel slab, and the K loop accumulates a_hi w_hi + a_hi w_lo + a_lo w_hi
 * in f32 on v_mfma_f32_16x16x32_f16: 22-bit operands at a third of the f16 MFMA rate.  Meets the 1e-4 head tolerance
 * (tests/test_x3_gpu.py); the layers with cin < 64 (stem, the first block's stride-2 convs) run as PPN_F32 and their
 * outputs are converted by ppn_split_f16x3. */
enum { PPN_F32 = 0, PPN_BF16 = 1, PPN_F16 = 2, PPN_F16X3 = 3 };

typedef struct ppn_conv_desc {
    int32_t dtype;               /* PPN_F32 (exact-f32 MFMA, parity mode), PPN_BF16 or PPN_F16 (16-bit MFMA, f32 accumulate) */
    int32_t batch, in_h, in_w, cin;
    int32_t out_h, out_w, cout;
    int32_t ksize, stride, dilation, pad;
    int32_t k_total;             /* padded GEMM depth of the packed weight rows (multiple of the K step) */
    int32_t cout_pad;            /* rows in the packed weight matrix (multiple of the channel tile)  */
    int32_t act1, act2;
    int32_t out_nchw_f32;        /* 1: out_raw is f32 NCHW [B,cout,H,W] (the head, model.py:136)     */
    const void* src;             /* NHWC [B,in_h,in_w,cin] dtype                                      */
    const void* weight;          /* packed [cout_pad][k_total] dtype, depth order per ppn_conv_tiling  */
    const float* scale1;         /* [cout] or NULL (=1)                                               */
    const float* shift1;         /* [cout] or NULL (=0)                                               */
    const void* residual;        /* NHWC [B,out_h,out_w,cout] dtype or NULL                           */
    void* out_raw;               /* NHWC dtype (or NCHW f32) or NULL                                  */
    const float* scale2;         /* [cout] or NULL                                                    */
    const float* shift2;         /* [cout] or NULL                                                    */
    void* out_act;               /* NHWC dtype or NULL                                                */
    const void* zero_page;       /* >= 256 B of zeros on device (padding source)                      */
    /* Fused decode front end for the head conv (out_nchw_f32 = 1): when argmax_keys != NULL the epilogue also
     * writes the first `unary_channels` (= 6K) sigmoid outputs to `unary_out` f32 [B,unary_channels,H,W] and folds
     * every limb channel into `argmax_keys` u64 [B,E,H,W] (key = value bits << 32 | ~window index; the caller
     * zeroes the buffer before the launch, see ppn_plan_add_memset).  out_raw may then be NULL: the 17.5 MB/image
     * head tensor is never written (rt_test.py:109-120 copies it to the host instead).  ppn_decode_fused consumes
     * the two buffers. */
    /* Fused 1x1 projection shortcut (BasicBlock.downsample, drn.py:53-54): when src2 != NULL the GEMM depth is
     * extended by cin2 values gathered from src2 NHWC [B,in2_h,in2_w,cin2] at (oy*stride2, ox*stride2); the packed
     * weight rows carry [main conv | shortcut 1x1 weights pre-multiplied by the shortcut BN scale] and shift1
     * carries the shortcut BN shift, so  out = conv(src) + bn_ds(conv1x1(src2))  costs no extra launch and no
     * residual tensor.  k_total = k_main + cin2, both multiples of the K step; scale1 must be NULL. */
    const void* src2;
    int32_t in2_h, in2_w, cin2, stride2;
    float* unary_out;
    uint64_t* argmax_keys;
    int32_t unary_channels;      /* 6K = 108                                                          */
    int32_t limb_window;         /* sH*sW = 441                                                       */
    /* Output-pixel range of this launch, in the flattened [batch*out_h*out_w] order: pixels m_begin .. m_begin +
     * m_count - 1 are computed, the rest of the output tensors is left untouched.  m_count = 0 (default): the whole
     * tensor (under tile policy 2 ppn_conv2d_fused then cuts the range itself where ppn_conv_split says so; a plan
     * lists the two ranges as separate entries so that each launch is timed and named). */
    int32_t m_begin, m_count;
    /* Edge-aligned limb part of the head conv (the fast fused path): limb_edge_pad = 448 (the only size built: windows of
     * 385..448 values, e.g. 21 x 21 = 441) says that `weight`, `scale1` and `shift1` hold ONLY the limb channels, edge e's
     * window in rows [e * limb_edge_pad, e * limb_edge_pad + limb_window) and padding after it (cout = E * limb_window,
     * cout_pad = E * limb_edge_pad).  One workgroup then owns a whole window of 128 cells, reduces its arg-max on the
     * accumulators and STORES argmax_keys u64 [B,E,H,W] (same key format; no atomics, the buffer need not be zeroed).
     * out_raw, unary_out must be NULL and unary_channels 0: the 6K unary channels are an ordinary NCHW launch of their
     * own (cout = 6K, out_raw = the compact unary tensor).  0 (default): the chunked epilogue with atomicMax keys. */
    int32_t limb_edge_pad;
    /* PPN_CONV_* bits, 0 by default.  Carried by the descriptor (and therefore by each plan entry), never process-wide:
     * PPN_CONV_NO_FILTER_BANK routes a 64 -> 64 3x3 stride-1 convolution of the 16-bit modes through the generic
     * implicit-GEMM kernel instead of the register-resident filter-bank kernel (csrc/conv64.hip), whose workgroups own a
     * CU's whole LDS and register file -- the choice of a plan that shares the GPU with other lanes' plans
     * (rt.MultiLaneInference).  Results are bit-identical either way. */
    int32_t flags;
    /* Prefetch hint (round 5; large-tile kernel only, ignored elsewhere): `prefetch_bytes` bytes at `prefetch` -- the packed
     * weights of the NEXT launch of a plan -- are touched once (one dword per 128-byte line, shared out over the workgroups)
     * before this launch's epilogue, which puts them into the Infinity Cache.  In the forward pass a layer's weights were
     * last read one whole pass (> 1 GB of traffic) ago, so its first round of workgroups otherwise fetches every K step's
     * weight slab from HBM in lockstep (profiles/r05/cold_operands.txt: 80 vs 100 us on the 24 x 24 512-wide layers).
     * Results do not depend on it.  NULL / 0: none. */
    const void* prefetch;
    int64_t prefetch_bytes;
    /* Train-mode BatchNorm statistics from this launch's epilogue (round 5; large-tile kernel, 16-bit single-output NHWC launches
     * without a residual -- elsewhere the launch runs as usual and reports 0 tiles).  Each pixel tile folds, per output channel,
     * two sums over ITS pixels of the values it stores (v, after rounding to the tensor's type) and writes them as f64 pairs
     * stats_partial[(tile * cout + c) * 2 + {0, 1}] -- the layout ppn_bn_train_fwd / ppn_bn_train_bwd fold their own reduction
     * pass from, so that pass is skipped (ppn_bn_desc.stats_blocks):
     *   stats_mode 1 (the convolution that FEEDS a BatchNorm, drn.py:47-63):   { sum v, sum v^2 }
     *   stats_mode 2 (the input-gradient convolution whose result is dy of a BatchNorm + activation over stats_x):
     *                { sum g, sum g * xhat },  g = v * act'(x * gamma * rstd + beta - mean * gamma * rstd),  xhat = (x - mean) * rstd
     * *stats_tiles (HOST int) receives the number of pixel tiles written, 0 if this launch does not produce statistics. */
    void* stats_partial;
    int32_t stats_mode, stats_act;
    const void* stats_x;                       /* mode 2: the BatchNorm's input, NHWC like this launch's output */
    const float *stats_gamma, *stats_beta, *stats_mean, *stats_rstd;
    int32_t* stats_tiles;
} ppn_conv_desc;
#define PPN_CONV_NO_FILTER_BANK 1
/* PPN_CONV_SHARED_GPU: this launch runs beside other streams' launches (rt.MultiLaneInference): the tile chooser then
 * leaves out the tiles that only shorten a LONE launch by filling every CU with smaller, less efficient workgroups (the
 * 144 x 256 tile of the 24 x 24 layers: -11 % in sequence, +19 % CU-time, -3.5 % images/s with three lanes). */
#define PPN_CONV_SHARED_GPU 2
/* PPN_CONV_OUT_BF16: a PPN_F16 launch (large-tile kernel, NHWC) stores out_raw / out_act as bf16 -- the last launch of an
 * IEEE-half PREFIX in front of a bf16 trunk.  The bf16 mode runs the stem and layer3-4 (6.9 % of DRN-D-22's FLOPs) in half
 * since round 4: same kernels, same rate, and the rounding noise injected there is what every later layer amplifies. */
#define PPN_CONV_OUT_BF16 4
/* PPN_CONV_X3_PLAIN_OUT: a PPN_F16X3 launch stores out_raw / out_act as plain IEEE-half NHWC tensors [pixel][Cout] instead of
 * half pairs -- the last launch of an exact (f32 + float16x3) PREFIX in front of a float16 trunk
 * (PoseProposalNet(compute_dtype="float16", exact_prefix=3): stem + layer3 exact, 251 of the reference's 260 people). */
#define PPN_CONV_X3_PLAIN_OUT 8

/* GEMM-depth step / channel tile the packer must pad to for a conv of this shape and dtype, and the order of
 * the GEMM depth index in the packed weight rows:
 *   k_order 0:  k = (ky*ksize + kx)*cin + ci                      (tap-major)
 *   k_order 1:  k = ((ci / k_step)*ksize*ksize + ky*ksize + kx)*k_step + ci % k_step
 *               (channel-chunk-major: the 9 taps of one 64-channel slab are consecutive K steps, so the
 *                shifted re-reads of an input row hit L2 instead of the Infinity Cache)
 *   k_order 2:  direct small-channel kernel (Cin 16, 3x3): the weight stays in the reference layout
 *               [cout][cin][3][3] as f32 for either dtype; k_step = k_total = 144, cout_tile = cout        */
int ppn_conv_tiling(int32_t dtype, int32_t cin, int32_t cout, int32_t ksize, int32_t* k_step, int32_t* cout_tile,
                    int32_t* k_order);

int ppn_conv2d_fused(const ppn_conv_desc* d, void* stream);

/* One launch per 64-channel stride-1 pre-activation BasicBlock (drn.py:25-57; DRN-D's layer3 behind its first block), 16-bit
 * modes (csrc/block64.hip, round 5):
 *   mid     = act_mid(conv3x3(src, weight1) * scale_mid + shift_mid)        never written to HBM
 *   v       = act1(conv3x3(mid, weight2) * scale1 + shift1) (+ residual)
 *   out_raw = v;   out_act = act2(v * scale2 + shift2)
 * i.e. the two ppn_conv2d_fused launches of the block (conv1 with out_raw = mid, conv2 reading it) in one persistent kernel:
 * `src` is the block's pre-activated input relu(bn1(x)), `residual` the raw x.  Both weights are packed [64][576] with
 * k = tap * 64 + ci (ppn_pack_weight k_order 0).  Outputs are BIT-IDENTICAL to the two launches (same K order, same epilogue
 * arithmetic, mid rounded to the 16-bit type as the stored tensor was).  All tensors NHWC [batch][h][w][64] of `dtype`. */
typedef struct ppn_block_desc {
    int32_t dtype;               /* PPN_BF16 or PPN_F16 */
    int32_t batch, h, w, channels;   /* channels = 64 */
    const void* src;
    const void* residual;        /* or NULL */
    const void* weight1;
    const float* scale_mid;      /* [64] or NULL (=1) */
    const float* shift_mid;      /* [64] or NULL (=0) */
    int32_t act_mid;
    const void* weight2;
    const float* scale1;
    const float* shift1;
    int32_t act1;
    const float* scale2;
    const float* shift2;
    int32_t act2;
    void* out_raw;               /* or NULL */
    void* out_act;               /* or NULL */
    int32_t flags;               /* 0 */
    /* stride 2 (round 5; 0 / 1 = the stride-1 block above): the FIRST block of layer3 (drn.py:168-190) -- conv1 is a 3x3
     * stride-2 convolution from 32 channels, `src` = relu(bn1(x)) NHWC [batch][in_h][in_w][32], weight1 packed [64][w1_ld] with
     * k = tap * 32 + ci (ppn_pack_weight k_order 0); the shortcut is bn_ds(conv1x1_stride2(x)): `proj_src` = x AT THE EVEN
     * PIXELS, NHWC [batch][h][w][32] (what the fused stem writes under PPN_STEM_RAW_S2), proj_weight packed [64][proj_ld] with
     * k = ci, proj_scale / proj_shift the folded BN; `residual` must be NULL.  Outputs [batch][h][w][64], h = (in_h - 1) / 2 + 1.
     * Bit-identical to the three ppn_conv2d_fused launches it replaces (downsample, conv1, conv2 + residual). */
    int32_t stride, in_h, in_w;
    const void* proj_src;
    const void* proj_weight;
    const float* proj_scale;
    const float* proj_shift;
    int32_t w1_ld, proj_ld;
} ppn_block_desc;
int ppn_basicblock64_fused(const ppn_block_desc* d, void* stream);

/* PPN_F16X3 helpers.
 * ppn_pack_weight_x3: w f32 [cout][cin][k][k] (device) -> out half [cout_pad][3 * k_pad], k_pad = k*k*cin (cin % 64 == 0),
 *   row layout per 64-channel slab sl: [half(ws) taps x 64 | half(ws - half(ws)) taps x 64 | half(ws * 2^-11) taps x 64]
 *   with ws = w * 2^scale_log2; the caller multiplies scale1 by 2^-scale_log2 (ppn_conv_desc.k_total = 3 * k_pad).
 * ppn_split_f16x3: src f32 NHWC [pixels][channels] -> dst half [pixels][hi(channels) | lo'(channels)]; channels % 8 == 0. */
int ppn_pack_weight_x3(const float* w, int32_t cout, int32_t cin, int32_t ksize, int32_t cout_pad, int32_t scale_log2,
                       void* out, void* stream);
int ppn_split_f16x3(const float* src, int64_t pixels, int32_t channels, void* dst, void* stream);

/* Where the launcher would cut the output pixels [0, m) of a conv with this Cin/Cout into two launches under tile
 * policy 2 (ppn_set_conv_tile_policy): *m_split pixels run whole rounds (256 CUs) of the most efficient large tile, the
 * remaining m - *m_split a smaller tile that fills one more round (e.g. 512 -> 512 at 32 x 48 x 48 = 73 728 pixels:
 * 65 536 on 256 x 256 tiles = 2 rounds, 8 192 on 128 x 128 tiles = 1 round, instead of 3 rounds of 192 x 256).
 * *m_split = 0: a single launch -- always under policies 0 and 1.  Each output element accumulates its GEMM depth in
 * the same order whatever tile computes it, so results do not depend on the cut. */
int ppn_conv_split(int32_t dtype, int32_t cin, int32_t cout, int64_t m, int64_t* m_split);

/*
 * Frame ingest (rt_test.py:150-157 grab_frame): cv2.resize(frame, (dst_w, dst_h)) [INTER_LINEAR, 8-bit fixed point],
 * cv2.flip(.,0) + cv2.flip(.,1) when `flip`, cv2.COLOR_BGR2RGB when `swap_rb`, on the device.
 *   src_bgr  u8 [batch, src_h, src_w, 3]   camera frames as cv2.VideoCapture.read() delivers them
 *   dst_rgb  u8 [batch, dst_h, dst_w, 3]   e.g. the conv plan's own input buffer (PoseProposalNet.input_buffer)
 * The arithmetic is OpenCV's (oracle/ingest_ref.py restates it; parity unpinned: cv2 is not in this image). */
int ppn_ingest_frames(const void* src_bgr, int32_t batch, int32_t src_h, int32_t src_w, void* dst_rgb, int32_t dst_h,
                      int32_t dst_w, int32_t flip, int32_t swap_rb, void* stream);

/* Process-wide tile choice of the large-tile convolution kernel.  0 (default): one launch at a time -- tiles are
 * sized so that the workgroup count fills whole rounds of the 256 CUs.  1: several launches are in flight on
 * different streams (rt.MultiLaneInference) -- a partial last round is filled by the other stream's workgroups, so
 * the most efficient tile shape is taken regardless of the round count (measured +4..5 % with two lanes). */
/* 2: as 0, plus two-segment launches where ppn_conv_split cuts (opt-in: measured 2-4 % slower end to end because the
 * single round of small tiles is slower than modelled; conv_big.hip::big_split_for has the numbers). */
int ppn_set_conv_tile_policy(int32_t policy);

/* Force the (pixels x channels) tile of the large-tile convolution kernel for every later launch / plan entry whose
 * Cout class admits it: bp in {128,192,256}, bc in {64,128,256} (not 192x64); (0,0) restores the automatic choice.
 * Test and tuning hook: the automatic choice depends on B*Ho*Wo, so small parity problems would otherwise never
 * reach the instantiations a batch-32 384x384 forward runs (tests/test_conv_tiles_gpu.py).  Process-wide, like the
 * PPN_CONV_TILE="bp,bc" environment knob that supplies its initial value. */
int ppn_set_conv_tile_override(int32_t bp, int32_t bc);

/* Test and tuning hook: 0 routes the 64 -> 64 3x3 stride-1 convolutions of the 16-bit modes through the generic
 * implicit-GEMM kernels instead of the register-resident filter-bank kernel (csrc/conv64.hip); results are bit-identical
 * either way (tests/test_conv_tiles_gpu.py).  Process-wide; initial value 1 unless PPN_CONV64=0 is in the environment.
 * Product code uses ppn_conv_desc.flags (PPN_CONV_NO_FILTER_BANK) per descriptor / plan instead. */
int ppn_set_conv64_enabled(int32_t on);

/* Name of the kernel instantiation the calling thread's last successful ppn_conv2d_fused launched
 * (e.g. "conv_igemm_big_kernel<__bf16, 192, 256, 8, false>"); "" before the first call. */
const char* ppn_last_conv_kernel(void);

/*
 * First layer (drn.py:123-128 layer0: 7x7 conv 3->16, BN, ReLU) with the input normalisation of
 * rt_test.py:97-101 / aug.py:149-153 fused into the load.
 *   src      u8  [B,H,W,3]  RGB frame (src_is_u8=1): (x-mean_c)/std_c applied on the fly, or
 *            f32 [B,3,H,W]  already normalised NCHW input, the model.forward() argument (src_is_u8=0)
 *   weight   f32 [16,3,7,7] (reference layout, device), scale/shift f32[16] folded BN (device); both NULL =
 *            train mode: the plain convolution output is written (no affine, no ReLU)
 *   mean,std_ HOST pointers to 3 floats (read at call time; may be NULL when src_is_u8=0)
 *   out      NHWC [B,H,W,16] dtype
 */
int ppn_stem7x7(int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w,
                const float* weight, const float* scale, const float* shift, const float* mean, const float* std_,
                void* out, void* stream);

/*
 * layer0 + layer1 (drn.py:123-130) in one launch: as ppn_stem7x7, then 3x3 conv 16->16 + BN + ReLU on the tile that
 * is still in LDS; the 16-channel tensor between the two layers never goes to HBM.
 *   w1 f32 [16,16,3,3] (reference layout, device), scale1/shift1 f32[16] folded BN of layer1.  out NHWC [B,H,W,16].
 */
int ppn_stem01(int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w,
               const float* w0, const float* scale0, const float* shift0, const float* mean, const float* std_,
               const float* w1, const float* scale1, const float* shift1, void* out, void* stream);

/* A recorded sequence of launches (one forward pass): replayed in order on `stream`. */
typedef struct ppn_plan ppn_plan;
int ppn_plan_create(ppn_plan** out);
int ppn_plan_add_conv(ppn_plan* p, const ppn_conv_desc* d);
/* A whole 64-channel BasicBlock as one entry (ppn_basicblock64_fused). */
int ppn_plan_add_block(ppn_plan* p, const ppn_block_desc* d);
/* Zero `bytes` at `ptr` as a step of the plan (the arg-max keys of the fused head conv); runs as a KERNEL with
 * 16-byte stores, never a memset node of the captured graph: `ptr` must be 16-byte aligned (PPN_E_INVALID else). */
int ppn_plan_add_memset(ppn_plan* p, void* ptr, size_t bytes);
/* ppn_split_f16x3 as a step of the plan (PPN_F16X3 plans: behind the PPN_F32 launches whose outputs feed split convs). */
int ppn_plan_add_split(ppn_plan* p, const float* src, int64_t pixels, int32_t channels, void* dst);
int ppn_plan_add_stem(ppn_plan* p, int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h,
                      int32_t w, const float* weight, const float* scale, const float* shift, const float* mean,
                      const float* std_, void* out);
int ppn_plan_add_stem01(ppn_plan* p, int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h,
                        int32_t w, const float* w0, const float* scale0, const float* shift0, const float* mean,
                        const float* std_, const float* w1, const float* scale1, const float* shift1, void* out);
/* The whole stem in one launch, bf16 mode (csrc/stem012.hip): layer0 7x7 3->16, layer1 3x3 16->16, layer2 3x3 stride 2
 * 16->32, each + folded BN + ReLU (drn.py:123-133), input normalisation fused; out_raw = layer2's output, out_act =
 * relu(out_raw * scale3 + shift3), the pre-activation of the first BasicBlock (either may be NULL).  Outputs NHWC bf16
 * [B, (H+1)/2, (W+1)/2, 32].  Weights f32 in the reference layout; bit-identical to running ppn_stem7x7 +
 * two ppn_conv2d_fused launches in bf16 mode. */
int ppn_stem012(int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w, const float* w0,
                const float* scale0, const float* shift0, const float* mean, const float* std_, const float* w1,
                const float* scale1, const float* shift1, const float* w2, const float* scale2, const float* shift2,
                const float* scale3, const float* shift3, void* out_raw, void* out_act, void* stream);
int ppn_plan_add_stem012(ppn_plan* p, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w,
                         const float* w0, const float* scale0, const float* shift0, const float* mean, const float* std_,
                         const float* w1, const float* scale1, const float* shift1, const float* w2, const float* scale2,
                         const float* shift2, const float* scale3, const float* shift3, void* out_raw, void* out_act);
/* The same with the 16-bit type as an argument: dtype = PPN_BF16 (== the two entry points above) or PPN_F16 (IEEE half
 * storage and MFMA operands, outputs NHWC f16). */
/* dtype of the *_dt entry points: PPN_BF16 / PPN_F16 = the stem's MFMA operand type, on-chip storage type and output
 * type; PPN_STEM_IO(PPN_F16, PPN_BF16) = IEEE-half operands and on-chip tensors, bf16 OUTPUT tensors (what the bf16 mode runs
 * since round 4: the stem's rounding noise is amplified by every layer behind it, csrc/stem012.hip). */
#define PPN_STEM_IO(internal, out) ((internal) | (((out) + 1) << 8))
/* PPN_STEM_RAW_S2 (or-ed into the dtype of the *_dt entry points; round 5): out_raw receives only the pixels with even row AND
 * column, as NHWC [batch, (Ho + 1) / 2, (Wo + 1) / 2, 32] -- when the raw stem output's only reader is the first BasicBlock's
 * 1x1 stride-2 projection (drn.py:53-54, 176-181), which reads exactly those pixels: 19 instead of 75 MB written at batch 32,
 * and the projection becomes a stride-1 launch over a dense tensor instead of a gather of half-used 128-byte lines. */
#define PPN_STEM_RAW_S2 (1 << 16)
int ppn_stem012_dt(int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h, int32_t w, const float* w0,
                   const float* scale0, const float* shift0, const float* mean, const float* std_, const float* w1,
                   const float* scale1, const float* shift1, const float* w2, const float* scale2, const float* shift2,
                   const float* scale3, const float* shift3, void* out_raw, void* out_act, void* stream);
int ppn_plan_add_stem012_dt(ppn_plan* p, int32_t dtype, int32_t src_is_u8, const void* src, int32_t batch, int32_t h,
                            int32_t w, const float* w0, const float* scale0, const float* shift0, const float* mean,
                            const float* std_, const float* w1, const float* scale1, const float* shift1, const float* w2,
                            const float* scale2, const float* shift2, const float* scale3, const float* shift3,
                            void* out_raw, void* out_act);
/* Re-point the first layer's input (same shape/dtype as at ppn_plan_add_stem) before a run. */
int ppn_plan_set_input(ppn_plan* p, const void* src);
/* Issue every launch of the plan on `stream`.  After two launch-by-launch runs the sequence is captured into a
 * hipGraph per (input pointer, stream) and later runs are one hipGraphLaunch (PPN_PLAN_GRAPH=0 disables that; a
 * failed capture falls back to launch-by-launch -- the same kernels either way). */
int ppn_plan_run(ppn_plan* p, void* stream);
/* Same as ppn_plan_run but brackets every launch with HIP events on `stream`; ms[i] = duration of launch i.
 * Each launch is issued `repeats` (>= 1) times back to back between its two events and the elapsed time is
 * divided by `repeats`, which amortises the event/launch gap (launches are idempotent: no output aliases an
 * input).  Synchronises on the last event. */
int ppn_plan_run_timed(ppn_plan* p, void* stream, float* ms, int32_t n_ms, int32_t repeats);
/* Number of times the plan's launch sequence has been captured into a hipGraph (0 while it still launches directly).
 * A plan re-captures when its input pointer or its stream changes -- the old executable graph is retired only after
 * its stream has drained -- so callers that serve fresh frames keep ONE plan-owned input buffer per plan and copy
 * into it (PoseProposalNet._plan_for does); this counter lets tests assert that. */
int ppn_plan_graph_captures(const ppn_plan* p);
int ppn_plan_size(const ppn_plan* p);
/* Name of the kernel launch i dispatches (as it appears in rocprofv3 --kernel-trace). */
const char* ppn_plan_kernel_name(const ppn_plan* p, int32_t i);
int ppn_plan_destroy(ppn_plan* p);

/* Weight packing: reference layout f32 [cout,cin,k,k] -> [cout_pad][k_total] dtype, zero padded, with the
 * depth order/step reported by ppn_conv_tiling. */
int ppn_pack_weight(int32_t dtype, const float* w, int32_t cout, int32_t cin, int32_t ksize, int32_t cout_pad,
                    int32_t k_total, int32_t k_order, int32_t k_step, void* out, void* stream);
/* Same, for the input-gradient convolution of a layer whose FORWARD weight is w f32 [cin,cout,k,k]: packs
 * w'[co][ci][ky][kx] = w[ci][co][k-1-ky][k-1-kx] (cout/cin are those of the gradient convolution, i.e. the
 * forward layer's cin/cout).  ppn_conv2d_fused on dy with this weight is autograd's conv input gradient. */
int ppn_pack_weight_dgrad(int32_t dtype, const float* w, int32_t cout, int32_t cin, int32_t ksize, int32_t cout_pad,
                          int32_t k_total, int32_t k_order, int32_t k_step, void* out, void* stream);

/* All weight packs of a training iteration in ONE launch (round 4: a DRN-D-22 step packed its ~86 weight views -- each
 * layer's forward and input-gradient layout -- with one 8 us launch apiece, every optimiser step; the reference has no
 * counterpart: cuDNN reads its [cout,cin,k,k] parameters directly, /root/reference/main.py:643-777).
 * items (host): one entry per ppn_pack_weight / ppn_pack_weight_dgrad call it replaces, same arguments, same result.
 * ppn_pack_table_build lays the entries out for the device: `table` (host, n * PPN_PACK_ITEM_BYTES bytes) is what the
 * caller copies ONCE to device memory; total_blocks is the grid of ppn_pack_table_run, which re-packs every entry from
 * the current values of its w on `stream`.  Pointers are captured: rebuild the table when a w or out moves. */
typedef struct ppn_pack_item {
    const float* w;      /* device, reference layout f32 */
    void* out;           /* device, [cout_pad][k_total] of dtype (f32 for k_order 2) */
    int32_t dtype, cout, cin, ksize, cout_pad, k_total, k_order, k_step;
    int32_t transposed;  /* 1: ppn_pack_weight_dgrad's layout */
    int32_t reserved_;
} ppn_pack_item;
#define PPN_PACK_ITEM_BYTES 64
int ppn_pack_table_build(const ppn_pack_item* items, int32_t n, void* table, int32_t* total_blocks);
int ppn_pack_table_run(const void* table_dev, int32_t n, int32_t total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training building blocks (main.py:623-777).  Activations are NHWC
 * [pixels][channels] in `dtype`, channels a multiple of 8 and a power of two <= 2048 (every BN of the
 * DRN-D / PPN stack); per-channel parameters, statistics and gradients are f32.
 * ---------------------------------------------------------------------------------------- */

/*
 * A16: nn.BatchNorm2d in train mode (implicit in model.train(), main.py:643) fused with the activation that
 * follows it (drn.py:44-51 ReLU, model.py:113-131 LeakyReLU(0.1)):
 *     mean_c, var_c  = batch statistics over all pixels (biased variance)
 *     y              = act((x - mean_c) / sqrt(var_c + eps) * gamma_c + beta_c)
 *     running_mean   = (1-momentum) * running_mean + momentum * mean_c
 *     running_var    = (1-momentum) * running_var  + momentum * var_c * n/(n-1)
 * save_mean / save_rstd [C] are kept for the backward; scale / shift [C] are the folded affine
 * (y = act(x*scale + shift)) a fused consumer can use instead of y.  y, running_*, scale, shift may be NULL.
 */
typedef struct ppn_bn_desc {
    int32_t dtype;
    int32_t channels;
    int64_t pixels;
    int32_t act;                  /* PPN_ACT_NONE / RELU / LRELU(0.1) */
    float eps, momentum;
    const void* x;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* save_mean;
    float* save_rstd;
    float* scale;
    float* shift;
    void* y;
    void* workspace;              /* >= ppn_bn_workspace_bytes(channels) */
    /* > 0: the workspace already holds that many blocks of partial sums { sum x, sum x^2 } per channel, written by the
     * convolution that produced x (ppn_conv_desc.stats_mode 1, *stats_tiles): the reduction pass over x is skipped. */
    int32_t stats_blocks;
    /* non-NULL (HOST int, needs y): the apply pass also folds { sum y, sum y^2 } of the values it stores into the workspace, as
     * the reduction pass of a BatchNorm over y would (bit for bit); *emit_blocks receives the number of partial blocks -- the
     * stats_blocks of that next call (same channel count, same workspace, nothing in between).  A pre-activation block's bn1
     * behind a conv-BN-ReLU unit (drn.py:47-63 after drn.py:205-218). */
    int32_t* emit_blocks;
} ppn_bn_desc;

size_t ppn_bn_workspace_bytes(int32_t channels);
int ppn_bn_train_fwd(const ppn_bn_desc* d, void* stream);

/*
 * Backward of the same fused BN + activation.  dy is the gradient w.r.t. y; with z = x_hat*gamma + beta,
 * g = dy * act'(z):   dgamma = sum g*x_hat,  dbeta = sum g,
 *                     dx = gamma*rstd * (g - dbeta/n - x_hat*dgamma/n)  (+ dx_add, e.g. the skip-path gradient)
 * dgamma / dbeta are OVERWRITTEN.  dx may alias dy or dx_add.
 */
typedef struct ppn_bn_bwd_desc {
    int32_t dtype;
    int32_t channels;
    int64_t pixels;
    int32_t act;
    const void* x;
    const void* dy;
    const void* dx_add;           /* NULL or [pixels][channels] */
    const float* gamma;
    const float* beta;
    const float* save_mean;
    const float* save_rstd;
    float* dgamma;
    float* dbeta;
    void* dx;
    void* workspace;              /* >= ppn_bn_workspace_bytes(channels) */
    /* > 0: the workspace already holds that many blocks of { sum g, sum g * xhat } per channel, written by the input-gradient
     * convolution that produced dy (ppn_conv_desc.stats_mode 2): the reduction pass over x and dy is skipped (single stream only). */
    int32_t stats_blocks;
    /* next_x non-NULL: the dx this call writes is the dy of ANOTHER BatchNorm (+ next_act) over next_x [pixels][channels] --
     * the projection shortcut's BatchNorm of the block below, or the conv-BN-ReLU unit below.  The apply pass folds that
     * BatchNorm's { sum g, sum g * xhat } from the dx it stores into the workspace (bit for bit what its reduction pass would
     * compute); *next_blocks (HOST int) receives the number of partial blocks = the stats_blocks of that next call. */
    const void* next_x;
    const float *next_gamma, *next_beta, *next_mean, *next_rstd;
    int32_t next_act;
    int32_t* next_blocks;
} ppn_bn_bwd_desc;

int ppn_bn_train_bwd(const ppn_bn_bwd_desc* d, void* stream);

/* out[c] = sum over pixels of x[p][c] (NHWC, channels a power of two in [8,2048]); the bias gradient of a
 * convolution whose output gradient is x (model.py:91 conv2.bias).  workspace >= ppn_bn_workspace_bytes(channels). */
int ppn_colsum(int32_t dtype, const void* x, int64_t pixels, int32_t channels, float* out, void* workspace,
               void* stream);

/*
 * Backward through the head's sigmoid (model.py:134) + relayout for the conv3 backward kernels:
 *     dz[b][hw][c] = grad_head[b][c][hw] * s*(1-s),  s = head[b][c][hw]          (NCHW f32 -> NHWC `dtype`)
 * Only the first channels_used (<= channels) channels of every image are converted (a probe pass of a unary loss
 * uses 6K of the 7605); dz has channels_pad (multiple of 64 >= channels_used) channels, the padding is zeros;
 * dbias (optional, f32[channels]) = sum over b,hw of the same quantity = d/d(conv3.bias).
 */
int ppn_head_grad(int32_t dtype, const float* head, const float* grad_head, int32_t batch, int32_t channels,
                  int32_t hw, int32_t channels_used, int32_t channels_pad, void* dz, float* dbias, void* stream);

/* Bottleneck tail (drn.py:92-95): out = relu(z + r), and its backward dz = dout * (out > 0) (+ add); n elements,
 * a multiple of 8, any contiguous layout. */
int ppn_add_relu(int32_t dtype, const void* z, const void* r, int64_t n, void* out, void* stream);
int ppn_relu_mask(int32_t dtype, const void* out, const void* dout, const void* add, int64_t n, void* dz, void* stream);

/* Data movement of the stride-2 input gradients and of the 7x7 layer's weight-gradient input in ONE pass each (torch did a
 * fill + a strided copy_, or four strided copies; main.py:677-683 leaves all of it to autograd / cuDNN):
 * ppn_upsample_zero:     dst [B][dst_h][dst_w][C] = src [B][src_h][src_w][C] at the pixels (y, x) = stride * (i, j), 0 elsewhere
 *                        (a pixel = a multiple of 16 bytes);
 * ppn_interleave_parity: dx [B][h][w][C] from the four parity sub-convolutions o_{py,px} [B][Ho+1][Wo+1][C] (Ho = (h+2-3)/2+1)
 *                        of a stride-2 3x3 input gradient: dx[y][x] = o_{y&1,x&1}[(y + (y&1)) / 2][(x + (x&1)) / 2];
 * ppn_image_to_nhwc:     f32 NCHW [B][3][h][w] -> NHWC [B][h][w][channels_pad] of dtype (channels_pad 4 or 8, channels 3.. zero). */
int ppn_upsample_zero(int32_t dtype, const void* src, int32_t batch, int32_t src_h, int32_t src_w, int32_t channels,
                      int32_t stride, int32_t dst_h, int32_t dst_w, void* dst, void* stream);
int ppn_interleave_parity(int32_t dtype, const void* o00, const void* o01, const void* o10, const void* o11, int32_t batch,
                          int32_t h, int32_t w, int32_t channels, void* dx, void* stream);
/* ppn_interleave_parity with the four sub-convolutions stacked along the channels of one tensor o [B][Ho+1][Wo+1][4 C]
 * (parity (py, px) in channel block 2 py + px): the output of ONE convolution with the four 2 x 2 filter sets stacked along
 * its output channels, which reads dy once instead of four times. */
int ppn_interleave_parity_stacked(int32_t dtype, const void* o, int32_t batch, int32_t h, int32_t w, int32_t channels, void* dx,
                                  void* stream);
int ppn_image_to_nhwc(int32_t dtype, const float* src, int32_t batch, int32_t h, int32_t w, int32_t channels_pad, void* dst,
                      void* stream);

/*
 * A15: one torch.optim.Adam step (main.py:278-279: betas (0.9, 0.999), eps 1e-8, weight_decay 0, no amsgrad)
 * over a flat f32 buffer -- the whole model is one launch.  `step` is the 1-based step count.
 *     m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g
 *     p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * grad_scale multiplies g first (1/world_size after the SUM all-reduce, main.py:1233-1238).
 * param_lp (optional) receives the updated parameters rounded to bf16 (the copy the bf16 kernels read).
 * Hyper-parameters are doubles (Python floats): torch forms 1-beta and the bias corrections in double.
 */
int ppn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int32_t step, float grad_scale,
                  void* param_lp, void* stream);

/* sum of squares of a f32 buffer, written (not accumulated) to out[0]; deterministic.  n <= 2^31.
 * workspace >= 1024 doubles.  Building block of G_i = ||d(w_i L_i)/dW||_2 (main.py:704-717). */
int ppn_sumsq(const float* x, int64_t n, float* out, void* workspace, void* stream);

/*
 * The limb loss's probe gradient by linearity of the backward pass (the four unary probe gradients g_0..3 =
 * dL_i/dW of main.py:704-708 and total = sum_i c_i dL_i/dW from loss.backward(), main.py:677-683) and every norm
 * the GradNorm step needs of them, in ONE pass:
 *     gw4 = (total - sum_{i<4} c_i g_i) / c_4                      (f32[n] device, written)
 *     stats[0..3] = ||g_i||^2,  stats[4] = ||gw4||^2,  stats[5] = ||total - sum c_i g_i||^2,  stats[6] = ||total||^2
 * coeff: HOST pointer to 5 floats c_i, c_4 != 0.  Every sum equals ppn_sumsq of that tensor (same mapping and
 * reduction order; deterministic).  workspace >= 1024 * 7 doubles.
 */
int ppn_gradnorm_probe_stats(const float* g0, const float* g1, const float* g2, const float* g3, const float* total,
                             const float* coeff, int64_t n, float* gw4, float* stats, void* workspace, void* stream);

/*
 * A13: the task-weight half of the GradNorm step (main.py:717-765), all on device, one launch:
 *     l_i = w_i*L_i;  G_i = w_i*gnorm_i;  G_avg = mean G;  lhat_i = l_i/base_i;  r_i = lhat_i/mean(lhat)
 *     C_i = G_avg * r_i^alpha (constant);  Lgrad = sum |G_i - C_i|;  dLgrad/dw_i = sign(G_i - C_i)*gnorm_i
 *     Adam step on w (optimizerR)
 * gnorm_i = ||dL_i/dW||_2 for the probe weight W (head conv1.weight).  out5x4 (optional, f32[20]) receives
 * G, C, dw and [Lgrad, G_avg, 0, 0, 0] for logging / tests.
 */
int ppn_gradnorm_weight_step(float* w, const float* losses, const float* gnorm, const float* base, float alpha,
                             float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                             int32_t step, float* out5x4, void* stream);
/* main.py:767-777 after the all-reduce(SUM): w = clamp(w/world, min 0);  w /= mean(w). */
int ppn_gradnorm_renorm(float* w, int32_t world_size, void* stream);

/*
 * Convolution weight gradient (autograd of nn.Conv2d inside loss.backward(), main.py:677-683):
 *     dw[co][ci][ky][kx] = beta*dw + sum_{b,oy,ox} dy[b,oy,ox,co] * x[b, oy*stride+ky*dil-pad, ox*stride+kx*dil-pad, ci]
 * x, dy NHWC in `dtype` (cin, cout multiples of 8 for bf16 / 4 for f32), dw f32 in the REFERENCE layout
 * [cout,cin,k,k] (the layout of the flat gradient buffer the all-reduce and Adam work on).  Deterministic.
 * The input gradient needs no entry point of its own: it is ppn_conv2d_fused on dy with the weights transposed
 * and flipped (see pytorch_pose_proposal_network_amd/train.py: conv_dgrad).
 */
typedef struct ppn_wgrad_desc {
    int32_t dtype;
    int32_t batch, in_h, in_w, cin;
    int32_t out_h, out_w, cout;
    int32_t ksize, stride, dilation, pad;
    float beta;                   /* 0: overwrite dw, 1: accumulate into it */
    const void* x;
    const void* dy;
    float* dw;
    void* workspace;
    uint64_t workspace_bytes;     /* >= ppn_conv_wgrad_workspace_bytes(desc) */
} ppn_wgrad_desc;

/* The workspace holds one f32 partial [k*k][cout][cin] per pixel split.  bf16: a split covers at most 8 096 (>= 256-wide
 * layers) or 4 032 (narrower ones) output pixels -- its x-row offset table lives in LDS -- so the workspace grows with the
 * pixel count: 132 MB for a 512 -> 512 3x3 layer at 32 x 48 x 48 pixels (14 splits), ~75 splits per 600 k pixels. */
size_t ppn_conv_wgrad_workspace_bytes(const ppn_wgrad_desc* d);
int ppn_conv_wgrad(const ppn_wgrad_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Second-order pieces: GradNorm's Lgrad.backward() (main.py:759) differentiates the probe gradients
 * G_iR = d(w_i L_i)/dW (created with create_graph=True, main.py:704-708) once more.  With v_i = g_i/||g_i||,
 *     d||g_i||/dtheta = d/dtheta <grad_s L_i(s), sdot_i>,   sdot_i = forward-mode tangent of the head along the
 * weight direction v_i; only the head's tail (bn0_2 .. sigmoid) carries a tangent.  These entry points are the
 * non-linear building blocks of the reverse pass over that dual computation (the convolutions reuse
 * ppn_conv2d_fused / ppn_conv_wgrad on both streams).
 * ---------------------------------------------------------------------------------------- */

/* dst[b][hw][c] = src[b][c][hw] (f32 NCHW -> NHWC `dtype`, channels_pad a multiple of 64, padding zeroed). */
int ppn_nchw_to_nhwc(int32_t dtype, const float* src, int32_t batch, int32_t channels, int32_t hw,
                     int32_t channels_used, int32_t channels_pad, void* dst, void* stream);

/* dx = dy * act'(x*scale + shift) with the BN affine of (gamma, beta, save_mean, save_rstd): the activation mask of
 * a tangent that has passed the BN (its forward-mode image is ppn_bn_train_bwd with act NONE applied to the
 * tangent).  Uses x, dy, gamma, beta, save_mean, save_rstd, act, dx of the descriptor. */
int ppn_bn_act_mask(const ppn_bn_bwd_desc* d, void* stream);

/* Adjoint of the train-mode BN tangent  ydot = gamma*rstd*(xdot - mean(xdot) - xhat*mean(xhat*xdot))  followed by
 * the activation mask, with respect to x and gamma:  d->dy is the adjoint arriving at the masked tangent, `xdot`
 * the tangent that entered the BN.  d->dx is ACCUMULATED into (add it to the ordinary ppn_bn_train_bwd result);
 * dgamma_tan f32[C] is overwritten.  (The adjoint w.r.t. xdot itself is ppn_bn_train_bwd(x, dy, act).dx.)
 * d->workspace >= ppn_bn_dual_workspace_bytes(channels). */
size_t ppn_bn_dual_workspace_bytes(int32_t channels);
int ppn_bn_dual_bwd(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, void* stream);

/* The three calls above for `nstreams` gradient / tangent streams over the SAME x in one set of launches (round 4: the
 * second-order tail of the GradNorm step pushes five tangent streams -- one per loss, /root/reference/main.py:700-759 --
 * through two train-mode BNs; stream by stream that was ~160 launches of ~8 us per iteration).  Every per-pixel tensor
 * other than x (dy, dx, dx_add, xdot) holds the streams back to back: [nstreams][pixels][channels]; d->pixels is ONE
 * stream's pixel count; dgamma, dbeta, dgamma_tan are [nstreams][channels]; d->workspace >= nstreams *
 * ppn_bn_workspace_bytes (ppn_bn_dual_workspace_bytes for the dual call).  Stream s's results are bit-identical to the
 * single-stream call on its slices. */
int ppn_bn_train_bwd_streams(const ppn_bn_bwd_desc* d, int32_t nstreams, void* stream);
int ppn_bn_act_mask_streams(const ppn_bn_bwd_desc* d, int32_t nstreams, void* stream);
int ppn_bn_dual_bwd_streams(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, int32_t nstreams, void* stream);
/* ... with ONE d->dx [pixels][channels] that accumulates the term of every stream, in stream order (what the tail needs is
 * the adjoint at x summed over the streams: the ordinary backward is linear in dy, so its streams are summed BEFORE the
 * convolutions in front of it -- the primal adjoint chain of the second-order tail is a single stream). */
int ppn_bn_dual_bwd_streams_sum(const ppn_bn_bwd_desc* d, const void* xdot, float* dgamma_tan, int32_t nstreams,
                                void* stream);

/* Head-space seeds for one coefficient vector c (HOST, 5 floats): with s = head, sdot = s(1-s)*tz,
 *     tzbar = sdot_bar * sig'          sdot_bar = d(sum c_i L_i)/ds
 *     zbar  = s_bar*sig' + sdot_bar*sig''*tz,   s_bar = (d2(sum c_i L_i)/ds2) sdot   (dual-number evaluation)
 * all f32.  tz, zbar, tzbar have the head layout [B][C][H*W]; with unary_only != 0 they are compact tensors
 * [B][6K][H*W] holding only the unary channels (c[4] must be 0; weight_ij / te may be NULL). */
int ppn_loss_dual(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch, const float* delta,
                  const float* weight, const float* weight_ij, const float* tx_half, const float* ty_half,
                  const float* tx, const float* ty, const float* tw, const float* th, const float* te,
                  const float* coeff, int32_t unary_only, float* zbar, float* tzbar, void* stream);

/*
 * The limb loss's stream of ppn_loss_dual -- coefficient vector (0, 0, 0, 0, c4) -- with the outputs in the layout the
 * convolutions read: zb, tzb `dtype` [B][H*W][cpad] NHWC (cpad a multiple of 64, >= 6K + E*sH*sW; channels outside the
 * limb range are zero) and zsum f32 [B][ceil(H*W/64)][cpad], the per-block pixel sums of zbar (summed over the first two
 * axes they are conv3.bias' second-order gradient).  Same arithmetic as ppn_loss_dual + ppn_nchw_to_nhwc without the two
 * f32 head-layout intermediates (2 x 17 MB per image written and re-read).
 */
int ppn_loss_limb_dual_nhwc(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch,
                            const float* weight_ij, const float* te, float c4, int32_t dtype, int32_t cpad, void* zb,
                            void* tzb, float* zsum, void* stream);
int ppn_loss_limb_dual_nhwc_c(const ppn_loss_cfg* cfg, const float* head, const float* tz, int32_t batch,
                              const uint8_t* limb_compact, float c4, int32_t dtype, int32_t cpad, void* zb, void* tzb,
                              float* zsum, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PPN_H */
