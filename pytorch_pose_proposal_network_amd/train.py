"""Host side of the training building blocks (reference main.py:623-777).

Thin wrappers over libppn's C ABI -- every function launches hand-written HIP kernels on torch's current stream
and fails loudly when the library is missing (there is no CPU path):

  bn_train_forward / bn_train_backward   nn.BatchNorm2d in train mode + the activation behind it   (A16)
  FlatAdam                               torch.optim.Adam over ONE flat f32 buffer, one launch       (A15)
  allreduce_mean_                        SUM all-reduce over RCCL then 1/world (main.py:1233-1238)   (A14)
  GradNormWeights                        the task-weight half of the GradNorm step (main.py:717-777) (A13)

Activations are NHWC `[N, H, W, C]` (or any `[..., C]` contiguous tensor) in f32 or bf16.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as L

ACT = {"none": 0, "relu": 1, "lrelu": 2}


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.PPN_F32
    if t.dtype == torch.bfloat16:
        return L.PPN_BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _f32(t: torch.Tensor, n: int, name: str) -> int:
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or not t.is_cuda:
        raise ValueError(f"{name}: expected a contiguous f32 device tensor of {n} elements")
    return t.data_ptr()


_ws_cache = {}


def _workspace(channels: int, device, nstreams: int = 1) -> torch.Tensor:
    key = (channels, str(device), torch.cuda.current_stream(device).cuda_stream, nstreams)   # one per stream (partial sums)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(nstreams * L.load().ppn_bn_workspace_bytes(channels), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class ConvStats:
    """BatchNorm partial sums a convolution's epilogue left in the BatchNorm workspace of its stream (ppn_conv_desc.stats_mode):
    `blocks` pixel tiles (0: the launch had no such epilogue -- the BatchNorm call then runs its own reduction pass), in the
    workspace tensor `ws`.  Hand it to the bn_train_forward / bn_train_backward call that FOLLOWS the convolution on the same
    stream, with no other BatchNorm of that channel count in between (they share the workspace)."""
    __slots__ = ("ws", "blocks", "mode", "tensor")

    def __init__(self, ws, blocks, mode, tensor):
        self.ws, self.blocks, self.mode, self.tensor = ws, blocks, mode, tensor


_FUSE_STATS = os.environ.get("PPN_TRAIN_FUSE_STATS", "1") != "0"      # A/B switch of the trainer's use of ConvStats
_FUSE_STATS_BWD = os.environ.get("PPN_TRAIN_FUSE_STATS", "1") != "2"  # "2": the forward sums only


def _stats_blocks(stats, ws, mode: int, tensor: torch.Tensor) -> int:
    if stats is None or stats.blocks == 0:
        return 0
    if stats.mode != mode or stats.ws is not ws or stats.tensor.data_ptr() != tensor.data_ptr():
        raise ValueError("ConvStats belongs to another tensor / workspace / pass")
    return stats.blocks


class BnSaved:
    """What the backward needs: batch mean, 1/sqrt(var+eps) and the folded affine (y = act(x*scale+shift))."""

    def __init__(self, c, device):
        self.mean = torch.empty(c, dtype=torch.float32, device=device)
        self.rstd = torch.empty(c, dtype=torch.float32, device=device)
        self.scale = torch.empty(c, dtype=torch.float32, device=device)
        self.shift = torch.empty(c, dtype=torch.float32, device=device)


def bn_train_forward(x: torch.Tensor, gamma, beta, running_mean=None, running_var=None, act: str = "none",
                     eps: float = 1e-5, momentum: float = 0.1, out: Optional[torch.Tensor] = None,
                     want_output: bool = True, stats: Optional[ConvStats] = None, emit_stats: bool = False):
    """y = act(batch_norm(x)) with batch statistics; running stats updated in place (nn.BatchNorm2d defaults).

    Returns (y, BnSaved).  x: NHWC contiguous, channels last.  stats: the ConvStats of the convolution that produced x
    (conv2d_nhwc(..., stats="fwd")): its epilogue already folded the per-tile sums, the reduction pass over x is skipped.
    emit_stats: y is itself the input of a BatchNorm (a pre-activation block behind a conv-BN-ReLU unit): the apply pass folds
    that BatchNorm's sums as it stores y; returns (y, BnSaved, ConvStats) -- hand the ConvStats to that next call."""
    lib = L.load()
    if not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be a contiguous device tensor (channels last)")
    c = x.shape[-1]
    saved = BnSaved(c, x.device)
    d = L.BnDesc()
    d.dtype, d.channels, d.pixels, d.act = _dtype_code(x), c, x.numel() // c, ACT[act]
    d.eps, d.momentum = eps, momentum
    d.x, d.gamma, d.beta = x.data_ptr(), _f32(gamma, c, "gamma"), _f32(beta, c, "beta")
    if running_mean is not None:
        d.running_mean, d.running_var = _f32(running_mean, c, "running_mean"), _f32(running_var, c, "running_var")
    d.save_mean, d.save_rstd = saved.mean.data_ptr(), saved.rstd.data_ptr()
    d.scale, d.shift = saved.scale.data_ptr(), saved.shift.data_ptr()
    y = None
    if want_output:
        y = out if out is not None else torch.empty_like(x)
        d.y = y.data_ptr()
    ws = _workspace(c, x.device)
    d.workspace = ws.data_ptr()
    d.stats_blocks = _stats_blocks(stats, ws, 1, x)
    if emit_stats:
        if y is None:
            raise ValueError("bn_train_forward: emit_stats needs the output")
        blocks = C.c_int32(0)
        d.emit_blocks = C.pointer(blocks)
    L.check(lib.ppn_bn_train_fwd(C.byref(d), L.current_stream_ptr()), "ppn_bn_train_fwd")
    if emit_stats:
        return y, saved, ConvStats(ws, blocks.value, 1, y)
    return y, saved


def _stacked(x: torch.Tensor, t: torch.Tensor, nstreams: int, name: str):
    """t holds `nstreams` tensors of x's shape back to back along the batch dimension."""
    if (t.shape[0] != nstreams * x.shape[0] or t.shape[1:] != x.shape[1:] or t.dtype != x.dtype or
            not t.is_contiguous()):
        raise ValueError(f"{name} must be {nstreams} x-shaped tensors stacked along dim 0")


def bn_train_backward(x: torch.Tensor, dy: torch.Tensor, gamma, beta, saved: BnSaved, act: str = "none",
                      dx_add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                      dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None, nstreams: int = 1,
                      stats: Optional[ConvStats] = None, next_bn=None):
    """Returns (dx, dgamma, dbeta) for y = act(batch_norm(x)); dx_add (same shape) is added to dx.
    stats: the ConvStats of the input-gradient convolution that produced dy (conv_dgrad(..., bn=...)): skips the reduction pass.
    next_bn = (x2, gamma2, beta2, BnSaved2, act2): dx is the dy of that BatchNorm (+ activation) over x2 (same shape as dx): the
    apply pass folds ITS sums as it stores dx; returns (dx, dgamma, dbeta, ConvStats) -- the ConvStats for that next backward.
    dgamma / dbeta: optional f32[C] destinations (e.g. views of the flat gradient buffer), overwritten.
    nstreams > 1: dy (dx, dx_add) hold that many gradient streams over the SAME x, stacked along dim 0; one set of
    launches (ppn_bn_train_bwd_streams); dgamma / dbeta are [nstreams, C]; stream s == the single call on its slices."""
    lib = L.load()
    c = x.shape[-1]
    _stacked(x, dy, nstreams, "dy")
    if dgamma is None:
        dgamma = torch.empty(nstreams * c, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(nstreams * c, dtype=torch.float32, device=x.device)
    dx = out if out is not None else torch.empty_like(dy)
    _stacked(x, dx, nstreams, "out")
    d = L.BnBwdDesc()
    d.dtype, d.channels, d.pixels, d.act = _dtype_code(x), c, x.numel() // c, ACT[act]
    d.x, d.dy = x.data_ptr(), dy.data_ptr()
    if dx_add is not None:
        _stacked(x, dx_add, nstreams, "dx_add")
        d.dx_add = dx_add.data_ptr()
    d.gamma, d.beta = _f32(gamma, c, "gamma"), _f32(beta, c, "beta")
    d.save_mean, d.save_rstd = saved.mean.data_ptr(), saved.rstd.data_ptr()
    d.dgamma, d.dbeta, d.dx = _f32(dgamma, nstreams * c, "dgamma"), _f32(dbeta, nstreams * c, "dbeta"), dx.data_ptr()
    ws = _workspace(c, x.device, nstreams)
    d.workspace = ws.data_ptr()
    d.stats_blocks = _stats_blocks(stats, ws, 2, dy)
    if next_bn is not None:
        x2, gamma2, beta2, saved2, act2 = next_bn
        if nstreams != 1 or x2.shape != dx.shape or x2.dtype != dx.dtype or not x2.is_contiguous():
            raise ValueError("bn_train_backward: next_bn needs a single stream and a BatchNorm input of dx's shape and type")
        nblocks = C.c_int32(0)
        d.next_x, d.next_act, d.next_blocks = x2.data_ptr(), ACT[act2], C.pointer(nblocks)
        d.next_gamma, d.next_beta = _f32(gamma2, c, "gamma2"), _f32(beta2, c, "beta2")
        d.next_mean, d.next_rstd = saved2.mean.data_ptr(), saved2.rstd.data_ptr()
        L.check(lib.ppn_bn_train_bwd(C.byref(d), L.current_stream_ptr()), "ppn_bn_train_bwd")
        return dx, dgamma, dbeta, ConvStats(ws, nblocks.value, 2, dx)
    if nstreams == 1:
        L.check(lib.ppn_bn_train_bwd(C.byref(d), L.current_stream_ptr()), "ppn_bn_train_bwd")
    else:
        L.check(lib.ppn_bn_train_bwd_streams(C.byref(d), nstreams, L.current_stream_ptr()), "ppn_bn_train_bwd_streams")
        dgamma, dbeta = dgamma.view(nstreams, c), dbeta.view(nstreams, c)
    return dx, dgamma, dbeta


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) (main.py:278-279) over one flat buffer.

    `param` and `grad` are flat f32 device tensors (views of the individual parameters live inside them, which
    is also what makes the gradient all-reduce a single RCCL call); `param_lp` optionally mirrors the updated
    parameters in bf16 for the bf16 kernels."""

    def __init__(self, param: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, param_lp: Optional[torch.Tensor] = None):
        n = param.numel()
        _f32(param, n, "param")
        self.param, self.lr, self.betas, self.eps, self.weight_decay = param, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.param_lp = param_lp
        self.step_count = 0

    def step(self, grad: torch.Tensor, grad_scale: float = 1.0):
        n = self.param.numel()
        _f32(grad, n, "grad")
        self.step_count += 1
        bump_param_version()                          # packed copies of the old parameters are stale
        lp = self.param_lp.data_ptr() if self.param_lp is not None else None
        L.check(L.load().ppn_adam_step(self.param.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), n, self.lr, self.betas[0], self.betas[1],
                                       self.eps, self.weight_decay, self.step_count, grad_scale, lp,
                                       L.current_stream_ptr()), "ppn_adam_step")

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}


def sumsq(x: torch.Tensor) -> torch.Tensor:
    """sum(x*x) of a f32 device tensor as a 1-element f32 device tensor (deterministic)."""
    n = x.numel()
    _f32(x, n, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024, dtype=torch.float64, device=x.device)
    L.check(L.load().ppn_sumsq(x.data_ptr(), n, out.data_ptr(), ws.data_ptr(), L.current_stream_ptr()), "ppn_sumsq")
    return out


def probe_stats(g4, total: torch.Tensor, coeff):
    """(gw4, stats f32[7]) of ppn_gradnorm_probe_stats: the limb probe gradient by linearity and the seven sums of squares
    (||g_0..3||^2, ||gw4||^2, ||unscaled remainder||^2, ||total||^2) in one pass over the five tensors."""
    n = total.numel()
    for t in list(g4) + [total]:
        _f32(t, n, "probe gradient")
    gw4 = torch.empty_like(total)
    st = torch.empty(7, dtype=torch.float32, device=total.device)
    ws = torch.empty(1024 * 7, dtype=torch.float64, device=total.device)
    cf = (C.c_float * 5)(*[float(v) for v in coeff])
    L.check(L.load().ppn_gradnorm_probe_stats(g4[0].data_ptr(), g4[1].data_ptr(), g4[2].data_ptr(), g4[3].data_ptr(),
                                              total.data_ptr(), cf, n, gw4.data_ptr(), st.data_ptr(), ws.data_ptr(),
                                              L.current_stream_ptr()), "ppn_gradnorm_probe_stats")
    return gw4, st


def allreduce_mean_(flat: torch.Tensor, group=None) -> float:
    """SUM all-reduce of one flat buffer (RCCL when the tensor is on a GPU, gloo on CPU); returns the 1/world
    factor the caller folds into its next kernel (FlatAdam.step(grad_scale=...)) instead of a separate divide --
    reduce_tensor of main.py:1233-1238 without the clone and without the elementwise pass."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


class GradNormWeights:
    """The five task weights of main.py:252-265 (`nn.Linear(5,1,bias=False)` filled with 1.0) and their update
    (main.py:717-777).  The caller supplies the five loss values and gnorm_i = ||dL_i/dW||_2 for the probe
    weight (head conv1.weight, params[-13])."""

    def __init__(self, device, lr: float, alpha: float = 0.12, betas=(0.9, 0.999), eps: float = 1e-8):
        self.w = torch.ones(5, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(5, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(5, dtype=torch.float32, device=device)
        self.lr, self.alpha, self.betas, self.eps = lr, alpha, betas, eps
        self.step_count = 0
        self.log = torch.zeros(20, dtype=torch.float32, device=device)
        # staged host copy of w (host_weights): valid only for the generation it was staged in.  EVERY method that lets a
        # kernel write `w` through its raw pointer bumps _gen (torch's tensor version counter never sees those writes).
        self._pin, self._pin_event, self._pin_version, self._pin_gen, self._gen = None, None, -1, -1, 0

    def host_weights(self):
        """The five weights as Python floats (coefficients of the loss kernel).  renorm() stages them into pinned memory
        behind an event, so the next iteration reads them WITHOUT draining the queue: a plain w.tolist() at the start
        of an iteration made the host wait for the whole forward pass it had just enqueued to its own stream and lose
        its run-ahead -- the loss, the head backward and the GradNorm probes were then enqueued at launch latency while
        the GPU idled (~1.5 ms per iteration).  Any in-place write to `w` by the caller invalidates the staged copy."""
        if self._pin_event is not None and self._pin_gen == self._gen and self._pin_version == self.w._version:
            self._pin_event.synchronize()
            return self._pin.tolist()
        return self.w.tolist()

    def bind(self, view: torch.Tensor):
        """Move the five weights into `view` (f32[5] on the same device, e.g. the prefix of the trainer's gradient store,
        so that the last gradient bucket of the data-parallel exchange carries them): same values, same object otherwise."""
        if view.numel() != 5 or view.dtype != torch.float32 or view.device != self.w.device:
            raise ValueError("GradNormWeights.bind: f32[5] on the weights' device")
        view.copy_(self.w)
        self.w = view
        self._gen += 1

    def touched(self):
        """Call after ANY raw-pointer write to `w` outside this class (a direct lib.ppn_gradnorm_* call, a fused kernel):
        invalidates the staged host copy."""
        self._gen += 1

    def local_step(self, losses: torch.Tensor, gnorm: torch.Tensor, base: torch.Tensor):
        """This rank's optimizerR.step() (main.py:717-768): G_i, C_i, Lgrad, dLgrad/dw, Adam on w -- before the
        all-reduce and the renormalisation.  losses, gnorm, base: f32[5] device tensors."""
        lib = L.load()
        self.step_count += 1
        self._gen += 1                                       # w changes behind torch's back: staged copy is stale
        self._pin_event = None
        L.check(lib.ppn_gradnorm_weight_step(self.w.data_ptr(), _f32(losses, 5, "losses"), _f32(gnorm, 5, "gnorm"),
                                             _f32(base, 5, "base"), self.alpha, self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.lr, self.betas[0], self.betas[1],
                                             self.eps, self.step_count, self.log.data_ptr(), L.current_stream_ptr()),
                "ppn_gradnorm_weight_step")
        return self.log

    def renorm(self, world: int = 1):
        """main.py:769-777 after the SUM all-reduce: w / world, clamp, renormalise to sum 5."""
        L.check(L.load().ppn_gradnorm_renorm(self.w.data_ptr(), world, L.current_stream_ptr()), "ppn_gradnorm_renorm")
        self._gen += 1
        if self.w.is_cuda:
            if self._pin is None:
                self._pin = torch.empty(5, dtype=torch.float32, pin_memory=True)
            self._pin.copy_(self.w, non_blocking=True)
            self._pin_event = torch.cuda.Event()
            self._pin_event.record()
            self._pin_version, self._pin_gen = self.w._version, self._gen

    def step(self, losses: torch.Tensor, gnorm: torch.Tensor, base: torch.Tensor, group=None):
        """local_step, SUM all-reduce of the five weights over `group` (main.py:769-771), renorm.
        Returns the log tensor (G, C, dw, [Lgrad, G_avg,...])."""
        import torch.distributed as dist
        self.local_step(losses, gnorm, base)
        world = 1
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
            if world > 1 or _force_dp():
                dist.all_reduce(self.w, op=dist.ReduceOp.SUM, group=group)      # main.py:769-771
        self.renorm(world)
        return self.log


# ---- convolution forward / input gradient / weight gradient on NHWC tensors -------------------------------------

def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, stride: int = 1, dilation: int = 1, pad: int = 0,
                add: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act: int = 0,
                nchw_f32: bool = False, dgrad_of: bool = False, out: Optional[torch.Tensor] = None, stats=None):
    """Raw convolution (no folded BN: train mode keeps BN separate) of an NHWC tensor with a reference-layout
    f32 weight [cout,cin,k,k] on the device, + `add` (NHWC, the residual).  Packs the weight for the MFMA
    kernels on the fly: in training the weights change every step anyway.

    stats: ask the epilogue for the BatchNorm partial sums of the output (ppn_conv_desc.stats_mode) -- "fwd" ({sum v, sum v^2}: the
    output feeds a BatchNorm) or a tuple (x, gamma, beta, BnSaved, act) ({sum g, sum g * xhat}: the output is dy of that BatchNorm
    + activation over x).  Returns (out, ConvStats) then; ConvStats.blocks == 0 where the launch has no such epilogue."""
    lib = L.load()
    dt = _dtype_code(x)
    B, H, W, cin = x.shape
    if dgrad_of:          # w is the FORWARD weight [cin, cout, k, k]: the packer transposes and rotates it
        cin_w, cout, k, _ = w.shape
    else:
        cout, cin_w, k, _ = w.shape
    if cin_w != cin or w.dtype != torch.float32 or not w.is_contiguous() or not x.is_contiguous():
        raise ValueError("conv2d_nhwc: x must be NHWC contiguous and w f32 [cout,cin,k,k] contiguous")
    eff = dilation * (k - 1) + 1
    Ho, Wo = (H + 2 * pad - eff) // stride + 1, (W + 2 * pad - eff) // stride + 1
    kstep, _, korder, ktot, cpad = L.conv_tiling(dt, cin, cout, k)
    st = L.current_stream_ptr()
    # A parameter is packed once per optimiser step and layout (forward / input-gradient): the probe passes and the
    # second-order tail run the same convolutions several times per iteration.  Keyed on the weight's storage, the
    # parameter version (bumped by FlatAdam.step / load_state_dict) and the stream the pack kernel was queued on.
    sp = w.untyped_storage().data_ptr()
    key = (w.data_ptr(), tuple(w.shape), bool(dgrad_of), dt, st) if sp in _param_storages else None
    packed = _pack_cache.get(key) if key is not None else None
    if packed is None or packed[0] != _param_version[0]:
        buf = packed[1] if packed is not None else torch.empty(cpad, ktot, dtype=torch.float32 if korder == 2 else x.dtype,
                                                               device=x.device)
        pack = lib.ppn_pack_weight_dgrad if dgrad_of else lib.ppn_pack_weight
        L.check(pack(dt, w.data_ptr(), cout, cin, k, cpad, ktot, korder, kstep, buf.data_ptr(), st), "ppn_pack_weight")
        if key is not None:
            if packed is None:
                _pack_tables.clear()                # a new entry: repack_all() rebuilds its table
            _pack_cache[key] = (_param_version[0], buf, sp,
                                (w.data_ptr(), buf.data_ptr(), dt, cout, cin, k, cpad, ktot, korder, kstep, 1 if dgrad_of else 0, 0))
        packed = (_param_version[0], buf)
    packed = packed[1]
    oshape, odt = ((B, cout, Ho, Wo), torch.float32) if nchw_f32 else ((B, Ho, Wo, cout), x.dtype)
    if out is None:                               # nchw_f32: the head tensor the loss / decode kernels read (model.py:134-136)
        out = torch.empty(oshape, dtype=odt, device=x.device)
    elif tuple(out.shape) != oshape or out.dtype != odt or not out.is_contiguous():
        raise ValueError("conv2d_nhwc: `out` must be a contiguous tensor of the output's shape and type")
    zero = _zero_page(x.device)
    d = L.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.cin = dt, B, H, W, cin
    d.out_h, d.out_w, d.cout = Ho, Wo, cout
    d.ksize, d.stride, d.dilation, d.pad = k, stride, dilation, pad
    d.k_total, d.cout_pad, d.act1, d.out_nchw_f32 = ktot, cpad, act, 1 if nchw_f32 else 0
    d.src, d.weight, d.zero_page, d.out_raw = x.data_ptr(), packed.data_ptr(), zero.data_ptr(), out.data_ptr()
    # input-gradient convolutions run beside the side stream's weight-gradient kernels: leave out the tiles that only shorten a
    # LONE launch (PPN_CONV_SHARED_GPU: the 144 x 256 tile costs +19 % CU-time); PPN_TRAIN_SHARED=0 / 2: never / every convolution
    if _TRAIN_SHARED == 2 or (_TRAIN_SHARED == 1 and dgrad_of):
        d.flags = L.PPN_CONV_SHARED_GPU
    if bias is not None:
        d.shift1 = _f32(bias, cout, "bias")       # v = act(conv + bias)
    if add is not None:
        if add.shape != out.shape or add.dtype != x.dtype or not add.is_contiguous():
            raise ValueError("conv2d_nhwc: `add` must match the output")
        d.residual = add.data_ptr()
    if _PREFETCH and key is not None:
        # prefetch hint (ppn_conv_desc.prefetch): a training iteration launches the same convolutions in the same order on each
        # stream, so the packed weight that FOLLOWED this one on this stream in the previous iteration is the one to pull into the
        # Infinity Cache now (a wrong guess costs a few MB of reads, nothing else).  Only cached packs take part: their buffers
        # live as long as their trainer's registration (unregister_param_storage clears the chain).
        pp = packed.data_ptr()
        prev = _chain_last.get(st)
        if prev is not None and prev != pp:
            _chain_next[(st, prev)] = (pp, packed.numel() * packed.element_size())
        _chain_last[st] = pp
        nxt = _chain_next.get((st, pp))
        if nxt is not None:
            d.prefetch, d.prefetch_bytes = nxt
    if stats is None:
        L.check(lib.ppn_conv2d_fused(C.byref(d), st), "ppn_conv2d_fused")
        return out
    if nchw_f32:
        raise ValueError("conv2d_nhwc: statistics are for NHWC outputs")
    ws = _workspace(cout, x.device)
    tiles = C.c_int32(0)
    d.stats_partial, d.stats_tiles = ws.data_ptr(), C.pointer(tiles)
    if isinstance(stats, str):
        if stats != "fwd":
            raise ValueError("conv2d_nhwc: stats is 'fwd' or (x, gamma, beta, saved, act)")
        d.stats_mode = 1
    else:
        bx, gamma, beta, saved, bact = stats
        if bx.shape != out.shape or bx.dtype != out.dtype or not bx.is_contiguous():
            raise ValueError("conv2d_nhwc: the BatchNorm input of `stats` must match the output")
        d.stats_mode, d.stats_act, d.stats_x = 2, ACT[bact], bx.data_ptr()
        d.stats_gamma, d.stats_beta = _f32(gamma, cout, "gamma"), _f32(beta, cout, "beta")
        d.stats_mean, d.stats_rstd = saved.mean.data_ptr(), saved.rstd.data_ptr()
    L.check(lib.ppn_conv2d_fused(C.byref(d), st), "ppn_conv2d_fused")
    return out, ConvStats(ws, tiles.value, d.stats_mode, out)


_PREFETCH = os.environ.get("PPN_PREFETCH", "1") != "0"
_TRAIN_SHARED = int(os.environ.get("PPN_TRAIN_SHARED", "1"))
_chain_next: dict = {}              # (stream, packed weight) -> (packed weight launched next on that stream, bytes)
_chain_last: dict = {}              # stream -> packed weight of its latest launch


_zero_pages = {}
# packed-weight cache of conv2d_nhwc (see there).  Only views of a buffer a trainer REGISTERED (its flat parameter
# buffer, register_param_storage) are cached: any other weight tensor -- free-standing or a view of a temporary the
# caching allocator may hand out again at the same address -- is packed on every call.
_pack_cache: dict = {}
_param_version = [0]
_param_storages: dict = {}          # storage data_ptr -> number of live registrations


def register_param_storage(flat: torch.Tensor):
    """The caller owns `flat` for its lifetime and calls bump_param_version() after every in-place edit of it."""
    sp = flat.untyped_storage().data_ptr()
    _param_storages[sp] = _param_storages.get(sp, 0) + 1
    return sp


def unregister_param_storage(sp):
    """Drop a registration and every packed copy made of views of that storage (a dead trainer pins nothing)."""
    n = _param_storages.get(sp, 0) - 1
    if n > 0:
        _param_storages[sp] = n
        return
    _param_storages.pop(sp, None)
    for key in [k for k, v in _pack_cache.items() if v[2] == sp]:
        del _pack_cache[key]
    _pack_tables.clear()
    _chain_next.clear()                 # no prefetch hint may outlive the buffers it points at
    _chain_last.clear()


def bump_param_version():
    """Parameters changed (optimiser step, load_state_dict): packed weights of older versions are stale."""
    _param_version[0] += 1


_pack_tables: dict = {}             # (device, storages) -> (device table, entries, grid, keys): rebuilt when the cache gains a key
_BATCHED_PACK = os.environ.get("PPN_TRAIN_BATCHED_PACK", "1") != "0"


def repack_all(device, storages=None) -> int:
    """Refresh the cached packed weights on `device` from the current parameter values with one launch on the current
    stream (ppn_pack_table_run) and stamp them with the current parameter version; returns the number of entries.
    storages: the registered storages (register_param_storage keys) whose views to refresh -- a trainer passes ITS OWN (its flat
    parameter buffer and its padded conv3 copies), so another live trainer's packed weights, which its kernels may still be
    reading on its streams, are never rewritten from here; None: every registered storage.
    conv2d_nhwc packs a weight view the first time it meets it and on every parameter version after that -- ~86 launches
    of ~8 us per DRN-D-22 iteration (each layer's forward and input-gradient layout); called at the head of a pass (after
    bump_param_version, when every consumer of the previous copies has finished: the trainer joins its side streams
    before the optimiser step) the same work is one ~50 us launch.  Entries packed for another stream are refreshed too:
    their consumers wait for an event of this stream before they run (PPNTrainer._on_side, the probe stream)."""
    if not _BATCHED_PACK:
        return 0
    device = torch.device(device)
    if device.index is None:
        device = torch.device(device.type, torch.cuda.current_device())
    scope = None if storages is None else frozenset(storages)
    keys = [k for k, v in _pack_cache.items() if v[1].device == device and (scope is None or v[2] in scope)]
    if not keys:
        return 0
    lib = L.load()
    tkey = (device, scope)
    tab = _pack_tables.get(tkey)
    if tab is None or tab[3] != keys:
        items = (L.PackItem * len(keys))()
        for it, k in zip(items, keys):
            (it.w, it.out, it.dtype, it.cout, it.cin, it.ksize, it.cout_pad, it.k_total, it.k_order, it.k_step,
             it.transposed, it.reserved_) = _pack_cache[k][3]
        host = torch.empty(len(keys) * L.PPN_PACK_ITEM_BYTES, dtype=torch.uint8)
        grid = C.c_int32(0)
        L.check(lib.ppn_pack_table_build(items, len(keys), host.data_ptr(), C.byref(grid)), "ppn_pack_table_build")
        tab = _pack_tables[tkey] = (host.to(device), len(keys), grid.value, keys)
    L.check(lib.ppn_pack_table_run(tab[0].data_ptr(), tab[1], tab[2], L.current_stream_ptr()), "ppn_pack_table_run")
    ver = _param_version[0]
    for k in keys:
        v = _pack_cache[k]
        _pack_cache[k] = (ver,) + v[1:]
    return len(keys)



def _zero_page(device) -> torch.Tensor:
    z = _zero_pages.get(str(device))
    if z is None:
        z = _zero_pages[str(device)] = torch.zeros(64, dtype=torch.float32, device=device)
    return z


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, in_hw, stride: int = 1, dilation: int = 1, pad: int = 0,
               add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, bn=None):
    """dL/dx of y = conv2d(x, w, stride, dilation, pad) given dy (NHWC): the same implicit-GEMM kernel run on dy
    with the weights transposed (cin <-> cout) and flipped; a strided convolution first spreads dy over a
    zero-filled grid (the transposed-convolution identity).  `add` (NHWC like x) is added (skip-path gradient)."""
    H, W = in_hw
    cout, cin, k, _ = w.shape
    eff = dilation * (k - 1) + 1
    B, Ho, Wo, _ = dy.shape
    # measured at batch 32 (tools/bench_dgrad_s2.py, bit-identical results): 1x1 projections 119 -> 51 and 74 -> 37 us,
    # layer2 (16 -> 32) 402 -> 359 us; the 3x3 layers with >= 32 input channels do not gain (181 -> 187, 106 -> 130 us:
    # four short-K launches and four strided copies cost what the 2.25x fewer FLOPs save) and keep the zero-upsampled form
    # With a skip-path gradient to add, the 16-bit modes keep the zero-upsampled form: its epilogue adds in f32 and rounds
    # ONCE, whereas the parity form could only add after its sub-convolutions have been rounded (two roundings).
    # bn = (x, gamma, beta, BnSaved, act): the result is dy of that BatchNorm + activation over x; returns (dx, ConvStats) for
    # bn_train_backward(..., stats=) -- blocks == 0 wherever the launch that writes dx cannot fold the sums
    if (_S2_PARITY and stride == 2 and dilation == 1 and pad == k // 2 and (k == 1 or (k == 3 and cin <= 16)) and
            (add is None or dy.dtype == torch.float32) and
            Ho == (H + 2 * pad - k) // 2 + 1 and Wo == (W + 2 * pad - k) // 2 + 1):
        if out is not None:
            raise ValueError("conv_dgrad: `out` is for the stride-1 form")
        dx = _dgrad_stride2(dy, w, H, W, add)
        return dx if bn is None else (dx, ConvStats(None, 0, 2, dx))
    hup, wup = H + 2 * pad - eff + 1, W + 2 * pad - eff + 1
    if stride > 1 or (hup, wup) != (Ho, Wo):
        dy = upsample_zero(dy, stride, hup, wup)
    return conv2d_nhwc(dy, w, 1, dilation, eff - 1 - pad, add=add, dgrad_of=True, out=out, stats=bn)


def upsample_zero(src: torch.Tensor, stride: int, dst_h: int, dst_w: int) -> torch.Tensor:
    """dst [B, dst_h, dst_w, C]: src [B, h, w, C] at the pixels stride * (i, j), zero elsewhere -- one write pass
    (ppn_upsample_zero) where torch took a fill and a strided copy_."""
    B, h, w, ch = src.shape
    if not src.is_contiguous():
        src = src.contiguous()
    if (ch * src.element_size()) % 16 != 0:
        dst = torch.zeros(B, dst_h, dst_w, ch, dtype=src.dtype, device=src.device)
        dst[:, ::stride, ::stride][:, :h, :w] = src
        return dst
    dst = torch.empty(B, dst_h, dst_w, ch, dtype=src.dtype, device=src.device)
    L.check(L.load().ppn_upsample_zero(_dtype_code(src), src.data_ptr(), B, h, w, ch, stride, dst_h, dst_w, dst.data_ptr(),
                                       L.current_stream_ptr()), "ppn_upsample_zero")
    return dst


import os as _os
_S2_PARITY = _os.environ.get("PPN_DGRAD_S2_PARITY", "1") != "0"
_S2_STACKED = _os.environ.get("PPN_DGRAD_S2_STACKED", "1") != "0"
_S2_IDX = {}


def _dgrad_stride2(dy: torch.Tensor, w: torch.Tensor, H: int, W: int, add: Optional[torch.Tensor]) -> torch.Tensor:
    """Input gradient of a stride-2 convolution (k = 3 pad 1, or k = 1) WITHOUT the zero-upsampled dy: the input pixels of
    each parity (i mod 2, j mod 2) only ever meet the taps of the matching parity, so dx splits into four stride-1
    convolutions of dy itself with 2 x 2 kernels (k = 3:  dx[2j] = w[1] dy[j],  dx[2j+1] = w[2] dy[j] + w[0] dy[j+1]  per
    axis; absent taps are zero) -- 16 tap-pixel products per 2 x 2 input pixels instead of 36, and no 4x larger zero-filled
    tensor written, scattered into and read back (0.9 ms of the training step's main stream went into the five stride-2
    input gradients of DRN-D-22 that way).  k = 1: dx[2j] = w^T dy[j], odd positions zero."""
    B, Ho, Wo, cout = dy.shape
    cin = w.shape[1]
    k = w.shape[2]
    dev = dy.device
    if k == 1:
        wt = w[:, :, 0, 0].t().contiguous().view(cin, cout, 1, 1)
        o = conv2d_nhwc(dy, wt, 1, 1, 0)
        if o.shape[1] != (H + 1) // 2 or o.shape[2] != (W + 1) // 2:
            o = o[:, :(H + 1) // 2, :(W + 1) // 2].contiguous()
        dx = upsample_zero(o, 2, H, W)
    else:
        idx = _S2_IDX.get(dev)
        if idx is None:
            # tap of the forward filter that kernel position u of parity p multiplies; 3 = the zero tap appended below
            idx = _S2_IDX[dev] = torch.tensor([[3, 1], [2, 0]], dtype=torch.long, device=dev)
        w4 = torch.nn.functional.pad(w.permute(1, 0, 2, 3), (0, 1, 0, 1))            # [cin, cout, 4, 4], index 3 = 0
        wall = w4[:, :, idx[:, :, None, None], idx[None, None, :, :]]                # [cin, cout, py, uy, px, ux]
        wall = wall.permute(2, 4, 0, 1, 3, 5).contiguous()                           # [py, px, cin, cout, uy, ux]
        dx = torch.empty(B, H, W, cin, dtype=dy.dtype, device=dev)
        if _S2_STACKED and cin <= 16 and (cin * dy.element_size()) % 16 == 0:
            # ONE convolution with the four parities' 2 x 2 filters stacked along its output channels (dy is read once, not
            # four times), then one interleaving pass: layer2's input gradient (16 <- 32 channels at 384 x 384, batch 32)
            # 292 -> 176 us, bit-identical.  (cin <= 16: the only shape conv_dgrad sends here; wider ones would change the
            # convolution kernel and with it the f32 summation order the zero-upsampled form is compared against.)
            o4 = conv2d_nhwc(dy, wall.view(4 * cin, cout, 2, 2), 1, 1, 1)                  # [B, Ho + 1, Wo + 1, 4 cin]
            L.check(L.load().ppn_interleave_parity_stacked(_dtype_code(dy), o4.data_ptr(), B, H, W, cin, dx.data_ptr(),
                                                           L.current_stream_ptr()), "ppn_interleave_parity_stacked")
            if add is not None:
                dx += add
            return dx
        o = [[conv2d_nhwc(dy, wall[py, px], 1, 1, 1) for px in (0, 1)] for py in (0, 1)]   # each [B, Ho + 1, Wo + 1, cin]
        if (cin * dy.element_size()) % 16 == 0:
            # dx[:, py::2, px::2] = o[py][px][:, py:py + ny, px:px + nx] for the four parities, in one pass
            L.check(L.load().ppn_interleave_parity(_dtype_code(dy), o[0][0].data_ptr(), o[0][1].data_ptr(), o[1][0].data_ptr(),
                                                   o[1][1].data_ptr(), B, H, W, cin, dx.data_ptr(), L.current_stream_ptr()),
                    "ppn_interleave_parity")
        else:
            for py in (0, 1):
                for px in (0, 1):
                    ny, nx = (H - py + 1) // 2, (W - px + 1) // 2
                    if ny > 0 and nx > 0:
                        dx[:, py::2, px::2] = o[py][px][:, py:py + ny, px:px + nx]
    if add is not None:
        dx += add
    return dx


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, ksize: int, stride: int = 1, dilation: int = 1, pad: int = 0,
               out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dL/dw [cout,cin,k,k] f32 of y = conv2d(x, w) from NHWC x and dy (hand-written MFMA kernel, wgrad.hip)."""
    lib = L.load()
    B, H, W, cin = x.shape
    _, Ho, Wo, cout = dy.shape
    if dy.dtype != x.dtype or not x.is_contiguous() or not dy.is_contiguous():
        raise ValueError("conv_wgrad: x and dy must be contiguous NHWC tensors of one dtype")
    dw = out if out is not None else torch.empty(cout, cin, ksize, ksize, dtype=torch.float32, device=x.device)
    if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.numel() != cout * cin * ksize * ksize:
        raise ValueError("conv_wgrad: out must be a contiguous f32 [cout,cin,k,k] tensor")
    d = L.WgradDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.cin = _dtype_code(x), B, H, W, cin
    d.out_h, d.out_w, d.cout = Ho, Wo, cout
    d.ksize, d.stride, d.dilation, d.pad = ksize, stride, dilation, pad
    d.beta = 1.0 if accumulate else 0.0
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    need = lib.ppn_conv_wgrad_workspace_bytes(C.byref(d))
    if need == 0:
        raise L.PPNError("ppn_conv_wgrad: " + (lib.ppn_last_error() or b"").decode())
    ws = _wgrad_ws(need, x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(lib.ppn_conv_wgrad(C.byref(d), L.current_stream_ptr()), "ppn_conv_wgrad")
    return dw


_wgrad_cache = {}


def _wgrad_ws(nbytes: int, device) -> torch.Tensor:
    # one workspace per stream: launches on one stream are ordered, two streams must not share partial sums
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _wgrad_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _wgrad_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def _force_dp() -> bool:
    import os
    return os.environ.get("PPN_FORCE_DP", "0") == "1"


class BucketedAllReduce:
    """SUM all-reduce of one flat gradient buffer in a few large buckets, each issued (async) as soon as the
    backward pass has produced every gradient in it, so the exchange runs underneath the rest of the backward.

    The flat buffer is in forward (named_parameters) order and the backward fills it from the END towards the front:
    `ready(offset)` says "every element at index >= offset is final".  Buckets are contiguous tail-to-head slices of
    `bucket_elems` elements (default 8 M f32 = 32 MB: xGMI is point-to-point, a ring all-reduce is bound per link, so
    a handful of large messages beats many small ones).  `finish()` issues what is left, waits for all buckets and
    returns the 1/world factor the optimiser kernel applies (reduce_tensor of main.py:1233-1238 without its clone and
    divide passes).  Without an initialised process group (or world size 1) everything is a no-op."""

    def __init__(self, flat: torch.Tensor, bucket_elems: int = 8 * 1024 * 1024, group=None, store=None, before_last=None):
        """store: the allocation `flat` is a trailing view of (trainer: 16 leading floats whose first five are the GradNorm
        task weights); the LAST bucket -- the one that starts at flat[0] -- is then issued from store[0], i.e. the prefix
        rides on it.  before_last(): called once right before that bucket is issued (the task weights' local step)."""
        import torch.distributed as dist
        self.flat, self.group = flat, group
        self.store, self.before_last = store, before_last
        self.prefix = 0
        if store is not None:
            self.prefix = flat.storage_offset() - store.storage_offset()
            if (self.prefix < 0 or store.numel() != self.prefix + flat.numel() or
                    store.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr()):
                raise ValueError("BucketedAllReduce: `flat` must be the trailing view of `store`")
        # PPN_FORCE_DP=1: run the exchange even in a one-rank group (a SUM over one rank is the identity) -- how the
        # RCCL code path (async bucketed all-reduce against side-stream weight gradients) is exercised on a 1-GPU box
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _force_dp())
        self.world = dist.get_world_size(group) if self.enabled else 1
        n = flat.numel()
        self.bounds = list(range(n, 0, -bucket_elems)) + [0]          # n = b0 > b1 > ... > 0
        self.next = 0                                                  # bucket [bounds[next+1], bounds[next]) is next
        self.handles = []

    def ready(self, offset: int, before_issue=None):
        """Everything at index >= offset is final.  `before_issue()` (optional) is called once before a bucket is
        issued -- e.g. to make the current stream wait for side streams that also wrote gradients."""
        if not self.enabled:
            return
        import torch.distributed as dist
        called = False
        while self.next + 1 < len(self.bounds) and self.bounds[self.next + 1] >= offset:
            if before_issue is not None and not called:
                before_issue()
                called = True
            lo, hi = self.bounds[self.next + 1], self.bounds[self.next]
            buf = self.flat[lo:hi]
            if lo == 0 and self.store is not None and self.before_last is not None:
                self.before_last()                                    # e.g. the task weights' local step into the prefix
                buf = self.store[:self.prefix + hi]                   # prefix + first bucket: one message
            self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.next += 1

    # Exposed (non-overlapped) exchange time: BucketedAllReduce.measure = True makes finish() bracket its waits with two
    # timing events on the current stream -- the time that stream sits idle until the last bucket has arrived, i.e. what the
    # all-reduce adds to the step beyond what ran under the backward (bench.py --workload train --gpus N reports the mean).
    measure = False
    exposed_events: list = []

    @classmethod
    def exposed_ms(cls, reset: bool = True):
        """Per finish() call since the last reset: milliseconds the issuing stream waited for the exchange."""
        out = [a.elapsed_time(b) for a, b in cls.exposed_events]
        if reset:
            cls.exposed_events = []
        return out

    def finish(self, before_issue=None) -> float:
        self.ready(0, before_issue)
        timed = self.enabled and BucketedAllReduce.measure and self.flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self.handles:
            h.wait()
        if timed:
            e1.record()
            BucketedAllReduce.exposed_events.append((e0, e1))
        self.handles, self.next = [], 0
        return 1.0 / self.world


# ---- second-order building blocks (GradNorm's Lgrad.backward(), main.py:759) --------------------------------------

def _bwd_desc(x, dy, gamma, beta, saved, act, dx):
    c = x.shape[-1]
    d = L.BnBwdDesc()
    d.dtype, d.channels, d.pixels, d.act = _dtype_code(x), c, x.numel() // c, ACT[act]
    d.x, d.dy, d.dx = x.data_ptr(), dy.data_ptr(), dx.data_ptr()
    d.gamma, d.beta = _f32(gamma, c, "gamma"), _f32(beta, c, "beta")
    d.save_mean, d.save_rstd = saved.mean.data_ptr(), saved.rstd.data_ptr()
    return d


def bn_tangent(x, xdot, gamma, beta, saved: BnSaved, act: str, out: Optional[torch.Tensor] = None,
               nstreams: int = 1) -> torch.Tensor:
    """Forward-mode image of y = act(bn_train(x)) for an input tangent xdot: act'(z) * gamma*rstd*P(xdot).
    nstreams > 1: xdot / out hold that many tangents of the same x stacked along dim 0 (4 launches for all of them)."""
    lib = L.load()
    jvp, _, _ = bn_train_backward(x, xdot, gamma, beta, saved, act="none", nstreams=nstreams)   # the BN tangent IS the backward formula
    if out is None:
        out = torch.empty_like(xdot)
    _stacked(x, out, nstreams, "out")
    d = _bwd_desc(x, jvp, gamma, beta, saved, act, out)
    if nstreams == 1:
        L.check(lib.ppn_bn_act_mask(C.byref(d), L.current_stream_ptr()), "ppn_bn_act_mask")
    else:
        L.check(lib.ppn_bn_act_mask_streams(C.byref(d), nstreams, L.current_stream_ptr()), "ppn_bn_act_mask_streams")
    return out


_dual_ws = {}


def bn_dual_backward(x, xdot, dy, dyt, gamma, beta, saved: BnSaved, act: str, out_dx=None, out_dxdot=None,
                     nstreams: int = 1, dy_dyt: Optional[torch.Tensor] = None, out_both: Optional[torch.Tensor] = None):
    """Adjoint of the pair (y, ydot) = (act(bn(x)), bn_tangent(x, xdot)) for adjoints (dy, dyt).
    Returns (dx, dxdot, dgamma, dbeta): the ordinary backward of dy plus the tangent stream's contributions.
    nstreams > 1: xdot, dy, dyt (and the outputs) hold that many streams over the same x stacked along dim 0; dgamma /
    dbeta come back as [nstreams, C].  dy_dyt: ONE tensor [dy streams | dyt streams] (2 * nstreams stacked; dy / dyt are
    then ignored) -- with out_both of the same layout the two ordinary backward passes are a single set of launches."""
    lib = L.load()
    c = x.shape[-1]
    B = x.shape[0]
    if dy_dyt is not None:
        _stacked(x, dy_dyt, 2 * nstreams, "dy_dyt")
        both, dgb, dbb = bn_train_backward(x, dy_dyt, gamma, beta, saved, act=act, out=out_both, nstreams=2 * nstreams)
        dx, dxdot = both[:nstreams * B], both[nstreams * B:]
        dyt = dy_dyt[nstreams * B:]
        dgamma, dbeta = dgb.view(2 * nstreams, c)[:nstreams], dbb.view(2 * nstreams, c)[:nstreams]
    else:
        dx, dgamma, dbeta = bn_train_backward(x, dy, gamma, beta, saved, act=act, out=out_dx, nstreams=nstreams)
        dxdot, _, _ = bn_train_backward(x, dyt, gamma, beta, saved, act=act, out=out_dxdot, nstreams=nstreams)   # (gamma*rstd) P(dyt * act')
    _stacked(x, xdot, nstreams, "xdot")
    key = (c, str(x.device), torch.cuda.current_stream(x.device).cuda_stream, nstreams)
    ws = _dual_ws.get(key)
    if ws is None:
        ws = _dual_ws[key] = torch.empty(nstreams * lib.ppn_bn_dual_workspace_bytes(c), dtype=torch.uint8, device=x.device)
    dg_tan = torch.empty(nstreams * c, dtype=torch.float32, device=x.device)
    d = _bwd_desc(x, dyt, gamma, beta, saved, act, dx)
    d.workspace = ws.data_ptr()
    if nstreams == 1:
        L.check(lib.ppn_bn_dual_bwd(C.byref(d), xdot.data_ptr(), dg_tan.data_ptr(), L.current_stream_ptr()),
                "ppn_bn_dual_bwd")
    else:
        L.check(lib.ppn_bn_dual_bwd_streams(C.byref(d), xdot.data_ptr(), dg_tan.data_ptr(), nstreams,
                                            L.current_stream_ptr()), "ppn_bn_dual_bwd_streams")
        dg_tan = dg_tan.view(nstreams, c)
    return dx, dxdot, dgamma + dg_tan, dbeta


def bn_dual_backward_summed(x, xdot, dysum_dyt, gamma, beta, saved: BnSaved, act: str, nstreams: int,
                            out_both: Optional[torch.Tensor] = None):
    """bn_dual_backward for `nstreams` streams when only the SUM over the streams of the adjoint at x is needed (it is: the
    tail adds the streams' adjoints).  The ordinary backward is linear in dy, so the caller passes the primal adjoints
    ALREADY SUMMED: dysum_dyt = [sum_i dy_i | dyt_0 .. dyt_{n-1}] (1 + nstreams tensors of x's shape stacked along dim 0).
    Returns (dx_sum [x's shape] = BNbwd(sum dy) + sum_i dual_i, dxdot [nstreams stacked], dgamma f32[C], dbeta f32[C]).
    One ordinary backward over 1 + n streams, one dual reduce / finalize, n accumulating dual applies (stream order)."""
    lib = L.load()
    c, B = x.shape[-1], x.shape[0]
    _stacked(x, dysum_dyt, 1 + nstreams, "dysum_dyt")
    _stacked(x, xdot, nstreams, "xdot")
    both, dgb, dbb = bn_train_backward(x, dysum_dyt, gamma, beta, saved, act=act, out=out_both, nstreams=1 + nstreams)
    dx, dxdot = both[:B], both[B:]
    key = (c, str(x.device), torch.cuda.current_stream(x.device).cuda_stream, nstreams)
    ws = _dual_ws.get(key)
    if ws is None:
        ws = _dual_ws[key] = torch.empty(nstreams * lib.ppn_bn_dual_workspace_bytes(c), dtype=torch.uint8, device=x.device)
    dg_tan = torch.empty(nstreams * c, dtype=torch.float32, device=x.device)
    d = _bwd_desc(x, dysum_dyt[B:], gamma, beta, saved, act, dx)
    d.workspace = ws.data_ptr()
    L.check(lib.ppn_bn_dual_bwd_streams_sum(C.byref(d), xdot.data_ptr(), dg_tan.data_ptr(), nstreams,
                                            L.current_stream_ptr()), "ppn_bn_dual_bwd_streams_sum")
    dgb, dbb = dgb.view(1 + nstreams, c), dbb.view(1 + nstreams, c)
    return dx, dxdot, dgb[0] + dg_tan.view(nstreams, c).sum(0), dbb[0]


def nchw_to_nhwc(src: torch.Tensor, dtype: torch.dtype, channels_used=None) -> torch.Tensor:
    """f32 [B,C,H,W] -> `dtype` [B,H,W,Cpad] (Cpad = channels_used rounded up to 64, padding zeroed)."""
    B, Cn, H, W = src.shape
    used = Cn if channels_used is None else channels_used
    cpad = (used + 63) // 64 * 64
    out = torch.empty(B, H, W, cpad, dtype=dtype, device=src.device)
    L.check(L.load().ppn_nchw_to_nhwc(L.PPN_F32 if dtype == torch.float32 else L.PPN_BF16, src.data_ptr(), B, Cn,
                                      H * W, used, cpad, out.data_ptr(), L.current_stream_ptr()), "ppn_nchw_to_nhwc")
    return out


def add_relu(z: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """relu(z + r): the tail of a Bottleneck (drn.py:92-95)."""
    out = torch.empty_like(z)
    L.check(L.load().ppn_add_relu(_dtype_code(z), z.data_ptr(), r.data_ptr(), z.numel(), out.data_ptr(),
                                  L.current_stream_ptr()), "ppn_add_relu")
    return out


def relu_mask(out: torch.Tensor, dout: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dout * (out > 0) (+ add): backward of out = relu(.)."""
    dz = torch.empty_like(out)
    L.check(L.load().ppn_relu_mask(_dtype_code(out), out.data_ptr(), dout.data_ptr(),
                                   add.data_ptr() if add is not None else None, out.numel(), dz.data_ptr(),
                                   L.current_stream_ptr()), "ppn_relu_mask")
    return dz
