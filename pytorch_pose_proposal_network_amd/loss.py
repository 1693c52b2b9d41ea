"""``PPNLoss`` with the reference's surface (main.py:147-216) over the fused HIP loss kernels.

    criterion = PPNLoss(insize=(384, 384), outsize=(24, 24), local_grid_size=(21, 21))
    l_resp, l_iou, l_coor, l_size, l_limb = criterion(image, feature_map, delta, weight, weight_ij,
                                                      tx_half, ty_half, tx, ty, tw, th, te)   # main.py:180, 665-666
    losses, grad = criterion.forward_backward(feature_map, targets, coeff=w / 5)              # main.py:668-683

There is no autograd graph here: ``forward_backward`` returns d(sum_i coeff_i L_i)/d(feature_map) computed by the
same pass that evaluates the losses (the reference gets it from ``loss.backward()``).  With a one-hot ``coeff``
it yields the per-loss gradient that seeds the GradNorm partial backward passes (main.py:704-708).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch

from . import config as cfg
from . import lib as L

_LIMB_COMPACT = os.environ.get("PPN_LOSS_COMPACT_TARGETS", "1") != "0"      # A/B switch: read te / weight_ij as f32
_CHECK_LIMB_C = os.environ.get("PPN_CHECK_LIMB_C", "0") == "1"

TARGET_KEYS = ("delta", "weight", "weight_ij", "tx_half", "ty_half", "tx", "ty", "tw", "th", "te")


class PPNLoss:
    def __init__(self, insize=(384, 384), outsize=(24, 24), keypoint_names=cfg.KEYPOINT_NAMES,
                 local_grid_size=(21, 21), edges=cfg.EDGES):
        self.insize, self.outsize = insize, outsize
        self.keypoint_names, self.edges, self.local_grid_size = keypoint_names, edges, local_grid_size
        inW, inH = insize
        outW, outH = outsize
        self.gridsize = (int(inW / outW), int(inH / outH))
        c = L.LossCfg()
        c.K, c.E = len(keypoint_names), len(edges)
        c.sW, c.sH = local_grid_size
        c.W, c.H = outsize
        c.inW, c.inH = insize
        self._cfg = c
        self._lib = None
        self._ws: Dict[int, torch.Tensor] = {}

    def _check(self, name, t, shape):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
            raise ValueError(f"{name} must be a float32 CUDA tensor")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
        return t.contiguous()       # the reference's torch ops accept any strides; the kernels need dense tensors

    def forward_backward(self, feature_map: torch.Tensor, targets: Dict[str, torch.Tensor],
                         coeff: Optional[Sequence[float]] = None, want_grad: bool = True, coeff_dev=None):
        """-> (losses f32[5] on the device, grad like feature_map or None).
        coeff_dev = (f32[5] device tensor, divisor): the coefficients are read on the device when the kernels run
        (c_i = tensor[i] / divisor) instead of being passed by value -- no host read-back of the task weights."""
        lib = self._lib = self._lib or L.load()
        c = self._cfg
        B = feature_map.shape[0]
        C_ = 6 * c.K + c.E * c.sH * c.sW
        feature_map = self._check("feature_map", feature_map, (B, C_, c.H, c.W))
        t = {}
        for k in TARGET_KEYS:
            shape = (B, c.E, c.sH, c.sW, c.H, c.W) if k in ("weight_ij", "te") else (B, c.K, c.H, c.W)
            t[k] = self._check(k, targets[k], shape)
        dev = feature_map.device
        ws = self._ws.get(B)
        if ws is None or ws.device != dev:
            n = lib.ppn_loss_workspace_bytes(C.byref(c), B)
            ws = self._ws[B] = torch.empty(max(n, 16) // 4, dtype=torch.float32, device=dev)
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        grad = torch.empty_like(feature_map) if want_grad else None
        cf = None
        if want_grad and coeff_dev is not None:
            cw, div = coeff_dev
            if not (cw.is_cuda and cw.dtype == torch.float32 and cw.numel() == 5 and cw.is_contiguous()):
                raise ValueError("coeff_dev[0] must be a contiguous float32 CUDA tensor of 5 elements")
            L.check(lib.ppn_loss_fwd_bwd_dev(C.byref(c), feature_map.data_ptr(), B, t["delta"].data_ptr(),
                                             t["weight"].data_ptr(), t["weight_ij"].data_ptr(), t["tx_half"].data_ptr(),
                                             t["ty_half"].data_ptr(), t["tx"].data_ptr(), t["ty"].data_ptr(),
                                             t["tw"].data_ptr(), t["th"].data_ptr(), t["te"].data_ptr(),
                                             cw.data_ptr(), float(div), losses.data_ptr(), grad.data_ptr(),
                                             ws.data_ptr(), L.current_stream_ptr()), "ppn_loss_fwd_bwd_dev")
            return losses, grad
        if want_grad:
            if coeff is None:
                raise ValueError("coeff (5 floats) is required for the backward pass")
            cf = (C.c_float * 5)(*[float(v) for v in (coeff.tolist() if isinstance(coeff, torch.Tensor) else coeff)])
        L.check(lib.ppn_loss_fwd_bwd(C.byref(c), feature_map.data_ptr(), B, t["delta"].data_ptr(),
                                     t["weight"].data_ptr(), t["weight_ij"].data_ptr(), t["tx_half"].data_ptr(),
                                     t["ty_half"].data_ptr(), t["tx"].data_ptr(), t["ty"].data_ptr(),
                                     t["tw"].data_ptr(), t["th"].data_ptr(), t["te"].data_ptr(), cf,
                                     losses.data_ptr(), grad.data_ptr() if grad is not None else None, ws.data_ptr(),
                                     L.current_stream_ptr()), "ppn_loss_fwd_bwd")
        return losses, grad

    def unary_backward(self, feature_map: torch.Tensor, targets: Dict[str, torch.Tensor], coeff4: Sequence[float],
                       out: torch.Tensor) -> torch.Tensor:
        """d(sum_{i<4} coeff4_i L_i)/d(feature_map[:, :6K]) written into `out` (a tensor like feature_map; its limb
        channels are left untouched).  The cheap seed of the GradNorm probe passes for losses 0..3."""
        lib = self._lib = self._lib or L.load()
        c = self._cfg
        B = feature_map.shape[0]
        C_ = 6 * c.K + c.E * c.sH * c.sW
        feature_map = self._check("feature_map", feature_map, (B, C_, c.H, c.W))
        if out.shape != feature_map.shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 tensor like feature_map")
        t = {k: self._check(k, targets[k], (B, c.K, c.H, c.W)) for k in TARGET_KEYS if k not in ("weight_ij", "te")}
        ws = self._ws.get(B)
        if ws is None or ws.device != feature_map.device:
            n = lib.ppn_loss_workspace_bytes(C.byref(c), B)
            ws = self._ws[B] = torch.empty(max(n, 16) // 4, dtype=torch.float32, device=feature_map.device)
        cf = (C.c_float * 4)(*[float(v) for v in coeff4])
        L.check(lib.ppn_loss_unary_bwd(C.byref(c), feature_map.data_ptr(), B, t["delta"].data_ptr(),
                                       t["weight"].data_ptr(), t["tx_half"].data_ptr(), t["ty_half"].data_ptr(),
                                       t["tx"].data_ptr(), t["ty"].data_ptr(), t["tw"].data_ptr(), t["th"].data_ptr(),
                                       cf, out.data_ptr(), ws.data_ptr(), L.current_stream_ptr()),
                "ppn_loss_unary_bwd")
        return out

    def dual(self, feature_map: torch.Tensor, tz: torch.Tensor, targets: Dict[str, torch.Tensor],
             coeff: Sequence[float], unary_only: bool = False, out=None):
        """Second-order seeds in head space (GradNorm's Lgrad.backward(), main.py:759): for the logit tangent `tz`
        (head layout) returns (zbar, tzbar), the adjoints of the logits and of their tangents of
        F = <d(sum c_i L_i)/ds, sig'(z)*tz>  (see ppn_loss_dual).  unary_only: tz and the results are compact
        [B,6K,H,W] tensors (the four unary losses only touch those channels)."""
        lib = self._lib = self._lib or L.load()
        c = self._cfg
        B = feature_map.shape[0]
        C_ = 6 * c.K + c.E * c.sH * c.sW
        feature_map = self._check("feature_map", feature_map, (B, C_, c.H, c.W))
        Cd = 6 * c.K if unary_only else C_                    # unary passes work on compact [B,6K,H,W] dual tensors
        tz = self._check("tz", tz, (B, Cd, c.H, c.W))
        t = {}
        for k in TARGET_KEYS:
            if unary_only and k in ("weight_ij", "te"):
                continue
            shape = (B, c.E, c.sH, c.sW, c.H, c.W) if k in ("weight_ij", "te") else (B, c.K, c.H, c.W)
            t[k] = self._check(k, targets[k], shape)
        zbar, tzbar = out if out is not None else (torch.empty_like(tz), torch.empty_like(tz))
        cf = (C.c_float * 5)(*[float(v) for v in coeff])
        if unary_only and float(coeff[4]) != 0.0:
            raise ValueError("unary_only needs coeff[4] == 0")
        L.check(lib.ppn_loss_dual(C.byref(c), feature_map.data_ptr(), tz.data_ptr(), B, t["delta"].data_ptr(),
                                  t["weight"].data_ptr(), t["weight_ij"].data_ptr() if "weight_ij" in t else None,
                                  t["tx_half"].data_ptr(), t["ty_half"].data_ptr(), t["tx"].data_ptr(),
                                  t["ty"].data_ptr(), t["tw"].data_ptr(), t["th"].data_ptr(),
                                  t["te"].data_ptr() if "te" in t else None, cf, 1 if unary_only else 0,
                                  zbar.data_ptr(), tzbar.data_ptr(), L.current_stream_ptr()), "ppn_loss_dual")
        return zbar, tzbar

    def forward_backward_dz(self, feature_map: torch.Tensor, targets: Dict[str, torch.Tensor], coeff_dev, dtype: torch.dtype):
        """Losses + the gradient w.r.t. conv3's logits in NHWC (ppn_loss_fwd_bwd_dz): -> (losses f32[5], dz `dtype`
        [B,H,W,cpad], dbsum f32 [B*ceil(HW/64), cpad] whose column sums are d loss / d conv3.bias).  Equals
        forward_backward(coeff_dev=...) followed by ppn_head_grad, without the f32 head-layout gradient."""
        lib = self._lib = self._lib or L.load()
        c = self._cfg
        B = feature_map.shape[0]
        C_ = 6 * c.K + c.E * c.sH * c.sW
        feature_map = self._check("feature_map", feature_map, (B, C_, c.H, c.W))
        t = {}
        for k in TARGET_KEYS:
            shape = (B, c.E, c.sH, c.sW, c.H, c.W) if k in ("weight_ij", "te") else (B, c.K, c.H, c.W)
            t[k] = self._check(k, targets[k], shape)
        cw, div = coeff_dev
        if not (cw.is_cuda and cw.dtype == torch.float32 and cw.numel() == 5 and cw.is_contiguous()):
            raise ValueError("coeff_dev[0] must be a contiguous float32 CUDA tensor of 5 elements")
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be float32 or bfloat16")
        dev = feature_map.device
        cpad = (C_ + 63) // 64 * 64
        key = ("dz", B, cpad)
        ws = self._ws.get(key)
        if ws is None or ws.device != dev:
            n = lib.ppn_loss_dz_workspace_bytes(C.byref(c), B, cpad)
            ws = self._ws[key] = torch.empty(max(n, 16) // 4, dtype=torch.float32, device=dev)
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        gu = torch.empty(B, 6 * c.K, c.H, c.W, dtype=torch.float32, device=dev)
        dz = torch.empty(B, c.H, c.W, cpad, dtype=dtype, device=dev)
        dbsum = torch.empty(B * ((c.H * c.W + 63) // 64), cpad, dtype=torch.float32, device=dev)
        lc = self._limb_compact(targets, B)
        dcode = L.PPN_F32 if dtype == torch.float32 else L.PPN_BF16
        if lc is not None:
            L.check(lib.ppn_loss_fwd_bwd_dz_c(C.byref(c), feature_map.data_ptr(), B, t["delta"].data_ptr(),
                                              t["weight"].data_ptr(), lc.data_ptr(), t["tx_half"].data_ptr(),
                                              t["ty_half"].data_ptr(), t["tx"].data_ptr(), t["ty"].data_ptr(),
                                              t["tw"].data_ptr(), t["th"].data_ptr(), cw.data_ptr(), float(div),
                                              losses.data_ptr(), gu.data_ptr(), dcode, cpad, dz.data_ptr(),
                                              dbsum.data_ptr(), ws.data_ptr(), L.current_stream_ptr()),
                    "ppn_loss_fwd_bwd_dz_c")
            return losses, dz, dbsum
        L.check(lib.ppn_loss_fwd_bwd_dz(C.byref(c), feature_map.data_ptr(), B, t["delta"].data_ptr(),
                                        t["weight"].data_ptr(), t["weight_ij"].data_ptr(), t["tx_half"].data_ptr(),
                                        t["ty_half"].data_ptr(), t["tx"].data_ptr(), t["ty"].data_ptr(),
                                        t["tw"].data_ptr(), t["th"].data_ptr(), t["te"].data_ptr(), cw.data_ptr(),
                                        float(div), losses.data_ptr(), gu.data_ptr(), dcode, cpad, dz.data_ptr(),
                                        dbsum.data_ptr(), ws.data_ptr(), L.current_stream_ptr()), "ppn_loss_fwd_bwd_dz")
        return losses, dz, dbsum

    def _limb_compact(self, targets, B):
        """targets["limb_c"] (u8, te's shape; targets.encode_targets writes it) when present and enabled, else None."""
        lc = targets.get("limb_c") if _LIMB_COMPACT else None
        if lc is None:
            return None
        c = self._cfg
        if (lc.dtype != torch.uint8 or tuple(lc.shape) != (B, c.E, c.sH, c.sW, c.H, c.W) or not lc.is_cuda or
                not lc.is_contiguous()):
            raise ValueError("targets['limb_c'] must be a contiguous uint8 CUDA tensor of te's shape")
        # `limb_c` is DERIVED from te / weight_ij (bit 0: te == 1, bit 1: weight_ij == 1; targets.encode_targets writes all three)
        # and must be dropped from the dict whenever either is edited or replaced: the kernels that stream it never look at the
        # f32 maps, the GradNorm probes and the untrusted fallback do.  PPN_CHECK_LIMB_C=1 verifies the agreement on every call
        # (two full-size comparisons: a debugging aid, not for timed runs).
        if _CHECK_LIMB_C:
            te, wij = targets["te"], targets["weight_ij"]
            want = (te == 1).to(torch.uint8) | ((wij == 1).to(torch.uint8) << 1)
            if not torch.equal(want, lc) or bool(((te != 0) & (te != 1)).any()):
                raise ValueError("targets['limb_c'] does not agree with targets['te'] / ['weight_ij'] (edited after "
                                 "encode_targets? drop 'limb_c' from the dict)")
        return lc

    def limb_dual_nhwc(self, feature_map: torch.Tensor, tz: torch.Tensor, targets: Dict[str, torch.Tensor], c4: float,
                       dtype: torch.dtype):
        """The limb stream of dual() -- coefficients (0,0,0,0,c4) -- with NHWC outputs: (zb, tzb `dtype` [B,H,W,cpad],
        zsum f32 [B*ceil(HW/64), cpad] whose column sums are the per-channel sums of zbar).  Equals dual() followed by
        train.nchw_to_nhwc on both results, without the two f32 head-layout intermediates."""
        lib = self._lib = self._lib or L.load()
        c = self._cfg
        B = feature_map.shape[0]
        C_ = 6 * c.K + c.E * c.sH * c.sW
        feature_map = self._check("feature_map", feature_map, (B, C_, c.H, c.W))
        tz = self._check("tz", tz, (B, C_, c.H, c.W))
        wij = self._check("weight_ij", targets["weight_ij"], (B, c.E, c.sH, c.sW, c.H, c.W))
        te = self._check("te", targets["te"], (B, c.E, c.sH, c.sW, c.H, c.W))
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be float32 or bfloat16")
        cpad = (C_ + 63) // 64 * 64
        dev = feature_map.device
        zb = torch.empty(B, c.H, c.W, cpad, dtype=dtype, device=dev)
        tzb = torch.empty_like(zb)
        npy = (c.H * c.W + 63) // 64
        zsum = torch.empty(B * npy, cpad, dtype=torch.float32, device=dev)
        lc = self._limb_compact(targets, B)
        dcode = L.PPN_F32 if dtype == torch.float32 else L.PPN_BF16
        if lc is not None:
            L.check(lib.ppn_loss_limb_dual_nhwc_c(C.byref(c), feature_map.data_ptr(), tz.data_ptr(), B, lc.data_ptr(),
                                                  float(c4), dcode, cpad, zb.data_ptr(), tzb.data_ptr(), zsum.data_ptr(),
                                                  L.current_stream_ptr()), "ppn_loss_limb_dual_nhwc_c")
            return zb, tzb, zsum
        L.check(lib.ppn_loss_limb_dual_nhwc(C.byref(c), feature_map.data_ptr(), tz.data_ptr(), B, wij.data_ptr(),
                                            te.data_ptr(), float(c4), dcode, cpad, zb.data_ptr(), tzb.data_ptr(),
                                            zsum.data_ptr(), L.current_stream_ptr()), "ppn_loss_limb_dual_nhwc")
        return zb, tzb, zsum

    def forward(self, image, feature_map, delta, weight, weight_ij, tx_half, ty_half, tx, ty, tw, th, te):
        """Reference signature (main.py:180); `image` is only used for its batch size there and is ignored here."""
        targets = dict(delta=delta, weight=weight, weight_ij=weight_ij, tx_half=tx_half, ty_half=ty_half, tx=tx, ty=ty,
                       tw=tw, th=th, te=te)
        losses, _ = self.forward_backward(feature_map, targets, want_grad=False)
        return tuple(losses[i] for i in range(5))

    __call__ = forward
