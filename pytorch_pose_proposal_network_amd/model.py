"""``PoseProposalNet`` with the reference's surface (model.py:51-136) over the HIP conv stack.

    backbone = drn.drn_d_22()                       # spec object, mirrors rt_test.py:56-61
    model = PoseProposalNet(backbone, local_grid_size=(21, 21)).cuda()
    model.load_state_dict(checkpoint['state_dict']) # reference names
    model.eval()
    head = model(image)                             # f32 [B,3,S,S] cuda -> f32 [B,7605,S/16,S/16]

``forward`` launches one fused HIP kernel per convolution through libppn.so (ctypes, C ABI); PyTorch
is used for device memory and streams only.  ``compute_dtype='float32'`` is the exact-f32 MFMA parity
mode (1e-4 on the head), ``'bfloat16'`` the performance mode (bf16 operands, f32 accumulation, f32 head).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import arch as A
from . import config as cfg
from . import lib as L


class DRNSpec:
    """What ``drn.drn_d_22()`` returns here: the architecture name (there are no nn.Modules to hold)."""

    def __init__(self, arch: str):
        if arch not in A.DRN_D:
            raise ValueError(f"unknown DRN-D variant {arch!r}; have {sorted(A.DRN_D)}")
        self.arch = arch

    def children(self):            # `nn.Sequential(*list(model.children())[:-2])` keeps working
        return [self, None, None]

    def __repr__(self):
        return f"DRNSpec({self.arch})"


def _arch_of(backbone) -> str:
    if isinstance(backbone, str):
        return backbone
    if isinstance(backbone, DRNSpec):
        return backbone.arch
    if isinstance(backbone, (list, tuple)) and backbone and isinstance(backbone[0], DRNSpec):
        return backbone[0].arch
    raise TypeError("backbone must be a DRN-D name or the object returned by drn.drn_d_*()")


class _Plan:
    def __init__(self, handle, buffers, head, entries, flops, input):
        self.handle, self.buffers, self.head, self.flops = handle, buffers, head, flops
        self.entries = entries          # [(name, flops)] aligned with the plan's launches
        self.n_ops = len(entries)
        self.input = input              # plan-owned input buffer: the captured hipGraph never depends on a caller's pointer


class PoseProposalNet:
    def __init__(self, backbone="drn_d_22", insize=(384, 384), outsize=(24, 24),
                 keypoint_names=cfg.KEYPOINT_NAMES, local_grid_size=(21, 21), edges=cfg.EDGES,
                 compute_dtype: str = "float32", fuse_stem=None, fuse_shortcut: Optional[bool] = None,
                 stem_dtype: Optional[str] = None, half_prefix: Optional[int] = None, exact_prefix: int = -1,
                 fuse_block: Optional[bool] = None):
        self.arch = _arch_of(backbone)
        self.insize = insize
        self.outsize = outsize
        self.keypoint_names = keypoint_names
        self.edges = edges
        self.local_grid_size = local_grid_size
        inW, inH = insize
        outW, outH = outsize
        sW, sH = local_grid_size
        self.gridsize = (int(inW / outW), int(inH / outH))
        self.lastsize = 6 * len(keypoint_names) + sW * sH * len(edges)          # model.py:64
        self.compute_dtype = {"float32": L.PPN_F32, "fp32": L.PPN_F32, "bfloat16": L.PPN_BF16,
                              "bf16": L.PPN_BF16, "float16": L.PPN_F16, "fp16": L.PPN_F16, "f16": L.PPN_F16,
                              "float16x3": L.PPN_F16X3, "f16x3": L.PPN_F16X3}[compute_dtype]
        self.training = False
        # 64-channel stride-1 BasicBlocks (layer3 behind its first block) as one launch each (csrc/block64.hip; PPN_BLOCK64=0 /
        # fuse_block=False keep the two launches; results are bit-identical)
        self.fuse_block = (os.environ.get("PPN_BLOCK64", "1") != "0") if fuse_block is None else bool(fuse_block)
        self.device = torch.device("cuda")
        # Type the FUSED stem (csrc/stem012.hip, 16-bit modes) computes in: its MFMA operands and on-chip tensors; its two
        # output tensors are always stored in the trunk's type.  The bf16 mode defaults to IEEE half (round 4): the stem
        # is 1.8 % of the FLOPs, but its rounding noise passes through every layer behind it -- half internals take the
        # bf16 pipeline from 95 to ~150 of the reference's 260 people at the same speed (PPN_STEM_DTYPE=bfloat16 / the
        # argument restore the all-bf16 stem).
        sdt = stem_dtype or (os.environ.get("PPN_STEM_DTYPE") if self.compute_dtype == L.PPN_BF16 else None) or \
            ("float16" if self.compute_dtype in (L.PPN_BF16, L.PPN_F16) else None)
        self.stem_dtype = {None: None, "float16": L.PPN_F16, "fp16": L.PPN_F16, "f16": L.PPN_F16, "bfloat16": L.PPN_BF16,
                           "bf16": L.PPN_BF16}[sdt]
        if self.compute_dtype == L.PPN_F16 and self.stem_dtype == L.PPN_BF16:
            raise ValueError("the float16 mode has no bfloat16 stem")
        # bf16 mode, round 4: the launches of backbone.0 .. backbone.{half_prefix} (default 4: stem + layer3 + layer4 = 6.9 % of
        # DRN-D-22's FLOPs) run in IEEE half -- the same kernels at the same rate -- and the last of them stores its outputs
        # as bf16 (PPN_CONV_OUT_BF16).  Rounding noise injected in the first layers is amplified by every layer behind
        # them: with this prefix the bf16 pipeline reproduces ~200 instead of 95 of the reference's 260 people
        # (tests/precision_study_mixed.py; measured numbers in DESIGN.md section 2).  half_prefix=-1 / PPN_BF16_HALF_PREFIX=-1:
        # pure bf16 (with stem_dtype="bfloat16": the round-3 behaviour).
        explicit = half_prefix is not None
        if half_prefix is None:
            half_prefix = int(os.environ.get("PPN_BF16_HALF_PREFIX", "4"))
        self.half_prefix = half_prefix if self.compute_dtype == L.PPN_BF16 else -1
        if self.half_prefix >= 3 and self.stem_dtype == L.PPN_BF16:
            if explicit:
                raise ValueError("half_prefix >= 3 needs the IEEE-half stem (stem_dtype='float16')")
            self.half_prefix = -1                              # an all-bf16 stem was asked for: pure bf16
        self._half_names = tuple(f"backbone.{i}." for i in range(self.half_prefix + 1)) if self.half_prefix >= 3 else ()
        # float16 mode with an EXACT prefix (round 4): the launches of backbone.0 .. backbone.{exact_prefix} run as in the
        # float16x3 mode (f32 where cin < 64, split-f16 elsewhere) and the last of them stores plain half for the f16 trunk
        # (PPN_CONV_X3_PLAIN_OUT).  exact_prefix=3 (stem + layer3, 4.3 % of the FLOPs): 251 of the reference's 260 people
        # where the plain f16 mode reproduces 233 (emulated: tests/precision_study_mixed.py).
        self.exact_prefix = exact_prefix if self.compute_dtype == L.PPN_F16 else -1
        if exact_prefix >= 0 and (self.compute_dtype != L.PPN_F16 or exact_prefix < 3):
            raise ValueError("exact_prefix is an option of the float16 mode and covers at least backbone.0 .. backbone.3")
        self._exact_names = tuple(f"backbone.{i}." for i in range(self.exact_prefix + 1)) if self.exact_prefix >= 3 else ()
        if self._exact_names:
            if fuse_stem not in (None, False, True):
                raise ValueError("an exact prefix runs the stem as f32 launches (fuse_stem False / True)")
            fuse_stem = bool(fuse_stem)
            if fuse_shortcut is None:
                fuse_shortcut = os.environ.get("PPN_FUSE_SHORTCUT", "1") != "0"
            if fuse_shortcut and not callable(fuse_shortcut):
                names = self._exact_names
                fuse_shortcut = lambda prefix: not (prefix + ".").startswith(names)        # noqa: E731  (no fused shortcut in split launches)
        if fuse_stem is None:
            # bf16 mode: the three stem layers share one launch (csrc/stem012.hip; PPN_FUSE_STEM=0 keeps them apart);
            # the exact-f32 parity mode runs them layer by layer
            fuse_stem = "all" if (self.compute_dtype == L.PPN_F16 or (
                self.compute_dtype == L.PPN_BF16 and os.environ.get("PPN_FUSE_STEM", "1") != "0")) else False
        if self.compute_dtype == L.PPN_F16X3:
            # split-f16 mode: the layers with cin < 64 (stem, first block's stride-2 convs) run as exact f32, launch by
            # launch; the split kernel has no fused-shortcut instantiation
            if fuse_stem or fuse_shortcut:
                raise ValueError("the float16x3 mode runs the stem layer by layer and without fused shortcuts")
            fuse_stem, fuse_shortcut = False, False
        if fuse_stem == "all" and self.compute_dtype in (L.PPN_F32, L.PPN_F16X3):
            raise ValueError("fuse_stem='all' (csrc/stem012.hip) is a 16-bit-mode kernel")
        if self.compute_dtype == L.PPN_F16 and fuse_stem != "all" and not self._exact_names:
            raise ValueError("the float16 mode runs the stem through csrc/stem012.hip only (fuse_stem='all')")
        if fuse_stem != "all":                                # the half prefix starts with the fused stem's half outputs
            self._half_names, self.half_prefix = (), -1
        if fuse_shortcut is None:                             # tuning knob: PPN_FUSE_SHORTCUT=0 keeps the 1x1 shortcuts apart
            fuse_shortcut = os.environ.get("PPN_FUSE_SHORTCUT", "1") != "0"
        self._ops: List[A.ConvOp] = A.build_program(self.arch, self.lastsize, fuse_stem=fuse_stem,
                                                    fuse_shortcut=fuse_shortcut)
        self._spec = dict(A.param_spec(self.arch, self.lastsize))
        self._sd: Dict[str, torch.Tensor] = {}
        self._dev: Dict[str, torch.Tensor] = {}      # packed weights / folded BN on the device
        self._plans: Dict[tuple, _Plan] = {}
        self._lib = None
        self._trainer = None                          # trainer.PPNTrainer behind train(): same object, both modes
        self._trainer_dirty = False
        self._mean = (C.c_float * 3)(*cfg.MEAN)
        self._std = (C.c_float * 3)(*cfg.STD)

    # ---- nn.Module-like surface -------------------------------------------------------------
    def cuda(self, device=None):
        self.device = torch.device("cuda" if device is None else device)
        return self

    def eval(self):
        return self.train(False)

    def train(self, mode: bool = True):
        """nn.Module.train(): the SAME object serves both modes, as in main.py:643 / rt_test.py:94.

        mode=True: forward() runs the train-mode network (batch statistics, running statistics updated with momentum
        0.1, activations taped) of ``self.trainer`` -- a trainer.PPNTrainer created on first use from this model's
        state_dict; the training loop then calls ``model.trainer.train_step(x, targets)``.
        mode=False: back to the folded-BN inference plan; parameters and running statistics the trainer changed are
        folded again first."""
        if mode and self.compute_dtype in (L.PPN_F16, L.PPN_F16X3):
            raise RuntimeError("PoseProposalNet.train(): the float16 / float16x3 modes are inference only (train in "
                               "bfloat16 / float32)")
        if mode and not self.training:
            if not self._sd:
                raise RuntimeError("PoseProposalNet.train(): call load_state_dict() first")
            if self._trainer is None:
                from .trainer import PPNTrainer
                self._trainer = PPNTrainer(self.arch, self._sd, compute_dtype=self.compute_dtype, insize=self.insize,
                                           device=self.device)
            self._trainer_dirty = True                # train_step / train-mode forwards change parameters and statistics
        if not mode and self.training and self._trainer is not None and self._trainer_dirty:
            self._trainer_dirty = False
            self.load_state_dict(self._trainer.state_dict(), _from_trainer=True)
        self.training = bool(mode)
        return self

    @property
    def trainer(self):
        """The PPNTrainer behind train mode (None before the first train())."""
        return self._trainer

    def state_dict(self):
        if self._trainer is not None and self._trainer_dirty:        # train mode: the trainer holds the live values
            return {k: v.detach().cpu() for k, v in self._trainer.state_dict().items()}
        return dict(self._sd)

    def load_state_dict(self, state_dict, strict: bool = True, _from_trainer: bool = False):
        """Accepts the reference checkpoint's ``state_dict`` (rt_test.py:74-75); ``module.``-prefixed DDP
        checkpoints are stripped as main.py:311-318 does."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("module."):
                k = k[len("module."):]
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
            sd[k] = t.detach().cpu()
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} ({len(missing)}), "
                               f"unexpected {unexpected[:5]} ({len(unexpected)})")
        for k, shape in self._spec.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"load_state_dict: {k} has shape {tuple(sd[k].shape)}, expected {tuple(shape)}")
        self._sd = sd
        self._prepare()
        if self._trainer is not None and not _from_trainer:
            self._trainer.load_state_dict(sd)
        return self

    # ---- weight preparation -------------------------------------------------------------------
    def _fold_bn(self, prefix: str):
        sd = self._sd
        g = sd[prefix + ".weight"].double()
        b = sd[prefix + ".bias"].double()
        m = sd[prefix + ".running_mean"].double()
        v = sd[prefix + ".running_var"].double()
        s = g / torch.sqrt(v + 1e-5)                         # nn.BatchNorm2d eps
        return s, b - m * s

    def _prepare(self):
        lib = self._lib = L.load()
        dev = self.device
        self._dev.clear()
        for p in self._plans.values():
            lib.ppn_plan_destroy(p.handle)
        self._plans.clear()
        self._dev["zero"] = torch.zeros(64, dtype=torch.float32, device=dev)
        stream = L.current_stream_ptr()
        tdt = self._tdt()
        for op in self._ops:
            w = self._sd[op.weight].float().contiguous()
            s1 = b1 = None
            if op.bn1:
                s1, b1 = self._fold_bn(op.bn1)
            if op.bias:
                bias = self._sd[op.bias].double()
                b1 = bias * s1 + b1 if s1 is not None else bias
            if s1 is not None:
                self._dev[op.name + ".s1"] = s1.float().to(dev)
            if b1 is not None:
                self._dev[op.name + ".b1"] = b1.float().to(dev)
            if op.bn2:
                s2, b2 = self._fold_bn(op.bn2)
                self._dev[op.name + ".s2"] = s2.float().to(dev)
                self._dev[op.name + ".b2"] = b2.float().to(dev)
            if op.k == 7:                                     # stem keeps the reference layout in f32
                self._dev[op.name + ".w"] = w.to(dev)
                if op.next3x3 is not None:                     # layer1 fused into the same launch
                    n = op.next3x3
                    sn, bn_ = self._fold_bn(n.bn1)
                    self._dev[op.name + ".w1"] = self._sd[n.weight].float().contiguous().to(dev)
                    self._dev[op.name + ".s1b"] = sn.float().to(dev)
                    self._dev[op.name + ".b1b"] = bn_.float().to(dev)
                if op.next_s2 is not None:                     # ... and layer2
                    m = op.next_s2
                    sm, bm = self._fold_bn(m.bn1)
                    self._dev[op.name + ".w2"] = self._sd[m.weight].float().contiguous().to(dev)
                    self._dev[op.name + ".s1c"] = sm.float().to(dev)
                    self._dev[op.name + ".b1c"] = bm.float().to(dev)
                continue
            odt = self._op_dtype(op)
            if odt == L.PPN_F16X3:
                # split-f16 weights: three half copies of w * 2^s per 64-channel slab (csrc/conv_big.hip, X3); 2^-s is
                # folded into scale1 (a power of two: exact)
                _, _, _, kpad, cpad = L.conv_tiling(odt, op.cin, op.cout, op.k)
                wmax = float(w.abs().max())
                sl2 = int(np.floor(np.log2(32768.0 / wmax))) if wmax > 0 else 0
                sl2 = max(-24, min(24, sl2))
                wd = w.to(dev)
                packed = torch.empty(cpad, 3 * kpad, dtype=torch.float16, device=dev)
                L.check(lib.ppn_pack_weight_x3(wd.data_ptr(), op.cout, op.cin, op.k, cpad, sl2, packed.data_ptr(), stream),
                        "ppn_pack_weight_x3")
                base = s1 if s1 is not None else torch.ones(op.cout, dtype=torch.float64)
                self._dev[op.name + ".s1"] = (base * (2.0 ** -sl2)).float().to(dev)
                self._dev[op.name + ".w"] = packed
                self._dev[op.name + ".geom"] = (3 * kpad, cpad)
                torch.cuda.synchronize(dev)
                continue
            kstep, _, korder, ktot, cpad = L.conv_tiling(odt, op.cin, op.cout, op.k)
            wd = w.to(dev)
            tdt = self._tdt(odt)
            packed = torch.empty(cpad, ktot, dtype=torch.float32 if korder == 2 else tdt, device=dev)
            L.check(lib.ppn_pack_weight(odt, wd.data_ptr(), op.cout, op.cin, op.k, cpad, ktot,
                                        korder, kstep, packed.data_ptr(), stream), "ppn_pack_weight")
            if op.ds_src:
                # fused projection shortcut: [main | 1x1 weights * BN scale] per packed row, BN shift -> shift1
                assert korder == 1 and s1 is None and op.ds_cin % kstep == 0
                sds, bds = self._fold_bn(op.ds_bn)
                wds = (self._sd[op.ds_weight].double() * sds.view(-1, 1, 1, 1)).float().contiguous().to(dev)
                pds = torch.empty(cpad, op.ds_cin, dtype=tdt, device=dev)
                L.check(lib.ppn_pack_weight(odt, wds.data_ptr(), op.cout, op.ds_cin, 1, cpad,
                                            op.ds_cin, 1, kstep, pds.data_ptr(), stream), "ppn_pack_weight")
                packed = torch.cat([packed, pds], dim=1).contiguous()
                ktot += op.ds_cin
                self._dev[op.name + ".b1"] = (bds if b1 is None else b1 + bds).float().to(dev)
            self._dev[op.name + ".w"] = packed
            self._dev[op.name + ".geom"] = (ktot, cpad)
            if op.nchw_f32_out and self._head_edge_pad():
                # fused-decode plans (forward_u8(fused_decode=True)): the head conv as TWO launches -- the 6K unary
                # channels as an ordinary NCHW conv of their own, and the limb channels with one 448-row channel tile per
                # edge (ppn_conv_desc.limb_edge_pad), whose epilogue reduces each window's arg-max on the accumulators
                ep, nun = self._head_edge_pad(), 6 * len(self.keypoint_names)
                win, ne = self.local_grid_size[0] * self.local_grid_size[1], len(self.edges)
                bias = self._sd[op.bias].float() if op.bias else torch.zeros(op.cout)
                assert op.bn1 is None and op.k == 1 and op.cout == nun + ne * win
                wu = w[:nun].contiguous().to(dev)
                ks_u, _, ko_u, kt_u, cp_u = L.conv_tiling(self.compute_dtype, op.cin, nun, 1)
                pu = torch.empty(cp_u, kt_u, dtype=tdt, device=dev)
                L.check(lib.ppn_pack_weight(self.compute_dtype, wu.data_ptr(), nun, op.cin, 1, cp_u, kt_u, ko_u, ks_u,
                                            pu.data_ptr(), stream), "ppn_pack_weight")
                self._dev[op.name + ".w_unary"], self._dev[op.name + ".geom_unary"] = pu, (kt_u, cp_u)
                self._dev[op.name + ".b_unary"] = bias[:nun].contiguous().to(dev)
                we = torch.zeros(ne, ep, op.cin, 1, 1)
                we[:, :win] = w[nun:].view(ne, win, op.cin, 1, 1)
                we = we.view(ne * ep, op.cin, 1, 1).contiguous().to(dev)
                pe = torch.empty(ne * ep, ktot, dtype=tdt, device=dev)
                L.check(lib.ppn_pack_weight(self.compute_dtype, we.data_ptr(), ne * ep, op.cin, 1, ne * ep, ktot, korder,
                                            kstep, pe.data_ptr(), stream), "ppn_pack_weight")
                be = torch.zeros(ne, ep)
                be[:, :win] = bias[nun:].view(ne, win)
                self._dev[op.name + ".w_edge"], self._dev[op.name + ".b_edge"] = pe, be.view(-1).contiguous().to(dev)
                torch.cuda.synchronize(dev)               # `wu` / `we` die here: their pack kernels must have run
        torch.cuda.synchronize(dev)

    def _tdt(self, dtype=None):
        return {L.PPN_F32: torch.float32, L.PPN_BF16: torch.bfloat16, L.PPN_F16: torch.float16,
                L.PPN_F16X3: torch.float16}[self.compute_dtype if dtype is None else dtype]

    def _op_dtype(self, op) -> int:
        """The dtype a launch runs in: the model's, except that the float16x3 mode runs the convolutions the split
        kernel does not cover (cin not a multiple of 64: the stem and the first block's stride-2 convs) as exact f32."""
        if self.compute_dtype == L.PPN_BF16:
            return L.PPN_F16 if (self._half_names and op.name.startswith(self._half_names)) else L.PPN_BF16
        exact = self.compute_dtype == L.PPN_F16X3 or (self._exact_names and op.name.startswith(self._exact_names))
        if not exact:
            return self.compute_dtype
        return L.PPN_F16X3 if (op.k != 7 and op.cin % 64 == 0 and op.cout >= 64) else L.PPN_F32

    def _block64_pair(self, oi: int, store_dt) -> bool:
        """Do ops oi, oi + 1 form a 64-channel stride-1 BasicBlock that csrc/block64.hip runs as one launch?  (16-bit modes;
        conv1 64 -> 64 3x3 -> bn2 -> ReLU -> conv2 64 -> 64 3x3 (+ x) with the mid tensor read by conv2 alone.)"""
        if not self.fuse_block or oi + 1 >= len(self._ops):
            return False
        c1, c2 = self._ops[oi], self._ops[oi + 1]
        odt = self._op_dtype(c1)
        if odt not in (L.PPN_BF16, L.PPN_F16) or self._op_dtype(c2) != odt:
            return False
        for c in (c1, c2):
            if not (c.cin == 64 and c.cout == 64 and c.k == 3 and c.stride == 1 and c.dilation == 1 and c.pad == 1 and
                    not c.ds_src and not c.nchw_f32_out and c.next3x3 is None and
                    self._dev.get(c.name + ".geom") == (576, 64)):
                return False
        if not (c1.out_raw and c2.src == c1.out_raw and not c1.out_act and not c1.residual and c1.bias is None):
            return False
        if sum(1 for o in self._ops if c1.out_raw in (o.src, o.residual, o.ds_src)) != 1:
            return False
        outs = [store_dt[n] for n in (c2.out_raw, c2.out_act) if n]
        return bool(outs) and all(o == odt for o in outs) and all(a in (A.ACT_NONE, A.ACT_RELU, A.ACT_LRELU)
                                                                  for a in (c1.act1, c2.act1, c2.act2))

    def _block64_first(self, oi: int, store_dt, s2_tensor) -> bool:
        """Do ops oi .. oi + 2 form layer3's first block -- 1x1 stride-2 projection of the (subsampled) raw stem output, conv1 3x3
        stride 2 from 32 channels, conv2 64 -> 64 + the projection -- that csrc/block64.hip runs as one launch?"""
        if not self.fuse_block or s2_tensor is None or oi + 2 >= len(self._ops):
            return False
        ds, c1, c2 = self._ops[oi], self._ops[oi + 1], self._ops[oi + 2]
        odt = self._op_dtype(ds)
        if odt not in (L.PPN_BF16, L.PPN_F16) or self._op_dtype(c1) != odt or self._op_dtype(c2) != odt:
            return False
        if not (ds.src == s2_tensor and ds.k == 1 and ds.stride == 2 and ds.cin == 32 and ds.cout == 64 and ds.out_raw and
                not ds.out_act and ds.act1 == A.ACT_NONE and ds.bias is None and not ds.residual):
            return False
        if not (c1.cin == 32 and c1.cout == 64 and c1.k == 3 and c1.stride == 2 and c1.dilation == 1 and c1.pad == 1 and
                c1.out_raw and not c1.out_act and not c1.residual and c1.bias is None and not c1.ds_src and
                c1.src == self._ops[0].out_act):
            return False
        if not (c2.src == c1.out_raw and c2.residual == ds.out_raw and c2.cin == 64 and c2.cout == 64 and c2.k == 3 and
                c2.stride == 1 and c2.dilation == 1 and c2.pad == 1 and not c2.ds_src and not c2.nchw_f32_out and
                self._dev.get(c2.name + ".geom") == (576, 64)):
            return False
        for t_ in (ds.out_raw, c1.out_raw):                    # read by conv2 alone
            if sum(1 for o in self._ops if t_ in (o.src, o.residual, o.ds_src)) != 1:
                return False
        for c in (ds, c1):                                     # tap-major packed rows [64][k_total]
            _, _, korder, _, cpad = L.conv_tiling(odt, c.cin, c.cout, c.k)
            if korder != 0 or cpad < 64:
                return False
        outs = [store_dt[n] for n in (c2.out_raw, c2.out_act) if n]
        return bool(outs) and all(o == odt for o in outs) and all(a in (A.ACT_NONE, A.ACT_RELU, A.ACT_LRELU)
                                                                  for a in (c1.act1, c2.act1, c2.act2))

    def _head_edge_pad(self) -> int:
        """Rows per edge of the edge-aligned limb tile (448) when the limb window fits it (385..448 values, e.g. the
        reference's 21 x 21), else 0: the chunked epilogue with atomicMax keys.  PPN_HEAD_EDGE=0 forces the latter."""
        win = self.local_grid_size[0] * self.local_grid_size[1]
        if os.environ.get("PPN_HEAD_EDGE", "1") == "0" or not (384 < win <= 448) or self.compute_dtype == L.PPN_F16X3:
            return 0
        kstep, _, korder, _, _ = L.conv_tiling(self.compute_dtype, 512, 512, 1)
        return 448 if korder == 1 else 0

    # ---- plans ------------------------------------------------------------------------------------
    def _ptr(self, key: Optional[str]):
        t = self._dev.get(key) if key else None
        return t.data_ptr() if t is not None else None

    def _build_plan(self, batch: int, h: int, w: int, src_is_u8: bool, fused: bool = False, conv_flags: int = 0) -> _Plan:
        lib = self._lib
        dev = self.device
        # the plan's own input buffer: u8 [B,H,W,3] frames or the f32 [B,3,H,W] normalised image of model.forward
        src = (torch.empty(batch, h, w, 3, dtype=torch.uint8, device=dev) if src_is_u8 else
               torch.empty(batch, 3, h, w, dtype=torch.float32, device=dev))
        tdt = self._tdt()
        shapes = A.tensor_shapes(self._ops, h, w)
        bufs: Dict[str, torch.Tensor] = {}
        producer = {}                                 # tensor name -> dtype of the launch that writes it
        for op in self._ops:
            for name in (op.out_raw, op.out_act):
                if name:
                    producer[name] = self._op_dtype(op)
        readers: Dict[str, set] = {}                  # tensor name -> dtypes of the launches that read it
        for op in self._ops:
            for name in (op.src, op.residual, op.ds_src):
                if name and name != "input":
                    readers.setdefault(name, set()).add(self._op_dtype(op))
        # Storage of every tensor.  f32 launches write f32 (plus a half-PAIR copy made by a split launch where a float16x3
        # launch reads it); float16x3 launches write half pairs, or plain half when only float16 launches read the tensor
        # (the last launch of an exact prefix); float16 launches write half, or bf16 when only bf16 launches read it (the
        # last launch of the bf16 mode's half prefix).  A tensor is read by launches of ONE type (f32 + split excepted).
        F32, BF16, F16, X3 = L.PPN_F32, L.PPN_BF16, L.PPN_F16, L.PPN_F16X3
        store_dt, need_split = {}, set()
        for name, p_ in producer.items():
            rs = readers.get(name, set())
            if p_ == F32:
                assert rs <= {F32, X3}, f"{name}: an f32 tensor read by {rs}"
                store_dt[name] = F32
                if X3 in rs:
                    need_split.add(name)
            elif p_ == X3:
                if rs and rs <= {F16}:
                    store_dt[name] = F16                      # PPN_CONV_X3_PLAIN_OUT
                else:
                    assert rs <= {X3}, f"{name}: a half-pair tensor read by {rs}"
                    store_dt[name] = X3
            elif p_ == F16:
                if rs == {BF16}:
                    store_dt[name] = BF16                     # PPN_CONV_OUT_BF16
                else:
                    assert rs <= {F16}, f"{name}: a half tensor read by {rs}"
                    store_dt[name] = F16
            else:
                assert rs <= {BF16}, f"{name}: a bf16 tensor read by {rs}"
                store_dt[name] = BF16
        x3 = bool(need_split) or any(v == X3 for v in store_dt.values())       # the plan holds split launches
        # round 5: when the fused stem's RAW output is read by nothing but the first BasicBlock's 1x1 stride-2 projection
        # (drn.py:53-54), the stem writes only the pixels that projection reads (even row and column: PPN_STEM_RAW_S2) and the
        # projection runs at stride 1 over the dense quarter-size tensor -- same values, 19 instead of 75 MB written and read
        s2_tensor = None
        stem = self._ops[0]
        if (stem.k == 7 and stem.next_s2 is not None and stem.out_raw and os.environ.get("PPN_STEM_RAW_S2", "1") != "0"):
            rd_ops = [o for o in self._ops if stem.out_raw in (o.src, o.residual, o.ds_src)]
            if (len(rd_ops) == 1 and rd_ops[0].src == stem.out_raw and rd_ops[0].k == 1 and rd_ops[0].stride == 2 and
                    rd_ops[0].pad == 0 and not rd_ops[0].ds_src and store_dt[stem.out_raw] in (BF16, F16)):
                s2_tensor = stem.out_raw
        for name, (th, tw, tc) in shapes.items():
            if name == "input":
                continue
            if name == s2_tensor:
                th, tw = (th + 1) // 2, (tw + 1) // 2
            if name == "head":
                bufs[name] = torch.empty(batch, tc, th, tw, dtype=torch.float32, device=dev)
            elif store_dt[name] == X3:
                bufs[name] = torch.empty(batch, th, tw, 2 * tc, dtype=torch.float16, device=dev)   # [hi(C) | lo'(C)]
            else:
                bufs[name] = torch.empty(batch, th, tw, tc, dtype=self._tdt(store_dt[name]), device=dev)
                if name in need_split:
                    bufs[name + "#x3"] = torch.empty(batch, th, tw, 2 * tc, dtype=torch.float16, device=dev)

        def rd(name, odt):                            # the buffer a launch of dtype `odt` reads tensor `name` from
            return bufs[name + "#x3"] if (odt == L.PPN_F16X3 and store_dt.get(name) == L.PPN_F32) else bufs[name]

        def add_splits(op):                           # behind an f32 launch: convert the outputs split launches read
            for name in (op.out_raw, op.out_act):
                if name and name in need_split:
                    th_, tw_, tc_ = shapes[name]
                    L.check(lib.ppn_plan_add_split(handle, bufs[name].data_ptr(), batch * th_ * tw_, tc_,
                                                   bufs[name + "#x3"].data_ptr()), "ppn_plan_add_split")
                    entries.append((f"split({name})", 0))
        handle = C.c_void_p()
        L.check(lib.ppn_plan_create(C.byref(handle)), "ppn_plan_create")
        entries = []
        if fused:
            # decode front end fused into the head conv: the head tensor is never materialised
            th, tw, _ = shapes["head"]
            n_unary = 6 * len(self.keypoint_names)
            del bufs["head"]
            bufs["unary"] = torch.empty(batch, n_unary, th, tw, dtype=torch.float32, device=dev)
            bufs["keys"] = torch.empty(batch, len(self.edges), th, tw, dtype=torch.int64, device=dev)
        # prefetch hint (ppn_conv_desc.prefetch, round 5): every large-tile launch touches the packed weights of the NEXT
        # launch before its epilogue -- a layer's weights were last read a whole pass ago and its first round of workgroups
        # otherwise fetches them from HBM in lockstep (PPN_PREFETCH=0 switches the hint off; results do not depend on it)
        use_pf = os.environ.get("PPN_PREFETCH", "1") != "0"

        def set_prefetch(d, t):
            if use_pf and t is not None:
                d.prefetch, d.prefetch_bytes = t.data_ptr(), t.numel() * t.element_size()
        edge_head = fused and bool(self._head_edge_pad())
        skip = set()
        for oi, op in enumerate(self._ops):
            if oi in skip:
                continue
            ih, iw, _ = shapes[op.src]
            oh, ow = A.out_hw(op, ih, iw)
            entries.append((op.name, A.op_flops(op, shapes) * batch))
            odt = self._op_dtype(op)
            nxt = self._ops[oi + 1] if oi + 1 < len(self._ops) else None
            nxt_w = None if nxt is None else self._dev.get(
                nxt.name + (".w_unary" if (nxt.nchw_f32_out and edge_head) else ".w"))
            if self._block64_first(oi, store_dt, s2_tensor):
                # layer3's FIRST block as one launch (csrc/block64.hip, stride 2): the 1x1 stride-2 projection + BN of the raw
                # stem output (read at the even pixels the stem wrote), conv1 3x3 stride 2 from the pre-activated stem output,
                # bn2 + ReLU, conv2 + shortcut, second output.  Bit-identical to the three launches.
                ds, c1, c2 = self._ops[oi], self._ops[oi + 1], self._ops[oi + 2]
                skip.update((oi + 1, oi + 2))
                entries[-1] = (f"{ds.name}+conv1+conv2", sum(A.op_flops(o, shapes) for o in (ds, c1, c2)) * batch)
                ih1, iw1, _ = shapes[c1.src]
                oh1, ow1 = A.out_hw(c1, ih1, iw1)
                bd = L.BlockDesc()
                bd.dtype, bd.batch, bd.h, bd.w, bd.channels, bd.stride, bd.in_h, bd.in_w = odt, batch, oh1, ow1, 64, 2, ih1, iw1
                bd.src, bd.proj_src = rd(c1.src, odt).data_ptr(), bufs[ds.src].data_ptr()
                bd.weight1, bd.w1_ld = self._ptr(c1.name + ".w"), self._dev[c1.name + ".geom"][0]
                bd.scale_mid, bd.shift_mid, bd.act_mid = self._ptr(c1.name + ".s1"), self._ptr(c1.name + ".b1"), c1.act1
                bd.proj_weight, bd.proj_ld = self._ptr(ds.name + ".w"), self._dev[ds.name + ".geom"][0]
                bd.proj_scale, bd.proj_shift = self._ptr(ds.name + ".s1"), self._ptr(ds.name + ".b1")
                bd.weight2, bd.scale1, bd.shift1, bd.act1 = (self._ptr(c2.name + ".w"), self._ptr(c2.name + ".s1"),
                                                             self._ptr(c2.name + ".b1"), c2.act1)
                bd.scale2, bd.shift2, bd.act2 = self._ptr(c2.name + ".s2"), self._ptr(c2.name + ".b2"), c2.act2
                bd.out_raw = bufs[c2.out_raw].data_ptr() if c2.out_raw else None
                bd.out_act = bufs[c2.out_act].data_ptr() if c2.out_act else None
                L.check(lib.ppn_plan_add_block(handle, C.byref(bd)), f"ppn_plan_add_block({ds.name})")
                continue
            if self._block64_pair(oi, store_dt):
                # a whole 64-channel stride-1 BasicBlock as ONE launch (csrc/block64.hip, round 5): conv1 -> bn2 -> ReLU -> conv2
                # (+ x, second output); the tensor between the convolutions stays in LDS.  Bit-identical to the two launches.
                c2 = self._ops[oi + 1]
                skip.add(oi + 1)
                entries[-1] = (f"{op.name}+conv2", (A.op_flops(op, shapes) + A.op_flops(c2, shapes)) * batch)
                bd = L.BlockDesc()
                bd.dtype, bd.batch, bd.h, bd.w, bd.channels = odt, batch, ih, iw, 64
                bd.src, bd.residual = rd(op.src, odt).data_ptr(), (rd(c2.residual, odt).data_ptr() if c2.residual else None)
                bd.weight1, bd.scale_mid, bd.shift_mid, bd.act_mid = (self._ptr(op.name + ".w"), self._ptr(op.name + ".s1"),
                                                                     self._ptr(op.name + ".b1"), op.act1)
                bd.weight2, bd.scale1, bd.shift1, bd.act1 = (self._ptr(c2.name + ".w"), self._ptr(c2.name + ".s1"),
                                                             self._ptr(c2.name + ".b1"), c2.act1)
                bd.scale2, bd.shift2, bd.act2 = self._ptr(c2.name + ".s2"), self._ptr(c2.name + ".b2"), c2.act2
                bd.out_raw = bufs[c2.out_raw].data_ptr() if c2.out_raw else None
                bd.out_act = bufs[c2.out_act].data_ptr() if c2.out_act else None
                L.check(lib.ppn_plan_add_block(handle, C.byref(bd)), f"ppn_plan_add_block({op.name})")
                continue
            if op.k == 7 and op.next_s2 is not None:
                assert op.src == "input" and self.compute_dtype in (L.PPN_BF16, L.PPN_F16)
                out_dt = store_dt[op.out_raw or op.out_act]
                sdt = self.stem_dtype if self.stem_dtype is not None else self.compute_dtype
                if sdt != out_dt:
                    sdt = sdt | ((out_dt + 1) << 8)                                  # PPN_STEM_IO(internal, out)
                if s2_tensor is not None:
                    sdt = sdt | L.PPN_STEM_RAW_S2
                L.check(lib.ppn_plan_add_stem012_dt(handle, sdt, 1 if src_is_u8 else 0, src.data_ptr(), batch, h, w,
                                                 self._ptr(op.name + ".w"), self._ptr(op.name + ".s1"),
                                                 self._ptr(op.name + ".b1"), self._mean, self._std,
                                                 self._ptr(op.name + ".w1"), self._ptr(op.name + ".s1b"),
                                                 self._ptr(op.name + ".b1b"), self._ptr(op.name + ".w2"),
                                                 self._ptr(op.name + ".s1c"), self._ptr(op.name + ".b1c"),
                                                 self._ptr(op.name + ".s2"), self._ptr(op.name + ".b2"),
                                                 bufs[op.out_raw].data_ptr() if op.out_raw else None,
                                                 bufs[op.out_act].data_ptr() if op.out_act else None),
                        "ppn_plan_add_stem012_dt")
                continue
            if op.k == 7 and op.next3x3 is not None:
                assert op.src == "input" and op.out_act is None
                L.check(lib.ppn_plan_add_stem01(handle, odt, 1 if src_is_u8 else 0, src.data_ptr(),
                                                batch, h, w, self._ptr(op.name + ".w"), self._ptr(op.name + ".s1"),
                                                self._ptr(op.name + ".b1"), self._mean, self._std,
                                                self._ptr(op.name + ".w1"), self._ptr(op.name + ".s1b"),
                                                self._ptr(op.name + ".b1b"), bufs[op.out_raw].data_ptr()),
                        "ppn_plan_add_stem01")
                add_splits(op)
                continue
            if op.k == 7:
                assert op.src == "input" and op.out_act is None
                L.check(lib.ppn_plan_add_stem(handle, odt, 1 if src_is_u8 else 0, src.data_ptr(),
                                              batch, h, w, self._ptr(op.name + ".w"), self._ptr(op.name + ".s1"),
                                              self._ptr(op.name + ".b1"), self._mean, self._std,
                                              bufs[op.out_raw].data_ptr()), "ppn_plan_add_stem")
                add_splits(op)
                continue
            d = L.ConvDesc()
            d.dtype = odt
            d.flags = conv_flags
            outs = [store_dt[n] for n in (op.out_raw, op.out_act) if n and n in store_dt and n != "head"]
            if odt == L.PPN_F16 and outs:
                if all(o == L.PPN_BF16 for o in outs):
                    d.flags |= L.PPN_CONV_OUT_BF16                                   # last launch of the IEEE-half prefix
                else:
                    assert all(o == L.PPN_F16 for o in outs), f"{op.name}: outputs of mixed storage types"
            if odt == L.PPN_F16X3 and outs:
                if all(o == L.PPN_F16 for o in outs):
                    d.flags |= L.PPN_CONV_X3_PLAIN_OUT                               # last launch of an exact prefix
                else:
                    assert all(o == L.PPN_F16X3 for o in outs), f"{op.name}: outputs of mixed storage types"
            d.batch, d.in_h, d.in_w, d.cin = batch, ih, iw, op.cin
            d.out_h, d.out_w, d.cout = oh, ow, op.cout
            d.ksize, d.stride, d.dilation, d.pad = op.k, op.stride, op.dilation, op.pad
            if op.src == s2_tensor:                  # the stem wrote only the pixels this 1x1 stride-2 projection reads
                d.in_h, d.in_w, d.stride = (ih + 1) // 2, (iw + 1) // 2, 1
            d.k_total, d.cout_pad = self._dev[op.name + ".geom"]
            d.act1, d.act2 = op.act1, op.act2
            d.out_nchw_f32 = 1 if op.nchw_f32_out else 0
            d.src = rd(op.src, odt).data_ptr()
            d.weight = self._ptr(op.name + ".w")
            d.scale1, d.shift1 = self._ptr(op.name + ".s1"), self._ptr(op.name + ".b1")
            d.residual = rd(op.residual, odt).data_ptr() if op.residual else None
            if op.ds_src:
                sh2, sw2, sc2 = shapes[op.ds_src]
                d.src2, d.in2_h, d.in2_w, d.cin2, d.stride2 = bufs[op.ds_src].data_ptr(), sh2, sw2, sc2, op.ds_stride
            d.out_raw = bufs[op.out_raw].data_ptr() if (op.out_raw and op.out_raw in bufs) else None
            set_prefetch(d, nxt_w)
            if fused and op.nchw_f32_out and self._head_edge_pad():
                # (1) the unary channels: an ordinary sigmoid NCHW conv straight into the compact unary tensor
                keys, nun = bufs["keys"], bufs["unary"].shape[1]
                name, flops = entries.pop()
                d.cout = nun
                d.k_total, d.cout_pad = self._dev[op.name + ".geom_unary"]
                d.weight, d.shift1 = self._ptr(op.name + ".w_unary"), self._ptr(op.name + ".b_unary")
                d.out_raw = bufs["unary"].data_ptr()
                d.zero_page = self._dev["zero"].data_ptr()
                set_prefetch(d, self._dev[op.name + ".w_edge"])
                L.check(lib.ppn_plan_add_conv(handle, C.byref(d)), f"ppn_plan_add_conv({op.name}.unary)")
                d.prefetch, d.prefetch_bytes = None, 0
                entries.append((name + ".unary", flops * nun // op.cout))
                # (2) the limb channels, one channel tile per edge: keys are stored, not accumulated -- no zero fill
                ep = self._head_edge_pad()
                d.cout = op.cout - nun
                d.k_total, d.cout_pad = self._dev[op.name + ".geom"][0], len(self.edges) * ep
                d.weight, d.shift1 = self._ptr(op.name + ".w_edge"), self._ptr(op.name + ".b_edge")
                d.out_raw, d.argmax_keys, d.limb_edge_pad = None, keys.data_ptr(), ep
                d.limb_window = self.local_grid_size[0] * self.local_grid_size[1]
                L.check(lib.ppn_plan_add_conv(handle, C.byref(d)), f"ppn_plan_add_conv({op.name}.limbs)")
                entries.append((name + ".limbs", flops - flops * nun // op.cout))
                continue
            if fused and op.nchw_f32_out:
                keys = bufs["keys"]
                L.check(lib.ppn_plan_add_memset(handle, keys.data_ptr(), keys.numel() * 8), "ppn_plan_add_memset")
                entries.insert(len(entries) - 1, ("zero arg-max keys", 0))
                d.unary_out, d.argmax_keys = bufs["unary"].data_ptr(), keys.data_ptr()
                d.unary_channels = bufs["unary"].shape[1]
                d.limb_window = self.local_grid_size[0] * self.local_grid_size[1]
            d.scale2, d.shift2 = self._ptr(op.name + ".s2"), self._ptr(op.name + ".b2")
            d.out_act = bufs[op.out_act].data_ptr() if op.out_act else None
            d.zero_page = self._dev["zero"].data_ptr()
            # two launches with different tiles where the launcher would cut the pixel range (ppn_conv_split): listed as
            # two plan entries so that each launch is timed and named by itself
            m_all, cut = batch * oh * ow, C.c_int64(0)
            L.check(lib.ppn_conv_split(self.compute_dtype, op.cin, op.cout, m_all, C.byref(cut)), "ppn_conv_split")
            if cut.value:
                name, flops = entries.pop()
                for lo, n in ((0, cut.value), (cut.value, m_all - cut.value)):
                    d.m_begin, d.m_count = lo, n
                    L.check(lib.ppn_plan_add_conv(handle, C.byref(d)), f"ppn_plan_add_conv({op.name})")
                    entries.append((f"{name}[{lo}:{lo + n}]", flops * n // m_all))
                continue
            L.check(lib.ppn_plan_add_conv(handle, C.byref(d)), f"ppn_plan_add_conv({op.name})")
            if odt == L.PPN_F32:
                add_splits(op)
        head = (bufs["unary"], bufs["keys"]) if fused else bufs["head"]
        return _Plan(handle, bufs, head, entries, A.conv_flops(self._ops, h, w) * batch, src)

    def _get_plan(self, b: int, h: int, w: int, src_is_u8: bool, fused: bool = False, slot: int = 0,
                  conv_flags: int = 0) -> _Plan:
        if not self._dev:
            raise RuntimeError("PoseProposalNet: call load_state_dict() first")
        # slot: independent output buffers (pipelined serving); conv_flags: ppn_conv_desc.flags of every conv of the plan
        key = (b, h, w, src_is_u8, fused, slot) if not conv_flags else (b, h, w, src_is_u8, fused, slot, conv_flags)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._build_plan(b, h, w, src_is_u8, fused, conv_flags)
        return plan

    def _plan_for(self, x: torch.Tensor, src_is_u8: bool, fused: bool = False, slot: int = 0, conv_flags: int = 0) -> _Plan:
        if src_is_u8:
            b, h, w, _ = x.shape
        else:
            b, _, h, w = x.shape
        plan = self._get_plan(b, h, w, src_is_u8, fused, slot, conv_flags)
        # Frames go through the plan's own input buffer (a D2D copy on the caller's stream: 442 KB per 384x384 u8
        # frame), so the hipGraph captured for this plan is replayed whatever tensor the caller hands in -- a server
        # that uploads a fresh tensor per frame would otherwise re-capture ~35 nodes on every call.  Callers that
        # want zero copies write into `input_buffer(...)` and pass that tensor.
        if x.data_ptr() != plan.input.data_ptr():
            plan.input.copy_(x)
        return plan

    def input_buffer(self, batch: int, h: int, w: int, u8: bool = True, fused_decode: bool = False,
                     slot: int = 0, conv_flags: int = 0) -> torch.Tensor:
        """The plan-owned input tensor for this shape (u8 [B,H,W,3] or f32 [B,3,H,W]): fill it (e.g. an H2D copy
        straight into it) and pass it to forward_u8 / forward to skip the D2D copy."""
        return self._get_plan(batch, h, w, u8, fused_decode, slot, conv_flags).input

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """model.py:104-136: f32 [B,3,H,W] normalised image -> sigmoid head f32 [B,lastsize,H/16,W/16].

        The returned tensor is owned by the model's plan for this input buffer and is overwritten by the
        next forward of the same shape (clone it to keep it)."""
        if not (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.shape[1] == 3):
            raise ValueError("forward expects a float32 CUDA tensor [B,3,H,W]")
        if self.training:                                  # model.train(): batch statistics, running stats advance
            return self._trainer.forward(input)
        x = input.contiguous()
        plan = self._plan_for(x, False)
        L.check(self._lib.ppn_plan_run(plan.handle, L.current_stream_ptr()), "ppn_plan_run")
        return plan.head

    __call__ = forward

    def forward_u8(self, frames: torch.Tensor, fused_decode: bool = False, slot: int = 0, conv_flags: int = 0):
        """Fused rt_test.py:97-101 + forward: u8 [B,H,W,3] RGB frames on the device -> head.

        With ``fused_decode=True`` the head conv's epilogue runs the decode's limb arg-max itself and the
        17.5 MB/image head is never written: returns ``(unary f32 [B,6K,H,W], keys i64 [B,E,H,W])`` for
        ``Decoder.decode_fused`` (results bit-identical to decoding the materialised head).  Plans of different
        ``slot`` own different output buffers, so a consumer on another stream may still be reading slot 0's
        outputs while slot 1's forward runs (rt.InferencePipeline).  ``conv_flags``: ppn_conv_desc.flags of the plan's
        convolutions (lib.PPN_CONV_NO_FILTER_BANK: the choice of a plan that shares the GPU with other lanes)."""
        if not (frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3):
            raise ValueError("forward_u8 expects a uint8 CUDA tensor [B,H,W,3]")
        if self.training:
            raise RuntimeError("forward_u8 is the inference entry point (folded BN): call model.eval() first")
        x = frames.contiguous()
        plan = self._plan_for(x, True, fused_decode, slot, conv_flags)
        L.check(self._lib.ppn_plan_run(plan.handle, L.current_stream_ptr()), "ppn_plan_run")
        return plan.head

    def profile_layers(self, x: torch.Tensor, src_is_u8: bool = False, repeats: int = 1, fused_decode: bool = False,
                       conv_flags: int = 0):
        """Per-launch durations (ms) measured with HIP events on the launch stream: [(op name, kernel, ms, flops)].
        `repeats` launches of each op are issued back to back between its events (amortises the event gap).
        conv_flags: as forward_u8 -- pass the timed path's flags to time the kernels THAT path runs."""
        plan = self._plan_for(x.contiguous(), src_is_u8, fused_decode, conv_flags=conv_flags)
        ms = (C.c_float * plan.n_ops)()
        L.check(self._lib.ppn_plan_run_timed(plan.handle, L.current_stream_ptr(), ms, plan.n_ops, repeats),
                "ppn_plan_run_timed")
        return [(name, self._lib.ppn_plan_kernel_name(plan.handle, i).decode(), float(ms[i]), fl)
                for i, (name, fl) in enumerate(plan.entries)]

    def half_range_report(self, frames: torch.Tensor) -> Dict[str, float]:
        """max |value| / 65504 of every tensor a plan stores in IEEE half for these u8 frames (the bf16 mode's half PREFIX --
        stem + layer3-4 -- and every tensor of the float16 mode).  Half stores CLAMP at +-65504 instead of overflowing
        (csrc/conv_common.h clamp_f16), so a checkpoint whose early activations exceed that range saturates silently: a
        value of 1.0 here says it did (tests/test_16bit_floors_gpu.py::test_half_prefix_saturates_at_65504); run such a
        checkpoint with half_prefix=-1, stem_dtype="bfloat16" (pure bf16: the f32 exponent range) or in float32."""
        self.forward_u8(frames)
        b, h, w, _ = frames.shape
        plan = self._get_plan(b, h, w, True)
        torch.cuda.synchronize(self.device)
        return {name: float(t.abs().max().item()) / 65504.0 for name, t in plan.buffers.items()
                if isinstance(t, torch.Tensor) and t.dtype == torch.float16 and not name.endswith("#x3")}

    def graph_captures(self) -> Dict[tuple, int]:
        """How often each plan (batch, h, w, u8, fused, slot) has captured its launch sequence into a hipGraph."""
        return {k: int(self._lib.ppn_plan_graph_captures(p.handle)) for k, p in self._plans.items()}

    def __del__(self):
        try:
            for p in self._plans.values():
                self._lib.ppn_plan_destroy(p.handle)
        except Exception:
            pass
