"""Training step of the Pose Proposal Network on one MI355X (main.py:623-777).

`PPNTrainer` keeps every parameter of the reference's PoseProposalNet(arch) in ONE flat f32 buffer (in
`model.named_parameters()` order, so `params[-13]` is the head's conv1.weight as in main.py:704) with a matching
flat gradient buffer: the data-parallel exchange is a few 32 MB RCCL all-reduces of slices of that buffer, issued while
the backward pass is still running, and the optimiser a single launch.
Forward and backward are explicit: a tape of saved activations per unit (arch._units), train-mode BatchNorm with
batch statistics (train.py / csrc/train.hip), convolutions, input gradients and weight gradients on the MFMA
kernels (csrc/conv*.hip, csrc/wgrad.hip), the loss and its gradient from csrc/loss.hip.  There is no autograd and
no CPU path.

One iteration (main.py:664-777):
    head              = forward(x)                                  train-mode BN, running stats updated
    L_i, dhead        = PPNLoss fwd+bwd with coeff w_i/5            loss = sum w_i L_i / 5
    grad              = backward(dhead)                             d loss / d theta
    gnorm_i           = ||d L_i / d W||, W = head conv1.weight      5 partial backward passes (head only)
    w                <- GradNorm weight step (Adam), all-reduce, clamp, renormalise
    grad             <- all-reduce(grad) / world;  theta <- Adam(theta, grad)

main.py:759 back-propagates Lgrad through the graph of the G_i (create_graph=True), which adds d Lgrad / d theta -- a
second-order term -- to the model gradients.  With second_order=True (default) the trainer adds exactly that term: the
probe weight only influences the head's tail, so the term is a reverse pass over a forward-mode (dual) evaluation of
that tail per loss, whose adjoints join the first-order ones before the backbone is back-propagated once
(_second_order_tail).  second_order=False applies d loss / d theta only (35 % faster per step).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import arch as A
from . import config as cfg
from . import lib as L
from . import train as T
from .loss import PPNLoss

_BUFFER_SUFFIXES = (".running_mean", ".running_var", ".num_batches_tracked")


class PPNTrainer:
    def __init__(self, arch: str = "drn_d_22", state_dict: Optional[Dict] = None, compute_dtype: int = L.PPN_BF16,
                 lr: float = 7e-4, lr_weights: Optional[float] = None, alpha: float = 0.12, insize=(384, 384),
                 device="cuda", second_order: bool = True):
        L.load()                                               # fail loudly without libppn.so
        self.arch, self.compute_dtype, self.device = arch, compute_dtype, torch.device(device)
        self.tdt = torch.float32 if compute_dtype == L.PPN_F32 else torch.bfloat16
        self.units = A._units(arch)
        self.insize = insize
        spec = A.param_spec(arch)
        self.param_names = [n for n, _ in spec if not n.endswith(_BUFFER_SUFFIXES)]
        shapes = dict(spec)
        n_total = sum(int(np.prod(shapes[n])) for n in self.param_names)
        self.flat = torch.zeros(n_total, dtype=torch.float32, device=self.device)
        # The flat gradient buffer sits behind a 16-float prefix of its own allocation: the five GradNorm task weights
        # live in prefix[0:5] (GradNormWeights.bind), so the LAST gradient bucket of the data-parallel exchange -- the
        # lowest addresses, issued when the backward is complete -- carries them along and the separate 20-byte,
        # latency-bound all-reduce of main.py:769-771 disappears (one collective fewer per step).
        self._grad_store = torch.zeros(16 + n_total, dtype=torch.float32, device=self.device)
        self.grad = self._grad_store[16:]
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        self.offset: Dict[str, int] = {}
        o = 0
        for n in self.param_names:
            k = int(np.prod(shapes[n]))
            self.P[n] = self.flat[o:o + k].view(shapes[n])
            self.G[n] = self.grad[o:o + k].view(shapes[n])
            self.offset[n] = o
            o += k
        self.buffers: Dict[str, torch.Tensor] = {}
        for n, shp in spec:
            if n.endswith((".running_mean", ".running_var")):
                self.buffers[n] = torch.zeros(shp, dtype=torch.float32, device=self.device)
                if n.endswith("var"):
                    self.buffers[n].fill_(1.0)
        self.num_batches_tracked = 0
        self._storage_key = T.register_param_storage(self.flat)               # packed-weight cache (train.conv2d_nhwc)
        if state_dict is not None:
            self.load_state_dict(state_dict)
        self.opt = T.FlatAdam(self.flat, lr=lr)                               # optimizerM, main.py:278
        # weight_model + optimizerR: the reference gives both optimisers args.lr (main.py:278-279)
        self.task = T.GradNormWeights(self.device, lr=lr if lr_weights is None else lr_weights, alpha=alpha)
        # an ALIAS of store[0:5], not a view: a view would share the store's version counter, and every torch op on a
        # gradient slice would then invalidate the staged host copy of the weights (GradNormWeights.host_weights)
        self.task.bind(torch.empty(0, dtype=torch.float32, device=self.device).set_(
            self._grad_store.untyped_storage(), self._grad_store.storage_offset(), (5,), (1,)))
        self.criterion = PPNLoss(insize=insize, outsize=(insize[0] // 16, insize[1] // 16))
        self.base: Optional[torch.Tensor] = None
        self._tape = None
        import os
        self._side = (torch.cuda.Stream(device=self.device)
                      if os.environ.get("PPN_TRAIN_SIDE_STREAM", "1") != "0" else None)
        # (A high-priority probe stream was tried -- PPN_TRAIN_PROBE_PRIORITY=-1 -- because the main stream waits ~0.65 ms
        # for the four probe passes, which take 0.5 ms each beside the head backward and 0.23 ms alone: no measurable
        # change, 1.48 vs 1.52 ms from the end of the loss to the fourth probe; tools/host_timeline.py.)
        self._stacked_probes = os.environ.get("PPN_TRAIN_STACKED_PROBES", "1") != "0"
        # second-order tail: enqueue its forward-mode half before the host reads the probe norms (A/B: =0)
        self._speculate_tail = os.environ.get("PPN_TRAIN_SPECULATE_TAIL", "1") != "0"
        self._so_pin = None
        self._w3p = {}                               # _w3_padded
        self._wg0_main = os.environ.get("PPN_TRAIN_WG0_MAIN", "1") != "0"
        self._fuse_stats = T._FUSE_STATS and self.tdt == torch.bfloat16
        self._tail_wgrad_side = os.environ.get("PPN_TRAIN_TAIL_WGRAD_SIDE", "1") != "0"
        pri = int(os.environ.get("PPN_TRAIN_PROBE_PRIORITY", "0"))
        self._probe_stream = torch.cuda.Stream(device=self.device, priority=pri) if self._side is not None else None
        self._probe_scratch = None
        self._conv1_local = None
        # second_order: add d Lgrad / d theta (main.py:759, the double backward through the probe gradients) to the
        # model gradients.  Everything then runs on one stream (the second-order pass accumulates into gradients the
        # first-order pass has just written).
        self.second_order = second_order

    # ---- state ------------------------------------------------------------------------------------------------
    def __del__(self):
        try:
            T.unregister_param_storage(self._storage_key)
            for ent in getattr(self, "_w3p", {}).values():
                T.unregister_param_storage(ent[2])
        except Exception:
            pass

    def load_state_dict(self, sd):
        T.bump_param_version()                     # parameters are overwritten in place: packed copies are stale
        for n in self.param_names:
            v = sd[n] if n in sd else sd["module." + n]
            self.P[n].copy_(torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).float())
        for n, b in self.buffers.items():
            v = sd.get(n, sd.get("module." + n))
            if v is not None:
                b.copy_(torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).float())

    def state_dict(self):
        """Checkpoint-compatible dict (main.py:1175-1181 'state_dict' entry)."""
        sd = {n: self.P[n].detach().clone() for n in self.param_names}
        for n, b in self.buffers.items():
            sd[n] = b.clone()
            if n.endswith(".running_var"):
                sd[n[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(self.num_batches_tracked)
        return sd

    # ---- checkpoints in the reference's format (main.py:436-451 save, 302-331 resume) ---------------------------------
    def checkpoint(self, epoch: int = 0, best_AP: float = 0.0, arch_field: str = "resnet18") -> Dict:
        """The dict main.py:443-451 hands to torch.save: model state_dict, weight_model state_dict ([1,5] `weight`),
        and torch.optim.Adam state_dicts of optimizerM / optimizerR (per-parameter step / exp_avg / exp_avg_sq, CPU
        tensors).  `arch_field` is what the reference stores there: its unused --arch default (main.py:68, 445)."""
        def adam_state(opt_lr, betas, eps, wd, step, items):
            state = {i: {"step": torch.tensor(float(step)), "exp_avg": m.detach().cpu().clone(),
                         "exp_avg_sq": v.detach().cpu().clone()} for i, (m, v) in enumerate(items)} if step > 0 else {}
            group = {"lr": opt_lr, "betas": tuple(betas), "eps": eps, "weight_decay": wd, "amsgrad": False,
                     "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                     "params": list(range(len(items)))}
            return {"state": state, "param_groups": [group]}

        shapes = {n: self.P[n].shape for n in self.param_names}
        items = []
        for n in self.param_names:
            o, k = self.offset[n], self.P[n].numel()
            items.append((self.opt.exp_avg[o:o + k].view(shapes[n]), self.opt.exp_avg_sq[o:o + k].view(shapes[n])))
        return {
            "epoch": epoch, "arch": arch_field, "best_AP": best_AP,
            "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
            "weight_state_dict": {"weight": self.task.w.detach().cpu().view(1, 5).clone()},
            "optimizerM": adam_state(self.opt.lr, self.opt.betas, self.opt.eps, self.opt.weight_decay,
                                     self.opt.step_count, items),
            "optimizerR": adam_state(self.task.lr, self.task.betas, self.task.eps, 0.0, self.task.step_count,
                                     [(self.task.exp_avg.view(1, 5), self.task.exp_avg_sq.view(1, 5))]),
        }

    def load_checkpoint(self, ckpt: Dict) -> int:
        """Resume from a reference-format checkpoint (with or without DDP's `module.` prefixes); returns the epoch."""
        self.load_state_dict(ckpt["state_dict"])
        self.task.w.copy_(torch.as_tensor(ckpt["weight_state_dict"]["weight"]).reshape(5).float())

        def restore(sd, items, set_step):
            st = sd.get("state", {})
            step = 0
            for i, (m, v) in enumerate(items):
                e = st.get(i, st.get(str(i)))
                if e is None:
                    m.zero_(); v.zero_()
                    continue
                m.copy_(torch.as_tensor(e["exp_avg"]).reshape(m.shape).float())
                v.copy_(torch.as_tensor(e["exp_avg_sq"]).reshape(v.shape).float())
                step = max(step, int(float(e["step"])))
            set_step(step, sd["param_groups"][0])

        items = []
        for n in self.param_names:
            o, k = self.offset[n], self.P[n].numel()
            items.append((self.opt.exp_avg[o:o + k], self.opt.exp_avg_sq[o:o + k]))

        def set_m(step, g):
            self.opt.step_count, self.opt.lr = step, float(g["lr"])

        def set_r(step, g):
            self.task.step_count, self.task.lr = step, float(g["lr"])

        restore(ckpt["optimizerM"], items, set_m)
        restore(ckpt["optimizerR"], [(self.task.exp_avg, self.task.exp_avg_sq)], set_r)
        return int(ckpt.get("epoch", 0))

    def get_baseloss(self, batches):
        """main.py:578-621: the five losses in EVAL mode (running BN statistics) averaged over `batches`, an iterable
        of (x f32[B,3,H,W], targets dict); sets and returns self.base (f32[5] on the device)."""
        from . import drn, model
        net = model.PoseProposalNet(getattr(drn, self.arch)(), insize=self.insize,
                                    outsize=(self.insize[0] // 16, self.insize[1] // 16),
                                    compute_dtype="float32" if self.compute_dtype == L.PPN_F32 else "bfloat16")
        net = net.cuda(self.device)
        net.load_state_dict(self.state_dict())
        net.eval()
        total = torch.zeros(5, dtype=torch.float32, device=self.device)
        n = 0
        for x, targets in batches:
            losses, _ = self.criterion.forward_backward(net(x), targets, want_grad=False)
            total += losses
            n += 1
        if n == 0:
            raise ValueError("get_baseloss needs at least one batch")
        self.base = total / n
        return self.base

    def adjust_learning_rate(self, epoch: int):
        """main.py:1222-1231, called on optimizerM once per epoch (main.py:406): halve the rate every 300 epochs."""
        if epoch % 300 == 0 and epoch > 1:
            self.opt.lr = 0.5 * self.opt.lr
        return self.opt.lr

    # ---- small helpers ------------------------------------------------------------------------------------------
    def _bn(self, x, prefix, act, stats=None, emit=False):
        """emit: the output is the input of the next unit's BatchNorm -> (y, saved, ConvStats) (train.bn_train_forward)"""
        return T.bn_train_forward(x, self.P[prefix + ".weight"], self.P[prefix + ".bias"],
                                  self.buffers[prefix + ".running_mean"], self.buffers[prefix + ".running_var"],
                                  act=act, stats=stats, emit_stats=emit and self._fuse_stats)

    def _bn_bwd(self, x, dy, prefix, act, saved, dx_add=None, keep=True, stats=None, next_bn=None):
        """next_bn = (x2, prefix2, act2, saved2): dx is the dy of that BatchNorm -> (dx, ConvStats) (train.bn_train_backward)"""
        # parameter gradients land directly in the flat gradient buffer (probe passes discard them)
        if next_bn is not None and self._fuse_stats:
            x2, p2, act2, saved2 = next_bn
            nb = (x2, self.P[p2 + ".weight"], self.P[p2 + ".bias"], saved2, act2)
        else:
            nb = None
        r = T.bn_train_backward(x, dy, self.P[prefix + ".weight"], self.P[prefix + ".bias"], saved, act=act,
                                dx_add=dx_add, dgamma=self.G[prefix + ".weight"] if keep else None,
                                dbeta=self.G[prefix + ".bias"] if keep else None, stats=stats, next_bn=nb)
        if next_bn is not None:
            return r[0], (r[3] if nb is not None else None)
        return r[0]

    # BatchNorm statistics from the neighbouring convolution's epilogue (train.ConvStats; 16-bit mode, PPN_TRAIN_FUSE_STATS=0: off):
    # a forward convolution that feeds a BatchNorm folds {sum y, sum y^2} per pixel tile, an input-gradient convolution whose result
    # is a BatchNorm's dy folds {sum g, sum g * xhat} -- the BatchNorm call that follows then skips its own pass over the tensor(s).
    def _conv_bn(self, x, wname, *args, **kw):
        """conv2d_nhwc whose output feeds a BatchNorm: (y, ConvStats or None)"""
        if self._fuse_stats:
            return T.conv2d_nhwc(x, self.P[wname], *args, stats="fwd", **kw)
        return T.conv2d_nhwc(x, self.P[wname], *args, **kw), None

    def _dgrad_bn(self, dy, wname_or_w, in_hw, stride, dil, pad, x, prefix, act, saved, fuse=True):
        """conv_dgrad whose result is dy of the BatchNorm `prefix` (+ act) over x: (dx, ConvStats or None)"""
        w = self.P[wname_or_w] if isinstance(wname_or_w, str) else wname_or_w
        if self._fuse_stats and fuse and T._FUSE_STATS_BWD:
            return T.conv_dgrad(dy, w, in_hw, stride, dil, pad,
                                bn=(x, self.P[prefix + ".weight"], self.P[prefix + ".bias"], saved, act))
        return T.conv_dgrad(dy, w, in_hw, stride, dil, pad), None

    def _on_side(self, fn, *tensors):
        """Run fn() on the side stream once everything queued on the current stream so far is done.  Weight gradients
        are leaves of the backward graph: nothing downstream waits for them, so they run beside the input-gradient
        chain and fill the CUs its launches leave idle.  backward() joins the streams at its end."""
        if self._side is None:
            return fn()
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        for t in tensors:
            t.record_stream(self._side)                      # the caching allocator must not recycle them early
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            fn()

    def _tail_side(self, fn, *tensors):
        """_on_side for the second-order tail's weight gradients (PPN_TRAIN_TAIL_WGRAD_SIDE=0: on the main stream, for A/B)."""
        if self._tail_wgrad_side:
            return self._on_side(fn, *tensors)
        return fn()

    def _wgrad(self, name, x, dy, k, stride=1, dil=1, pad=0):
        self._on_side(lambda: T.conv_wgrad(x, dy, k, stride, dil, pad, out=self.G[name]), x, dy)

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x f32 [B,3,H,W] normalised (the model.forward argument) -> head f32 [B,7605,H/16,W/16]; train-mode BN."""
        lib = L.load()
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda:
            raise ValueError("x must be a float32 [B,3,H,W] device tensor")
        x = x.contiguous()
        B, _, H, W = x.shape
        T.bump_param_version()        # parameters may have been edited in place since the last pass: repack once per forward
        for used in list(self._w3p):  # the padded copies of conv3.weight follow the parameters BEFORE they are repacked
            self._w3_padded(used)
        # ... every weight view the previous passes of THIS trainer met, in one launch
        T.repack_all(self.device, [self._storage_key] + [ent[2] for ent in self._w3p.values()])
        tape = []
        # the 7x7 stem reads NCHW f32; its weight gradient reads an NHWC copy padded with zero channels: 4 channels for
        # the dedicated bf16 kernel (csrc/stem_wgrad.hip: two MFMAs per filter row), 8 for the generic f32 kernel
        xin8 = torch.empty(B, H, W, 4 if self.tdt == torch.bfloat16 else 8, dtype=self.tdt, device=self.device)
        L.check(lib.ppn_image_to_nhwc(self.compute_dtype, x.data_ptr(), B, H, W, xin8.shape[-1], xin8.data_ptr(),
                                      L.current_stream_ptr()), "ppn_image_to_nhwc")
        cur = None
        cur_st = None                                  # ConvStats of `cur` where the next unit opens with a BatchNorm over it
        for ui, u in enumerate(self.units):
            nxt_kind = self.units[ui + 1].kind if ui + 1 < len(self.units) else None
            if u.kind == "cbr":
                wn = f"{u.prefix}.{u.conv_idx}.weight"
                bnp = f"{u.prefix}.{u.conv_idx + 1}"
                d = u.dil[0]
                if u.k == 7:
                    y = torch.empty(B, H, W, u.cout, dtype=self.tdt, device=self.device)
                    L.check(lib.ppn_stem7x7(self.compute_dtype, 0, x.data_ptr(), B, H, W, self.P[wn].data_ptr(), None,
                                            None, None, None, y.data_ptr(), L.current_stream_ptr()), "ppn_stem7x7")
                    src = xin8
                    yst = None
                else:
                    src = cur
                    y, yst = self._conv_bn(cur, wn, u.stride, d, d)
                emit = self._fuse_stats and nxt_kind in ("basic", "head")   # those open with a BatchNorm over this unit's output
                r = self._bn(y, bnp, "relu", stats=yst, emit=emit)
                z, saved, cur_st = r[0], r[1], (r[2] if emit else None)
                tape.append(("cbr", u, dict(x=src, y=y, saved=saved)))
                cur = z
            elif u.kind == "basic":
                p = u.prefix
                a, s1 = self._bn(cur, p + ".bn1", "relu", stats=cur_st)
                cur_st = None
                c1, c1st = self._conv_bn(a, p + ".conv1.weight", u.stride, u.dil[0], u.dil[0])
                b, s2 = self._bn(c1, p + ".bn2", "relu", stats=c1st)
                ctx = dict(x=cur, a=a, s1=s1, c1=c1, b=b, s2=s2)
                if u.downsample:
                    dsy, dst = self._conv_bn(cur, p + ".downsample.0.weight", u.stride, 1, 0)
                    r, s3 = self._bn(dsy, p + ".downsample.1", "none", stats=dst)
                    ctx.update(dsy=dsy, s3=s3)
                else:
                    r = cur
                out = T.conv2d_nhwc(b, self.P[p + ".conv2.weight"], 1, u.dil[1], u.dil[1], add=r)
                tape.append(("basic", u, ctx))
                cur = out
            elif u.kind == "bottleneck":                              # drn.py:77-97 (post-activation)
                p, pl = u.prefix, u.planes
                y1, st1 = self._conv_bn(cur, p + ".conv1.weight")
                h1, s1 = self._bn(y1, p + ".bn1", "relu", stats=st1)
                y2, st2 = self._conv_bn(h1, p + ".conv2.weight", u.stride, u.dil[1], u.dil[1])
                h2, s2 = self._bn(y2, p + ".bn2", "relu", stats=st2)
                y3, st3 = self._conv_bn(h2, p + ".conv3.weight")
                z3, s3 = self._bn(y3, p + ".bn3", "none", stats=st3)
                ctx = dict(x=cur, y1=y1, h1=h1, s1=s1, y2=y2, h2=h2, s2=s2, y3=y3, s3=s3)
                if u.downsample:
                    yd, std = self._conv_bn(cur, p + ".downsample.0.weight", u.stride, 1, 0)
                    r, sd = self._bn(yd, p + ".downsample.1", "none", stats=std)
                    ctx.update(yd=yd, sd=sd)
                else:
                    r = cur
                out = T.add_relu(z3, r)
                ctx["out"] = out
                tape.append(("bottleneck", u, ctx))
                cur = out
            else:                                                     # PPN head, model.py:113-136
                R = cur
                h0, s0 = self._bn(R, "bn0_1", "lrelu", stats=cur_st)
                cur_st = None
                a1, a1st = self._conv_bn(h0, "conv1x1_1.weight")
                h1, s1 = self._bn(a1, "bn1", "lrelu", stats=a1st)
                a2, a2st = self._conv_bn(h1, "conv1.weight", 1, 1, 1)
                h2, s2 = self._bn(a2, "bn0_2", "lrelu", stats=a2st)
                a3 = T.conv2d_nhwc(h2, self.P["conv1x1_2.weight"], add=R)
                c2, c2st = self._conv_bn(a3, "conv2.weight", 1, 1, 1, bias=self.P["conv2.bias"])
                h3, s3 = self._bn(c2, "bn2", "lrelu", stats=c2st)
                head = T.conv2d_nhwc(h3, self.P["conv3.weight"], bias=self.P["conv3.bias"], act=L.PPN_ACT_SIGMOID,
                                     nchw_f32=True)
                tape.append(("head", u, dict(R=R, h0=h0, s0=s0, a1=a1, h1=h1, s1=s1, a2=a2, h2=h2, s2=s2, a3=a3,
                                             c2=c2, h3=h3, s3=s3, head=head)))
                cur = head
        self.num_batches_tracked += 1
        self._tape = tape
        # flat-buffer offset of the unit that follows each unit in forward order (for the bucketed exchange)
        self._next_offset = {}
        nxt = self.flat.numel()
        for kind, u, _ in reversed(tape):
            self._next_offset[id(u)] = nxt
            nxt = self._unit_offset(kind, u)
        return cur

    # ---- backward ------------------------------------------------------------------------------------------------
    def _head_backward(self, c, grad_head, probe_only: bool, channels_used: Optional[int] = None, so=None, next_bn=None):
        """Backward of the head unit.  probe_only: stop at conv1.weight and return its gradient (GradNorm);
        channels_used: only the first so many head channels carry a gradient (6K for the unary losses)."""
        lib = L.load()
        keep = not probe_only
        head = c["head"]
        B, Ch, Ho, Wo = head.shape
        used = Ch if channels_used is None else channels_used
        assert not keep or used == Ch
        cpad = (used + 63) // 64 * 64
        if keep and so is not None and so.get("dz") is not None:
            dz, dbsum = so["dz"]                                      # PPNLoss.forward_backward_dz did this step already
            self.G["conv3.bias"].copy_(dbsum.sum(0)[:Ch])
        else:
            dz = torch.empty(B, Ho, Wo, cpad, dtype=self.tdt, device=self.device)
            dbias3 = self.G["conv3.bias"] if keep else None
            L.check(lib.ppn_head_grad(self.compute_dtype, head.data_ptr(), grad_head.data_ptr(), B, Ch, Ho * Wo, used,
                                      cpad, dz.data_ptr(), dbias3.data_ptr() if keep else None, L.current_stream_ptr()),
                    "ppn_head_grad")
        w3p = self._w3_padded(used)
        if keep:
            def wg3():
                dw3 = T.conv_wgrad(c["h3"], dz, 1)
                self.G["conv3.weight"].copy_(dw3[:Ch])
            self._on_side(wg3, c["h3"], dz)
        # (the probe passes keep the separate reduction pass: their stacked form -- _stacked_unary_probe_grads, several gradient
        # streams per launch -- has no fused sums, and the two forms stay bit-identical)
        dh3, st3 = self._dgrad_bn(dz, w3p, (Ho, Wo), 1, 1, 0, c["c2"], "bn2", "lrelu", c["s3"], fuse=keep)
        dc2 = self._bn_bwd(c["c2"], dh3, "bn2", "lrelu", c["s3"], keep=keep, stats=st3)
        if keep:
            ws = T._workspace(dc2.shape[-1], self.device)
            L.check(lib.ppn_colsum(self.compute_dtype, dc2.data_ptr(), dc2.numel() // dc2.shape[-1], dc2.shape[-1],
                                   self.G["conv2.bias"].data_ptr(), ws.data_ptr(), L.current_stream_ptr()),
                    "ppn_colsum")
            self._wgrad("conv2.weight", c["a3"], dc2, 3, 1, 1, 1)
        da3 = T.conv_dgrad(dc2, self.P["conv2.weight"], (Ho, Wo), 1, 1, 1)
        if keep:
            self._wgrad("conv1x1_2.weight", c["h2"], da3, 1)
        dh2, st2 = self._dgrad_bn(da3, "conv1x1_2.weight", (Ho, Wo), 1, 1, 0, c["a2"], "bn0_2", "lrelu", c["s2"], fuse=keep)
        da2 = self._bn_bwd(c["a2"], dh2, "bn0_2", "lrelu", c["s2"], keep=keep, stats=st2)
        if probe_only:
            return T.conv_wgrad(c["h1"], da2, 3, 1, 1, 1)
        self._wgrad("conv1.weight", c["h1"], da2, 3, 1, 1, 1)
        second = so is not None and "head" in so
        if second:                                       # the second-order adjoint is added to dh1 below: no sums of the bare dh1
            dh1, st1 = T.conv_dgrad(da2, self.P["conv1.weight"], (Ho, Wo), 1, 1, 1), None
        else:
            dh1, st1 = self._dgrad_bn(da2, "conv1.weight", (Ho, Wo), 1, 1, 1, c["a1"], "bn1", "lrelu", c["s1"])
        skip = da3
        if so is not None and "head" in so:
            # GradNorm's Lgrad.backward(): the second-order adjoints at h1 and at the skip tensor join the first-order
            # ones here, so everything upstream is back-propagated once
            if so.get("launch_probes") is not None:
                so.pop("launch_probes")()
            h1_sum, r_bar = self._second_order_tail(c, so, dh1)
            if h1_sum is not None:
                dh1 = h1_sum                             # first-order seed + second-order adjoint (added in a conv epilogue)
                skip = da3 + r_bar
        da1 = self._bn_bwd(c["a1"], dh1, "bn1", "lrelu", c["s1"], stats=st1)
        self._wgrad("conv1x1_1.weight", c["h0"], da1, 1)
        dh0, st0 = self._dgrad_bn(da1, "conv1x1_1.weight", (Ho, Wo), 1, 1, 0, c["R"], "bn0_1", "lrelu", c["s0"])
        # next_bn: the BatchNorm of the unit below that takes this gradient as its dy -> (dR, ConvStats)
        return self._bn_bwd(c["R"], dh0, "bn0_1", "lrelu", c["s0"], dx_add=skip, stats=st0, next_bn=next_bn)

    # ---- second order ------------------------------------------------------------------------------------------------
    def _second_order_tail(self, c, so, dh1=None):
        """d Lgrad / d theta restricted to what the probe weight W = conv1.weight can influence: the head's tail.
        so = dict(head, targets, losses, coeff, unary=[dL_i/dW for i < 4]).  Accumulates into the tail parameters'
        gradients and returns the adjoints (at h1, at the skip tensor R) to be added to the first-order seeds.
        Derivation: DESIGN.md section 7 item 2; building blocks: train.bn_tangent / bn_dual_backward, PPNLoss.dual."""
        head, targets = so["head"], so["targets"]
        coeff = so["coeff"] if "coeff" in so else so["coeff_fn"]()
        B, Ch, Ho, Wo = head.shape
        main = torch.cuda.current_stream(self.device)
        if self._side is not None:
            main.wait_stream(self._side)                 # the first-order tail gradients this pass accumulates into
        if so.get("stream") is not None:
            main.wait_stream(so["stream"])                # the unary probe gradients
        gw = [g.contiguous() for g in so["unary"]]
        # The limb probe gradient by linearity, the five probe norms, the trust test of the bf16 mode and GradNorm's
        # C_i / signs: ONE kernel pass over the five weight-sized tensors and ONE host read-back (the 5 x 9.4 MB tensors
        # went through ~45 tiny torch launches and three read-backs, each draining the queue: 0.8 ms of an idle GPU per
        # iteration, tools/train_timeline.py).
        # Round 4: that read-back no longer idles the GPU either.  The forward-mode half of the tail (tangents through
        # bn0_2 .. conv3) is linear in the unit directions v_i = g_i / ||g_i|| and needs none of the host values (kappa
        # and the list of active streams only enter at the loss), so it is ENQUEUED FIRST, for all five streams, with
        # v_i divided on the device; the host then waits for the 17 floats (pinned, asynchronous copy) while the GPU
        # works through ~1.5 ms of tangent convolutions, and enqueues the reverse half behind them.  If the host values
        # say otherwise (a stream inactive, the limb remainder untrusted: rare), the speculative tensors are dropped and
        # the tangents are rebuilt for the streams that are active -- bitwise the results of the non-speculative order (both multiply by
        # the SAME device-computed reciprocal, see inv_dev).
        total = self.G["conv1.weight"]
        trusted = coeff[4] > 1e-3 * max(coeff)
        spec = None
        if trusted:
            gw4, st = T.probe_stats(gw, total.contiguous(), coeff)
            # 1 / ||g_i|| is computed ONCE, on the device, and travels to the host with the other scalars: the speculative
            # tangents (g_i * inv_i with the device value) and the rebuilt ones (g_i * inv_i with the same bits read back) are then
            # bitwise equal, so whether a step keeps its speculation never changes its result
            inv_dev = torch.reciprocal(torch.sqrt(st[:5]))
            dev_vec = torch.cat([st, so["losses"].reshape(5), self.base.reshape(5), inv_dev])
            if self._so_pin is None:
                self._so_pin = torch.empty(22, dtype=torch.float32, pin_memory=True)
            self._so_pin.copy_(dev_vec, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
            if self._speculate_tail:
                spec = self._tail_tangents(c, [(g * inv_dev[i]) for i, g in enumerate(gw + [gw4])], list(range(5)))
            ev.synchronize()
            host = self._so_pin.tolist()
            if self.compute_dtype != L.PPN_F32:
                # bf16: `total` and the unary probes come from differently rounded passes (relative noise ~2^-8 each);
                # a remainder that is not clearly above that noise cannot be trusted (see _limb_probe)
                trusted = host[5] > (16.0 * 2.0 ** -8) ** 2 * host[6]
        inv_h = None
        if trusted:
            ss = host[:5]
            inv_h = host[17:22]
            gw.append(gw4)
        else:
            spec = None
            _, g4 = self.criterion.forward_backward(head, targets, coeff=[0.0, 0.0, 0.0, 0.0, 1.0])
            gw.append(self.probe_grad(g4))
            host = torch.cat([torch.cat([T.sumsq(g.contiguous().view(-1)) for g in gw]), so["losses"].reshape(5),
                              self.base.reshape(5)]).tolist()
            ss = host[:5]
            host = ss + [0.0, 0.0] + host[5:]
        import numpy as _np
        f32 = _np.float32
        gn_np = _np.sqrt(_np.asarray(ss, dtype=f32))
        gn = torch.from_numpy(gn_np).to(self.device, non_blocking=True)
        so["gnorm"] = gn
        w_np = _np.asarray(self.task.host_weights(), dtype=f32)
        loss_np, base_np = _np.asarray(host[7:12], dtype=f32), _np.asarray(host[12:17], dtype=f32)
        G = _np.abs(w_np) * gn_np                                                 # main.py:717-721
        lhat = w_np * loss_np / base_np
        Cc = G.mean(dtype=f32) * (lhat / lhat.mean(dtype=f32)) ** f32(self.task.alpha)   # main.py:726-753 (constant)
        kappa = (_np.sign(G - Cc) * _np.sign(w_np) * _np.abs(w_np)).tolist()      # d Lgrad / d||g_i||, times w_i
        gn_h = gn_np.tolist()
        k6 = 6 * cfg.K
        P, Gd = self.P, self.G
        act = [i for i in range(5) if kappa[i] != 0.0 and gn_h[i] != 0.0]
        if not act:
            return None, None
        n = len(act)
        if spec is None or act != [0, 1, 2, 3, 4]:
            if inv_h is None:                                     # (the untrusted path has no device reciprocals: exact host ones)
                inv_h = [float(f32(1.0) / f32(v)) if v != 0.0 else 0.0 for v in gn_h]
            spec = self._tail_tangents(c, [(gw[i] * inv_h[i]).contiguous() for i in act], act)
        vs, u2, TH2, TA3, TC2, TH3, tz_groups = spec

        # ---- head space per stream: gradient and Hessian-vector product of loss i through the sigmoid.  The unary
        # losses only touch the first 6K channels: their streams share 128-channel conv3 launches.
        # The PRIMAL adjoint chain is a single stream: everything between the loss and h1 that the primal adjoints pass
        # through is linear in them (convolutions, the ordinary BN backward) and the tail only needs their sum over the
        # streams, so they are summed where they are born (zsum: the weight gradient needs it anyway) and travel as ONE
        # batch-B tensor beside the n tangent adjoints: (1 + n) B rows per convolution instead of 2 n B.
        HB = torch.empty((1 + n) * B, *TH3.shape[1:], dtype=self.tdt, device=self.device)   # [sum of primal adj | tangent adj] at h3
        H3sum, TH3bar = HB[:B], HB[B:]
        first_group = True
        for js, used, w3u, th3, t_z in tz_groups:                                 # t_z: logit tangents [m*B, used, H, W]
            m = len(js)
            if used == Ch and self.tdt in (torch.float32, torch.bfloat16):
                # the limb stream (one stream, the whole head): dual seeds, relayout and the pixel sums of zbar in
                # one pass -- no f32 head-layout zbar / tzbar (2 x 541 MB written and re-read at batch 32)
                zb, tzb, zpart = self.criterion.limb_dual_nhwc(head, t_z, targets, kappa[4], self.tdt)
                zbias = zpart.sum(0)[:used]
            else:
                zbar, tzbar = torch.empty_like(t_z), torch.empty_like(t_z)
                for q, j in enumerate(js):
                    ci = [0.0] * 5
                    ci[act[j]] = kappa[act[j]]
                    sl = slice(q * B, (q + 1) * B)
                    self.criterion.dual(head, t_z[sl], targets, ci, unary_only=used != Ch, out=(zbar[sl], tzbar[sl]))
                zb, tzb = T.nchw_to_nhwc(zbar, self.tdt), T.nchw_to_nhwc(tzbar, self.tdt)
                zbias = zbar.sum((0, 2, 3))
            cpad = zb.shape[-1]
            w3p = self._w3_padded(used)
            zsum = zb if m == 1 else zb.view(m, B, Ho, Wo, cpad).sum(0)          # (m == 1: a 280 MB no-op reduction)
            # Weight gradients are leaves here too: on the side stream (idle during the tail) instead of ~0.9 ms of the
            # main stream; they accumulate into gradients the side stream wrote, so the order is the stream's own.
            def wg3(zsum=zsum, th3=th3, tzb=tzb, used=used):
                dw3 = T.conv_wgrad(c["h3"], zsum, 1)                              # primal stream: same h3 for all
                T.conv_wgrad(th3, tzb, 1, out=dw3, accumulate=True)              # tangent stream: stacked batch
                Gd["conv3.weight"][:used] += dw3[:used]
            self._tail_side(wg3, c["h3"], zsum, th3, tzb)
            Gd["conv3.bias"][:used] += zbias
            T.conv_dgrad(zsum, w3p, (Ho, Wo), add=None if first_group else H3sum, out=H3sum)
            first_group = False
            TH3bar[js[0] * B:(js[-1] + 1) * B] = T.conv_dgrad(tzb, w3p, (Ho, Wo))
        # ---- reverse pass over the dual tail --------------------------------------------------------------------------
        # Every BN adjoint covers all streams in one set of launches (the stream is a grid dimension; the ordinary backward
        # of the summed primal adjoint and of the n tangent adjoints are 1 + n streams of one call; the n dual terms
        # accumulate into the one primal result): 8 + n launches per BN layer where the stream-by-stream form took 12 n.
        both = torch.empty((1 + n) * B, *TC2.shape[1:], dtype=self.tdt, device=self.device)   # [sum of primal adj | tangent adj]
        c2sum, TC2bar, dg, db = T.bn_dual_backward_summed(c["c2"], TC2, HB, P["bn2.weight"], P["bn2.bias"], c["s3"], "lrelu",
                                                          n, out_both=both)
        Gd["bn2.weight"] += dg
        Gd["bn2.bias"] += db
        Gd["conv2.bias"] += c2sum.float().sum((0, 1, 2))
        def wg2(c2sum=c2sum, TC2bar=TC2bar):
            T.conv_wgrad(c["a3"], c2sum, 3, 1, 1, 1, out=Gd["conv2.weight"], accumulate=True)
            T.conv_wgrad(TA3, TC2bar, 3, 1, 1, 1, out=Gd["conv2.weight"], accumulate=True)
        self._tail_side(wg2, c["a3"], c2sum, TA3, TC2bar)
        both = T.conv_dgrad(both, P["conv2.weight"], (Ho, Wo), 1, 1, 1)
        a3sum, TA3bar = both[:B], both[B:]
        def wg12(a3sum=a3sum, TA3bar=TA3bar):
            T.conv_wgrad(c["h2"], a3sum, 1, out=Gd["conv1x1_2.weight"], accumulate=True)
            T.conv_wgrad(TH2, TA3bar, 1, out=Gd["conv1x1_2.weight"], accumulate=True)
        self._tail_side(wg12, c["h2"], a3sum, TH2, TA3bar)
        both = T.conv_dgrad(both, P["conv1x1_2.weight"], (Ho, Wo))                # [sum of H2bar | TH2bar]
        a2sum, U_bar, dg, db = T.bn_dual_backward_summed(c["a2"], u2, both, P["bn0_2.weight"], P["bn0_2.bias"], c["s2"],
                                                         "lrelu", n)
        Gd["bn0_2.weight"] += dg
        Gd["bn0_2.bias"] += db
        self._tail_side(lambda: T.conv_wgrad(c["h1"], a2sum, 3, 1, 1, 1, out=Gd["conv1.weight"], accumulate=True),
                        c["h1"], a2sum)
        # adjoint at h1: through W (primal) and through u_i = conv(h1, v_i) per stream; every convolution adds the sum so
        # far in its epilogue (f32, one rounding) instead of a separate elementwise pass
        # (dh1: the first-order gradient at h1 joins here, in the first epilogue of the chain, instead of a pass of its own)
        h1_bar = T.conv_dgrad(a2sum, P["conv1.weight"], (Ho, Wo), 1, 1, 1, add=dh1)
        for j, v in enumerate(vs):
            h1_bar = T.conv_dgrad(U_bar[j * B:(j + 1) * B], v, (Ho, Wo), 1, 1, 1, add=h1_bar)
        r_bar = a3sum
        return h1_bar, r_bar

    def _tail_tangents(self, c, vs, act):
        """Forward-mode half of the second-order tail for the streams `act` with unit directions `vs` on W = conv1.weight:
        tangents through bn0_2 -> conv1x1_2 -> conv2 -> bn2 -> conv3.  Every stream is linear in its tangent, so the
        streams are stacked along the batch dimension for the convolutions (one launch for all of them); the BN tangents
        need per-stream batch statistics: the stream is a grid dimension of their kernels.  The unary losses only touch the first 6K head
        channels: their streams share 128-channel conv3 launches.  Needs no host value (see _second_order_tail)."""
        P = self.P
        n = len(act)
        B = c["h1"].shape[0]
        Ch = c["head"].shape[1]
        k6 = 6 * cfg.K
        vs = [v.contiguous() for v in vs]
        u2 = torch.empty(n * B, *c["a2"].shape[1:], dtype=self.tdt, device=self.device)   # the streams back to back
        for j, v in enumerate(vs):
            T.conv2d_nhwc(c["h1"], v, 1, 1, 1, out=u2[j * B:(j + 1) * B])
        # the BN tangents of all streams in one set of launches (ppn_bn_*_streams: per-stream batch statistics, shared x)
        TH2 = T.bn_tangent(c["a2"], u2, P["bn0_2.weight"], P["bn0_2.bias"], c["s2"], "lrelu", nstreams=n)
        TA3 = T.conv2d_nhwc(TH2, P["conv1x1_2.weight"])
        TC2 = T.conv2d_nhwc(TA3, P["conv2.weight"], 1, 1, 1)
        TH3 = T.bn_tangent(c["c2"], TC2, P["bn2.weight"], P["bn2.bias"], c["s3"], "lrelu", nstreams=n)
        tz_groups = []
        for js, used in (([j for j, i in enumerate(act) if i < 4], k6), ([j for j, i in enumerate(act) if i == 4], Ch)):
            if not js:
                continue
            w3u = P["conv3.weight"] if used == Ch else P["conv3.weight"][:used].contiguous()
            th3 = TH3[js[0] * B:(js[-1] + 1) * B]                                 # the group's streams are adjacent
            tz_groups.append((js, used, w3u, th3, T.conv2d_nhwc(th3, w3u, nchw_f32=True)))
        return vs, u2, TH2, TA3, TC2, TH3, tz_groups

    def _w3_padded(self, used: int) -> torch.Tensor:
        """conv3.weight[:used] zero-padded to a multiple of 64 rows (what the input-gradient convolution of the 7605- /
        108-channel head reads as its input width), in a PERSISTENT buffer registered with the packed-weight cache: padded and
        packed once per parameter version instead of a zeros + copy + 33 us pack at each of its three uses per iteration.
        forward() refreshes the existing buffers before it repacks."""
        ent = self._w3p.get(used)
        if ent is None:
            w3 = self.P["conv3.weight"]
            cpad = (used + 63) // 64 * 64
            buf = torch.zeros(cpad, w3.shape[1], 1, 1, dtype=torch.float32, device=self.device)
            ent = self._w3p[used] = [-1, buf, T.register_param_storage(buf)]
        if ent[0] != T._param_version[0]:
            ent[1][:used].copy_(self.P["conv3.weight"][:used])
            ent[0] = T._param_version[0]
        return ent[1]

    def _unit_offset(self, kind, u) -> int:
        """First element of the flat buffer that belongs to this unit (its parameters are contiguous)."""
        if kind == "head":
            return self.offset["conv1x1_1.weight"]
        if kind in ("basic", "bottleneck"):
            return self.offset[u.prefix + ".conv1.weight"]
        return self.offset[f"{u.prefix}.{u.conv_idx}.weight"]

    def backward(self, grad_head: torch.Tensor, exchange: Optional["T.BucketedAllReduce"] = None, so=None):
        """d(sum_i coeff_i L_i)/d(theta) into self.grad, given d/d(head) from the loss kernel.  `exchange`: buckets
        of the flat buffer are all-reduced as soon as the units that own them are done (the buffer is in forward
        order, the backward completes it from its end)."""
        if self._tape is None:
            raise RuntimeError("backward() needs a forward() first")

        def join_side():
            if self._side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._side)

        g = grad_head
        gst = None                                       # ConvStats of g where g is the dy of the NEXT unit's leading BatchNorm
        order = list(reversed(self._tape))
        for ui, (kind, u, c) in enumerate(order):
            # the unit BEFORE this one in forward order: a conv-BN-ReLU unit's BatchNorm takes this unit's input gradient as its dy
            below = order[ui + 1] if ui + 1 < len(order) else None
            below_cbr = below is not None and below[0] == "cbr"
            # ... and a block with a projection shortcut takes it as the dy of the shortcut's BatchNorm: (x2, prefix2, act2, saved2)
            if below_cbr:
                below_bn = (below[2]["y"], f"{below[1].prefix}.{below[1].conv_idx + 1}", "relu", below[2]["saved"])
            elif below is not None and below[0] == "basic" and below[1].downsample:
                below_bn = (below[2]["dsy"], below[1].prefix + ".downsample.1", "none", below[2]["s3"])
            else:
                below_bn = None
            if exchange is not None and kind != "head":
                # every unit AFTER this one in forward order has been processed: its slice of the buffer is final
                nxt = self._next_offset[id(u)]
                exchange.ready(nxt, before_issue=join_side)
            if kind == "head":
                if below_bn is not None:
                    g, gst = self._head_backward(c, g, probe_only=False, so=so, next_bn=below_bn)
                else:
                    g = self._head_backward(c, g, probe_only=False, so=so)
                if exchange is not None and exchange.enabled:
                    # the GradNorm probes need THIS rank's d loss/d conv1.weight; keep it before its bucket is summed
                    join_side()
                    self._conv1_local = self.G["conv1.weight"].clone()
            elif kind == "basic":
                p = u.prefix
                hw = c["x"].shape[1:3]
                if u.downsample:
                    # first: its BatchNorm may take the sums the unit above folded while it wrote g (gst), and they sit in the
                    # workspace every BatchNorm / statistics epilogue of this channel count uses
                    dds = self._bn_bwd(c["dsy"], g, p + ".downsample.1", "none", c["s3"], stats=gst)
                    self._wgrad(p + ".downsample.0.weight", c["x"], dds, 1, u.stride, 1, 0)
                    dxr = T.conv_dgrad(dds, self.P[p + ".downsample.0.weight"], hw, u.stride, 1, 0)
                else:
                    dxr = g
                gst = None
                self._wgrad(p + ".conv2.weight", c["b"], g, 3, 1, u.dil[1], u.dil[1])
                db, dbst = self._dgrad_bn(g, p + ".conv2.weight", c["b"].shape[1:3], 1, u.dil[1], u.dil[1],
                                          c["c1"], p + ".bn2", "relu", c["s2"])
                dc1 = self._bn_bwd(c["c1"], db, p + ".bn2", "relu", c["s2"], stats=dbst)
                self._wgrad(p + ".conv1.weight", c["a"], dc1, 3, u.stride, u.dil[0], u.dil[0])
                da, dast = self._dgrad_bn(dc1, p + ".conv1.weight", hw, u.stride, u.dil[0], u.dil[0],
                                          c["x"], p + ".bn1", "relu", c["s1"])
                if below_bn is not None:
                    g, gst = self._bn_bwd(c["x"], da, p + ".bn1", "relu", c["s1"], dx_add=dxr, stats=dast, next_bn=below_bn)
                else:
                    g = self._bn_bwd(c["x"], da, p + ".bn1", "relu", c["s1"], dx_add=dxr, stats=dast)
            elif kind == "bottleneck":
                p = u.prefix
                hw = c["x"].shape[1:3]
                dsum = T.relu_mask(c["out"], g)                              # through relu(z3 + r)
                dy3 = self._bn_bwd(c["y3"], dsum, p + ".bn3", "none", c["s3"])
                self._wgrad(p + ".conv3.weight", c["h2"], dy3, 1)
                dh2, dh2st = self._dgrad_bn(dy3, p + ".conv3.weight", c["h2"].shape[1:3], 1, 1, 0, c["y2"], p + ".bn2", "relu", c["s2"])
                dy2 = self._bn_bwd(c["y2"], dh2, p + ".bn2", "relu", c["s2"], stats=dh2st)
                self._wgrad(p + ".conv2.weight", c["h1"], dy2, 3, u.stride, u.dil[1], u.dil[1])
                dh1, dh1st = self._dgrad_bn(dy2, p + ".conv2.weight", c["h1"].shape[1:3], u.stride, u.dil[1], u.dil[1],
                                            c["y1"], p + ".bn1", "relu", c["s1"])
                dy1 = self._bn_bwd(c["y1"], dh1, p + ".bn1", "relu", c["s1"], stats=dh1st)
                self._wgrad(p + ".conv1.weight", c["x"], dy1, 1)
                if u.downsample:
                    dyd = self._bn_bwd(c["yd"], dsum, p + ".downsample.1", "none", c["sd"])
                    self._wgrad(p + ".downsample.0.weight", c["x"], dyd, 1, u.stride, 1, 0)
                    dxr = T.conv_dgrad(dyd, self.P[p + ".downsample.0.weight"], hw, u.stride, 1, 0)
                else:
                    dxr = dsum
                g = T.conv_dgrad(dy1, self.P[p + ".conv1.weight"], hw, add=dxr)
                gst = None
            else:  # cbr
                wn = f"{u.prefix}.{u.conv_idx}.weight"
                bnp = f"{u.prefix}.{u.conv_idx + 1}"
                d = u.dil[0]
                dy = self._bn_bwd(c["y"], g, bnp, "relu", c["saved"], stats=gst)
                gst = None
                if u.k == 7:
                    def wg0(x8=c["x"], dy=dy, wn=wn):
                        dw8 = T.conv_wgrad(x8, dy, 7, 1, 1, 3)            # [16, 4 | 8, 7, 7]; input channels 3.. are zero
                        self.G[wn].copy_(dw8[:, :3])
                    # the LAST weight gradient of the pass: on the main stream (idle from here on) beside the side stream's
                    # layer1 / layer2 weight gradients instead of behind them (PPN_TRAIN_WG0_MAIN=0: on the side stream)
                    if self._wg0_main:
                        wg0()
                    else:
                        self._on_side(wg0, c["x"], dy)
                    g = None                                               # the input needs no gradient
                else:
                    self._wgrad(wn, c["x"], dy, 3, u.stride, d, d)
                    if below_bn is not None:                               # g is the dy of a BatchNorm of the unit below
                        g, gst = self._dgrad_bn(dy, wn, c["x"].shape[1:3], u.stride, d, d, below_bn[0], below_bn[1],
                                                below_bn[2], below_bn[3])
                    else:
                        g = T.conv_dgrad(dy, self.P[wn], c["x"].shape[1:3], u.stride, d, d)
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)   # every weight gradient has landed
        return self.grad

    def probe_grad(self, grad_head: torch.Tensor, channels_used: Optional[int] = None) -> torch.Tensor:
        """dL/dW for W = head conv1.weight (params[-13]) from d L/d(head): the partial backward of main.py:704-708."""
        kind, _, c = self._tape[-1]
        assert kind == "head"
        return self._head_backward(c, grad_head, probe_only=True, channels_used=channels_used)

    def _stacked_unary_probe_grads(self, head, targets, scratch):
        """[dL_i/dW for i < 4] (W = conv1.weight, the four unary losses): the four probe passes of probe_grad() with their
        convolutions STACKED along the batch dimension -- one conv3 / conv2 / conv1x1_2 input-gradient launch for all four
        (they are per-image operations) -- the BN backward passes (per-pass batch statistics) take the pass as a grid dimension
        and only the four weight gradients run pass by pass: 25 launches instead of 68, and three well-filled convolution launches instead of twelve
        at a quarter of the GPU.  Same arithmetic per element as probe_grad(): the results are bit-identical."""
        kind, _, c = self._tape[-1]
        assert kind == "head"
        lib = L.load()
        B, Ch, Ho, Wo = head.shape
        k6 = 6 * cfg.K
        cpad = (k6 + 63) // 64 * 64
        dz4 = torch.empty(4 * B, Ho, Wo, cpad, dtype=self.tdt, device=self.device)
        for i in range(4):
            self.criterion.unary_backward(head, targets, [1.0 if j == i else 0.0 for j in range(4)], out=scratch)
            L.check(lib.ppn_head_grad(self.compute_dtype, head.data_ptr(), scratch.data_ptr(), B, Ch, Ho * Wo, k6, cpad,
                                      dz4[i * B:(i + 1) * B].data_ptr(), None, L.current_stream_ptr()), "ppn_head_grad")
        dh3 = T.conv_dgrad(dz4, self._w3_padded(k6), (Ho, Wo))
        # the BN backward of the four passes in one set of launches (per-pass batch statistics: the pass is a grid dimension)
        dc2, _, _ = T.bn_train_backward(c["c2"], dh3, self.P["bn2.weight"], self.P["bn2.bias"], c["s3"], act="lrelu", nstreams=4)
        da3 = T.conv_dgrad(dc2, self.P["conv2.weight"], (Ho, Wo), 1, 1, 1)
        dh2 = T.conv_dgrad(da3, self.P["conv1x1_2.weight"], (Ho, Wo))
        da2, _, _ = T.bn_train_backward(c["a2"], dh2, self.P["bn0_2.weight"], self.P["bn0_2.bias"], c["s2"], act="lrelu",
                                        nstreams=4)
        return [T.conv_wgrad(c["h1"], da2[i * B:(i + 1) * B], 3, 1, 1, 1) for i in range(4)]

    def _unary_probes(self, head, targets, coeff, scratch):
        """The four cheap probe passes: (gnorm[0:4] f32[4], sum_{i<4} coeff_i dL_i/dW).  Independent of backward()."""
        k6 = 6 * cfg.K
        gn = torch.empty(4, dtype=torch.float32, device=self.device)
        acc = torch.zeros_like(self.G["conv1.weight"])
        for i in range(4):
            self.criterion.unary_backward(head, targets, [1.0 if j == i else 0.0 for j in range(4)], out=scratch)
            gw = self.probe_grad(scratch, channels_used=k6)
            gn[i:i + 1] = T.sumsq(gw.view(-1)).sqrt()
            acc.add_(gw, alpha=float(coeff[i]))
        return gn, acc

    def _limb_probe(self, total, acc, coeff, head, targets) -> torch.Tensor:
        """dL_4/dW (limb loss) = (sum_i coeff_i dL_i/dW - sum_{i<4} coeff_i dL_i/dW) / coeff_4 by linearity of the
        backward pass, or the direct fifth pass when that remainder cannot be trusted: coeff_4 too small to divide by,
        or -- bf16 mode, where `total` and `acc` come from differently rounded passes (relative noise ~2^-8 each) -- a
        remainder that is not clearly above the rounding noise of the total."""
        if coeff[4] > 1e-3 * max(coeff):
            rest = total - acc
            trusted = True
            if self.compute_dtype != L.PPN_F32:
                # both norms in one device tensor: ONE read-back (a host sync on the main stream) instead of two
                n_rest, n_tot = torch.stack([T.sumsq(rest.contiguous().view(-1)).reshape(()),
                                             T.sumsq(total.contiguous().view(-1)).reshape(())]).tolist()
                trusted = n_rest > (16.0 * 2.0 ** -8) ** 2 * n_tot
            if trusted:
                return rest / float(coeff[4])
        _, g4 = self.criterion.forward_backward(head, targets, coeff=[0.0, 0.0, 0.0, 0.0, 1.0])
        return self.probe_grad(g4)

    def probe_norms(self, head, targets, coeff, ghead, unary=None) -> torch.Tensor:
        """gnorm_i = ||dL_i/dW||_2, i = 0..4, AFTER backward() ran with `coeff` (so self.G['conv1.weight'] holds
        sum_i coeff_i dL_i/dW).  The four unary losses touch only the first 6K head channels, so their probe passes
        cost a 128-channel conv3 backward instead of a 7616-channel one; the backward pass is linear in the head
        gradient (BN statistics are fixed by the forward), hence the limb loss's probe gradient is what remains:
            dL_4/dW = (sum_i coeff_i dL_i/dW - sum_{i<4} coeff_i dL_i/dW) / coeff_4
        -- no fifth pass.  Falls back to the direct pass when coeff_4 is too small to divide by.
        `unary`: the result of _unary_probes when it already ran (train_step runs it beside backward())."""
        gn4, acc = unary if unary is not None else self._unary_probes(head, targets, coeff, ghead)
        gn = torch.empty(5, dtype=torch.float32, device=self.device)
        gn[:4] = gn4
        local = self._conv1_local if self._conv1_local is not None else self.G["conv1.weight"]
        rest = self._limb_probe(local, acc, coeff, head, targets)
        gn[4:5] = T.sumsq(rest.contiguous().view(-1)).sqrt()
        return gn

    # ---- one iteration ------------------------------------------------------------------------------------------
    def _init_base(self, losses: torch.Tensor, group=None):
        """No get_baseloss() / explicit base before the first step: fall back to this minibatch's train-mode losses
        (the reference uses eval-mode losses averaged over the loader, main.py:578-621, which changes the GradNorm
        targets C_i) -- loudly, and identical on every rank."""
        import warnings
        import torch.distributed as dist
        warnings.warn("PPNTrainer: GradNorm base losses not set (call get_baseloss() or assign trainer.base as "
                      "main.py:403 does); using the first minibatch's train-mode losses", RuntimeWarning, stacklevel=3)
        base = losses.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(base, op=dist.ReduceOp.SUM, group=group)
            base /= dist.get_world_size(group)
        self.base = base

    def local_pass(self, x: torch.Tensor, targets: Dict[str, torch.Tensor], group=None, w_step=None):
        """Everything of main.py:664-759 that happens on this rank's shard: train-mode forward, PPNLoss fwd+bwd, the
        backward pass (with the second-order term when enabled) into self.grad -- its 32 MB buckets all-reduced
        (SUM) over `group` while the backward runs -- and the five probe-gradient norms.
        Returns (losses f32[5], gnorm f32[5], scale = 1/world for the optimiser).
        `w_step(losses, gnorm)` (train_step passes it): where the exchange is enabled and the probe norms are known before
        the last bucket goes out (the second-order path), it is called right before that bucket is issued so that the task
        weights' local step can ride on it; it is NOT called otherwise (the caller then steps and all-reduces them)."""
        head = self.forward(x)
        ev_head = torch.cuda.Event()
        ev_head.record(torch.cuda.current_stream(self.device))        # the probe passes need the head, not the loss
        if self.second_order:
            # the loss kernels read w_i / 5 from the device; the host copy (needed by the second-order tail only) is
            # fetched there, when the previous iteration is long over -- nothing here waits for the GPU
            if self.tdt in (torch.float32, torch.bfloat16):
                # ... and hand the gradient over already differentiated through the sigmoid, in the NHWC layout conv3's
                # backward reads (no f32 head-layout gradient: 17 MB per image written once and read twice)
                losses, dz_pre, dbsum = self.criterion.forward_backward_dz(head, targets, (self.task.w, 5.0), self.tdt)
                ghead = None
            else:
                losses, ghead = self.criterion.forward_backward(head, targets, coeff_dev=(self.task.w, 5.0))
                dz_pre = None
            coeff = None
        elif self._probe_stream is not None and self.tdt in (torch.float32, torch.bfloat16):
            # first-order step, same plumbing: device-side coefficients, gradient straight to NHWC; the host values of
            # the task weights are read after the backward, where nothing waits for them
            losses, dz_pre, dbsum = self.criterion.forward_backward_dz(head, targets, (self.task.w, 5.0), self.tdt)
            ghead, coeff = None, None
        else:
            w = self.task.host_weights()                                 # 5 floats (coefficients of the loss kernel)
            losses, ghead = self.criterion.forward_backward(head, targets, coeff=[v / 5.0 for v in w])
            coeff = [v / 5.0 for v in w]
            dz_pre = None
        if self.base is None:
            self._init_base(losses, group)
        unary = None
        if self.second_order:
            # the probe gradients themselves (not only their norms) steer the second-order pass inside backward()
            if self._probe_scratch is None or self._probe_scratch.shape != head.shape:
                self._probe_scratch = torch.empty_like(head)
            grads = []
            main = torch.cuda.current_stream(self.device)
            pst = self._probe_stream if self._probe_stream is not None else main
            ev = ev_head                                  # probes start beside the loss kernels (0.45 ms of head start)

            def launch_probes():
                # The four cheap probe passes run beside the head backward.  They only need the forward (the event
                # above), but the HOST enqueues them after the head's first-order backward (_head_backward calls this
                # just before the second-order tail): their ~66 small launches take the host ~1 ms, during which the
                # main stream sat idle when they were enqueued first (rocprofv3 kernel trace, round 3).
                with torch.cuda.stream(pst):
                    pst.wait_event(ev)
                    if self._stacked_probes:
                        gs = self._stacked_unary_probe_grads(head, targets, self._probe_scratch)
                    else:
                        gs = []
                        for i in range(4):
                            self.criterion.unary_backward(head, targets, [1.0 if j == i else 0.0 for j in range(4)],
                                                          out=self._probe_scratch)
                            gs.append(self.probe_grad(self._probe_scratch, channels_used=6 * cfg.K))
                    for gi in gs:
                        gi.record_stream(main)
                        grads.append(gi)

            so = dict(head=head, targets=targets, losses=losses, unary=grads,
                      coeff_fn=lambda: [v / 5.0 for v in self.task.host_weights()],
                      stream=pst if pst is not main else None, launch_probes=launch_probes)
            if dz_pre is not None:
                so["dz"] = (dz_pre, dbsum)
            # the task weights' local step (it needs the probe norms, which the head backward has produced long before
            # the last bucket) runs just before the LAST gradient bucket is issued and rides on it
            before_last = (lambda: w_step(losses, so["gnorm"])) if w_step is not None else None
            exchange = T.BucketedAllReduce(self.grad, group=group, store=self._grad_store, before_last=before_last)
            self.backward(ghead, exchange, so=so)
            scale = exchange.finish()
            gn = so["gnorm"]
        elif coeff is None:
            # first-order step on the fused plumbing: stacked probe passes beside the backward (they need the head only),
            # then ONE pass over the five probe-sized tensors for the limb probe gradient and all norms
            main = torch.cuda.current_stream(self.device)
            if self._probe_scratch is None or self._probe_scratch.shape != head.shape:
                self._probe_scratch = torch.empty_like(head)
            with torch.cuda.stream(self._probe_stream):
                self._probe_stream.wait_event(ev_head)
                gs = self._stacked_unary_probe_grads(head, targets, self._probe_scratch)
            exchange = T.BucketedAllReduce(self.grad, group=group)
            self.backward(None, exchange, so=dict(dz=(dz_pre, dbsum)))
            scale = exchange.finish()
            main.wait_stream(self._probe_stream)
            for t in gs:
                t.record_stream(main)
            coeff = [v / 5.0 for v in self.task.host_weights()]
            local = self._conv1_local if self._conv1_local is not None else self.G["conv1.weight"]
            trusted = coeff[4] > 1e-3 * max(coeff)
            if trusted:
                gw4, st = T.probe_stats([g.contiguous() for g in gs], local.contiguous(), coeff)
                if self.compute_dtype != L.PPN_F32:
                    n_rest, n_tot = st[5:7].tolist()                    # the trust test of _limb_probe
                    trusted = n_rest > (16.0 * 2.0 ** -8) ** 2 * n_tot
            if trusted:
                gn = torch.sqrt(st[:5])
            else:
                _, g4 = self.criterion.forward_backward(head, targets, coeff=[0.0, 0.0, 0.0, 0.0, 1.0])
                gn = torch.sqrt(torch.cat([T.sumsq(g.contiguous().view(-1)) for g in gs] +
                                          [T.sumsq(self.probe_grad(g4).contiguous().view(-1))]))
        else:
            if self._probe_stream is not None:
                # the four unary probe passes are small launches that depend only on the forward: they run on their
                # own stream underneath the backward pass
                main = torch.cuda.current_stream(self.device)
                if self._probe_scratch is None or self._probe_scratch.shape != head.shape:
                    self._probe_scratch = torch.empty_like(head)
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(self._probe_stream):
                    self._probe_stream.wait_event(ev)
                    unary = self._unary_probes(head, targets, coeff, self._probe_scratch)
            # gradient exchange: 32 MB buckets of the flat buffer go out over RCCL as the backward completes them
            exchange = T.BucketedAllReduce(self.grad, group=group)
            self.backward(ghead, exchange)
            scale = exchange.finish()
            if unary is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._probe_stream)
                for t in unary:
                    t.record_stream(torch.cuda.current_stream(self.device))
            gn = self.probe_norms(head, targets, coeff, ghead, unary=unary)
        self._tape = None
        self._conv1_local = None
        return losses, gn, scale

    def train_step(self, x: torch.Tensor, targets: Dict[str, torch.Tensor], group=None):
        """main.py:664-777 for one minibatch shard.  Returns (losses f32[5], task weights f32[5]) device tensors."""
        rode = []

        def w_step(losses_, gn_):
            self.task.local_step(losses_, gn_, self.base)                # optimizerR.step on this rank's values ...
            rode.append(True)                                            # ... summed over the ranks with the last bucket

        losses, gn, scale = self.local_pass(x, targets, group, w_step=w_step)
        if rode:
            self.task.renorm(int(round(1.0 / scale)))                    # / world, clamp, renormalise (main.py:772-777)
        else:
            self.task.step(losses, gn, self.base, group=group)           # optimizerR.step + all-reduce + renormalise
        self.opt.step(self.grad, grad_scale=scale)                       # optimizerM.step
        return losses, self.task.w
