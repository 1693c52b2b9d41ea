"""ORACLE (test infrastructure, never shipped or measured as the product).

Executes the *fused conv program* (pytorch_pose_proposal_network_amd.arch.build_program) with plain
PyTorch-CPU ops, one F.conv2d per ConvOp followed by the same epilogue algebra the HIP kernel applies:

    v = act1(conv(x, w) * scale1 + shift1) (+ residual);  out_raw = v;  out_act = act2(v * scale2 + shift2)

Two uses:
  * fp32: must agree with oracle/forward_ref.py (the op-by-op restatement of model.py:104-136 /
    drn.py:42-57,77-97) -- a CPU-only check of the host lowering (BN folding, pre-activation second
    output, Bottleneck ReLU-after-add), no GPU needed.
  * emulate_bf16=True: weights and every stored activation are rounded to bf16 exactly where the HIP bf16
    mode stores bf16 (accumulation and epilogue stay fp32, the head stays fp32), giving a tight reference
    for the bf16 kernels that is independent of how much a random network amplifies rounding noise.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from pytorch_pose_proposal_network_amd import arch as A

_ACT = {A.ACT_NONE: lambda t: t, A.ACT_RELU: F.relu, A.ACT_LRELU: lambda t: F.leaky_relu(t, 0.1),
        A.ACT_SIGMOID: torch.sigmoid}


def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _fold(sd, prefix):
    g, b = _t(sd[prefix + ".weight"]).double(), _t(sd[prefix + ".bias"]).double()
    m, v = _t(sd[prefix + ".running_mean"]).double(), _t(sd[prefix + ".running_var"]).double()
    s = g / torch.sqrt(v + 1e-5)
    return s, b - m * s


def _bf16(t):
    return t.to(torch.bfloat16).float()


def fused_forward_ref(sd, x: torch.Tensor, arch: str = "drn_d_22", emulate_bf16: bool = False, taps=None,
                      fuse_stem=False, emulate_dtype=None, residual_dtype="same", fuse_shortcut: bool = True,
                      stem_dtype=None, half_prefix: int = -1, exact_prefix: int = -1, exact_input: bool = False):
    """x f32 [B,3,H,W] (normalised) -> head f32 [B,C,H/16,W/16], walking the fused program.
    emulate_dtype=torch.float16 emulates the PPN_F16 mode's storage roundings the way emulate_bf16 does bf16's.
    residual_dtype (precision study, tests/precision_study.py): storage type of the tensors that are ONLY ever read as a
    residual (never as a convolution operand) -- "same" = emulate_dtype, None = f32, or a torch dtype.
    stem_dtype: the type the stem (layer0-2) computes in -- its weights, input patch and the tensors between its layers --
    while its outputs are stored in emulate_dtype (the bf16 mode's default since round 4: torch.float16).
    half_prefix: the launches of backbone.0 .. backbone.{half_prefix} compute and store in IEEE half, and a tensor is stored
    in the type of the launches that read it (the bf16 mode's default since round 4: 4 = stem + layer3 + layer4).
    exact_input: the stem's input patch is not rounded (the u8 path of the fused half stem keeps the integer x - 128 and folds
    the normalisation into layer 0's weights, csrc/stem012.hip; the f32-input path of model.forward() rounds the patch).
    exact_prefix: the launches of backbone.0 .. backbone.{exact_prefix} are exact (f32 / float16x3 on the GPU: no rounding
    emulated); what they hand to the 16-bit trunk is rounded to its type (PoseProposalNet(exact_prefix=))."""
    ops = A.build_program(arch, fuse_stem=fuse_stem, fuse_shortcut=fuse_shortcut)
    if emulate_dtype is None and emulate_bf16:
        emulate_dtype = torch.bfloat16
    q = (lambda t: t.to(emulate_dtype).float()) if emulate_dtype is not None else (lambda t: t)
    operands = {o.src for o in ops} | {o.ds_src for o in ops if o.ds_src}
    if residual_dtype == "same":
        q_res = q
    else:
        q_res = (lambda t: t.to(residual_dtype).float()) if residual_dtype is not None else (lambda t: t)
    q_trunk = q
    half_names = tuple(f"backbone.{i}." for i in range(half_prefix + 1)) if (half_prefix >= 3 and emulate_dtype is not None) else ()
    q_half = lambda t: t.to(torch.float16).float()          # noqa: E731
    in_half = lambda o: bool(half_names) and o.name.startswith(half_names)          # noqa: E731
    exact_names = tuple(f"backbone.{i}." for i in range(exact_prefix + 1)) if (exact_prefix >= 3 and emulate_dtype is not None) else ()
    q_exact = lambda t: t                                   # noqa: E731
    in_exact = lambda o: bool(exact_names) and o.name.startswith(exact_names)       # noqa: E731
    read_q = {}                                             # tensor -> rounding of the launches that read it
    for o in ops:
        for name in (o.src, o.residual, o.ds_src):
            if name:
                read_q[name] = q_exact if in_exact(o) else (q_half if in_half(o) else q_trunk)
    qs = (lambda t: t.to(stem_dtype).float()) if (stem_dtype is not None and emulate_dtype is not None) else None
    stem_names = {o.name for o in ops[:3] if o.cin in (3, 16)} if qs is not None else set()
    tensors = {"input": x.float()}
    with torch.no_grad():
        for op in ops:
            # inside the stem: weights, patch and the tensors between its layers in stem_dtype; what LEAVES the stem (the
            # outputs of its last layer) in the trunk's type
            in_stem = op.name in stem_names
            q = q_exact if in_exact(op) else (qs if in_stem else (q_half if in_half(op) else q_trunk))
            in_stem = in_stem and not in_exact(op)
            last_stem = in_stem and (op.next_s2 is not None or (op.cin == 16 and op.cout == 32))
            q_store = qs if (in_stem and not last_stem) else None      # None: the type of the launches that read the tensor
            src = tensors[op.src]
            w = _t(sd[op.weight]).float()
            if op.k == 7:
                src_q, w_q = (src if exact_input else q(src)), q(w)   # the stem stages a 16-bit patch / weight fragments
            else:
                src_q, w_q = src, q(w)             # other inputs are already-stored (rounded) tensors
            acc = F.conv2d(src_q, w_q, None, op.stride, op.pad, op.dilation)
            s1 = b1 = None
            if op.ds_src:
                # fused projection shortcut: the BN scale is folded into the 1x1 weights BEFORE they are stored
                sds, bds = _fold(sd, op.ds_bn)
                wds = _t(sd[op.ds_weight]).float() * sds.float().view(-1, 1, 1, 1)
                acc = acc + F.conv2d(tensors[op.ds_src], q(wds), None, op.ds_stride, 0, 1)
                b1 = bds
            if op.bn1:
                s1, b1 = _fold(sd, op.bn1)
            if op.bias:
                bias = _t(sd[op.bias]).double()
                b1 = bias * s1 + b1 if s1 is not None else bias
            v = acc
            if s1 is not None:
                v = v * s1.float().view(1, -1, 1, 1)
            if b1 is not None:
                v = v + b1.float().view(1, -1, 1, 1)
            v = _ACT[op.act1](v)
            if op.next3x3 is not None:
                # layer0 + layer1 in one launch: the intermediate lives in LDS in the storage dtype
                n = op.next3x3
                s2n, b2n = _fold(sd, n.bn1)
                acc2 = F.conv2d(q(v), q(_t(sd[n.weight]).float()), None, n.stride, n.pad, n.dilation)
                v = _ACT[n.act1](acc2 * s2n.float().view(1, -1, 1, 1) + b2n.float().view(1, -1, 1, 1))
                if op.next_s2 is not None:           # ... and layer2 as well (csrc/stem012.hip)
                    m = op.next_s2
                    s3n, b3n = _fold(sd, m.bn1)
                    acc3 = F.conv2d(q(v), q(_t(sd[m.weight]).float()), None, m.stride, m.pad, m.dilation)
                    v = _ACT[m.act1](acc3 * s3n.float().view(1, -1, 1, 1) + b3n.float().view(1, -1, 1, 1))
            if op.residual:
                v = v + tensors[op.residual]
            if op.out_raw:
                qo = q_store or read_q.get(op.out_raw, q_trunk)
                tensors[op.out_raw] = v if op.nchw_f32_out else (qo(v) if op.out_raw in operands else q_res(v))
            if op.out_act:
                u = v
                if op.bn2:
                    s2, b2 = _fold(sd, op.bn2)
                    u = u * s2.float().view(1, -1, 1, 1) + b2.float().view(1, -1, 1, 1)
                tensors[op.out_act] = (q_store or read_q.get(op.out_act, q_trunk))(_ACT[op.act2](u))
            if taps is not None:
                taps[op.name] = tensors[op.out_raw or op.out_act]
    return tensors["head"]
