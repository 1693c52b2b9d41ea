"""Host-side description of DRN-D + PPN head as a flat program of fused convolutions.

What the reference builds with nn.Module objects (``drn.py:102-202`` DRN/_make_layer/
_make_conv_layers, ``drn.py:25-97`` BasicBlock/Bottleneck, ``model.py:51-136``
PoseProposalNet) is restated here as data:

* ``param_spec(arch)``   -- every state_dict entry (name, shape) the reference model owns,
                            so ``load_state_dict`` can validate real checkpoints
                            (names: ``backbone.{0..8}...``, ``basicblock{1,2}.*``,
                            ``conv1x1_{1,2}``, ``conv{1,2,3}``, ``bn0_{1,2}``, ``bn{1,2}``).
* ``build_program(arch)`` -- the list of ``ConvOp`` the HIP executor launches, one fused
                            kernel per convolution: implicit-GEMM conv, then
                            ``v = act1(acc*scale1+shift1) (+residual)``, optional raw store,
                            optional second output ``act2(v*scale2+shift2)`` which is the
                            *pre-activation* BN->ReLU of the next BasicBlock.  Zero padding
                            of a pre-activated input therefore happens after BN/ReLU, as in
                            the reference (drn.py:45-51).

No torch import here: this is pure host logic, usable by the oracle and the tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from . import config as cfg

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3

# layers[0..7] and block type per DRN-D variant (drn.py:345-398)
DRN_D = {
    "drn_d_22": ("basic", [1, 1, 2, 2, 2, 2, 1, 1]),
    "drn_d_24": ("basic", [1, 1, 2, 2, 2, 2, 2, 2]),
    "drn_d_38": ("basic", [1, 1, 3, 4, 6, 3, 1, 1]),
    "drn_d_40": ("basic", [1, 1, 3, 4, 6, 3, 2, 2]),
    "drn_d_54": ("bottleneck", [1, 1, 3, 4, 6, 3, 1, 1]),
    "drn_d_56": ("bottleneck", [1, 1, 3, 4, 6, 3, 2, 2]),
    "drn_d_105": ("bottleneck", [1, 1, 3, 4, 23, 3, 1, 1]),
    "drn_d_107": ("bottleneck", [1, 1, 3, 4, 23, 3, 2, 2]),
}
CHANNELS = (16, 32, 64, 128, 256, 512, 512, 512)  # drn.py:105

@dataclass
class ConvOp:
    """One fused launch.  Tensor names refer to NHWC activation buffers."""
    name: str
    src: str
    weight: str                       # state_dict key of the conv weight [Cout,Cin,k,k]
    cin: int
    cout: int
    k: int
    stride: int = 1
    dilation: int = 1
    pad: int = 0
    bias: Optional[str] = None        # state_dict key
    bn1: Optional[str] = None         # BN applied to the conv output (foldable), key prefix
    act1: int = ACT_NONE
    residual: Optional[str] = None    # tensor added after act1
    out_raw: Optional[str] = None     # store v
    bn2: Optional[str] = None         # BN of the *consumer's* pre-activation
    act2: int = ACT_NONE
    out_act: Optional[str] = None     # store act2(bn2(v))
    nchw_f32_out: bool = False        # head: sigmoid output in the reference's NCHW f32 layout
    next3x3: Optional["ConvOp"] = None  # layer0 only: layer1 (3x3 16->16 conv+BN+ReLU) fused into the same launch
    next_s2: Optional["ConvOp"] = None  # layer0 only, with next3x3: layer2 (3x3 stride 2 16->32) fused too (stem012.hip)
    # fused projection shortcut: out = conv(src) + bn_ds(conv1x1_stride(ds_src))  (BasicBlock.downsample, drn.py:53-54)
    ds_src: Optional[str] = None
    ds_weight: Optional[str] = None
    ds_bn: Optional[str] = None
    ds_cin: int = 0
    ds_stride: int = 1

@dataclass
class Unit:
    kind: str                         # 'cbr' | 'basic' | 'bottleneck' | 'head'
    prefix: str
    cin: int
    cout: int
    stride: int = 1
    dil: Tuple[int, int] = (1, 1)
    downsample: bool = False
    k: int = 3
    conv_idx: int = 0                 # for 'cbr': index of conv inside its nn.Sequential
    planes: int = 0                   # bottleneck inner width

def _units(arch: str) -> List[Unit]:
    block, layers = DRN_D[arch]
    exp = 1 if block == "basic" else 4
    units: List[Unit] = []
    inpl = CHANNELS[0]
    # layer0: 7x7 conv-BN-ReLU (drn.py:123-128)
    units.append(Unit("cbr", "backbone.0", 3, CHANNELS[0], k=7, conv_idx=0))
    # layer1, layer2: _make_conv_layers (drn.py:192-202)
    for li, (ch, stride) in enumerate(((CHANNELS[0], 1), (CHANNELS[1], 2)), start=1):
        for i in range(layers[li - 1]):
            units.append(Unit("cbr", f"backbone.{li}", inpl, ch, stride=stride if i == 0 else 1,
                              conv_idx=3 * i))
            inpl = ch
    # layer3..6: _make_layer (drn.py:168-190); layers 5,6 are dilated with new_level=False
    for li, (planes, nblk, stride, dil) in enumerate(
            ((CHANNELS[2], layers[2], 2, 1), (CHANNELS[3], layers[3], 2, 1),
             (CHANNELS[4], layers[4], 1, 2), (CHANNELS[5], layers[5], 1, 4)), start=3):
        for b in range(nblk):
            first = b == 0
            ds = first and (stride != 1 or inpl != planes * exp)
            d = (1, 1) if dil == 1 else (dil, dil)
            units.append(Unit(block, f"backbone.{li}.{b}", inpl, planes * exp,
                              stride=stride if first else 1, dil=d, downsample=ds, planes=planes))
            inpl = planes * exp
    # layer7, layer8: conv layers with dilation 2 then 1 (drn.py:150-154)
    for li, (ch, n, dil) in enumerate(((CHANNELS[6], layers[6], 2), (CHANNELS[7], layers[7], 1)), start=7):
        for i in range(n):
            units.append(Unit("cbr", f"backbone.{li}", inpl, ch, dil=(dil, dil), conv_idx=3 * i))
            inpl = ch
    # PPN head (model.py:66-93)
    units.append(Unit("basic", "basicblock1", 512, 512, stride=2, downsample=True, planes=512))
    units.append(Unit("basic", "basicblock2", 512, 512, planes=512))
    units.append(Unit("head", "", 512, cfg.lastsize()))
    return units

def _bn(prefix: str, c: int):
    return [(f"{prefix}.weight", (c,)), (f"{prefix}.bias", (c,)),
            (f"{prefix}.running_mean", (c,)), (f"{prefix}.running_var", (c,)),
            (f"{prefix}.num_batches_tracked", ())]

def param_spec(arch: str = "drn_d_22", head_channels: Optional[int] = None) -> List[Tuple[str, tuple]]:
    """(name, shape) of every state_dict entry of the reference PoseProposalNet(arch)."""
    spec: List[Tuple[str, tuple]] = []
    hc = head_channels or cfg.lastsize()
    for u in _units(arch):
        if u.kind == "cbr":
            spec.append((f"{u.prefix}.{u.conv_idx}.weight", (u.cout, u.cin, u.k, u.k)))
            spec += _bn(f"{u.prefix}.{u.conv_idx + 1}", u.cout)
        elif u.kind == "basic":
            # drn.py:33 -- bn1 has `inplanes` features (model.py:22 uses planes; equal there)
            spec.append((f"{u.prefix}.conv1.weight", (u.cout, u.cin, 3, 3)))
            spec += _bn(f"{u.prefix}.bn1", u.cin)
            spec.append((f"{u.prefix}.conv2.weight", (u.cout, u.cout, 3, 3)))
            spec += _bn(f"{u.prefix}.bn2", u.cout)
            if u.downsample:
                spec.append((f"{u.prefix}.downsample.0.weight", (u.cout, u.cin, 1, 1)))
                spec += _bn(f"{u.prefix}.downsample.1", u.cout)
        elif u.kind == "bottleneck":
            p = u.planes
            spec.append((f"{u.prefix}.conv1.weight", (p, u.cin, 1, 1)))
            spec += _bn(f"{u.prefix}.bn1", p)
            spec.append((f"{u.prefix}.conv2.weight", (p, p, 3, 3)))
            spec += _bn(f"{u.prefix}.bn2", p)
            spec.append((f"{u.prefix}.conv3.weight", (4 * p, p, 1, 1)))
            spec += _bn(f"{u.prefix}.bn3", 4 * p)
            if u.downsample:
                spec.append((f"{u.prefix}.downsample.0.weight", (u.cout, u.cin, 1, 1)))
                spec += _bn(f"{u.prefix}.downsample.1", u.cout)
        else:  # head, in model.py __init__ attribute order (model.py:80-93)
            spec.append(("conv1x1_1.weight", (128, 512, 1, 1)))
            spec.append(("conv1x1_2.weight", (512, 128, 1, 1)))
            spec.append(("conv1.weight", (128, 128, 3, 3)))
            spec.append(("conv2.weight", (512, 512, 3, 3)))
            spec.append(("conv2.bias", (512,)))
            spec.append(("conv3.weight", (hc, 512, 1, 1)))
            spec.append(("conv3.bias", (hc,)))
            spec += _bn("bn0_1", 512) + _bn("bn0_2", 128) + _bn("bn1", 128) + _bn("bn2", 512)
    return spec

def _needs(u: Optional[Unit]):
    """What unit `u` wants from its producer: (needs_raw, (bn_prefix, act) or None)."""
    if u is None:
        return True, None
    if u.kind == "basic":
        return True, (f"{u.prefix}.bn1", ACT_RELU)       # raw x feeds residual / downsample
    if u.kind == "head":
        return True, ("bn0_1", ACT_LRELU)                # R is re-added at model.py:127
    return True, None                                    # cbr / bottleneck read raw x

def build_program(arch: str = "drn_d_22", head_channels: Optional[int] = None, fuse_stem=False,
                  fuse_shortcut: bool = True) -> List[ConvOp]:
    """Lower the module list into fused conv launches.

    fuse_stem: True -- layer0 (7x7) and the first conv of layer1 (3x3 16->16) share one launch (csrc/stem01.hip):
    the 16x384x384 tensor between them never goes to HBM (measured slower than the two launches, kept as an option).
    "all" -- layer0, layer1 and layer2 (3x3 stride 2 16->32) in one launch (csrc/stem012.hip, bf16 mode): neither
    16-channel full-resolution tensor goes to HBM; needs one conv per stem layer (every DRN-D variant has that)."""
    # fuse_shortcut (bool, or a predicate of the unit's prefix): a BasicBlock's 1x1 projection shortcut (+BN) becomes extra GEMM depth of its second conv
    # (no separate launch, no residual tensor) when its input width is a multiple of 64 and the block is narrow.
    hc = head_channels or cfg.lastsize()
    units = _units(arch)
    ops: List[ConvOp] = []
    raw, act = "input", None          # current tensors: raw x and (optionally) its pre-activation
    n = 0

    def t(tag):
        nonlocal n
        n += 1
        return f"t{n}_{tag}"

    for i, u in enumerate(units):
        nxt = units[i + 1] if i + 1 < len(units) else None
        need_raw, pre = _needs(nxt)

        def finish(op: ConvOp, tag: str):
            """Attach the outputs the next unit needs to the unit's last conv."""
            nonlocal raw, act
            if need_raw:
                op.out_raw = t(tag)
            if pre is not None:
                op.bn2, op.act2 = pre
                op.out_act = t(tag + "_pre")
            ops.append(op)
            raw, act = op.out_raw, op.out_act

        if u.kind == "cbr":
            w = f"{u.prefix}.{u.conv_idx}"
            d = u.dil[0]
            op = ConvOp(w, raw, f"{w}.weight", u.cin, u.cout, u.k, u.stride, d,
                        pad=(3 if u.k == 7 else d), bn1=f"{u.prefix}.{u.conv_idx + 1}", act1=ACT_RELU)
            if (fuse_stem and i == 1 and ops and ops[-1].k == 7 and ops[-1].next3x3 is None and u.cin == 16 and
                    u.cout == 16 and u.k == 3 and u.stride == 1 and d == 1 and pre is None):
                first = ops.pop()                      # layer0: its only consumer is this conv
                first.next3x3 = op
                first.name = first.name + "+" + op.name
                first.out_raw = None
                op.src = "(on chip)"
                finish(first, w.replace(".", "_"))
                continue
            if (fuse_stem == "all" and i == 2 and len(ops) == 1 and ops[-1].next3x3 is not None and
                    ops[-1].next_s2 is None and u.cin == 16 and u.cout == 32 and u.k == 3 and u.stride == 2 and d == 1):
                first = ops.pop()                      # layer0+layer1: their only consumer is this conv
                first.next_s2 = op
                first.name = first.name + "+" + op.name
                first.out_raw = first.out_act = None
                first.bn2, first.act2 = None, ACT_NONE
                op.src = "(on chip)"
                finish(first, w.replace(".", "_"))
                continue
            finish(op, w.replace(".", "_"))
        elif u.kind == "basic":
            p = u.prefix
            assert act is not None, "pre-activation tensor missing for BasicBlock"
            res = raw
            # measured on MI355X (batch 32, whole step, same box): fusing the 64->128 and 128->256 projections is
            # worth 0-3 % of the step (their separate 1x1 launches under-fill the GPU); for the 512-wide blocks it
            # loses 6-30 us each (the two-source loader slows every K step of a launch that is already efficient)
            # -> fuse only up to 256 output channels
            fuse_ok = fuse_shortcut(u.prefix) if callable(fuse_shortcut) else fuse_shortcut
            fuse_ds = u.downsample and fuse_ok and u.cin % 64 == 0 and 64 <= u.cout <= 256
            if u.downsample and not fuse_ds:
                res = t(p.replace(".", "_") + "_ds")
                ops.append(ConvOp(f"{p}.downsample", raw, f"{p}.downsample.0.weight", u.cin, u.cout, 1,
                                  u.stride, 1, 0, bn1=f"{p}.downsample.1", out_raw=res))
            mid = t(p.replace(".", "_") + "_c1")
            ops.append(ConvOp(f"{p}.conv1", act, f"{p}.conv1.weight", u.cin, u.cout, 3, u.stride,
                              u.dil[0], u.dil[0], bn1=f"{p}.bn2", act1=ACT_RELU, out_raw=mid))
            op = ConvOp(f"{p}.conv2", mid, f"{p}.conv2.weight", u.cout, u.cout, 3, 1, u.dil[1], u.dil[1],
                        residual=None if fuse_ds else res)
            if fuse_ds:
                op.name = f"{p}.conv2+downsample"
                op.ds_src, op.ds_weight, op.ds_bn = raw, f"{p}.downsample.0.weight", f"{p}.downsample.1"
                op.ds_cin, op.ds_stride = u.cin, u.stride
            finish(op, p.replace(".", "_"))
        elif u.kind == "bottleneck":
            p, pl = u.prefix, u.planes
            assert pre is None, "a pre-activation consumer directly after a Bottleneck is not supported"
            res = raw
            if u.downsample:
                res = t(p.replace(".", "_") + "_ds")
                ops.append(ConvOp(f"{p}.downsample", raw, f"{p}.downsample.0.weight", u.cin, u.cout, 1,
                                  u.stride, 1, 0, bn1=f"{p}.downsample.1", out_raw=res))
            m1 = t(p.replace(".", "_") + "_c1")
            ops.append(ConvOp(f"{p}.conv1", raw, f"{p}.conv1.weight", u.cin, pl, 1, 1, 1, 0,
                              bn1=f"{p}.bn1", act1=ACT_RELU, out_raw=m1))
            m2 = t(p.replace(".", "_") + "_c2")
            ops.append(ConvOp(f"{p}.conv2", m1, f"{p}.conv2.weight", pl, pl, 3, u.stride, u.dil[1], u.dil[1],
                              bn1=f"{p}.bn2", act1=ACT_RELU, out_raw=m2))
            # out = relu(bn3(conv3) + residual): the ReLU comes after the add (drn.py:92-95),
            # so it is expressed as the second (identity-affine) output.
            o = t(p.replace(".", "_"))
            ops.append(ConvOp(f"{p}.conv3", m2, f"{p}.conv3.weight", pl, 4 * pl, 1, 1, 1, 0,
                              bn1=f"{p}.bn3", residual=res, act2=ACT_RELU, out_act=o))
            raw, act = o, None
        else:  # PPN head, model.py:113-134
            R, Rpre = raw, act
            a1 = t("neck1")
            ops.append(ConvOp("conv1x1_1", Rpre, "conv1x1_1.weight", 512, 128, 1, bn1="bn1",
                              act1=ACT_LRELU, out_raw=a1))
            a2 = t("neck2")
            ops.append(ConvOp("conv1", a1, "conv1.weight", 128, 128, 3, pad=1, bn1="bn0_2",
                              act1=ACT_LRELU, out_raw=a2))
            a3 = t("neck3")
            ops.append(ConvOp("conv1x1_2", a2, "conv1x1_2.weight", 128, 512, 1, residual=R, out_raw=a3))
            c = t("conv2")
            ops.append(ConvOp("conv2", a3, "conv2.weight", 512, 512, 3, pad=1, bias="conv2.bias",
                              bn1="bn2", act1=ACT_LRELU, out_raw=c))
            ops.append(ConvOp("conv3", c, "conv3.weight", 512, hc, 1, bias="conv3.bias",
                              act1=ACT_SIGMOID, out_raw="head", nchw_f32_out=True))
    return ops

def out_hw(op: ConvOp, h: int, w: int) -> Tuple[int, int]:
    eff = op.dilation * (op.k - 1) + 1
    return (h + 2 * op.pad - eff) // op.stride + 1, (w + 2 * op.pad - eff) // op.stride + 1

def tensor_shapes(ops: List[ConvOp], h: int, w: int) -> Dict[str, Tuple[int, int, int]]:
    """(H, W, C) of every activation tensor for an input of h x w."""
    shapes: Dict[str, Tuple[int, int, int]] = {"input": (h, w, 3)}
    for op in ops:
        ih, iw, ic = shapes[op.src]
        assert ic == op.cin, (op.name, ic, op.cin)
        oh, ow = out_hw(op, ih, iw)
        cout = op.next3x3.cout if op.next3x3 else op.cout
        if op.next_s2:
            oh, ow = out_hw(op.next_s2, oh, ow)
            cout = op.next_s2.cout
        for name in (op.out_raw, op.out_act):
            if name:
                shapes[name] = (oh, ow, cout)
        if op.residual:
            assert shapes[op.residual] == (oh, ow, op.cout), (op.name, shapes[op.residual], (oh, ow, op.cout))
    return shapes

def conv_flops(ops: List[ConvOp], h: int, w: int) -> int:
    """2*MACs of all convolutions for one image (the 95.304 GFLOP figure of BASELINE.md)."""
    shapes = tensor_shapes(ops, h, w)
    total = 0
    for op in ops:
        total += op_flops(op, shapes)
    return total

def op_flops(op: ConvOp, shapes) -> int:
    """2*MACs of one launch for one image (a fused layer0+layer1 launch counts both convolutions)."""
    oh, ow, _ = shapes[op.out_raw or op.out_act]
    if op.next_s2:                                   # layer0 and layer1 run at the input resolution
        m = op.next_s2
        fl2 = 2 * m.cin * m.cout * m.k * m.k * oh * ow
        ih, iw, _ = shapes[op.src]
        oh, ow = out_hw(op, ih, iw)
    else:
        fl2 = 0
    fl = 2 * op.cin * op.cout * op.k * op.k * oh * ow + fl2
    if op.ds_src:
        fl += 2 * op.ds_cin * op.cout * oh * ow
    if op.next3x3:
        n = op.next3x3
        fl += 2 * n.cin * n.cout * n.k * n.k * oh * ow
    return fl
