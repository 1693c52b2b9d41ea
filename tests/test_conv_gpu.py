"""GPU parity of the fused conv kernels (through the C ABI) against plain PyTorch fp32 ops of the same
layer (torch CPU), covering every shape class of SURVEY.md 2.1: stride 2, dilation 2/4, 1x1, Cin < K step,
residual, dual (pre-activation) output, bias+LReLU, and the NCHW sigmoid head with Cout = 7605."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

F32_TOL = 2e-5      # relative to the output scale; exact-f32 MFMA vs torch CPU summation order
BF16_TOL = 2e-2
F16_TOL = 3e-3      # IEEE half keeps 11 significant bits (bf16: 8)


def _act(v, a):
    return [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.1), torch.sigmoid][a](v)


def run_conv(x, w, dtype, stride=1, dil=1, pad=0, s1=None, b1=None, act1=0, residual=None, s2=None, b2=None, act2=0,
             want_raw=True, want_act=False, nchw=False, shortcut=None, argmax=None, info=None, ranges=None):
    """x [B,Cin,H,W], w [Cout,Cin,k,k] CPU f32 -> (raw, act) as NCHW CPU f32 tensors via libppn.

    argmax=(unary_channels, window): NCHW head mode with the decode's limb arg-max fused into the epilogue; the
    compact unary tensor and the u64 keys are returned through ``info`` (a dict, which also receives the name of
    the kernel instantiation that ran).  ranges=[(m_begin, m_count, (bp, bc) or None), ...]: one launch per entry over
    that range of the flattened output pixels (ppn_conv_desc.m_begin/m_count), each with its own forced tile; pixels
    outside every range keep the NaN fill."""
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda")
    tdt = {L.PPN_F32: torch.float32, L.PPN_BF16: torch.bfloat16, L.PPN_F16: torch.float16}[dtype]
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    eff = dil * (k - 1) + 1
    Ho, Wo = (H + 2 * pad - eff) // stride + 1, (W + 2 * pad - eff) // stride + 1
    kstep, _, korder, ktot, cpad = L.conv_tiling(dtype, Cin, Cout, k)
    st = torch.cuda.current_stream().cuda_stream
    wd = w.contiguous().to(dev)
    packed = torch.empty(cpad, ktot, dtype=torch.float32 if korder == 2 else tdt, device=dev)
    L.check(lib.ppn_pack_weight(dtype, wd.data_ptr(), Cout, Cin, k, cpad, ktot, korder, kstep, packed.data_ptr(), st))
    xs = x.permute(0, 2, 3, 1).contiguous().to(dev, tdt)
    zero = torch.zeros(64, device=dev)
    keep = [wd, packed, xs, zero]
    d = L.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.cin = dtype, B, H, W, Cin
    d.out_h, d.out_w, d.cout = Ho, Wo, Cout
    d.ksize, d.stride, d.dilation, d.pad = k, stride, dil, pad
    d.k_total, d.cout_pad, d.act1, d.act2, d.out_nchw_f32 = ktot, cpad, act1, act2, int(nchw)
    d.src, d.weight, d.zero_page = xs.data_ptr(), packed.data_ptr(), zero.data_ptr()

    def dv(t):
        if t is None:
            return None
        t = t.float().contiguous().to(dev)
        keep.append(t)
        return t.data_ptr()

    d.scale1, d.shift1, d.scale2, d.shift2 = dv(s1), dv(b1), dv(s2), dv(b2)
    if shortcut is not None:
        # fused projection shortcut: x2 [B,Cin2,H2,W2], w2 [Cout,Cin2,1,1], stride2 -- extra GEMM depth
        x2, w2, stride2 = shortcut
        cin2 = x2.shape[1]
        assert korder == 1 and cin2 % kstep == 0
        w2d = w2.contiguous().to(dev)
        p2 = torch.empty(cpad, cin2, dtype=tdt, device=dev)
        L.check(lib.ppn_pack_weight(dtype, w2d.data_ptr(), Cout, cin2, 1, cpad, cin2, 1, kstep, p2.data_ptr(), st))
        packed = torch.cat([packed, p2], dim=1).contiguous()
        xs2 = x2.permute(0, 2, 3, 1).contiguous().to(dev, tdt)
        keep += [w2d, packed, xs2]
        d.weight, d.k_total = packed.data_ptr(), ktot + cin2
        d.src2, d.in2_h, d.in2_w, d.cin2, d.stride2 = xs2.data_ptr(), x2.shape[2], x2.shape[3], cin2, stride2
    if residual is not None:
        r = residual.permute(0, 2, 3, 1).contiguous().to(dev, tdt)
        keep.append(r)
        d.residual = r.data_ptr()
    raw = act = None
    if want_raw:
        raw = (torch.full((B, Cout, Ho, Wo), float("nan"), device=dev) if nchw
               else torch.full((B, Ho, Wo, Cout), float("nan"), device=dev).to(tdt))
        d.out_raw = raw.data_ptr()
    if want_act:
        act = torch.full((B, Ho, Wo, Cout), float("nan"), device=dev).to(tdt)
        d.out_act = act.data_ptr()
    if argmax is not None:
        uch, win = argmax
        unary = torch.full((B, uch, Ho, Wo), float("nan"), device=dev)
        keys = torch.zeros(B, (Cout - uch) // win, Ho, Wo, dtype=torch.int64, device=dev)
        keep += [unary, keys]
        d.unary_out, d.argmax_keys, d.unary_channels, d.limb_window = unary.data_ptr(), keys.data_ptr(), uch, win
    kernels = []
    for lo, n, tile in (ranges if ranges is not None else [(0, 0, None)]):
        d.m_begin, d.m_count = lo, n
        if tile is not None:
            L.check(lib.ppn_set_conv_tile_override(*tile), "ppn_set_conv_tile_override")
        try:
            L.check(lib.ppn_conv2d_fused(C.byref(d), st), "ppn_conv2d_fused")
        finally:
            if tile is not None:
                L.check(lib.ppn_set_conv_tile_override(0, 0), "ppn_set_conv_tile_override")
        kernels.append(lib.ppn_last_conv_kernel().decode())
    torch.cuda.synchronize()
    if info is not None:
        info["kernel"] = kernels[0]
        info["kernels"] = kernels
        if argmax is not None:
            info["unary"], info["keys"] = unary.cpu(), keys.cpu()
    out = []
    for t in (raw, act):
        if t is None:
            out.append(None)
        elif nchw:
            out.append(t.float().cpu())
        else:
            out.append(t.float().cpu().permute(0, 3, 1, 2).contiguous())
    return out


def run_edge_argmax(x, w, bias, dtype, uch, win, edge_pad=448):
    """The limb part of a conv3-shaped launch through the edge-aligned tile (ppn_conv_desc.limb_edge_pad): one channel
    tile per edge, arg-max reduced on the accumulators, keys STORED into a buffer that is NOT zeroed first.  x [B,Cin,H,W],
    w [uch + E*win, Cin, 1, 1], bias [uch + E*win] CPU f32 -> (keys i64 [B,E,H,W] CPU, kernel name)."""
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda")
    tdt = {L.PPN_F32: torch.float32, L.PPN_BF16: torch.bfloat16, L.PPN_F16: torch.float16}[dtype]
    B, Cin, H, W = x.shape
    E = (w.shape[0] - uch) // win
    kstep, _, korder, ktot, _ = L.conv_tiling(dtype, Cin, 512, 1)
    assert korder == 1
    st = torch.cuda.current_stream().cuda_stream
    we = torch.zeros(E, edge_pad, Cin, 1, 1)
    we[:, :win] = w[uch:].view(E, win, Cin, 1, 1)
    we = we.view(E * edge_pad, Cin, 1, 1).contiguous().to(dev)
    packed = torch.empty(E * edge_pad, ktot, dtype=tdt, device=dev)
    L.check(lib.ppn_pack_weight(dtype, we.data_ptr(), E * edge_pad, Cin, 1, E * edge_pad, ktot, korder, kstep,
                                packed.data_ptr(), st))
    be = torch.full((E, edge_pad), 1000.0)                     # a pad row that competed would win every window
    be[:, :win] = bias[uch:].view(E, win)
    be = be.view(-1).contiguous().to(dev)
    xs = x.permute(0, 2, 3, 1).contiguous().to(dev, tdt)
    zero = torch.zeros(64, device=dev)
    keys = torch.full((B, E, H, W), 0x7EADBEEF7EADBEEF, dtype=torch.int64, device=dev)   # garbage: must be overwritten
    d = L.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.cin = dtype, B, H, W, Cin
    d.out_h, d.out_w, d.cout = H, W, E * win
    d.ksize, d.stride, d.dilation, d.pad = 1, 1, 1, 0
    d.k_total, d.cout_pad, d.act1, d.out_nchw_f32 = ktot, E * edge_pad, 3, 1
    d.src, d.weight, d.zero_page, d.shift1 = xs.data_ptr(), packed.data_ptr(), zero.data_ptr(), be.data_ptr()
    d.argmax_keys, d.limb_window, d.limb_edge_pad = keys.data_ptr(), win, edge_pad
    L.check(lib.ppn_conv2d_fused(C.byref(d), st), "ppn_conv2d_fused")
    torch.cuda.synchronize()
    return keys.cpu(), lib.ppn_last_conv_kernel().decode()


def ref_conv(x, w, stride=1, dil=1, pad=0, s1=None, b1=None, act1=0, residual=None, s2=None, b2=None, act2=0):
    y = F.conv2d(x.double(), w.double(), None, stride, pad, dil)
    if s1 is not None:
        y = y * s1.double().view(1, -1, 1, 1)
    if b1 is not None:
        y = y + b1.double().view(1, -1, 1, 1)
    y = _act(y, act1)
    if residual is not None:
        y = y + residual.double()
    u = y
    if s2 is not None:
        u = u * s2.double().view(1, -1, 1, 1) + b2.double().view(1, -1, 1, 1)
    u = _act(u, act2)
    return y.float(), u.float()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """Round inputs to the storage type first in the 16-bit modes so that only accumulation/epilogue error is measured."""
    from pytorch_pose_proposal_network_amd import lib as L
    if dtype == L.PPN_F16:
        return t.to(torch.float16).float()
    return t.to(torch.bfloat16).float() if dtype == L.PPN_BF16 else t


CASES = [
    # name,           B, Cin, Cout, H,  W,  k, s, d, p
    ("L6_512_d4",      1, 512, 512, 12, 12, 3, 1, 4, 4),
    ("L5_128_256_d2",  2, 128, 256, 10, 14, 3, 1, 2, 2),
    ("L4_s2",          2, 64, 128, 18, 22, 3, 2, 1, 1),
    ("ds_1x1_s2",      2, 64, 128, 18, 22, 1, 2, 1, 0),
    ("L3_ds_cin32",    1, 32, 64, 20, 20, 1, 2, 1, 0),
    ("L3_c1_cin32",    1, 32, 64, 20, 24, 3, 2, 1, 1),
    ("L1_cin16",       1, 16, 16, 24, 40, 3, 1, 1, 1),
    ("L2_cin16_s2",    2, 16, 32, 26, 30, 3, 2, 1, 1),
    ("neck_1x1",       3, 512, 128, 6, 6, 1, 1, 1, 0),
    ("odd_M",          1, 64, 64, 7, 9, 3, 1, 1, 1),
    ("d54_2048_512",   1, 2048, 512, 6, 6, 3, 1, 2, 2),
]


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bn_relu(case, dtype_name):
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    _, B, Cin, Cout, H, W, k, s, dl, p = case
    x = q(rnd(B, Cin, H, W, seed=1), dtype)
    w = q(rnd(Cout, Cin, k, k, seed=2, scale=(2.0 / (Cin * k * k)) ** 0.5), dtype)
    s1 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(3))
    b1 = rnd(Cout, seed=4, scale=0.3)
    raw, _ = run_conv(x, w, dtype, s, dl, p, s1, b1, act1=1)
    ref, _ = ref_conv(x, w, s, dl, p, s1, b1, act1=1)
    tol = (F32_TOL if dtype == L.PPN_F32 else BF16_TOL) * max(1.0, float(ref.abs().max()))
    assert raw.shape == ref.shape
    assert float((raw - ref).abs().max()) <= tol


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_preact_block_epilogue(dtype_name):
    """conv2 of a BasicBlock: raw = acc + residual, act = relu(bn_next(raw))  (drn.py:42-57)."""
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    B, Cc, H, W = 2, 128, 12, 12
    x, res = q(rnd(B, Cc, H, W, seed=5), dtype), q(rnd(B, Cc, H, W, seed=6), dtype)
    w = q(rnd(Cc, Cc, 3, 3, seed=7, scale=0.03), dtype)
    s2, b2 = 0.5 + torch.rand(Cc), rnd(Cc, seed=8, scale=0.2)
    raw, act = run_conv(x, w, dtype, 1, 2, 2, residual=res, s2=s2, b2=b2, act2=1, want_act=True)
    rr, ra = ref_conv(x, w, 1, 2, 2, residual=res, s2=s2, b2=b2, act2=1)
    tol = F32_TOL * 10 if dtype == L.PPN_F32 else BF16_TOL * 2
    assert float((raw - rr).abs().max()) <= tol * max(1.0, float(rr.abs().max()))
    assert float((act - ra).abs().max()) <= tol * max(1.0, float(ra.abs().max()))
    # act only (no raw store) -- Bottleneck tail: relu(bn3(conv) + residual)
    _, act2 = run_conv(x, w, dtype, 1, 2, 2, s1=s2, b1=b2, residual=res, act2=1, want_raw=False, want_act=True)
    y = F.relu(F.conv2d(x, w, None, 1, 2, 2) * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1) + res)
    assert float((act2 - y).abs().max()) <= tol * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_head_nchw_sigmoid_7605(dtype_name):
    """conv3: 1x1 512->7605 + bias -> sigmoid, f32 NCHW output (model.py:133-136)."""
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    for (B, H, W) in ((2, 24, 24), (1, 6, 6), (1, 5, 7)):
        x = q(rnd(B, 512, H, W, seed=9), dtype)
        w = q(rnd(7605, 512, 1, 1, seed=10, scale=0.06), dtype)
        bias = rnd(7605, seed=11, scale=0.1)
        raw, _ = run_conv(x, w, dtype, b1=bias, act1=3, nchw=True)
        ref, _ = ref_conv(x, w, b1=bias, act1=3)
        assert raw.shape == (B, 7605, H, W)
        assert float((raw - ref).abs().max()) <= (2e-6 if dtype == L.PPN_F32 else 5e-3)


def test_lrelu_bias_bn():
    from pytorch_pose_proposal_network_amd import lib as L
    x, w = rnd(1, 512, 6, 6, seed=12), rnd(512, 512, 3, 3, seed=13, scale=0.02)
    s1, b1 = 0.5 + torch.rand(512), rnd(512, seed=14)
    raw, _ = run_conv(x, w, L.PPN_F32, 1, 1, 1, s1, b1, act1=2)
    ref, _ = ref_conv(x, w, 1, 1, 1, s1, b1, act1=2)
    assert float((raw - ref).abs().max()) <= F32_TOL * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
@pytest.mark.parametrize("u8", [True, False])
def test_stem7x7(dtype_name, u8):
    from pytorch_pose_proposal_network_amd import lib as L, prng
    lib = L.load()
    dtype = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    tdt = torch.float32 if dtype == L.PPN_F32 else torch.bfloat16
    B, H, W = 2, 37, 70            # not multiples of the 16x64 tile
    frames = torch.from_numpy(prng.u8_frames(5, B, (H, W)))
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    xn = (frames.permute(0, 3, 1, 2).float() - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    w = rnd(16, 3, 7, 7, seed=15, scale=0.002)
    s1, b1 = 0.5 + torch.rand(16), rnd(16, seed=16, scale=0.3)
    dev = torch.device("cuda")
    out = torch.full((B, H, W, 16), float("nan"), device=dev).to(tdt)
    src = frames.to(dev) if u8 else xn.contiguous().to(dev)
    wd, sd_, bd = w.to(dev), s1.to(dev), b1.to(dev)
    m3, s3 = (C.c_float * 3)(0.485, 0.456, 0.406), (C.c_float * 3)(0.229, 0.224, 0.225)
    L.check(lib.ppn_stem7x7(dtype, int(u8), src.data_ptr(), B, H, W, wd.data_ptr(), sd_.data_ptr(), bd.data_ptr(),
                            m3, s3, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "ppn_stem7x7")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(xn.double(), w.double(), None, 1, 3) * s1.double().view(1, -1, 1, 1) + b1.double().view(1, -1, 1, 1)).float()
    tol = (1e-5 if dtype == L.PPN_F32 else 2e-2) * float(ref.abs().max())
    assert float((got - ref).abs().max()) <= tol


def test_conv_rejects_bad_descriptors():
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()
    d = L.ConvDesc()
    assert lib.ppn_conv2d_fused(C.byref(d), None) != 0
    assert b"" != lib.ppn_last_error()
    assert lib.ppn_conv2d_fused(None, None) != 0


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_stem3x3_dual_output(dtype_name):
    """backbone.2.0 (16->32, stride 2) with the second, pre-activation output of layer3's first block."""
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    x = q(rnd(2, 16, 37, 70, seed=21), dtype)
    w = q(rnd(32, 16, 3, 3, seed=22, scale=0.1), dtype)
    s1, b1 = 0.5 + torch.rand(32), rnd(32, seed=23, scale=0.3)
    s2, b2 = 0.5 + torch.rand(32), rnd(32, seed=24, scale=0.3)
    raw, act = run_conv(x, w, dtype, 2, 1, 1, s1, b1, act1=1, s2=s2, b2=b2, act2=1, want_act=True)
    rr, ra = ref_conv(x, w, 2, 1, 1, s1, b1, act1=1, s2=s2, b2=b2, act2=1)
    tol = (F32_TOL if dtype == L.PPN_F32 else BF16_TOL) * max(1.0, float(rr.abs().max()))
    assert raw.shape == rr.shape == (2, 32, 19, 35)
    assert float((raw - rr).abs().max()) <= tol and float((act - ra).abs().max()) <= 2 * tol


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    # (B, Cmid, Cin2, Hout, stride2, dil)   main conv: 3x3 Cmid->Cout on [Hout,Hout]; shortcut 1x1 stride2 on x2
    (2, 128, 64, 24, 2, 1),      # layer4.0: 64 -> 128, stride 2
    (1, 256, 128, 12, 2, 1),     # layer5.0
    (2, 512, 256, 9, 1, 2),      # layer6.0: stride 1, dilated main conv, ragged pixel count
    (1, 128, 192, 7, 2, 1),      # odd source size (H2 = 13), 3 shortcut slabs
])
def test_fused_projection_shortcut(dtype_name, case):
    """conv2 + BasicBlock.downsample (drn.py:53-54) as one GEMM: out = conv3x3(mid) + conv1x1_s(x2) + shift,
    then the pre-activation second output."""
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    B, Cm, C2, Ho, s2, dil = case
    H2 = Ho * s2 - (1 if s2 == 2 and Ho % 2 else 0)
    assert (H2 - 1) // s2 + 1 == Ho
    x = rnd(B, Cm, Ho, Ho, seed=1)
    w = rnd(Cm, Cm, 3, 3, seed=2, scale=(Cm * 9) ** -0.5)
    x2 = rnd(B, C2, H2, H2, seed=3)
    w2 = rnd(Cm, C2, 1, 1, seed=4, scale=C2 ** -0.5)
    b1, sc2, sh2 = rnd(Cm, seed=5), rnd(Cm, seed=6) + 1.5, rnd(Cm, seed=7)
    if dtype == L.PPN_BF16:
        x, w, x2, w2 = (t.to(torch.bfloat16).float() for t in (x, w, x2, w2))
    raw, act = run_conv(x, w, dtype, 1, dil, dil, b1=b1, s2=sc2, b2=sh2, act2=1, want_act=True,
                        shortcut=(x2, w2, s2))
    y = F.conv2d(x.double(), w.double(), None, 1, dil, dil) + F.conv2d(x2.double(), w2.double(), None, s2)
    y = y + b1.double().view(1, -1, 1, 1)
    u = torch.relu(y * sc2.double().view(1, -1, 1, 1) + sh2.double().view(1, -1, 1, 1))
    tol = 2e-5 if dtype == L.PPN_F32 else 2e-2
    assert (raw.double() - y).abs().max() <= tol * max(1.0, y.abs().max().item())
    assert (act.double() - u).abs().max() <= tol * max(1.0, u.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 37, 70), (1, 96, 96), (3, 384, 384), (1, 33, 200)], ids=str)
@pytest.mark.parametrize("u8", [True, False])
def test_fused_stem_equals_layer_by_layer(shape, u8):
    """csrc/stem012.hip (layer0 + layer1 + layer2 in one launch, bf16) == ppn_stem7x7 -> stem3x3<16,1> -> stem3x3<32,2>
    BIT FOR BIT (same MFMA sequences, same bf16 rounding points), both outputs, for ragged sizes, several bands and
    strips; and within bf16 tolerance of the fp64 reference of drn.py:123-133 + the first block's relu(bn1(x))."""
    from pytorch_pose_proposal_network_amd import lib as L, prng
    lib = L.load()
    B, H, W = shape
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    frames = torch.from_numpy(prng.u8_frames(17, B, (H, W)))
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    xn = ((frames.permute(0, 3, 1, 2).float() - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)).contiguous()
    w0, w1, w2 = rnd(16, 3, 7, 7, seed=31, scale=0.002), rnd(16, 16, 3, 3, seed=32, scale=0.12), rnd(32, 16, 3, 3, seed=33, scale=0.12)
    gen = torch.Generator().manual_seed(34)
    s = [0.5 + torch.rand(n, generator=gen) for n in (16, 16, 32, 32)]
    b = [rnd(n, seed=35 + i, scale=0.3) for i, n in enumerate((16, 16, 32, 32))]
    d = lambda t: t.contiguous().to(dev)
    src = d(frames) if u8 else d(xn)
    w0d, w1d, w2d = d(w0), d(w1), d(w2)
    sd_, bd = [d(t) for t in s], [d(t) for t in b]
    m3, s3 = (C.c_float * 3)(0.485, 0.456, 0.406), (C.c_float * 3)(0.229, 0.224, 0.225)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    nan = lambda *sh: torch.full(sh, float("nan"), device=dev).to(torch.bfloat16)
    # ---- one launch ----
    raw_f, act_f = nan(B, Ho, Wo, 32), nan(B, Ho, Wo, 32)
    L.check(lib.ppn_stem012(int(u8), src.data_ptr(), B, H, W, w0d.data_ptr(), sd_[0].data_ptr(), bd[0].data_ptr(), m3, s3,
                            w1d.data_ptr(), sd_[1].data_ptr(), bd[1].data_ptr(), w2d.data_ptr(), sd_[2].data_ptr(),
                            bd[2].data_ptr(), sd_[3].data_ptr(), bd[3].data_ptr(), raw_f.data_ptr(), act_f.data_ptr(), st),
            "ppn_stem012")
    # ---- layer by layer ----
    t0, t1 = nan(B, H, W, 16), nan(B, H, W, 16)
    raw_s, act_s = nan(B, Ho, Wo, 32), nan(B, Ho, Wo, 32)
    L.check(lib.ppn_stem7x7(L.PPN_BF16, int(u8), src.data_ptr(), B, H, W, w0d.data_ptr(), sd_[0].data_ptr(),
                            bd[0].data_ptr(), m3, s3, t0.data_ptr(), st), "ppn_stem7x7")
    zero = torch.zeros(64, device=dev)

    def conv3(x, wd, cout, stride, s1, b1, s2, b2, out_raw, out_act):
        dsc = L.ConvDesc()
        dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = L.PPN_BF16, B, H, W, 16
        dsc.out_h, dsc.out_w, dsc.cout = out_raw.shape[1], out_raw.shape[2], cout
        dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = 3, stride, 1, 1
        dsc.k_total, dsc.cout_pad, dsc.act1, dsc.act2 = 144, cout, 1, (1 if out_act is not None else 0)
        dsc.src, dsc.weight, dsc.zero_page = x.data_ptr(), wd.data_ptr(), zero.data_ptr()
        dsc.scale1, dsc.shift1, dsc.out_raw = s1.data_ptr(), b1.data_ptr(), out_raw.data_ptr()
        if out_act is not None:
            dsc.scale2, dsc.shift2, dsc.out_act = s2.data_ptr(), b2.data_ptr(), out_act.data_ptr()
        L.check(lib.ppn_conv2d_fused(C.byref(dsc), st), "ppn_conv2d_fused")
        assert lib.ppn_last_conv_kernel().decode().startswith("stem3x3_kernel")

    conv3(t0, w1d, 16, 1, sd_[1], bd[1], None, None, t1, None)
    conv3(t1, w2d, 32, 2, sd_[2], bd[2], sd_[3], bd[3], raw_s, act_s)
    torch.cuda.synchronize()
    assert not torch.isnan(raw_f.float()).any() and not torch.isnan(act_f.float()).any()
    assert torch.equal(raw_f, raw_s), float((raw_f.float() - raw_s.float()).abs().max())
    assert torch.equal(act_f, act_s)
    # ---- fp64 reference (bf16 tolerance) ----
    v = lambda t: t.double().view(1, -1, 1, 1)
    y = F.relu(F.conv2d(xn.double(), w0.double(), None, 1, 3) * v(s[0]) + v(b[0]))
    y = F.relu(F.conv2d(y, w1.double(), None, 1, 1) * v(s[1]) + v(b[1]))
    y = F.relu(F.conv2d(y, w2.double(), None, 2, 1) * v(s[2]) + v(b[2]))
    u = F.relu(y * v(s[3]) + v(b[3]))
    got_y, got_u = raw_f.float().cpu().permute(0, 3, 1, 2).double(), act_f.float().cpu().permute(0, 3, 1, 2).double()
    assert float((got_y - y).abs().max()) <= 4e-2 * max(1.0, float(y.abs().max()))
    assert float((got_u - u).abs().max()) <= 4e-2 * max(1.0, float(u.abs().max()))


def test_f16_stores_stay_finite_past_65504():
    """The f16 mode stores RAW pre-activation residual streams in IEEE half: a value past 65504 must be stored as the
    largest finite half, not +-inf (an inf becomes a NaN in the next residual add / BN affine and a garbage decode).
    Both outputs of the chunked epilogue, the single-output fast epilogue and the small-channel kernel."""
    from pytorch_pose_proposal_network_amd import lib as L
    g = torch.Generator().manual_seed(3)
    for cin, cout in ((64, 128), (32, 64)):
        x = torch.randn(2, cin, 12, 12, generator=g) * 100.0
        w = torch.randn(cout, cin, 3, 3, generator=g)
        res = torch.randn(2, cout, 12, 12, generator=g) * 1000.0
        # dual-output launch with a residual (chunked f32 epilogue): |v| reaches ~2e5 here
        raw, act = run_conv(x, w, L.PPN_F16, pad=1, s1=torch.full((cout,), 16.0), residual=res, s2=torch.full((cout,), 2.0),
                            act2=1, want_act=True)
        ref = F.conv2d(x.half().float(), w.half().float(), None, 1, 1) * 16.0 + res.half().float()
        assert float(ref.abs().max()) > 1e5                                   # the case really overflows half
        assert torch.isfinite(raw).all() and torch.isfinite(act).all()
        over = ref.abs() > 70000.0
        assert torch.equal(raw[over].abs(), torch.full_like(raw[over], 65504.0))
        assert torch.equal(torch.sign(raw[over]), torch.sign(ref[over]))
        ok = ref.abs() < 60000.0
        assert float((raw[ok] - ref[ok]).abs().max()) <= 3e-3 * 60000.0
        # single-output launch (the bf16/f16 fast epilogue where the tile has one)
        raw1, _ = run_conv(x, w, L.PPN_F16, pad=1, s1=torch.full((cout,), 16.0))
        assert torch.isfinite(raw1).all() and float(raw1.abs().max()) == 65504.0
