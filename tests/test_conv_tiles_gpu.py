"""GPU parity of EVERY tile instantiation of the large-tile convolution kernel (csrc/conv_big.hip).

The tile of a launch is picked from M = B*Ho*Wo (big_tile_for): the small problems of test_conv_gpu.py only ever
reach 128x128 / 128x64, while a batch-32 384x384 forward (bench.py, BASELINE configs[1]) runs 192x256, 192x128
(plain and with the fused projection shortcut), 256x128 and 128x64.  Here

* every (bp, bc) is FORCED through ppn_set_conv_tile_override on problems with a ragged last pixel tile and a
  partial last channel tile, in f32 and bf16, for each epilogue: single output (bf16: the single-pass epilogue where
  it fits, else the chunked one), residual + second (pre-activation) output, fused projection shortcut, and the
  NCHW sigmoid head with the decode's arg-max keys;
* the layer shapes of the batch-32 forward run at FULL size with the automatic choice and are compared with an
  fp64 reference on sampled output rows (first / last tile, tile seams, the ragged tail).

Every test asserts and prints the kernel instantiation it hit (ppn_last_conv_kernel).
Reference: plain PyTorch CPU fp64 conv of the same layer (/root/reference/drn.py:42-57, model.py:104-136)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_conv_gpu import BF16_TOL, F16_TOL, F32_TOL, q, ref_conv, rnd, run_conv, run_edge_argmax

pytestmark = pytest.mark.gpu

TILES = [(256, 256), (192, 256), (144, 256), (128, 256), (256, 128), (192, 128), (128, 128), (256, 64), (128, 64)]


@pytest.fixture
def force_tile():
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()

    def force(bp, bc):
        L.check(lib.ppn_set_conv_tile_override(bp, bc), "ppn_set_conv_tile_override")

    yield force
    L.check(lib.ppn_set_conv_tile_override(0, 0), "ppn_set_conv_tile_override")


def _dt(name):
    from pytorch_pose_proposal_network_amd import lib as L
    return {"f32": L.PPN_F32, "bf16": L.PPN_BF16, "f16": L.PPN_F16}[name]


def _tol(name):
    return {"f32": F32_TOL, "bf16": BF16_TOL, "f16": F16_TOL}[name]


def _kname(dtype_name, bp, bc, sc=False):
    return "conv_igemm_big_kernel<%s, %d, %d, 8, %s>" % ({"f32": "float", "bf16": "__bf16", "f16": "_Float16"}[dtype_name], bp, bc,
                                                         "true" if sc else "false")


def _cout_for(bc):
    # two channel tiles, the second one partial (cout_pad is a multiple of the class's largest tile)
    return {256: 320, 128: 200, 64: 200}[bc]


@pytest.mark.parametrize("dtype_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("tile", TILES, ids=["%dx%d" % t for t in TILES])
def test_forced_tile_single_output(force_tile, tile, dtype_name):
    """conv + BN + ReLU, one output: bf16 takes the single-pass epilogue where the tile fits it."""
    bp, bc = tile
    dtype = _dt(dtype_name)
    force_tile(bp, bc)
    B, Cin, H, W, Cout = 2, 128, 20, 23, _cout_for(bc)          # M = 920: ragged for 128 / 192 / 256
    x = q(rnd(B, Cin, H, W, seed=31), dtype)
    w = q(rnd(Cout, Cin, 3, 3, seed=32, scale=(2.0 / (Cin * 9)) ** 0.5), dtype)
    s1 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(33))
    b1 = rnd(Cout, seed=34, scale=0.3)
    info = {}
    raw, _ = run_conv(x, w, dtype, 1, 2, 2, s1, b1, act1=1, info=info)
    ref, _ = ref_conv(x, w, 1, 2, 2, s1, b1, act1=1)
    print(info["kernel"])
    assert info["kernel"] == _kname(dtype_name, bp, bc)
    tol = _tol(dtype_name) * max(1.0, float(ref.abs().max()))
    assert not torch.isnan(raw).any()
    assert float((raw - ref).abs().max()) <= tol


@pytest.mark.parametrize("dtype_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("tile", TILES, ids=["%dx%d" % t for t in TILES])
def test_forced_tile_residual_dual_output(force_tile, tile, dtype_name):
    """conv2 of a BasicBlock: raw = acc + residual, act = relu(bn_next(raw)) -- the chunked f32 epilogue."""
    bp, bc = tile
    dtype = _dt(dtype_name)
    force_tile(bp, bc)
    B, Cc, H, W = 3, _cout_for(bc), 17, 19                        # M = 969
    Cin = 64
    x, res = q(rnd(B, Cin, H, W, seed=41), dtype), q(rnd(B, Cc, H, W, seed=42), dtype)
    w = q(rnd(Cc, Cin, 3, 3, seed=43, scale=0.04), dtype)
    s2, b2 = 0.5 + torch.rand(Cc, generator=torch.Generator().manual_seed(44)), rnd(Cc, seed=45, scale=0.2)
    info = {}
    raw, act = run_conv(x, w, dtype, 1, 1, 1, residual=res, s2=s2, b2=b2, act2=1, want_act=True, info=info)
    rr, ra = ref_conv(x, w, 1, 1, 1, residual=res, s2=s2, b2=b2, act2=1)
    print(info["kernel"])
    assert info["kernel"] == _kname(dtype_name, bp, bc)
    tol = F32_TOL * 10 if dtype_name == "f32" else _tol(dtype_name) * 2
    assert not torch.isnan(raw).any() and not torch.isnan(act).any()
    assert float((raw - rr).abs().max()) <= tol * max(1.0, float(rr.abs().max()))
    assert float((act - ra).abs().max()) <= tol * max(1.0, float(ra.abs().max()))
    # second output only (Bottleneck tail, relu(bn3(conv) + residual)): no raw store
    _, act2 = run_conv(x, w, dtype, 1, 1, 1, s1=s2, b1=b2, residual=res, act2=1, want_raw=False, want_act=True)
    y = F.relu(F.conv2d(x.double(), w.double(), None, 1, 1, 1) * s2.double().view(1, -1, 1, 1) +
               b2.double().view(1, -1, 1, 1) + res.double()).float()
    assert float((act2 - y).abs().max()) <= tol * max(1.0, float(y.abs().max()))


SC_TILES = [t for t in TILES if t[1] >= 128]


@pytest.mark.parametrize("dtype_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("tile", SC_TILES, ids=["%dx%d" % t for t in SC_TILES])
def test_forced_tile_fused_shortcut(force_tile, tile, dtype_name):
    """conv2 + BasicBlock.downsample as one GEMM (the SC=true instantiations), with the pre-activation output."""
    bp, bc = tile
    dtype = _dt(dtype_name)
    force_tile(bp, bc)
    B, Cm, C2, Ho, s2 = 2, (256 if bc == 256 else 128), 64, 21, 2    # M = 882, source 41x41 (odd)
    H2 = Ho * s2 - 1
    x = q(rnd(B, Cm, Ho, Ho, seed=51), dtype)
    w = q(rnd(Cm, Cm, 3, 3, seed=52, scale=(Cm * 9) ** -0.5), dtype)
    x2 = q(rnd(B, C2, H2, H2, seed=53), dtype)
    w2 = q(rnd(Cm, C2, 1, 1, seed=54, scale=C2 ** -0.5), dtype)
    b1, sc2, sh2 = rnd(Cm, seed=55), rnd(Cm, seed=56) + 1.5, rnd(Cm, seed=57)
    info = {}
    raw, act = run_conv(x, w, dtype, 1, 1, 1, b1=b1, s2=sc2, b2=sh2, act2=1, want_act=True, shortcut=(x2, w2, s2),
                        info=info)
    print(info["kernel"])
    assert info["kernel"] == _kname(dtype_name, bp, bc, sc=True)
    y = F.conv2d(x.double(), w.double(), None, 1, 1, 1) + F.conv2d(x2.double(), w2.double(), None, s2)
    y = y + b1.double().view(1, -1, 1, 1)
    u = torch.relu(y * sc2.double().view(1, -1, 1, 1) + sh2.double().view(1, -1, 1, 1))
    tol = _tol(dtype_name)
    assert (raw.double() - y).abs().max() <= tol * max(1.0, y.abs().max().item())
    assert (act.double() - u).abs().max() <= tol * max(1.0, u.abs().max().item())


def _check_keys(info, head, uch, win):
    """keys[b,e,cell] = (value bits << 32) | ~s must be the FIRST maximum of head[b, uch+e*win : uch+(e+1)*win, cell]
    (np.argmax semantics, /root/reference/datatest.py:113), and the compact unary tensor the first uch channels."""
    keys = info["keys"].numpy().astype(np.int64)
    B, E = keys.shape[:2]
    hv = head.numpy()
    val = ((keys >> 32) & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    idx = (0xFFFFFFFF - (keys & 0xFFFFFFFF)).astype(np.int64)
    limbs = hv[:, uch:uch + E * win].reshape(B, E, win, *hv.shape[2:])
    assert np.array_equal(idx, limbs.argmax(axis=2))
    assert np.array_equal(val, limbs.max(axis=2))
    assert np.array_equal(info["unary"].numpy(), hv[:, :uch])


HEAD_TILES = [t for t in TILES if t[1] >= 128]


@pytest.mark.parametrize("dtype_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("tile", HEAD_TILES, ids=["%dx%d" % t for t in HEAD_TILES])
def test_forced_tile_head_nchw_argmax(force_tile, tile, dtype_name):
    """conv3-shaped launch (1x1 + bias + sigmoid, f32 NCHW) with the fused arg-max keys, head materialised too."""
    bp, bc = tile
    dtype = _dt(dtype_name)
    force_tile(bp, bc)
    uch, win, E = 108, 441, 3
    Cout = uch + E * win                                            # 1431: partial last channel tile
    B, H, W = 3, 15, 13                                             # M = 585
    x = q(rnd(B, 512, H, W, seed=61), dtype)
    w = q(rnd(Cout, 512, 1, 1, seed=62, scale=0.06), dtype)
    bias = rnd(Cout, seed=63, scale=0.1)
    info = {}
    raw, _ = run_conv(x, w, dtype, b1=bias, act1=3, nchw=True, argmax=(uch, win), info=info)
    print(info["kernel"])
    assert info["kernel"] == _kname(dtype_name, bp, bc)
    ref, _ = ref_conv(x, w, b1=bias, act1=3)
    assert float((raw - ref).abs().max()) <= {"f32": 2e-6, "bf16": 5e-3, "f16": 8e-4}[dtype_name]
    _check_keys(info, raw, uch, win)
    if (bp, bc) == (192, 128):                                    # the edge-aligned tile takes the same decisions
        keys_e, kn = run_edge_argmax(x, w, bias, dtype, uch, win)
        assert "head_limb_argmax_kernel" in kn and torch.equal(keys_e, info["keys"])
    # keys only (the benchmarked path: the head tensor is not written)
    info2 = {}
    run_conv(x, w, dtype, b1=bias, act1=3, nchw=True, argmax=(uch, win), want_raw=False, info=info2)
    assert torch.equal(info2["keys"], info["keys"]) and torch.equal(info2["unary"], info["unary"])


# ---- the batch-32 384x384 layer shapes at full size, automatic tile choice ----------------------------------------

def _sampled_rows(M, tiles=(128, 192, 256), n_random=96, seed=0):
    """output pixel indices at tile seams of every tile height, the first / last rows and a random sample."""
    rows = {0, 1, M - 1, M - 2}
    for t in tiles:
        last = (M - 1) // t * t
        for base in (t, 2 * t, last, (M // t // 2) * t):
            for d in (-1, 0, 1, t - 1):
                r = base + d
                if 0 <= r < M:
                    rows.add(r)
    g = np.random.default_rng(seed)
    rows.update(int(v) for v in g.integers(0, M, n_random))
    if M > 65536:                                                    # the two-segment launch's cut (ppn_conv_split)
        rows.update(65536 + d for d in (-256, -129, -2, -1, 0, 1, 127, 128) if 65536 + d < M)
        rows.update(int(v) for v in g.integers(65536, M, 32))
    return np.array(sorted(rows))


def _ref_rows(x, w, rows, Ho, Wo, stride, dil, pad):
    """fp64 conv outputs [len(rows), Cout] at flat output pixels `rows` (b*Ho*Wo + oy*Wo + ox); x NCHW, w OIHW."""
    B, Cin, H, W = x.shape
    k = w.shape[2]
    xp = F.pad(x.double(), (pad, pad, pad, pad))
    wm = w.double().reshape(w.shape[0], -1)                         # [Cout, Cin*k*k]
    out = torch.empty(len(rows), w.shape[0], dtype=torch.float64)
    for i, r in enumerate(rows):
        b, rem = divmod(int(r), Ho * Wo)
        oy, ox = divmod(rem, Wo)
        patch = xp[b, :, oy * stride: oy * stride + dil * (k - 1) + 1: dil, ox * stride: ox * stride + dil * (k - 1) + 1: dil]
        out[i] = wm @ patch.reshape(-1)
    return out


FULL = [
    # name,                    B,  Cin, Cout, H,  W,  k, s, d, p, expected kernel tile
    ("L7_512_d2_48x48",        32, 512, 512, 48, 48, 3, 1, 2, 2, (192, 256)),
    ("L6_c1_256_512_d4",       32, 256, 512, 48, 48, 3, 1, 4, 4, (192, 256)),
    ("L5_c2_256_d2",           32, 256, 256, 48, 48, 3, 1, 2, 2, (192, 128)),     # 3 rounds; 144x256 (2 rounds) measured slower
    ("L4_c1_64_128_s2",        32, 64, 128, 96, 96, 3, 2, 1, 1, (192, 128)),
    ("L3_c2_64_96x96",         32, 64, 64, 96, 96, 3, 1, 1, 1, None),
    # M = 18 432: 128 x 2 tiles of 144 x 256 = exactly one workgroup per CU (192 x 256 leaves 64 CUs idle)
    ("B1_c1_512_s2_24x24",     32, 512, 512, 48, 48, 3, 2, 1, 1, (144, 256)),
    ("B2_c1_512_24x24",        32, 512, 512, 24, 24, 3, 1, 1, 1, (144, 256)),
    ("ragged_M_31",            31, 512, 512, 47, 45, 3, 1, 2, 2, None),
    # layer3's first block on the 32-channel stem output (bf16: the small-channel kernel of csrc/conv.hip) at full size
    ("L3_c1_32_64_s2_192",     32, 32, 64, 192, 192, 3, 2, 1, 1, None),
    ("L3_ds_32_64_1x1_s2",     32, 32, 64, 192, 192, 1, 2, 1, 0, None),
    ("c32_5x5_d2_ragged",      3, 32, 96, 45, 51, 5, 1, 2, 4, None),
]


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
@pytest.mark.parametrize("case", FULL, ids=[c[0] for c in FULL])
def test_full_size_layer_sampled(case, dtype_name):
    """Batch-32 layer shapes of DRN-D-22 at 384x384 with the AUTOMATIC tile (the instantiations bench.py runs):
    single-output conv+BN+ReLU, then the same conv with residual + second output; sampled rows vs fp64."""
    name, B, Cin, Cout, H, W, k, s, dl, p, want = case
    dtype = _dt(dtype_name)
    eff = dl * (k - 1) + 1
    Ho, Wo = (H + 2 * p - eff) // s + 1, (W + 2 * p - eff) // s + 1
    M = B * Ho * Wo
    x = q(rnd(B, Cin, H, W, seed=71), dtype)
    w = q(rnd(Cout, Cin, k, k, seed=72, scale=(2.0 / (Cin * k * k)) ** 0.5), dtype)
    s1 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(73))
    b1 = rnd(Cout, seed=74, scale=0.3)
    rows = _sampled_rows(M)
    acc = _ref_rows(x, w, rows, Ho, Wo, s, dl, p)
    info = {}
    raw, _ = run_conv(x, w, dtype, s, dl, p, s1, b1, act1=1, info=info)
    print(name, dtype_name, "M =", M, info["kernel"])
    if want is not None:
        assert info["kernel"] == _kname(dtype_name, *want)

    got = raw.permute(0, 2, 3, 1).reshape(M, Cout)[rows].double()
    ref = torch.relu(acc * s1.double() + b1.double())
    tol = _tol(dtype_name) * max(1.0, float(ref.abs().max()))
    assert not torch.isnan(raw).any()
    assert float((got - ref).abs().max()) <= tol
    # residual + second output on the same shape (chunked epilogue of the same instantiation)
    res = q(rnd(B, Cout, Ho, Wo, seed=75), dtype)
    s2, b2 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(76)), rnd(Cout, seed=77, scale=0.2)
    raw2, act2 = run_conv(x, w, dtype, s, dl, p, residual=res, s2=s2, b2=b2, act2=1, want_act=True)
    rr = acc + res.permute(0, 2, 3, 1).reshape(M, Cout)[rows].double()
    ra = torch.relu(rr * s2.double() + b2.double())
    tol2 = (F32_TOL * 10 if dtype_name == "f32" else BF16_TOL * 2)
    assert not torch.isnan(raw2).any() and not torch.isnan(act2).any()
    assert float((raw2.permute(0, 2, 3, 1).reshape(M, Cout)[rows].double() - rr).abs().max()) <= tol2 * max(1.0, float(rr.abs().max()))
    assert float((act2.permute(0, 2, 3, 1).reshape(M, Cout)[rows].double() - ra).abs().max()) <= tol2 * max(1.0, float(ra.abs().max()))


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_full_size_fused_shortcut_sampled(dtype_name):
    """layer4.0 conv2 + projection shortcut at batch 32 (M = 73 728): the 192x128 SC=true instantiation."""
    dtype = _dt(dtype_name)
    B, Cm, C2, Ho, s2 = 32, 128, 64, 48, 2
    H2 = Ho * s2
    M = B * Ho * Ho
    x = q(rnd(B, Cm, Ho, Ho, seed=81), dtype)
    w = q(rnd(Cm, Cm, 3, 3, seed=82, scale=(Cm * 9) ** -0.5), dtype)
    x2 = q(rnd(B, C2, H2, H2, seed=83), dtype)
    w2 = q(rnd(Cm, C2, 1, 1, seed=84, scale=C2 ** -0.5), dtype)
    b1, sc2, sh2 = rnd(Cm, seed=85), rnd(Cm, seed=86) + 1.5, rnd(Cm, seed=87)
    info = {}
    raw, act = run_conv(x, w, dtype, 1, 1, 1, b1=b1, s2=sc2, b2=sh2, act2=1, want_act=True, shortcut=(x2, w2, s2),
                        info=info)
    print(info["kernel"])
    assert info["kernel"] == _kname(dtype_name, 192, 128, sc=True)
    rows = _sampled_rows(M)
    y = _ref_rows(x, w, rows, Ho, Ho, 1, 1, 1) + _ref_rows(x2, w2, rows, Ho, Ho, s2, 1, 0) + b1.double()
    u = torch.relu(y * sc2.double() + sh2.double())
    tol = 2e-5 if dtype_name == "f32" else 2e-2
    assert (raw.permute(0, 2, 3, 1).reshape(M, Cm)[rows].double() - y).abs().max() <= tol * max(1.0, y.abs().max().item())
    assert (act.permute(0, 2, 3, 1).reshape(M, Cm)[rows].double() - u).abs().max() <= tol * max(1.0, u.abs().max().item())


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_full_size_head_conv3_argmax(dtype_name):
    """conv3 at batch 32 (512 -> 7605 on 24x24, the 192x128 head tile): materialised head on sampled pixels vs fp64,
    arg-max keys vs the materialised head everywhere, and keys-only == keys of the materialising launch."""
    dtype = _dt(dtype_name)
    B, H, W, Cout, uch, win = 32, 24, 24, 7605, 108, 441
    M = B * H * W
    x = q(rnd(B, 512, H, W, seed=91), dtype)
    w = q(rnd(Cout, 512, 1, 1, seed=92, scale=0.06), dtype)
    bias = rnd(Cout, seed=93, scale=0.1)
    info = {}
    raw, _ = run_conv(x, w, dtype, b1=bias, act1=3, nchw=True, argmax=(uch, win), info=info)
    print(info["kernel"])
    assert info["kernel"] == _kname(dtype_name, 192, 128)
    rows = _sampled_rows(M, n_random=32)
    ref = torch.sigmoid(_ref_rows(x, w, rows, H, W, 1, 1, 0) + bias.double())
    got = raw.permute(0, 2, 3, 1).reshape(M, Cout)[rows].double()
    assert float((got - ref).abs().max()) <= (2e-6 if dtype_name == "f32" else 5e-3)
    _check_keys(info, raw, uch, win)
    info2 = {}
    run_conv(x, w, dtype, b1=bias, act1=3, nchw=True, argmax=(uch, win), want_raw=False, info=info2)
    assert torch.equal(info2["keys"], info["keys"]) and torch.equal(info2["unary"], info["unary"])
    # the edge-aligned tile (what the fused-decode plans launch): same keys, stored instead of accumulated
    keys_e, kn = run_edge_argmax(x, w, bias, dtype, uch, win)
    print(kn)
    assert "head_limb_argmax_kernel" in kn and torch.equal(keys_e, info["keys"])


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
@pytest.mark.parametrize("mode", ["saturated", "near_ties", "equal_logits"])
def test_head_argmax_ambiguous_segments(force_tile, mode, dtype_name):
    """The arg-max epilogue decides a window from its LOGITS when the runner-up is clearly below the maximum and falls
    back to comparing the sigmoid values otherwise.  Force the fallback: (a) saturated heads -- many logits above 16.6
    all give sigmoid == 1.0f and the LOWEST index must win (np.argmax, /root/reference/datatest.py:113); (b) logits a few
    ulps apart whose sigmoids may round to one float; (c) exactly equal logits (duplicated weight rows).  In every
    case keys == first maximum of the materialised head of the same launch, and keys-only == keys."""
    dtype = _dt(dtype_name)
    force_tile(192, 128)
    uch, win, E = 108, 441, 3
    Cout = uch + E * win
    B, H, W = 2, 12, 16
    x = q(rnd(B, 512, H, W, seed=101), dtype)
    w = q(rnd(Cout, 512, 1, 1, seed=102, scale=0.06), dtype)
    bias = rnd(Cout, seed=103, scale=0.1)
    if mode == "saturated":
        bias = bias + 30.0                                    # every logit far above 16.6: sigmoid == 1.0f everywhere
        bias[uch + 5] -= 60.0                                 # ... except a few channels
        bias[uch + win + 100: uch + win + 140] -= 25.0
    elif mode == "near_ties":
        w = w.clone()
        w[uch + 7::50] = w[uch + 3]                           # same weights ...
        bias[uch + 7::50] = bias[uch + 3] + 3e-7              # ... and a bias a few ulps above: logits a few ulps apart
        bias[uch + 3] += 4.0                                  # make them the window's maxima
        bias[uch + 7::50] += 4.0
    else:
        w = w.clone()
        w[uch + 9], w[uch + 200], w[uch + 440] = w[uch + 2], w[uch + 2], w[uch + 2]
        for c in (uch + 9, uch + 200, uch + 440):
            bias[c] = bias[uch + 2]
        for c in (uch + 2, uch + 9, uch + 200, uch + 440):
            bias[c] += 5.0
    info = {}
    raw, _ = run_conv(x, w, dtype, b1=bias, act1=3, nchw=True, argmax=(uch, win), info=info)
    ref, _ = ref_conv(x, w, b1=bias, act1=3)
    assert float((raw - ref).abs().max()) <= (2e-6 if dtype_name == "f32" else 5e-3)
    _check_keys(info, raw, uch, win)
    limbs = raw.numpy()[:, uch:].reshape(B, E, win, H, W)
    ties = (limbs == limbs.max(axis=2, keepdims=True)).sum(axis=2)
    print(mode, dtype_name, "windows with a tied maximum:", int((ties > 1).sum()), "of", ties.size)
    if mode != "near_ties":
        assert int((ties > 1).sum()) > 0                      # the case really exercises the tie rule
    info2 = {}
    run_conv(x, w, dtype, b1=bias, act1=3, nchw=True, argmax=(uch, win), want_raw=False, info=info2)
    assert torch.equal(info2["keys"], info["keys"]) and torch.equal(info2["unary"], info["unary"])
    keys_e, _ = run_edge_argmax(x, w, bias, dtype, uch, win)   # the edge-aligned tile takes the same decisions
    assert torch.equal(keys_e, info["keys"])


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_pixel_ranges_with_different_tiles_equal_one_launch(dtype_name):
    """ppn_conv_desc.m_begin / m_count: a conv cut into pixel ranges, each launched with a DIFFERENT tile, gives the
    bits of the single launch (every output accumulates its GEMM depth in the same order whatever tile computes it);
    cuts fall inside an image row and off every tile multiple; a launch leaves the pixels outside its range alone."""
    dtype = _dt(dtype_name)
    B, Cin, H, W, Cout = 3, 128, 20, 23, 320                    # M = 1380
    x = q(rnd(B, Cin, H, W, seed=71), dtype)
    w = q(rnd(Cout, Cin, 3, 3, seed=72, scale=(2.0 / (Cin * 9)) ** 0.5), dtype)
    s1 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(73))
    b1 = rnd(Cout, seed=74, scale=0.3)
    res = q(rnd(B, Cout, H, W, seed=75), dtype)
    s2 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(76))
    b2 = rnd(Cout, seed=77, scale=0.3)
    kw = dict(stride=1, dil=2, pad=2, s1=s1, b1=b1, act1=1, residual=res, s2=s2, b2=b2, act2=1, want_act=True)
    whole_raw, whole_act = run_conv(x, w, dtype, **kw)
    info = {}
    M = B * H * W
    cuts = [(0, 512, (256, 256)), (512, 301, (128, 128)), (813, M - 813, (192, 256))]
    raw, act = run_conv(x, w, dtype, info=info, ranges=cuts, **kw)
    print(info["kernels"])
    assert info["kernels"] == [_kname(dtype_name, 256, 256), _kname(dtype_name, 128, 128), _kname(dtype_name, 192, 256)]
    assert torch.equal(raw, whole_raw) and torch.equal(act, whole_act)
    # a single range: everything outside it keeps the NaN fill, everything inside equals the whole launch
    raw1, _ = run_conv(x, w, dtype, ranges=[(512, 301, (128, 128))], **kw)
    flat1 = raw1.permute(0, 2, 3, 1).reshape(M, Cout)
    flatw = whole_raw.permute(0, 2, 3, 1).reshape(M, Cout)
    assert torch.isnan(flat1[:512]).all() and torch.isnan(flat1[813:]).all()
    assert torch.equal(flat1[512:813], flatw[512:813])


def test_conv_split_plan_and_automatic_two_segment_launch():
    """Tile policy 2 (opt-in; measured slower end to end, conv_big.hip::big_split_for): ppn_conv_split reports where the
    launcher cuts the pixel range (whole rounds of 256x256 tiles, then one round of a small tile), a whole-tensor
    ppn_conv2d_fused at such a size performs the two launches itself and equals the same conv forced onto a single
    192x256 launch; under the default policy nothing is cut.  128 -> 512 3x3 at 32 x 48 x 48 (M = 73 728)."""
    import ctypes as C
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()
    cut = C.c_int64(-1)
    L.check(lib.ppn_conv_split(L.PPN_BF16, 512, 512, 73728, C.byref(cut)), "ppn_conv_split")
    assert cut.value == 0
    dtype = L.PPN_BF16
    B, Cin, H, W, Cout = 32, 128, 48, 48, 512                      # 3x3: 18 K steps (1x1 layers take 128-channel tiles)
    x = q(rnd(B, Cin, H, W, seed=81), dtype)
    w = q(rnd(Cout, Cin, 3, 3, seed=82, scale=(2.0 / (9 * Cin)) ** 0.5), dtype)
    b1 = rnd(Cout, seed=83, scale=0.3)
    info = {}
    L.check(lib.ppn_set_conv_tile_policy(2), "ppn_set_conv_tile_policy")
    try:
        for cin, cout, m, want in [(512, 512, 73728, 65536), (256, 256, 73728, 65536), (128, 128, 294912, 0),
                                   (512, 512, 18432, 0), (512, 7605, 18432, 0), (512, 512, 4608, 0)]:
            L.check(lib.ppn_conv_split(L.PPN_BF16, cin, cout, m, C.byref(cut)), "ppn_conv_split")
            assert cut.value == want, (cin, cout, m, cut.value)
        auto, _ = run_conv(x, w, dtype, 1, 1, 1, b1=b1, act1=1, info=info)
    finally:
        L.check(lib.ppn_set_conv_tile_policy(0), "ppn_set_conv_tile_policy")
    print(info["kernels"])
    assert info["kernel"] == _kname("bf16", 256, 256)            # the first of the two launches
    single, _ = run_conv(x, w, dtype, 1, 1, 1, b1=b1, act1=1, ranges=[(0, 0, (192, 256))])
    assert not torch.isnan(auto).any() and torch.equal(auto, single)


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_pixel_ranges_small_channel_kernel(dtype_name):
    """The same range interface through the small-channel kernel (Cin below the K step: csrc/conv.hip), stride 2."""
    dtype = _dt(dtype_name)
    B, Cin, H, W, Cout = 3, 16, 38, 42, 64
    x = q(rnd(B, Cin, H, W, seed=95), dtype)
    w = q(rnd(Cout, Cin, 3, 3, seed=96, scale=(2.0 / (Cin * 9)) ** 0.5), dtype)
    s1 = 0.5 + torch.rand(Cout, generator=torch.Generator().manual_seed(97))
    b1 = rnd(Cout, seed=98, scale=0.3)
    kw = dict(stride=2, dil=1, pad=1, s1=s1, b1=b1, act1=1)
    info = {}
    whole, _ = run_conv(x, w, dtype, info=info, **kw)
    assert info["kernel"].startswith("conv_igemm_kernel<"), info["kernel"]
    M = B * 19 * 21
    parts, _ = run_conv(x, w, dtype, ranges=[(0, 130, None), (130, 700, None), (830, M - 830, None)], **kw)
    assert not torch.isnan(whole).any() and torch.equal(parts, whole)
    one, _ = run_conv(x, w, dtype, ranges=[(130, 700, None)], **kw)
    flat = one.permute(0, 2, 3, 1).reshape(M, Cout)
    assert torch.isnan(flat[:130]).all() and torch.isnan(flat[830:]).all()
    assert torch.equal(flat[130:830], whole.permute(0, 2, 3, 1).reshape(M, Cout)[130:830])


def test_pixel_ranges_nchw_head_need_aligned_cuts():
    """NCHW f32 head output (model.py:134-136 layout): the epilogue stores four consecutive pixels of a channel plane as
    one float4 when Ho*Wo % 4 == 0, so a pixel range must begin and end on multiples of 4 (ppn_conv2d_fused rejects any
    other cut: it would store misaligned and write into the neighbouring range); aligned cuts equal the single launch
    and leave the rest of the tensor untouched."""
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_BF16
    B, Cin, H, W, Cout = 3, 128, 12, 12, 200                    # HoWo = 144 (multiple of 4), M = 432
    x = q(rnd(B, Cin, H, W, seed=171), dtype)
    w = q(rnd(Cout, Cin, 1, 1, seed=172, scale=(2.0 / Cin) ** 0.5), dtype)
    bias = rnd(Cout, seed=173, scale=0.3)
    kw = dict(b1=bias, act1=3, nchw=True)
    whole, _ = run_conv(x, w, dtype, **kw)
    M = B * H * W
    parts, _ = run_conv(x, w, dtype, ranges=[(0, 100, (128, 128)), (100, 200, (192, 128)), (300, M - 300, (128, 128))], **kw)
    assert not torch.isnan(whole).any() and torch.equal(parts, whole)
    one, _ = run_conv(x, w, dtype, ranges=[(100, 200, (128, 128))], **kw)
    flat = one.permute(0, 2, 3, 1).reshape(M, Cout)
    assert torch.isnan(flat[:100]).all() and torch.isnan(flat[300:]).all()
    assert torch.equal(flat[100:300], whole.permute(0, 2, 3, 1).reshape(M, Cout)[100:300])
    for lo, n in ((2, 200), (100, 201), (101, 99)):             # misaligned begin / end
        with pytest.raises(RuntimeError, match="multiple of 4"):
            run_conv(x, w, dtype, ranges=[(lo, n, (128, 128))], **kw)
    # a grid whose planes are not a multiple of 4 pixels takes the scalar store path: any cut is fine there
    x2 = q(rnd(2, Cin, 7, 9, seed=174), dtype)
    whole2, _ = run_conv(x2, w, dtype, **kw)
    parts2, _ = run_conv(x2, w, dtype, ranges=[(0, 61, (128, 128)), (61, 126 - 61, (128, 128))], **kw)
    assert torch.equal(parts2, whole2)


@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(3, 24, 32), (2, 19, 37), (1, 8, 16), (2, 96, 96)], ids=lambda s: "%dx%dx%d" % s)
def test_conv64_filter_bank_kernel_equals_generic(dtype_name, shape):
    """csrc/conv64.hip (64 -> 64 3x3 stride 1, filter bank in registers, persistent 8x16 tiles with a resident input
    patch) against the generic implicit-GEMM kernel on the same launch: BIT-identical outputs (same K order, same
    epilogue arithmetic) in the three forms layer3 uses -- single output with BN + ReLU, residual + second output,
    second output only -- on image sizes that are and are not multiples of the tile; and close to fp64."""
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()
    dtype = _dt(dtype_name)
    B, H, W = shape
    x = q(rnd(B, 64, H, W, seed=201), dtype)
    w = q(rnd(64, 64, 3, 3, seed=202, scale=(2.0 / 576) ** 0.5), dtype)
    res = q(rnd(B, 64, H, W, seed=203), dtype)
    s1 = 0.5 + torch.rand(64, generator=torch.Generator().manual_seed(204))
    b1 = rnd(64, seed=205, scale=0.3)
    s2 = 0.5 + torch.rand(64, generator=torch.Generator().manual_seed(206))
    b2 = rnd(64, seed=207, scale=0.3)
    forms = [dict(s1=s1, b1=b1, act1=1), dict(residual=res, s2=s2, b2=b2, act2=1, want_act=True),
             dict(s1=s2, b1=b2, residual=res, act2=1, want_raw=False, want_act=True), dict()]
    for kw in forms:
        got, names = {}, {}
        for on in (1, 0):
            L.check(lib.ppn_set_conv64_enabled(on), "ppn_set_conv64_enabled")
            try:
                info = {}
                got[on] = run_conv(x, w, dtype, 1, 1, 1, info=info, **kw)
                names[on] = info["kernel"]
            finally:
                L.check(lib.ppn_set_conv64_enabled(1), "ppn_set_conv64_enabled")
        assert names[1].startswith("conv64_kernel<") and not names[0].startswith("conv64_kernel<"), names
        for a_, b_ in zip(got[1], got[0]):
            assert (a_ is None) == (b_ is None)
            if a_ is not None:
                assert not torch.isnan(a_).any() and torch.equal(a_, b_), (kw.keys(), float((a_ - b_).abs().max()))
    raw, _ = run_conv(x, w, dtype, 1, 1, 1, s1=s1, b1=b1, act1=1)
    ref, _ = ref_conv(x, w, 1, 1, 1, s1, b1, act1=1)
    assert float((raw - ref).abs().max()) <= _tol(dtype_name) * max(1.0, float(ref.abs().max()))
