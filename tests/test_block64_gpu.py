"""GPU parity of the one-launch 64-channel BasicBlock (csrc/block64.hip, ppn_basicblock64_fused):

* BIT-IDENTICAL to the two ppn_conv2d_fused launches it replaces (conv1 -> bn2 -> ReLU -> mid tensor; conv2 + residual +
  second output), in f16 and bf16, on full tiles, ragged images (H, W not multiples of the 8 x 16 tile, smaller than a
  tile), several tiles per workgroup (persistent loop, both pipeline buffers) and every output / residual combination;
* against an fp64 reference of the block (/root/reference/drn.py:42-57) within the 16-bit tolerance;
* in the model: PoseProposalNet(fuse_block=True) == fuse_block=False bit for bit (head and people), and the plan runs it.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(dt_name, B, H, W, seed):
    from pytorch_pose_proposal_network_amd import lib as L
    dt, tdt = {"f16": (L.PPN_F16, torch.float16), "bf16": (L.PPN_BF16, torch.bfloat16)}[dt_name]
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda")
    t = {}
    t["x_raw"] = torch.randn(B, H, W, 64, generator=g).to(tdt)
    t["x_act"] = torch.relu(torch.randn(B, H, W, 64, generator=g)).to(tdt)
    for n in ("w1", "w2"):
        t[n] = (torch.randn(64, 64, 3, 3, generator=g) * 0.06)
    for n in ("sm", "s2"):
        t[n] = torch.rand(64, generator=g) + 0.5
    for n in ("bm", "b2"):
        t[n] = torch.randn(64, generator=g) * 0.2
    t = {k: v.to(dev) for k, v in t.items()}
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    kstep, _, korder, ktot, cpad = L.conv_tiling(dt, 64, 64, 3)
    assert (ktot, cpad) == (576, 64)               # one 64-channel slab: k = tap * 64 + ci in either depth order
    for n in ("w1", "w2"):
        p = torch.empty(cpad, ktot, dtype=tdt, device=dev)
        L.check(lib.ppn_pack_weight(dt, t[n].contiguous().data_ptr(), 64, 64, 3, cpad, ktot, korder, kstep, p.data_ptr(), st))
        t[n + "p"] = p
    torch.cuda.synchronize()
    return L, lib, dt, tdt, t, st


def _two_launches(L, lib, dt, tdt, t, st, B, H, W, residual, raw, act, flags=0):
    dev = t["x_act"].device
    mid = torch.empty(B, H, W, 64, dtype=tdt, device=dev)
    o_raw = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    o_act = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    zero = torch.zeros(64, device=dev)

    def desc(src, w):
        d = L.ConvDesc()
        d.dtype, d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = dt, B, H, W, 64, H, W, 64
        d.ksize, d.stride, d.dilation, d.pad, d.k_total, d.cout_pad = 3, 1, 1, 1, 576, 64
        d.src, d.weight, d.zero_page, d.flags = src.data_ptr(), w.data_ptr(), zero.data_ptr(), flags
        return d
    d1 = desc(t["x_act"], t["w1p"])
    d1.scale1, d1.shift1, d1.act1, d1.out_raw = t["sm"].data_ptr(), t["bm"].data_ptr(), L.PPN_ACT_RELU, mid.data_ptr()
    L.check(lib.ppn_conv2d_fused(C.byref(d1), st))
    d2 = desc(mid, t["w2p"])
    if residual:
        d2.residual = t["x_raw"].data_ptr()
    if raw:
        d2.out_raw = o_raw.data_ptr()
    if act:
        d2.scale2, d2.shift2, d2.act2, d2.out_act = t["s2"].data_ptr(), t["b2"].data_ptr(), L.PPN_ACT_RELU, o_act.data_ptr()
    L.check(lib.ppn_conv2d_fused(C.byref(d2), st))
    torch.cuda.synchronize()
    return mid, o_raw, o_act


def _one_launch(L, lib, dt, tdt, t, st, B, H, W, residual, raw, act):
    dev = t["x_act"].device
    o_raw = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    o_act = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    d = L.BlockDesc()
    d.dtype, d.batch, d.h, d.w, d.channels = dt, B, H, W, 64
    d.src, d.residual = t["x_act"].data_ptr(), (t["x_raw"].data_ptr() if residual else None)
    d.weight1, d.scale_mid, d.shift_mid, d.act_mid = t["w1p"].data_ptr(), t["sm"].data_ptr(), t["bm"].data_ptr(), L.PPN_ACT_RELU
    d.weight2 = t["w2p"].data_ptr()
    if raw:
        d.out_raw = o_raw.data_ptr()
    if act:
        d.scale2, d.shift2, d.act2, d.out_act = t["s2"].data_ptr(), t["b2"].data_ptr(), L.PPN_ACT_RELU, o_act.data_ptr()
    L.check(lib.ppn_basicblock64_fused(C.byref(d), st))
    torch.cuda.synchronize()
    return o_raw, o_act


SHAPES = [(2, 96, 96), (1, 8, 16), (3, 5, 7), (2, 50, 37), (1, 17, 33), (40, 24, 40)]


@pytest.mark.parametrize("dt_name", ["f16", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=["%dx%dx%d" % s for s in SHAPES])
def test_block_equals_its_two_launches(dt_name, shape):
    B, H, W = shape
    L, lib, dt, tdt, t, st = _setup(dt_name, B, H, W, 100 + H)
    for residual, raw, act in ((True, True, True), (True, True, False), (False, False, True), (False, True, False)):
        # the generic kernel is the reference (PPN_CONV_NO_FILTER_BANK); conv64.hip is bit-identical to it by its own tests
        _, r_raw, r_act = _two_launches(L, lib, dt, tdt, t, st, B, H, W, residual, raw, act, flags=L.PPN_CONV_NO_FILTER_BANK)
        o_raw, o_act = _one_launch(L, lib, dt, tdt, t, st, B, H, W, residual, raw, act)
        for name, a, b in (("out_raw", o_raw, r_raw), ("out_act", o_act, r_act)):
            a, b = a.view(torch.int16).cpu().numpy(), b.view(torch.int16).cpu().numpy()
            bad = np.argwhere(a != b)
            assert bad.size == 0, (dt_name, shape, residual, raw, act, name, len(bad), bad[:5])


@pytest.mark.parametrize("dt_name", ["f16", "bf16"])
def test_block_vs_fp64(dt_name):
    B, H, W = 2, 40, 52
    L, lib, dt, tdt, t, st = _setup(dt_name, B, H, W, 5)
    o_raw, o_act = _one_launch(L, lib, dt, tdt, t, st, B, H, W, True, True, True)

    def q(v):                                         # values as the kernels see them
        return v.to(tdt).double().cpu()
    x = q(t["x_act"]).permute(0, 3, 1, 2)
    w1, w2 = q(t["w1"]), q(t["w2"])
    mid = F.conv2d(x, w1, padding=1) * t["sm"].double().cpu().view(1, -1, 1, 1) + t["bm"].double().cpu().view(1, -1, 1, 1)
    mid = q(torch.relu(mid))
    v = F.conv2d(mid, w2, padding=1) + q(t["x_raw"]).permute(0, 3, 1, 2)
    u = torch.relu(v * t["s2"].double().cpu().view(1, -1, 1, 1) + t["b2"].double().cpu().view(1, -1, 1, 1))
    tol = 2e-2 if dt_name == "bf16" else 3e-3
    for got, ref in ((o_raw, v), (o_act, u)):
        err = (got.double().cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err <= tol * ref.abs().max().item(), (dt_name, err, ref.abs().max().item())


def test_block_rejects_bad_arguments():
    from pytorch_pose_proposal_network_amd import lib as L
    lib = L.load()
    d = L.BlockDesc()
    assert lib.ppn_basicblock64_fused(C.byref(d), None) != 0
    d.dtype, d.batch, d.h, d.w, d.channels = L.PPN_F32, 1, 8, 8, 64
    assert lib.ppn_basicblock64_fused(C.byref(d), None) != 0 and b"16-bit" in lib.ppn_last_error()
    d.dtype, d.channels = L.PPN_F16, 128
    assert lib.ppn_basicblock64_fused(C.byref(d), None) != 0 and b"64 channels" in lib.ppn_last_error()


@pytest.mark.parametrize("mode", ["bfloat16", "float16"])
@pytest.mark.parametrize("arch", ["drn_d_22", "drn_d_38"])
def test_model_with_fused_blocks_is_bit_identical(mode, arch):
    """The plan runs layer3's stride-1 blocks as one launch each and nothing changes: head bits equal."""
    from pytorch_pose_proposal_network_amd import drn, model as M, prng, synth
    S, B = 96, 3
    sd = synth.make_state_dict(arch, 0)
    frames = torch.from_numpy(prng.u8_frames(11, B, (S, S))).cuda()
    heads, kernels = [], []
    for fb in (False, True):
        net = M.PoseProposalNet(getattr(drn, arch)(), insize=(S, S), outsize=(S // 16, S // 16), compute_dtype=mode,
                                fuse_block=fb).cuda()
        net.load_state_dict(sd)
        heads.append(net.forward_u8(frames).clone())
        kernels.append([k for _, k, _, _ in net.profile_layers(frames, True)])
    assert not any("block64" in k for k in kernels[0])
    n_blocks = sum("block64_kernel" in k for k in kernels[1])
    n_first = sum("block64s2" in k for k in kernels[1])
    assert n_blocks == {"drn_d_22": 1, "drn_d_38": 2}[arch] and n_first == 1, kernels[1]
    assert len(kernels[1]) == len(kernels[0]) - n_blocks - 2 * n_first
    assert torch.equal(heads[0].view(torch.int32), heads[1].view(torch.int32))


@pytest.mark.parametrize("mode", ["bfloat16", "float16"])
@pytest.mark.parametrize("size", [(96, 96), (112, 208), (104, 72)])
def test_plan_shortcuts_leave_the_head_bits_unchanged(mode, size, monkeypatch):
    """Round 5's plan-level shortcuts are value-neutral: the weight-prefetch hint (PPN_PREFETCH), the subsampled raw stem output
    for the first block's 1x1 stride-2 projection (PPN_STEM_RAW_S2, incl. odd half-resolution sizes) and the one-launch
    BasicBlock (PPN_BLOCK64) each switched off give the same head, bit for bit (/root/reference/drn.py:42-57,176-181)."""
    from pytorch_pose_proposal_network_amd import drn, model as M, prng, synth
    H, W = size
    sd = synth.make_state_dict("drn_d_22", 0)
    frames = torch.from_numpy(prng.u8_frames(3, 2, (H, W))).cuda()

    def head(env):
        for k in ("PPN_PREFETCH", "PPN_STEM_RAW_S2", "PPN_BLOCK64"):
            monkeypatch.setenv(k, env.get(k, "1"))
        net = M.PoseProposalNet(drn.drn_d_22(), insize=(W, H), outsize=(W // 16, H // 16), compute_dtype=mode).cuda()
        net.load_state_dict(sd)
        return net.forward_u8(frames).clone()
    ref = head({})
    for off in ("PPN_PREFETCH", "PPN_STEM_RAW_S2", "PPN_BLOCK64"):
        got = head({off: "0"})
        assert torch.equal(ref.view(torch.int32), got.view(torch.int32)), off


# ---- the FIRST block of layer3 as one launch: 3x3 stride-2 conv1 from 32 channels + 1x1 stride-2 projection shortcut -------------------
def _setup_s2(dt_name, B, Hi, Wi, seed):
    from pytorch_pose_proposal_network_amd import lib as L
    dt, tdt = {"f16": (L.PPN_F16, torch.float16), "bf16": (L.PPN_BF16, torch.bfloat16)}[dt_name]
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda")
    t = {"x_raw": torch.randn(B, Hi, Wi, 32, generator=g).to(tdt), "x_act": torch.relu(torch.randn(B, Hi, Wi, 32, generator=g)).to(tdt),
         "w1": torch.randn(64, 32, 3, 3, generator=g) * 0.08, "wd": torch.randn(64, 32, 1, 1, generator=g) * 0.2,
         "w2": torch.randn(64, 64, 3, 3, generator=g) * 0.06}
    for n in ("sm", "s2", "sd"):
        t[n] = torch.rand(64, generator=g) + 0.5
    for n in ("bm", "b2", "bd"):
        t[n] = torch.randn(64, generator=g) * 0.2
    t = {k: v.to(dev) for k, v in t.items()}
    t["x_raw_s2"] = t["x_raw"][:, ::2, ::2].contiguous()
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    geom = {}
    for n, cin, k in (("w1", 32, 3), ("wd", 32, 1), ("w2", 64, 3)):
        kstep, _, korder, ktot, cpad = L.conv_tiling(dt, cin, 64, k)
        p = torch.empty(cpad, ktot, dtype=tdt, device=dev)
        L.check(lib.ppn_pack_weight(dt, t[n].contiguous().data_ptr(), 64, cin, k, cpad, ktot, korder, kstep, p.data_ptr(), st))
        t[n + "p"], geom[n] = p, (ktot, cpad, korder)
    torch.cuda.synchronize()
    assert geom["w1"][2] == 0 and geom["wd"][2] == 0 and geom["w2"][:2] == (576, 64)
    return L, lib, dt, tdt, t, st, geom


def _three_launches(L, lib, dt, tdt, t, st, geom, B, Hi, Wi, raw, act):
    dev = t["x_act"].device
    H, W = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    zero = torch.zeros(64, device=dev)
    ds, mid = (torch.empty(B, H, W, 64, dtype=tdt, device=dev) for _ in range(2))
    o_raw = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    o_act = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)

    def desc(src, w, wn, ih, iw, cin, k, s, pad):
        d = L.ConvDesc()
        d.dtype, d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = dt, B, ih, iw, cin, H, W, 64
        d.ksize, d.stride, d.dilation, d.pad, d.k_total, d.cout_pad = k, s, 1, pad, geom[wn][0], geom[wn][1]
        d.src, d.weight, d.zero_page, d.flags = src.data_ptr(), w.data_ptr(), zero.data_ptr(), L.PPN_CONV_NO_FILTER_BANK
        return d
    d0 = desc(t["x_raw"], t["wdp"], "wd", Hi, Wi, 32, 1, 2, 0)
    d0.scale1, d0.shift1, d0.out_raw = t["sd"].data_ptr(), t["bd"].data_ptr(), ds.data_ptr()
    L.check(lib.ppn_conv2d_fused(C.byref(d0), st))
    d1 = desc(t["x_act"], t["w1p"], "w1", Hi, Wi, 32, 3, 2, 1)
    d1.scale1, d1.shift1, d1.act1, d1.out_raw = t["sm"].data_ptr(), t["bm"].data_ptr(), L.PPN_ACT_RELU, mid.data_ptr()
    L.check(lib.ppn_conv2d_fused(C.byref(d1), st))
    d2 = desc(mid, t["w2p"], "w2", H, W, 64, 3, 1, 1)
    d2.residual = ds.data_ptr()
    if raw:
        d2.out_raw = o_raw.data_ptr()
    if act:
        d2.scale2, d2.shift2, d2.act2, d2.out_act = t["s2"].data_ptr(), t["b2"].data_ptr(), L.PPN_ACT_RELU, o_act.data_ptr()
    L.check(lib.ppn_conv2d_fused(C.byref(d2), st))
    torch.cuda.synchronize()
    return o_raw, o_act


def _one_launch_s2(L, lib, dt, tdt, t, st, geom, B, Hi, Wi, raw, act):
    dev = t["x_act"].device
    H, W = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    o_raw = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    o_act = torch.full((B, H, W, 64), 7.0, dtype=tdt, device=dev)
    d = L.BlockDesc()
    d.dtype, d.batch, d.h, d.w, d.channels, d.stride, d.in_h, d.in_w = dt, B, H, W, 64, 2, Hi, Wi
    d.src, d.proj_src = t["x_act"].data_ptr(), t["x_raw_s2"].data_ptr()
    d.weight1, d.w1_ld, d.scale_mid, d.shift_mid, d.act_mid = t["w1p"].data_ptr(), geom["w1"][0], t["sm"].data_ptr(), t["bm"].data_ptr(), L.PPN_ACT_RELU
    d.proj_weight, d.proj_ld, d.proj_scale, d.proj_shift = t["wdp"].data_ptr(), geom["wd"][0], t["sd"].data_ptr(), t["bd"].data_ptr()
    d.weight2 = t["w2p"].data_ptr()
    if raw:
        d.out_raw = o_raw.data_ptr()
    if act:
        d.scale2, d.shift2, d.act2, d.out_act = t["s2"].data_ptr(), t["b2"].data_ptr(), L.PPN_ACT_RELU, o_act.data_ptr()
    L.check(lib.ppn_basicblock64_fused(C.byref(d), st))
    torch.cuda.synchronize()
    return o_raw, o_act


S2_SHAPES = [(2, 192, 192), (1, 8, 32), (3, 10, 14), (2, 100, 74), (1, 33, 65), (24, 48, 80)]


@pytest.mark.parametrize("dt_name", ["f16", "bf16"])
@pytest.mark.parametrize("shape", S2_SHAPES, ids=["%dx%dx%d" % s for s in S2_SHAPES])
def test_first_block_equals_its_three_launches(dt_name, shape):
    """ppn_basicblock64_fused with stride 2 == downsample (1x1 s2 + BN) + conv1 (3x3 s2 + bn2 + ReLU) + conv2 (+ residual, second
    output), bit for bit, incl. odd input sizes, ragged tiles, several tiles per workgroup, repeated runs (determinism)."""
    B, Hi, Wi = shape
    L, lib, dt, tdt, t, st, geom = _setup_s2(dt_name, B, Hi, Wi, 200 + Hi)
    for raw, act in ((True, True), (True, False), (False, True)):
        r_raw, r_act = _three_launches(L, lib, dt, tdt, t, st, geom, B, Hi, Wi, raw, act)
        for rep in range(2):
            o_raw, o_act = _one_launch_s2(L, lib, dt, tdt, t, st, geom, B, Hi, Wi, raw, act)
            for name, a, b in (("out_raw", o_raw, r_raw), ("out_act", o_act, r_act)):
                a, b = a.view(torch.int16).cpu().numpy(), b.view(torch.int16).cpu().numpy()
                bad = np.argwhere(a != b)
                assert bad.size == 0, (dt_name, shape, raw, act, rep, name, len(bad), bad[:5])
