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
    n_blocks = sum("block64" in k for k in kernels[1])
    assert n_blocks == {"drn_d_22": 1, "drn_d_38": 2}[arch], kernels[1]
    assert len(kernels[1]) == len(kernels[0]) - n_blocks
    assert torch.equal(heads[0].view(torch.int32), heads[1].view(torch.int32))
