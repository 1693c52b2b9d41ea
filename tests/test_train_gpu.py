"""GPU parity of the training building blocks against plain PyTorch-CPU ops (what the reference's autograd /
torch.optim.Adam / nn.BatchNorm2d execute: main.py:278-279, 643, 664-777)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(t, a):
    return {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[a](t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,act", [((2, 24, 24, 16), "relu"), ((3, 12, 12, 512), "lrelu"),
                                        ((2, 7, 5, 64), "none"), ((1, 3, 3, 2048), "relu"),
                                        ((4, 96, 96, 32), "relu")])
def test_bn_train_forward_backward(dtype, shape, act):
    from pytorch_pose_proposal_network_amd import train as T
    g = torch.Generator().manual_seed(11)
    c = shape[-1]
    x = (torch.randn(*shape, generator=g) * 1.7 + 0.6).to(dtype)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    dy = torch.randn(*shape, generator=g).to(dtype)
    skip = torch.randn(*shape, generator=g).to(dtype)

    # reference: nn.BatchNorm2d semantics in f64 on the (already rounded) inputs
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    yr = _act(z, act)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    dx_ref = xr.grad.permute(0, 2, 3, 1) + skip.double()

    dev = torch.device("cuda")
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
    y, saved = T.bn_train_forward(xd, gd, bd, rm_d, rv_d, act=act)
    dx, dgamma, dbeta = T.bn_train_backward(xd, dy.to(dev), gd, bd, saved, act=act, dx_add=skip.to(dev))
    torch.cuda.synchronize()

    lo = dtype == torch.bfloat16
    n = x.numel() // c
    assert torch.allclose(saved.mean.cpu().double(), x.double().reshape(-1, c).mean(0), atol=2e-6, rtol=1e-6)
    assert torch.allclose(rm_d.cpu().double(), rm_r, atol=2e-6, rtol=1e-6)
    assert torch.allclose(rv_d.cpu().double(), rv_r, atol=1e-6, rtol=3e-6)
    tol = 2e-2 if lo else 2e-5
    yref = yr.detach().permute(0, 2, 3, 1)
    assert (y.cpu().double() - yref).abs().max() <= tol * max(1.0, yref.abs().max().item())
    # parameter gradients are f32 sums of n terms
    scale = max(1.0, n ** 0.5)
    assert (dgamma.cpu().double() - gr.grad).abs().max() <= (3e-2 if lo else 2e-5) * scale
    assert (dbeta.cpu().double() - br.grad).abs().max() <= (3e-2 if lo else 2e-5) * scale
    assert (dx.cpu().double() - dx_ref).abs().max() <= (6e-2 if lo else 5e-5) * max(1.0, dx_ref.abs().max().item())


def test_bn_rejects_bad_shapes():
    from pytorch_pose_proposal_network_amd import train as T, lib as L
    dev = torch.device("cuda")
    x = torch.zeros(2, 4, 4, 24, device=dev)           # 24 channels: not a power of two
    with pytest.raises(L.PPNError):
        T.bn_train_forward(x, torch.ones(24, device=dev), torch.zeros(24, device=dev))


@pytest.mark.parametrize("n", [5, 1000, 128 * 128 * 9 + 3, 4_000_037])
def test_flat_adam_matches_torch(n):
    from pytorch_pose_proposal_network_amd import train as T
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=7e-4)              # train.sh:5
    dev = torch.device("cuda")
    pd = p0.to(dev)
    lp = torch.empty(n, dtype=torch.bfloat16, device=dev)
    mine = T.FlatAdam(pd, lr=7e-4, param_lp=lp)
    for it in range(4):
        grad = torch.randn(n, generator=g) * (10.0 ** (it - 2))
        ref.grad = grad.clone() / 2                      # world_size 2: SUM all-reduce then /world
        opt.step()
        mine.step(grad.to(dev), grad_scale=0.5)
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), ref.detach(), rtol=2e-6, atol=2e-7)
    st = opt.state[ref]
    assert torch.allclose(mine.exp_avg.cpu(), st["exp_avg"], rtol=2e-6, atol=1e-9)
    assert torch.allclose(mine.exp_avg_sq.cpu(), st["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    assert torch.equal(lp.cpu(), pd.cpu().to(torch.bfloat16))


def test_sumsq():
    from pytorch_pose_proposal_network_amd import train as T
    x = torch.randn(128 * 128 * 9, generator=torch.Generator().manual_seed(3))
    out = T.sumsq(x.cuda()).cpu().item()
    assert abs(out - (x.double() ** 2).sum().item()) <= 1e-6 * out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gradnorm_weight_step_matches_autograd(seed):
    """main.py:668-777 restated with torch autograd on 5 scalars: l_i = w_i L_i, G_i = ||w_i g_i||, C_i detached,
    Lgrad = sum L1(G_i, C_i), Adam on w, clamp, renormalise."""
    from pytorch_pose_proposal_network_amd import train as T
    g = torch.Generator().manual_seed(seed)
    losses = torch.rand(5, generator=g) * 3 + 0.1
    base = torch.rand(5, generator=g) * 3 + 0.1
    gvec = [torch.randn(64, generator=g) * (0.2 + i) for i in range(5)]     # dL_i/dW stand-ins
    alpha, lr = 0.12, 0.025
    wm = torch.nn.Linear(5, 1, bias=False)
    wm.weight.data.fill_(1.0)
    if seed == 2:
        wm.weight.data = torch.tensor([[1.3, 0.4, 1.1, 0.9, 1.3]])
    opt = torch.optim.Adam(wm.parameters(), lr=lr)
    dev = torch.device("cuda")
    mine = T.GradNormWeights(dev, lr=lr, alpha=alpha)
    mine.w.copy_(wm.weight.data[0])
    for it in range(3):
        l = [wm.weight[0][i] * losses[i] for i in range(5)]
        G = [torch.norm(wm.weight[0][i] * gvec[i], 2) for i in range(5)]
        G_avg = sum(G) / 5
        lhat = [l[i] / base[i] for i in range(5)]
        lhat_avg = sum(lhat) / 5
        Cc = [(G_avg * (lhat[i] / lhat_avg) ** alpha).detach() for i in range(5)]
        opt.zero_grad()
        Lgrad = sum(F.l1_loss(G[i], Cc[i]) for i in range(5))
        Lgrad.backward()
        opt.step()
        with torch.no_grad():
            wm.weight.clamp_(min=0.0)
            wm.weight.div_(torch.mean(wm.weight))
        gn = torch.stack([torch.norm(v, 2) for v in gvec])
        log = mine.step(losses.to(dev), gn.to(dev), base.to(dev)).cpu()
        assert torch.allclose(log[0:5], torch.stack(G).detach(), rtol=1e-5)
        assert torch.allclose(log[5:10], torch.stack(Cc), rtol=1e-5)
        assert abs(log[15].item() - Lgrad.item()) <= 1e-5 * max(1.0, abs(Lgrad.item()))
        assert torch.allclose(mine.w.cpu(), wm.weight.data[0], rtol=2e-5, atol=1e-6)
        losses = losses * 0.9 + torch.rand(5, generator=g) * 0.1


CONV_CASES = [
    # (B, Cin, Cout, H, k, stride, dil, pad)
    (2, 64, 64, 12, 3, 1, 1, 1),
    (2, 128, 256, 9, 3, 1, 2, 2),         # dilated, ragged pixel count, two cout tiles
    (1, 512, 128, 6, 1, 1, 1, 0),         # 1x1 neck conv
    (2, 64, 128, 16, 3, 2, 1, 1),         # strided 3x3 (layer4.0.conv1)
    (2, 64, 128, 16, 1, 2, 1, 0),         # strided 1x1 projection shortcut
    (1, 160, 72, 7, 3, 1, 4, 4),          # channel counts that are not tile multiples; dilation > image/2
    (3, 32, 64, 20, 3, 2, 1, 1),          # layer3.0.conv1: 32 input channels
    (1, 256, 512, 10, 3, 1, 2, 2),        # >= 256-wide: the 256 x 256 tile (8 waves)
    (2, 512, 256, 6, 1, 1, 1, 0),
    (1, 264, 320, 7, 3, 1, 1, 1),         # 256-tile with ragged channel counts
    (2, 8, 16, 70, 7, 1, 1, 3),           # layer0 7x7 on the 8-channel padded input (bf16: dedicated stem kernel)
    (3, 4, 16, 77, 7, 1, 1, 3),           # ... on the 4-channel input the trainer keeps (bf16: two MFMAs per filter row)
    (2, 16, 16, 70, 3, 1, 1, 1),          # layer1
    (2, 16, 32, 141, 3, 2, 1, 1),         # layer2, stride 2, odd input size
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(dtype, case):
    from pytorch_pose_proposal_network_amd import train as T
    B, ci, co, H, k, s, dil, pad = case
    if ci == 4 and dtype == torch.float32:
        pytest.skip("the 4-channel layer-0 input exists for the dedicated bf16 kernel only")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, H, ci, generator=g).to(dtype)
    eff = dil * (k - 1) + 1
    Ho = (H + 2 * pad - eff) // s + 1
    dy = torch.randn(B, Ho, Ho, co, generator=g).to(dtype)
    w = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w, None, s, pad, dil)
    y.backward(dy.double().permute(0, 3, 1, 2))
    dev = torch.device("cuda")
    pre = torch.full((co, ci, k, k), 0.5, device=dev)
    dw = T.conv_wgrad(x.to(dev), dy.to(dev), k, s, dil, pad)
    dw2 = T.conv_wgrad(x.to(dev), dy.to(dev), k, s, dil, pad, out=pre, accumulate=True)
    torch.cuda.synchronize()
    # inputs are already rounded, products are exact in f32, only the f32 accumulation order differs
    tol = 3e-5 * (B * Ho * Ho) ** 0.5
    assert (dw.cpu().double() - w.grad).abs().max() <= tol
    assert (dw2.cpu().double() - 0.5 - w.grad).abs().max() <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES[:5] + [(1, 64, 64, 11, 3, 2, 1, 1)])
def test_conv_dgrad(dtype, case):
    from pytorch_pose_proposal_network_amd import train as T
    B, ci, co, H, k, s, dil, pad = case
    g = torch.Generator().manual_seed(6)
    eff = dil * (k - 1) + 1
    Ho = (H + 2 * pad - eff) // s + 1
    w = (torch.randn(co, ci, k, k, generator=g) * (ci * k * k) ** -0.5)
    if dtype == torch.bfloat16:
        w = w.to(dtype).float()
    dy = torch.randn(B, Ho, Ho, co, generator=g).to(dtype)
    skip = torch.randn(B, H, H, ci, generator=g).to(dtype)
    x = torch.zeros(B, ci, H, H, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, s, pad, dil).backward(dy.double().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1) + skip.double()
    dev = torch.device("cuda")
    dx = T.conv_dgrad(dy.to(dev), w.to(dev), (H, H), s, dil, pad, add=skip.to(dev))
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-5
    assert dx.shape == ref.shape
    assert (dx.cpu().double() - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())


FULL_WGRAD = [
    # (B, Cin, Cout, H, k, stride, dil, pad): D-22 layer shapes at 384x384, batch 32 -- the 8-wave ping-pong kernel
    (32, 512, 512, 48, 3, 1, 4, 4),
    (32, 256, 512, 48, 1, 1, 1, 0),
    (32, 512, 512, 24, 3, 1, 1, 1),
    (8, 512, 1312, 24, 1, 1, 1, 0),
    (3, 264, 320, 37, 3, 1, 2, 2),       # ragged everything: channel tiles, 4107 pixels (not a multiple of 32), odd width
    (32, 64, 64, 96, 3, 1, 1, 1),        # layer3: the 4-wave kernel with its split capped by the LDS offset table (74 splits)
    (32, 512, 512, 128, 1, 1, 1, 0),     # 524 k pixels: the 8-wave kernel's split cap binds too (65 splits instead of 64)
]


@pytest.mark.parametrize("case", FULL_WGRAD)
def test_conv_wgrad_fullsize_bf16_against_f32_kernel(case):
    """Full-size weight gradients: the bf16 kernel (ping-pong loop, LDS-DMA ring, transposed reads) against the exact-f32
    kernel (a different loop) on the same bf16-rounded operands -- the products are exact in f32, so only the
    accumulation order differs -- and against itself run twice (bitwise: no atomics, no races)."""
    from pytorch_pose_proposal_network_amd import train as T
    B, ci, co, H, k, s, dil, pad = case
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(9)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // s + 1
    x = torch.randn(B, H, H, ci, generator=g).to(torch.bfloat16).to(dev)
    dy = torch.randn(B, Ho, Ho, co, generator=g).to(torch.bfloat16).to(dev)
    a = T.conv_wgrad(x, dy, k, s, dil, pad).clone()
    b = T.conv_wgrad(x, dy, k, s, dil, pad).clone()
    r = T.conv_wgrad(x.float(), dy.float(), k, s, dil, pad)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    tol = 3e-5 * (B * Ho * Ho) ** 0.5
    assert float((a - r).abs().max()) <= tol, (float((a - r).abs().max()), tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 16, 32, 20, 3), (1, 64, 64, 11, 3), (2, 32, 64, 13, 1), (1, 64, 128, 12, 1),
                                  (3, 32, 64, 7, 3)])
def test_stride2_dgrad_by_parity_equals_zero_upsampled(dtype, case):
    """train._dgrad_stride2 (four stride-1 sub-convolutions of dy with 2 x 2 kernels, one per input-pixel parity; k = 1: one
    half-resolution 1 x 1) against the zero-upsampled transposed convolution it replaces and against autograd in f64;
    even and odd input sizes."""
    from pytorch_pose_proposal_network_amd import train as T
    B, ci, co, H, k = case
    g = torch.Generator().manual_seed(11)
    pad = k // 2
    Ho = (H + 2 * pad - k) // 2 + 1
    dy = torch.randn(B, Ho, Ho, co, generator=g).to(dtype)
    w = torch.randn(co, ci, k, k, generator=g) * 0.2
    x = torch.zeros(B, ci, H, H, dtype=torch.float64, requires_grad=True)
    wq = w.to(dtype).double() if dtype == torch.bfloat16 else w.double()
    F.conv2d(x, wq, None, 2, pad).backward(dy.double().permute(0, 3, 1, 2))
    dev = torch.device("cuda")
    got = T._dgrad_stride2(dy.to(dev), w.to(dev), H, H, None)
    old = T._S2_PARITY
    T._S2_PARITY = False
    try:
        ref = T.conv_dgrad(dy.to(dev), w.to(dev), (H, H), 2, 1, pad)
    finally:
        T._S2_PARITY = old
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    err = (got.cpu().double() - x.grad.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= tol * max(1.0, x.grad.abs().max().item()), err
    # with a skip-path gradient to add: conv_dgrad must give the fused (add in f32, ONE rounding) result whatever the
    # PPN_DGRAD_S2_PARITY switch says -- the parity form could only add after rounding its sub-convolutions
    add = torch.randn(B, H, H, ci, generator=g).to(dtype).to(dev)
    with_add = T.conv_dgrad(dy.to(dev), w.to(dev), (H, H), 2, 1, pad, add=add)
    T._S2_PARITY = False
    try:
        ref_add = T.conv_dgrad(dy.to(dev), w.to(dev), (H, H), 2, 1, pad, add=add)
    finally:
        T._S2_PARITY = old
    torch.cuda.synchronize()
    assert torch.equal(with_add, ref_add)
    exact = x.grad.permute(0, 2, 3, 1) + add.cpu().double()
    assert (with_add.cpu().double() - exact).abs().max().item() <= tol * max(1.0, exact.abs().max().item())


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_batched_weight_pack_equals_the_single_packs(dtype_name):
    """ppn_pack_table_run (every weight view of a training iteration in one launch) == ppn_pack_weight /
    ppn_pack_weight_dgrad per view, byte for byte: layers of every k_order the training step meets (16-channel stem
    layers stay in the reference layout, 64-multiples go channel-chunk-major), forward and input-gradient layouts,
    padded cout (the 7605-channel conv3)."""
    import ctypes as C
    from pytorch_pose_proposal_network_amd import lib as L
    dtype = L.PPN_F32 if dtype_name == "float32" else L.PPN_BF16
    lib = L.load()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    st = L.current_stream_ptr()
    tdt = torch.float32 if dtype == L.PPN_F32 else torch.bfloat16
    cases = [(16, 16, 3, 0), (32, 16, 3, 0), (64, 32, 3, 0), (64, 64, 3, 1), (128, 64, 1, 0), (512, 256, 3, 0), (256, 512, 3, 1),
             (7605, 512, 1, 0), (512, 7616, 1, 0), (512, 512, 3, 1)]
    items = (L.PackItem * len(cases))()
    ws, singles, outs = [], [], []
    for it, (cout, cin, k, tr) in zip(items, cases):
        # tr: the FORWARD weight is [cin, cout, k, k]; the packed matrix is the gradient convolution's [cout][cin taps]
        w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g).to(dev)
        kstep, _, korder, ktot, cpad = L.conv_tiling(dtype, cin, cout, k)
        odt = torch.float32 if korder == 2 else tdt
        single = torch.full((cpad, ktot), 7.0, dtype=odt, device=dev)
        fn = lib.ppn_pack_weight_dgrad if tr else lib.ppn_pack_weight
        L.check(fn(dtype, w.data_ptr(), cout, cin, k, cpad, ktot, korder, kstep, single.data_ptr(), st), "pack")
        out = torch.full((cpad, ktot), -3.0, dtype=odt, device=dev)
        (it.w, it.out, it.dtype, it.cout, it.cin, it.ksize, it.cout_pad, it.k_total, it.k_order, it.k_step, it.transposed,
         it.reserved_) = (w.data_ptr(), out.data_ptr(), dtype, cout, cin, k, cpad, ktot, korder, kstep, tr, 0)
        ws.append(w); singles.append(single); outs.append(out)
    host = torch.empty(len(cases) * L.PPN_PACK_ITEM_BYTES, dtype=torch.uint8)
    grid = C.c_int32(0)
    L.check(lib.ppn_pack_table_build(items, len(cases), host.data_ptr(), C.byref(grid)), "build")
    assert grid.value > 0          # (element chunks of 2048, or 8-row x 64-channel tiles for the MFMA layers' 16-bit packs)
    table = host.to(dev)
    L.check(lib.ppn_pack_table_run(table.data_ptr(), len(cases), grid.value, st), "run")
    torch.cuda.synchronize()
    for (cout, cin, k, tr), a, b in zip(cases, singles, outs):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (cout, cin, k, tr)
    # bad entries are refused on the host
    items[0].cout_pad = 1
    assert lib.ppn_pack_table_build(items, len(cases), host.data_ptr(), C.byref(grid)) != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_streams_equal_the_stream_by_stream_calls(dtype):
    """ppn_bn_train_bwd_streams / ppn_bn_act_mask_streams / ppn_bn_dual_bwd_streams (the stream is a grid dimension; the
    second-order tail's five tangent streams share x): every stream's dx, tangent, adjoints, dgamma and dbeta bitwise equal
    to the single-stream calls on its slices, including the [dy | dyt] form that runs both ordinary backward passes as 2n
    streams of one call."""
    from pytorch_pose_proposal_network_amd import train as T
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    n, B, H, C = 3, 4, 12, 128
    x = torch.randn(B, H, H, C, generator=g).to(dtype).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    _, saved = T.bn_train_forward(x, gamma, beta, act="lrelu")
    xdot, dy, dyt = [torch.randn(n * B, H, H, C, generator=g).to(dtype).to(dev) for _ in range(3)]
    sl = [slice(j * B, (j + 1) * B) for j in range(n)]
    # backward with a skip gradient
    add = torch.randn(n * B, H, H, C, generator=g).to(dtype).to(dev)
    dx_m, dg_m, db_m = T.bn_train_backward(x, dy, gamma, beta, saved, act="lrelu", dx_add=add, nstreams=n)
    for j in range(n):
        dx, dg, db = T.bn_train_backward(x, dy[sl[j]], gamma, beta, saved, act="lrelu", dx_add=add[sl[j]])
        assert torch.equal(dx, dx_m[sl[j]]) and torch.equal(dg, dg_m[j]) and torch.equal(db, db_m[j])
    # tangent
    t_m = T.bn_tangent(x, xdot, gamma, beta, saved, "lrelu", nstreams=n)
    for j in range(n):
        assert torch.equal(T.bn_tangent(x, xdot[sl[j]], gamma, beta, saved, "lrelu"), t_m[sl[j]])
    # dual adjoint, both forms
    a_m = T.bn_dual_backward(x, xdot, dy, dyt, gamma, beta, saved, "lrelu", nstreams=n)
    b_m = T.bn_dual_backward(x, xdot, None, None, gamma, beta, saved, "lrelu", nstreams=n, dy_dyt=torch.cat([dy, dyt]))
    for j in range(n):
        ref = T.bn_dual_backward(x, xdot[sl[j]], dy[sl[j]], dyt[sl[j]], gamma, beta, saved, "lrelu")
        for m in (a_m, b_m):
            assert torch.equal(ref[0], m[0][sl[j]]) and torch.equal(ref[1], m[1][sl[j]])
            assert torch.equal(ref[2], m[2][j]) and torch.equal(ref[3], m[3][j])
    torch.cuda.synchronize()


def test_bn_dual_backward_summed_equals_the_sum_of_the_streams():
    """bn_dual_backward_summed (the primal adjoints arrive already summed, the dual terms accumulate into the one result):
    in f32 the adjoint at x equals the sum over the streams of bn_dual_backward's per-stream results to rounding (the
    ordinary backward is linear in dy), the tangent adjoints are bitwise the per-stream ones, dgamma / dbeta the sums."""
    from pytorch_pose_proposal_network_amd import train as T
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(12)
    n, B, H, C = 5, 2, 10, 64
    x = torch.randn(B, H, H, C, generator=g).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    _, saved = T.bn_train_forward(x, gamma, beta, act="lrelu")
    xdot, dy, dyt = [torch.randn(n * B, H, H, C, generator=g).to(dev) for _ in range(3)]
    ref = T.bn_dual_backward(x, xdot, dy, dyt, gamma, beta, saved, "lrelu", nstreams=n)
    dysum = dy.view(n, B, H, H, C).sum(0)
    dx, dxdot, dgam, dbet = T.bn_dual_backward_summed(x, xdot, torch.cat([dysum, dyt]), gamma, beta, saved, "lrelu", n)
    torch.cuda.synchronize()
    want = ref[0].view(n, B, H, H, C).double().sum(0)
    assert float((dx.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert torch.equal(dxdot, ref[1])
    assert torch.allclose(dgam, ref[2].sum(0), rtol=1e-4, atol=1e-4) and torch.allclose(dbet, ref[3].sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_data_movement_kernels_equal_the_torch_copies(dtype):
    """ppn_upsample_zero, ppn_interleave_parity, ppn_image_to_nhwc against the torch indexing they replace (bitwise: pure
    data movement), odd sizes included."""
    import ctypes as C
    from pytorch_pose_proposal_network_amd import train as T, lib as L
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    code = L.PPN_F32 if dtype == torch.float32 else L.PPN_BF16
    for (B, h, w, ch, s, dh, dw) in [(2, 5, 7, 16, 2, 10, 13), (1, 6, 6, 64, 2, 11, 11), (3, 4, 9, 32, 1, 6, 10), (2, 7, 5, 8, 3, 19, 15)]:
        src = torch.randn(B, h, w, ch, generator=g).to(dtype).to(dev)
        want = torch.zeros(B, dh, dw, ch, dtype=dtype, device=dev)
        want[:, ::s, ::s][:, :h, :w] = src
        assert torch.equal(T.upsample_zero(src, s, dh, dw), want), (B, h, w, ch, s)
    for (B, H, W, ch) in [(2, 9, 12, 16), (1, 8, 8, 32), (2, 7, 7, 8)]:
        Ho1, Wo1 = (H + 2 - 3) // 2 + 2, (W + 2 - 3) // 2 + 2
        o = [[torch.randn(B, Ho1, Wo1, ch, generator=g).to(dtype).to(dev) for _ in range(2)] for _ in range(2)]
        want = torch.empty(B, H, W, ch, dtype=dtype, device=dev)
        for py in (0, 1):
            for px in (0, 1):
                ny, nx = (H - py + 1) // 2, (W - px + 1) // 2
                want[:, py::2, px::2] = o[py][px][:, py:py + ny, px:px + nx]
        dx = torch.empty_like(want)
        L.check(L.load().ppn_interleave_parity(code, o[0][0].data_ptr(), o[0][1].data_ptr(), o[1][0].data_ptr(), o[1][1].data_ptr(),
                                               B, H, W, ch, dx.data_ptr(), L.current_stream_ptr()), "interleave")
        assert torch.equal(dx, want), (B, H, W, ch)
    x = torch.randn(3, 3, 11, 14, generator=g).to(dev)
    for cp in (4, 8):
        want = torch.zeros(3, 11, 14, cp, dtype=dtype, device=dev)
        want[..., :3] = x.permute(0, 2, 3, 1)
        got = torch.empty_like(want)
        L.check(L.load().ppn_image_to_nhwc(code, x.data_ptr(), 3, 11, 14, cp, got.data_ptr(), L.current_stream_ptr()), "image")
        assert torch.equal(got, want)
    torch.cuda.synchronize()


def test_repack_all_is_scoped_to_the_given_storages():
    """train.repack_all(device, storages): one trainer's batched repack must not rewrite (and stamp as current) the packed
    weights of another registered storage, whose kernels may still be reading them on its own streams."""
    from pytorch_pose_proposal_network_amd import train as T
    dev = torch.device("cuda")
    fa = torch.randn(128 * 128 * 9, device=dev) * 0.05
    fb = torch.randn(128 * 128 * 9, device=dev) * 0.05
    ka, kb = T.register_param_storage(fa), T.register_param_storage(fb)
    try:
        x = torch.randn(1, 8, 8, 128, device=dev).to(torch.bfloat16)
        wa, wb = fa.view(128, 128, 3, 3), fb.view(128, 128, 3, 3)
        T.conv2d_nhwc(x, wa, 1, 1, 1)
        T.conv2d_nhwc(x, wb, 1, 1, 1)
        ents = {v[2]: k for k, v in T._pack_cache.items() if v[2] in (ka, kb)}
        assert set(ents) == {ka, kb}
        before_b = T._pack_cache[ents[kb]][1].clone()
        fa.mul_(2.0); fb.mul_(3.0)
        T.bump_param_version()
        n = T.repack_all(dev, [ka])
        assert n == 1
        ver = T._param_version[0]
        assert T._pack_cache[ents[ka]][0] == ver and T._pack_cache[ents[kb]][0] != ver
        assert torch.equal(T._pack_cache[ents[kb]][1], before_b)             # untouched
        ya = T.conv2d_nhwc(x, wa, 1, 1, 1)                                   # the refreshed copy
        yb = T.conv2d_nhwc(x, wb, 1, 1, 1)                                   # stale entry: repacked on use
        assert T._pack_cache[ents[kb]][0] == ver
        torch.testing.assert_close(ya.float(), T.conv2d_nhwc(x, wa.clone(), 1, 1, 1).float(), rtol=0, atol=0)
        torch.testing.assert_close(yb.float(), T.conv2d_nhwc(x, wb.clone(), 1, 1, 1).float(), rtol=0, atol=0)
    finally:
        T.unregister_param_storage(ka)
        T.unregister_param_storage(kb)
