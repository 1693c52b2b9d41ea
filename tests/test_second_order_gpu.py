"""GPU parity of the second-order building blocks (GradNorm's Lgrad.backward(), main.py:759) against PyTorch-CPU
double-backward / forward-mode autograd in f64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(t, a):
    return {"none": lambda v: v, "relu": F.relu, "lrelu": lambda v: F.leaky_relu(v, 0.1)}[a](t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,act", [((2, 6, 6, 128), "lrelu"), ((3, 5, 7, 512), "lrelu"), ((2, 9, 9, 64), "relu"),
                                        ((2, 4, 4, 16), "none")])
def test_bn_tangent_and_dual_backward(dtype, shape, act):
    from pytorch_pose_proposal_network_amd import train as T
    g = torch.Generator().manual_seed(5)
    c = shape[-1]
    x = (torch.randn(*shape, generator=g) * 1.3 + 0.4).to(dtype)
    xdot = torch.randn(*shape, generator=g).to(dtype)
    dy = torch.randn(*shape, generator=g).to(dtype)
    dyt = torch.randn(*shape, generator=g).to(dtype)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3

    # reference: the tangent written out as a differentiable function of x (torch.func.jvp of batch_norm keeps the
    # batch statistics out of the graph, so its x-gradient is incomplete), then plain autograd
    xr = x.double().reshape(-1, c).clone().requires_grad_(True)
    xdr = xdot.double().reshape(-1, c).clone().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mu, var = xr.mean(0), xr.var(0, unbiased=False)
    r = 1 / torch.sqrt(var + 1e-5)
    xh = (xr - mu) * r
    zb = xh * gr + br
    y = _act(zb, act)
    slope = {"none": torch.ones_like(zb), "relu": (zb > 0).double(), "lrelu": torch.where(zb > 0, 1.0, 0.1)}[act].detach()
    ydot = slope * gr * r * (xdr - xdr.mean(0) - xh * (xdr * xh).mean(0))
    S = (y * dy.double().reshape(-1, c)).sum() + (ydot * dyt.double().reshape(-1, c)).sum()
    gx, gxd, gg, gb = torch.autograd.grad(S, [xr, xdr, gr, br])
    gx, gxd, ydot = (t_.reshape(shape).permute(0, 3, 1, 2) for t_ in (gx, gxd, ydot))

    dev = torch.device("cuda")
    xd, xdd, gd, bd = x.to(dev), xdot.to(dev), gamma.to(dev), beta.to(dev)
    _, saved = T.bn_train_forward(xd, gd, bd, act=act)
    tan = T.bn_tangent(xd, xdd, gd, bd, saved, act)
    dx, dxdot, dgamma, dbeta = T.bn_dual_backward(xd, xdd, dy.to(dev), dyt.to(dev), gd, bd, saved, act)
    torch.cuda.synchronize()
    lo = dtype == torch.bfloat16
    tol = 3e-2 if lo else 3e-5

    def close(a, b, scale=1.0):
        b = b.detach()
        return (a.double().cpu() - b).abs().max() <= tol * scale * max(1.0, b.abs().max().item())

    assert close(tan, ydot.permute(0, 2, 3, 1))
    assert close(dxdot, gxd.permute(0, 2, 3, 1))
    assert close(dx, gx.permute(0, 2, 3, 1), 2.0)
    n = x.numel() // c
    assert close(dgamma, gg, max(1.0, n ** 0.5) / 4)
    assert close(dbeta, gb, max(1.0, n ** 0.5) / 4)


@pytest.mark.parametrize("coeff", [[1.0, 0, 0, 0, 0], [0, 1.0, 0, 0, 0], [0, 0, 1.0, 0, 0], [0, 0, 0, 1.0, 0],
                                   [0, 0, 0, 0, 1.0], [0.3, 0.2, 0.25, 0.15, 0.1]])
def test_loss_dual_matches_double_backward(coeff):
    """(zbar, tzbar) = d/d(z, tz) of  F = < d(sum c_i L_i)/ds , sig'(z) tz >  with s = sigmoid(z), by torch double backward."""
    from pytorch_pose_proposal_network_amd import loss, prng, config as cfg
    from oracle import loss_ref as Lr, targets_ref as Tg
    B = 2
    tg = Tg.synthetic_batch(50, B)
    Cn = cfg.lastsize()
    z0 = (prng.uniform(prng.stream_seed(9, 1), B * Cn * 576, -3.0, 3.0)).reshape(B, Cn, 24, 24).astype(np.float32)
    # overlapping predictions so that the IoU branch has curvature
    on = tg["delta"] > 0
    for lo, key, a, b in ((36, "tx", 0.9, 0.03), (54, "ty", 0.95, 0.02), (72, "tw", 1.2, 0.01), (90, "th", 0.8, 0.01)):
        v = np.clip(tg[key][on] * a + b, 1e-3, 1 - 1e-3)
        z0[:, lo:lo + 18][on] = np.log(v / (1 - v)).astype(np.float32)
    tz0 = prng.uniform(prng.stream_seed(9, 2), B * Cn * 576, -1.0, 1.0).reshape(B, Cn, 24, 24).astype(np.float32)
    unary_only = coeff[4] == 0
    z = torch.from_numpy(z0).double().requires_grad_(True)
    tz = torch.from_numpy(tz0).double().requires_grad_(True)
    s = torch.sigmoid(z)
    t64 = {k: torch.from_numpy(v).double() for k, v in tg.items()}
    losses = Lr.ppn_loss_ref(s, t64)
    total = sum(float(c) * l for c, l in zip(coeff, losses))
    gs, = torch.autograd.grad(total, s, create_graph=True)
    Fv = (gs * (s * (1 - s) * tz)).sum()
    zbar_ref, tzbar_ref = torch.autograd.grad(Fv, [z, tz])

    dev = torch.device("cuda")
    head = torch.sigmoid(torch.from_numpy(z0).to(dev))
    crit = loss.PPNLoss()
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    sl = slice(0, 108) if unary_only else slice(None)     # unary passes use compact [B,6K,H,W] dual tensors
    tzd = torch.from_numpy(np.ascontiguousarray(tz0[:, sl])).to(dev)
    zbar, tzbar = crit.dual(head, tzd, tgd, coeff, unary_only=unary_only)
    torch.cuda.synchronize()
    for mine, ref in ((zbar, zbar_ref), (tzbar, tzbar_ref)):
        m, r = mine.cpu().double(), ref[:, sl]
        assert (m - r).abs().max() <= 2e-4 * max(1e-3, r.abs().max().item()), ((m - r).abs().max(), r.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_limb_dual_nhwc_equals_dual_plus_relayout(dtype):
    """ppn_loss_limb_dual_nhwc (what the trainer's limb stream uses) against the pieces it fuses -- PPNLoss.dual with
    coefficients (0,0,0,0,c4), checked above against torch double backward, then train.nchw_to_nhwc: the same arithmetic,
    so the NHWC tensors must be EQUAL; the pixel-sum partials against an f64 sum of zbar."""
    from pytorch_pose_proposal_network_amd import loss, prng, train as T, config as cfg
    from oracle import targets_ref as Tg
    B, c4 = 3, -0.37
    tg = {k: torch.from_numpy(v).cuda() for k, v in Tg.synthetic_batch(51, B).items()}
    Cn = cfg.lastsize()
    head = torch.sigmoid(torch.from_numpy(prng.uniform(prng.stream_seed(10, 1), B * Cn * 576, -3.0, 3.0)
                                          .reshape(B, Cn, 24, 24).astype(np.float32)).cuda())
    tz = torch.from_numpy(prng.uniform(prng.stream_seed(10, 2), B * Cn * 576, -1.0, 1.0)
                          .reshape(B, Cn, 24, 24).astype(np.float32)).cuda()
    crit = loss.PPNLoss()
    zbar, tzbar = crit.dual(head, tz, tg, [0.0, 0.0, 0.0, 0.0, c4])
    zb_ref, tzb_ref = T.nchw_to_nhwc(zbar, dtype), T.nchw_to_nhwc(tzbar, dtype)
    zb, tzb, zsum = crit.limb_dual_nhwc(head, tz, tg, c4, dtype)
    torch.cuda.synchronize()
    assert zb.shape == zb_ref.shape and torch.equal(zb, zb_ref) and torch.equal(tzb, tzb_ref)
    assert float(zbar[:, :6 * cfg.K].abs().max()) == 0.0                 # the limb stream leaves the unary channels alone
    ref = zbar.double().sum((0, 2, 3)).cpu()
    got = zsum.double().sum(0)[:Cn].cpu()
    assert (got - ref).abs().max() <= 1e-5 * max(1e-6, float(ref.abs().max()))
    assert float(zsum[:, Cn:].abs().max()) == 0.0
