"""GPU parity of the fused PPN loss (forward values + gradient w.r.t. the head) against the golden values
produced by the reference's own PPNLoss + autograd (tests/golden/loss_cases.npz) and against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref as Lr, targets_ref as T
from pytorch_pose_proposal_network_amd import config as cfg, prng

pytestmark = pytest.mark.gpu


def _case(g, tag):
    seed, batch = int(g[f"{tag}/seed"]), int(g[f"{tag}/batch"])
    tg = T.synthetic_batch(seed, batch)
    head = prng.uniform(prng.stream_seed(seed, 7), batch * cfg.lastsize() * 576, 0.02, 0.98).reshape(
        batch, cfg.lastsize(), 24, 24)
    on = tg["delta"] > 0
    for lo, key, a, b in ((36, "tx", 0.9, 0.03), (54, "ty", 0.95, 0.02), (72, "tw", 1.2, 0.01), (90, "th", 0.8, 0.01)):
        head[:, lo:lo + 18][on] = (tg[key][on] * a + b).astype(np.float32)
    return head, tg


@pytest.mark.parametrize("tag", ["a", "b", "limb_only", "iou_only"])
def test_loss_golden(golden_dir, tag):
    from pytorch_pose_proposal_network_amd import loss
    g = np.load(os.path.join(golden_dir, "loss_cases.npz"))
    head, tg = _case(g, tag)
    crit = loss.PPNLoss()
    fm = torch.from_numpy(head).cuda()
    tt = {k: torch.from_numpy(v).cuda() for k, v in tg.items()}
    coeff = g[f"{tag}/coeff"]
    losses, grad = crit.forward_backward(fm, tt, coeff)
    losses = losses.cpu().numpy()
    assert np.allclose(losses, g[f"{tag}/losses"], rtol=2e-5), (losses, g[f"{tag}/losses"])
    gh = grad.cpu().numpy()
    ref = g[f"{tag}/grad_val"]
    got = gh.reshape(-1)[g[f"{tag}/grad_idx"]]
    scale = max(1e-6, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= 2e-5 * scale, np.abs(got - ref).max()
    assert np.allclose(np.abs(gh.astype(np.float64)).sum(axis=(2, 3)), g[f"{tag}/grad_abs_sum"], rtol=1e-4, atol=1e-6)
    # reference-shaped forward() returns the same five scalars and is bitwise reproducible
    five = crit(None, fm, tt["delta"], tt["weight"], tt["weight_ij"], tt["tx_half"], tt["ty_half"], tt["tx"], tt["ty"],
                tt["tw"], tt["th"], tt["te"])
    assert [float(v) for v in five] == [float(v) for v in losses]


def test_loss_matches_oracle_full_gradient():
    """Every element of the gradient (not a sample) against torch autograd on the CPU oracle."""
    from pytorch_pose_proposal_network_amd import loss
    tg = T.synthetic_batch(123, 2)
    head = prng.uniform(prng.stream_seed(5, 1), 2 * cfg.lastsize() * 576, 0.01, 0.99).reshape(2, cfg.lastsize(), 24, 24)
    coeff = [0.25, 0.15, 0.3, 0.2, 0.1]
    ref_l, ref_g = Lr.loss_and_grad_ref(head, tg, coeff)
    losses, grad = loss.PPNLoss().forward_backward(torch.from_numpy(head).cuda(),
                                                   {k: torch.from_numpy(v).cuda() for k, v in tg.items()}, coeff)
    assert np.allclose(losses.cpu().numpy(), ref_l, rtol=2e-5)
    d = np.abs(grad.cpu().numpy() - ref_g)
    assert d.max() <= 2e-5 * max(1.0, float(np.abs(ref_g).max())), d.max()


def test_loss_device_coefficients_equal_host_coefficients():
    """ppn_loss_fwd_bwd_dev (coefficients w_i / 5 read on the device, what PPNTrainer uses so that the host never waits
    for the task weights) against ppn_loss_fwd_bwd with the same values passed by value: same losses, same gradient."""
    from pytorch_pose_proposal_network_amd import loss
    tg = {k: torch.from_numpy(v).cuda() for k, v in T.synthetic_batch(77, 3).items()}
    head = torch.from_numpy(prng.uniform(prng.stream_seed(6, 2), 3 * cfg.lastsize() * 576, 0.01, 0.99)
                            .reshape(3, cfg.lastsize(), 24, 24)).cuda()
    w = torch.tensor([1.3, 0.8, 1.1, 0.7, 1.1], dtype=torch.float32).cuda()
    crit = loss.PPNLoss()
    l0, g0 = crit.forward_backward(head, tg, coeff=[float(np.float32(v) / np.float32(5.0)) for v in w.tolist()])
    l1, g1 = crit.forward_backward(head, tg, coeff_dev=(w, 5.0))
    assert torch.equal(l0, l1)
    assert torch.equal(g0, g1)
    with pytest.raises(ValueError):
        crit.forward_backward(head, tg, coeff_dev=(w.double(), 5.0))


def test_loss_argument_validation():
    from pytorch_pose_proposal_network_amd import loss
    crit = loss.PPNLoss()
    fm = torch.zeros(1, cfg.lastsize(), 24, 24, device="cuda")
    with pytest.raises((ValueError, KeyError)):
        crit.forward_backward(fm, {}, [1] * 5)


@pytest.mark.parametrize("size,batch,seed", [(384, 3, 11), (96, 4, 50), (384, 2, 1234)])
def test_target_encoder_bit_exact(size, batch, seed):
    """csrc/encode.hip vs the NumPy restatement of dataset.py:96-185: every one of the ten tensors identical."""
    from pytorch_pose_proposal_network_amd import targets, synth
    outsize = (size // 16, size // 16)
    lists = [synth.synthetic_people(seed + i, insize=(size, size)) for i in range(batch)]
    # overlapping people exercise the "later person overwrites" rule: duplicate the first person with another size
    twin = dict(lists[0][0])
    twin["size"] = np.float32(19.5)
    lists[0] = list(lists[0]) + [twin]
    ref = [T.encode_targets(p, insize=(size, size), outsize=outsize) for p in lists]
    got = targets.encode_targets(targets.pack_people(lists), (size, size), outsize)
    for k in targets.TARGET_KEYS:
        exp = np.stack([r[k] for r in ref])
        assert np.array_equal(got[k].cpu().numpy(), exp), k
    assert float(got["te"].sum()) > 0 and float(got["delta"].sum()) > 0


def test_target_encoder_equals_reference_fixture(golden_dir):
    """csrc/encode.hip vs the tensors the reference's own KeypointsDataset.__getitem__ produced (targets_cases.npz,
    dataset.py:96-200): all cases in one batch (ragged people counts), bit for bit."""
    from pytorch_pose_proposal_network_amd import targets
    g = np.load(os.path.join(golden_dir, "targets_cases.npz"))
    n = int(g["n_cases"])
    lists = [[dict(bbox=tuple(g[f"case{i}/bbox"][p]), points=g[f"case{i}/points"][p], visible=g[f"case{i}/visible"][p],
                   size=g[f"case{i}/size"][p]) for p in range(len(g[f"case{i}/size"]))] for i in range(n)]
    got = targets.encode_targets(targets.pack_people(lists))
    for k in targets.TARGET_KEYS:
        exp = np.stack([g[f"case{i}/{k}"] for i in range(n)])
        assert np.array_equal(got[k].cpu().numpy(), exp), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_loss_fwd_bwd_dz_equals_loss_then_head_grad(dtype):
    """ppn_loss_fwd_bwd_dz (losses + gradient w.r.t. conv3's logits as NHWC + bias partial sums, one pass; what the
    trainer uses) against the two steps it fuses: ppn_loss_fwd_bwd_dev and ppn_head_grad.  Same arithmetic per element, so
    dz must be EQUAL; loss_limb and the bias gradient are summed in another order (1e-6 / 1e-5 relative)."""
    import ctypes as C
    from pytorch_pose_proposal_network_amd import loss, lib as L
    B = 3
    tg = {k: torch.from_numpy(v).cuda() for k, v in T.synthetic_batch(78, B).items()}
    Cn = cfg.lastsize()
    head = torch.from_numpy(prng.uniform(prng.stream_seed(6, 3), B * Cn * 576, 0.01, 0.99).reshape(B, Cn, 24, 24)).cuda()
    w = torch.tensor([1.3, 0.8, 1.1, 0.7, 1.1], dtype=torch.float32).cuda()
    crit = loss.PPNLoss()
    l0, g0 = crit.forward_backward(head, tg, coeff_dev=(w, 5.0))
    cpad = (Cn + 63) // 64 * 64
    dz0 = torch.empty(B, 24, 24, cpad, dtype=dtype, device="cuda")
    db0 = torch.empty(Cn, dtype=torch.float32, device="cuda")
    code = L.PPN_F32 if dtype == torch.float32 else L.PPN_BF16
    L.check(L.load().ppn_head_grad(code, head.data_ptr(), g0.data_ptr(), B, Cn, 576, Cn, cpad, dz0.data_ptr(),
                                   db0.data_ptr(), L.current_stream_ptr()), "ppn_head_grad")
    l1, dz1, dbsum = crit.forward_backward_dz(head, tg, (w, 5.0), dtype)
    torch.cuda.synchronize()
    assert torch.equal(dz0, dz1)
    assert torch.equal(l0[:4], l1[:4])
    assert abs(float(l0[4]) - float(l1[4])) <= 1e-6 * abs(float(l0[4]))
    db1 = dbsum.sum(0)
    assert float(db1[Cn:].abs().max()) == 0.0
    assert float((db1[:Cn] - db0).abs().max()) <= 1e-5 * float(db0.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_compact_limb_targets_give_the_f32_targets_results(dtype):
    """targets["limb_c"] (te | weight_ij in two bits per element, written by ppn_encode_targets_c) decodes to exactly the f32
    tensors the encoder writes beside it, and the two kernels that read it -- ppn_loss_fwd_bwd_dz_c, ppn_loss_limb_dual_nhwc_c
    -- return bitwise what their f32-target forms return."""
    from pytorch_pose_proposal_network_amd import targets, loss as LS, config as cfg
    dev = torch.device("cuda")
    size, B = 192, 3
    tg = targets.synthetic_targets(77, B, (size, size), device=dev)
    lc = tg["limb_c"]
    assert lc.dtype == torch.uint8 and lc.shape == tg["te"].shape
    assert torch.equal((lc & 1).float(), tg["te"]) and int(tg["te"].sum()) > 0
    assert torch.equal(torch.where((lc & 2) != 0, 1.0, 0.0005).float(), tg["weight_ij"]) and float(tg["weight_ij"].max()) == 1.0
    crit = LS.PPNLoss(insize=(size, size), outsize=(size // 16, size // 16))
    g = torch.Generator().manual_seed(5)
    Ch = 6 * cfg.K + cfg.E * 21 * 21
    head = torch.sigmoid(torch.randn(B, Ch, size // 16, size // 16, generator=g)).to(dev)
    tz = torch.randn(B, Ch, size // 16, size // 16, generator=g).to(dev)
    w = torch.tensor([1.1, 0.9, 1.0, 0.8, 1.2], device=dev)
    plain = {k: v for k, v in tg.items() if k != "limb_c"}
    a = crit.forward_backward_dz(head, tg, (w, 5.0), dtype)
    b = crit.forward_backward_dz(head, plain, (w, 5.0), dtype)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    a = crit.limb_dual_nhwc(head, tz, tg, 0.37, dtype)
    b = crit.limb_dual_nhwc(head, tz, plain, 0.37, dtype)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    torch.cuda.synchronize()
