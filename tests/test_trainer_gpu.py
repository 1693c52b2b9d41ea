"""End-to-end GPU parity of the training step (PPNTrainer: train-mode forward, loss, backward, GradNorm task
weights, Adam) against the CPU restatement of main.py:664-777 (oracle/train_ref.py), which the golden fixture
train_d22_96.npz pins to the imported reference.

Tolerance: this randomly initialised network in train mode amplifies f32 rounding strongly (BN over 72 samples,
ReLU gates): the reference's own f32 gradients differ from the exact (f64) ones by up to ~5 % of a tensor's
largest entry (isolated entries; in L2 norm the gap is ~1e-4).  The HIP f32 path is therefore held, per tensor and
in relative L2 norm, to max(3 x that measured noise, 1e-2) against the f64 values (a ReLU gate that flips in one
implementation but not in the other moves a 72-sample BN-bias gradient by ~0.3 %) -- wiring errors (a missing skip-path gradient, a wrong saved tensor) are O(1) and still fail."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup():
    from pytorch_pose_proposal_network_amd import synth, prng
    from oracle import forward_ref as Fr, targets_ref as T
    g = np.load(os.path.join(GOLDEN, "train_d22_96.npz"))
    size, batch = int(g["size"]), int(g["batch"])
    sd = synth.make_state_dict(str(g["arch"]), int(g["seed_w"]))
    x = Fr.normalize_u8(prng.u8_frames(int(g["seed_in"]), batch, (size, size)))
    tg = T.synthetic_batch(int(g["seed_t"]), batch, insize=(size, size), outsize=(size // 16, size // 16))
    return g, sd, x, tg, size


def _rel(a, b):
    """relative L2 error; a tensor whose exact value is ~0 (conv2.bias: BN removes its effect) is measured absolutely"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-4, np.sqrt((b ** 2).sum())))


def test_trainer_f32_matches_oracle():
    from pytorch_pose_proposal_network_amd import lib as L
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    from oracle import train_ref
    g, sd, x, tg, size = _setup()
    w0, base = g["w_before"], g["base"]
    torch.set_num_threads(8)
    r64 = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), float(g["alpha"]))
    r32 = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), float(g["alpha"]),
                                        dtype=torch.float32)
    dev = torch.device("cuda")
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(size, size), lr_weights=float(g["lr_w"]),
                    alpha=float(g["alpha"]))
    tr.task.w.copy_(torch.from_numpy(w0))
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    head = tr.forward(xd)
    assert np.abs(head.cpu().numpy() - r64["head"]).max() <= 1e-4          # BASELINE north_star tolerance
    for n, b in tr.buffers.items():                                        # running stats after one train-mode pass
        assert np.allclose(b.cpu().numpy(), r64["buffers"][n], rtol=2e-4, atol=2e-5), n
    losses, ghead = tr.criterion.forward_backward(head, tgd, coeff=[float(v) / 5 for v in w0])
    assert np.allclose(losses.cpu().numpy(), r64["losses"], rtol=1e-4)
    tr.backward(ghead)
    torch.cuda.synchronize()
    bad = []
    for n in tr.param_names:
        mine = tr.G[n].cpu().numpy().astype(np.float64)
        noise = _rel(r32["grads"][n], r64["grads"][n])
        err = _rel(mine, r64["grads"][n])
        if err > max(3 * noise, 1e-2):
            bad.append((n, err, noise))
    assert not bad, bad[:8]
    # probe gradients of the five losses (GradNorm, main.py:704-717)
    gn = []
    for i in range(5):
        onehot = [1.0 if j == i else 0.0 for j in range(5)]
        _, gi = tr.criterion.forward_backward(head, tgd, coeff=onehot)
        gn.append(float(tr.probe_grad(gi).double().norm().cpu()))
    # the production path: four 6K-channel probes + the limb probe by linearity of the backward pass
    coeff = [float(v) / 5 for v in w0]
    gn2 = tr.probe_norms(head, tgd, coeff, ghead).cpu().numpy()
    assert np.allclose(gn2, np.array(gn), rtol=2e-3), (gn2, gn)
    noise = np.abs(r32["gnorm"] - r64["gnorm"]) / r64["gnorm"]
    assert np.all(np.abs(np.array(gn) - r64["gnorm"]) / r64["gnorm"] <= np.maximum(3 * noise, 1e-2)), (gn, r64["gnorm"])


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_train_step_updates(dtype_name):
    """A full iteration: task weights follow the reference's (fixture values from the reference itself), every
    parameter moves by an Adam step of its gradient, BN statistics advance, a second step runs on the new state."""
    from pytorch_pose_proposal_network_amd import lib as L
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    g, sd, x, tg, size = _setup()
    dev = torch.device("cuda")
    dt = L.PPN_F32 if dtype_name == "f32" else L.PPN_BF16
    lr = 7e-4
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=dt, insize=(size, size), lr=lr, lr_weights=float(g["lr_w"]),
                    alpha=float(g["alpha"]))
    tr.task.w.copy_(torch.from_numpy(g["w_before"]))
    tr.base = torch.from_numpy(g["base"]).to(dev)
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    p0 = tr.flat.clone()
    losses, w = tr.train_step(xd, tgd)
    torch.cuda.synchronize()
    tol = 2e-4 if dtype_name == "f32" else 3e-2
    assert np.allclose(losses.cpu().numpy(), g["losses"], rtol=tol)
    # G_i = w_i * ||dL_i/dW|| carry the f32 rounding noise of this network (see module docstring): the bound is the
    # oracle's own f32-vs-f64 distance on the same quantity, as for the gradients above -- not a flat percentage
    from oracle import train_ref
    torch.set_num_threads(8)
    r64 = train_ref.train_iteration_ref(sd, x, tg, g["w_before"], g["base"], "drn_d_22", (size, size), float(g["alpha"]))
    r32 = train_ref.train_iteration_ref(sd, x, tg, g["w_before"], g["base"], "drn_d_22", (size, size), float(g["alpha"]),
                                        dtype=torch.float32)
    noise = np.abs(r32["gnorm"] - r64["gnorm"]) / r64["gnorm"]
    G = tr.task.log[:5].cpu().numpy().astype(np.float64)
    relG = np.abs(G - g["G_f64"]) / g["G_f64"]
    print(f"{dtype_name}: |G - G_f64| / G_f64 = {[float('%.2e' % v) for v in relG]}, oracle f32-vs-f64 noise "
          f"{[float('%.2e' % v) for v in noise]}")
    # measured: f32 < 1e-4 on every G_i (the NORMS are far less noisy than single gradient entries), bf16 0.4-1.8 %
    if dtype_name == "f32":
        assert np.all(relG <= np.maximum(3 * noise, 1e-4)), (relG, noise)
    else:
        assert np.all(relG <= 0.05), relG
    dw = np.abs(w.cpu().numpy() - g["w_final"]).max()
    print(f"{dtype_name}: max |w - w_final| = {dw:.2e}")
    # measured 1.2e-7 in both modes: Adam's first step moves every w_i by exactly lr_w (2.5e-2) in the direction of
    # sign(dLgrad/dw_i), so this line checks the SIGNS of the GradNorm weight gradient and the renormalisation
    assert dw <= 1e-5
    assert abs(float(w.mean().cpu()) - 1.0) < 1e-6
    # Adam, step 1: |delta| = lr * |g| / (|g| + eps) <= lr, and == lr wherever the gradient is not tiny
    delta = (tr.flat - p0).abs()
    gabs = tr.grad.abs()
    assert float(delta.max()) <= lr * 1.0001
    big = gabs > 1e-4
    assert float(big.float().mean()) > 0.5
    assert torch.allclose(delta[big], torch.full_like(delta[big], lr), rtol=2e-3)
    assert torch.equal(torch.sign(tr.flat - p0)[big], -torch.sign(tr.grad)[big])
    assert tr.num_batches_tracked == 1
    losses2, _ = tr.train_step(xd, tgd)
    assert torch.isfinite(losses2).all() and torch.isfinite(tr.flat).all()
    sd2 = tr.state_dict()
    assert len(sd2) == 207 and "backbone.0.1.num_batches_tracked" in sd2


def test_bf16_gradients_track_f32():
    """bf16 MFMA mode (bf16 activations / packed weights, f32 accumulation, f32 master weights and gradients) against
    the f32 mode.  This randomly initialised network is chaotic under 0.4 % perturbations: merely rounding the FORWARD
    tensors of the exact f32 pipeline to bf16 (weights, conv outputs, BN outputs) already turns the gradients by
    cos 0.89 (stem) .. 0.999 (conv3) -- rounding the backward tensors changes nothing (cos 1.0000, measured).  The bf16
    kernels must do no worse than that emulation by more than a small margin, tensor by tensor."""
    from pytorch_pose_proposal_network_amd import lib as L, train as T
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    g, sd, x, tg, size = _setup()
    dev = torch.device("cuda")
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    orig_conv, orig_bnf = T.conv2d_nhwc, T.bn_train_forward

    def rb(t):
        return t.to(torch.bfloat16).float()

    def grads_of(dt, emulate):
        def convf(xx, w, *a, **k):
            if k.get("dgrad_of") or not emulate:
                return orig_conv(xx, w, *a, **k)
            o = orig_conv(xx, rb(w), *a, **k)
            return o if k.get("nchw_f32") else rb(o)

        def bnf(*a, **k):
            r = orig_bnf(*a, **k)              # (y, saved) or, with emit_stats (16-bit mode only), (y, saved, ConvStats)
            if emulate and r[0] is not None:
                r[0].copy_(rb(r[0]))
            return r

        T.conv2d_nhwc, T.bn_train_forward = convf, bnf
        try:
            tr = PPNTrainer("drn_d_22", sd, compute_dtype=dt, insize=(size, size))
            head = tr.forward(xd)
            _, gh = tr.criterion.forward_backward(head, tgd, coeff=[0.2] * 5)
            tr.backward(gh)
        finally:
            T.conv2d_nhwc, T.bn_train_forward = orig_conv, orig_bnf
        return {n: tr.G[n].double().cpu() for n in tr.param_names}

    def cos(a, b):
        a, b = a.reshape(-1), b.reshape(-1)
        return float(torch.dot(a, b) / (a.norm() * b.norm()))

    f32, emu, bf = grads_of(L.PPN_F32, False), grads_of(L.PPN_F32, True), grads_of(L.PPN_BF16, False)
    checked = 0
    for n in f32:
        if f32[n].numel() < 64 or float(f32[n].norm()) < 1e-4:
            continue
        c_bf, c_emu = cos(bf[n], f32[n]), cos(emu[n], f32[n])
        ratio = float(bf[n].norm() / f32[n].norm())
        if f32[n].numel() >= 4096:                                 # convolution weights
            assert c_bf >= c_emu - 0.08 and c_bf >= 0.75 and 0.9 <= ratio <= 1.1, (n, c_bf, c_emu, ratio)
        else:                                                      # 64..512-element BN vectors scatter more
            assert c_bf >= c_emu - 0.2 and c_bf >= 0.6 and 0.7 <= ratio <= 1.4, (n, c_bf, c_emu, ratio)
        checked += 1
    assert checked > 60 and cos(bf["conv3.weight"], f32["conv3.weight"]) > 0.995


def test_get_baseloss_matches_eval_oracle():
    """main.py:578-621: eval-mode losses averaged over the batches (here two), vs forward_ref (eval BN) + loss_ref."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    from oracle import forward_ref as Fr, loss_ref as Lr, targets_ref as Tg
    g = np.load(os.path.join(GOLDEN, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    sd = synth.make_state_dict("drn_d_22", int(g["seed_w"]), bn_stats=stats)
    dev = torch.device("cuda")
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(96, 96))
    batches, ref = [], np.zeros(5)
    for i in range(2):
        x = Fr.normalize_u8(prng.u8_frames(40 + i, 2, (96, 96)))
        tg = Tg.synthetic_batch(60 + 2 * i, 2, insize=(96, 96), outsize=(6, 6))
        head = Fr.forward_ref(sd, x, "drn_d_22")
        ref += np.array([float(v) for v in Lr.ppn_loss_ref(head, {k: torch.from_numpy(v) for k, v in tg.items()},
                                                          insize=(96, 96))])
        batches.append((torch.as_tensor(x).to(dev), {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}))
    base = tr.get_baseloss(batches).cpu().numpy()
    assert np.allclose(base, ref / 2, rtol=2e-4), (base, ref / 2)


def test_checkpoint_roundtrip_and_torch_compatibility(tmp_path):
    """PPNTrainer.checkpoint() is the dict of main.py:443-451: torch.optim.Adam / nn.Linear / the reference's state_dict
    names load it; resuming from it continues bit-identically."""
    from pytorch_pose_proposal_network_amd import lib as L, arch as A
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    g, sd, x, tg, size = _setup()
    dev = torch.device("cuda")
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(size, size))
    tr.train_step(xd, tgd)
    tr.train_step(xd, tgd)
    path = tmp_path / "PPN_model_2.pth.tar"
    torch.save(tr.checkpoint(epoch=2, best_AP=0.25), path)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "arch", "state_dict", "weight_state_dict", "best_AP", "optimizerM", "optimizerR"}
    assert len(ck["state_dict"]) == 207
    # torch objects of the reference accept it
    wm = torch.nn.Linear(5, 1, bias=False)
    wm.load_state_dict(ck["weight_state_dict"])
    params = [torch.nn.Parameter(torch.zeros(shp)) for n, shp in A.param_spec("drn_d_22")
              if not n.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    optM = torch.optim.Adam(params, lr=7e-4)
    optM.load_state_dict(ck["optimizerM"])
    assert int(optM.state[params[0]]["step"]) == 2
    optR = torch.optim.Adam(wm.parameters(), lr=7e-4)
    optR.load_state_dict(ck["optimizerR"])
    # resume: a fresh trainer loaded from the checkpoint takes the same third step
    tr2 = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(size, size))
    assert tr2.load_checkpoint(ck) == 2
    tr2.base = tr.base.clone()
    tr2.num_batches_tracked = tr.num_batches_tracked
    l1, w1 = tr.train_step(xd, tgd)
    l2, w2 = tr2.train_step(xd, tgd)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(w1, w2)
    assert torch.equal(tr.flat, tr2.flat) and torch.equal(tr.opt.exp_avg_sq, tr2.opt.exp_avg_sq)


def test_trainer_other_basicblock_depth():
    """DRN-D-38 (3/4/6/3 BasicBlocks): same units, more of them -- forward, losses and a spread of gradients vs the
    CPU autograd restatement."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    from oracle import forward_ref as Fr, targets_ref as T, train_ref
    arch, size, batch = "drn_d_38", 96, 2
    sd = synth.make_state_dict(arch, 9)
    x = Fr.normalize_u8(prng.u8_frames(21, batch, (size, size)))
    tg = T.synthetic_batch(31, batch, insize=(size, size), outsize=(6, 6))
    w0, base = [1.0] * 5, [1.0] * 5
    torch.set_num_threads(8)
    r64 = train_ref.train_iteration_ref(sd, x, tg, w0, base, arch, (size, size))
    r32 = train_ref.train_iteration_ref(sd, x, tg, w0, base, arch, (size, size), dtype=torch.float32)
    dev = torch.device("cuda")
    tr = PPNTrainer(arch, sd, compute_dtype=L.PPN_F32, insize=(size, size))
    head = tr.forward(torch.as_tensor(x).to(dev))
    assert np.abs(head.cpu().numpy() - r64["head"]).max() <= 1e-4
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    losses, gh = tr.criterion.forward_backward(head, tgd, coeff=[0.2] * 5)
    assert np.allclose(losses.cpu().numpy(), r64["losses"], rtol=1e-4)
    tr.backward(gh)
    torch.cuda.synchronize()
    assert len(tr.param_names) == len(r64["grads"])
    bad = [(n, _rel(tr.G[n].cpu().numpy(), r64["grads"][n]), _rel(r32["grads"][n], r64["grads"][n]))
           for n in tr.param_names]
    bad = [b for b in bad if b[1] > max(3 * b[2], 1e-2)]
    assert not bad, bad[:6]


@pytest.mark.parametrize("second_order", [True, False])
def test_training_converges_on_a_fixed_batch(second_order):
    """Functional check of the whole step (forward, loss, backward, GradNorm weights, Adam): 40 iterations on one
    fixed batch drive every loss down by more than an order of magnitude, and the bf16 mode follows the f32 mode.
    (With the reference's second-order term the descent is a little slower: Lgrad's gradient also moves the weights.)"""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng, targets
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    dev = torch.device("cuda")
    size, B = 96, 4
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(5, B, (size, size)))).to(dev)
    tg = targets.synthetic_targets(11, B, (size, size), device=dev)
    final = {}
    for name, dt in (("f32", L.PPN_F32), ("bf16", L.PPN_BF16)):
        tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 3), compute_dtype=dt, insize=(size, size),
                        lr=7e-4, second_order=second_order)
        first = None
        for it in range(40):
            losses, w = tr.train_step(x, tg)
            if first is None:
                first = losses.clone()
        torch.cuda.synchronize()
        assert torch.isfinite(losses).all() and torch.isfinite(tr.flat).all()
        lim0 = 0.1 if second_order else 0.05
        assert float(losses[0]) < lim0 * float(first[0]) and float(losses[4]) < 0.02 * float(first[4]), (first, losses)
        assert abs(float(w.mean()) - 1.0) < 1e-5 and float(w.min()) > 0.5
        final[name] = losses.cpu().numpy()
    # Same iteration count on a steeply falling curve (the third loss halves every ~8 iterations here): a change of the
    # f32 accumulation ORDER inside the bf16 weight-gradient kernel moved bf16 / f32 at iteration 40 from 1.15 to 1.59
    # on that component while both runs kept converging (0.067 and 0.085 four iterations later, f32 0.065).
    assert np.allclose(final["bf16"], final["f32"], rtol=1.0), final


def test_second_order_gradients_match_oracle():
    """second_order=True: the model gradients are d loss/d theta + d Lgrad/d theta, the exact semantics of main.py:755-759
    (Lgrad.backward() through the create_graph probe gradients), against the CPU autograd restatement, which in f32 was
    bit-identical to the imported reference when the fixture was made (fixture values g2*)."""
    from pytorch_pose_proposal_network_amd import lib as L
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    from oracle import train_ref
    g, sd, x, tg, size = _setup()
    w0, base = g["w_before"], g["base"]
    torch.set_num_threads(8)
    r64 = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), float(g["alpha"]),
                                        second_order=True)
    r32 = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), float(g["alpha"]),
                                        dtype=torch.float32, second_order=True)
    first = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), float(g["alpha"]))
    dev = torch.device("cuda")
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(size, size), lr_weights=float(g["lr_w"]),
                    alpha=float(g["alpha"]), second_order=True)
    tr.task.w.copy_(torch.from_numpy(w0))
    tr.base = torch.from_numpy(base).to(dev)
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    head = tr.forward(xd)
    coeff = [float(v) / 5 for v in w0]
    losses, ghead = tr.criterion.forward_backward(head, tgd, coeff=coeff)
    scratch = torch.empty_like(head)
    grads = []
    for i in range(4):
        tr.criterion.unary_backward(head, tgd, [1.0 if j == i else 0.0 for j in range(4)], out=scratch)
        grads.append(tr.probe_grad(scratch, channels_used=108))
    so = dict(head=head, targets=tgd, losses=losses, coeff=coeff, unary=grads)
    tr.backward(ghead, so=so)
    torch.cuda.synchronize()
    assert np.allclose(so["gnorm"].cpu().numpy(), r64["gnorm"], rtol=1e-2)
    bad, moved = [], 0
    for n in tr.param_names:
        mine = tr.G[n].cpu().numpy().astype(np.float64)
        noise = _rel(r32["grads"][n], r64["grads"][n])
        err = _rel(mine, r64["grads"][n])
        if err > max(3 * noise, 2e-2):
            bad.append((n, err, noise))
        if _rel(first["grads"][n], r64["grads"][n]) > 0.05:
            moved += 1                      # the second-order term really changes this tensor
    assert moved > 50, moved
    assert not bad, bad[:8]


def test_trainer_bottleneck_arch():
    """DRN-D-54 (Bottleneck units, drn.py:59-97; 2048-channel trunk into the PPN head): forward, losses and every
    gradient of the first-order pass vs the CPU autograd restatement, then one full (second-order) train_step."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    from oracle import forward_ref as Fr, targets_ref as T, train_ref
    arch, size, batch = "drn_d_54", 96, 2
    sd = synth.make_state_dict(arch, 4)
    x = Fr.normalize_u8(prng.u8_frames(22, batch, (size, size)))
    tg = T.synthetic_batch(32, batch, insize=(size, size), outsize=(6, 6))
    torch.set_num_threads(8)
    r64 = train_ref.train_iteration_ref(sd, x, tg, [1.0] * 5, [1.0] * 5, arch, (size, size))
    r32 = train_ref.train_iteration_ref(sd, x, tg, [1.0] * 5, [1.0] * 5, arch, (size, size), dtype=torch.float32)
    dev = torch.device("cuda")
    tr = PPNTrainer(arch, sd, compute_dtype=L.PPN_F32, insize=(size, size))
    xd = torch.as_tensor(x).to(dev)
    head = tr.forward(xd)
    noise_h = np.abs(r32["head"] - r64["head"]).max()
    assert np.abs(head.cpu().numpy() - r64["head"]).max() <= max(1e-4, 1.5 * noise_h)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    losses, gh = tr.criterion.forward_backward(head, tgd, coeff=[0.2] * 5)
    assert np.allclose(losses.cpu().numpy(), r64["losses"], rtol=2e-4)
    tr.backward(gh)
    torch.cuda.synchronize()
    assert len(tr.param_names) == len(r64["grads"])
    bad = [(n, _rel(tr.G[n].cpu().numpy(), r64["grads"][n]), _rel(r32["grads"][n], r64["grads"][n]))
           for n in tr.param_names]
    # 54 layers deep: the f32-vs-f64 gap of the restatement itself is ~1 % in L2 norm here, and two f32 implementations
    # sit a few per cent apart; a wiring error would show as tens of per cent
    bad = [b for b in bad if b[1] > max(5 * b[2], 5e-2)]
    assert not bad, bad[:6]
    losses2, w = tr.train_step(xd, tgd)
    assert torch.isfinite(losses2).all() and torch.isfinite(tr.flat).all() and abs(float(w.mean()) - 1) < 1e-5


def test_train_step_at_full_size_batch32_384():
    """BASELINE configs[3] per-GPU shard at its real size (batch 32, 384x384, bf16): the step runs the tile
    instantiations the benchmark runs (tests/test_conv_tiles_gpu.py holds each of them to an fp64 reference);
    here: two identical steps from the same state give bit-identical losses, gradients and parameters
    (reproducible kernels: fixed-order folds, no float atomics), the bf16 losses agree with the exact-f32 mode's,
    and the gradient norms of both modes agree (bf16 storage noise on a random network)."""
    from pytorch_pose_proposal_network_amd import lib as L, prng, synth, targets
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    B, S = 32, 384
    dev = torch.device("cuda")
    sd = synth.make_state_dict("drn_d_22", 0)
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(99, B, (S, S)))).to(dev)
    tg = targets.synthetic_targets(99, B, (S, S), device=dev)
    base = torch.tensor([2.0, 1.5, 0.6, 0.4, 3.0], device=dev)
    runs = {}
    for name, dt in (("bf16_a", L.PPN_BF16), ("bf16_b", L.PPN_BF16), ("f32", L.PPN_F32)):
        tr = PPNTrainer("drn_d_22", sd, compute_dtype=dt, insize=(S, S))
        tr.base = base.clone()
        losses, w = tr.train_step(x, tg)
        torch.cuda.synchronize()
        runs[name] = (losses.cpu(), tr.grad.clone(), tr.flat.clone(), w.cpu())
        assert torch.isfinite(losses).all() and torch.isfinite(tr.grad).all() and torch.isfinite(tr.flat).all()
        del tr
        torch.cuda.empty_cache()
    a, b, f = runs["bf16_a"], runs["bf16_b"], runs["f32"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    print("losses bf16", a[0].numpy().round(4).tolist(), "f32", f[0].numpy().round(4).tolist())
    assert torch.allclose(a[0], f[0], rtol=1e-2)                 # measured: 0.4 % at most
    gn_a, gn_f = float(a[1].double().norm()), float(f[1].double().norm())
    cos = float((a[1].double() @ f[1].double()) / (gn_a * gn_f))
    print(f"|grad| bf16 {gn_a:.4e} f32 {gn_f:.4e} cos {cos:.4f}")
    assert abs(gn_a / gn_f - 1.0) < 0.02 and cos > 0.93          # measured: ratio 0.9997, cosine 0.961


def test_bf16_limb_probe_by_linearity_vs_direct_pass():
    """GradNorm's limb probe gradient is taken by linearity, (dL/dW - sum_{i<4} c_i dL_i/dW) / c_4, unless that
    remainder drowns in bf16 rounding noise (then the direct fifth pass runs, PPNTrainer._limb_probe).  In bf16 mode the
    norm from the production path must agree with the direct pass, for balanced task weights and for a limb weight so
    small that the subtraction would be pure noise."""
    from pytorch_pose_proposal_network_amd import lib as L
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    g, sd, x, tg, size = _setup()
    dev = torch.device("cuda")
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    for w in ([1.0, 1.0, 1.0, 1.0, 1.0], [1.6, 1.2, 1.2, 0.99, 0.01]):
        tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_BF16, insize=(size, size), second_order=False)
        head = tr.forward(xd)
        coeff = [v / 5 for v in w]
        _, ghead = tr.criterion.forward_backward(head, tgd, coeff=coeff)
        tr.backward(ghead)
        gn = tr.probe_norms(head, tgd, coeff, ghead).cpu().numpy()
        _, g4 = tr.criterion.forward_backward(head, tgd, coeff=[0.0, 0.0, 0.0, 0.0, 1.0])
        direct = float(tr.probe_grad(g4).double().norm().cpu())
        print(f"task weights {w}: gnorm_4 production {gn[4]:.5e}  direct pass {direct:.5e}")
        assert abs(gn[4] / direct - 1.0) < 0.05, (gn[4], direct)


def test_trainer_f32_matches_oracle_at_384_batch4():
    """The training step at the benchmarked RESOLUTION (384x384; batch 4 so that the CPU autograd oracle of
    main.py:664-777 finishes in seconds on the box's 16 threads), f32 mode: head 1e-4, losses 1e-4, every parameter
    gradient by the noise rule of this module, G_i 1e-4 (or 3x the oracle's own f32-vs-f64 noise).  The batch-32 shard
    then follows from the additivity checked in test_batch32_step_is_assembled_from_its_halves."""
    from pytorch_pose_proposal_network_amd import lib as L, prng, synth
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    from oracle import forward_ref as Fr, targets_ref as T, train_ref
    size, batch, alpha = 384, 4, 0.12
    sd = synth.make_state_dict("drn_d_22", 0)
    x = Fr.normalize_u8(prng.u8_frames(4242, batch, (size, size)))
    tg = T.synthetic_batch(4343, batch, insize=(size, size), outsize=(size // 16, size // 16))
    w0, base = np.array([1.3, 0.8, 1.1, 0.7, 1.1]), np.array([2.0, 1.5, 0.6, 0.4, 3.0])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r64 = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), alpha)
    r32 = train_ref.train_iteration_ref(sd, x, tg, w0, base, "drn_d_22", (size, size), alpha, dtype=torch.float32)
    dev = torch.device("cuda")
    tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(size, size), lr_weights=0.025, alpha=alpha)
    tr.task.w.copy_(torch.from_numpy(w0).float())
    xd = torch.as_tensor(x).to(dev)
    tgd = {k: torch.from_numpy(v).to(dev) for k, v in tg.items()}
    head = tr.forward(xd)
    err = float(np.abs(head.cpu().numpy() - r64["head"]).max())
    print(f"384x384 batch 4 train-mode head: max|hip - f64 oracle| = {err:.3e}")
    assert err <= 1e-4
    for n, b in tr.buffers.items():
        assert np.allclose(b.cpu().numpy(), r64["buffers"][n], rtol=2e-4, atol=2e-5), n
    coeff = [float(v) / 5 for v in w0]
    losses, ghead = tr.criterion.forward_backward(head, tgd, coeff=coeff)
    assert np.allclose(losses.cpu().numpy(), r64["losses"], rtol=1e-4), (losses, r64["losses"])
    tr.backward(ghead)
    torch.cuda.synchronize()
    bad, worst = [], (None, 0.0, 0.0)
    for n in tr.param_names:
        noise = _rel(r32["grads"][n], r64["grads"][n])
        e = _rel(tr.G[n].cpu().numpy().astype(np.float64), r64["grads"][n])
        if e > worst[1]:
            worst = (n, e, noise)
        if e > max(3 * noise, 1e-2):
            bad.append((n, e, noise))
    print(f"gradients: worst relative L2 error {worst[1]:.2e} ({worst[0]}; the oracle's own f32-vs-f64 gap there {worst[2]:.2e})")
    assert not bad, bad[:8]
    gn = tr.probe_norms(head, tgd, coeff, ghead).cpu().numpy().astype(np.float64)
    noise = np.abs(r32["gnorm"] - r64["gnorm"]) / r64["gnorm"]
    rel = np.abs(gn - r64["gnorm"]) / r64["gnorm"]
    print(f"||dL_i/dW||: relative error {[float('%.2e' % v) for v in rel]} (oracle f32 noise {[float('%.2e' % v) for v in noise]})")
    assert np.all(rel <= np.maximum(3 * noise, 1e-4)), (rel, noise)


def test_batch32_step_is_assembled_from_its_halves():
    """What carries the batch-4 oracle check to the benchmarked batch 32 (f32, 384x384): the loss is a batch MEAN and the
    BN sums are ADDITIVE, so a batch-32 pass equals what its two halves give where the algebra allows it -- the five
    losses and the head gradient of a batch-32 head from its halves, and the first BN's batch statistics (its input does
    not depend on the batch) from the halves' statistics."""
    from pytorch_pose_proposal_network_amd import lib as L, prng, synth, targets
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    size, B = 384, 32
    dev = torch.device("cuda")
    sd = synth.make_state_dict("drn_d_22", 0)
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(77, B, (size, size)))).to(dev)
    tg = targets.synthetic_targets(78, B, (size, size), device=dev)
    stats = []
    for lo, hi in ((0, B), (0, B // 2), (B // 2, B)):
        tr = PPNTrainer("drn_d_22", sd, compute_dtype=L.PPN_F32, insize=(size, size))
        head = tr.forward(x[lo:hi].contiguous())
        n = (hi - lo) * size * size
        mean = tr.buffers["backbone.0.1.running_mean"].double() / 0.1                     # momentum 0.1, start 0
        var = (tr.buffers["backbone.0.1.running_var"].double() - 0.9) / 0.1 * (n - 1) / n   # unbiased -> biased
        stats.append((mean.cpu(), var.cpu(), head if lo == 0 and hi == B else None, tr if hi - lo == B else None))
        if hi - lo != B:
            del tr
        torch.cuda.empty_cache()
    (m, v, head, tr), (ma, va, _, _), (mb, vb, _, _) = stats
    m_exp = (ma + mb) / 2
    v_exp = (va + ma ** 2 + vb + mb ** 2) / 2 - m_exp ** 2
    assert torch.allclose(m, m_exp, rtol=1e-5, atol=1e-6), float((m - m_exp).abs().max())
    assert torch.allclose(v, v_exp, rtol=1e-4, atol=1e-6), float((v - v_exp).abs().max())
    coeff = [0.26, 0.16, 0.22, 0.14, 0.22]
    l32, g32 = tr.criterion.forward_backward(head, tg, coeff=coeff)
    g32 = g32.clone()
    parts = []
    for lo, hi in ((0, B // 2), (B // 2, B)):
        th = {k: t[lo:hi].contiguous() for k, t in tg.items()}
        lh, gh = tr.criterion.forward_backward(head[lo:hi].contiguous(), th, coeff=coeff)
        parts.append((lh.clone(), gh.clone()))
    assert torch.allclose(l32, (parts[0][0] + parts[1][0]) / 2, rtol=1e-5), (l32, parts)
    gcat = torch.cat([parts[0][1], parts[1][1]]) / 2
    assert torch.allclose(g32, gcat, rtol=1e-5, atol=1e-9), float((g32 - gcat).abs().max())


@pytest.mark.parametrize("dtype_code", ["f32", "bf16"])
def test_stacked_probe_grads_equal_the_four_probe_passes(dtype_code):
    """PPNTrainer._stacked_unary_probe_grads (the four GradNorm probe passes with their convolutions stacked along the
    batch dimension; what train_step runs) against probe_grad() pass by pass: bit-identical gradients."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng, targets, config as cfg
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    dev = torch.device("cuda")
    size, B = 96, 3
    dt = L.PPN_F32 if dtype_code == "f32" else L.PPN_BF16
    tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 5), compute_dtype=dt, insize=(size, size))
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(8, B, (size, size)))).to(dev)
    tg = targets.synthetic_targets(13, B, (size, size), device=dev)
    head = tr.forward(x)
    scratch = torch.empty_like(head)
    ref = []
    for i in range(4):
        tr.criterion.unary_backward(head, tg, [1.0 if j == i else 0.0 for j in range(4)], out=scratch)
        ref.append(tr.probe_grad(scratch, channels_used=6 * cfg.K).clone())
    got = tr._stacked_unary_probe_grads(head, tg, scratch)
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert float(b.abs().max()) > 0 and torch.equal(a, b)


def test_train_step_bitwise_reproducible_at_full_size():
    """Two trainers from one checkpoint, six full-size steps on one batch: every loss, task weight and parameter bitwise
    equal.  Three streams, the ping-pong weight-gradient kernel and the fixed-order folds leave no room for run-to-run
    differences; a missing event or barrier would (tools/soak_train.py runs the same screen for longer)."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng, targets
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    dev = torch.device("cuda")
    size, B = 384, 32
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(21, B, (size, size)))).to(dev)
    tg = targets.synthetic_targets(22, B, (size, size), device=dev)
    runs = []
    for _ in range(2):
        tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 7), compute_dtype=L.PPN_BF16, insize=(size, size),
                        lr=2e-4)
        hist = []
        for _it in range(6):
            losses, w = tr.train_step(x, tg)
            hist.append(torch.cat([losses, w]).clone())
        torch.cuda.synchronize()
        runs.append((torch.stack(hist).cpu(), tr.flat.clone().cpu()))
        del tr
    (h0, p0), (h1, p1) = runs
    assert torch.isfinite(h0).all() and torch.isfinite(p0).all()
    assert torch.equal(h0, h1) and torch.equal(p0, p1)
    assert float(h0[-1, 4]) < float(h0[0, 4])                      # and it trains: the limb loss falls


def test_batched_weight_pack_leaves_the_step_bitwise_unchanged(monkeypatch):
    """train.repack_all (one ppn_pack_table_run launch at the head of every pass, round 4) against the per-view packs
    of conv2d_nhwc (PPN_TRAIN_BATCHED_PACK=0): four steps at 192 x 192, every loss, task weight and parameter bitwise
    equal -- also after an in-place edit of a parameter between two steps (the next pass must see it)."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng, targets, train as T
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    dev = torch.device("cuda")
    size, B = 192, 4
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(31, B, (size, size)))).to(dev)
    tg = targets.synthetic_targets(32, B, (size, size), device=dev)
    runs = []
    for batched in (False, True):
        monkeypatch.setattr(T, "_BATCHED_PACK", batched)
        tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 7), compute_dtype=L.PPN_BF16, insize=(size, size),
                        lr=2e-4)
        hist, packed = [], []
        for it in range(4):
            if it == 2:
                tr.P["conv2.weight"].mul_(1.01)                                  # in-place edit between two passes
            losses, w = tr.train_step(x, tg)
            hist.append(torch.cat([losses, w]).clone())
            packed.append(T.repack_all(dev) if batched else 0)
        torch.cuda.synchronize()
        runs.append((torch.stack(hist).cpu(), tr.flat.clone().cpu()))
        if batched:
            assert packed[0] >= 60 and packed[-1] == packed[0], packed           # every weight view of the step is in the table
        del tr
    (h0, p0), (h1, p1) = runs
    assert torch.isfinite(h0).all() and torch.equal(h0, h1) and torch.equal(p0, p1)


@pytest.mark.parametrize("zero_task", [None, 2])
def test_speculative_tail_equals_the_read_back_first_order(zero_task):
    """The second-order tail enqueues its forward-mode half (tangents through bn0_2 .. conv3, all five streams, unit
    directions divided on the device) BEFORE the host has read the probe norms, and keeps it when the host values agree
    (PPN_TRAIN_SPECULATE_TAIL, round 4).  Against the read-back-first order (speculation off) on the same trainer state:
    * all five streams active: every gradient BITWISE equal (round 5: both orders multiply g_i by the same device-computed
      reciprocal 1 / ||g_i||, which travels to the host with the other scalars), losses identical;
    * one task weight exactly zero (kappa_i = 0: that stream is inactive): the speculative tensors are dropped and the
      tangents rebuilt for the four active streams -- bitwise the same gradients as without speculation."""
    from pytorch_pose_proposal_network_amd import lib as L, synth, prng, targets
    from pytorch_pose_proposal_network_amd.trainer import PPNTrainer
    dev = torch.device("cuda")
    size, B = 192, 4
    x = torch.from_numpy(synth.normalized_frames(prng.u8_frames(41, B, (size, size)))).to(dev)
    tg = targets.synthetic_targets(42, B, (size, size), device=dev)
    out = []
    for spec in (False, True):
        tr = PPNTrainer("drn_d_22", synth.make_state_dict("drn_d_22", 7), compute_dtype=L.PPN_F32, insize=(size, size))
        tr._speculate_tail = spec
        tr.base = torch.tensor([1.0, 0.8, 1.2, 0.9, 1.1], device=dev)
        if zero_task is not None:
            tr.task.w[zero_task] = 0.0
            tr.task.touched()
        losses, gn, _ = tr.local_pass(x, tg)
        torch.cuda.synchronize()
        out.append((losses.cpu(), gn.cpu(), tr.grad.clone().cpu()))
        del tr
    (l0, g0, p0), (l1, g1, p1) = out
    assert torch.equal(l0, l1) and torch.equal(g0, g1) and torch.isfinite(p0).all() and float(p0.abs().max()) > 0
    # round 5: both orders multiply by the same device-computed reciprocal 1 / ||g_i|| -> bitwise equal in every case
    assert torch.equal(p0, p1)
