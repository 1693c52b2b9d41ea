"""Generate the golden fixtures by running the REFERENCE itself (this container only).

Usage:  python tests/golden/make_golden.py [--only forward|decode|nms|loss]

The reference (/root/reference, read-only, never copied) is imported with stub modules for
its absent optional dependencies (recipe: SURVEY.md Appendix B).  For every fixture the
script (1) builds inputs from the repo's integer PRNG, (2) runs the reference's own code
(model.PoseProposalNet / datatest.get_humans_by_feature / datatest.non_maximum_suppression),
(3) checks the repo's CPU oracle (oracle/*.py) against it, and (4) writes inputs' seeds and
expected outputs as small .npz files next to this script.  Fixtures are data only.

The GPU box never sees /root/reference: tests replay the fixtures from seeds.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"


def import_reference():
    sys.path.insert(0, REF)
    sys.argv = ["main.py"]
    import matplotlib
    matplotlib.use("Agg")

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    stub("torchsummary", summary=lambda *a, **k: None)
    stub("gelu", GELU=object)
    tv = stub("torchvision")
    for s in ("transforms", "utils", "datasets", "models"):
        setattr(tv, s, stub("torchvision." + s))
    sk = stub("skimage")
    sk.io = stub("skimage.io")
    sk.transform = stub("skimage.transform")
    stub("skimage.color", gray2rgb=None)
    stub("cv2")
    ia = stub("imgaug")
    ia.augmenters = stub("imgaug.augmenters")
    sh = stub("shapely")
    sh.geometry = stub("shapely.geometry")
    import drn, model, datatest  # noqa: E401
    return drn, model, datatest


def import_reference_main():
    """main.py additionally needs apex / visdom stubs and hard-codes .cuda() (main.py:173): SURVEY App. B."""
    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m
    ap = stub("apex")
    stub("apex.parallel", DistributedDataParallel=object)
    stub("apex.fp16_utils")
    ap.amp = stub("apex.amp")
    ap.optimizers = stub("apex.optimizers")
    stub("apex.multi_tensor_apply", multi_tensor_applier=None)
    stub("visdom", Visdom=object)
    import torch
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.nn.Module.cuda = lambda s, *a, **k: s
    import main
    return main


import numpy as np  # noqa: E402
import torch  # noqa: E402

from pytorch_pose_proposal_network_amd import arch as A, config as cfg, prng, synth  # noqa: E402
from oracle import decode_ref as D, forward_ref as Fr  # noqa: E402


def build_ref_model(drn, model, arch_name, sd_np):
    import torch.nn as nn
    bb = getattr(drn, arch_name)()
    net = model.PoseProposalNet(nn.Sequential(*list(bb.children())[:-2]), local_grid_size=(21, 21))
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in sd_np.items()}
    missing = net.load_state_dict(sd, strict=True)
    return net


def calibrate(net, frames):
    """One train-mode pass with momentum 1.0 -> running stats = batch stats."""
    import torch.nn as nn
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = 1.0
    net.train()
    with torch.no_grad():
        net(frames)
    net.eval()
    return {k: v.detach().numpy().copy() for k, v in net.state_dict().items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def make_forward(drn, model):
    cases = [  # (fixture name, arch, size, batch, full head?)
        ("forward_d22_96", "drn_d_22", 96, 2, True),
        ("forward_d22_384", "drn_d_22", 384, 2, False),
        ("forward_d54_96", "drn_d_54", 96, 1, True),
        ("forward_d38_96", "drn_d_38", 96, 1, True),
    ]
    for name, arch_name, size, batch, full in cases:
        seed_w, seed_cal, seed_in = 0, 1, 1234
        sd = synth.make_state_dict(arch_name, seed_w)
        # the reference's own state_dict must have exactly our names/shapes
        net = build_ref_model(drn, model, arch_name, sd)
        ref_keys = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        spec = dict(A.param_spec(arch_name))
        assert ref_keys == {k: tuple(s) for k, s in spec.items()}, "param_spec mismatch with reference"
        cal = Fr.normalize_u8(prng.u8_frames(seed_cal, 4, (size, size)))
        stats = calibrate(net, cal)
        x_u8 = prng.u8_frames(seed_in, batch, (size, size))
        x = Fr.normalize_u8(x_u8)
        with torch.no_grad():
            ref = net(x).numpy()
        # oracle vs reference (same torch build -> expect ~bit equality)
        sd_cal = synth.make_state_dict(arch_name, seed_w, bn_stats=stats)
        taps = {}
        mine = Fr.forward_ref(sd_cal, x, arch_name, taps=taps).numpy()
        err = float(np.abs(mine - ref).max())
        sat = float(((ref == 0) | (ref == 1)).mean())
        print(f"{name}: oracle-vs-reference max|diff| = {err:.3e}; saturated frac {sat:.4f}; "
              f"head mean {ref.mean():.4f}")
        assert err <= 1e-6, err
        # calibration by the oracle's own train-mode BN must reproduce the reference's stats
        sd2 = synth.make_state_dict(arch_name, seed_w)
        Fr.forward_ref(sd2, cal, arch_name, train_bn=True, momentum=1.0)
        for k, v in stats.items():
            np.testing.assert_allclose(sd2[k], v, rtol=1e-4, atol=1e-5)
        out = {"seed_w": seed_w, "seed_cal": seed_cal, "seed_in": seed_in, "size": size, "batch": batch,
               "arch": arch_name}
        for k, v in stats.items():
            out["bn/" + k] = v
        for k, v in taps.items():
            t = v.numpy().astype(np.float64)
            out["tap/" + k] = np.array([t.mean(), np.abs(t).mean(), t.std()])
        # fp64 evaluation of the same network: the reference's own fp32 rounding noise on this fixture
        sd64 = {k: (torch.from_numpy(v).double() if v.dtype != np.int64 else torch.from_numpy(v))
                for k, v in sd_cal.items()}
        h64 = Fr.forward_ref(sd64, x.double(), arch_name).numpy()
        out["ref_f32_noise"] = float(np.abs(ref - h64).max())
        print(f"   reference fp32 vs fp64 evaluation: max|diff| = {out['ref_f32_noise']:.3e}")
        if full:
            out["head"] = ref
            out["head_f64"] = h64.astype(np.float32)
        else:
            n = 40000
            idx = (prng.raw_u64(prng.stream_seed(99, 0), n) % np.uint64(ref.size)).astype(np.int64)
            out["head_idx"] = idx
            out["head_val"] = ref.reshape(-1)[idx]
            out["head_chan_sum"] = ref.astype(np.float64).sum(axis=(2, 3))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        if size == 384:
            # the calibrated BN statistics double as the synthetic "checkpoint" of bench.py
            d = os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data")
            os.makedirs(d, exist_ok=True)
            np.savez_compressed(os.path.join(d, f"bn_calib_{arch_name}_seed{seed_w}.npz"), **stats)
    # D-54 calibrated at 384 for the config-5 end-to-end run (stats only)
    sd = synth.make_state_dict("drn_d_54", 0)
    net = build_ref_model(drn, model, "drn_d_54", sd)
    stats = calibrate(net, Fr.normalize_u8(prng.u8_frames(1, 2, (384, 384))))
    d = os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data")
    np.savez_compressed(os.path.join(d, "bn_calib_drn_d_54_seed0.npz"), **stats)


def compare_humans(ref_h, ref_s, res):
    humans, scores = D.humans_from_compact(res)
    assert len(humans) == len(ref_h), (len(humans), len(ref_h))
    for a, b, sa, sb in zip(humans, ref_h, scores, ref_s):
        assert sorted(a.keys()) == sorted(b.keys()), (sorted(a.keys()), sorted(b.keys()))
        for k in a:
            assert np.array_equal(a[k], b[k]), (k, a[k], b[k])
            assert sa[k] == sb[k]


def make_decode(datatest):
    heads = []
    for seed in range(7, 7 + 12):
        heads.append(("crowd", seed, synth.planted_crowd_head(seed)))
    for seed in range(100, 104):               # unstructured heads: many candidates, NMS heavy
        C = cfg.lastsize()
        h = prng.uniform01(prng.stream_seed(seed, 0), C * 576).reshape(C, 24, 24)
        h[0:36] = prng.uniform(prng.stream_seed(seed, 1), 36 * 576, 0.2, 1.0).reshape(36, 24, 24)
        h[72:108] = prng.uniform(prng.stream_seed(seed, 2), 36 * 576, 0.05, 0.3).reshape(36, 24, 24)
        heads.append(("random", seed, h.astype(np.float32)))
    out = {}
    for i, (kind, seed, head) in enumerate(heads):
        delta, x, y, w, h, e = D.split_head(head)
        ref_h, ref_s = datatest.get_humans_by_feature(delta, x, y, w, h, e, detection_thresh=0.15)
        res = D.decode_ref(head)
        compare_humans(ref_h, ref_s, res)
        sc = delta[0][delta[0] > np.float32(0.15)]
        assert len(np.unique(sc)) == len(sc), "fixture must have distinct candidate scores"
        print(f"decode {kind} seed {seed}: {len(res['cand'])} candidates, {len(res['selected'])} after NMS, "
              f"{res['n']} humans, {int((res['kp_cell'] >= 0).sum())} keypoints")
        out[f"{i}/kind"] = kind
        out[f"{i}/seed"] = seed
        for k in ("n", "root_cell", "kp_cell", "limb_arg", "bbox", "score", "cand", "selected"):
            out[f"{i}/{k}"] = res[k]
    out["count"] = len(heads)
    np.savez_compressed(os.path.join(HERE, "decode_heads.npz"), **out)


def make_nms(datatest):
    out = {}
    cases = []
    # hand-made boxes: disjoint, nested, identical, touching edges, and IoU straddling 0.3
    base = np.array([[10, 10, 50, 50], [12, 12, 52, 52], [100, 100, 140, 160], [10, 50, 50, 90],
                     [20, 20, 40, 40], [100, 100, 140, 160], [0, 0, 384, 384]], np.float32)
    cases.append((base, np.array([0.9, 0.8, 0.7, 0.6, 0.95, 0.65, 0.5], np.float32)))
    # pairs whose IoU is swept across the 0.3 knife edge in 1-ulp steps of one coordinate
    a = np.array([0, 0, 100, 100], np.float32)
    xs = []
    x = np.float32(53.846153)          # shift giving IoU ~ 0.3 for two 100x100 boxes
    for _ in range(8):
        x = np.nextafter(x, np.float32(0))
    for _ in range(16):
        xs.append(x)
        x = np.nextafter(x, np.float32(1000))
    for x in xs:
        b = np.array([0, x, 100, x + np.float32(100)], np.float32)
        cases.append((np.stack([a, b]), np.array([0.9, 0.8], np.float32)))
    # random crowds
    for seed in range(5):
        n = 40 + 30 * seed
        u = prng.uniform01(prng.stream_seed(500 + seed, 0), n * 5).reshape(n, 5)
        cy, cx = u[:, 0] * 384, u[:, 1] * 384
        hh, ww = 20 + u[:, 2] * 150, 20 + u[:, 3] * 150
        bb = np.stack([cy - hh / 2, cx - ww / 2, cy + hh / 2, cx + ww / 2], 1).astype(np.float32)
        cases.append((bb, u[:, 4].astype(np.float32)))
    for i, (bb, sc) in enumerate(cases):
        ref = datatest.non_maximum_suppression(bb, 0.3, sc)
        mine = D.nms_ref(bb, 0.3, sc)
        assert np.array_equal(ref, mine), (i, ref, mine)
        out[f"{i}/bbox"], out[f"{i}/score"], out[f"{i}/sel"] = bb, sc, ref
        if i < 20:
            ref_l = datatest.non_maximum_suppression(bb, 0.3, sc, limit=1)
            assert np.array_equal(ref_l, D.nms_ref(bb, 0.3, sc, limit=1))
    # score=None path (input order)
    ref = datatest.non_maximum_suppression(base, 0.3)
    assert np.array_equal(ref, D.nms_ref(base, 0.3))
    out["noscore/bbox"], out["noscore/sel"] = base, ref
    out["count"] = len(cases)
    print(f"nms: {len(cases)} cases; knife-edge selections:",
          [len(out[f'{i}/sel']) for i in range(1, 17)])
    np.savez_compressed(os.path.join(HERE, "nms_cases.npz"), **out)


def make_loss():
    """PPNLoss (main.py:125-216): forward values and d(sum c_i L_i)/d(feature_map) from the reference itself."""
    from oracle import loss_ref as Lr, targets_ref as T
    main_mod = import_reference_main()
    crit = main_mod.PPNLoss()
    out = {}
    cases = [("a", 50, 2, [0.2, 0.2, 0.2, 0.2, 0.2]), ("b", 60, 3, [0.31, 0.07, 0.4, 0.12, 0.1]),
             ("limb_only", 70, 1, [0, 0, 0, 0, 1.0]), ("iou_only", 71, 1, [0, 1.0, 0, 0, 0])]
    for tag, seed, batch, coeff in cases:
        tg = T.synthetic_batch(seed, batch)
        head = prng.uniform(prng.stream_seed(seed, 7), batch * cfg.lastsize() * 576, 0.02, 0.98).reshape(
            batch, cfg.lastsize(), 24, 24)
        # make some predictions overlap their targets so that the IoU branch is exercised with gradients
        on = tg["delta"] > 0
        for lo, key, a, b in ((36, "tx", 0.9, 0.03), (54, "ty", 0.95, 0.02), (72, "tw", 1.2, 0.01), (90, "th", 0.8, 0.01)):
            head[:, lo:lo + 18][on] = (tg[key][on] * a + b).astype(np.float32)
        fm = torch.from_numpy(head).clone().requires_grad_(True)
        tt = {k: torch.from_numpy(v) for k, v in tg.items()}
        image = torch.zeros(batch, 3, 384, 384)
        ref = crit(image, fm, tt["delta"], tt["weight"], tt["weight_ij"], tt["tx_half"], tt["ty_half"], tt["tx"],
                   tt["ty"], tt["tw"], tt["th"], tt["te"])
        total = sum(float(c) * l for c, l in zip(coeff, ref))
        total.backward()
        ref_l = np.array([float(l.detach()) for l in ref], np.float32)
        ref_g = fm.grad.numpy()
        my_l, my_g = Lr.loss_and_grad_ref(head, tg, coeff)
        assert np.allclose(my_l, ref_l, rtol=1e-6), (my_l, ref_l)
        assert np.abs(my_g - ref_g).max() <= 1e-7 * max(1.0, np.abs(ref_g).max()), np.abs(my_g - ref_g).max()
        n = 30000
        idx = (prng.raw_u64(prng.stream_seed(seed, 99), n) % np.uint64(ref_g.size)).astype(np.int64)
        # always include every unary position (they carry the IoU / size / coordinate gradients)
        un = np.arange(batch * cfg.lastsize() * 576).reshape(batch, cfg.lastsize(), 576)[:, :108].reshape(-1)
        idx = np.unique(np.concatenate([idx, un]))
        out[f"{tag}/seed"], out[f"{tag}/batch"], out[f"{tag}/coeff"] = seed, batch, np.array(coeff, np.float32)
        out[f"{tag}/losses"] = ref_l
        out[f"{tag}/grad_idx"], out[f"{tag}/grad_val"] = idx, ref_g.reshape(-1)[idx]
        out[f"{tag}/grad_abs_sum"] = np.abs(ref_g.astype(np.float64)).sum(axis=(2, 3))
        print(f"loss {tag}: losses {ref_l}, |grad| max {np.abs(ref_g).max():.4f}; oracle == reference")
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "loss_cases.npz"), **out)


def make_train(drn, model):
    """One training iteration of the reference itself (main.py:664-777) on D-22 at 96x96, batch 2: the imported
    PoseProposalNet in train() mode, the imported PPNLoss, loss.backward(), the five create_graph probe gradients,
    GradNorm, Lgrad.backward(), Adam on the task weights.  Stored: losses, gradient fingerprints of every parameter
    (L2 norm + 24 sampled values) for BOTH d loss/d theta alone (after main.py:683) and the reference's final
    .grad (after main.py:759, with the second-order term), G_i, C_i, Lgrad, d Lgrad/d w, w after
    optimizerR.step() and after the clamp/renormalise, and the BN running stats after the forward."""
    from oracle import targets_ref as T, train_ref
    main_mod = import_reference_main()
    arch_name, size, batch, seed_w, seed_in, seed_t = "drn_d_22", 96, 2, 3, 5, 11
    alpha, lr_w = 0.12, 0.025
    sd_np = synth.make_state_dict(arch_name, seed_w)
    net = build_ref_model(drn, model, arch_name, sd_np)
    net.train()
    x = Fr.normalize_u8(prng.u8_frames(seed_in, batch, (size, size)))
    tg = T.synthetic_batch(seed_t, batch, insize=(size, size), outsize=(size // 16, size // 16))
    tt = {k: torch.from_numpy(v) for k, v in tg.items()}
    crit = main_mod.PPNLoss(insize=(size, size), outsize=(size // 16, size // 16))
    weight_model = torch.nn.Linear(5, 1, bias=False)
    weight_model.weight.data = torch.tensor([[1.2, 0.7, 1.1, 0.9, 1.1]])
    optR = torch.optim.Adam(weight_model.parameters(), lr=lr_w)
    base = np.array([150.0, 0.5, 0.9, 0.6, 200.0], np.float32)
    output = net(x)
    losses = crit(x, output, tt["delta"], tt["weight"], tt["weight_ij"], tt["tx_half"], tt["ty_half"], tt["tx"],
                  tt["ty"], tt["tw"], tt["th"], tt["te"])
    l = [torch.mul(weight_model.weight[0][i], losses[i]) for i in range(5)]
    loss = torch.div(l[0] + l[1] + l[2] + l[3] + l[4], 5)
    net.zero_grad()
    loss.backward(retain_graph=True)
    names = [n for n, _ in net.named_parameters()]
    first = {n: p.grad.detach().clone().numpy() for n, p in net.named_parameters()}
    param = list(net.parameters())
    assert names[-13] == "conv1.weight"
    GR = [torch.autograd.grad(l[i], param[-13], retain_graph=True, create_graph=True) for i in range(5)]
    G = [torch.norm(GR[i][0], 2) for i in range(5)]
    G_avg = torch.div(G[0] + G[1] + G[2] + G[3] + G[4], 5)
    lhat = [torch.div(l[i], float(base[i])) for i in range(5)]
    lhat_avg = torch.div(lhat[0] + lhat[1] + lhat[2] + lhat[3] + lhat[4], 5)
    C = [(G_avg * (torch.div(lhat[i], lhat_avg)) ** alpha).detach().squeeze() for i in range(5)]
    optR.zero_grad()
    gl = torch.nn.L1Loss()
    Lgrad = gl(G[0], C[0]) + gl(G[1], C[1]) + gl(G[2], C[2]) + gl(G[3], C[3]) + gl(G[4], C[4])
    Lgrad.backward()
    dw = weight_model.weight.grad.detach().clone().numpy()[0]
    w_before = weight_model.weight.detach().clone().numpy()[0]
    optR.step()
    w_adam = weight_model.weight.detach().clone().numpy()[0]
    with torch.no_grad():
        weight_model.weight.clamp_(min=0.0)
        weight_model.weight.div_(torch.mean(weight_model.weight))
    w_final = weight_model.weight.detach().clone().numpy()[0]
    total = {n: p.grad.detach().clone().numpy() for n, p in net.named_parameters()}

    # The CPU restatement run in f32 executes the same torch ops in the same order as the reference: it must agree
    # to rounding in both modes.  Run in f64 it gives the noise-free values; the f32-vs-f64 gap of this randomly
    # initialised network in train mode (tiny BN batches, ReLU gates) reaches several per cent on some tensors --
    # that gap, not 1e-6, is the honest tolerance for any f32 implementation of this step.
    f64 = {}
    for so, ref in ((False, first), (True, total)):
        r = train_ref.train_iteration_ref(sd_np, x, tg, w_before, base, arch_name, (size, size), alpha,
                                          dtype=torch.float32, second_order=so)
        assert np.allclose(r["losses"], [float(v.detach()) for v in losses], rtol=1e-6)
        errs = {n: np.abs(r["grads"][n] - ref[n]).max() / max(1e-3, np.abs(ref[n]).max()) for n in names}
        worst = max(errs.values())
        assert worst < 1e-5, (so, sorted(errs.items(), key=lambda kv: -kv[1])[:6])
        assert np.allclose(r["G"], [float(g.detach()) for g in G], rtol=1e-5)
        assert np.allclose(r["C"], [float(c) for c in C], rtol=1e-5)
        assert np.allclose(r["dw"], dw, rtol=1e-5, atol=1e-6)
        print(f"train oracle f32 (second_order={so}) == reference: worst relative gradient error {worst:.2e}")
        f64[so] = train_ref.train_iteration_ref(sd_np, x, tg, w_before, base, arch_name, (size, size), alpha,
                                                second_order=so)
        noise = {n: np.abs(f64[so]["grads"][n] - ref[n]).max() / max(1e-3, np.abs(ref[n]).max()) for n in names}
        print(f"  reference f32 vs restatement f64: worst {max(noise.values()):.3f}, "
              f"median {np.median(list(noise.values())):.4f}")

    out = dict(arch=arch_name, size=size, batch=batch, seed_w=seed_w, seed_in=seed_in, seed_t=seed_t, alpha=alpha,
               lr_w=lr_w, base=base, w_before=w_before, w_adam=w_adam, w_final=w_final, dw=dw,
               losses=np.array([float(v.detach()) for v in losses], np.float32),
               G=np.array([float(g.detach()) for g in G], np.float32), C=np.array([float(c) for c in C], np.float32),
               Lgrad=np.float32(float(Lgrad.detach())), names=np.array(names))
    out["G_f64"], out["C_f64"], out["dw_f64"] = f64[False]["G"], f64[False]["C"], f64[False]["dw"]
    out["gnorm_f64"], out["losses_f64"] = f64[False]["gnorm"], f64[False]["losses"]
    for tag, gd in (("g1", first), ("g2", total), ("g1_f64", f64[False]["grads"]), ("g2_f64", f64[True]["grads"])):
        out[tag + "/norm"] = np.array([np.sqrt((gd[n].astype(np.float64) ** 2).sum()) for n in names])
        idx, val = [], []
        for i, n in enumerate(names):
            k = (prng.raw_u64(prng.stream_seed(1000 + i, 1), 24) % np.uint64(gd[n].size)).astype(np.int64)
            idx.append(k)
            val.append(gd[n].reshape(-1)[k])
        out[tag + "/idx"], out[tag + "/val"] = np.stack(idx), np.stack(val).astype(np.float64 if "f64" in tag else np.float32)
    for n, v in net.state_dict().items():
        if n.endswith("running_mean") or n.endswith("running_var"):
            out["buf/" + n] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "train_d22_96.npz"), **out)
    print("train fixture written: losses", out["losses"], "G", out["G"], "w_final", w_final)


def make_eval(datatest):
    """AP values of the reference's own datatest.evaluation (-> eval_helpers / evaluateAP) on synthetic pck_objects."""
    import contextlib
    import copy
    import io
    from pytorch_pose_proposal_network_amd import evaluate
    seeds = [1, 2, 3, 4, 5, 6]
    sizes = [12, 12, 20, 6, 30, 1]
    exp = []
    for seed, n in zip(seeds, sizes):
        obj = synth.eval_case(seed, n)
        with contextlib.redirect_stdout(io.StringIO()):
            ref = datatest.evaluation(copy.deepcopy(obj))
        mine = evaluate.evaluation(obj)
        assert np.allclose(mine, ref, rtol=0, atol=1e-9, equal_nan=True), (seed, mine, ref)
        exp.append(ref)
        print(f"eval case seed {seed} ({n} images): AP {np.round(ref, 3).tolist()}; evaluate.py == reference")
    np.savez_compressed(os.path.join(HERE, "eval_cases.npz"), seeds=np.array(seeds), sizes=np.array(sizes),
                        ap=np.array(exp, np.float64))


MARGIN_KEYS = ("cand", "order", "iou", "argmax", "hop")


def make_e2e(drn, model, datatest):
    """tests/golden/e2e_d22_384.npz: frames -> people through the REFERENCE pipeline (rt_test.py:87-147: normalise,
    model.forward, the seven head slices, resp*conf, datatest.get_humans_by_feature) for 8 calibrated D-22 frames
    at 384x384, as compact indices (oracle decode of the reference head, cross-checked dict by dict against the
    reference's humans/scores), plus per frame the decision margins of the reference head (oracle/decode_ref.py).
    Also forward_d54_384.npz: sampled reference head of D-54 at 384x384 (BASELINE configs[4] end to end)."""
    arch_name, size, batch, seed_in = "drn_d_22", 384, 8, 4242
    g = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch_name}_seed0.npz"))
    stats = {k: g[k] for k in g.files}
    sd = synth.make_state_dict(arch_name, 0, bn_stats=stats)
    net = build_ref_model(drn, model, arch_name, sd).eval()
    x_u8 = prng.u8_frames(seed_in, batch, (size, size))
    with torch.no_grad():
        feat = net(Fr.normalize_u8(x_u8))                                   # rt_test.py:104
    out = {"arch": arch_name, "size": size, "batch": batch, "seed_in": seed_in, "seed_w": 0}
    K = cfg.K
    for i in range(batch):
        fm = feat[i:i + 1]
        # rt_test.py:106-130, verbatim slicing of the reference driver
        resp = fm[:, 0 * K:1 * K].numpy()[0]
        conf = fm[:, 1 * K:2 * K].numpy()[0]
        x = fm[:, 2 * K:3 * K].numpy()[0]
        y = fm[:, 3 * K:4 * K].numpy()[0]
        w = fm[:, 4 * K:5 * K].numpy()[0]
        h = fm[:, 5 * K:6 * K].numpy()[0]
        e = fm[:, 6 * K:].reshape(1, len(cfg.EDGES), 21, 21, 24, 24).numpy()[0]
        delta = resp * conf
        ref_h, ref_s = datatest.get_humans_by_feature(delta, x, y, w, h, e, detection_thresh=0.15)
        margins = {}
        res = D.decode_ref(feat[i].numpy(), insize=(size, size), margins=margins)
        compare_humans(ref_h, ref_s, res)
        for k in ("n", "root_cell", "kp_cell", "limb_arg", "bbox", "score"):
            out[f"{i}/{k}"] = res[k]
        out[f"{i}/margins"] = np.array([margins.get(k, np.inf) for k in MARGIN_KEYS], np.float64)
        print(f"e2e frame {i}: {len(res['cand'])} candidates, {res['n']} people, "
              f"{int((res['kp_cell'] >= 0).sum())} keypoints; margins "
              + ", ".join(f"{k} {margins.get(k, np.inf):.2e}" for k in MARGIN_KEYS))
    out["margin_keys"] = np.array(MARGIN_KEYS)
    np.savez_compressed(os.path.join(HERE, "e2e_d22_384.npz"), **out)

    # D-54 at 384x384, batch 1: sampled head + per-channel sums of the reference
    arch_name = "drn_d_54"
    g = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch_name}_seed0.npz"))
    sd = synth.make_state_dict(arch_name, 0, bn_stats={k: g[k] for k in g.files})
    net = build_ref_model(drn, model, arch_name, sd).eval()
    x_u8 = prng.u8_frames(777, 1, (384, 384))
    x = Fr.normalize_u8(x_u8)
    with torch.no_grad():
        ref = net(x).numpy()
    mine = Fr.forward_ref(sd, x, arch_name).numpy()
    assert np.abs(mine - ref).max() <= 1e-6
    sd64 = {k: (torch.from_numpy(v).double() if v.dtype != np.int64 else torch.from_numpy(v)) for k, v in sd.items()}
    h64 = Fr.forward_ref(sd64, x.double(), arch_name).numpy()
    n = 40000
    idx = (prng.raw_u64(prng.stream_seed(98, 0), n) % np.uint64(ref.size)).astype(np.int64)
    noise = float(np.abs(ref - h64).max())
    print(f"forward_d54_384: oracle == reference; reference fp32 vs fp64 max|diff| = {noise:.3e}")
    np.savez_compressed(os.path.join(HERE, "forward_d54_384.npz"), arch=arch_name, size=384, batch=1, seed_in=777,
                        seed_w=0, head_idx=idx, head_val=ref.reshape(-1)[idx],
                        head_val_f64=h64.reshape(-1)[idx].astype(np.float32), ref_f32_noise=noise,
                        head_chan_sum=ref.astype(np.float64).sum(axis=(2, 3)))


TUNED_KEYS = ("bn2.weight", "bn2.bias", "conv3.bias")


def make_e2e_tuned(drn, model, datatest):
    """tests/golden/e2e_tuned_d22_384.npz: the same 8 frames -> people through the REFERENCE pipeline, on a checkpoint
    that is less adversarial for reduced-precision modes than the raw random initialisation (where ~490 of 576 cells
    are near-tied root candidates).  Still entirely reference-generated: the imported reference model is fine-tuned
    with the reference's own PPNLoss (main.py:125-216) and torch.optim.Adam (main.py:278) on synthetic targets, but
    ONLY bn2.weight / bn2.bias / conv3.bias (8 629 parameters -- they travel inside the fixture; every other weight
    stays the seed-0 PRNG stream), until the densest frame has fewer than 40 root candidates.  conv2's output is
    computed once (nothing before it is trained) and the tail bn2 -> LeakyReLU -> conv3 -> sigmoid runs through the
    reference's own modules every step."""
    from oracle import targets_ref as T
    main_mod = import_reference_main()
    arch_name, size, batch, seed_in, seed_t = "drn_d_22", 384, 8, 4242, 9100
    g = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch_name}_seed0.npz"))
    sd = synth.make_state_dict(arch_name, 0, bn_stats={k: g[k] for k in g.files})
    net = build_ref_model(drn, model, arch_name, sd).eval()
    x_u8 = prng.u8_frames(seed_in, batch, (size, size))
    x = Fr.normalize_u8(x_u8)
    keep = {}
    hook = net.conv2.register_forward_hook(lambda m, i, o: keep.__setitem__("z", o.detach()))
    with torch.no_grad():
        net(x)
    hook.remove()
    z = keep["z"]
    for p_ in net.parameters():
        p_.requires_grad_(False)
    params = [net.bn2.weight, net.bn2.bias, net.conv3.bias]
    for p_ in params:
        p_.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=0.01)
    crit = main_mod.PPNLoss()
    tg = {k: torch.from_numpy(v) for k, v in T.synthetic_batch(seed_t, batch, insize=(size, size),
                                                               outsize=(size // 16, size // 16)).items()}
    # Synthetic targets for THIS purpose: roots (keypoint 0) only where the synthetic people stand, but every cell is told
    # to hold every other keypoint -- with three per-channel parameter sets the network cannot learn where people are in
    # 30 steps, and pushing all responses down together (the plain targets) leaves root candidates without any accepted
    # hop, i.e. no people at all.  This way the root channel thins out while hops stay accepted.
    tg["delta"][:, 1:] = 1.0
    tg["weight"][:, 1:] = 1.0
    tg["tx_half"][:, 1:] = tg["tx"][:, 1:]
    tg["ty_half"][:, 1:] = tg["ty"][:, 1:]
    image = torch.zeros(batch, 3, size, size)
    K = cfg.K

    def tail():
        return net.sigmoid(net.conv3(net.lRelu(net.bn2(z))))

    def candidates(fm):
        return ((fm[:, 0] * fm[:, K]) > 0.15).flatten(1).sum(1)

    steps = 0
    for steps in range(1, 401):
        fm = tail()
        ls = crit(image, fm, tg["delta"], tg["weight"], tg["weight_ij"], tg["tx_half"], tg["ty_half"], tg["tx"],
                  tg["ty"], tg["tw"], tg["th"], tg["te"])
        loss = ls[0]          # main.py:668-674 with task weights (5, 0, 0, 0, 0): only the response loss (the IoU loss
                              # would drive conf of the all-ones keypoints, whose target boxes are empty, to zero)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if steps % 10 == 0 or steps > 25:
            with torch.no_grad():
                c = candidates(tail())
            print(f"tune step {steps}: losses {[round(float(l.detach()), 3) for l in ls]}, root candidates per frame "
                  f"{c.tolist()}", flush=True)
            if int(c.max()) < 40:                                 # first step at which the densest frame is below 40
                break
    for p_ in params:
        p_.requires_grad_(False)
    over = {k: net.state_dict()[k].detach().numpy().astype(np.float32).copy() for k in TUNED_KEYS}
    sd2 = dict(sd)
    sd2.update(over)
    with torch.no_grad():
        feat = net(x)
    mine = Fr.forward_ref(sd2, x, arch_name).numpy()              # the oracle on (seed-0 weights + overrides)
    assert np.abs(mine - feat.numpy()).max() == 0.0, np.abs(mine - feat.numpy()).max()
    out = {"arch": arch_name, "size": size, "batch": batch, "seed_in": seed_in, "seed_w": 0, "tune_steps": steps,
           "seed_targets": seed_t}
    for k in TUNED_KEYS:
        out["override/" + k] = over[k]
    for i in range(batch):
        fm = feat[i:i + 1]
        resp, conf = fm[:, 0 * K:1 * K].numpy()[0], fm[:, 1 * K:2 * K].numpy()[0]
        xx, yy = fm[:, 2 * K:3 * K].numpy()[0], fm[:, 3 * K:4 * K].numpy()[0]
        ww, hh = fm[:, 4 * K:5 * K].numpy()[0], fm[:, 5 * K:6 * K].numpy()[0]
        e = fm[:, 6 * K:].reshape(1, len(cfg.EDGES), 21, 21, 24, 24).numpy()[0]
        ref_h, ref_s = datatest.get_humans_by_feature(resp * conf, xx, yy, ww, hh, e, detection_thresh=0.15)
        margins = {}
        res = D.decode_ref(feat[i].numpy(), insize=(size, size), margins=margins)
        compare_humans(ref_h, ref_s, res)
        for k in ("n", "root_cell", "kp_cell", "limb_arg", "bbox", "score"):
            out[f"{i}/{k}"] = res[k]
        out[f"{i}/margins"] = np.array([margins.get(k, np.inf) for k in MARGIN_KEYS], np.float64)
        print(f"e2e tuned frame {i}: {len(res['cand'])} candidates, {res['n']} people, "
              f"{int((res['kp_cell'] >= 0).sum())} keypoints; margins "
              + ", ".join(f"{k} {margins.get(k, np.inf):.2e}" for k in MARGIN_KEYS))
    out["margin_keys"] = np.array(MARGIN_KEYS)
    total = sum(int(out[f"{i}/n"]) for i in range(batch))
    assert total >= 20, f"only {total} people left: the fixture would test nothing"
    np.savez_compressed(os.path.join(HERE, "e2e_tuned_d22_384.npz"), **out)


def target_cases():
    """People lists of the target-encoder fixture: synthetic crowds (synth.synthetic_people) plus the edge cases of
    dataset.py:108-152 -- an unlabeled root (w = 0), keypoints left/above the frame (int() truncates towards zero,
    so x in (-16, 0) lands in column 0 with a negative offset), keypoints right/below the frame (dropped), limbs
    longer than the 21x21 window (dropped), an invisible joint, and two people sharing cells (the later one wins)."""
    cases = []
    for seed in (11, 12, 13):
        cases.append([dict(p) for p in synth.synthetic_people(seed)])
    ppl = [dict(p) for p in synth.synthetic_people(14)]
    a = ppl[0]
    a["points"] = a["points"].copy()
    a["points"][0] = (-5.0, 37.5)          # left of the frame, truncation -> column 0, tx = -0.3125
    a["points"][1] = (391.25, 100.0)       # right of the frame -> dropped
    a["points"][2] = (200.0, -3.0)         # above the frame -> row 0, ty = -0.1875
    a["points"][3] = (10.0, 383.9)         # last row
    a["points"][4] = (383.99, 383.99)      # last cell
    a["visible"] = a["visible"].copy()
    a["visible"][5] = False
    twin = dict(a)                          # same cells, other sizes: overwrites a's entries
    twin["size"] = np.float32(19.5)
    twin["bbox"] = (a["bbox"][0], a["bbox"][1], np.float32(0.0), a["bbox"][3])   # w = 0: root unlabeled
    cases.append(ppl + [twin])
    return cases


def make_targets():
    """tests/golden/targets_cases.npz: the ten target tensors of the reference's own
    KeypointsDataset.__getitem__ (dataset.py:70-200) for synthetic annotation files.  The image loader is stubbed
    (skimage.io.imread -> zeros; the encoder never reads pixels), `np.bool` (removed from NumPy, dataset.py:53) is
    aliased to bool, and the IAA augmentation (aug.py:13-134, needs imgaug) is bypassed by a transform that hands
    over the annotated coordinates in the containers IAA produces (keypoints f32 [P,17,2], bbox [[cx,cy,w,h]]) to the
    reference's own aug.ToNormalizedTensor."""
    import json
    import tempfile
    if not hasattr(np, "bool"):
        np.bool = bool                      # dataset.py:53
    sys.modules["skimage.io"].imread = lambda path: np.zeros((384, 384, 3), np.uint8)
    sys.modules["skimage"].io.imread = sys.modules["skimage.io"].imread
    import aug
    import dataset
    from oracle import targets_ref as T
    cases = target_cases()
    annos = []
    for i, people in enumerate(cases):
        for p in people:
            annos.append({"file_name": f"img_{i:03d}.jpg",
                          "keypoints": [[float(x), float(y)] for x, y in np.asarray(p["points"], np.float32)],
                          "bbox": [float(v) for v in p["bbox"]],
                          "is_visible": [int(bool(v)) for v in p["visible"]],
                          "size": float(p["size"])})
    to_tensor = aug.ToNormalizedTensor()

    def bypass_iaa(sample):
        kp = np.asarray(sample["keypoints"], np.float32).reshape(-1, 17, 2)
        bb = [[float(v) for v in b] for b in sample["bbox"]]
        return to_tensor({"image": sample["image"], "keypoints": kp, "bbox": bb, "is_visible": sample["is_visible"],
                          "size": sample["size"]})

    with tempfile.TemporaryDirectory() as td:
        jf = os.path.join(td, "anno.json")
        with open(jf, "w") as f:
            json.dump({"annotations": annos}, f)
        ds = dataset.KeypointsDataset(json_file=jf, root_dir=td, transform=bypass_iaa, insize=(384, 384),
                                      outsize=(24, 24), local_grid_size=(21, 21))
        assert len(ds) == len(cases)
        names = ["delta", "weight", "weight_ij", "tx", "ty", "tx_half", "ty_half", "tw", "th", "te"]   # dataset.py:200
        out = {}
        for i, people in enumerate(cases):
            item = ds[i]
            ref = {n: item[1 + j].numpy() for j, n in enumerate(names)}
            mine = T.encode_targets(people)
            for n in names:
                assert ref[n].dtype == np.float32 and ref[n].shape == mine[n].shape, n
                assert np.array_equal(ref[n], mine[n]), (i, n, np.abs(ref[n] - mine[n]).max())
                out[f"case{i}/{n}"] = ref[n]
            out[f"case{i}/bbox"] = np.array([p["bbox"] for p in people], np.float32)
            out[f"case{i}/points"] = np.stack([np.asarray(p["points"], np.float32) for p in people])
            out[f"case{i}/visible"] = np.stack([np.asarray(p["visible"], bool) for p in people])
            out[f"case{i}/size"] = np.array([p["size"] for p in people], np.float32)
            print(f"targets case {i}: {len(people)} people, delta {int(ref['delta'].sum())} cells, te {int(ref['te'].sum())} "
                  f"limbs, weight_ij==1 {int((ref['weight_ij'] == 1).sum())}; oracle == reference (10 tensors, bitwise)")
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "targets_cases.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    drn, model, datatest = import_reference()
    torch.set_num_threads(8)
    if args.only in (None, "nms"):
        make_nms(datatest)
    if args.only in (None, "decode"):
        make_decode(datatest)
    if args.only in (None, "forward"):
        make_forward(drn, model)
    if args.only in (None, "loss"):
        make_loss()
    if args.only in (None, "train"):
        make_train(drn, model)
    if args.only in (None, "eval"):
        make_eval(datatest)
    if args.only in (None, "targets"):
        make_targets()
    if args.only in (None, "e2e"):
        make_e2e(drn, model, datatest)
    if args.only in (None, "e2e_tuned"):
        make_e2e_tuned(drn, model, datatest)


if __name__ == "__main__":
    main()
