"""GPU: frames -> people, the whole hot path against the REFERENCE pipeline's people lists.

Fixture tests/golden/e2e_d22_384.npz holds what the reference itself returns for 8 calibrated DRN-D-22 frames at
384x384 (/root/reference/rt_test.py:87-147 -> datatest.py:74-132: forward, head slices, resp*conf,
get_humans_by_feature), as compact indices.  These synthetic heads are dense (~490 root candidates and ~30 people per
frame), so some of the reference's own decisions sit closer to their flip point than the 1e-4 head tolerance
(sorted candidate scores 1e-7 apart, IoUs 2e-5 from 0.3): a head that is within tolerance cannot be required to
reproduce those.  The test therefore checks the five relations the decode is a function of -- candidate set, candidate
order, pairwise suppression (IoU >= 0.3), limb arg-max map, hop acceptance map -- between the reference head (the CPU
oracle forward, bit-identical to the reference and re-pinned here by the fixture) and the HIP head:

* f32 mode: every relation entry that differs must be a knife edge of the REFERENCE head (margin below what a 1e-4
  head perturbation can move); if none differs the people lists must be identical to the fixture;
* both modes: the share of reference people reproduced exactly (root cell, every keypoint cell, every limb arg-max)
  is printed and gated; bench.py reports the bf16 number in its JSON line (`bf16_agreement`)."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref as D, forward_ref as Fr
from pytorch_pose_proposal_network_amd import prng, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THR, NMS_THR = np.float32(0.15), np.float32(0.3)
# what a head perturbation of 1e-4 per element can move: delta = resp*conf (two factors <= 1) by 2e-4; a limb
# value by 1e-4 (gap between two of them 2e-4); a box edge by (16 + 192) * 1e-4 px, i.e. the IoU of >= 8 px boxes
# by < 3e-3
TOL_DELTA, TOL_GAP, TOL_IOU = 3e-4, 3e-4, 3e-3


def _relations(head):
    """The decode's decision inputs for one head f32 [7605,24,24] (same f32 arithmetic as oracle/decode_ref.py)."""
    delta, x, y, w, h, e = D.split_head(head)
    bbox = D.build_bbox(x, y, w, h)[0].reshape(-1, 4)                      # root boxes, [576, 4]
    d0 = delta[0].reshape(-1)
    area = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
    tl = np.maximum(bbox[:, None, :2], bbox[None, :, :2])
    br = np.minimum(bbox[:, None, 2:], bbox[None, :, 2:])
    inter = ((br[..., 0] - tl[..., 0]) * (br[..., 1] - tl[..., 1]) * ((tl < br).all(-1))).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / ((area[:, None] + area[None, :]) - inter)
    win = e.reshape(D.E, -1, 576)                                          # [E, 441, cells]
    arg = win.argmax(axis=1)
    top2 = np.partition(win.astype(np.float64), -2, axis=1)[:, -2:]
    return dict(delta=delta.reshape(D.K, -1), d0=d0, cand=d0 > THR, accept=~(delta.reshape(D.K, -1) < THR), iou=iou,
                arg=arg, gap=top2[:, 1] - top2[:, 0])


def _flips(ref, got):
    """Differences between the decision relations of the reference head and another head, with the REFERENCE margin
    of each: {kind: array of margins}."""
    out = {}
    c = ref["cand"] != got["cand"]
    out["cand"] = np.abs(ref["d0"][c].astype(np.float64) - float(THR))
    both = np.where(ref["cand"] & got["cand"])[0]
    sr, sg = ref["d0"][both].astype(np.float64), got["d0"][both].astype(np.float64)
    inv = ((sr[:, None] - sr[None, :]) * (sg[:, None] - sg[None, :])) < 0
    out["order"] = np.abs(sr[:, None] - sr[None, :])[inv]
    union = np.where(ref["cand"] | got["cand"])[0]
    ir, ig = ref["iou"][np.ix_(union, union)], got["iou"][np.ix_(union, union)]
    s = (ir >= NMS_THR) != (ig >= NMS_THR)
    out["iou"] = np.abs(ir[s].astype(np.float64) - float(NMS_THR))
    a = ref["arg"] != got["arg"]
    out["argmax"] = ref["gap"][a]
    h = ref["accept"] != got["accept"]
    out["hop"] = np.abs(ref["delta"][h].astype(np.float64) - float(THR))
    return out


def _people(res):
    """{root cell: (kp_cell row, limb_arg row, bbox rows, score row)} of one image's compact result."""
    return {int(res["kp_cell"][i, 0]): (res["kp_cell"][i], res["limb_arg"][i], res["bbox"][i], res["score"][i])
            for i in range(res["n"])}


def _agreement(exp, got):
    from pytorch_pose_proposal_network_amd import decode
    return decode.people_agreement(exp, got)


FIXTURES = ["e2e_d22_384", "e2e_tuned_d22_384"]


def _setup(dtype, fixture="e2e_d22_384"):
    """`e2e_tuned_d22_384`: the same frames and seed-0 weights, except bn2.weight / bn2.bias / conv3.bias, which the
    REFERENCE fine-tuned with its own PPNLoss + Adam until the densest frame has < 40 root candidates (values inside the
    fixture, tests/golden/make_golden.py::make_e2e_tuned): 19-39 candidates and 5-13 people per frame instead of ~490 / ~32."""
    from pytorch_pose_proposal_network_amd import drn, model
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    arch, size, batch = str(g["arch"]), int(g["size"]), int(g["batch"])
    st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch}_seed0.npz"))
    sd = synth.make_state_dict(arch, int(g["seed_w"]), bn_stats={k: st[k] for k in st.files})
    for k in g.files:
        if k.startswith("override/"):
            sd[k[len("override/"):]] = g[k]
    net = model.PoseProposalNet(getattr(drn, arch)(), insize=(size, size), outsize=(size // 16, size // 16),
                                compute_dtype=dtype).cuda()
    net.load_state_dict(sd)
    u8 = prng.u8_frames(int(g["seed_in"]), batch, (size, size))
    exp = [{k: g[f"{i}/{k}"] for k in ("n", "root_cell", "kp_cell", "limb_arg", "bbox", "score")} for i in range(batch)]
    for e_ in exp:
        e_["n"] = int(e_["n"])
    return g, sd, net.eval(), u8, exp, arch


@pytest.mark.parametrize("fixture", FIXTURES)
def test_f32_pipeline_reproduces_reference_people_up_to_knife_edges(fixture):
    from pytorch_pose_proposal_network_amd import decode, rt
    g, sd, net, u8, exp, arch = _setup("float32", fixture)
    frames = torch.from_numpy(u8).cuda()
    torch.set_num_threads(min(16, os.cpu_count() or 1))   # a 1-GPU box owns a 16-core share
    ref_head = Fr.forward_ref(sd, Fr.normalize_u8(u8), arch).numpy()     # == the reference's head (make_golden.py)
    hip_head = net.forward_u8(frames).cpu().numpy()
    err = float(np.abs(hip_head - ref_head).max())
    print(f"f32 head: max|hip - reference| = {err:.3e}")
    assert err <= 1e-4
    got_fused = rt.inference_batch(frames, net).to_host()                 # the benchmarked path (no head tensor)
    got = decode.decode_heads(net.forward_u8(frames)).to_host()
    tot = dict(people=0, exact=0, flips=0)
    for i in range(len(exp)):
        # the fixture is what the oracle decodes from the oracle head: the CPU stand-in IS the reference pipeline
        mine = D.decode_ref(ref_head[i])
        assert mine["n"] == exp[i]["n"] and np.array_equal(mine["kp_cell"], exp[i]["kp_cell"])
        assert np.array_equal(mine["limb_arg"], exp[i]["limb_arg"]) and np.array_equal(mine["bbox"], exp[i]["bbox"])
        # fused path == stand-alone decode of the materialised head, bit for bit
        assert got_fused[i]["n"] == got[i]["n"]
        for k in ("kp_cell", "limb_arg", "bbox", "score"):
            assert np.array_equal(got_fused[i][k], got[i][k]), k
        fl = _flips(_relations(ref_head[i]), _relations(hip_head[i]))
        nfl = sum(len(v) for v in fl.values())
        for kind, tol in (("cand", TOL_DELTA), ("order", 2 * TOL_DELTA), ("iou", TOL_IOU), ("argmax", TOL_GAP),
                          ("hop", TOL_DELTA)):
            assert len(fl[kind]) == 0 or float(fl[kind].max()) < tol, \
                f"frame {i}: a {kind} decision flipped although the reference margin is {float(fl[kind].max()):.2e}"
        n, exact, same_root, kp_eq, kp_all = _agreement(exp[i], got[i])
        tot["people"] += n; tot["exact"] += exact; tot["flips"] += nfl
        print(f"frame {i}: {n} reference people, {exact} reproduced exactly, {same_root} same root, keypoint cells "
              f"{kp_eq}/{kp_all}; knife-edge flips: " + ", ".join(f"{k} {len(v)}" for k, v in fl.items()))
        if nfl == 0:                                                      # no decision differs -> identical people
            pe, pg = _people(exp[i]), _people(got[i])
            assert pe.keys() == pg.keys() and exp[i]["n"] == got[i]["n"]
            assert [int(c) for c in exp[i]["kp_cell"][:, 0]] == [int(c) for c in got[i]["kp_cell"][:, 0]]
            for r in pe:
                assert np.array_equal(pe[r][0], pg[r][0]) and np.array_equal(pe[r][1], pg[r][1])
                assert np.abs(pe[r][2] - pg[r][2]).max() <= 384 * 2e-4 and np.abs(pe[r][3] - pg[r][3]).max() <= 2e-4
    print(f"f32 mode: {tot['exact']}/{tot['people']} reference people reproduced exactly; {tot['flips']} relation "
          f"entries differ, all on knife edges of the reference head")
    assert tot["exact"] >= 0.97 * tot["people"]


@pytest.mark.parametrize("fixture", FIXTURES)
@pytest.mark.parametrize("mode", ["bfloat16", "float16"])
def test_bf16_pipeline_agreement_with_reference_people(mode, fixture):
    """bf16 is the benchmarked mode: how many of the reference's people it returns (stated, gated, and repeated in
    bench.py's JSON line); f16 (same MFMA rate, 3 more mantissa bits) beside it."""
    from pytorch_pose_proposal_network_amd import rt
    g, sd, net, u8, exp, arch = _setup(mode, fixture)
    got = rt.inference_batch(torch.from_numpy(u8).cuda(), net).to_host()
    tot = np.zeros(5, np.int64)
    for i in range(len(exp)):
        tot += np.array(_agreement(exp[i], got[i]))
        print(f"frame {i}: reference {exp[i]['n']} people, {mode} {got[i]['n']}")
    n, exact, same_root, kp_eq, kp_all = (int(v) for v in tot)
    print(f"{fixture}: {mode} mode vs reference people: {exact}/{n} exact ({exact / n:.3f}), same root {same_root}/{n} "
          f"({same_root / n:.3f}), keypoint cells among same-root people {kp_eq}/{kp_all} ({kp_eq / max(kp_all, 1):.3f})")
    min_root, min_kp, min_exact = GATES[(fixture, mode)]
    assert same_root >= min_root * n and kp_eq >= min_kp * kp_all and exact >= min_exact * n


@pytest.mark.parametrize("fixture", FIXTURES)
def test_ap_of_each_mode_against_reference_people(fixture):
    """What a reduced-precision mode costs in the TASK metric (BASELINE metric "PCKh@0.5 vs ref"): the reference
    pipeline's people taken as ground truth (keypoint = box centre, head box = instance box), the HIP pipeline's people
    scored with the reference's own matcher and metric (evaluate.evaluation == datatest.evaluation,
    /root/reference/datatest.py:278-369, eval_helpers.py:300-468).  On these dense synthetic crowds the metric's ceiling
    -- the reference people scored against themselves -- is below 100 (overlapping people tie in assignGTmulti); f32
    must reach that ceiling up to the knife edges, bf16 is gated just under its measured value."""
    from pytorch_pose_proposal_network_amd import evaluate, rt
    names = ["head", "shoulder", "elbow", "wrist", "hip", "knee", "ankle", "total"]
    aps = {}
    for mode in ("float32", "bfloat16", "float16"):
        g, sd, net, u8, exp, arch = _setup(mode, fixture)
        got = rt.inference_batch(torch.from_numpy(u8).cuda(), net).to_host()
        aps[mode] = np.array(evaluate.ap_against_people(exp, got))
        if "self" not in aps:
            aps["self"] = np.array(evaluate.ap_against_people(exp, exp))
    for k, v in aps.items():
        print(f"{fixture}: AP vs reference people, {k:9s}: " + ", ".join(f"{n} {x:.2f}" for n, x in zip(names, v)))
    assert np.all(np.abs(aps["float32"] - aps["self"]) <= AP_F32_MAX_GAP), (aps["float32"], aps["self"])
    max_bf16, max_f16 = AP_MAX_LOSS[fixture]
    assert aps["bfloat16"][-1] >= aps["self"][-1] - max_bf16, (aps["bfloat16"], aps["self"])
    assert aps["float16"][-1] >= aps["self"][-1] - max_f16, (aps["float16"], aps["self"])


# AP gates (see the test above): f32 within 1 point of the ceiling per joint group (knife edges move single people);
# bf16 total AP at most this far below the ceiling (measured on MI355X, round 3: see profiles/README.md)
AP_F32_MAX_GAP = 1.0
AP_MAX_LOSS = {"e2e_d22_384": (45.0, 10.0),          # measured losses of total AP: bf16 41.2, f16 7.6
               "e2e_tuned_d22_384": (50.0, 11.0)}    # ceiling 92.2: bf16 46.6 (loss 45.6), f16 83.8 (loss 8.4)

# bf16 gates: measured on MI355X (see profiles/README.md, round 2), set just below the measurement.  The synthetic
# checkpoint is a randomly initialised network: ~490 of 576 cells are root candidates with near-equal scores, so which
# of two overlapping roots survives NMS is decided by differences far below bf16 resolution.
BF16_MIN_SAME_ROOT, BF16_MIN_KP = 0.62, 0.92            # measured 0.650 / 0.937 (95 of 260 people exact)
# f16 (same MFMA rate, 11 significant bits): measured 0.969 / 0.9958, 233 of 260 people exact
F16_MIN_SAME_ROOT, F16_MIN_KP, F16_MIN_EXACT = 0.95, 0.99, 0.85
# (same root, keypoint cells among same-root people, reproduced exactly) as fractions of the reference people
GATES = {("e2e_d22_384", "bfloat16"): (BF16_MIN_SAME_ROOT, BF16_MIN_KP, 0.33),
         ("e2e_d22_384", "float16"): (F16_MIN_SAME_ROOT, F16_MIN_KP, F16_MIN_EXACT),
         ("e2e_tuned_d22_384", "bfloat16"): (0.70, 0.92, 0.40),      # measured 0.737 / 0.938 / 0.434 (33 of 76)
         ("e2e_tuned_d22_384", "float16"): (0.94, 0.99, 0.88)}       # measured 0.961 / 0.997 / 0.921 (70 of 76)


def test_d54_384_f32_head_vs_reference(golden_dir):
    """BASELINE configs[4]'s network at full resolution: DRN-D-54 (Bottleneck trunk) 384x384 f32 head on sampled
    positions vs the reference head.  The reference's own fp32 result is `ref_f32_noise` (4e-4) away from an fp64
    evaluation of the same network, so -- as for D-54 at 96x96 -- the HIP head must be within 1e-4 of the reference
    OR at least as close to the fp64 evaluation as 1.5x the reference itself is (a stated north_star deviation)."""
    from pytorch_pose_proposal_network_amd import drn, model
    g = np.load(os.path.join(golden_dir, "forward_d54_384.npz"))
    st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", "bn_calib_drn_d_54_seed0.npz"))
    net = model.PoseProposalNet(drn.drn_d_54(), compute_dtype="float32").cuda()
    net.load_state_dict(synth.make_state_dict("drn_d_54", 0, bn_stats={k: st[k] for k in st.files}))
    u8 = prng.u8_frames(int(g["seed_in"]), 1, (384, 384))
    head = net.forward_u8(torch.from_numpy(u8).cuda()).cpu().numpy()
    assert head.shape == (1, 7605, 24, 24)
    v = head.reshape(-1)[g["head_idx"]]
    err, err64, noise = np.abs(v - g["head_val"]).max(), np.abs(v - g["head_val_f64"]).max(), float(g["ref_f32_noise"])
    print(f"D-54 @384 f32: |hip-ref| {err:.3e}  |hip-f64| {err64:.3e}  |ref-f64| {noise:.3e}")
    assert err <= 1e-4 or err64 <= 1.5 * noise, (err, err64, noise)
    assert np.allclose(head.astype(np.float64).sum(axis=(2, 3)), g["head_chan_sum"], atol=5e-2)
