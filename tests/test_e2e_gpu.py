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


def _emulated(fixture, mode):
    """What the EMULATED-STORAGE ORACLE (oracle/fused_ref.py with this mode's roundings, torch-CPU, f32 accumulation) returns
    on this fixture: tests/golden/e2e_emulated.npz, written by tests/golden/make_emulated.py and re-checked on the CPU by
    tests/test_oracle.py.  Keys: agreement (people, exact, same root, kp equal, kp compared), ap[8], ap_self[8], head_err."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "e2e_emulated.npz"))
    return {k: z[f"{fixture}/{mode}/{k}"] for k in ("agreement", "ap", "head_err")} | {"ap_self": z[f"{fixture}/ap_self"]}


# ---- 16-bit modes: gates DERIVED from the requirement, not from this implementation's past measurements ---------------
# A 16-bit storage policy cannot meet north_star's 1e-4 / bit-exact tolerance by construction (bf16 keeps 8 significant
# bits); what a 16-bit KERNEL can be required to do is to be no worse than a correct implementation of the same policy.
# That implementation is the emulated-storage oracle; the HIP pipeline may fall short of it only by what two correct
# implementations differ by -- different f32 summation orders flip single 16-bit roundings, which this chaotic random
# network then amplifies like any other storage noise:
#   people reproduced exactly / same root   >= 85 % of the oracle's own count
#   keypoint-cell agreement                 >= the oracle's fraction - 0.02
#   total AP against the reference people   >= the oracle's AP - (15 % of the AP the policy itself loses + 2 points;
#                                              one person on these 8 frames moves the total by ~0.4-1.3 points)
PEOPLE_SHARE, KP_SLACK, AP_LOSS_SHARE, AP_SLACK = 0.85, 0.02, 0.15, 2.0


@pytest.mark.parametrize("fixture", FIXTURES)
@pytest.mark.parametrize("mode", ["bfloat16", "float16"])
def test_16bit_pipeline_no_worse_than_emulated_oracle(mode, fixture):
    """bf16 is BASELINE configs[1]'s dtype (the benchmarked mode), f16 runs at the same MFMA rate with 3 more mantissa
    bits: people and AP of each HIP pipeline against the reference pipeline's people, gated relative to what the
    emulated-storage oracle of the same mode achieves (rule above); bench.py prints the same numbers."""
    from pytorch_pose_proposal_network_amd import evaluate, rt
    g, sd, net, u8, exp, arch = _setup(mode, fixture)
    got = rt.inference_batch(torch.from_numpy(u8).cuda(), net).to_host()
    tot = np.zeros(5, np.int64)
    for i in range(len(exp)):
        tot += np.array(_agreement(exp[i], got[i]))
    n, exact, same_root, kp_eq, kp_all = (int(v) for v in tot)
    emu = _emulated(fixture, mode)
    en, eexact, esame, ekp, ekpall = (int(v) for v in emu["agreement"])
    ap = np.array(evaluate.ap_against_people(exp, got))
    ceiling, eap = float(emu["ap_self"][-1]), float(emu["ap"][-1])
    ap_floor = eap - (AP_LOSS_SHARE * (ceiling - eap) + AP_SLACK)
    print(f"{fixture} {mode}: HIP vs reference people: exact {exact}/{n}, same root {same_root}/{n}, keypoint cells "
          f"{kp_eq}/{kp_all} ({kp_eq / max(kp_all, 1):.3f}), total AP {ap[-1]:.2f} | emulated oracle: exact {eexact}/{en}, "
          f"same root {esame}, cells {ekp}/{ekpall} ({ekp / ekpall:.3f}), AP {eap:.2f} (ceiling {ceiling:.2f}) -> AP floor {ap_floor:.2f}")
    assert n == en
    assert exact >= PEOPLE_SHARE * eexact and same_root >= PEOPLE_SHARE * esame
    assert kp_eq / max(kp_all, 1) >= ekp / ekpall - KP_SLACK
    assert ap[-1] >= ap_floor, (ap, eap, ceiling)


@pytest.mark.parametrize("fixture", FIXTURES)
def test_f32_ap_reaches_the_ceiling(fixture):
    """The TASK metric of the parity mode (BASELINE metric "PCKh@0.5 vs ref"): the reference pipeline's people taken as
    ground truth (keypoint = box centre, head box = instance box), the HIP f32 pipeline's people scored with the
    reference's own matcher and metric (evaluate.evaluation == datatest.evaluation, /root/reference/datatest.py:278-369,
    eval_helpers.py:300-468).  On these dense synthetic crowds the metric's ceiling -- the reference people scored
    against themselves -- is below 100 (overlapping people tie in assignGTmulti); f32 must reach that ceiling up to the
    knife edges (one person moves a joint group's AP by <= 1 point)."""
    from pytorch_pose_proposal_network_amd import evaluate, rt
    g, sd, net, u8, exp, arch = _setup("float32", fixture)
    got = rt.inference_batch(torch.from_numpy(u8).cuda(), net).to_host()
    ap, ceiling = np.array(evaluate.ap_against_people(exp, got)), np.array(evaluate.ap_against_people(exp, exp))
    print(f"{fixture}: AP vs reference people, f32 {np.round(ap, 2).tolist()} ceiling {np.round(ceiling, 2).tolist()}")
    assert np.all(np.abs(ap - ceiling) <= AP_F32_MAX_GAP), (ap, ceiling)


AP_F32_MAX_GAP = 1.0


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
