"""GPU parity: HIP decode / NMS / limb parse (through the C ABI) vs the oracle and the golden
fixtures generated from the reference.  Bit-exact on every index, bbox and score."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref as D
from pytorch_pose_proposal_network_amd import config as cfg, prng, synth

pytestmark = pytest.mark.gpu

from test_oracle import make_head  # noqa: E402  (same seeded head generators as the CPU tests)


def _decode_mod():
    from pytorch_pose_proposal_network_amd import decode
    return decode


def _assert_same(res, exp, tag):
    assert res["n"] == int(exp["n"]), (tag, res["n"], int(exp["n"]))
    for k in ("root_cell", "kp_cell", "limb_arg", "bbox", "score"):
        assert np.array_equal(res[k], exp[k]), (tag, k)


def test_golden_heads_batch(golden_dir):
    g = np.load(os.path.join(golden_dir, "decode_heads.npz"))
    n = int(g["count"])
    heads = np.stack([make_head(str(g[f"{i}/kind"]), int(g[f"{i}/seed"])) for i in range(n)])
    dec = _decode_mod()
    out = dec.decode_heads(torch.from_numpy(heads).cuda()).to_host()
    for i in range(n):
        exp = {k: g[f"{i}/{k}"] for k in ("n", "root_cell", "kp_cell", "limb_arg", "bbox", "score")}
        _assert_same(out[i], exp, i)


def test_limb_argmax_dense_matches_numpy():
    dec = _decode_mod()
    heads = np.stack([make_head("random", 200 + i) for i in range(3)])
    d = dec.Decoder(3)
    am = d.limb_argmax(torch.from_numpy(heads).cuda()).cpu().numpy()
    for i in range(3):
        assert np.array_equal(am[i], D.limb_argmax_dense(heads[i]))


def test_argmax_ties_take_lowest_index():
    dec = _decode_mod()
    head = np.zeros((2, cfg.lastsize(), 24, 24), np.float32)
    head[:, 108:] = 0.25
    head[0, 108 + 441 * 3 + 17, 5, 6] = 0.9
    head[0, 108 + 441 * 3 + 400, 5, 6] = 0.9      # tie across different row slices
    head[1, 108 + 440, 0, 0] = 0.5                 # last channel of edge 0
    d = dec.Decoder(2)
    am = d.limb_argmax(torch.from_numpy(head).cuda()).cpu().numpy()
    assert am[0, 3, 5, 6] == 17
    assert am[1, 0, 0, 0] == 440
    assert am[0, 0, 0, 0] == 0                     # all equal -> first
    assert np.array_equal(am[0], D.limb_argmax_dense(head[0]))


def test_saturated_head_full_candidate_list():
    """Degenerate head (every cell a candidate, saturated values): n = 576 candidates, ties in e."""
    dec = _decode_mod()
    C = cfg.lastsize()
    u = prng.uniform01(prng.stream_seed(77, 0), C * 576).reshape(C, 24, 24)
    head = (u > 0.5).astype(np.float32)            # exactly 0.0 / 1.0 like an uncalibrated net
    # distinct root scores so that the NMS order is defined
    sc = 0.2 + 0.8 * (np.random.RandomState(0).permutation(576).astype(np.float32) + 1) / 577.0
    head[0] = sc.reshape(24, 24)
    head[18] = 1.0
    out = dec.decode_heads(torch.from_numpy(head[None]).cuda()).to_host()[0]
    exp = D.decode_ref(head)
    assert len(exp["cand"]) == 576
    _assert_same(out, exp, "saturated")


def test_empty_and_threshold_edges():
    dec = _decode_mod()
    C = cfg.lastsize()
    head = np.full((2, C, 24, 24), 0.1, np.float32)   # delta = 0.01: no candidates anywhere
    # image 1: one root whose delta is exactly 0.15 (not > thr) and one just above
    head[1, 0, 2, 2], head[1, 18, 2, 2] = np.float32(0.15), np.float32(1.0)
    head[1, 0, 9, 9], head[1, 18, 9, 9] = np.nextafter(np.float32(0.15), np.float32(1)), np.float32(1.0)
    # limb 0 of the second root points to a target whose delta == thr exactly (passes: `<` is strict)
    head[1, 108 + 10 * 21 + 12, 9, 9] = 0.9           # dh=0, dw=+2
    head[1, 15, 9, 11], head[1, 18 + 15, 9, 11] = np.float32(0.15), np.float32(1.0)
    res = dec.decode_heads(torch.from_numpy(head).cuda()).to_host()
    assert res[0]["n"] == 0
    exp = D.decode_ref(head[1])
    assert exp["n"] == 1 and exp["kp_cell"][0, 15] == 9 * 24 + 11
    _assert_same(res[1], exp, "edges")
    # batch 0 is a no-op
    z = dec.decode_heads(torch.zeros(0, C, 24, 24, device="cuda"))
    assert z.count.numel() == 0


def test_batch32_crowd_vs_oracle():
    """BASELINE config 5 size: 32 planted-crowd heads decoded in one call, every image checked."""
    dec = _decode_mod()
    heads = np.stack([synth.planted_crowd_head(7 + i) for i in range(32)])
    out = dec.decode_heads(torch.from_numpy(heads).cuda()).to_host()
    total = 0
    for i in range(32):
        exp = D.decode_ref(heads[i])
        _assert_same(out[i], exp, i)
        total += exp["n"]
    assert total > 32 * 8


def test_other_grid_sizes():
    """96x96 input -> 6x6 grid (scalar tail path) and a 320x320 -> 20x20 grid."""
    dec = _decode_mod()
    for hw, ins in (((6, 6), (96, 96)), ((20, 20), (320, 320)), ((5, 7), (80, 112))):
        H, W = hw
        C = cfg.lastsize()
        u = prng.uniform01(prng.stream_seed(31 + H, 0), C * H * W).reshape(C, H, W).astype(np.float32)
        u[0:36] = prng.uniform(prng.stream_seed(31 + H, 1), 36 * H * W, 0.3, 1.0).reshape(36, H, W)
        u[72:108] *= 0.3
        out = dec.decode_heads(torch.from_numpy(u[None]).cuda(), insize_hw=ins).to_host()[0]
        exp = D.decode_ref(u, insize=(ins[1], ins[0]))
        _assert_same(out, exp, hw)


def test_reference_shaped_get_humans_by_feature():
    dec = _decode_mod()
    head = synth.planted_crowd_head(42)
    delta, x, y, w, h, e = D.split_head(head)
    humans, scores = dec.get_humans_by_feature(delta, x, y, w, h, e, detection_thresh=0.15)
    exp_h, exp_s = D.humans_from_compact(D.decode_ref(head))
    assert len(humans) == len(exp_h) > 0
    for a, b, sa, sb in zip(humans, exp_h, scores, exp_s):
        assert sorted(a) == sorted(b)
        for k in a:
            assert np.array_equal(a[k], b[k]) and sa[k] == sb[k]


def test_nms_golden(golden_dir):
    dec = _decode_mod()
    g = np.load(os.path.join(golden_dir, "nms_cases.npz"))
    for i in range(int(g["count"])):
        sel = dec.non_maximum_suppression(g[f"{i}/bbox"], 0.3, g[f"{i}/score"])
        assert sel.dtype == np.int32 and np.array_equal(sel, g[f"{i}/sel"]), i
        one = dec.non_maximum_suppression(g[f"{i}/bbox"], 0.3, g[f"{i}/score"], limit=1)
        assert np.array_equal(one, D.nms_ref(g[f"{i}/bbox"], 0.3, g[f"{i}/score"], limit=1))
    assert np.array_equal(dec.non_maximum_suppression(g["noscore/bbox"], 0.3), g["noscore/sel"])
    assert dec.non_maximum_suppression(np.zeros((0, 4), np.float32), 0.3).shape == (0,)
    # negative and mixed-sign scores keep the descending order
    bb = g["0/bbox"]
    sc = np.array([-0.5, 0.25, -0.1, 0.0, 0.9, -2.0, 0.3], np.float32)
    assert np.array_equal(dec.non_maximum_suppression(bb, 0.3, sc), D.nms_ref(bb, 0.3, sc))


def test_device_decode_feeds_ap_evaluation():
    """SURVEY 8f-3 / datatest.py:298-327 -> evaluation: the device DecodeResult (to_humans()) goes through
    evaluate.evaluation unchanged.  Ground truth = the planted people (centres of the oracle-decoded boxes of every
    second person, slightly shifted), so AP is neither 0 nor trivially 1; the 8 AP values must equal those obtained
    from the oracle's decode of the same heads (and the evaluator itself is pinned to the reference by
    tests/test_oracle.py::test_ap_evaluation_golden)."""
    from pytorch_pose_proposal_network_amd import decode, evaluate
    heads = np.stack([synth.planted_crowd_head(40 + i) for i in range(6)])
    res = decode.decode_heads(torch.from_numpy(heads).cuda())
    dev_people = res.to_humans()
    obj_dev, obj_ref = [[] for _ in range(7)], [[] for _ in range(7)]
    for i in range(len(heads)):
        exp = D.decode_ref(heads[i])
        ref_h, ref_s = D.humans_from_compact(exp)
        gt_kps, gt_boxes, vis, sizes = [], [], [], []
        for p in range(0, exp["n"], 2):
            bb = exp["bbox"][p]                                            # [K,4] (ymin,xmin,ymax,xmax)
            pts = np.stack([(bb[1:, 1] + bb[1:, 3]) / 2 + 1.5, (bb[1:, 0] + bb[1:, 2]) / 2 - 1.0], 1).astype(np.float32)
            gt_kps.append(pts)
            cy, cx = (bb[0, 0] + bb[0, 2]) / 2, (bb[0, 1] + bb[0, 3]) / 2
            gt_boxes.append((np.float32(cx), np.float32(cy), np.float32(bb[0, 3] - bb[0, 1]), np.float32(bb[0, 2] - bb[0, 0])))
            vis.append(np.ones(17, bool))
            sizes.append(np.float32(12.0))
        for obj, (hm, sc) in ((obj_dev, dev_people[i]), (obj_ref, (ref_h, ref_s))):
            obj[0].append(f"img_{i}.jpg")
            obj[1].append(np.stack(gt_kps) if gt_kps else np.zeros((0, 17, 2), np.float32))
            obj[2].append(hm); obj[3].append(sc); obj[4].append(gt_boxes); obj[5].append(vis); obj[6].append(sizes)
    ap_dev = np.asarray(evaluate.evaluation(obj_dev), np.float64)
    ap_ref = np.asarray(evaluate.evaluation(obj_ref), np.float64)
    print("AP from the device decode:", np.round(ap_dev, 3).tolist())
    assert np.array_equal(ap_dev, ap_ref, equal_nan=True)
    assert np.isfinite(ap_dev[-1]) and 0.0 < ap_dev[-1] <= 100.0


def test_early_root_nms_candidate_cap_boundary():
    """The stand-alone decode runs candidates + root NMS inside the arg-max launch for images with <= 128 root
    candidates and inside the parse kernel beyond that (csrc/decode.hip early_root_nms).  One batch mixing images with
    0, 1, 127, 128, 129, 200 and 576 candidates (random heads whose root delta is pushed under the threshold except in
    the chosen cells): every image equals the NumPy oracle whichever kernel ran its NMS."""
    dec = _decode_mod()
    counts = [0, 1, 127, 128, 129, 200, 576, 64]
    heads = []
    for i, n in enumerate(counts):
        h = make_head("random", 700 + i)
        g = np.random.default_rng(900 + i)
        cells = g.permutation(576)[:n]
        resp = np.full(576, 0.05, np.float32)                    # delta = resp * conf stays far below 0.15
        resp[cells] = g.uniform(0.6, 1.0, n).astype(np.float32)   # distinct scores
        h[0] = resp.reshape(24, 24)
        h[18] = np.float32(0.9)                                   # conf of the root keypoint
        heads.append(h)
    heads = np.stack(heads)
    out = dec.decode_heads(torch.from_numpy(heads).cuda()).to_host()
    for i, n in enumerate(counts):
        exp = D.decode_ref(heads[i])
        assert len(exp["cand"]) == n, (i, len(exp["cand"]), n)
        _assert_same(out[i], exp, f"{n} candidates")


def _unary_and_keys(heads):
    """What the fused head conv leaves for Decoder.decode_fused: the 6K unary channels and one u64 key per (image, edge,
    cell) = sigmoid value bits << 32 | ~(first arg-max index of the 21x21 window)."""
    h = torch.from_numpy(heads).cuda()
    B = h.shape[0]
    e = h[:, 6 * cfg.K:].reshape(B, len(cfg.EDGES), -1, h.shape[2], h.shape[3])
    val, _ = e.max(dim=2)
    first = (e == val.unsqueeze(2)).float().argmax(dim=2)                              # lowest index among ties
    keys = (val.contiguous().view(torch.int32).to(torch.int64) << 32) | (0xFFFFFFFF - first)
    return h[:, :6 * cfg.K].contiguous(), keys.contiguous()


def test_fused_decode_spread_root_nms_candidate_counts():
    """Fused path (round 4): the pairwise-IoU bit matrix of an image's root candidates is computed by 8 workgroups per image
    in front of the parse kernel (csrc/decode.hip root_mask_kernel) when the image has >= 128 candidates, by the parse
    workgroup itself below that.  One batch mixing 0, 1, 64, 127, 128, 129, 200, 490 and 576 candidates: every image equals
    the NumPy oracle (datatest.py:74-160 restated) AND the single-kernel entry point, bit for bit, whichever kernel built
    its matrix."""
    import ctypes as C
    from pytorch_pose_proposal_network_amd import lib as L
    dec = _decode_mod()
    counts = [0, 1, 127, 128, 129, 200, 576, 64, 490]
    heads = []
    for i, n in enumerate(counts):
        h = make_head("random", 700 + i)
        g = np.random.default_rng(900 + i)
        cells = g.permutation(576)[:n]
        resp = np.full(576, 0.05, np.float32)
        resp[cells] = g.uniform(0.6, 1.0, n).astype(np.float32)
        h[0] = resp.reshape(24, 24)
        h[18] = np.float32(0.9)
        heads.append(h)
    heads = np.stack(heads)
    unary, keys = _unary_and_keys(heads)
    d = dec.Decoder(len(counts))
    got = d.decode_fused(unary, keys).to_host()
    got = [{k: (v.copy() if hasattr(v, "copy") else v) for k, v in r.items()} for r in got]
    for i, n in enumerate(counts):
        exp = D.decode_ref(heads[i])
        assert len(exp["cand"]) == n, (i, len(exp["cand"]), n)
        _assert_same(got[i], exp, f"fused, {n} candidates")
    # the single-kernel entry point (no workspace): same people
    o, c = d.out, d.cfg
    L.check(d.lib.ppn_decode_fused(C.byref(c), unary.data_ptr(), keys.data_ptr(), len(counts), o.count.data_ptr(),
                                   o.kp_cell.data_ptr(), o.limb_arg.data_ptr(), o.bbox.data_ptr(), o.score.data_ptr(),
                                   L.current_stream_ptr()), "ppn_decode_fused")
    one = o.to_host()
    for i in range(len(counts)):
        assert one[i]["n"] == got[i]["n"]
        for k in ("kp_cell", "limb_arg", "bbox", "score"):
            assert np.array_equal(one[i][k], got[i][k]), (i, k)
