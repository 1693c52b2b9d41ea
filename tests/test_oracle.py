"""CPU: the oracle (oracle/*.py) replays every golden fixture produced from the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref as D, forward_ref as Fr
from pytorch_pose_proposal_network_amd import arch as A, config as cfg, prng, synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_skeleton_tables_match_reference_values():
    # resolved values probed from the reference config.py
    assert cfg.EDGES == D.EDGES
    assert cfg.DIRECTED_GRAPHS == D.DIRECTED_GRAPHS
    assert cfg.lastsize() == 7605 and cfg.K == 18 and cfg.E == 17
    src, dst, order = cfg.tree_tables()
    assert order == D.tree_edges()
    # every keypoint except the root has exactly one parent edge
    assert sorted(dst) == list(range(1, 18))


def test_nms_golden(golden_dir):
    g = _load(golden_dir, "nms_cases.npz")
    for i in range(int(g["count"])):
        sel = D.nms_ref(g[f"{i}/bbox"], 0.3, g[f"{i}/score"])
        assert sel.dtype == np.int32
        assert np.array_equal(sel, g[f"{i}/sel"]), i
    assert np.array_equal(D.nms_ref(g["noscore/bbox"], 0.3), g["noscore/sel"])
    assert D.nms_ref(np.zeros((0, 4), np.float32), 0.3).shape == (0,)


def _golden_heads(g):
    for i in range(int(g["count"])):
        kind, seed = str(g[f"{i}/kind"]), int(g[f"{i}/seed"])
        yield i, kind, seed, make_head(kind, seed)


def make_head(kind, seed):
    if kind == "crowd":
        return synth.planted_crowd_head(seed)
    C = cfg.lastsize()
    h = prng.uniform01(prng.stream_seed(seed, 0), C * 576).reshape(C, 24, 24)
    h[0:36] = prng.uniform(prng.stream_seed(seed, 1), 36 * 576, 0.2, 1.0).reshape(36, 24, 24)
    h[72:108] = prng.uniform(prng.stream_seed(seed, 2), 36 * 576, 0.05, 0.3).reshape(36, 24, 24)
    return h.astype(np.float32)


def test_decode_golden(golden_dir):
    g = _load(golden_dir, "decode_heads.npz")
    for i, kind, seed, head in _golden_heads(g):
        res = D.decode_ref(head)
        assert res["n"] == int(g[f"{i}/n"])
        for k in ("root_cell", "kp_cell", "limb_arg", "bbox", "score", "cand", "selected"):
            assert np.array_equal(res[k], g[f"{i}/{k}"]), (i, kind, seed, k)


def test_decode_tree_walk_equals_chain_replay():
    """The single tree walk must equal replaying the five chains (datatest.py:103-127)."""
    head = synth.planted_crowd_head(3)
    res = D.decode_ref(head)
    delta, x, y, w, h, e = D.split_head(head)
    for i in range(res["n"]):
        rh, rw = divmod(int(res["root_cell"][i]), 24)
        found = {0: (rh, rw)}
        for es, ts in D.DIRECTED_GRAPHS:
            ih, iw = rh, rw
            for ei, t in zip(es, ts):
                u = int(np.argmax(e[ei, :, :, ih, iw]))
                jh, jw = ih + u // 21 - 10, iw + u % 21 - 10
                if jh < 0 or jw < 0 or jh >= 24 or jw >= 24 or delta[t, jh, jw] < np.float32(0.15):
                    break
                found[t] = (jh, jw)
                ih, iw = jh, jw
        mine = {k: divmod(int(c), 24) for k, c in enumerate(res["kp_cell"][i]) if c >= 0}
        assert mine == found


def test_argmax_first_index_rule():
    head = np.zeros((cfg.lastsize(), 24, 24), np.float32)
    head[6 * 18 + 5, 3, 4] = 0.7
    head[6 * 18 + 9, 3, 4] = 0.7            # tie: lowest s wins (np.argmax)
    am = D.limb_argmax_dense(head)
    assert am[0, 3, 4] == 5 and am[0, 0, 0] == 0


@pytest.mark.parametrize("name", ["forward_d22_96", "forward_d54_96", "forward_d38_96"])
def test_forward_golden_small(golden_dir, name):
    g = _load(golden_dir, name + ".npz")
    arch_name, size, batch = str(g["arch"]), int(g["size"]), int(g["batch"])
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    sd = synth.make_state_dict(arch_name, int(g["seed_w"]), bn_stats=stats)
    x = Fr.normalize_u8(prng.u8_frames(int(g["seed_in"]), batch, (size, size)))
    torch.set_num_threads(8)
    out = Fr.forward_ref(sd, x, arch_name).numpy()
    # same arithmetic as the reference modules; allow last-bit differences across CPUs
    assert np.abs(out - g["head"]).max() <= 2e-6


def test_forward_golden_384(golden_dir):
    g = _load(golden_dir, "forward_d22_384.npz")
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    sd = synth.make_state_dict("drn_d_22", int(g["seed_w"]), bn_stats=stats)
    x = Fr.normalize_u8(prng.u8_frames(int(g["seed_in"]), int(g["batch"]), (384, 384)))
    torch.set_num_threads(8)
    out = Fr.forward_ref(sd, x, "drn_d_22").numpy()
    assert out.shape == (2, 7605, 24, 24)
    assert np.abs(out.reshape(-1)[g["head_idx"]] - g["head_val"]).max() <= 2e-6
    assert np.allclose(out.astype(np.float64).sum(axis=(2, 3)), g["head_chan_sum"], atol=1e-3)


def test_program_flops_and_shapes():
    ops = A.build_program("drn_d_22")
    assert len(ops) == 33                                        # 2 projection shortcuts ride on their conv2
    plain = A.build_program("drn_d_22", fuse_shortcut=False)
    assert len(plain) == 35 and A.conv_flops(plain, 384, 384) == A.conv_flops(ops, 384, 384)
    fused = A.build_program("drn_d_22", fuse_stem=True)
    assert len(fused) == 32 and fused[0].next3x3 is not None     # layer0 + layer1 share one launch
    assert A.conv_flops(fused, 384, 384) == A.conv_flops(ops, 384, 384)
    assert abs(A.conv_flops(ops, 384, 384) / 1e9 - 95.304) < 0.01
    shapes = A.tensor_shapes(ops, 384, 384)
    assert shapes["head"] == (24, 24, 7605)
    ops54 = A.build_program("drn_d_54")
    assert len(ops54) == 67
    assert abs(A.conv_flops(ops54, 384, 384) / 1e9 - 186.13) < 0.01
    assert len(A.param_spec("drn_d_22")) == 207


def test_prng_is_stable():
    # known-answer values pin the generator so fixtures replay on any machine
    assert prng.raw_u64(0, 2).tolist() == [16294208416658607535, 7960286522194355700]
    f = prng.u8_frames(1234, 1, (4, 4))
    assert f.shape == (1, 4, 4, 3) and f.dtype == np.uint8


@pytest.mark.parametrize("name", ["forward_d22_96", "forward_d54_96"])
def test_fused_program_equals_reference_order(golden_dir, name):
    """Host lowering (arch.build_program: BN folding, pre-activation second output, Bottleneck tail) executed
    with torch-CPU ops must reproduce the reference head."""
    from oracle import fused_ref
    g = _load(golden_dir, name + ".npz")
    arch_name = str(g["arch"])
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    sd = synth.make_state_dict(arch_name, int(g["seed_w"]), bn_stats=stats)
    x = Fr.normalize_u8(prng.u8_frames(int(g["seed_in"]), int(g["batch"]), (96, 96)))
    out = fused_ref.fused_forward_ref(sd, x, arch_name).numpy()
    err, err64 = np.abs(out - g["head"]).max(), np.abs(out - g["head_f64"]).max()
    assert err <= 1e-4 or err64 <= 1.5 * float(g["ref_f32_noise"]), (err, err64)


def test_loss_oracle_golden(golden_dir):
    """oracle/loss_ref.py (PPNLoss restatement + autograd) replays the values the reference itself produced."""
    from oracle import loss_ref as Lr, targets_ref as T
    g = _load(golden_dir, "loss_cases.npz")
    for tag in ("a", "iou_only"):
        seed, batch = int(g[f"{tag}/seed"]), int(g[f"{tag}/batch"])
        tg = T.synthetic_batch(seed, batch)
        head = prng.uniform(prng.stream_seed(seed, 7), batch * cfg.lastsize() * 576, 0.02, 0.98).reshape(
            batch, cfg.lastsize(), 24, 24)
        on = tg["delta"] > 0
        for lo, key, a, b in ((36, "tx", 0.9, 0.03), (54, "ty", 0.95, 0.02), (72, "tw", 1.2, 0.01), (90, "th", 0.8, 0.01)):
            head[:, lo:lo + 18][on] = (tg[key][on] * a + b).astype(np.float32)
        losses, grad = Lr.loss_and_grad_ref(head, tg, g[f"{tag}/coeff"])
        assert np.allclose(losses, g[f"{tag}/losses"], rtol=1e-5)
        ref = g[f"{tag}/grad_val"]
        assert np.abs(grad.reshape(-1)[g[f"{tag}/grad_idx"]] - ref).max() <= 1e-6 * max(1.0, float(np.abs(ref).max()))


def test_target_encoder_roundtrip():
    """Known-answer property (idea of datatest.py:403-412): targets encoded by the dataset rules, fed to the
    decoder as a perfect head, come back as the planted people."""
    from oracle import targets_ref as T
    people = T.synthetic_people(321)
    tg = T.encode_targets(people)
    head = np.concatenate([tg["delta"], np.ones_like(tg["delta"]), tg["tx"], tg["ty"], tg["tw"], tg["th"],
                           tg["te"].reshape(17 * 441, 24, 24)], 0).astype(np.float32)
    res = D.decode_ref(head)
    roots = {(int(p["bbox"][1] // 16), int(p["bbox"][0] // 16)) for p in people}
    got = {divmod(int(c), 24) for c in res["root_cell"]}
    assert got <= roots and len(got) >= 1


def test_train_oracle_golden(golden_dir):
    """oracle/train_ref.py (one training iteration, main.py:664-777, restated with CPU autograd) replays the values
    the imported reference produced (tests/golden/train_d22_96.npz).  Run in f32 the restatement was bit-identical
    to the reference when the fixture was made (asserted by make_golden.py); the stable f64 run is compared here,
    and the reference's own f32 numbers must lie within the f32-vs-f64 noise recorded beside them."""
    from oracle import train_ref, targets_ref as T
    g = _load(golden_dir, "train_d22_96.npz")
    size, batch = int(g["size"]), int(g["batch"])
    sd = synth.make_state_dict(str(g["arch"]), int(g["seed_w"]))
    x = Fr.normalize_u8(prng.u8_frames(int(g["seed_in"]), batch, (size, size)))
    tg = T.synthetic_batch(int(g["seed_t"]), batch, insize=(size, size), outsize=(size // 16, size // 16))
    names = [str(n) for n in g["names"]]
    assert names[-13] == "conv1.weight" and len(names) == 105
    for so, tag in ((False, "g1_f64"), (True, "g2_f64")):
        r = train_ref.train_iteration_ref(sd, x, tg, g["w_before"], g["base"], str(g["arch"]), (size, size),
                                          float(g["alpha"]), second_order=so)
        for i, n in enumerate(names):
            got = r["grads"][n].reshape(-1)[g[tag + "/idx"][i]]
            scale = max(1e-6, float(g[tag + "/norm"][i]))
            assert np.abs(got - g[tag + "/val"][i]).max() <= 1e-6 * scale, (n, so)
            assert abs(np.sqrt((r["grads"][n] ** 2).sum()) - g[tag + "/norm"][i]) <= 1e-6 * scale, (n, so)
    assert np.allclose(r["losses"], g["losses"], rtol=2e-5)
    assert np.allclose(r["G"], g["G_f64"], rtol=1e-6) and np.allclose(r["C"], g["C_f64"], rtol=1e-6)
    # the reference's f32 run vs the f64 values: G, C within a few per cent, and so the task weights
    assert np.allclose(g["G"], g["G_f64"], rtol=0.2) and np.allclose(g["dw"], g["dw_f64"], rtol=0.2)
    rel = np.abs(g["g1/norm"] - g["g1_f64/norm"]) / np.maximum(g["g1_f64/norm"], 1e-6)
    assert np.median(rel) < 0.02 and rel[g["g1_f64/norm"] > 1e-4].max() < 0.5    # conv2.bias: exactly 0 in theory


def test_ap_evaluation_golden(golden_dir):
    """evaluate.evaluation == the reference's datatest.evaluation -> eval_helpers.assignGTmulti / computeRPC / VOCap
    -> getCum on the synthetic pck_objects of synth.eval_case (expected AP values produced by the imported reference,
    tests/golden/eval_cases.npz)."""
    from pytorch_pose_proposal_network_amd import evaluate
    g = _load(golden_dir, "eval_cases.npz")
    for seed, n, exp in zip(g["seeds"], g["sizes"], g["ap"]):
        got = evaluate.evaluation(synth.eval_case(int(seed), int(n)))
        assert np.allclose(got, exp, rtol=0, atol=1e-9), (seed, got, exp)
    # structural properties: perfect predictions score 100, an empty prediction set scores 0
    obj = synth.eval_case(3, 8)
    perfect = [list(o) for o in obj]
    for i, (kps, boxes) in enumerate(zip(obj[1], obj[4])):
        hs, ss = [], []
        for p in range(len(boxes)):
            hm = {0: np.zeros(4, np.float32)}
            sm = {0: np.float32(0.9)}
            for k in range(1, 18):
                x, y = kps[p][k - 1]
                hm[k] = np.array([y - 4, x - 4, y + 4, x + 4], np.float32)
                sm[k] = np.float32(0.9)
            hs.append(hm)
            ss.append(sm)
        perfect[2][i], perfect[3][i] = hs, ss
    assert np.allclose(evaluate.evaluation(perfect), 100.0)
    empty = [list(o) for o in obj]
    empty[2], empty[3] = [[] for _ in obj[2]], [[] for _ in obj[3]]
    assert np.allclose(evaluate.evaluation(empty), 0.0)


def test_target_encoder_oracle_equals_reference_fixture(golden_dir):
    """oracle/targets_ref.encode_targets == the reference's KeypointsDataset.__getitem__ (dataset.py:96-200) on the
    fixture's people lists: ten tensors, bit for bit (fixture generated by make_golden.py --only targets)."""
    from oracle import targets_ref as T
    g = np.load(os.path.join(golden_dir, "targets_cases.npz"))
    n = int(g["n_cases"])
    assert n >= 4
    for i in range(n):
        people = [dict(bbox=tuple(g[f"case{i}/bbox"][p]), points=g[f"case{i}/points"][p],
                       visible=g[f"case{i}/visible"][p], size=g[f"case{i}/size"][p])
                  for p in range(len(g[f"case{i}/size"]))]
        mine = T.encode_targets(people)
        for k in ("delta", "weight", "weight_ij", "tx", "ty", "tx_half", "ty_half", "tw", "th", "te"):
            assert np.array_equal(mine[k], g[f"case{i}/{k}"]), (i, k)
    # the edge cases are really in there: a negative offset (keypoint left of the frame, int() truncation)
    assert float(g["case3/tx"].min()) < 0 and float(g["case3/ty"].min()) < 0


def test_ingest_oracle_properties():
    """oracle/ingest_ref.py (rt_test.py:150-157; parity unpinned -- cv2 is absent): identity size is a pure 180-degree
    rotation + channel swap; any resize stays within 1 LSB of real-valued bilinear interpolation with OpenCV's
    half-pixel centres; exact halving is the 2x2 box mean."""
    from oracle import ingest_ref as I
    frame = prng.u8_frames(9, 1, (384, 384))[0]
    assert np.array_equal(I.grab_frame_ref(frame), frame[::-1, ::-1, ::-1])
    for (hs, ws) in ((480, 640), (720, 1280), (150, 200), (385, 383)):
        src = prng.u8_frames(10 + hs, 1, (hs, ws))[0]
        got = I.resize_linear_u8(src, (384, 384)).astype(np.float64)
        fy = np.clip((np.arange(384) + 0.5) * hs / 384 - 0.5, 0, hs - 1)
        fx = np.clip((np.arange(384) + 0.5) * ws / 384 - 0.5, 0, ws - 1)
        y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
        y1, x1 = np.minimum(y0 + 1, hs - 1), np.minimum(x0 + 1, ws - 1)
        wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
        s = src.astype(np.float64)
        ref = (s[y0][:, x0] * (1 - wy) * (1 - wx) + s[y0][:, x1] * (1 - wy) * wx + s[y1][:, x0] * wy * (1 - wx) +
               s[y1][:, x1] * wy * wx)
        assert np.abs(got - ref).max() <= 1.0, (hs, ws, np.abs(got - ref).max())
    src = prng.u8_frames(12, 1, (768, 768))[0].astype(np.int64)
    box = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
    assert np.array_equal(I.resize_linear_u8(src.astype(np.uint8), (384, 384)), box.astype(np.uint8))


def test_emulated_oracle_fixture_frame0(golden_dir):
    """tests/golden/e2e_emulated.npz (what the emulated-storage oracle returns on the end-to-end fixtures: the yardstick of
    the 16-bit GPU gates, tests/test_e2e_gpu.py) is what tests/golden/make_emulated.py computes: frame 0 of the tuned
    fixture recomputed here in both modes -- same people, same agreement with the reference pipeline's people."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_emulated", os.path.join(golden_dir, "make_emulated.py"))
    me = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(me)
    z = np.load(os.path.join(golden_dir, "e2e_emulated.npz"))
    fx = "e2e_tuned_d22_384"
    sd, arch, size, u8, exp = me.setup(fx)
    for mode in me.MODES:
        emax, emean, people, tot = me.emulate(sd, arch, size, u8, exp, mode, frames=[0])
        assert people[0]["n"] == int(z[f"{fx}/{mode}/frame0/n"])
        assert np.array_equal(people[0]["kp_cell"], z[f"{fx}/{mode}/frame0/kp_cell"])
        assert np.array_equal(people[0]["limb_arg"], z[f"{fx}/{mode}/frame0/limb_arg"])
        assert emax <= float(z[f"{fx}/{mode}/head_err"][0]) + 1e-12
        assert 0 < tot[1] <= tot[0] and tot[0] == int(exp[0]["n"])
    # the yardstick is what DESIGN.md section 2 quotes
    assert int(z["e2e_d22_384/bfloat16/agreement"][0]) == 260 and int(z["e2e_tuned_d22_384/float16/agreement"][0]) == 76
