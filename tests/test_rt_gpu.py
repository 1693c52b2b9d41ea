"""GPU: the rt_test-shaped entry points (network / inference / inference_batch) end to end."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref as D, forward_ref as Fr
from pytorch_pose_proposal_network_amd import prng, synth

pytestmark = pytest.mark.gpu


def test_inference_matches_oracle_pipeline(golden_dir):
    from pytorch_pose_proposal_network_amd import rt
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    sd = synth.make_state_dict("drn_d_22", 0, bn_stats=stats)
    model, outsize, lgs = rt.network(image_size=96, state_dict=sd)
    assert outsize == (6, 6) and lgs == (21, 21) and model.lastsize == 7605 and model.gridsize == (16, 16)
    frame = prng.u8_frames(99, 1, (96, 96))[0]
    humans, scores = rt.inference(frame, model, outsize, lgs)
    # oracle pipeline on the HIP head (index parity is defined given the head; the head has its own test)
    head = model.forward_u8(torch.from_numpy(frame[None]).cuda()).cpu().numpy()[0]
    ref_head = Fr.forward_ref(sd, Fr.normalize_u8(frame[None]), "drn_d_22").numpy()[0]
    assert np.abs(head - ref_head).max() <= 1e-4
    exp_h, exp_s = D.humans_from_compact(D.decode_ref(head, insize=(96, 96)))
    assert len(humans) == len(exp_h)
    for a, b, sa, sb in zip(humans, exp_h, scores, exp_s):
        assert sorted(a) == sorted(b)
        for k in a:
            assert np.array_equal(a[k], b[k]) and sa[k] == sb[k]
    with pytest.raises(ValueError):
        rt.inference(frame.astype(np.float32), model, outsize, lgs)


def test_inference_batch_shards_are_independent(golden_dir):
    """Sharding property: decoding frames in one batch == decoding each frame alone (what N GPUs would do)."""
    from pytorch_pose_proposal_network_amd import rt, shard
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    model, _, _ = rt.network(image_size=96, state_dict=synth.make_state_dict("drn_d_22", 0, bn_stats=stats))
    frames = torch.from_numpy(prng.u8_frames(5, 4, (96, 96))).cuda()
    full = rt.inference_batch(frames, model).to_host()
    parts = []
    for r in range(2):
        idx = shard.frame_shard(4, r, 2)
        parts.append(rt.inference_batch(frames[idx].contiguous(), model).to_host())
    merged = shard.merge_shards(parts, 4)
    for a, b in zip(full, merged):
        assert a["n"] == b["n"]
        for k in ("kp_cell", "limb_arg", "bbox", "score"):
            assert np.array_equal(a[k], b[k])


def test_pipelined_inference_equals_serial(golden_dir):
    """rt.InferencePipeline (decode of batch i on a side stream under the conv stack of batch i+1, two slots of
    output buffers) must return exactly what the serial path returns, batch after batch, including slot reuse."""
    from pytorch_pose_proposal_network_amd import rt
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    model, _, _ = rt.network(image_size=96, state_dict=synth.make_state_dict("drn_d_22", 0, bn_stats=stats))
    batches = [torch.from_numpy(prng.u8_frames(100 + i, 3, (96, 96))).cuda() for i in range(5)]
    serial = [rt.inference_batch(b, model).to_host() for b in batches]
    pipe = rt.InferencePipeline(model, 3, (96, 96))
    results = []
    # keep two batches in flight: read batch i only after batch i+1 has been submitted
    prev = None
    for b in batches:
        cur = pipe.submit(b)
        if prev is not None:
            prev.ready.synchronize()
            results.append(prev.to_host())
        prev = cur
    prev.ready.synchronize()
    results.append(prev.to_host())
    pipe.flush()
    assert sum(r["n"] for res in serial for r in res) > 0
    for s_, p_ in zip(serial, results):
        for a, b in zip(s_, p_):
            assert a["n"] == b["n"]
            for k in ("kp_cell", "limb_arg", "bbox", "score"):
                assert np.array_equal(a[k], b[k])


@pytest.mark.parametrize("lanes", [2, 3])
def test_multi_lane_inference_equals_serial(golden_dir, lanes):
    """rt.MultiLaneInference (batches round-robin over independent stream lanes) returns exactly the serial results."""
    from pytorch_pose_proposal_network_amd import rt
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    model, _, _ = rt.network(image_size=96, state_dict=synth.make_state_dict("drn_d_22", 0, bn_stats=stats))
    batches = [torch.from_numpy(prng.u8_frames(200 + i, 3, (96, 96))).cuda() for i in range(7)]
    serial = [rt.inference_batch(b, model).to_host() for b in batches]
    pipe = rt.MultiLaneInference(model, 3, (96, 96), lanes=lanes)
    pending, results = [], []
    for b in batches:
        pending.append(pipe.submit(b))
        if len(pending) == lanes:                 # a lane's result must be read before the lane is reused
            r = pending.pop(0)
            r.ready.synchronize()
            results.append(r.to_host())
    for r in pending:
        r.ready.synchronize()
        results.append(r.to_host())
    pipe.close()                                  # restores the single-stream tile policy
    assert len(results) == len(serial)
    # seven DIFFERENT input tensors went through the lanes: a plan copies them into its own input buffer, so its
    # hipGraph is captured once (third run of a plan on a non-default stream), not once per fresh tensor
    caps = {k: c for k, c in model.graph_captures().items() if k[4]}
    assert all(c <= 1 for c in caps.values()), caps
    assert lanes != 2 or any(c == 1 for c in caps.values()), caps
    for s_, p_ in zip(serial, results):
        for a, b in zip(s_, p_):
            assert a["n"] == b["n"]
            for k in ("kp_cell", "limb_arg", "bbox", "score"):
                assert np.array_equal(a[k], b[k])


@pytest.mark.parametrize("hs,ws", [(480, 640), (720, 1280), (768, 768), (384, 384), (150, 200), (385, 383)])
def test_frame_ingest_bit_exact_vs_oracle(hs, ws):
    """SURVEY 8f-4, rt_test.py:150-157: resize + both flips + BGR->RGB on the device == oracle/ingest_ref.py,
    every byte, for down-scaling, the exact-halving path, identity and up-scaling (batch of 3)."""
    from oracle import ingest_ref as I
    from pytorch_pose_proposal_network_amd import rt
    src = prng.u8_frames(1000 + hs, 3, (hs, ws))
    got = rt.ingest_frames(torch.from_numpy(src).cuda()).cpu().numpy()
    assert got.shape == (3, 384, 384, 3)
    for b in range(3):
        assert np.array_equal(got[b], I.grab_frame_ref(src[b])), (hs, ws, b)


def test_grab_frame_feeds_inference_without_copies(golden_dir):
    """grab_frame(cap) -> inference: the camera frame is resized straight into the model's own input buffer and
    the result equals inference on the oracle-ingested frame."""
    from oracle import ingest_ref as I
    from pytorch_pose_proposal_network_amd import rt
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    model, outsize, lgs = rt.network(image_size=96, state_dict=synth.make_state_dict("drn_d_22", 0, bn_stats=stats))
    cam = prng.u8_frames(321, 1, (240, 320))[0]

    class Cap:
        def read(self):
            return True, cam

    buf = model.input_buffer(1, 96, 96)
    ret, frame = rt.grab_frame(Cap(), size=96, out=buf)
    assert ret and frame.data_ptr() == buf.data_ptr()
    humans, scores = rt.inference(frame, model, outsize, lgs)
    exp_h, exp_s = rt.inference(I.grab_frame_ref(cam, 96), model, outsize, lgs)
    assert len(humans) == len(exp_h)
    for a, b, sa, sb in zip(humans, exp_h, scores, exp_s):
        assert sorted(a) == sorted(b)
        for k in a:
            assert np.array_equal(a[k], b[k]) and sa[k] == sb[k]


def test_multi_lane_pinned_frames_and_async_readback(golden_dir):
    """MultiLaneInference.submit(pinned host frames, to_host=True): H2D straight into the lane's input buffer and D2H
    of the compact result into pinned buffers, both on the lane's stream, no host synchronisation inside -- results
    equal the serial device path; an image with more people than the staging capacity falls back to the full read."""
    from pytorch_pose_proposal_network_amd import decode, rt
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    model, _, _ = rt.network(image_size=96, state_dict=synth.make_state_dict("drn_d_22", 0, bn_stats=stats))
    hosts = [torch.from_numpy(prng.u8_frames(300 + i, 3, (96, 96))).pin_memory() for i in range(6)]
    serial = [rt.inference_batch(h.cuda(), model).to_host() for h in hosts]
    pipe = rt.MultiLaneInference(model, 3, (96, 96), lanes=2)
    with pytest.raises(ValueError):
        pipe.submit(torch.from_numpy(prng.u8_frames(1, 3, (96, 96))))          # pageable host memory
    pending, results = [], []
    for h in hosts:
        pending.append(pipe.submit(h, to_host=True))
        if len(pending) == 2:
            r = pending.pop(0)
            r.ready.synchronize()
            results.append(r.hosted.unpack())
    for r in pending:
        r.ready.synchronize()
        results.append(r.hosted.unpack())
    pipe.close()
    assert sum(x["n"] for res in serial for x in res) > 0
    for s_, p_ in zip(serial, results):
        for a, b in zip(s_, p_):
            assert a["n"] == b["n"]
            for k in ("kp_cell", "limb_arg", "bbox", "score"):
                assert np.array_equal(a[k], b[k])
    # capacity overflow -> synchronous fallback with the same content
    st = decode.HostStage(3, cap=1)
    res = rt.inference_batch(hosts[0].cuda(), model)
    res.to_host_async(st)
    torch.cuda.synchronize()
    for a, b in zip(serial[0], st.unpack()):
        assert a["n"] == b["n"] and np.array_equal(a["kp_cell"], b["kp_cell"])


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_graph_replay_equals_direct_launches(golden_dir, dtype):
    """The fused-decode plan launches directly twice and is then captured into a hipGraph (csrc/plan.hip): every replay
    must leave the unary tensor, the arg-max keys and the people lists bit-equal to the direct run -- the check that a
    graph node running out of order (the memset-node bug of round 2, csrc/plan.hip zero_fill_kernel) cannot pass."""
    from pytorch_pose_proposal_network_amd import decode, rt
    g = np.load(os.path.join(golden_dir, "forward_d22_96.npz"))
    stats = {k[3:]: g[k] for k in g.files if k.startswith("bn/")}
    model, _, _ = rt.network(image_size=96, state_dict=synth.make_state_dict("drn_d_22", 0, bn_stats=stats),
                             compute_dtype=dtype)
    st = torch.cuda.Stream()                                  # the legacy default stream cannot be captured
    with torch.cuda.stream(st):
        sets = [torch.from_numpy(prng.u8_frames(300 + i, 4, (96, 96))).cuda() for i in range(2)]
        dec = decode.Decoder(4, (6, 6), (96, 96), device="cuda")
        first = {}
        for rnd_ in range(6):                                 # runs 0-1 direct, the capture happens in run 2
            for i, fr in enumerate(sets):
                unary, keys = model.forward_u8(fr, fused_decode=True)
                people = dec.decode_fused(unary, keys).to_host()
                snap = (unary.clone(), keys.clone(), people)
                if rnd_ == 0:
                    first[i] = snap
                    continue
                assert torch.equal(snap[0], first[i][0]) and torch.equal(snap[1], first[i][1]), (rnd_, i)
                for a, b in zip(people, first[i][2]):
                    assert a["n"] == b["n"]
                    for k in ("kp_cell", "limb_arg", "bbox", "score"):
                        assert np.array_equal(a[k], b[k]), (rnd_, i, k)
        assert not torch.equal(first[0][1], first[1][1])      # the two frame sets really differ
    caps = model.graph_captures()
    assert sum(caps.values()) >= 1, caps                      # the later runs were graph replays
