"""Soak of round 4's new paths on the GPU (not a pytest module; a few seconds):
  * fused decode with the root NMS spread over 8 workgroups per image: 96 heads (planted crowds + dense random heads) x 5
    repeats == the NumPy oracle, bit for bit;
  * the bf16 default (IEEE-half prefix), float16 + exact prefix and float16x3 models under THREE stream lanes: 30 submits of
    the same frames give 30 identical results, equal to the single-stream path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import decode_ref as D
from pytorch_pose_proposal_network_amd import config as cfg, decode, drn, model, prng, rt, synth
from test_oracle import make_head

heads = np.stack([synth.planted_crowd_head(100 + i) for i in range(64)] + [make_head("random", 300 + i) for i in range(32)])
exp = [D.decode_ref(hd) for hd in heads]
h = torch.from_numpy(heads).cuda()
e = h[:, 6 * cfg.K:].reshape(96, len(cfg.EDGES), -1, 24, 24)
val, _ = e.max(dim=2)
first = (e == val.unsqueeze(2)).float().argmax(dim=2)
keys = ((val.contiguous().view(torch.int32).to(torch.int64) << 32) | (0xFFFFFFFF - first)).contiguous()
unary = h[:, :6 * cfg.K].contiguous()
d = decode.Decoder(96)
for rep in range(5):
    out = d.decode_fused(unary, keys).to_host()
    for i, (r, x) in enumerate(zip(out, exp)):
        assert r["n"] == int(x["n"]), (rep, i)
        for k in ("kp_cell", "limb_arg", "bbox", "score"):
            assert np.array_equal(r[k], x[k]), (rep, i, k)
print(f"fused decode (spread root NMS): 96 heads x 5 repeats == oracle ({sum(int(x['n']) for x in exp)} people, "
      f"{sum(1 for x in exp if len(x['cand']) >= 128)} heads through root_mask_kernel)", flush=True)

st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", "bn_calib_drn_d_22_seed0.npz"))
sd = synth.make_state_dict("drn_d_22", 0, bn_stats={k: st[k] for k in st.files})
frames = torch.from_numpy(prng.u8_frames(77, 32, (384, 384))).cuda()
for name, kw in (("bfloat16 (half prefix)", dict(compute_dtype="bfloat16")), ("float16 + exact prefix 3", dict(compute_dtype="float16", exact_prefix=3)),
                 ("float16x3", dict(compute_dtype="float16x3"))):
    net = model.PoseProposalNet(drn.drn_d_22(), **kw).cuda()
    net.load_state_dict(sd)
    ref = rt.inference_batch(frames, net).to_host()
    ref = [{k: (v.copy() if hasattr(v, "copy") else v) for k, v in r.items()} for r in ref]
    pipe = rt.MultiLaneInference(net, 32, (384, 384), lanes=3)
    res = [pipe.submit(frames) for _ in range(30)]
    for n_, r in enumerate(res):
        r.ready.synchronize()
        if n_ >= 27:                                   # a lane's buffers hold its LAST result: check the final three
            got = r.to_host()
            for x, y in zip(got, ref):
                assert x["n"] == y["n"] and all(np.array_equal(x[k], y[k]) for k in ("kp_cell", "limb_arg", "bbox", "score")), name
    pipe.close()
    print(f"{name}: three lanes x 30 submits == single stream ({sum(r['n'] for r in ref)} people)", flush=True)
    del net, pipe
