"""Diagnostic: phase times inside parse_kernel from the -DPPN_STAMP build (tools/stamp_conv.py --build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPN_LIB"] = os.path.join(ROOT, "tools", "bin", "libppn_stamp.so")
import numpy as np, torch
from pytorch_pose_proposal_network_amd import decode, prng, config as cfg, synth
def dense(seed):
    C = cfg.lastsize()
    h = prng.uniform01(prng.stream_seed(seed, 0), C * 576).reshape(C, 24, 24)
    h[0:36] = prng.uniform(prng.stream_seed(seed, 1), 36 * 576, 0.2, 1.0).reshape(36, 24, 24)
    h[72:108] = prng.uniform(prng.stream_seed(seed, 2), 36 * 576, 0.05, 0.3).reshape(36, 24, 24)
    return h.astype(np.float32)
for name, heads in (("dense", np.stack([dense(100 + i % 4) for i in range(32)])),
                    ("crowd", np.stack([synth.planted_crowd_head(7 + i % 8) for i in range(32)]))):
    dec = decode.Decoder(32)
    h = torch.from_numpy(heads).cuda()
    for _ in range(3):
        out = dec(h)
    torch.cuda.synchronize()
    raw = out.bbox[:, -1].reshape(32, -1).cpu().numpy().view(np.uint64)[:, :7].astype(np.int64)
    print("lib", __import__("pytorch_pose_proposal_network_amd.lib", fromlist=["x"]).LIB_PATH, "raw0", raw[0]); d = np.diff(raw, axis=1).mean(0)
    print(name, "cycles per phase [cand, sort, nms, stage-sync, stage, walk, output]:", d.astype(int), "total", int(d.sum()))
