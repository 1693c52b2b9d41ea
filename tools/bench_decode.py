"""Micro-benchmark of the decode kernels on planted-crowd heads (BASELINE config 5 shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_pose_proposal_network_amd import synth, decode

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
if "--dense" in sys.argv:   # unstructured heads: ~500 root candidates / image (worst case for NMS + parse)
    from pytorch_pose_proposal_network_amd import prng, config as cfg
    def dense(seed):
        C = cfg.lastsize()
        h = prng.uniform01(prng.stream_seed(seed, 0), C * 576).reshape(C, 24, 24)
        h[0:36] = prng.uniform(prng.stream_seed(seed, 1), 36 * 576, 0.2, 1.0).reshape(36, 24, 24)
        h[72:108] = prng.uniform(prng.stream_seed(seed, 2), 36 * 576, 0.05, 0.3).reshape(36, 24, 24)
        return h.astype(np.float32)
    heads = np.stack([dense(100 + (i % 4)) for i in range(B)])
else:
    heads = np.stack([synth.planted_crowd_head(7 + (i % 8)) for i in range(B)])
# 8 distinct copies of the batch so that consecutive iterations do not hit the 256 MB infinity cache
hs = [torch.from_numpy(heads).cuda() for _ in range(4)]
dec = decode.Decoder(B)
for h in hs: dec(h)
torch.cuda.synchronize()
for name, fn in (("limb_argmax", dec.limb_argmax), ("decode(total)", dec)):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(hs[i % 4]); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    byt = heads.nbytes
    print(f"{name}: median {ms[len(ms)//2]*1e3:.1f} us  min {ms[0]*1e3:.1f} us  -> {byt/ms[len(ms)//2]/1e6:.1f} GB/s "
          f"(algorithmic {byt/1e6:.1f} MB)")
