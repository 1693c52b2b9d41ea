"""Fused-path decode on the benchmark's own dense heads (random-network heads, ~490 root candidates per frame): the
root_mask_kernel + parse_kernel pair against the single parse kernel (PPN_DECODE_SPREAD=0).  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_pose_proposal_network_amd import decode, drn, model, prng, synth

B, S = 32, 384
st = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch_pose_proposal_network_amd",
                          "data", "bn_calib_drn_d_22_seed0.npz"))
net = model.PoseProposalNet(drn.drn_d_22(), compute_dtype="bfloat16").cuda()
net.load_state_dict(synth.make_state_dict("drn_d_22", 0, bn_stats={k: st[k] for k in st.files}))
frames = torch.from_numpy(prng.u8_frames(1234, B, (S, S))).cuda()
unary, keys = net.forward_u8(frames, fused_decode=True)
unary, keys = unary.clone(), keys.clone()
d = decode.Decoder(B)
for _ in range(5):
    d.decode_fused(unary, keys)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); d.decode_fused(unary, keys); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print(f"decode_fused on {B} dense heads ({int(d.out.count.sum())} people): median {ms[15]*1e3:.1f} us, min {ms[0]*1e3:.1f} us "
      f"(PPN_DECODE_SPREAD={os.environ.get('PPN_DECODE_SPREAD', '8')})")
