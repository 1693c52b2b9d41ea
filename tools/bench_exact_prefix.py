"""float16 trunk behind an exact prefix: per-launch table (one launch in flight) and one-lane rate, stem as three f32 launches
vs layer0+layer1 fused (csrc/stem01.hip, f32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from pytorch_pose_proposal_network_amd import decode, drn, model, prng, synth

B, S = 32, 384
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", "bn_calib_drn_d_22_seed0.npz"))
sd = synth.make_state_dict("drn_d_22", 0, bn_stats={k: st[k] for k in st.files})
frames = torch.from_numpy(prng.u8_frames(1234, B, (S, S))).cuda()
for fuse in (False, True):
    net = model.PoseProposalNet(drn.drn_d_22(), compute_dtype="float16", exact_prefix=3, fuse_stem=fuse).cuda()
    net.load_state_dict(sd)
    d = decode.Decoder(B)
    def step():
        u, k = net.forward_u8(frames, fused_decode=True)
        d.decode_fused(u, k)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"fuse_stem={fuse}: {B / dt:.0f} images/s one lane ({dt * 1e3:.3f} ms)")
    net.profile_layers(frames, src_is_u8=True, fused_decode=True)
    for name, kern, ms, fl in net.profile_layers(frames, src_is_u8=True, fused_decode=True)[:12]:
        print(f"   {name:34s} {ms * 1e3:8.1f} us  {kern}")
    del net
