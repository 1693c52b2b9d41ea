"""Where does the bf16 head error come from?  Per launch of the fused program: the HIP bf16 activation (read from the
plan's buffers) against (a) the fp32 oracle and (b) the oracle with bf16 storage emulated at the same points
(oracle/fused_ref.py) -- relative L2 error and max error normalised by the tensor's rms.  (a) is quantisation noise
accumulating through the randomly initialised network, (b) is what the KERNELS add to it.
Test infrastructure (imports oracle/): run by hand, output kept under profiles/.

    python tools/bf16_layer_errors.py [arch] [size]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import forward_ref as Fr, fused_ref
from pytorch_pose_proposal_network_amd import drn, model, prng, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arch = sys.argv[1] if len(sys.argv) > 1 else "drn_d_22"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 384
g = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch}_seed0.npz"))
sd = synth.make_state_dict(arch, 0, bn_stats={k: g[k] for k in g.files})
u8 = prng.u8_frames(1234, 2, (size, size))
torch.set_num_threads(min(16, os.cpu_count() or 1))
x = Fr.normalize_u8(u8)
taps32, taps16 = {}, {}
fused_ref.fused_forward_ref(sd, x, arch, emulate_bf16=False, taps=taps32, fuse_stem="all")
fused_ref.fused_forward_ref(sd, x, arch, emulate_bf16=True, taps=taps16, fuse_stem="all")
net = model.PoseProposalNet(getattr(drn, arch)(), insize=(size, size), outsize=(size // 16, size // 16),
                            compute_dtype="bfloat16").cuda()
net.load_state_dict(sd)
head = net.forward_u8(torch.from_numpy(u8).cuda())
torch.cuda.synchronize()
plan = next(iter(net._plans.values()))
print(f"{arch} {size}x{size} bf16 mode, batch 2: per launch  rel-L2 / max|err|/rms  vs fp32 oracle | vs bf16-emulating oracle")
for op in net._ops:
    name = op.out_raw or op.out_act
    t = plan.buffers["head" if op.nchw_f32_out else name].float().cpu()
    if not op.nchw_f32_out:
        t = t.permute(0, 3, 1, 2)
    def err(ref):
        ref = ref.float()
        d = (t - ref).double()
        rms = float(ref.double().pow(2).mean().sqrt()) + 1e-30
        return float(d.pow(2).sum().sqrt() / (ref.double().pow(2).sum().sqrt() + 1e-30)), float(d.abs().max()) / rms
    a, b = err(taps32[op.name]), err(taps16[op.name])
    print(f"{op.name[:44]:44s} {a[0]:9.2e} {a[1]:9.2e} | {b[0]:9.2e} {b[1]:9.2e}")
