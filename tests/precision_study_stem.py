"""CPU precision study 3 (not a test): WHICH roundings of the 16-bit stem cost people -- its input patch, its weights, or the
tensors between its layers -- with an f16 tail behind it (emulated on torch-CPU; python tests/precision_study_stem.py).
Result (profiles/r04/precision_stem_roundings.txt): input and weights; the inner tensors do not matter."""
import os, sys, numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decode_ref as D, forward_ref as Fr, fused_ref
from pytorch_pose_proposal_network_amd import arch as A, decode, prng, synth
torch.set_num_threads(8)
g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_d22_384.npz"))
arch, size, batch = str(g["arch"]), int(g["size"]), int(g["batch"])
st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch}_seed0.npz"))
sd = synth.make_state_dict(arch, int(g["seed_w"]), bn_stats={k: st[k] for k in st.files})
u8 = prng.u8_frames(int(g["seed_in"]), batch, (size, size))
x = torch.as_tensor(Fr.normalize_u8(u8))
exp = [{k: g[f"{i}/{k}"] for k in ("n", "kp_cell", "limb_arg")} for i in range(batch)]
h16 = lambda t: t.to(torch.float16).float()
ident = lambda t: t
def fwd(xi, q_input, q_w, q_mid, q_out, tail=torch.float16):
    ops = A.build_program(arch, fuse_stem=False, fuse_shortcut=False)
    qt = lambda t: t.to(tail).float()
    tensors = {"input": xi.float()}
    with torch.no_grad():
        for i, op in enumerate(ops):
            stem = i < 3
            qw = q_w if stem else qt
            qi = (q_input if i == 0 else (ident if stem else qt))     # stem inner tensors were stored by q_mid already
            qo = (q_mid if i < 2 else q_out) if stem else qt
            w = qw(fused_ref._t(sd[op.weight]).float())
            acc = F.conv2d(qi(tensors[op.src]), w, None, op.stride, op.pad, op.dilation)
            s1 = b1 = None
            if op.bn1: s1, b1 = fused_ref._fold(sd, op.bn1)
            if op.bias:
                bias = fused_ref._t(sd[op.bias]).double(); b1 = bias * s1 + b1 if s1 is not None else bias
            v = acc
            if s1 is not None: v = v * s1.float().view(1, -1, 1, 1)
            if b1 is not None: v = v + b1.float().view(1, -1, 1, 1)
            v = fused_ref._ACT[op.act1](v)
            if op.residual: v = v + qt(tensors[op.residual])
            if op.out_raw: tensors[op.out_raw] = v if op.nchw_f32_out else qo(v)
            if op.out_act:
                u = v
                if op.bn2:
                    s2, b2 = fused_ref._fold(sd, op.bn2); u = u * s2.float().view(1, -1, 1, 1) + b2.float().view(1, -1, 1, 1)
                tensors[op.out_act] = qo(fused_ref._ACT[op.act2](u))
    return tensors["head"]
cases = [("stem all f16", h16, h16, h16, h16), ("exact input only", ident, h16, h16, h16), ("exact input + weights", ident, ident, h16, h16),
         ("exact input + weights + inner tensors (only the output rounded)", ident, ident, ident, h16), ("exact inner tensors only", h16, h16, ident, h16),
         ("exact weights only", h16, ident, h16, h16)]
for name, qi, qw, qm, qo in cases:
    head = np.concatenate([fwd(x[i:i+1], qi, qw, qm, qo).numpy() for i in range(batch)])
    tot = np.zeros(5, np.int64)
    for i in range(batch):
        tot += np.array(decode.people_agreement(exp[i], D.decode_ref(head[i], insize=(size, size))))
    print(f"f16 tail, {name:62s}: exact {tot[1]}/{tot[0]}, same root {tot[2]}", flush=True)
