"""CPU study 4 (not a test; VERDICT r4 item 9, PAPER STUDY): what would a fused F(2x2, 3x3) Winograd form of the dilated 512-channel
layers (backbone.6.* / 7 / 8: 74 % of DRN-D-22's FLOPs) cost in the TASK metric of a 16-bit mode?  A dilation-d 3x3 convolution is d^2
independent dense 3x3 convolutions on the interleaved sub-grids x[a::d, b::d]; each runs as Winograd F(2x2, 3x3): V = B^T d B per 4x4
input tile, U = G g G^T per filter (f32, precomputed), M = sum_cin U * V per position (16 GEMMs, 16/36 = 1/2.25 of the direct form's
multiply-adds), Y = A^T M A.  The 16-bit kernel would hold V and U in the 16-bit type (MFMA operands) and accumulate M in f32: that is
what is emulated here (V and U rounded, everything else f32), inside the emulated-storage oracle of tests/precision_study_mixed.py.

    python tests/precision_study_winograd.py [--tail float16|bfloat16] [--frames 8]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decode_ref as D, forward_ref as Fr, fused_ref  # noqa: E402
from pytorch_pose_proposal_network_amd import arch as A, decode, prng, synth  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_dense(x, w, q):
    """3x3 stride-1 pad-1 convolution of x [B,C,H,W] (H, W even) with w [O,C,3,3] as F(2x2,3x3); q rounds the MFMA operands."""
    B_, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    t = F.unfold(xp, kernel_size=4, stride=2).view(B_, C, 4, 4, -1)                  # [B,C,4,4,T]
    V = q(torch.einsum("ip,bcpqt,jq->bcijt", BT, t, BT))                             # B^T d B
    U = q(torch.einsum("ip,ocpq,jq->ocij", G, w, G))                                 # G g G^T
    M = torch.einsum("ocij,bcijt->boijt", U, V)                                      # 16 GEMMs over cin, f32 accumulation
    Y = torch.einsum("pi,boijt,qj->bopqt", AT, M, AT)                                # A^T M A: [B,O,2,2,T]
    O = w.shape[0]
    return F.fold(Y.reshape(B_, O * 4, -1), (H, W), kernel_size=2, stride=2)


def winograd_dilated(x, w, d, q):
    out = torch.empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3])
    for a in range(d):
        for b in range(d):
            out[:, :, a::d, b::d] = winograd_dense(x[:, :, a::d, b::d].contiguous(), w, q)
    return out


def forward(sd, x, arch, dt, wino):
    """The fused program with every launch rounding weights and stored tensors to `dt` (the plain 16-bit policy); the launches
    named in `wino` run their convolution as Winograd with 16-bit V / U instead of 16-bit x / w."""
    ops = A.build_program(arch, fuse_stem=False, fuse_shortcut=False)
    q = lambda t: t.to(dt).float()          # noqa: E731
    tensors = {"input": x.float()}
    with torch.no_grad():
        for op in ops:
            w = fused_ref._t(sd[op.weight]).float()
            src = q(tensors[op.src])
            if op.name in wino:
                acc = winograd_dilated(src, w, op.dilation, q)
            else:
                acc = F.conv2d(src, q(w), None, op.stride, op.pad, op.dilation)
            s1 = b1 = None
            if op.bn1:
                s1, b1 = fused_ref._fold(sd, op.bn1)
            if op.bias:
                bias = fused_ref._t(sd[op.bias]).double()
                b1 = bias * s1 + b1 if s1 is not None else bias
            v = acc
            if s1 is not None:
                v = v * s1.float().view(1, -1, 1, 1)
            if b1 is not None:
                v = v + b1.float().view(1, -1, 1, 1)
            v = fused_ref._ACT[op.act1](v)
            if op.residual:
                v = v + q(tensors[op.residual])
            if op.out_raw:
                tensors[op.out_raw] = v if op.nchw_f32_out else q(v)
            if op.out_act:
                u = v
                if op.bn2:
                    s2, b2 = fused_ref._fold(sd, op.bn2)
                    u = u * s2.float().view(1, -1, 1, 1) + b2.float().view(1, -1, 1, 1)
                tensors[op.out_act] = q(fused_ref._ACT[op.act2](u))
    return tensors["head"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--tail", default="float16")
    ap.add_argument("--fixture", default="e2e_d22_384")
    args = ap.parse_args()
    dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[args.tail]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = np.load(os.path.join(ROOT, "tests", "golden", args.fixture + ".npz"))
    arch, size, batch = str(g["arch"]), int(g["size"]), min(int(g["batch"]), args.frames)
    st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch}_seed0.npz"))
    sd = synth.make_state_dict(arch, int(g["seed_w"]), bn_stats={k: st[k] for k in st.files})
    for k in g.files:
        if k.startswith("override/"):
            sd[k[len("override/"):]] = g[k]
    u8 = prng.u8_frames(int(g["seed_in"]), int(g["batch"]), (size, size))[:batch]
    x = torch.as_tensor(Fr.normalize_u8(u8))
    exp = [{k: g[f"{i}/{k}"] for k in ("n", "kp_cell", "limb_arg")} for i in range(batch)]
    ops = A.build_program(arch, fuse_stem=False, fuse_shortcut=False)
    dil = [o.name for o in ops if o.k == 3 and o.stride == 1 and o.cin == 512 and o.cout == 512 and o.dilation in (2, 4)]
    # self-check of the algebra: exact arithmetic reproduces the direct convolution
    xs, ws = torch.randn(1, 8, 16, 16), torch.randn(4, 8, 3, 3)
    err = (winograd_dilated(xs, ws, 2, lambda t: t) - F.conv2d(xs, ws, None, 1, 2, 2)).abs().max()
    assert err < 1e-4, err
    for name, wino in (("direct (the plain 16-bit mode)", ()), (f"Winograd F(2x2,3x3) on {len(dil)} launches ({', '.join(dil)})", tuple(dil))):
        head = np.concatenate([forward(sd, x[i:i + 1], arch, dt, wino).numpy() for i in range(batch)])
        tot = np.zeros(5, np.int64)
        for i in range(batch):
            tot += np.array(decode.people_agreement(exp[i], D.decode_ref(head[i], insize=(size, size))))
        n, exact, same, kp_eq, kp_all = (int(v) for v in tot)
        print(f"{args.fixture}, tail {args.tail}, {name}: people exact {exact}/{n}, same root {same}/{n}, kp cells {kp_eq}/{kp_all}", flush=True)


if __name__ == "__main__":
    main()
