"""CPU precision study 2 (not a test): a MIXED inference mode -- the first layers exact (float16x3 / f32), the rest in a
16-bit mode.  Rounding noise injected early is amplified by every layer behind it, and the early layers are the cheap ones
(stem + layer3-5 = 17 % of DRN-D-22's FLOPs), so: how many of the reference pipeline's people does a 16-bit tail reproduce
when the trunk up to a given block is exact?  Emulated with the torch-CPU oracle (storage roundings of oracle/fused_ref.py
applied from a given launch on).    python tests/precision_study_mixed.py [--frames N] [--tail float16|bfloat16]"""
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


def forward_mixed(sd, x, arch, first_16bit, dt, exact_from=0, stem_dt=None, prefix_dt=None):
    """fused program without stem / shortcut fusion; launches [exact_from, first_16bit) are exact, the others round weights
    and stored outputs to `dt`."""
    ops = A.build_program(arch, fuse_stem=False, fuse_shortcut=False)
    tensors = {"input": x.float()}
    with torch.no_grad():
        for i, op in enumerate(ops):
            q = (lambda t: t.to(dt).float()) if (i >= first_16bit or i < exact_from) else (lambda t: t)
            if prefix_dt is not None and exact_from <= i < first_16bit:
                q = lambda t: t.to(prefix_dt).float()           # noqa: E731  the "exact" range runs in prefix_dt instead
            if stem_dt is not None and i < 3:
                # the stem computes in `stem_dt` (its weights, its input patch, the tensors between its layers); what leaves
                # it (launch 2's outputs) is stored in the trunk's type
                qs = lambda t: t.to(stem_dt).float()            # noqa: E731
                q_in, q_out = qs, (qs if i < 2 else (lambda t: t.to(dt).float()))
            else:
                q_in = q_out = q
            w = q_in(fused_ref._t(sd[op.weight]).float())
            src = q_in(tensors[op.src])                    # an exact producer's output enters a 16-bit launch rounded
            acc = F.conv2d(src, w, None, op.stride, op.pad, op.dilation)
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
                v = v + q_in(tensors[op.residual])
            if op.out_raw:
                tensors[op.out_raw] = v if op.nchw_f32_out else q_out(v)
            if op.out_act:
                u = v
                if op.bn2:
                    s2, b2 = fused_ref._fold(sd, op.bn2)
                    u = u * s2.float().view(1, -1, 1, 1) + b2.float().view(1, -1, 1, 1)
                tensors[op.out_act] = q_out(fused_ref._ACT[op.act2](u))
    return tensors["head"], ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--tail", default="float16")
    ap.add_argument("--fixture", default="e2e_d22_384")
    ap.add_argument("--exact-from", type=int, default=0, help="first exact launch (3 = behind the three stem layers)")
    ap.add_argument("--cuts", default="")
    ap.add_argument("--stem", default="", help="float16 / float32: the three stem launches compute in this type")
    ap.add_argument("--prefix", default="", help="float16: launches [exact-from, cut) run in this 16-bit type instead of exactly")
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
    shapes = A.tensor_shapes(ops, size, size)
    fl = [A.op_flops(o, shapes) for o in ops]
    names = [o.name for o in ops]
    cuts = [0] + [names.index(n) for n in ("backbone.4.0.downsample", "backbone.5.0.downsample", "backbone.6.0.downsample",
                                           "backbone.7.0", "basicblock1.downsample", "conv1x1_1")] + [len(ops)]
    if args.cuts:
        cuts = [int(c) for c in args.cuts.split(",")]
    for cut in cuts:
        stem_dt = {"": None, "float16": torch.float16, "float32": torch.float32}[args.stem]
        prefix_dt = {"": None, "float16": torch.float16}[args.prefix]
        head = np.concatenate([forward_mixed(sd, x[i:i + 1], arch, cut, dt, args.exact_from, stem_dt, prefix_dt)[0].numpy()
                               for i in range(batch)])
        tot = np.zeros(5, np.int64)
        for i in range(batch):
            tot += np.array(decode.people_agreement(exp[i], D.decode_ref(head[i], insize=(size, size))))
        n, exact, same, kp_eq, kp_all = (int(v) for v in tot)
        share = sum(fl[args.exact_from:cut]) / sum(fl)
        # cost model: exact launches at 3x the 16-bit cost (float16x3)
        cost = (2 * sum(fl[args.exact_from:cut]) + sum(fl)) / sum(fl)
        print((f"[stem in {args.stem}] " if args.stem else "") + (f"[prefix in {args.prefix}] " if args.prefix else "") +
              f"{args.fixture} exact from launch {args.exact_from} up to launch {cut:2d} ({names[cut] if cut < len(ops) else 'end':26s}) = {share:5.1%} of the FLOPs, "
              f"tail {args.tail}: people exact {exact}/{n}, same root {same}/{n}, kp cells {kp_eq}/{kp_all}; relative MFMA cost {cost:.2f}",
              flush=True)


if __name__ == "__main__":
    main()
