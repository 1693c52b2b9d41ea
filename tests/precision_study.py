"""CPU precision study (not a test; `python tests/precision_study.py [--frames N]`): what each storage / operand
precision policy of the fused program costs in PEOPLE on the reference-generated end-to-end fixtures, emulated with the
torch-CPU oracle (oracle/fused_ref.py: weights and stored tensors rounded where the HIP modes round them, f32
accumulation).  Answers VERDICT r3 item 3(b) -- "residual / skip stream stored f16 or f32, MFMA operands bf16" -- before
any kernel is written: only tensors that are never a convolution operand can be kept wider than the MFMA operand type
(LDS-DMA cannot convert on load), which in DRN-D-22 are the raw sums between the two blocks of a stage and the head's R.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decode_ref as D, forward_ref as Fr, fused_ref  # noqa: E402
from pytorch_pose_proposal_network_amd import decode, prng, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--fixture", default="e2e_d22_384")
    args = ap.parse_args()
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
    ref = Fr.forward_ref(sd, x, arch).numpy()
    policies = [("bf16 everywhere (shipped bf16 mode)", dict(emulate_dtype=torch.bfloat16, fuse_stem="all")),
                ("bf16 operands, residual-only tensors f16", dict(emulate_dtype=torch.bfloat16, fuse_stem="all", residual_dtype=torch.float16)),
                ("bf16 operands, residual-only tensors f32", dict(emulate_dtype=torch.bfloat16, fuse_stem="all", residual_dtype=None)),
                ("f16 everywhere (shipped f16 mode)", dict(emulate_dtype=torch.float16, fuse_stem="all")),
                ("f16 operands, residual-only tensors f32", dict(emulate_dtype=torch.float16, fuse_stem="all", residual_dtype=None))]
    for name, kw in policies:
        head = np.concatenate([fused_ref.fused_forward_ref(sd, x[i:i + 1], arch, **kw).numpy() for i in range(batch)])
        tot = np.zeros(5, np.int64)
        for i in range(batch):
            tot += np.array(decode.people_agreement(exp[i], D.decode_ref(head[i], insize=(size, size))))
        n, exact, same, kp_eq, kp_all = (int(v) for v in tot)
        d = np.abs(head - ref)
        print(f"{args.fixture} {name:45s}: people exact {exact}/{n}, same root {same}/{n}, kp cells {kp_eq}/{kp_all}; "
              f"head max {d.max():.4f} mean {d.mean():.5f}", flush=True)


if __name__ == "__main__":
    main()
