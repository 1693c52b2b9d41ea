"""CPU precision study 3 (not a test): WHERE is the 16-bit error injected in DRN-D-54's Bottleneck trunk?
The round-4 study (tests/precision_study_mixed.py) was run on DRN-D-22 only; its half / exact prefix (stem + layer3-4, 6.9 % of
the FLOPs) was derived there, and on D-54 the 16-bit modes reproduce < 10 % of the f32 pipeline's people.  Same method: the
torch-CPU oracle program with the launches up to a cut exact and the rest rounding weights and stored tensors to the 16-bit
type, people decoded from each head and compared with the ALL-EXACT head's people (no reference-generated people fixture
exists for D-54 at 384 x 384; the f32 head is pinned by tests/golden/forward_d54_384.npz).

    python tests/precision_study_d54.py [--frames 2] [--tail float16|bfloat16]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import decode_ref as D, forward_ref as Fr  # noqa: E402
from precision_study_mixed import forward_mixed  # noqa: E402
from pytorch_pose_proposal_network_amd import arch as A, decode, prng, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--tail", default="float16")
    ap.add_argument("--arch", default="drn_d_54")
    ap.add_argument("--size", type=int, default=384)
    args = ap.parse_args()
    dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[args.tail]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    arch, size, batch = args.arch, args.size, args.frames
    st = np.load(os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "data", f"bn_calib_{arch}_seed0.npz"))
    sd = synth.make_state_dict(arch, 0, bn_stats={k: st[k] for k in st.files})
    u8 = prng.u8_frames(1234, batch, (size, size))            # bench.py's frames (rank 0, batch 0)
    x = torch.as_tensor(Fr.normalize_u8(u8))
    ops = A.build_program(arch, fuse_stem=False, fuse_shortcut=False)
    shapes = A.tensor_shapes(ops, size, size)
    fl = [A.op_flops(o, shapes) for o in ops]
    names = [o.name for o in ops]

    def first(prefix):
        return next(i for i, n in enumerate(names) if n.startswith(prefix))
    cuts = [0, 3] + [first(p) for p in ("backbone.4.", "backbone.5.", "backbone.6.", "backbone.7.", "basicblock1.", "conv1x1_1")]

    def people(head):
        return [D.decode_ref(head[i], insize=(size, size)) for i in range(batch)]
    exact_head = np.concatenate([forward_mixed(sd, x[i:i + 1], arch, len(ops), dt)[0].numpy() for i in range(batch)])
    exp = people(exact_head)
    print(f"{arch} @ {size}, {batch} frames: the exact pipeline finds {sum(e['n'] for e in exp)} people", flush=True)
    for cut in cuts:
        head = np.concatenate([forward_mixed(sd, x[i:i + 1], arch, cut, dt)[0].numpy() for i in range(batch)])
        got = people(head)
        tot = np.zeros(5, np.int64)
        for i in range(batch):
            tot += np.array(decode.people_agreement(exp[i], got[i]))
        n, exact, same, kp_eq, kp_all = (int(v) for v in tot)
        err = np.abs(head - exact_head)
        print(f"{arch} exact up to launch {cut:3d} ({names[cut]:28s}) = {sum(fl[:cut]) / sum(fl):6.1%} of the FLOPs, tail {args.tail}: "
              f"people exact {exact}/{n}, same root {same}/{n}, kp cells {kp_eq}/{max(kp_all, 1)}; head err max {err.max():.4f} mean {err.mean():.5f}",
              flush=True)


if __name__ == "__main__":
    main()
