"""Diagnostic: is the all-bf16 pipeline (half_prefix=-1, stem_dtype='bfloat16') deterministic?  Runs the plan several times on the
same frames and reports the first tensor (in launch order) that differs between runs."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from test_16bit_floors_gpu import _fixture, _net
from pytorch_pose_proposal_network_amd import prng
g, sd = _fixture("e2e_d22_384")
size, batch = int(g["size"]), int(g["batch"])
u8 = torch.from_numpy(prng.u8_frames(int(g["seed_in"]), batch, (size, size))).cuda()
for name in sys.argv[1:] or ["pure_bf16"]:
    net = _net(name, sd, size=size)
    runs = []
    for rep in range(4):
        net.forward_u8(u8)
        torch.cuda.synchronize()
        plan = net._get_plan(batch, size, size, True)
        runs.append({k: v.clone() for k, v in plan.buffers.items()})
    order = []
    for op in net._ops:
        for t in (op.out_raw, op.out_act):
            if t and t in runs[0]:
                order.append((op.name, t))
    for opname, t in order:
        a = runs[0][t]
        diffs = [int((a.view(torch.int16 if a.element_size() == 2 else torch.int32) != r[t].view(torch.int16 if a.element_size() == 2 else torch.int32)).sum().item()) for r in runs[1:]]
        print(f"{name}: {opname:40s} {t:28s} differing elements vs run 0: {diffs}")
        if any(diffs):
            idx = (a != runs[1][t]).nonzero()[:5]
            print("   first differing indices:", idx.tolist())
            break
