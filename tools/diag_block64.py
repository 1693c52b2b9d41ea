import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from test_block64_gpu import _setup, _one_launch, _two_launches
dt_name = "bf16"
B, H, W = 8, 96, 96
L, lib, dt, tdt, t, st = _setup(dt_name, B, H, W, 3)
ref = _two_launches(L, lib, dt, tdt, t, st, B, H, W, True, True, True, flags=L.PPN_CONV_NO_FILTER_BANK)[1]
conv_only = _two_launches(L, lib, dt, tdt, t, st, B, H, W, False, True, True, flags=L.PPN_CONV_NO_FILTER_BANK)[1]
xr = t["x_raw"].float()
for i in range(3):
    o = _one_launch(L, lib, dt, tdt, t, st, B, H, W, True, True, True)[0]
    d = (o.view(torch.int16) != ref.view(torch.int16))
    idx = d.nonzero().cpu().numpy()
    print("run", i, len(idx), "bad")
    for (b, y, x, c) in idx[:12]:
        got, r, co, want_res = float(o[b, y, x, c]), float(ref[b, y, x, c]), float(conv_only[b, y, x, c]), float(xr[b, y, x, c])
        used = got - co
        # where does the used residual value come from?
        m = (xr - used).abs() < 2e-2 * max(1.0, abs(used))
        cand = m.nonzero()[:4].tolist()
        print(f"   [{b},{y},{x},{c}] got {got:.4f} ref {r:.4f} conv-only {co:.4f} residual wanted {want_res:.4f} used {used:.4f}  candidates {cand}")
