"""In-kernel stamps of the one-launch BasicBlock (python tools/build_variant.py clockb64 block64.hip -DPPN_CLOCK -fno-slp-vectorize):
per role, cycles in the convolution work, waiting for the LDS-DMA, at the phase barrier."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PPN_LIB", os.path.join(ROOT, "tools", "bin", "libppn_clockb64.so"))
import numpy as np, torch
from test_block64_gpu import _setup, _one_launch
from pytorch_pose_proposal_network_amd import lib as L_
B, H, W = 32, 96, 96
L, lib, dt, tdt, t, st = _setup("f16", B, H, W, 1)
raw = C.CDLL(os.environ["PPN_LIB"])
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
raw.ppn_block64_set_debug(C.c_void_p(dbg.data_ptr()))
for _ in range(5):
    _one_launch(L, lib, dt, tdt, t, st, B, H, W, True, True, True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
d = L.BlockDesc()
o_raw = torch.empty(B, H, W, 64, dtype=tdt, device="cuda"); o_act = torch.empty_like(o_raw)
d.dtype, d.batch, d.h, d.w, d.channels = dt, B, H, W, 64
d.src, d.residual = t["x_act"].data_ptr(), t["x_raw"].data_ptr()
d.weight1, d.scale_mid, d.shift_mid, d.act_mid = t["w1p"].data_ptr(), t["sm"].data_ptr(), t["bm"].data_ptr(), 1
d.weight2, d.out_raw, d.scale2, d.shift2, d.act2, d.out_act = t["w2p"].data_ptr(), o_raw.data_ptr(), t["s2"].data_ptr(), t["b2"].data_ptr(), 1, o_act.data_ptr()
for _ in range(20):
    L.check(lib.ppn_basicblock64_fused(C.byref(d), st))
e0.record()
for _ in range(20):
    L.check(lib.ppn_basicblock64_fused(C.byref(d), st))
e1.record(); torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) * 50:.1f} us (back to back, stamped build)")
v = dbg.cpu().numpy().reshape(256, 8, 8)
for role, name in ((0, "role 0 (conv1)"), (1, "role 1 (conv2)")):
    w = v[:, role * 4:(role + 1) * 4, :].reshape(-1, 8)
    n = w[:, 4].mean()
    print(f"{name}: tiles per workgroup {n:.1f}; per phase: work {np.median(w[:, 0]) / (n):8.0f} cycles, DMA wait {np.median(w[:, 1]) / (n + 1):6.0f}, "
          f"barrier {np.median(w[:, 2]) / (n + 1):6.0f}, issue {np.median(w[:, 3]) / n:5.0f}; kernel {np.median(w[:, 5]):8.0f} cycles")
