"""Diagnostic: cycle stamps of csrc/conv64.hip (python tools/build_variant.py clock64 conv64.hip -DPPN_CLOCK)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPN_LIB"] = os.environ.get("PPN_LIB", os.path.join(ROOT, "tools", "bin", "libppn_clock64.so"))
import numpy as np, torch
from pytorch_pose_proposal_network_amd import lib as L
lib = L.load()
B, H = 32, 96
dt, tdt = L.PPN_BF16, torch.bfloat16
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
kstep, _, korder, ktot, cpad = L.conv_tiling(dt, 64, 64, 3)
x = torch.randn(B, H, H, 64, device=dev).to(tdt); w = (torch.randn(cpad, ktot, device=dev) * 0.06).to(tdt)
res = torch.randn(B, H, H, 64, device=dev).to(tdt); o1, o2 = torch.empty_like(x), torch.empty_like(x)
sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.1
dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
for form in ("single", "residual+dual"):
    d = L.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = dt, B, H, H, 64, H, H, 64
    d.ksize, d.stride, d.dilation, d.pad, d.k_total, d.cout_pad = 3, 1, 1, 1, ktot, cpad
    d.src, d.weight, d.zero_page, d.out_raw = x.data_ptr(), w.data_ptr(), dbg.data_ptr(), o1.data_ptr()
    if form == "single": d.scale1, d.shift1, d.act1 = sc.data_ptr(), sh.data_ptr(), 1
    else: d.residual, d.scale2, d.shift2, d.act2, d.out_act = res.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, o2.data_ptr()
    for _ in range(20): L.check(lib.ppn_conv2d_fused(C.byref(d), st))
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(-1, 8); t = t[t[:, 4] > 0]
    per = t[:, 1:4] / t[:, 4:5]
    print(f"{form}: {len(t)} waves, tiles/WG {np.median(t[:,4]):.1f}; weights+first patch {np.median(t[:,0]):.0f} cycles; per tile: wait+barrier "
          f"{np.median(per[:,0]):.0f}, MFMA loop {np.median(per[:,1]):.0f} (MFMA alone 2304), epilogue {np.median(per[:,2]):.0f}; kernel {np.median(t[:,5]):.0f} cycles")
