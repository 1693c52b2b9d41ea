"""Diagnostic: where a tile of the persistent head kernel (csrc/conv_head.hip) spends its cycles.  Build with
    python tools/build_variant.py clockhead conv_head.hip -DPPN_CLOCK
and run this: s_memtime stamps per workgroup and wave, summed over its tiles -- wait for the first stage, K loop,
issue of the next tile's stages, arg-max epilogue -- and the in-kernel clock."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPN_LIB"] = os.environ.get("PPN_LIB", os.path.join(ROOT, "tools", "bin", "libppn_clockhead.so"))
import numpy as np
import torch
from pytorch_pose_proposal_network_amd import lib as L
lib = L.load()
B, H, cin, win, E, ep = 32, 24, 512, 441, 17, 448
dtype, tdt = L.PPN_BF16, torch.bfloat16
dev = torch.device("cuda")
kstep, _, korder, ktot, _ = L.conv_tiling(dtype, cin, 512, 1)
x = torch.randn(B, H, H, cin, device=dev).to(tdt)
w = (torch.randn(E * ep, ktot, device=dev) * 0.04).to(tdt)
bias = torch.randn(E * ep, device=dev) * 0.1
keys = torch.zeros(B, E, H, H, dtype=torch.int64, device=dev)
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
zero = torch.zeros(64, device=dev)
d = L.ConvDesc()
d.dtype, d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = dtype, B, H, H, cin, H, H, E * win
d.ksize, d.stride, d.dilation, d.pad, d.k_total, d.cout_pad, d.act1, d.out_nchw_f32 = 1, 1, 1, 0, ktot, E * ep, 3, 1
d.src, d.weight, d.zero_page, d.shift1 = x.data_ptr(), w.data_ptr(), zero.data_ptr(), bias.data_ptr()
d.argmax_keys, d.limb_window, d.limb_edge_pad = keys.data_ptr(), win, ep
d.shift2 = dbg.data_ptr()
st = torch.cuda.current_stream().cuda_stream
import time
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        L.check(lib.ppn_conv2d_fused(C.byref(d), st))
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.ppn_conv2d_fused(C.byref(d), st))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
t = dbg.cpu().numpy().reshape(-1, 8)
t = t[t[:, 4] > 0]
clk = np.median(t[:, 5] / t[:, 6]) * 100e6
per = t[:, :4] / t[:, 4:5]
fl = 2.0 * B * H * H * E * win * cin
print(f"launch {us:.1f} us = {fl / us / 1e6:.0f} TFLOP/s; in-kernel clock {clk / 1e9:.3f} GHz; {len(t)} waves, "
      f"{np.median(t[:, 4]):.0f} tiles per workgroup (median), kernel {np.median(t[:, 5]):.0f} cycles")
print("per tile (median over waves): wait for stage 0 %.0f, K loop %.0f (8 steps; MFMA alone 14336), issue next %.0f, "
      "epilogue %.0f cycles" % tuple(np.median(per, axis=0)))
