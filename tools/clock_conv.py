"""Diagnostic: in-kernel shader clock of the large-tile conv kernel under sustained load (build with -DPPN_CLOCK:
python tools/build_variant.py clock conv_big.hip -DPPN_CLOCK): s_memtime / s_memrealtime around the K loop after 2 s of
back-to-back launches on random data, and the cycles one K step takes."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "csrc")
LIB = os.environ.get("PPN_LIB", os.path.join(ROOT, "tools", "bin", "libppn_clock.so"))


os.environ["PPN_LIB"] = LIB
import torch
from pytorch_pose_proposal_network_amd import lib as L
lib = L.load()
B, cin, cout, H, k, s, d = 32, 512, 512, 48, 3, 1, 2
HEAD = "--head" in sys.argv          # the fused head conv (512 -> 7605, 1x1, 24x24)
if HEAD:
    cin, cout, H, k, s, d = 512, 7605, 24, 1, 1, 1
dtype, tdt = (L.PPN_F16, torch.float16) if "--f16" in sys.argv else (L.PPN_BF16, torch.bfloat16)   # --f16: the IEEE-half instantiation
dev = torch.device("cuda")
pad = d * (k - 1) // 2
kstep, _, korder, ktot, cpad = L.conv_tiling(dtype, cin, cout, k)
x = torch.randn(B, H, H, cin, device=dev).to(tdt)
w = (torch.randn(cpad, ktot, device=dev) * 0.02).to(tdt)
out = torch.empty(B, H, H, cout, device=dev, dtype=tdt)
dbg = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
unary = torch.empty(B, 108, H, H, device=dev); keys = torch.zeros(B, 17, H, H, dtype=torch.int64, device=dev)
bias = torch.zeros(cout, device=dev)
zero = torch.zeros(64, device=dev)
dsc = L.ConvDesc()
dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = dtype, B, H, H, cin
dsc.out_h, dsc.out_w, dsc.cout = H, H, cout
dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, s, d, pad
dsc.k_total, dsc.cout_pad, dsc.act1, dsc.act2, dsc.out_nchw_f32 = ktot, cpad, 1, 0, 0
dsc.src, dsc.weight, dsc.zero_page, dsc.out_raw = x.data_ptr(), w.data_ptr(), zero.data_ptr(), out.data_ptr()
dsc.shift2 = dbg.data_ptr()          # diagnostic channel of the stamped build
if HEAD:
    dsc.act1, dsc.out_nchw_f32, dsc.out_raw = 3, 1, None
    dsc.shift1 = bias.data_ptr()
    dsc.unary_out, dsc.argmax_keys, dsc.unary_channels, dsc.limb_window = unary.data_ptr(), keys.data_ptr(), 108, 441
st = torch.cuda.current_stream().cuda_stream
import time
t0 = time.time()
n = 0
while time.time() - t0 < 2.5:
    for _ in range(50):
        L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
    torch.cuda.synchronize(); n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
t = dbg.cpu().numpy().reshape(-1, 8)
t = t[t[:, 1] > 0]
import numpy as np
clk = np.median(t[:, 0] / t[:, 1]) * 100e6
nsteps = ktot // kstep - 1
fl = 2.0 * B * H * H * cout * cin * k * k
print(("f16 " if "--f16" in sys.argv else "bf16 ") + f"launch {us:.1f} us = {fl / us / 1e6:.0f} TFLOP/s; in-kernel clock {clk / 1e9:.3f} GHz (median over {len(t)} waves); "
      f"K loop {np.median(t[:, 0]) / nsteps:.0f} cycles per step ({nsteps} steps), ideal MFMA time 1536; "
      f"peak at this clock {256 * 4096 * clk / 1e15:.3f} PFLOP/s")
print(f"per workgroup: prologue {np.median(t[:, 2]):.0f} cycles, K loop {np.median(t[:, 0]):.0f}, epilogue {np.median(t[:, 4]):.0f} "
      f"[prologue: loader state ready at {np.median(t[:, 5]):.0f}, stage 0 landed + barrier at {np.median(t[:, 6]):.0f}] "
      f"[epilogue: accumulators -> LDS {np.median(t[:, 7] >> 32):.0f}, LDS -> scale/act -> stores {np.median(t[:, 7] & 0xffffffff):.0f}] "
      f"(= {np.median(t[:, 2]) / clk * 1e6:.1f} / {np.median(t[:, 0]) / clk * 1e6:.1f} / {np.median(t[:, 4]) / clk * 1e6:.1f} us)")
