"""Diagnostic: where a depth step of the weight-gradient kernel (csrc/wgrad.hip) spends its cycles.  Build with
    python tools/build_variant.py clockwg wgrad.hip -DPPN_CLOCK
and run this: s_memtime stamps per wave summed over its depth steps -- wait for the staged tile (vmcnt), barrier,
issue of the next stage, transposed LDS reads + MFMA -- and the in-kernel clock."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPN_LIB"] = os.environ.get("PPN_LIB", os.path.join(ROOT, "tools", "bin", "libppn_clockwg.so"))
import numpy as np
import torch
from pytorch_pose_proposal_network_amd import train as T

CASES = [("layer6 3x3 512->512 d4 @48", 32, 512, 512, 48, 3, 1, 4, 4, 8),
         ("layer5 3x3 256->256 d2 @48", 32, 256, 256, 48, 3, 1, 2, 2, 8),
         ("layer4 3x3 128->128 @48", 32, 128, 128, 48, 3, 1, 1, 1, 4),
         ("head block 3x3 512->512 @24", 32, 512, 512, 24, 3, 1, 1, 1, 8)]
dev = torch.device("cuda")
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[0]]
for name, B, ci, co, H, k, s, d, p, nw in CASES:
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    x = torch.randn(B, H, H, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, Ho, Ho, co, device=dev).to(torch.bfloat16)
    out = torch.empty(co, ci, k, k, device=dev)
    for _ in range(20):
        T.conv_wgrad(x, dy, k, s, d, p, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        T.conv_wgrad(x, dy, k, s, d, p, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    ws = list(T._wgrad_cache.values())[0]
    tiles = (co + (255 if nw == 8 else 127)) // (256 if nw == 8 else 128) * ((ci + (255 if nw == 8 else 127)) // (256 if nw == 8 else 128)) * k * k
    # the stamps sit behind the partials: find them from the workspace size the library asked for
    import ctypes as C
    from pytorch_pose_proposal_network_amd import lib as L
    dd = L.WgradDesc()
    dd.dtype, dd.batch, dd.in_h, dd.in_w, dd.cin = L.PPN_BF16, B, H, H, ci
    dd.out_h, dd.out_w, dd.cout, dd.ksize, dd.stride, dd.dilation, dd.pad = Ho, Ho, co, k, s, d, p
    need = L.load().ppn_conv_wgrad_workspace_bytes(C.byref(dd))
    # need = nsplit * (taps*cout*cin*4 + tiles*512)
    nsplit = need // (k * k * co * ci * 4 + tiles * 512)
    part = nsplit * k * k * co * ci * 4
    t = ws[part:part + tiles * nsplit * nw * 64].cpu().numpy().view(np.int64).reshape(-1, 8)
    t = t[t[:, 4] > 0]
    clk = np.median(t[:, 5] / t[:, 6]) * 100e6
    per = t[:, :4] / t[:, 4:5]
    fl = 2.0 * B * Ho * Ho * co * ci * k * k
    print(f"{name}: launch+fold {us:.1f} us = {fl / us / 1e6:.0f} TFLOP/s; clock {clk / 1e9:.3f} GHz; {tiles} tiles x {nsplit} splits, "
          f"{np.median(t[:, 4]):.0f} steps per workgroup, kernel {np.median(t[:, 5]):.0f} cycles per workgroup")
    if nw == 8:     # ping-pong loop: 32-pixel steps
        print("   per 32-pixel step (median over waves): read phase %.0f, barrier %.0f, MFMA phase %.0f, barrier %.0f cycles "
              "(MFMA issue alone: 512 per wave)" % tuple(np.median(per, axis=0)))
    else:
        print("   per 64-pixel step (median over waves): DMA wait %.0f, barrier %.0f, issue next %.0f, LDS reads + MFMA %.0f cycles"
              % tuple(np.median(per, axis=0)))
