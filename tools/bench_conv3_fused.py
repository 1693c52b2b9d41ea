"""conv3 in its fused-decode form (arg-max keys + unary, no head tensor) at batch 32: PPN_LIB variants A/B."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import lib as L
names = sys.argv[1:] or ["default"]
B, H, cin, cout, uch, win = 32, 24, 512, 7605, 108, 441
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
kstep, _, korder, ktot, cpad = L.conv_tiling(L.PPN_BF16, cin, cout, 1)
x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cpad, ktot, device=dev) * 0.04).to(torch.bfloat16)
bias = torch.randn(cout, device=dev) * 0.1
unary = torch.empty(B, uch, H, H, device=dev); keys = torch.zeros(B, 17, H, H, dtype=torch.int64, device=dev)
zero = torch.zeros(64, device=dev)
d = L.ConvDesc()
d.dtype, d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = L.PPN_BF16, B, H, H, cin, H, H, cout
d.ksize, d.stride, d.dilation, d.pad, d.k_total, d.cout_pad, d.act1, d.out_nchw_f32 = 1, 1, 1, 0, ktot, cpad, 3, 1
d.src, d.weight, d.zero_page, d.shift1 = x.data_ptr(), w.data_ptr(), zero.data_ptr(), bias.data_ptr()
d.unary_out, d.argmax_keys, d.unary_channels, d.limb_window = unary.data_ptr(), keys.data_ptr(), uch, win
libs = []
for n in names:
    path = L.LIB_PATH if n == "default" else os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", f"libppn_{n}.so")
    lib = C.CDLL(path); lib.ppn_conv2d_fused.restype, lib.ppn_conv2d_fused.argtypes = L._SIGNATURES["ppn_conv2d_fused"]
    libs.append((n, lib))
for n, lib in libs:
    for _ in range(3): assert lib.ppn_conv2d_fused(C.byref(d), st) == 0
torch.cuda.synchronize()
for r in range(3):
    for n, lib in libs:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): lib.ppn_conv2d_fused(C.byref(d), st)
        e.record(); torch.cuda.synchronize()
        print(f"round {r} {n:10s} {a.elapsed_time(e) * 100:8.1f} us", flush=True)
