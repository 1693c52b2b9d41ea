"""Micro-benchmark of ppn_conv2d_fused on the DRN-D-22 layer shapes (batch 32, bf16 unless --f32)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import lib as L

SHAPES = [  # name, Cin, Cout, H, k, stride, dil
    ("L6 512->512 d4 48", 512, 512, 48, 3, 1, 4),
    ("L7 512->512 d2 48", 512, 512, 48, 3, 1, 2),
    ("L6.0 256->512 d4 48", 256, 512, 48, 3, 1, 4),
    ("L5 256->256 d2 48", 256, 256, 48, 3, 1, 2),
    ("L5.0 128->256 d2 48", 128, 256, 48, 3, 1, 2),
    ("L4 128->128 48", 128, 128, 48, 3, 1, 1),
    ("L4.0 64->128 s2 96", 64, 128, 96, 3, 2, 1),
    ("L3 64->64 96", 64, 64, 96, 3, 1, 1),
    ("B1 512->512 s2 48", 512, 512, 48, 3, 2, 1),
    ("B2 512->512 24", 512, 512, 24, 3, 1, 1),
    ("ds 256->512 1x1 48", 256, 512, 48, 1, 1, 1),
    ("conv3 512->7605 1x1 24", 512, 7605, 24, 1, 1, 1),
    ("L1 16->16 384", 16, 16, 384, 3, 1, 1),
    ("L2 16->32 s2 384", 16, 32, 384, 3, 2, 1),
]

def main():
    lib = L.load()
    dtype = L.PPN_F32 if "--f32" in sys.argv else L.PPN_BF16
    tdt = torch.float32 if dtype == L.PPN_F32 else torch.bfloat16
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = 32
    dev = torch.device("cuda")
    zero = torch.zeros(64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for name, cin, cout, H, k, s, d in SHAPES:
        if only and not any(o in name for o in only):
            continue
        pad = d * (k - 1) // 2
        Ho = (H + 2 * pad - d * (k - 1) - 1) // s + 1
        kstep, _, korder, ktot, cpad = L.conv_tiling(dtype, cin, cout, k)
        x = torch.randn(B, H, H, cin, device=dev).to(tdt)
        w = (torch.randn(cpad, ktot, device=dev) * 0.02).to(torch.float32 if korder == 2 else tdt)
        nchw = cout == 7605
        out = torch.empty(B, cout, Ho, Ho, device=dev) if nchw else torch.empty(B, Ho, Ho, cout, device=dev, dtype=tdt)
        sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
        dsc = L.ConvDesc()
        dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = dtype, B, H, H, cin
        dsc.out_h, dsc.out_w, dsc.cout = Ho, Ho, cout
        dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, s, d, pad
        dsc.k_total, dsc.cout_pad, dsc.act1, dsc.act2, dsc.out_nchw_f32 = ktot, cpad, (3 if nchw else 1), 0, int(nchw)
        dsc.src, dsc.weight, dsc.zero_page = x.data_ptr(), w.data_ptr(), zero.data_ptr()
        dsc.scale1, dsc.shift1, dsc.out_raw = sc.data_ptr(), sh.data_ptr(), out.data_ptr()
        for _ in range(3):
            L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in evs:
            a.record(); L.check(lib.ppn_conv2d_fused(C.byref(dsc), st)); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        fl = 2.0 * B * Ho * Ho * cout * cin * k * k
        print(f"{name:26s} median {ms[5]*1e3:8.1f} us  min {ms[0]*1e3:8.1f} us  {fl/ms[5]/1e9:8.1f} TFLOP/s", flush=True)

main()
