"""Probe: which operand of a convolution launch is COLD in sequence?  One launch of a layer shape is timed
(HIP events) after the caches were flushed by streaming 1 GB, with (a) nothing warmed, (b) only the packed weights touched
(a reduction kernel reads them: they are then in the Infinity Cache), (c) only the input touched, (d) both, and (e) back to
back (the launch repeated).  In the forward pass the input was written by the previous launch and the weights were last read
one whole pass (> 1 GB of traffic) ago."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorch_pose_proposal_network_amd import lib as L
lib = L.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
flush_src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()       # 1 GiB
flush_dst = torch.empty_like(flush_src)


def run(name, B, cin, cout, H, k, d):
    dtype, tdt = L.PPN_BF16, torch.bfloat16
    pad = d * (k - 1) // 2
    kstep, _, korder, ktot, cpad = L.conv_tiling(dtype, cin, cout, k)
    x = torch.randn(B, H, H, cin, device=dev).to(tdt)
    w = (torch.randn(cpad, ktot, device=dev) * 0.02).to(tdt)
    out = torch.empty(B, H, H, cout, device=dev, dtype=tdt)
    zero = torch.zeros(64, device=dev)
    dsc = L.ConvDesc()
    dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = dtype, B, H, H, cin
    dsc.out_h, dsc.out_w, dsc.cout = H, H, cout
    dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, 1, d, pad
    dsc.k_total, dsc.cout_pad, dsc.act1, dsc.act2, dsc.out_nchw_f32 = ktot, cpad, 1, 0, 0
    dsc.src, dsc.weight, dsc.zero_page, dsc.out_raw = x.data_ptr(), w.data_ptr(), zero.data_ptr(), out.data_ptr()
    dsc.flags = L.PPN_CONV_NO_FILTER_BANK | L.PPN_CONV_SHARED_GPU

    def once(warm_w, warm_x, warm_by_write=False):
        flush_dst.copy_(flush_src)                      # 2 GiB of traffic: Infinity Cache and L2 hold none of the operands
        if warm_x:
            if warm_by_write:
                x.copy_(x.clone())                      # the input as the PREVIOUS launch leaves it: freshly written
            else:
                x.float().sum()
        if warm_w:
            w.float().sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    def med(*a):
        v = sorted(once(*a) for _ in range(7))
        return v[3]
    for _ in range(30):
        L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
    e1.record(); torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:28s} cold {med(False, False):7.1f}  weights warm {med(True, False):7.1f}  input warm (read) {med(False, True):7.1f}  "
          f"input warm (written) {med(False, True, True):7.1f}  both {med(True, True):7.1f}  written+weights {med(True, True, True):7.1f}  back to back {b2b:7.1f} us", flush=True)


run("24x24 512->512 3x3", 32, 512, 512, 24, 3, 1)
run("48x48 512->512 3x3 d2", 32, 512, 512, 48, 3, 2)
run("48x48 256->256 3x3 d2", 32, 256, 256, 48, 3, 2)
run("24x24 512->128 1x1", 32, 512, 128, 24, 1, 1)
run("48x48 128->128 3x3", 32, 128, 128, 48, 3, 1)
