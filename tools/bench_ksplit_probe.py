"""What a hybrid split-K launch of the 48x48 512-wide layers could gain, measured BEFORE building it:
  base  512->512 3x3 d2 at 32x48x48 on the shipped 192x256 tile (768 workgroups = 3 rounds)
  A     the first 65 536 pixels on 256x256 tiles (512 workgroups = 2 whole rounds), ppn_conv_desc.m_count
  B'    a stand-in for the remaining 64 tiles cut 4 ways along K: 256 workgroups of ONE 256x256 tile with 18 K steps
        (128->512 3x3 at 32x32x32: same tile, same step count, same epilogue; no partial-sum hand-off)
hybrid estimate = A + B' + hand-off (publish 256 KB f32 per workgroup, last arriver reads 3 x 256 KB: ~8-10 us by the
guide's price list)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import lib as L

lib = L.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
zero = torch.zeros(64, device=dev)


def run(name, cin, cout, H, d, B, tile=None, m_count=0, k=3):
    pad = d * (k - 1) // 2
    kstep, _, korder, ktot, cpad = L.conv_tiling(L.PPN_BF16, cin, cout, k)
    x = torch.randn(B, H, H, cin, device=dev).bfloat16()
    w = (torch.randn(cpad, ktot, device=dev) * 0.02).bfloat16()
    out = torch.empty(B, H, H, cout, device=dev, dtype=torch.bfloat16)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    dsc = L.ConvDesc()
    dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = L.PPN_BF16, B, H, H, cin
    dsc.out_h, dsc.out_w, dsc.cout = H, H, cout
    dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, 1, d, pad
    dsc.k_total, dsc.cout_pad, dsc.act1 = ktot, cpad, 1
    dsc.src, dsc.weight, dsc.zero_page = x.data_ptr(), w.data_ptr(), zero.data_ptr()
    dsc.scale1, dsc.shift1, dsc.out_raw = sc.data_ptr(), sh.data_ptr(), out.data_ptr()
    dsc.m_begin, dsc.m_count = 0, m_count
    if tile:
        L.check(lib.ppn_set_conv_tile_override(*tile))
    try:
        for _ in range(5):
            L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record(); L.check(lib.ppn_conv2d_fused(C.byref(dsc), st)); b.record()
        torch.cuda.synchronize()
    finally:
        L.check(lib.ppn_set_conv_tile_override(0, 0))
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{name:60s} median {ms[10]*1e3:7.1f} us  min {ms[0]*1e3:7.1f} us  {lib.ppn_last_conv_kernel().decode()}", flush=True)
    return ms[10] * 1e3


for rnd in range(2):
    base = run("base: 512->512 d2 48x48 B32, automatic tile", 512, 512, 48, 2, 32)
    a = run("A: first 65536 px on 256x256 (2 rounds)", 512, 512, 48, 2, 32, (256, 256), 65536)
    b = run("B': 256 workgroups x one 256x256 tile x 18 K steps", 128, 512, 32, 2, 32, (256, 256))
    full = run("256x256 on all 73728 px (2.25 rounds)", 512, 512, 48, 2, 32, (256, 256))
    print(f"round {rnd}: base {base:.1f}  A + B' = {a + b:.1f} (+ hand-off ~8-10)  256x256 everywhere {full:.1f}", flush=True)
