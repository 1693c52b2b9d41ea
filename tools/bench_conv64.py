"""conv64.hip vs the generic kernel on the three 64 -> 64 layer3 launches at batch 32 (96x96), interleaved rounds."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import lib as L
lib = L.load()
B, H = 32, 96
dt, tdt = L.PPN_BF16, torch.bfloat16
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
kstep, _, korder, ktot, cpad = L.conv_tiling(dt, 64, 64, 3)
x = torch.randn(B, H, H, 64, device=dev).to(tdt)
w = (torch.randn(cpad, ktot, device=dev) * 0.06).to(tdt)
res = torch.randn(B, H, H, 64, device=dev).to(tdt)
o1, o2 = torch.empty_like(x), torch.empty_like(x)
sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.1
zero = torch.zeros(64, device=dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)       # cache flusher
def desc(form):
    d = L.ConvDesc()
    d.dtype, d.batch, d.in_h, d.in_w, d.cin, d.out_h, d.out_w, d.cout = dt, B, H, H, 64, H, H, 64
    d.ksize, d.stride, d.dilation, d.pad, d.k_total, d.cout_pad = 3, 1, 1, 1, ktot, cpad
    d.src, d.weight, d.zero_page, d.out_raw = x.data_ptr(), w.data_ptr(), zero.data_ptr(), o1.data_ptr()
    if form == "single":
        d.scale1, d.shift1, d.act1 = sc.data_ptr(), sh.data_ptr(), 1
    else:
        d.residual, d.scale2, d.shift2, d.act2, d.out_act = res.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, o2.data_ptr()
    return d
for form in ("single", "residual+dual"):
    d = desc(form)
    for r in range(3):
        for on in (1, 0):
            lib.ppn_set_conv64_enabled(on)
            for cold in (False, True):
                ts = []
                for _ in range(5):
                    if cold: big.zero_()
                    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); L.check(lib.ppn_conv2d_fused(C.byref(d), st)); e.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(e) * 1e3)
                print(f"{form:14s} round {r} conv64={on} {'cold' if cold else 'warm'}: {sorted(ts)[2]:7.1f} us  ({lib.ppn_last_conv_kernel().decode()})", flush=True)
lib.ppn_set_conv64_enabled(1)
