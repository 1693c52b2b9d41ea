"""Can an HBM-bound kernel run in the shadow of the dominant conv kernel (another stream, same CUs)?

Host: the 512->512 3x3 d2 48x48 conv at batch 32 (conv_igemm_big_kernel<bf16,192,256,8>: 112 KB LDS, 8 waves x 192 VGPRs
per CU -> 48 KB LDS and 128 VGPRs per SIMD lane left).  Guests on a second stream: (a) a torch copy (no LDS, few VGPRs),
(b) the 64->64 3x3 96x96 conv (128x64 tile: 48 KB LDS, 8 waves), (c) the fused stem (54.6 KB LDS, 235 VGPRs).
Prints each alone, both back to back on one stream, and both concurrently on two streams."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pytorch_pose_proposal_network_amd import lib as L

lib = L.load()
dev = torch.device("cuda")
zero = torch.zeros(64, device=dev)
B = 32


def conv_job(cin, cout, H, k, s, d):
    pad = d * (k - 1) // 2
    Ho = (H + 2 * pad - d * (k - 1) - 1) // s + 1
    kstep, _, korder, ktot, cpad = L.conv_tiling(L.PPN_BF16, cin, cout, k)
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cpad, ktot, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(B, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    d_ = L.ConvDesc()
    d_.dtype, d_.batch, d_.in_h, d_.in_w, d_.cin = L.PPN_BF16, B, H, H, cin
    d_.out_h, d_.out_w, d_.cout = Ho, Ho, cout
    d_.ksize, d_.stride, d_.dilation, d_.pad = k, s, d, pad
    d_.k_total, d_.cout_pad, d_.act1 = ktot, cpad, 1
    d_.src, d_.weight, d_.zero_page = x.data_ptr(), w.data_ptr(), zero.data_ptr()
    d_.scale1, d_.shift1, d_.out_raw = sc.data_ptr(), sh.data_ptr(), out.data_ptr()
    keep = (x, w, out, sc, sh)

    def run(stream):
        L.check(lib.ppn_conv2d_fused(C.byref(d_), stream.cuda_stream), "conv")
    run.keep = keep
    return run


def copy_job(mb):
    a = torch.empty(mb * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16).normal_()
    b = torch.empty_like(a)

    def run(stream):
        with torch.cuda.stream(stream):
            b.copy_(a)
    run.keep = (a, b)
    return run


def timed(jobs):
    """jobs: [(fn, stream, count)] all started together; returns wall ms until every stream is done."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = max(c for _, _, c in jobs)
    for i in range(n):                      # interleave the submissions so neither stream starts late
        for fn, st, c in jobs:
            if i < c:
                fn(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def main():
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    host = conv_job(512, 512, 48, 3, 1, 2)
    guests = {"copy 75 MB (torch)": (copy_job(75), 60), "conv 64->64 3x3 96x96 (128x64 tile)": (conv_job(64, 64, 96, 3, 1, 1), 60)}
    NH = 20
    for _ in range(3):
        host(sa)
    torch.cuda.synchronize()
    t_host = min(timed([(host, sa, NH)]) for _ in range(3))
    print(f"host alone: {NH} launches {t_host:.3f} ms ({t_host / NH * 1e3:.1f} us each)")
    for name, (g, ng) in guests.items():
        for _ in range(3):
            g(sb)
        torch.cuda.synchronize()
        t_g = min(timed([(g, sb, ng)]) for _ in range(3))
        ng2 = max(1, int(ng * t_host / t_g))           # about as long as the host batch
        t_g = min(timed([(g, sb, ng2)]) for _ in range(3))
        t_serial = min(timed([(host, sa, NH), (g, sa, ng2)]) for _ in range(3))
        t_both = min(timed([(host, sa, NH), (g, sb, ng2)]) for _ in range(3))
        print(f"{name}: alone {ng2} x {t_g / ng2 * 1e3:.1f} us = {t_g:.3f} ms | one stream {t_serial:.3f} ms | two streams "
              f"{t_both:.3f} ms  -> overlap hides {100 * (t_host + t_g - t_both) / min(t_host, t_g):.0f} % of the shorter job")


if __name__ == "__main__":
    main()
