"""In-process A/B of libppn conv variants (cdna_hip_programming.md rule 24: interleaved rounds in ONE process).

    python tools/build_variant.py split conv_big.hip -DPPN_DMA_SPLIT=1
    python tools/ab_conv_inproc.py default split [--shapes "L7 512" "B2 512"] [--rounds 7]

Every variant is its own CDLL (own copy of the code object); per shape the variants are timed round-robin,
`--rounds` rounds of 10 back-to-back launches each between two HIP events; median and min over the rounds."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorch_pose_proposal_network_amd import lib as L

SHAPES = [  # name, Cin, Cout, H, k, stride, dil
    ("L7 512->512 d2 48", 512, 512, 48, 3, 1, 2),
    ("L6 512->512 d4 48", 512, 512, 48, 3, 1, 4),
    ("L6.0 256->512 d4 48", 256, 512, 48, 3, 1, 4),
    ("L5 256->256 d2 48", 256, 256, 48, 3, 1, 2),
    ("L4 128->128 48", 128, 128, 48, 3, 1, 1),
    ("L3 64->64 96", 64, 64, 96, 3, 1, 1),
    ("B2 512->512 24", 512, 512, 24, 3, 1, 1),
    ("conv3 512->7605 1x1 24", 512, 7605, 24, 1, 1, 1),
]


def load(name):
    path = L.LIB_PATH if name == "default" else os.path.join(ROOT, "tools", "bin", f"libppn_{name}.so")
    lib = C.CDLL(path)
    for fn in ("ppn_conv2d_fused", "ppn_conv_tiling", "ppn_last_error", "ppn_last_conv_kernel"):
        res, args = L._SIGNATURES[fn]
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--shapes", nargs="*", default=None)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    import torch  # noqa: F811
    libs = [(v, load(v)) for v in args.variants]
    dev = torch.device("cuda")
    zero = torch.zeros(64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    B = args.batch
    for name, cin, cout, H, k, s, d in SHAPES:
        if args.shapes and not any(o in name for o in args.shapes):
            continue
        pad = d * (k - 1) // 2
        Ho = (H + 2 * pad - d * (k - 1) - 1) // s + 1
        kstep, _, korder, ktot, cpad = L.conv_tiling(L.PPN_BF16, cin, cout, k)
        x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(cpad, ktot, device=dev) * 0.02).to(torch.bfloat16)
        nchw = cout == 7605
        out = torch.empty(B, cout, Ho, Ho, device=dev) if nchw else torch.empty(B, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
        sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
        dsc = L.ConvDesc()
        dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = L.PPN_BF16, B, H, H, cin
        dsc.out_h, dsc.out_w, dsc.cout = Ho, Ho, cout
        dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, s, d, pad
        dsc.k_total, dsc.cout_pad, dsc.act1, dsc.act2, dsc.out_nchw_f32 = ktot, cpad, (3 if nchw else 1), 0, int(nchw)
        dsc.src, dsc.weight, dsc.zero_page = x.data_ptr(), w.data_ptr(), zero.data_ptr()
        dsc.scale1, dsc.shift1, dsc.out_raw = sc.data_ptr(), sh.data_ptr(), out.data_ptr()
        ref = None
        times = {v: [] for v, _ in libs}
        for v, lib in libs:                                      # warm-up + results must agree bit for bit
            for _ in range(3):
                assert lib.ppn_conv2d_fused(C.byref(dsc), st) == 0, lib.ppn_last_error()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(out, ref), f"{v}: output differs from {libs[0][0]} on {name}"
        for r in range(args.rounds):
            for v, lib in libs:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    lib.ppn_conv2d_fused(C.byref(dsc), st)
                b.record()
                torch.cuda.synchronize()
                times[v].append(a.elapsed_time(b) / 10)
        fl = 2.0 * B * Ho * Ho * cout * cin * k * k
        line = f"{name:24s}"
        for v, _ in libs:
            t = sorted(times[v])
            line += f" | {v}: med {t[len(t)//2]*1e3:7.1f} us min {t[0]*1e3:7.1f} ({fl/t[len(t)//2]/1e9:6.0f} TF)"
        print(line, flush=True)


main()
