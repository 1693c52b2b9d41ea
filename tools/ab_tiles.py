"""In-process comparison of forced conv tiles (ppn_set_conv_tile_override) on the batch-32 layer shapes.

    python tools/ab_tiles.py "192,128" "192,256" "128,128" [--shapes L4 B2]
Prints, per shape, the automatic choice and every forced tile that the layer's Cout class admits."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pytorch_pose_proposal_network_amd import lib as L

SHAPES = [  # name, Cin, Cout, H, k, stride, dil
    ("L7 512->512 d2 48", 512, 512, 48, 3, 1, 2),
    ("L6.0 256->512 d4 48", 256, 512, 48, 3, 1, 4),
    ("L5 256->256 d2 48", 256, 256, 48, 3, 1, 2),
    ("L4 128->128 48", 128, 128, 48, 3, 1, 1),
    ("L4.0 64->128 s2 96", 64, 128, 96, 3, 2, 1),
    ("B2 512->512 24", 512, 512, 24, 3, 1, 1),
    ("B1 512->512 s2 48", 512, 512, 48, 3, 2, 1),
    ("neck 1x1 512->128 24", 512, 128, 24, 1, 1, 1),
    ("neck 3x3 128->128 24", 128, 128, 24, 3, 1, 1),
    ("neck 1x1 128->512 24", 128, 512, 24, 1, 1, 1),
    ("ds 1x1 256->512 48", 256, 512, 48, 1, 1, 1),
    ("ds 1x1 512->512 s2 48", 512, 512, 48, 1, 2, 1),
    ("ds 1x1 128->256 48", 128, 256, 48, 1, 1, 1),
    ("ds 1x1 64->128 s2 96", 64, 128, 96, 1, 2, 1),
]
ap = argparse.ArgumentParser()
ap.add_argument("tiles", nargs="*", default=["192,128", "192,256", "128,128"])
ap.add_argument("--shapes", nargs="*", default=None)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
lib = L.load()
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
    out = torch.empty(B, Ho, Ho, cout, device=dev, dtype=torch.bfloat16)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    dsc = L.ConvDesc()
    dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = L.PPN_BF16, B, H, H, cin
    dsc.out_h, dsc.out_w, dsc.cout = Ho, Ho, cout
    dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, s, d, pad
    dsc.k_total, dsc.cout_pad, dsc.act1 = ktot, cpad, 1
    dsc.src, dsc.weight, dsc.zero_page = x.data_ptr(), w.data_ptr(), zero.data_ptr()
    dsc.scale1, dsc.shift1, dsc.out_raw = sc.data_ptr(), sh.data_ptr(), out.data_ptr()
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    line, ref = f"{name:22s}", None
    for tile in ["0,0"] + args.tiles:
        bp, bc = (int(v) for v in tile.split(","))
        if bc > (256 if cout >= 256 else (128 if cout >= 128 else 64)) or (cout >= 256 and 0 < bc < 128):
            continue
        L.check(lib.ppn_set_conv_tile_override(bp, bc), "override")
        try:
            for _ in range(3):
                L.check(lib.ppn_conv2d_fused(C.byref(dsc), st), "conv")
            torch.cuda.synchronize()
            kn = lib.ppn_last_conv_kernel().decode()
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(out, ref), (name, tile)
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    lib.ppn_conv2d_fused(C.byref(dsc), st)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 10)
            t = sorted(ts)[len(ts) // 2]
            tag = kn.split("<")[1].split(">")[0].replace("__bf16, ", "").replace(", false", "") if bp == 0 else tile
            line += f" | {'auto=' if bp == 0 else ''}{tag}: {t * 1e3:6.1f} us {fl / t / 1e9:5.0f} TF"
        finally:
            L.check(lib.ppn_set_conv_tile_override(0, 0), "override")
    print(line, flush=True)
