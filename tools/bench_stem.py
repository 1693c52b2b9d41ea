"""Micro-benchmark of the one-launch stem (stem012) at batch 32, 384x384 u8 frames, in the variant the bf16 plan runs:
IEEE-half internals with the exact integer input, bf16 outputs (PPN_STEM_IO(PPN_F16, PPN_BF16)); STEM_DTYPE=bf16 times the
all-bf16 variant.  Arguments: names of diagnostic builds under tools/bin (tools/build_variant.py NAME stem012.hip
-DPPN_S012_SKIP=mask), "default" = the in-tree library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import lib as L, prng

def main():
    names = sys.argv[1:] or ["default"]
    B, H, W = 32, 384, 384
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    frames = torch.from_numpy(prng.u8_frames(1, B, (H, W))).to(dev)
    g = torch.Generator().manual_seed(0)
    w0 = (torch.randn(16, 3, 7, 7, generator=g) * 0.002).to(dev); w1 = (torch.randn(16, 16, 3, 3, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(32, 16, 3, 3, generator=g) * 0.1).to(dev)
    s = [torch.rand(n, generator=g).to(dev) + 0.5 for n in (16, 16, 32, 32)]
    b = [torch.randn(n, generator=g).to(dev) * 0.3 for n in (16, 16, 32, 32)]
    m3, s3 = (C.c_float * 3)(0.485, 0.456, 0.406), (C.c_float * 3)(0.229, 0.224, 0.225)
    raw = torch.empty(B, 192, 192, 32, dtype=torch.bfloat16, device=dev); act = torch.empty_like(raw)
    libs = []
    for n in names:
        path = L.LIB_PATH if n == "default" else os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", f"libppn_{n}.so")
        lib = C.CDLL(path)
        lib.ppn_stem012_dt.restype, lib.ppn_stem012_dt.argtypes = L._SIGNATURES["ppn_stem012_dt"]
        libs.append((n, lib))
    dt = L.PPN_BF16 if os.environ.get("STEM_DTYPE") == "bf16" else (L.PPN_F16 | ((L.PPN_BF16 + 1) << 8))
    def run(lib):
        rc = lib.ppn_stem012_dt(dt, 1, frames.data_ptr(), B, H, W, w0.data_ptr(), s[0].data_ptr(), b[0].data_ptr(), m3, s3,
                                w1.data_ptr(), s[1].data_ptr(), b[1].data_ptr(), w2.data_ptr(), s[2].data_ptr(), b[2].data_ptr(),
                                s[3].data_ptr(), b[3].data_ptr(), raw.data_ptr(), act.data_ptr(), st)
        assert rc == 0
    for n, lib in libs:
        for _ in range(3): run(lib)
    torch.cuda.synchronize()
    for r in range(3):
        for n, lib in libs:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): run(lib)
            e.record(); torch.cuda.synchronize()
            print(f"round {r} {n:12s} {a.elapsed_time(e) * 100:8.1f} us", flush=True)
main()
