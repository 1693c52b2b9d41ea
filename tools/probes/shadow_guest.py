"""What can be co-resident with a workgroup of the dominant conv kernel?  Host: the 512->512 3x3 d2 48x48 conv (batch 32,
192x256 tile: 112 KB LDS, 8 waves x 192 VGPRs per CU) on stream A; guest: tools/probes/guest_copy.hip (HBM-bound copy with
a parameterised LDS allocation / workgroup size / register payload) on stream B.  Prints, per guest shape, the time of
both alone and together, and how much of the shorter job the overlap hides."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pytorch_pose_proposal_network_amd import lib as L
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from shadow_overlap import conv_job, timed   # noqa: E402  (runs that probe's own measurement first)

g = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libguest.so"))
g.guest_copy.restype = C.c_int
g.guest_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda")
src = torch.empty(75 * 1024 * 1024 // 4, device=dev, dtype=torch.int32).random_()
dst = torch.empty_like(src)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
host = conv_job(512, 512, 48, 3, 1, 2)
NH = 20
for _ in range(3):
    host(sa)
torch.cuda.synchronize()
t_host = min(timed([(host, sa, NH)]) for _ in range(3))
print(f"host alone: {t_host / NH * 1e3:.1f} us per launch")
for lds_kb, threads, regs in [(0, 256, 8), (16, 256, 8), (32, 256, 8), (40, 256, 8), (44, 256, 8), (46, 256, 8), (47, 256, 8),
                              (48, 256, 8), (32, 512, 8), (32, 256, 32), (32, 256, 64), (32, 512, 32), (0, 512, 64), (0, 256, 64)]:
    def guest(stream, lds_kb=lds_kb, threads=threads, regs=regs):
        rc = g.guest_copy(src.data_ptr(), dst.data_ptr(), src.numel() // 4, lds_kb * 1024, threads, regs, 2048,
                          stream.cuda_stream)
        assert rc == 0, rc
    for _ in range(3):
        guest(sb)
    torch.cuda.synchronize()
    t1 = min(timed([(guest, sb, 40)]) for _ in range(3)) / 40
    ng = max(1, int(t_host / t1))
    t_g = min(timed([(guest, sb, ng)]) for _ in range(3))
    t_both = min(timed([(host, sa, NH), (guest, sb, ng)]) for _ in range(3))
    print(f"guest LDS {lds_kb:2d} KB, {threads} threads, payload {regs:2d} VGPRs: alone {t1 * 1e3:6.1f} us x {ng} = {t_g:.3f} ms | "
          f"with host {t_both:.3f} ms (host alone {t_host:.3f}) -> hides {100 * (t_host + t_g - t_both) / min(t_host, t_g):4.0f} %")
