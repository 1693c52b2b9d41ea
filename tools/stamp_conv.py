"""Diagnostic: build libppn with in-kernel s_memtime stamps (-DPPN_STAMP) and print where one wave of the
large-tile conv kernel spends a K step (shares only; the stamped build is slower than the real one)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "pytorch_pose_proposal_network_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "bin", "libppn_stamp.so")

def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    from pytorch_pose_proposal_network_amd import build as B
    srcs = [s_ for s_, _ in B.SOURCES]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DPPN_STAMP", "-shared", "-o", LIB]
    for s in srcs:
        cmd += (["-x", "hip"] if s.endswith(".cpp") else []) + [os.path.join(CSRC, s)]
    import importlib.util
    tl = os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib")
    cmd += ["-L" + tl, "-Wl,-rpath," + tl]
    subprocess.check_call(cmd)

if "--build" in sys.argv:
    build(); sys.exit(0)

os.environ["PPN_LIB"] = LIB
import torch
from pytorch_pose_proposal_network_amd import lib as L
lib = L.load()
B, cin, cout, H, k, s, d = 32, 512, 512, 48, 3, 1, 2
HEAD = "--head" in sys.argv          # the fused head conv (512 -> 7605, 1x1, 24x24)
if HEAD:
    cin, cout, H, k, s, d = 512, 7605, 24, 1, 1, 1
dtype, tdt = L.PPN_BF16, torch.bfloat16
dev = torch.device("cuda")
pad = d * (k - 1) // 2
kstep, _, korder, ktot, cpad = L.conv_tiling(dtype, cin, cout, k)
x = torch.randn(B, H, H, cin, device=dev).to(tdt)
w = (torch.randn(cpad, ktot, device=dev) * 0.02).to(tdt)
out = torch.empty(B, H, H, cout, device=dev, dtype=tdt)
dbg = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
unary = torch.empty(B, 108, H, H, device=dev); keys = torch.zeros(B, 17, H, H, dtype=torch.int64, device=dev)
bias = torch.zeros(cout, device=dev)
zero = torch.zeros(64, device=dev)
dsc = L.ConvDesc()
dsc.dtype, dsc.batch, dsc.in_h, dsc.in_w, dsc.cin = dtype, B, H, H, cin
dsc.out_h, dsc.out_w, dsc.cout = H, H, cout
dsc.ksize, dsc.stride, dsc.dilation, dsc.pad = k, s, d, pad
dsc.k_total, dsc.cout_pad, dsc.act1, dsc.act2, dsc.out_nchw_f32 = ktot, cpad, 1, 0, 0
dsc.src, dsc.weight, dsc.zero_page, dsc.out_raw = x.data_ptr(), w.data_ptr(), zero.data_ptr(), out.data_ptr()
dsc.shift2 = dbg.data_ptr()          # diagnostic channel of the stamped build
if HEAD:
    dsc.act1, dsc.out_nchw_f32, dsc.out_raw = 3, 1, None
    dsc.shift1 = bias.data_ptr()
    dsc.unary_out, dsc.argmax_keys, dsc.unary_channels, dsc.limb_window = unary.data_ptr(), keys.data_ptr(), 108, 441
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.check(lib.ppn_conv2d_fused(C.byref(dsc), st))
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(-1, 8)
t = t[t.sum(1) > 0]
print("epilogue cycles per wave: mean", int(t[:, 4].mean()), " main-loop cycles per wave:", int(t[:, :4].sum(1).mean()))
t = t[:, :4]
nsteps = ktot // kstep - 1
print("waves", len(t), "steps", nsteps)
names = ["batch1 (MFMA+DMA issue+reads A)", "batch2 (MFMA+reads B)", "s_waitcnt vmcnt/lgkmcnt", "barrier"]
tot = t.sum(1).mean() / nsteps
for i, n in enumerate(names):
    v = t[:, i] / nsteps
    print(f"{n:34s} mean {v.mean():8.0f} cyc  p10 {sorted(v)[len(v)//10]:8.0f}  p90 {sorted(v)[len(v)*9//10]:8.0f}   {100*v.mean()/tot:5.1f} %")
print(f"per step total {tot:.0f} cycles (s_memtime ticks = shader cycles)")
