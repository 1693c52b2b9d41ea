"""Build tools/bin/libppn_NAME.so with EXTRA FLAGS on several sources: python tools/build_variant_all.py NAME "src1.hip,src2.hip" -flag ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_pose_proposal_network_amd import build as B
B.build(verbose=False)
name, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
bindir = os.path.join(ROOT, "tools", "bin"); os.makedirs(bindir, exist_ok=True)
procs, objs = [], []
for s, extra in B.SOURCES:
    if s in srcs:
        obj = os.path.join(bindir, f"{name}_{os.path.splitext(s)[0]}.o")
        procs.append(subprocess.Popen([B.HIPCC] + B.COMMON + list(extra) + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj]))
    else:
        obj = os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o")
    objs.append(obj)
assert all(p.wait() == 0 for p in procs)
tl = B.torch_lib_dir()
out = os.path.join(bindir, f"libppn_{name}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-L" + tl, "-Wl,-rpath," + tl])
print(out)
