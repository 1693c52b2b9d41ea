"""Build a variant libppn into tools/bin/: python tools/build_variant.py NAME SOURCE.hip -DFOO=1 ...
Only SOURCE is recompiled with the extra flags; the other objects come from the in-tree build."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_pose_proposal_network_amd import build as B
B.build(verbose=False)
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
bindir = os.path.join(ROOT, "tools", "bin")
os.makedirs(bindir, exist_ok=True)
out = os.path.join(bindir, f"libppn_{name}.so")
obj = os.path.join(bindir, f"{name}_{os.path.splitext(src)[0]}.o")
extra = dict(B.SOURCES)[src]
subprocess.check_call([B.HIPCC] + B.COMMON + list(extra) + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o") for s, _ in B.SOURCES]
tl = B.torch_lib_dir()
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-L" + tl, "-Wl,-rpath," + tl])
os.remove(obj)
print(out)
