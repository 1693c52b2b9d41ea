"""BatchNorm forward / backward (train mode) on the D-22 tensor shapes at batch 32, bf16: time per call and the HBM rate of
its algorithmic bytes (forward: 2 reads + 1 write of the tensor; backward: 4 reads + 1 write)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import train as T
dev = torch.device("cuda")
for name, hw, c in (("layer0/1 16ch @384", 384, 16), ("layer2 32ch @192", 192, 32), ("layer3 64ch @96", 96, 64),
                    ("layer4 128ch @48", 48, 128), ("layer5 256ch @48", 48, 256), ("layer6-8 512ch @48", 48, 512),
                    ("head 512ch @24", 24, 512)):
    x = torch.randn(32, hw, hw, c, device=dev).to(torch.bfloat16)
    dy = torch.randn_like(x)
    g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    y, saved = T.bn_train_forward(x, g, b, rm, rv, act="relu")
    def timed(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = timed(lambda: T.bn_train_forward(x, g, b, rm, rv, act="relu", out=y))
    dx = torch.empty_like(x)
    tb = timed(lambda: T.bn_train_backward(x, dy, g, b, saved, act="relu", out=dx))
    nb = x.numel() * 2
    print(f"{name:22s} {nb / 1e6:7.1f} MB  fwd {tf:7.1f} us = {3 * nb / tf / 1e6:5.2f} TB/s   bwd {tb:7.1f} us = {5 * nb / tb / 1e6:5.2f} TB/s")
