"""Micro-benchmark of the weight-gradient kernels on the D-22 layer shapes at batch 32 (bf16).

    python tools/bench_wgrad.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import train as T

CASES = [  # name, B, cin, cout, H (input), k, stride, dil, pad
    ("layer6 3x3 512->512 d4 @48", 32, 512, 512, 48, 3, 1, 4, 4),
    ("layer5 3x3 256->256 d2 @48", 32, 256, 256, 48, 3, 1, 2, 2),
    ("layer4 3x3 128->128 @48", 32, 128, 128, 48, 3, 1, 1, 1),
    ("layer3 3x3 64->64 @96", 32, 64, 64, 96, 3, 1, 1, 1),
    ("head conv3 1x1 512->7616 @24", 32, 512, 7616, 24, 1, 1, 1, 0),
    ("layer1 3x3 16->16 @384", 32, 16, 16, 384, 3, 1, 1, 1),
    ("layer0 7x7 8->16 @384", 32, 8, 16, 384, 7, 1, 1, 3),
    ("layer0 7x7 4->16 @384", 32, 4, 16, 384, 7, 1, 1, 3),
    ("layer2 3x3 16->32 s2 @384", 32, 16, 32, 384, 3, 2, 1, 1),
    ("head block 3x3 512->512 @24", 32, 512, 512, 24, 3, 1, 1, 1),
    ("layer6.0 3x3 256->512 d4 @48", 32, 256, 512, 48, 3, 1, 4, 4),
    ("1x1 512->128 @24", 32, 512, 128, 24, 1, 1, 1, 0),
    ("1x1 128->512 @24", 32, 128, 512, 24, 1, 1, 1, 0),
    ("1x1 256->512 @48", 32, 256, 512, 48, 1, 1, 1, 0),
]
dev = torch.device("cuda")
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[0]]
for name, B, ci, co, H, k, s, d, p in CASES:
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    x = torch.randn(B, H, H, ci, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, Ho, Ho, co, device=dev).to(torch.bfloat16)
    out = torch.empty(co, ci, k, k, device=dev)
    for _ in range(3):
        T.conv_wgrad(x, dy, k, s, d, p, out=out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 20
    ev[0].record()
    for _ in range(n):
        T.conv_wgrad(x, dy, k, s, d, p, out=out)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / n * 1e3
    fl = 2.0 * B * Ho * Ho * co * ci * k * k
    print(f"{name:34s} {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s")
