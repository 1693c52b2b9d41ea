"""Stride-2 input gradients of DRN-D-22 at batch 32: parity sub-convolutions (train._dgrad_stride2) vs the zero-upsampled
form (PPN_DGRAD_S2_PARITY=0), values compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import train as T
dev = torch.device("cuda")
CASES = [("layer2 3x3 16->32 s2 @384", 16, 32, 384, 3), ("layer3.0.conv1 3x3 32->64 s2 @192", 32, 64, 192, 3),
         ("layer3.0.downsample 1x1 32->64 s2", 32, 64, 192, 1), ("layer4.0.conv1 3x3 64->128 s2 @96", 64, 128, 96, 3),
         ("layer4.0.downsample 1x1 64->128 s2", 64, 128, 96, 1)]
for name, ci, co, H, k in CASES:
    Ho = (H + 2 * (k // 2) - k) // 2 + 1
    dy = torch.randn(32, Ho, Ho, co, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, k, k, device=dev) * 0.1
    res = {}
    for mode in (True, False):
        T._S2_PARITY = mode
        for _ in range(3):
            out = T.conv_dgrad(dy, w, (H, H), 2, 1, k // 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = T.conv_dgrad(dy, w, (H, H), 2, 1, k // 2)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) * 100, out.float())
    d = (res[True][1] - res[False][1]).abs().max().item()
    print(f"{name:40s} parity {res[True][0]:7.1f} us   zero-upsampled {res[False][0]:7.1f} us   max|diff| {d:.3g} (max |dx| {res[False][1].abs().max().item():.3g})")
