"""Micro-benchmark of the fused PPN loss kernels at the per-GPU batch of BASELINE config 4 (B=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import loss, config as cfg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
fm = torch.rand(B, cfg.lastsize(), 24, 24, device=dev) * 0.9 + 0.05
t = {k: torch.rand(B, 18, 24, 24, device=dev) for k in ("delta", "weight", "tx_half", "ty_half", "tx", "ty", "tw", "th")}
t["te"] = (torch.rand(B, 17, 21, 21, 24, 24, device=dev) > 0.999).float()
t["weight_ij"] = torch.rand(B, 17, 21, 21, 24, 24, device=dev)
crit = loss.PPNLoss()
for want_grad in (False, True):
    for _ in range(3):
        crit.forward_backward(fm, t, [0.2] * 5, want_grad=want_grad)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); crit.forward_backward(fm, t, [0.2] * 5, want_grad=want_grad); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[5]
    limb = B * 17 * 441 * 576 * 4
    byt = limb * (4 if want_grad else 3)
    print(f"loss {'fwd+bwd' if want_grad else 'fwd'}: {ms*1e3:.1f} us  algorithmic {byt/1e9:.2f} GB -> {byt/ms/1e6:.0f} GB/s")
