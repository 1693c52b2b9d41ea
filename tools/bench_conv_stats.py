"""Cost of the BatchNorm-statistics epilogue (ppn_conv_desc.stats_mode) per launch, and what the BatchNorm call behind it saves:
conv alone / conv + sums (forward, backward form), BN forward / backward with its own reduction pass / with the sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_pose_proposal_network_amd import train as T
dev = torch.device("cuda")


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, hw, cin, cout, k, dil in (("512->512 3x3 d2 @48", 48, 512, 512, 3, 2), ("256->256 3x3 @48", 48, 256, 256, 3, 1),
                                    ("128->128 3x3 @48", 48, 128, 128, 3, 1), ("512->512 3x3 @24", 24, 512, 512, 3, 1),
                                    ("512->128 1x1 @24", 24, 512, 128, 1, 1)):
    B = 32
    x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    pad = dil * (k // 2)
    out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.bfloat16)
    xb = torch.randn(B, hw, hw, cout, device=dev).to(torch.bfloat16)
    g, b = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    y = torch.empty_like(out)
    _, saved = T.bn_train_forward(xb, g, b, act="relu", out=y)
    t0 = timed(lambda: T.conv2d_nhwc(x, w, 1, dil, pad, out=out))
    t1 = timed(lambda: T.conv2d_nhwc(x, w, 1, dil, pad, out=out, stats="fwd"))
    t2 = timed(lambda: T.conv2d_nhwc(x, w, 1, dil, pad, out=out, stats=(xb, g, b, saved, "relu")))
    f0 = timed(lambda: T.bn_train_forward(out, g, b, act="relu", out=y))
    def f_fused():
        _, st = T.conv2d_nhwc(x, w, 1, dil, pad, out=out, stats="fwd")
        T.bn_train_forward(out, g, b, act="relu", out=y, stats=st)
    f1 = timed(f_fused) - t1
    dx = torch.empty_like(out)
    b0 = timed(lambda: T.bn_train_backward(xb, out, g, b, saved, act="relu", out=dx))
    def b_fused():
        _, st = T.conv2d_nhwc(x, w, 1, dil, pad, out=out, stats=(xb, g, b, saved, "relu"))
        T.bn_train_backward(xb, out, g, b, saved, act="relu", out=dx, stats=st)
    b1 = timed(b_fused) - t2
    print(f"{name:22s} conv {t0:6.1f} | +fwd sums {t1:6.1f} ({t1 - t0:+5.1f}) BN fwd {f0:5.1f} -> {f1:5.1f} | "
          f"+bwd sums {t2:6.1f} ({t2 - t0:+5.1f}) BN bwd {b0:5.1f} -> {b1:5.1f}   net fwd {t1 - t0 + f1 - f0:+5.1f} bwd {t2 - t0 + b1 - b0:+5.1f} us")
