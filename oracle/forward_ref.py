"""ORACLE (test infrastructure, never shipped or measured as the product).

Plain PyTorch-CPU fp32 restatement of the reference forward pass, op by op in the
reference's own order (no fusion, no folding), driven directly by a state_dict with the
reference's names:

* DRN-D trunk layer0..layer8 ............ drn.py:122-154, 168-202 (children()[:-2], rt_test.py:60-61)
* pre-activation BasicBlock ............. drn.py:42-57 (== model.py:31-48)
* Bottleneck (D-54/56/105/107) .......... drn.py:77-97
* PPN head .............................. model.py:104-136
* input normalisation ................... rt_test.py:90-101 / aug.py:149-153

Parity pin: compared with the imported reference modules (same weights) by
tests/golden/make_golden.py; fixtures forward_*.npz.  For a floating-point kernel the
tier allows a torch fp32 reference; tolerance for the HIP fp32 path is 1e-4 abs on the
sigmoid head (BASELINE.json north_star).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = {
    "drn_d_22": ("basic", [1, 1, 2, 2, 2, 2, 1, 1]),
    "drn_d_24": ("basic", [1, 1, 2, 2, 2, 2, 2, 2]),
    "drn_d_38": ("basic", [1, 1, 3, 4, 6, 3, 1, 1]),
    "drn_d_40": ("basic", [1, 1, 3, 4, 6, 3, 2, 2]),
    "drn_d_54": ("bottleneck", [1, 1, 3, 4, 6, 3, 1, 1]),
    "drn_d_56": ("bottleneck", [1, 1, 3, 4, 6, 3, 2, 2]),
    "drn_d_105": ("bottleneck", [1, 1, 3, 4, 23, 3, 1, 1]),
    "drn_d_107": ("bottleneck", [1, 1, 3, 4, 23, 3, 2, 2]),
}


def to_torch(sd) -> Dict[str, torch.Tensor]:
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}


class _Net:
    def __init__(self, sd, train_bn=False, momentum=0.1):
        self.sd = to_torch(sd)
        self.train_bn = train_bn
        self.momentum = momentum
        self.taps = {}

    def conv(self, x, name, stride=1, dil=1, pad=0, bias=None):
        b = self.sd[bias] if bias else None
        return F.conv2d(x, self.sd[name + ".weight"], b, stride=stride, padding=pad, dilation=dil)

    def bn(self, x, p):
        sd = self.sd
        if self.train_bn:
            # nn.BatchNorm2d training semantics: batch stats, running stats updated in place
            return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                                sd[p + ".bias"], True, self.momentum, 1e-5)
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                            sd[p + ".bias"], False, 0.0, 1e-5)

    def basic(self, x, p, stride, dil, has_ds):                       # drn.py:42-57
        residual = x
        out = F.relu(self.bn(x, p + ".bn1"))
        out = self.conv(out, p + ".conv1", stride, dil[0], dil[0])
        out = F.relu(self.bn(out, p + ".bn2"))
        out = self.conv(out, p + ".conv2", 1, dil[1], dil[1])
        if has_ds:
            residual = self.bn(self.conv(x, p + ".downsample.0", stride), p + ".downsample.1")
        return out + residual

    def bottleneck(self, x, p, stride, dil, has_ds):                  # drn.py:77-97
        residual = x
        out = F.relu(self.bn(self.conv(x, p + ".conv1"), p + ".bn1"))
        out = F.relu(self.bn(self.conv(out, p + ".conv2", stride, dil[1], dil[1]), p + ".bn2"))
        out = self.bn(self.conv(out, p + ".conv3"), p + ".bn3")
        if has_ds:
            residual = self.bn(self.conv(x, p + ".downsample.0", stride), p + ".downsample.1")
        return F.relu(out + residual)


def forward_ref(sd, x: torch.Tensor, arch: str = "drn_d_22", train_bn: bool = False, momentum: float = 0.1,
                taps=None, grad: bool = False) -> torch.Tensor:
    """x: f32[B,3,H,W] normalised input -> sigmoid head f32[B,C,H/16,W/16].

    `taps`, if a dict, receives named intermediate tensors (for bisecting, fixture G2).
    With train_bn=True the BN layers use batch statistics and update the running stats in
    `sd` in place (used only to calibrate synthetic checkpoints).
    """
    block, layers = LAYERS[arch]
    exp = 1 if block == "basic" else 4
    channels = (16, 32, 64, 128, 256, 512, 512, 512)
    net = _Net(sd, train_bn, momentum)
    # grad=True keeps the autograd graph (oracle/train_ref.py differentiates this very function)
    with (torch.enable_grad() if grad else torch.no_grad()):
        def tap(name, v):
            if taps is not None:
                taps[name] = v
        # layer0 (drn.py:123-128)
        x = F.relu(net.bn(net.conv(x, "backbone.0.0", 1, 1, 3), "backbone.0.1")); tap("backbone.0", x)
        inpl = channels[0]
        for li, (ch, stride) in enumerate(((channels[0], 1), (channels[1], 2)), start=1):
            for i in range(layers[li - 1]):
                x = F.relu(net.bn(net.conv(x, f"backbone.{li}.{3 * i}", stride if i == 0 else 1, 1, 1),
                                  f"backbone.{li}.{3 * i + 1}"))
                inpl = ch
            tap(f"backbone.{li}", x)
        for li, (planes, nblk, stride, dil) in enumerate(
                ((channels[2], layers[2], 2, 1), (channels[3], layers[3], 2, 1),
                 (channels[4], layers[4], 1, 2), (channels[5], layers[5], 1, 4)), start=3):
            for b in range(nblk):
                first = b == 0
                has_ds = first and (stride != 1 or inpl != planes * exp)
                d = (1, 1) if dil == 1 else (dil, dil)
                fn = net.basic if block == "basic" else net.bottleneck
                x = fn(x, f"backbone.{li}.{b}", stride if first else 1, d, has_ds)
                inpl = planes * exp
            tap(f"backbone.{li}", x)
        for li, (ch, n, dil) in enumerate(((channels[6], layers[6], 2), (channels[7], layers[7], 1)), start=7):
            for i in range(n):
                x = F.relu(net.bn(net.conv(x, f"backbone.{li}.{3 * i}", 1, dil, dil), f"backbone.{li}.{3 * i + 1}"))
            tap(f"backbone.{li}", x)
        # PPN head (model.py:104-136)
        x = net.basic(x, "basicblock1", 2, (1, 1), True); tap("basicblock1", x)
        r = net.basic(x, "basicblock2", 1, (1, 1), False); tap("basicblock2", r)
        c = F.leaky_relu(net.bn(r, "bn0_1"), 0.1)
        c = net.conv(c, "conv1x1_1")
        c = F.leaky_relu(net.bn(c, "bn1"), 0.1)
        c = net.conv(c, "conv1", 1, 1, 1)
        c = F.leaky_relu(net.bn(c, "bn0_2"), 0.1)
        c = net.conv(c, "conv1x1_2")
        c = c + r; tap("neck", c)
        c = net.conv(c, "conv2", 1, 1, 1, bias="conv2.bias")
        c = F.leaky_relu(net.bn(c, "bn2"), 0.1); tap("conv2", c)
        z = net.conv(c, "conv3", bias="conv3.bias"); tap("logits", z)
        out = torch.sigmoid(z)
    # train_bn: F.batch_norm updated the running stats in place; numpy-backed entries of `sd`
    # share memory with the torch views, so the caller's dict already holds the new stats.
    return out


def normalize_u8(frames_u8: np.ndarray) -> torch.Tensor:
    """rt_test.py:97-101: HWC u8 -> CHW float, sub mean, div std (0-1 constants on 0-255 pixels)."""
    x = torch.from_numpy(frames_u8).permute(0, 3, 1, 2).float()
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return x.sub_(mean).div_(std)
