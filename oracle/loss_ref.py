"""ORACLE (test infrastructure, never shipped or measured as the product).

PyTorch-CPU restatement of the reference's PPNLoss (main.py:125-216: area / intersection / iou on centre-format
boxes, the five loss terms, sum over non-batch dims then mean over the batch) with autograd supplying the
gradient with respect to the head tensor.

Parity pin: compared with the imported main.PPNLoss (forward values and d/d(feature_map) of the weighted sum)
by tests/golden/make_golden.py; fixtures loss_*.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EPSILON = 1e-6      # config.py:82
K, E = 18, 17


def _iou(b0, b1):                                                   # main.py:125-144
    x0, y0, w0, h0 = b0
    x1, y1, w1, h1 = b1
    w = F.relu(torch.min(x0 + w0 / 2, x1 + w1 / 2) - torch.max(x0 - w0 / 2, x1 - w1 / 2))
    h = F.relu(torch.min(y0 + h0 / 2, y1 + h1 / 2) - torch.max(y0 - h0 / 2, y1 - h1 / 2))
    inter = w * h
    return inter / (w0 * h0 + w1 * h1 - inter + EPSILON)


def ppn_loss_ref(feature_map, t, insize=(384, 384), local_grid=(21, 21)):
    """feature_map f32 [B, 6K+E*sH*sW, H, W]; t: dict of target tensors (oracle/targets_ref.py names).
    Returns the 5 losses (resp, iou, coor, size, limb) as 0-d tensors (main.py:180-216)."""
    B, _, outH, outW = feature_map.shape
    inW, inH = insize
    sW, sH = local_grid
    gridW, gridH = int(inW / outW), int(inH / outH)
    resp, conf = feature_map[:, 0:K], feature_map[:, K:2 * K]
    x, y = feature_map[:, 2 * K:3 * K], feature_map[:, 3 * K:4 * K]
    w, h = feature_map[:, 4 * K:5 * K], feature_map[:, 5 * K:6 * K]
    e = feature_map[:, 6 * K:].reshape(B, E, sH, sW, outH, outW)
    X, Y = torch.meshgrid(torch.arange(outW, dtype=torch.float32), torch.arange(outH, dtype=torch.float32), indexing="xy")

    def rxy(a, b):
        return (a + X) * gridW, (b + Y) * gridH

    (rx, ry), (rw, rh) = rxy(x, y), (inW * w, inH * h)
    (rtx, rty), (rtw, rth) = rxy(t["tx"], t["ty"]), (inW * t["tw"], inH * t["th"])
    ious = _iou((rx, ry, rw, rh), (rtx, rty, rtw, rth))
    dims = (1, 2, 3)
    l_resp = torch.sum((resp - t["delta"]) ** 2, dims)
    l_iou = torch.sum(t["delta"] * (conf - ious) ** 2, dims)
    l_coor = torch.sum(t["weight"] * ((x - t["tx_half"]) ** 2 + (y - t["ty_half"]) ** 2), dims)
    l_size = torch.sum(t["weight"] * ((torch.sqrt(w + EPSILON) - torch.sqrt(t["tw"] + EPSILON)) ** 2 +
                                      (torch.sqrt(h + EPSILON) - torch.sqrt(t["th"] + EPSILON)) ** 2), dims)
    l_limb = torch.sum(t["weight_ij"] * (e - t["te"]) ** 2, (1, 2, 3, 4, 5))
    return tuple(torch.mean(v) for v in (l_resp, l_iou, l_coor, l_size, l_limb))


def loss_and_grad_ref(head: np.ndarray, targets: dict, coeff, **kw):
    """(losses f32[5], d(sum_i coeff_i L_i)/d(head) f32 like head) via autograd."""
    fm = torch.from_numpy(head).clone().requires_grad_(True)
    t = {k: torch.from_numpy(v) for k, v in targets.items()}
    losses = ppn_loss_ref(fm, t, **kw)
    total = sum(float(c) * l for c, l in zip(coeff, losses))
    total.backward()
    return np.array([float(l.detach()) for l in losses], np.float32), fm.grad.numpy()
