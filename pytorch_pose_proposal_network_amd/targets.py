"""Training targets on the device (reference: KeypointsDataset.__getitem__ dataset.py:96-185 + CustomBatch
dataset.py:233-248, shipped to the GPU by main.py:649-661).

    packed  = pack_people(batch_of_person_lists)            # a few hundred bytes per person, host
    targets = encode_targets(packed, device="cuda")         # the ten tensors PPNLoss / PPNTrainer take

A person is a dict(bbox=(cx, cy, w, h), points f32[K-1, 2] (x, y), visible bool[K-1], size float), the
fields dataset.py:108-117 reads.  The encoder kernel (csrc/encode.hip) is bit-exact with the host rules.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import config as cfg
from . import lib as L

TARGET_KEYS = ("delta", "weight", "weight_ij", "tx_half", "ty_half", "tx", "ty", "tw", "th", "te")


def pack_people(batch: Sequence[Sequence[dict]], pmax: int = 0):
    """-> (people f32[B,pmax,5+2(K-1)], visible i32[B,pmax], count i32[B]) NumPy arrays."""
    K = cfg.K
    B = len(batch)
    pmax = max(pmax, max((len(p) for p in batch), default=0), 1)
    people = np.zeros((B, pmax, 5 + 2 * (K - 1)), np.float32)
    visible = np.zeros((B, pmax), np.int32)
    count = np.zeros(B, np.int32)
    for b, plist in enumerate(batch):
        count[b] = len(plist)
        for i, person in enumerate(plist):
            people[b, i, 0:4] = person["bbox"]
            people[b, i, 4] = person["size"]
            people[b, i, 5:] = np.asarray(person["points"], np.float32).reshape(-1)
            bits = 0
            for k, v in enumerate(person["visible"]):
                bits |= int(bool(v)) << k
            visible[b, i] = bits
    return people, visible, count


def encode_targets(packed, insize=(384, 384), outsize=(24, 24), local_grid=(21, 21), device="cuda") -> Dict[str, torch.Tensor]:
    lib = L.load()
    people, visible, count = packed
    dev = torch.device(device)
    B, pmax, _ = people.shape
    K, E = cfg.K, cfg.E
    c = L.LossCfg()
    c.K, c.E = K, E
    c.sW, c.sH = local_grid
    c.W, c.H = outsize
    c.inW, c.inH = insize
    pd = torch.from_numpy(np.ascontiguousarray(people)).to(dev)
    vd = torch.from_numpy(np.ascontiguousarray(visible)).to(dev)
    cd = torch.from_numpy(np.ascontiguousarray(count)).to(dev)
    t = {}
    for k in TARGET_KEYS:
        shape = (B, E, c.sH, c.sW, c.H, c.W) if k in ("weight_ij", "te") else (B, K, c.H, c.W)
        t[k] = torch.empty(shape, dtype=torch.float32, device=dev)
    edges = (C.c_int32 * (2 * E))(*[int(v) for e in cfg.EDGES for v in e])
    # "limb_c" (u8, same shape as te): te | weight_ij in two bits per element -- what the two limb-streaming kernels of a
    # training iteration read instead of the two f32 tensors (PPNLoss.forward_backward_dz / limb_dual_nhwc use it when the
    # targets carry it; bit-identical results, 1.1 GB less HBM traffic per kernel at batch 32)
    limb_c = torch.empty((B, E, c.sH, c.sW, c.H, c.W), dtype=torch.uint8, device=dev)
    L.check(lib.ppn_encode_targets_c(C.byref(c), edges, pd.data_ptr(), vd.data_ptr(), cd.data_ptr(), B, pmax,
                                     *[t[k].data_ptr() for k in TARGET_KEYS], limb_c.data_ptr(), L.current_stream_ptr()),
            "ppn_encode_targets_c")
    t["limb_c"] = limb_c
    return t


def synthetic_targets(seed: int, batch: int, insize=(384, 384), device="cuda") -> Dict[str, torch.Tensor]:
    """1..4 synthetic people per image (synth.synthetic_people), encoded on the device."""
    from . import synth
    outsize = (insize[0] // 16, insize[1] // 16)
    lists: List[List[dict]] = [synth.synthetic_people(seed + i, insize=insize) for i in range(batch)]
    return encode_targets(pack_people(lists), insize, outsize, device=device)
