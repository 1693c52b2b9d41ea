"""Synthetic, reproducible checkpoints, frames and crowd heads (no dataset, no network).

* ``make_state_dict``  -- a state_dict with the reference's names/shapes (arch.param_spec),
  conv weights ~ kaiming-normal(a=0.1, fan_in) as in model.py:97-102 but drawn from the
  integer PRNG, BN affine slightly randomised so the scale/shift paths are exercised.
* ``planted_crowd_head`` -- SURVEY.md 8d config 5: a head tensor with planted people
  following the target-encoder rules of dataset.py:108-152, used for the decode stress.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

from . import arch as A
from . import config as cfg
from . import prng


def _name_stream(seed: int, name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return prng.stream_seed(seed, h & 0x7FFFFFFF)


def make_state_dict(arch: str = "drn_d_22", seed: int = 0, bn_stats: Optional[Dict[str, np.ndarray]] = None,
                    head_channels: Optional[int] = None) -> Dict[str, np.ndarray]:
    sd: Dict[str, np.ndarray] = {}
    for name, shape in A.param_spec(arch, head_channels):
        s = _name_stream(seed, name)
        n = int(np.prod(shape)) if shape else 1
        if name.endswith("num_batches_tracked"):
            sd[name] = np.array(1, dtype=np.int64)
        elif name.endswith("running_mean"):
            sd[name] = np.zeros(shape, np.float32)
        elif name.endswith("running_var"):
            sd[name] = np.ones(shape, np.float32)
        elif len(shape) == 4:                                   # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            std = math.sqrt(2.0 / (1.0 + 0.1 * 0.1)) / math.sqrt(fan_in)
            sd[name] = (prng.normalish(s, n) * np.float32(std)).reshape(shape)
        elif name in ("conv2.bias", "conv3.bias"):              # nn.Conv2d default bias init range
            bound = 1.0 / math.sqrt(512 * (9 if name == "conv2.bias" else 1))
            sd[name] = prng.uniform(s, n, -bound, bound).reshape(shape)
        elif name.endswith(".weight"):                          # BN gamma
            sd[name] = prng.uniform(s, n, 0.7, 1.3).reshape(shape)
        else:                                                   # BN beta
            sd[name] = prng.uniform(s, n, -0.2, 0.2).reshape(shape)
    if bn_stats:
        for k, v in bn_stats.items():
            assert k in sd and sd[k].shape == v.shape, k
            sd[k] = v.astype(np.float32)
    return sd


def normalized_frames(frames_u8: np.ndarray) -> np.ndarray:
    """u8[B,H,W,3] -> f32[B,3,H,W] with the reference's (x-mean)/std, no /255 (aug.py:149-153)."""
    x = frames_u8.astype(np.float32).transpose(0, 3, 1, 2)
    mean = np.array(cfg.MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.array(cfg.STD, np.float32).reshape(1, 3, 1, 1)
    # C-contiguous NCHW, as a DataLoader hands batches over (the arithmetic on the transposed view keeps the NHWC memory order, and
    # PPNTrainer.forward would then re-lay the batch out on the device in every step: 26 us at batch 32)
    return np.ascontiguousarray(((x - mean) / std).astype(np.float32))


def planted_crowd_head(seed: int, n_people: int = 16, n_decoys: int = 4, out_hw=(24, 24),
                       local_grid=(21, 21)) -> np.ndarray:
    """One head tensor f32[6K+E*sH*sW, H, W] with a planted crowd.

    Background: resp,conf ~ U(0,0.3) (delta < 0.09), x,y ~ U(0,1), w,h ~ U(0.02,0.1),
    e ~ U(0,0.5).  Each planted person gets a distinct root cell with resp=conf in
    U(0.8,1), an instance box of 0.15..0.4 of the frame, and for every limb of the
    skeleton tree an offset |dh|,|dw| <= 10 whose e-channel is raised to U(0.9,1)
    (te[ei, dh+10, dw+10, h, w] = 1 in dataset.py:136-152).  Some limbs are truncated
    (target delta < 0.15) or point outside the grid to exercise both `break`s of
    datatest.py:118-122; decoy roots have no limbs or heavily overlap a planted root.
    All values come from the integer PRNG; candidate scores are distinct by construction
    with overwhelming probability (24-bit uniforms) and the generator re-draws on a tie.
    """
    H, W = out_hw
    sW, sH = local_grid
    K, E = cfg.K, cfg.E
    C = 6 * K + E * sH * sW
    n = C * H * W
    st = lambda i: prng.stream_seed(seed, i)
    head = np.empty((C, H, W), np.float32)
    head[0:2 * K] = prng.uniform(st(1), 2 * K * H * W, 0.0, 0.3).reshape(2 * K, H, W)
    head[2 * K:4 * K] = prng.uniform01(st(2), 2 * K * H * W).reshape(2 * K, H, W)
    head[4 * K:6 * K] = prng.uniform(st(3), 2 * K * H * W, 0.02, 0.1).reshape(2 * K, H, W)
    head[6 * K:] = prng.uniform(st(4), E * sH * sW * H * W, 0.0, 0.5).reshape(E * sH * sW, H, W)

    r = prng.raw_u64(st(5), 4096)
    ri = [0]

    def rnd():                      # float in [0,1)
        v = float(r[ri[0]] >> np.uint64(40)) / 16777216.0
        ri[0] += 1
        return v

    def rint(lo, hi):               # integer in [lo, hi]
        return lo + int(rnd() * (hi - lo + 1))

    src, dst, order = cfg.tree_tables()
    cells = set()
    used_scores = set()

    def put_kp(k, h, w, strong=True):
        while True:
            a = 0.8 + 0.2 * rnd() if strong else 0.2 * rnd()
            val = np.float32(a)
            prod = float(val * val)
            if prod not in used_scores:
                used_scores.add(prod)
                break
        head[k, h, w] = val
        head[K + k, h, w] = val

    people = 0
    tries = 0
    while people < n_people + n_decoys and tries < 10000:
        tries += 1
        h0, w0 = rint(0, H - 1), rint(0, W - 1)
        if (h0, w0) in cells:
            continue
        cells.add((h0, w0))
        decoy = people >= n_people
        put_kp(0, h0, w0)
        head[4 * K, h0, w0] = np.float32(0.15 + 0.25 * rnd())
        head[5 * K, h0, w0] = np.float32(0.15 + 0.25 * rnd())
        people += 1
        if decoy and rnd() < 0.5:
            continue                                  # decoy without limbs -> dropped (datatest.py:129)
        pos = {0: (h0, w0)}
        for e in order:
            s, d = src[e], dst[e]
            if s not in pos:
                continue
            ph, pw = pos[s]
            mode = rnd()
            dh, dw = rint(-3, 3), rint(-3, 3)
            if mode < 0.06:                           # limb pointing out of the grid
                dh = -10 if ph < 10 else 10
                if 0 <= ph + dh < H:
                    dh = -ph - 1 if ph < 10 else H - ph
                    dh = max(-10, min(10, dh))
            jh, jw = ph + dh, pw + dw
            ch = 6 * K + e * sH * sW + (dh + sH // 2) * sW + (dw + sW // 2)
            head[ch, ph, pw] = np.float32(0.9 + 0.1 * rnd())
            if not (0 <= jh < H and 0 <= jw < W):
                continue
            if mode > 0.92:                           # truncated limb: target delta stays < 0.15
                continue
            put_kp(d, jh, jw)
            head[4 * K + d, jh, jw] = np.float32(0.02 + 0.08 * rnd())
            head[5 * K + d, jh, jw] = np.float32(0.02 + 0.08 * rnd())
            pos[d] = (jh, jw)
    return head


def synthetic_people(seed: int, insize=(384, 384), max_people: int = 4):
    """List of people: dict(bbox=(cx,cy,w,h), points f32[17,2] (x,y), visible bool[17], size float)."""
    inW, inH = insize
    r = prng.uniform01(prng.stream_seed(seed, 0), 1 + max_people * 64).astype(np.float64)
    n = 1 + int(r[0] * max_people)
    n = min(n, max_people)
    people, p = [], 1
    for _ in range(n):
        cx, cy = 40 + r[p] * (inW - 80), 40 + r[p + 1] * (inH - 80)
        w, h = 60 + r[p + 2] * 140, 60 + r[p + 3] * 140
        size = 8 + r[p + 4] * 16
        pts = np.stack([r[p + 5:p + 22] * (inW - 1), r[p + 22:p + 39] * (inH - 1)], 1).astype(np.float32)
        vis = r[p + 39:p + 56] > 0.15
        people.append(dict(bbox=(np.float32(cx), np.float32(cy), np.float32(w), np.float32(h)), points=pts,
                           visible=vis, size=np.float32(size)))
        p += 64
    return people


def eval_case(seed: int, n_images: int = 12, insize=(384, 384)):
    """A synthetic `pck_object` (main.py:899-990) for the AP evaluation: per image 0..3 ground-truth people and a set
    of predicted people made from them -- jittered joints (some beyond the PCKh@0.5 radius), dropped joints, a
    duplicate detection, false positives -- in the formats the reference hands to datatest.evaluation:
    [fnames, gt_kps f32[P,17,2], humans [{kp: f32[4] (ymin,xmin,ymax,xmax)}], scores [{kp: f32}], gt_bboxes
    [(cx,cy,w,h)], is_visible, size]."""
    inW, inH = insize
    obj = [[], [], [], [], [], [], []]
    for im in range(n_images):
        r = prng.uniform01(prng.stream_seed(seed, 10 + im), 4096).astype(np.float64)
        q = 0

        def nxt(k=1):
            nonlocal q
            v = r[q:q + k]
            q += k
            return v if k > 1 else float(v[0])

        n_gt = int(nxt() * 4)                                  # 0..3 (frames without people are dropped by the metric)
        gt_kps, gt_boxes, vis, sizes = [], [], [], []
        for _ in range(n_gt):
            cx, cy = 60 + nxt() * (inW - 120), 60 + nxt() * (inH - 120)
            w, h = 50 + nxt() * 120, 80 + nxt() * 160
            pts = np.stack([cx + (nxt(17) - 0.5) * w, cy + (nxt(17) - 0.5) * h], 1).astype(np.float32)
            gt_kps.append(pts)
            gt_boxes.append((np.float32(cx), np.float32(cy), np.float32(w), np.float32(h)))
            vis.append(nxt(17) > 0.2)
            sizes.append(np.float32(8 + nxt() * 16))
        humans, scores = [], []
        preds = []
        for g in range(n_gt):
            if nxt() < 0.85:
                preds.append((g, 0.02 + nxt() * 0.1))          # a detection of person g with small jitter
            if nxt() < 0.25:
                preds.append((g, 0.3 + nxt() * 0.5))           # a sloppy duplicate
        n_fp = int(nxt() * 3)
        for g, jit in preds:
            cx, cy, w, h = gt_boxes[g]
            diag = float(np.hypot(w, h))
            hm, sm = {0: np.array([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], np.float32)}, {0: np.float32(0.3 + 0.7 * nxt())}
            for k in range(1, 18):
                if nxt() < 0.12:
                    continue                                    # joint not found by the limb parse
                x = gt_kps[g][k - 1, 0] + (nxt() - 0.5) * 2 * jit * diag
                y = gt_kps[g][k - 1, 1] + (nxt() - 0.5) * 2 * jit * diag
                s_ = 6 + nxt() * 10
                hm[k] = np.array([y - s_, x - s_, y + s_, x + s_], np.float32)
                sm[k] = np.float32(0.15 + 0.85 * nxt())
            humans.append(hm)
            scores.append(sm)
        for _ in range(n_fp):
            hm, sm = {0: (nxt(4) * inW).astype(np.float32)}, {0: np.float32(nxt())}
            for k in range(1, 18):
                if nxt() < 0.5:
                    c = nxt(2) * inW
                    hm[k] = np.array([c[1] - 8, c[0] - 8, c[1] + 8, c[0] + 8], np.float32)
                    sm[k] = np.float32(0.15 + 0.5 * nxt())
            humans.append(hm)
            scores.append(sm)
        obj[0].append(f"img_{seed}_{im}.jpg")
        obj[1].append(np.stack(gt_kps) if gt_kps else np.zeros((0, 17, 2), np.float32))
        obj[2].append(humans)
        obj[3].append(scores)
        obj[4].append(gt_boxes)
        obj[5].append(vis)
        obj[6].append(sizes)
    return obj
