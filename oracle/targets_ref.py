"""ORACLE (test infrastructure, never shipped or measured as the product).

NumPy restatement of the reference's training-target encoder, dataset.py:96-185 (per image: delta, tx, ty, tw,
th [K,H,W]; te [E,sH,sW,H,W]; max_delta_ij -> weight_ij; weight; tx_half, ty_half), plus a synthetic crowd
generator (1..4 people per image, keypoints uniform in frame, part size 8..24 px,
instance box 60..200 px) driven by the repo's integer PRNG.

Parity pin: PINNED by tests/golden/targets_cases.npz -- the ten tensors returned by the reference's own
KeypointsDataset.__getitem__ (dataset.py:70-200), run in this container by tests/golden/make_golden.py on synthetic
annotation files (image loader stubbed, `np.bool` aliased, the imgaug stage bypassed with the annotated coordinates in
IAA's containers, the reference's own aug.ToNormalizedTensor kept).  encode_targets() below equals them bit for bit
on every case, edge cases included (unlabeled root, keypoints outside the frame and their int() truncation, limbs
beyond the window, invisible joints, people sharing cells); tests/test_oracle.py replays the fixture on CPU.
Not covered: dataset.py:44-55 drops annotations without a visible joint at load time (dataset loading is out of scope).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

from pytorch_pose_proposal_network_amd import prng
from oracle.decode_ref import EDGES, K, E


from pytorch_pose_proposal_network_amd.synth import synthetic_people  # noqa: E402,F401  (generator lives in the package)


def encode_targets(people, insize=(384, 384), outsize=(24, 24), local_grid=(21, 21)):
    """dataset.py:96-185 for one image.  Returns dict of float32 arrays."""
    inW, inH = insize
    outW, outH = outsize
    sW, sH = local_grid
    gridW, gridH = inW // outW, inH // outH
    delta = np.zeros((K, outH, outW), np.float32)
    tx, ty, tw, th = (np.zeros((K, outH, outW), np.float32) for _ in range(4))
    te = np.zeros((E, sH, sW, outH, outW), np.float32)
    for person in people:
        cx, cy, w, h = person["bbox"]
        points = [np.array([cx, cy], np.float32)] + [p for p in person["points"]]
        labeled = [bool(w > 0 and h > 0)] + [bool(v) for v in person["visible"]]
        for k, (xy, l) in enumerate(zip(points, labeled)):                      # dataset.py:119-134
            if not l:
                continue
            gx, gy = np.float32(xy[0]) / np.float32(gridW), np.float32(xy[1]) / np.float32(gridH)
            ix, iy = int(gx), int(gy)
            sizeW = w if k == 0 else person["size"]
            sizeH = h if k == 0 else person["size"]
            if 0 <= iy < outH and 0 <= ix < outW:
                delta[k, iy, ix] = 1
                tx[k, iy, ix] = gx - ix
                ty[k, iy, ix] = gy - iy
                tw[k, iy, ix] = sizeW / inW
                th[k, iy, ix] = sizeH / inH
        for ei, (s, t) in enumerate(EDGES):                                      # dataset.py:136-152
            if not labeled[s] or not labeled[t]:
                continue
            src, tar = points[s], points[t]
            iyx = (int(src[1] / gridH), int(src[0] / gridW))
            jyx = (int(tar[1] / gridH) - iyx[0] + sH // 2, int(tar[0] / gridW) - iyx[1] + sW // 2)
            if iyx[0] < 0 or iyx[1] < 0 or iyx[0] >= outH or iyx[1] >= outW:
                continue
            if jyx[0] < 0 or jyx[1] < 0 or jyx[0] >= sH or jyx[1] >= sW:
                continue
            te[ei, jyx[0], jyx[1], iyx[0], iyx[1]] = 1
    max_delta_ij = np.zeros((E, outH, outW, sH, sW), np.float32)                 # dataset.py:155-170
    for ei, (s, t) in enumerate(EDGES):
        max_delta_ij[ei][delta[s] != 0] = 1.0
        pad = np.pad(delta[t], (sH // 2, sW // 2), "constant")
        for r, c in zip(*np.where(delta[s] == 0)):
            max_delta_ij[ei][r, c] = pad[r:r + sH, c:c + sW]
    max_delta_ij = max_delta_ij.transpose(0, 3, 4, 1, 2)
    weight_ij = np.minimum(max_delta_ij + np.where(max_delta_ij < 0.5, np.float32(0.0005), np.float32(0)), 1.0)
    weight = np.minimum(delta + np.where(delta < 0.5, np.float32(0.0005), np.float32(0)), 1.0)
    half = np.where(delta < 0.5, np.float32(0.5), np.float32(0))
    return dict(delta=delta, weight=weight.astype(np.float32), weight_ij=weight_ij.astype(np.float32), tx=tx, ty=ty,
                tx_half=(tx + half).astype(np.float32), ty_half=(ty + half).astype(np.float32), tw=tw, th=th, te=te)


def synthetic_batch(seed: int, batch: int, **kw):
    """Stacked targets for `batch` images (seeds seed, seed+1, ...): dict of f32 [B, ...]."""
    per = [encode_targets(synthetic_people(seed + i), **kw) for i in range(batch)]
    return {k: np.ascontiguousarray(np.stack([p[k] for p in per])) for k in per[0]}
