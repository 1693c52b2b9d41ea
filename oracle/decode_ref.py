"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement in NumPy of the reference's grid-cell decode, root-box NMS and greedy
limb parse, returning *indices* so the HIP path can be compared bit-exactly:

* restore_xy / restore_size / bbox build ........ datatest.py:63-86
* candidate select ................................ datatest.py:87-92
* non_maximum_suppression ......................... datatest.py:134-160
* greedy limb parse over DIRECTED_GRAPHS .......... datatest.py:98-132, config.py:67-80
* head slicing + delta = resp*conf ................ rt_test.py:106-130

Parity pin: checked against the imported reference functions on planted-crowd and random
heads by tests/golden/make_golden.py (fixtures decode_*.npz) -- see tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Documented tie rule (the reference's `score.argsort()[::-1]` order is implementation
defined for equal scores): candidates are ordered by descending
score, equal scores by ascending row-major cell index.  Fixtures have distinct scores.
"""
from __future__ import annotations

import numpy as np

K = 18
E = 17
# (src, dst) per edge and the five chains -- values of config.py:44-80
EDGES = [[0, 15], [15, 13], [13, 1], [1, 3], [3, 5], [13, 2], [2, 4], [4, 6], [13, 17], [17, 14],
         [14, 7], [14, 8], [7, 9], [8, 10], [9, 11], [10, 12], [0, 16]]
DIRECTED_GRAPHS = [
    [[0, 1, 2, 3, 4], [15, 13, 1, 3, 5]],
    [[0, 1, 5, 6, 7], [15, 13, 2, 4, 6]],
    [[0, 1, 8, 9, 10, 12, 14], [15, 13, 17, 14, 7, 9, 11]],
    [[0, 1, 8, 9, 11, 13, 15], [15, 13, 17, 14, 8, 10, 12]],
    [[16], [16]],
]


def split_head(head: np.ndarray, local_grid=(21, 21)):
    """rt_test.py:109-130 for one image: head f32[6K+E*sH*sW, H, W] -> delta,x,y,w,h,e."""
    sW, sH = local_grid
    C, H, W = head.shape
    assert C == 6 * K + E * sH * sW
    resp, conf = head[0:K], head[K:2 * K]
    x, y, w, h = head[2 * K:3 * K], head[3 * K:4 * K], head[4 * K:5 * K], head[5 * K:6 * K]
    e = head[6 * K:].reshape(E, sH, sW, H, W)
    delta = (resp * conf).astype(np.float32)
    return delta, x, y, w, h, e


def build_bbox(x, y, w, h, insize=(384, 384)):
    """datatest.py:63-86.  Returns bbox f32[K,H,W,4] = (ymin, xmin, ymax, xmax)."""
    _, H, W = x.shape
    inW, inH = insize
    gridW, gridH = int(inW / W), int(inH / H)
    X, Y = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    rx, ry = (x + X) * np.float32(gridW), (y + Y) * np.float32(gridH)
    rw, rh = np.float32(inW) * w, np.float32(inH) * h
    ymin, ymax = ry - rh / np.float32(2), ry + rh / np.float32(2)
    xmin, xmax = rx - rw / np.float32(2), rx + rw / np.float32(2)
    return np.stack([ymin, xmin, ymax, xmax], axis=-1).astype(np.float32)


def _note(margins, key, value):
    if margins is not None and np.isfinite(value):
        margins[key] = min(margins.get(key, np.inf), float(value))


def nms_ref(bbox: np.ndarray, thresh: float, score=None, limit=None, margins=None) -> np.ndarray:
    """datatest.py:134-160 with fp32 arithmetic in the reference's operation order.
    `margins` (optional dict) receives the smallest distance of any decision from its flip point: 'iou' =
    min |iou - thresh| over the compared pairs, 'order' = smallest gap between consecutively sorted scores."""
    n = len(bbox)
    if n == 0:
        return np.zeros((0,), dtype=np.int32)
    bbox = np.asarray(bbox, np.float32)
    if score is not None:
        score = np.asarray(score, np.float32)
        order = np.argsort(-score, kind="stable")      # tie rule: see module docstring
        bbox = bbox[order]
        if n > 1:
            _note(margins, "order", np.min(-np.diff(score[order].astype(np.float64))))
    thr = np.float32(thresh)
    area = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])
    keep = []
    for i in range(n):
        b = bbox[i]
        ok = True
        if keep:
            s = bbox[keep]
            tl0, tl1 = np.maximum(b[0], s[:, 0]), np.maximum(b[1], s[:, 1])
            br0, br1 = np.minimum(b[2], s[:, 2]), np.minimum(b[3], s[:, 3])
            inter = (br0 - tl0) * (br1 - tl1) * ((tl0 < br0) & (tl1 < br1))
            inter = inter.astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = inter / ((area[i] + area[keep]) - inter)
            ok = not bool((iou >= thr).any())
            # (pairs after the first suppressing one are not decisive, but counting them only shrinks the margin)
            _note(margins, "iou", np.nanmin(np.abs(iou.astype(np.float64) - float(thr))))
        if ok:
            keep.append(i)
            if limit is not None and len(keep) >= limit:
                break
    sel = np.asarray(keep, dtype=np.int64)
    if score is not None:
        sel = order[sel]
    return sel.astype(np.int32)


def tree_edges():
    """Edges parent-before-child; equivalent to replaying the five chains (config.py:67-80)."""
    seen, order = set(), []
    for es, _ in DIRECTED_GRAPHS:
        for e in es:
            if e not in seen:
                seen.add(e)
                order.append(e)
    return order


def decode_ref(head: np.ndarray, det_thr=0.15, nms_thr=0.3, min_kp=1, insize=(384, 384),
               local_grid=(21, 21), margins=None):
    """Compact decode of one image's head tensor.

    `margins` (optional dict) receives, per kind of decision taken on THIS head, the smallest distance from its
    flip point: 'cand' |delta_root - thr| over all cells, 'order' / 'iou' (nms_ref), 'argmax' top1 - top2 of every
    evaluated limb window, 'hop' |delta_target - thr| of every evaluated hop.  A head within eps of this one must
    decode to the same indices when every margin exceeds the perturbation eps can cause (tests/test_e2e_gpu.py).

    Returns dict with
      n            number of humans kept
      root_cell    int32[n]     row-major cell (h*W+w) of each human's root, descending score
      kp_cell      int32[n,K]   cell of every accepted keypoint, -1 = absent
      limb_arg     int32[n,E]   argmax s = sh*sW+sw of every *evaluated* limb, -1 = not evaluated
      bbox         f32[n,K,4]   (ymin,xmin,ymax,xmax) of accepted keypoints, 0 elsewhere
      score        f32[n,K]     delta of accepted keypoints, 0 elsewhere
      cand         int32[m]     candidate cells (row-major), selected int32[s] NMS survivors (indices into cand)
    """
    sW, sH = local_grid
    delta, x, y, w, h, e = split_head(head, local_grid)
    _, H, W = delta.shape
    bbox = build_bbox(x, y, w, h, insize)
    thr = np.float32(det_thr)
    cand_h, cand_w = np.where(delta[0] > thr)
    cand = (cand_h * W + cand_w).astype(np.int32)
    selected = nms_ref(bbox[0][cand_h, cand_w], nms_thr, delta[0][cand_h, cand_w], margins=margins)
    _note(margins, "cand", np.min(np.abs(delta[0].astype(np.float64) - float(thr))))

    order = tree_edges()
    roots, kp_cells, limb_args, boxes, scores = [], [], [], [], []
    for ci in selected:
        rh_, rw_ = int(cand_h[ci]), int(cand_w[ci])
        cell = -np.ones(K, np.int32)
        larg = -np.ones(E, np.int32)
        cell[0] = rh_ * W + rw_
        for ei in order:
            s, t = EDGES[ei]
            if cell[s] < 0:
                continue
            i_h, i_w = divmod(int(cell[s]), W)
            win = e[ei, :, :, i_h, i_w]                       # [sH, sW], row-major argmax
            u = int(np.argmax(win))                            # first maximum (datatest.py:113)
            larg[ei] = u
            if margins is not None:
                top2 = np.partition(win.reshape(-1).astype(np.float64), -2)[-2:]
                _note(margins, "argmax", top2[1] - top2[0])
            j_h = i_h + u // sW - sH // 2
            j_w = i_w + u % sW - sW // 2
            if j_h < 0 or j_w < 0 or j_h >= H or j_w >= W:
                continue
            _note(margins, "hop", abs(float(delta[t, j_h, j_w]) - float(thr)))
            if delta[t, j_h, j_w] < thr:
                continue
            cell[t] = j_h * W + j_w
        if min_kp <= int((cell >= 0).sum()) - 1:
            bb = np.zeros((K, 4), np.float32)
            sc = np.zeros(K, np.float32)
            for k in range(K):
                if cell[k] >= 0:
                    kh, kw = divmod(int(cell[k]), W)
                    bb[k] = bbox[k, kh, kw]
                    sc[k] = delta[k, kh, kw]
            roots.append(cell[0]); kp_cells.append(cell); limb_args.append(larg)
            boxes.append(bb); scores.append(sc)
    n = len(roots)
    return dict(
        n=n,
        root_cell=np.asarray(roots, np.int32).reshape(n),
        kp_cell=np.asarray(kp_cells, np.int32).reshape(n, K),
        limb_arg=np.asarray(limb_args, np.int32).reshape(n, E),
        bbox=np.asarray(boxes, np.float32).reshape(n, K, 4),
        score=np.asarray(scores, np.float32).reshape(n, K),
        cand=cand, selected=selected,
    )


def humans_from_compact(res):
    """Compact result -> the reference's (humans, scores) lists of dicts (datatest.py:98-132)."""
    humans, scores = [], []
    for i in range(res["n"]):
        hm, sc = {}, {}
        for k in range(K):
            if res["kp_cell"][i, k] >= 0:
                hm[k] = res["bbox"][i, k].copy()
                sc[k] = res["score"][i, k]
        humans.append(hm)
        scores.append(sc)
    return humans, scores


def limb_argmax_dense(head: np.ndarray, local_grid=(21, 21)) -> np.ndarray:
    """Dense first-index argmax over the sH*sW window for every (edge, cell): int32[E,H,W]."""
    sW, sH = local_grid
    C, H, W = head.shape
    e = head[6 * K:].reshape(E, sH * sW, H, W)
    return np.argmax(e, axis=1).astype(np.int32)
