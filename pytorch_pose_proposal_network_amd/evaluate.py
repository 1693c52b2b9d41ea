"""Multi-person AP evaluation with the reference's semantics.

The reference scores a model with poseval-style code: datatest.evaluation (datatest.py:278-371) turns its
`pck_object` lists into annotation dicts, eval_helpers.assignGTmulti (eval_helpers.py:300-468) matches predicted to
ground-truth people by PCKh@0.5, evaluateAP.computeMetrics (evaluateAP.py:9-35) builds one precision/recall curve per
joint (eval_helpers.computeRPC / VOCap, eval_helpers.py:135-172) and eval_helpers.getCum (eval_helpers.py:103-114)
folds the 17 joints + mean into the 8 numbers main.py plots (head, shoulder, elbow, wrist, hip, knee, ankle, total).

This module is the same computation on arrays instead of nested dicts (host NumPy: the reference's is host Python
too, and it runs once per epoch on a few hundred people).  `evaluation(pck_object)` takes exactly the reference's
argument -- [fnames, gt_kps, humans, scores, gt_bboxes, is_visible, size] -- where humans/scores are what
decode.DecodeResult.to_humans() returns.  Quirks reproduced on purpose: every predicted person carries all 17 joints
(a missing one is the point (0, 0) with score 0, datatest.py:314-325); every ground-truth keypoint counts as annotated
(`is_visible` is ignored, datatest.py:338-340); frames without ground-truth people are dropped together with their
predictions (cleanupData, eval_helpers.py:202-218); the head size is 0.6 x the diagonal of the instance box whose
corners use floor division (datatest.py:331-334, eval_helpers.py:82-84); the f32 arithmetic of the reference's NumPy
scalars is kept so that distances exactly at the 0.5 threshold fall on the same side.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

N_JOINTS = 17
# eval_helpers.Joint (eval_helpers.py:11-29): joint ids of the 17 keypoints after the instance "keypoint"
_J = dict(left_shoulder=0, right_shoulder=1, left_elbow=2, right_elbow=3, left_wrist=4, right_wrist=5, left_hip=6,
          right_hip=7, left_knee=8, right_knee=9, left_ankle=10, right_ankle=11, thorax=12, pelvis=13, neck=14,
          top=15, stomach=16)
_CUM = [("top", "neck"), ("right_shoulder", "left_shoulder"), ("right_elbow", "left_elbow"),
        ("right_wrist", "left_wrist"), ("right_hip", "left_hip"), ("right_knee", "left_knee"),
        ("right_ankle", "left_ankle")]


def _pred_arrays(humans, scores):
    """-> (xy f32[P,17,2] joint centres, sc f32[P,17]) as datatest.py:309-327 builds them."""
    P = len(humans)
    xy = np.zeros((P, N_JOINTS, 2), np.float32)
    sc = np.zeros((P, N_JOINTS), np.float32)
    for i, (person, score) in enumerate(zip(humans, scores)):
        for num in range(1, N_JOINTS + 1):
            if num in person:
                b = np.asarray(person[num], np.float32)
                xy[i, num - 1, 1] = (b[0] + b[2]) / np.float32(2)          # y
                xy[i, num - 1, 0] = (b[1] + b[3]) / np.float32(2)          # x
                sc[i, num - 1] = np.float32(score[num])
    return xy, sc


def _head_sizes(gt_bboxes):
    """0.6 * || (x2-x1, y2-y1) || with x1 = cx - w//2 ... (datatest.py:329-334, eval_helpers.py:82-84), in f32."""
    out = np.zeros(len(gt_bboxes), np.float32)
    for i, (cx, cy, w, h) in enumerate(gt_bboxes):
        cx, cy, w, h = (np.float32(v) for v in (cx, cy, w, h))
        x1, x2 = cx - w // 2, cx + w // 2
        y1, y2 = cy - h // 2, cy + h // 2
        d = np.subtract(np.array([x2, y2], np.float32), np.array([x1, y1], np.float32))
        out[i] = np.float32(0.6) * np.linalg.norm(d)
    return out


def assign_gt_multi(frames, dist_thresh: float = 0.5):
    """eval_helpers.assignGTmulti on arrays.  frames: list of (pred_xy, pred_sc, gt_xy f32[G,17,2], head f32[G]).
    Returns (scores[j][img] arrays, labels[j][img] arrays, nGT[17, n_images])."""
    n = len(frames)
    scores_all = [[np.zeros(0, np.float32) for _ in range(n)] for _ in range(N_JOINTS)]
    labels_all = [[np.zeros(0, np.int8) for _ in range(n)] for _ in range(N_JOINTS)]
    n_gt = np.zeros((N_JOINTS, n))
    for img, (pxy, psc, gxy, head) in enumerate(frames):
        P, G = len(pxy), len(gxy)
        if P and G:
            diff = gxy[None, :, :, :] - pxy[:, None, :, :]                   # f32 [P,G,17,2]
            dist = np.sqrt((diff * diff).sum(-1, dtype=np.float32)) / head[None, :, None]
            match = dist.astype(np.float64) <= dist_thresh
            pck = match.sum(2).astype(np.float64) / float(N_JOINTS)          # every GT joint is annotated
            best_gt = np.argmax(pck, axis=1)                                 # preserve best GT match only
            keep = np.zeros_like(pck)
            keep[np.arange(P), best_gt] = pck[np.arange(P), best_gt]
            pr_to_gt = np.argmax(keep, axis=0)
            pr_to_gt[np.max(keep, axis=0) == 0] = -1
            for p in range(P):
                hit = np.flatnonzero(pr_to_gt == p)
                lab = match[p, hit[0]] if hit.size else np.zeros(N_JOINTS, bool)
                assert hit.size <= 1
                for j in range(N_JOINTS):
                    scores_all[j][img] = np.append(scores_all[j][img], psc[p, j])
                    labels_all[j][img] = np.append(labels_all[j][img], lab[j])
        # (frames without GT never reach here: cleanupData drops them; frames without predictions add nothing)
        n_gt[:, img] += G
    return scores_all, labels_all, n_gt


def _compute_rpc(scores, labels, total_pos):                                # eval_helpers.py:135-151
    idxs = np.array(scores).argsort()[::-1]
    ls = np.asarray(labels)[idxs]
    npos = np.cumsum(ls == 1)
    k = np.arange(1, len(idxs) + 1)
    return 1.0 * npos / k, 1.0 * npos / total_pos


def _voc_ap(rec, prec):                                                     # eval_helpers.py:155-171
    mpre = np.zeros(2 + len(prec))
    mpre[1:len(prec) + 1] = prec
    mrec = np.zeros(2 + len(rec))
    mrec[1:len(rec) + 1] = rec
    mrec[len(rec) + 1] = 1.0
    for i in range(mpre.size - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    i = np.flatnonzero(~np.equal(mrec[1:], mrec[:-1])) + 1
    return np.sum((mrec[i] - mrec[i - 1]) * mpre[i])


def compute_metrics(scores_all, labels_all, n_gt):                          # evaluateAP.py:9-35
    ap = np.zeros(N_JOINTS + 1)
    for j in range(N_JOINTS):
        scores = np.concatenate([np.asarray(s, np.float64) for s in scores_all[j]]) if n_gt.shape[1] else np.zeros(0)
        labels = np.concatenate([np.asarray(l, np.float64) for l in labels_all[j]]) if n_gt.shape[1] else np.zeros(0)
        total = n_gt[j].sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            prec, rec = _compute_rpc(scores, labels, total)
        if len(prec) > 0:
            ap[j] = _voc_ap(rec, prec) * 100
    valid = ~np.isnan(ap[:N_JOINTS])
    ap[N_JOINTS] = ap[:N_JOINTS][valid].mean()
    return ap


def get_cum(ap) -> List[float]:                                             # eval_helpers.py:103-114
    cum = [float(np.mean([ap[_J[a]], ap[_J[b]]])) for a, b in _CUM]
    return cum + [float(ap[N_JOINTS])]


def evaluation(list_: Sequence) -> List[float]:
    """datatest.evaluation (datatest.py:278-371): the 8 AP values [head, shoulder, elbow, wrist, hip, knee, ankle,
    total] for pck_object = [fnames, gt_kps, humans, scores, gt_bboxes, is_visible, size]."""
    _, gt_kps_list, humans_list, scores_list, gt_bboxes_list = list_[0], list_[1], list_[2], list_[3], list_[4]
    frames = []
    for humans, scores, gt_kps, gt_bboxes in zip(humans_list, scores_list, gt_kps_list, gt_bboxes_list):
        if len(gt_bboxes) == 0:
            continue                                                        # cleanupData: no GT people -> frame dropped
        pxy, psc = _pred_arrays(humans, scores)
        gxy = np.asarray(gt_kps, np.float32).reshape(len(gt_bboxes), N_JOINTS, 2)
        frames.append((pxy, psc, gxy, _head_sizes(gt_bboxes)))
    return get_cum(compute_metrics(*assign_gt_multi(frames)))


def people_as_ground_truth(res: dict):
    """One image's compact people list (DecodeResult.to_host() / oracle / fixture format: n, kp_cell [n,K], bbox
    [n,K,4] as [ymin,xmin,ymax,xmax], score [n,K]) as the ground truth of `evaluation`: every keypoint at its box centre
    (a keypoint the person does not have sits at (0, 0), which is also where datatest.py:314-325 puts a missing
    PREDICTED joint, so an identical people list scores 100), the instance box (keypoint 0) as (cx, cy, w, h)."""
    n = int(res["n"])
    humans, scores = [], []
    for i in range(n):
        hm = {k: np.asarray(res["bbox"][i, k], np.float32) for k in range(res["kp_cell"].shape[1]) if res["kp_cell"][i, k] >= 0}
        humans.append(hm)
        scores.append({k: np.float32(res["score"][i, k]) for k in hm})
    gxy, _ = _pred_arrays(humans, scores)
    boxes = []
    for i in range(n):
        ymin, xmin, ymax, xmax = (float(v) for v in res["bbox"][i, 0])
        boxes.append(((xmin + xmax) / 2, (ymin + ymax) / 2, xmax - xmin, ymax - ymin))
    return humans, scores, gxy.reshape(n, N_JOINTS * 2), boxes


def ap_against_people(expected: Sequence[dict], got: Sequence[dict]) -> List[float]:
    """The 8 AP values of `got` (per-image compact people lists) scored against `expected` taken as ground truth
    (people_as_ground_truth), with the reference's own matcher and metric (`evaluation`): what a reduced-precision
    mode costs in the task metric, instead of exact-match counts."""
    fnames, gt_kps, humans, scores, gt_boxes, vis, size = [], [], [], [], [], [], []
    for i, (e, g) in enumerate(zip(expected, got)):
        _, _, gxy, boxes = people_as_ground_truth(e)
        hm, sc, _, _ = people_as_ground_truth(g)
        fnames.append(f"frame{i}"); gt_kps.append(gxy); humans.append(hm); scores.append(sc); gt_boxes.append(boxes)
        vis.append(None); size.append(None)
    return evaluation([fnames, gt_kps, humans, scores, gt_boxes, vis, size])
