"""Skeleton constants and grid geometry of the Pose Proposal Network hot path.

Restates the *values* of the reference's ``config.py:3-82`` (KEYPOINT_NAMES, EDGES,
DIRECTED_GRAPHS, EPSILON) and the geometry globals of ``datatest.py:53-60`` /
``model.py:54-64``.  The five directed chains of the reference share prefixes, so the
limb parse is an ordinary tree walk from keypoint 0; the tree form below (parent edge,
depth-ordered edge list) is what the HIP parse kernel keeps in ``__constant__`` memory.
"""
from __future__ import annotations

KEYPOINT_NAMES = [
    "instance",
    "left_shoulder", "right_shoulder",
    "left_elbow", "right_elbow",
    "left_wrist", "right_wrist",
    "left_hip", "right_hip",
    "left_knee", "right_knee",
    "left_ankle", "right_ankle",
    "thorax", "pelvis", "neck", "top", "stomach",
]

EDGES_BY_NAME = [
    ("instance", "neck"), ("neck", "thorax"),
    ("thorax", "left_shoulder"), ("left_shoulder", "left_elbow"), ("left_elbow", "left_wrist"),
    ("thorax", "right_shoulder"), ("right_shoulder", "right_elbow"), ("right_elbow", "right_wrist"),
    ("thorax", "stomach"), ("stomach", "pelvis"),
    ("pelvis", "left_hip"), ("pelvis", "right_hip"),
    ("left_hip", "left_knee"), ("right_hip", "right_knee"),
    ("left_knee", "left_ankle"), ("right_knee", "right_ankle"),
    ("instance", "top"),
]

EDGES = [[KEYPOINT_NAMES.index(s), KEYPOINT_NAMES.index(d)] for s, d in EDGES_BY_NAME]

TRACK_ORDERS = [
    ["instance", "neck", "thorax", "left_shoulder", "left_elbow", "left_wrist"],
    ["instance", "neck", "thorax", "right_shoulder", "right_elbow", "right_wrist"],
    ["instance", "neck", "thorax", "stomach", "pelvis", "left_hip", "left_knee", "left_ankle"],
    ["instance", "neck", "thorax", "stomach", "pelvis", "right_hip", "right_knee", "right_ankle"],
    ["instance", "top"],
]


def _chains():
    out = []
    names = [list(e) for e in EDGES_BY_NAME]
    for order in TRACK_ORDERS:
        es = [names.index([a, b]) for a, b in zip(order[:-1], order[1:])]
        ts = [KEYPOINT_NAMES.index(b) for b in order[1:]]
        out.append([es, ts])
    return out


# [[edge index list], [target keypoint list]] per chain -- config.py:67-80
DIRECTED_GRAPHS = _chains()

EPSILON = 1e-6

K = len(KEYPOINT_NAMES)          # 18
E = len(EDGES)                   # 17
ROOT_NODE = 0

# Geometry at the BASELINE configs (datatest.py:53-60, model.py:54-64)
INSIZE = (384, 384)              # (inW, inH)
OUTSIZE = (24, 24)               # (outW, outH)
LOCAL_GRID_SIZE = (21, 21)       # (sW, sH)
DETECTION_THRESH = 0.15          # rt_test.py:133
NMS_THRESH = 0.3                 # datatest.py:94
MIN_NUM_KEYPOINTS = 1            # datatest.py:74

# Input normalisation (aug.py:149-153, rt_test.py:90-101): (u8 - mean)/std, NO /255.
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def lastsize(k: int = K, e: int = E, local_grid=LOCAL_GRID_SIZE) -> int:
    """Head channel count 6K + sW*sH*E  (model.py:64) = 7605 at the defaults."""
    sw, sh = local_grid
    return 6 * k + sw * sh * e


def tree_tables():
    """Tree form of DIRECTED_GRAPHS.

    Returns (edge_src, edge_dst, edge_order): for edge e, the limb goes from keypoint
    edge_src[e] to edge_dst[e]; edge_order lists the edges parent-before-child so a
    single pass visits every keypoint after its parent.  Each edge 0..16 is a tree edge
    exactly once.
    """
    src = [s for s, _ in EDGES]
    dst = [d for _, d in EDGES]
    seen, order = set(), []
    for es, _ in DIRECTED_GRAPHS:
        for e in es:
            if e not in seen:
                seen.add(e)
                order.append(e)
    assert sorted(order) == list(range(E))
    return src, dst, order
