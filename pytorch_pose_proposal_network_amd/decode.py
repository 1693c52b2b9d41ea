"""Host mirror of the reference's decode surface (datatest.py:62-160) over the HIP kernels.

* ``decode_heads(head)``                -- the fast path: device head tensor [B,7605,H,W] in, compact
                                           device buffers out (no D2H of the head, unlike rt_test.py:109-120).
* ``get_humans_by_feature(...)``        -- same name/arguments/return as datatest.py:74: B=1 arrays
                                           ``delta,x,y,w,h,e`` -> ``(humans, scores)`` lists of dicts.
* ``non_maximum_suppression(...)``      -- same as datatest.py:134.
* ``restore_xy`` / ``restore_size``     -- datatest.py:63-71 (host helpers for callers that draw).

Everything that computes runs in libppn.so; there is no NumPy fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import config as cfg
from . import lib as L


def make_cfg(out_hw=(24, 24), insize_hw=(384, 384), local_grid=(21, 21), det_thr=0.15, nms_thr=0.3, min_kp=1,
             max_humans: Optional[int] = None) -> L.DecodeCfg:
    c = L.DecodeCfg()
    c.K, c.E = cfg.K, cfg.E
    c.sW, c.sH = local_grid
    c.H, c.W = out_hw
    c.inH, c.inW = insize_hw
    c.det_thr, c.nms_thr, c.min_kp = det_thr, nms_thr, min_kp
    c.max_humans = max_humans if max_humans is not None else c.H * c.W
    src, dst, order = cfg.tree_tables()
    for i in range(cfg.E):
        c.edge_src[i], c.edge_dst[i], c.edge_order[i] = src[i], dst[i], order[i]
    return c


@dataclass
class DecodeResult:
    """Compact decode output, all on the device (see include/ppn.h ppn_decode)."""
    count: torch.Tensor      # i32 [B]
    kp_cell: torch.Tensor    # i32 [B, max_humans, K]
    limb_arg: torch.Tensor   # i32 [B, max_humans, E]
    bbox: torch.Tensor       # f32 [B, max_humans, K, 4]
    score: torch.Tensor      # f32 [B, max_humans, K]
    max_humans: int

    def to_host(self):
        """One small D2H (<= 80 KB/img): list per image of dicts of numpy arrays (oracle format)."""
        cnt = self.count.cpu().numpy()
        m = int(min(cnt.max(initial=0), self.max_humans))
        kp = self.kp_cell[:, :m].cpu().numpy()
        la = self.limb_arg[:, :m].cpu().numpy()
        bb = self.bbox[:, :m].cpu().numpy()
        sc = self.score[:, :m].cpu().numpy()
        out = []
        for b in range(len(cnt)):
            n = int(min(cnt[b], self.max_humans))
            out.append(dict(n=n, root_cell=kp[b, :n, 0].copy(), kp_cell=kp[b, :n], limb_arg=la[b, :n],
                            bbox=bb[b, :n], score=sc[b, :n]))
        return out

    def to_host_async(self, stage: "HostStage") -> "HostStage":
        """Queue the D2H of the compact result on the CURRENT stream into `stage`'s pinned buffers (the first
        `stage.cap` people slots per image: 32 x 64 slots = 1.0 MB) without waiting for the counts; call
        `stage.unpack()` once an event recorded behind this call has completed.  No host synchronisation here."""
        cap = stage.cap
        if cap > self.max_humans:
            raise ValueError(f"HostStage capacity {cap} exceeds the decoder's max_humans {self.max_humans}")
        dv = stage.device_side(self.count.device)
        # gather the first `cap` slots into contiguous device buffers (a strided D2H would be staged by torch anyway,
        # synchronously), then one asynchronous copy per array
        dv["count"].copy_(self.count)
        dv["kp_cell"].copy_(self.kp_cell[:, :cap]); dv["limb_arg"].copy_(self.limb_arg[:, :cap])
        dv["bbox"].copy_(self.bbox[:, :cap]); dv["score"].copy_(self.score[:, :cap])
        stage.flat.copy_(stage._dev_flat, non_blocking=True)          # ONE D2H for the whole compact result
        stage.source = self
        return stage

    def to_humans(self) -> List[Tuple[list, list]]:
        """Per image the reference's (humans, scores): lists of {kp: f32[4]} / {kp: f32} (datatest.py:98-132)."""
        res = []
        for r in self.to_host():
            humans, scores = [], []
            for i in range(r["n"]):
                hm, sm = {}, {}
                for k in range(cfg.K):
                    if r["kp_cell"][i, k] >= 0:
                        hm[k] = r["bbox"][i, k].copy()
                        sm[k] = r["score"][i, k]
                humans.append(hm)
                scores.append(sm)
            res.append((humans, scores))
        return res


class HostStage:
    """Pinned host buffers for the compact decode result of one batch (DecodeResult.to_host_async): ONE pinned
    allocation and one device mirror holding count | kp_cell | limb_arg | bbox | score back to back, so the read-back
    is a single D2H copy per batch (five separate copies cost five trips through the copy queue per step)."""

    def __init__(self, batch: int, cap: int = 64):
        self.cap = cap
        self._shapes = [("count", (batch,), torch.int32), ("kp_cell", (batch, cap, cfg.K), torch.int32),
                        ("limb_arg", (batch, cap, cfg.E), torch.int32), ("bbox", (batch, cap, cfg.K, 4), torch.float32),
                        ("score", (batch, cap, cfg.K), torch.float32)]
        self._words = sum(int(np.prod(shp)) for _, shp, _ in self._shapes)
        self.flat = torch.empty(self._words, dtype=torch.int32).pin_memory()
        for name, view in self._views(self.flat).items():
            setattr(self, name, view)
        self.source: Optional[DecodeResult] = None
        self._dev = None
        self._dev_flat = None

    def _views(self, flat):
        out, o = {}, 0
        for name, shp, dt in self._shapes:
            n = int(np.prod(shp))
            v = flat[o:o + n]
            out[name] = (v.view(torch.float32) if dt == torch.float32 else v).view(shp)
            o += n
        return out

    def device_side(self, device):
        if self._dev is None:
            self._dev_flat = torch.empty(self._words, dtype=torch.int32, device=device)
            self._dev = self._views(self._dev_flat)
        return self._dev

    def unpack(self):
        """Per-image dicts like DecodeResult.to_host() (the copies must have completed).  An image with more than
        `cap` people falls back to the synchronous full read-back."""
        cnt = self.count.numpy()
        if int(cnt.max(initial=0)) > self.cap:
            return self.source.to_host()
        kp, la, bb, sc = self.kp_cell.numpy(), self.limb_arg.numpy(), self.bbox.numpy(), self.score.numpy()
        out = []
        for b in range(len(cnt)):
            n = int(cnt[b])
            out.append(dict(n=n, root_cell=kp[b, :n, 0].copy(), kp_cell=kp[b, :n].copy(), limb_arg=la[b, :n].copy(),
                            bbox=bb[b, :n].copy(), score=sc[b, :n].copy()))
        return out


def people_agreement(expected: dict, got: dict):
    """How much of one image's `expected` compact result (DecodeResult.to_host() / oracle format) `got` reproduces,
    people matched by root cell: (expected people, reproduced exactly = same root, every keypoint cell and every limb
    arg-max, same root found, equal keypoint cells among same-root people, keypoint cells compared)."""
    def people(res):
        return {int(res["kp_cell"][i, 0]): (res["kp_cell"][i], res["limb_arg"][i]) for i in range(int(res["n"]))}
    pe, pg = people(expected), people(got)
    same = [r for r in pe if r in pg]
    exact = sum(1 for r in same if np.array_equal(pe[r][0], pg[r][0]) and np.array_equal(pe[r][1], pg[r][1]))
    kp_eq = sum(int((pe[r][0] == pg[r][0]).sum()) for r in same)
    return len(pe), exact, len(same), kp_eq, len(same) * cfg.K


class Decoder:
    """Owns the scratch/output buffers for a fixed (batch, grid) so repeated calls allocate nothing."""

    def __init__(self, batch: int, out_hw=(24, 24), insize_hw=(384, 384), local_grid=(21, 21), det_thr=0.15,
                 nms_thr=0.3, min_kp=1, max_humans: Optional[int] = None, device="cuda"):
        self.lib = L.load()
        self.cfg = make_cfg(out_hw, insize_hw, local_grid, det_thr, nms_thr, min_kp, max_humans)
        self.batch = batch
        c = self.cfg
        self.channels = 6 * c.K + c.E * c.sH * c.sW
        ws = self.lib.ppn_decode_workspace_bytes(C.byref(c), batch)
        dev = torch.device(device)
        self.workspace = torch.empty(max(ws, 16) // 4, dtype=torch.int32, device=dev)
        # root candidates + pairwise-IoU bit matrices of the fused path (ppn_decode_fused_ws)
        wsf = self.lib.ppn_decode_fused_workspace_bytes(C.byref(c), batch)
        self._fused_ws = torch.empty(max(wsf, 16) // 4, dtype=torch.int32, device=dev)
        m = c.max_humans
        self.out = DecodeResult(
            count=torch.zeros(batch, dtype=torch.int32, device=dev),
            kp_cell=torch.empty(batch, m, c.K, dtype=torch.int32, device=dev),
            limb_arg=torch.empty(batch, m, c.E, dtype=torch.int32, device=dev),
            bbox=torch.empty(batch, m, c.K, 4, dtype=torch.float32, device=dev),
            score=torch.empty(batch, m, c.K, dtype=torch.float32, device=dev),
            max_humans=m)

    def __call__(self, head: torch.Tensor) -> DecodeResult:
        c = self.cfg
        if not (head.is_cuda and head.dtype == torch.float32 and head.is_contiguous()):
            raise ValueError("head must be a contiguous float32 CUDA tensor [B,C,H,W]")
        if tuple(head.shape) != (self.batch, self.channels, c.H, c.W):
            raise ValueError(f"head shape {tuple(head.shape)} != {(self.batch, self.channels, c.H, c.W)}")
        o = self.out
        L.check(self.lib.ppn_decode(C.byref(c), head.data_ptr(), self.batch, self.workspace.data_ptr(),
                                    o.count.data_ptr(), o.kp_cell.data_ptr(), o.limb_arg.data_ptr(),
                                    o.bbox.data_ptr(), o.score.data_ptr(), L.current_stream_ptr()), "ppn_decode")
        return o

    def decode_fused(self, unary: torch.Tensor, keys: torch.Tensor) -> DecodeResult:
        """Decode from the fused head conv's outputs (PoseProposalNet.forward_u8(..., fused_decode=True))."""
        c = self.cfg
        if tuple(unary.shape) != (self.batch, 6 * c.K, c.H, c.W) or unary.dtype != torch.float32:
            raise ValueError(f"unary must be f32 {(self.batch, 6 * c.K, c.H, c.W)}")
        if tuple(keys.shape) != (self.batch, c.E, c.H, c.W) or keys.dtype != torch.int64:
            raise ValueError(f"keys must be i64 {(self.batch, c.E, c.H, c.W)}")
        o = self.out
        L.check(self.lib.ppn_decode_fused_ws(C.byref(c), unary.data_ptr(), keys.data_ptr(), self.batch,
                                             self._fused_ws.data_ptr(), o.count.data_ptr(), o.kp_cell.data_ptr(),
                                             o.limb_arg.data_ptr(), o.bbox.data_ptr(), o.score.data_ptr(),
                                             L.current_stream_ptr()), "ppn_decode_fused_ws")
        return o

    def limb_argmax(self, head: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        out = self.workspace[: self.batch * c.E * c.H * c.W].view(self.batch, c.E, c.H, c.W)
        L.check(self.lib.ppn_limb_argmax(C.byref(c), head.data_ptr(), self.batch, out.data_ptr(),
                                         L.current_stream_ptr()), "ppn_limb_argmax")
        return out


def decode_heads(head: torch.Tensor, insize_hw=(384, 384), local_grid=(21, 21), detection_thresh=0.15,
                 nms_thresh=0.3, min_num_keypoints=1, max_humans: Optional[int] = None) -> DecodeResult:
    """Decode a device head tensor [B, 6K+E*sH*sW, H, W] (model.forward output) in place on the GPU."""
    b, _, h, w = head.shape
    dec = Decoder(b, (h, w), insize_hw, local_grid, detection_thresh, nms_thresh, min_num_keypoints, max_humans,
                  head.device)
    return dec(head)


# ----------------------------------------------------------------------------------------------
# Reference-shaped functions (datatest.py)
# ----------------------------------------------------------------------------------------------
insize = cfg.INSIZE
outsize = cfg.OUTSIZE
local_grid_size = cfg.LOCAL_GRID_SIZE


def restore_xy(x, y):
    """datatest.py:63-67 (host helper, NumPy): grid-relative offsets -> pixel coordinates."""
    outW, outH = outsize
    gridW, gridH = int(insize[0] / outW), int(insize[1] / outH)
    X, Y = np.meshgrid(np.arange(outW, dtype=np.float32), np.arange(outH, dtype=np.float32))
    return (x + X) * gridW, (y + Y) * gridH


def restore_size(w, h):
    """datatest.py:69-71."""
    return insize[0] * w, insize[1] * h


def _dev(a) -> torch.Tensor:
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device="cuda", dtype=torch.float32)


def get_humans_by_feature(delta, x, y, w, h, e, detection_thresh=0.15, min_num_keypoints=1):
    """Drop-in for datatest.py:74-132 (single image).  `delta` is resp*conf as at rt_test.py:130.

    The arrays are assembled into one head tensor on the device with resp=delta, conf=1 (delta*1 is exact),
    decoded by the HIP kernels, and returned as the reference's (humans, scores).
    """
    delta, x, y, w, h = (_dev(a) for a in (delta, x, y, w, h))
    e = _dev(e)
    K, H, W = delta.shape
    E, sH, sW = e.shape[0], e.shape[1], e.shape[2]
    head = torch.cat([delta, torch.ones_like(delta), x, y, w, h, e.reshape(E * sH * sW, H, W)], 0).unsqueeze(0)
    inW, inH = insize
    res = decode_heads(head.contiguous(), (inH, inW), (sW, sH), detection_thresh, 0.3, min_num_keypoints)
    return res.to_humans()[0]


def non_maximum_suppression(bbox, thresh, score=None, limit=None):
    """Drop-in for datatest.py:134-160; returns int32 indices (NumPy), computed by ppn_nms on the GPU."""
    lib = L.load()
    bb = _dev(bbox).contiguous()
    n = bb.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int32)
    sc = _dev(score).contiguous() if score is not None else None
    sel = torch.empty(n, dtype=torch.int32, device=bb.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=bb.device)
    L.check(lib.ppn_nms(bb.data_ptr(), sc.data_ptr() if sc is not None else None, n, float(thresh),
                        int(limit) if limit is not None else 0, sel.data_ptr(), cnt.data_ptr(),
                        L.current_stream_ptr()), "ppn_nms")
    return sel[: int(cnt.item())].cpu().numpy().astype(np.int32)
