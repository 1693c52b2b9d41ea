"""Mirror of the reference's realtime driver entry points (rt_test.py:52-147) over the HIP path.

    model, outsize, local_grid_size = network(resume="PPN_model_best.pth.tar", image_size=384)
    humans, scores = inference(image_u8_hwc, model, outsize, local_grid_size)

``inference`` keeps the reference's argument list; what it does differently is *where* things run: the
uint8 frame is uploaded once (442 KB), normalisation + the whole conv stack + head run as HIP kernels, the
head tensor never leaves the device (the reference copies all 17.5 MB of it to the host in seven slices,
rt_test.py:109-120) and only the compact people list (<= 80 KB) comes back.  Webcam capture, matplotlib
animation and drawing (rt_test.py:150-205, datatest.py:162-232) are out of scope: ``inference`` returns the
reference's ``(humans, scores)`` instead of a PIL image; pass ``draw=callable`` to post-process on the host.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from . import config as cfg
from . import decode as D
from . import drn
from .model import PoseProposalNet


def network(resume: Optional[str] = None, image_size: int = 384, arch: str = "drn_d_22",
            compute_dtype: str = "float32", state_dict=None):
    """rt_test.py:52-85: build D-22 PPN, derive ``outsize`` and load ``checkpoint['state_dict']``.

    The reference discovers ``outsize`` with a dummy forward (rt_test.py:65-68); the network is fully
    convolutional with total stride 16, so it is ``image_size // 16`` here."""
    local_grid_size = (21, 21)
    outsize = (image_size // 16, image_size // 16)
    model = PoseProposalNet(getattr(drn, arch)(), insize=(image_size, image_size), outsize=outsize,
                            local_grid_size=local_grid_size, compute_dtype=compute_dtype).cuda()
    if resume is not None:
        ckpt = torch.load(resume, map_location="cpu")
        state_dict = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    if state_dict is not None:
        model.load_state_dict(state_dict)
    return model, outsize, local_grid_size


def ingest_frames(frames_bgr: torch.Tensor, size: int = 384, out: Optional[torch.Tensor] = None,
                  flip: bool = True, swap_rb: bool = True) -> torch.Tensor:
    """The cv2 calls of rt_test.py:150-157 on the device: u8 BGR camera frames [B,Hs,Ws,3] (CUDA) ->
    cv2.resize(dsize=(size,size)) -> flip both axes -> BGR2RGB -> u8 RGB [B,size,size,3].  `out` may be the model's
    own input buffer (``model.input_buffer(B, size, size, fused_decode=..., slot=...)``): forward_u8(out) then runs
    without any further copy."""
    from . import lib as L
    if not (frames_bgr.is_cuda and frames_bgr.dtype == torch.uint8 and frames_bgr.dim() == 4 and frames_bgr.shape[3] == 3):
        raise ValueError("ingest_frames expects a uint8 CUDA tensor [B,Hs,Ws,3]")
    x = frames_bgr.contiguous()
    b, hs, ws, _ = x.shape
    if out is None:
        out = torch.empty(b, size, size, 3, dtype=torch.uint8, device=x.device)
    if tuple(out.shape) != (b, size, size, 3) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous uint8 tensor {(b, size, size, 3)}")
    L.check(L.load().ppn_ingest_frames(x.data_ptr(), b, hs, ws, out.data_ptr(), size, size, int(flip), int(swap_rb),
                                       L.current_stream_ptr()), "ppn_ingest_frames")
    return out


def grab_frame(cap, size: int = 384, out: Optional[torch.Tensor] = None):
    """rt_test.py:150-157 with the resize / flips / colour conversion on the GPU: `cap` is anything with
    cv2.VideoCapture's ``read() -> (ret, frame_bgr_u8_hwc)``.  Returns ``(ret, frame)`` where frame is a u8 RGB
    [1,size,size,3] CUDA tensor (pass it to ``inference`` / ``model.forward_u8``)."""
    ret, frame = cap.read()
    if not ret or frame is None:
        return ret, None
    f = np.ascontiguousarray(np.asarray(frame))
    if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
        raise ValueError("cap.read() must deliver a uint8 HxWx3 BGR frame")
    return ret, ingest_frames(torch.from_numpy(f).unsqueeze(0).cuda(non_blocking=True), size, out)


def inference(image, model: PoseProposalNet, outsize, local_grid_size, detection_thresh: float = 0.15,
              draw: Optional[Callable] = None):
    """rt_test.py:87-147 for one RGB frame (u8 [S,S,3] array / PIL image / tensor).

    Returns ``(humans, scores)`` exactly as datatest.get_humans_by_feature does (lists of dicts
    keypoint -> [ymin,xmin,ymax,xmax] / keypoint -> delta), or ``draw(image, humans, scores)`` if given."""
    model.eval()
    if isinstance(image, torch.Tensor) and image.is_cuda:             # what grab_frame() returns: already on the device
        frames = image if image.dim() == 4 else image.unsqueeze(0)
        if frames.dtype != torch.uint8 or frames.shape[0] != 1 or frames.shape[3] != 3:
            raise ValueError("inference expects a uint8 [1,H,W,3] / [H,W,3] RGB frame")
        hw = (frames.shape[1], frames.shape[2])
    else:
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("inference expects a uint8 HxWx3 RGB frame")
        frames = torch.from_numpy(np.ascontiguousarray(img)).unsqueeze(0).cuda(non_blocking=True)
        hw = (img.shape[0], img.shape[1])
    head = model.forward_u8(frames)
    outW, outH = outsize
    res = D.decode_heads(head, insize_hw=hw, local_grid=local_grid_size,
                         detection_thresh=detection_thresh)
    humans, scores = res.to_humans()[0]
    if draw is not None:
        return draw(image, humans, scores)
    return humans, scores


def inference_batch(frames_u8: torch.Tensor, model: PoseProposalNet, decoder: Optional[D.Decoder] = None,
                    detection_thresh: float = 0.15) -> D.DecodeResult:
    """Batched device-side inference: u8 [B,S,S,3] CUDA tensor -> compact people lists on the device."""
    unary, keys = model.forward_u8(frames_u8, fused_decode=True)     # the head tensor is never materialised
    if decoder is None:
        b, _, h, w = unary.shape
        decoder = D.Decoder(b, (h, w), (frames_u8.shape[1], frames_u8.shape[2]), model.local_grid_size,
                            detection_thresh, device=unary.device)
    return decoder.decode_fused(unary, keys)


class InferencePipeline:
    """Two-deep software pipeline for batched serving: the conv stack of batch i+1 runs on the caller's stream
    while the latency-bound NMS + limb-parse kernel of batch i (one workgroup per image, 32 of 256 CUs) runs on a
    side stream.  Each in-flight batch owns its plan outputs and its Decoder (slot 0 / 1); HIP events order the
    hand-over in both directions, nothing blocks the host.

        pipe = InferencePipeline(model, batch, (S, S))
        for frames in source:                       # u8 [B,S,S,3] on the device
            done = pipe.submit(frames)              # DecodeResult of THIS batch, valid once done.ready is reached
        pipe.flush()                                # or torch.cuda.synchronize()

    `submit` returns the DecodeResult object of the batch just submitted; call `result.ready.synchronize()` (or
    make a stream wait on it) before reading it.  A slot is reused two submits later."""

    def __init__(self, model: PoseProposalNet, batch: int, insize_hw, detection_thresh: float = 0.15, device=None):
        self.model = model
        dev = device if device is not None else model.device
        h, w = insize_hw[0] // 16, insize_hw[1] // 16
        self.decoders = [D.Decoder(batch, (h, w), insize_hw, model.local_grid_size, detection_thresh, device=dev)
                         for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)                  # (a high-priority stream measured no better)
        self.fwd_done = [torch.cuda.Event() for _ in range(2)]
        self.dec_done = [None, None]
        self.k = 0

    def submit(self, frames_u8: torch.Tensor) -> D.DecodeResult:
        k = self.k
        self.k ^= 1
        main = torch.cuda.current_stream(frames_u8.device)
        if self.dec_done[k] is not None:
            main.wait_event(self.dec_done[k])                  # slot k's previous outputs have been consumed
        unary, keys = self.model.forward_u8(frames_u8, fused_decode=True, slot=k)
        self.fwd_done[k].record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.fwd_done[k])
            res = self.decoders[k].decode_fused(unary, keys)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.dec_done[k] = ev
        res.ready = ev
        return res

    def flush(self):
        self.side.synchronize()


_LANE_STREAMS: dict = {}


def _lane_streams(dev, lanes: int):
    """The lane streams of a device, created once per process and shared by every MultiLaneInference on it.  The runtime maps
    streams onto a handful of hardware queues (four by default); a pipeline whose lanes get fresh streams late in a process can
    find two of them on ONE queue, i.e. serialised: bench.py's `tile_policy_1` section, the third pipeline of its process, ran its
    two lanes at the one-lane rate (9.5 k instead of 11.6 k images/s, round 5).  Pipelines on the same device therefore reuse the
    first `lanes` streams of one pool (correct for any use -- a stream orders its work -- and what a server wants anyway)."""
    key = torch.device(dev)
    if key.index is None:
        key = torch.device(key.type, torch.cuda.current_device())
    pool = _LANE_STREAMS.setdefault(key, [])
    while len(pool) < lanes:
        pool.append(torch.cuda.Stream(device=key))
    return pool[:lanes]


class MultiLaneInference:
    """Independent inference lanes on separate HIP streams, batches go round-robin: while one lane is in a
    launch that cannot fill the GPU (the 14-22 us neck convolutions, the last partial round of workgroups of a
    layer, the one-workgroup-per-image parse kernel, the gap between two dependent launches) the other lane's
    workgroups take the idle CUs.  Each lane owns its plan outputs (slot) and Decoder; the input hand-over and the
    result hand-over are HIP events.  Results are bit-identical to the serial path."""

    def __init__(self, model: PoseProposalNet, batch: int, insize_hw, detection_thresh: float = 0.15, device=None,
                 lanes: int = 2, tile_policy: int = 0, shared_plan: Optional[bool] = None):
        self.model = model
        dev = device if device is not None else model.device
        h, w = insize_hw[0] // 16, insize_hw[1] // 16
        self.decoders = [D.Decoder(batch, (h, w), insize_hw, model.local_grid_size, detection_thresh, device=dev)
                         for _ in range(lanes)]
        self.streams = _lane_streams(dev, lanes)
        self._stages = [None] * lanes                     # pinned read-back buffers, created on first use
        self.k = 0
        # tile_policy 1: conv tiles chosen by efficiency alone instead of whole rounds of workgroups (process-wide
        # setting of libppn, restored by close()); measured +4 % with two lanes, but the per-launch (one in flight)
        # durations of those tiles are worse, so the default keeps the single-stream choice
        from . import lib as L
        self._policy = tile_policy
        L.check(L.load().ppn_set_conv_tile_policy(tile_policy), "ppn_set_conv_tile_policy")
        # csrc/conv64.hip (64 -> 64 3x3 with the filter bank in registers) shortens layer3's launches by ~7 us each when
        # one launch is in flight, but its workgroups own a CU's whole LDS and register file, so another lane's kernels
        # cannot share the CU: with several lanes the generic kernel gives the higher throughput (same-box A/B: 11.04 k
        # vs 10.96 k images/s with three lanes).  The choice travels in THIS pipeline's plans (ppn_conv_desc.flags), so
        # other plans, trainers and pipelines of the process -- and a user's PPN_CONV64 setting -- are untouched;
        # results are bit-identical either way.
        # shared_plan=True with ONE lane: the multi-lane plan (same kernels, same tiles) with one launch in flight -- what a
        # profiler run needs so that its per-kernel durations describe the kernels the multi-lane headline ran
        shared = lanes > 1 if shared_plan is None else bool(shared_plan)
        self._conv_flags = (L.PPN_CONV_NO_FILTER_BANK | L.PPN_CONV_SHARED_GPU) if shared else 0
        import os
        if shared and os.environ.get("PPN_LANES_LONE_TILES") == "1":      # A/B knob: keep the tiles that shorten a lone launch
            self._conv_flags = L.PPN_CONV_NO_FILTER_BANK

    def close(self):
        from . import lib as L
        self.flush()
        L.check(L.load().ppn_set_conv_tile_policy(0), "ppn_set_conv_tile_policy")

    def submit(self, frames_u8: torch.Tensor, to_host: bool = False) -> D.DecodeResult:
        """Queue one batch on the next lane.  `frames_u8`: u8 [B,S,S,3] on the device, or in PINNED host memory -- then
        the H2D copy goes straight into the lane's own input buffer on the lane's stream (it overlaps the other
        lanes' kernels and needs no staging tensor).  `to_host=True` also queues the D2H of the compact result into the
        lane's pinned buffers behind the decode (no host synchronisation): after `result.ready.synchronize()`,
        `result.hosted.unpack()` gives the per-image arrays."""
        k = self.k
        self.k = (k + 1) % len(self.streams)
        st = self.streams[k]
        on_host = not frames_u8.is_cuda
        if on_host:
            if not frames_u8.is_pinned():
                raise ValueError("host frames must be in pinned memory (tensor.pin_memory())")
            b, h, w, _ = frames_u8.shape
        else:
            main = torch.cuda.current_stream(frames_u8.device)
            ready = torch.cuda.Event()
            ready.record(main)                               # the frames are complete on the caller's stream
        with torch.cuda.stream(st):
            if on_host:
                buf = self.model.input_buffer(b, h, w, True, True, slot=k, conv_flags=self._conv_flags)
                buf.copy_(frames_u8, non_blocking=True)
                frames_u8 = buf
            else:
                st.wait_event(ready)
            unary, keys = self.model.forward_u8(frames_u8, fused_decode=True, slot=k, conv_flags=self._conv_flags)
            res = self.decoders[k].decode_fused(unary, keys)
            if to_host:
                if self._stages[k] is None:
                    self._stages[k] = D.HostStage(self.decoders[k].batch, cap=min(64, self.decoders[k].out.max_humans))
                res.hosted = res.to_host_async(self._stages[k])
            ev = torch.cuda.Event()
            ev.record(st)
        res.ready = ev
        return res

    def flush(self):
        for st in self.streams:
            st.synchronize()
