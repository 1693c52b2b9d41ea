"""ORACLE (test infrastructure, never shipped or measured as the product).

NumPy restatement of the reference's frame ingest, rt_test.py:150-157 (`grab_frame`):

    frame = cv2.resize(frame, dsize=(384, 384))      # INTER_LINEAR, 8-bit
    frame = cv2.flip(frame, 0); frame = cv2.flip(frame, 1)
    return cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)

Parity pin: **PARITY UNPINNED**.  The arithmetic lives in OpenCV, a third-party dependency that is neither under
/root/reference nor installed in this image, and the reference pins no version (README.md names none; there is no
requirements file).  What follows restates OpenCV's published 8-bit bilinear resize (modules/imgproc/src/resize.cpp,
3.x/4.x: `resizeGeneric_` with `HResizeLinear<uchar,int,short,2048>` and
`VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>`, whose scalar tail is the rule below and whose SIMD
bodies are written to give the same bytes):

    scale_x = Ws / Wd (double);  fx = float((dx + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx
    sx < 0 -> (sx, fx) = (0, 0);   sx >= Ws - 1 -> (sx, fx) = (Ws - 1, 0)          (same for y)
    a0 = saturate_short(round_half_even((1 - fx) * 2048)),  a1 = saturate_short(round_half_even(fx * 2048))
    row value   t = S[sx] * a0 + S[sx + 1] * a1                          (int, scale 2^11)
    pixel       d = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2
    exact halving in both directions (Ws = 2 Wd and Hs = 2 Hd): INTER_LINEAR is replaced by the INTER_AREA fast path,
    d = (s00 + s01 + s10 + s11 + 2) >> 2.

There is no reference fixture for this row (the reference has no tests and cv2 cannot run here); the CPU tests hold
the restatement to properties only (identity size = pure flip + channel swap; within 1 LSB of real-valued bilinear
interpolation).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np


def _axis_tables(n_src: int, n_dst: int):
    """(ofs int[n_dst], a0 int[n_dst], a1 int[n_dst]) of one axis."""
    scale = np.float64(n_src) / np.float64(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0.0
    a0 = np.clip(np.rint((np.float32(1.0) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
    a1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(src: np.ndarray, size_hw) -> np.ndarray:
    """cv2.resize(src, dsize=(W, H)) for u8 [Hs,Ws,C], INTER_LINEAR (see module docstring)."""
    Hs, Ws, _ = src.shape
    Hd, Wd = size_hw
    s = src.astype(np.int64)
    if Hs == 2 * Hd and Ws == 2 * Wd:
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xs, xa0, xa1 = _axis_tables(Ws, Wd)
    ys, yb0, yb1 = _axis_tables(Hs, Hd)
    x1 = np.minimum(xs + 1, Ws - 1)
    y1 = np.minimum(ys + 1, Hs - 1)
    rows = s[:, xs] * xa0[None, :, None] + s[:, x1] * xa1[None, :, None]          # [Hs, Wd, C], scale 2^11
    t0, t1 = rows[ys], rows[y1]
    d = (((yb0[:, None, None] * (t0 >> 4)) >> 16) + ((yb1[:, None, None] * (t1 >> 4)) >> 16) + 2) >> 2
    return np.clip(d, 0, 255).astype(np.uint8)


def grab_frame_ref(frame_bgr: np.ndarray, size: int = 384) -> np.ndarray:
    """rt_test.py:150-157: BGR u8 [Hs,Ws,3] camera frame -> RGB u8 [size,size,3], resized, rotated by 180 degrees."""
    r = resize_linear_u8(frame_bgr, (size, size))
    return np.ascontiguousarray(r[::-1, ::-1, ::-1])
