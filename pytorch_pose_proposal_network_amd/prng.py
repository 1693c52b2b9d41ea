"""Deterministic integer PRNG for synthetic weights / frames / heads.

Everything is splitmix64 in uint64 wrap-around arithmetic followed by exact
integer->float conversions, so the container that generates the golden fixtures and
the GPU box that replays them produce bit-identical tensors without shipping 128 MB of
weights.  No transcendental functions are used: the
"normal-ish" stream is a centred sum of four uniforms scaled to unit variance.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def stream_seed(base_seed: int, index: int) -> int:
    """Independent stream id for tensor number `index` of experiment `base_seed`."""
    with np.errstate(over="ignore"):
        z = np.array([np.uint64(base_seed & 0xFFFFFFFFFFFFFFFF)], dtype=np.uint64) * _GOLDEN
        z = _mix(z + np.uint64(index) * np.uint64(0xD1B54A32D192ED03) + np.uint64(1))
    return int(z[0])


def raw_u64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed`."""
    with np.errstate(over="ignore"):
        ctr = np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        return _mix(ctr)


def uniform01(seed: int, n: int) -> np.ndarray:
    """float32 uniform in [0,1) with 24 random bits (exact conversion)."""
    x = raw_u64(seed, n) >> np.uint64(40)
    return (x.astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def uniform(seed: int, n: int, lo: float, hi: float) -> np.ndarray:
    u = uniform01(seed, n).astype(np.float64)
    return (lo + (hi - lo) * u).astype(np.float32)


def normalish(seed: int, n: int) -> np.ndarray:
    """Unit-variance, zero-mean float32 (Irwin-Hall of 4 uniforms): no libm involved."""
    r = raw_u64(seed, 2 * n)
    a = (r[0::2] >> np.uint64(40)).astype(np.float64)
    b = ((r[0::2] >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)
    c = (r[1::2] >> np.uint64(40)).astype(np.float64)
    d = ((r[1::2] >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)
    s = (a + b + c + d) * (1.0 / 16777216.0) - 2.0
    return (s * 1.7320508075688772).astype(np.float32)


def u8_frames(seed: int, batch: int, size_hw=(384, 384)) -> np.ndarray:
    """Synthetic RGB frames u8[B,H,W,3], uniform {0..255}."""
    h, w = size_hw
    n = batch * h * w * 3
    r = raw_u64(stream_seed(seed, 0), (n + 7) // 8)
    return r.view(np.uint8)[:n].reshape(batch, h, w, 3).copy()
