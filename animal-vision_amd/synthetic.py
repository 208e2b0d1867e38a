"""Synthetic frame sources (SURVEY.md 8d "Synthetic inputs").

There is no codec on the box (cv2/ffmpeg absent), so the frame loop, the tests
and bench.py draw frames from here instead of `cv2.VideoCapture`
(reference: renderers/video.py:82-96 returns HxWx3 RGB uint8 frames).
Pure NumPy, no device code.
"""
from __future__ import annotations

import numpy as np


def noise_frame(i: int, H: int, W: int) -> np.ndarray:
    """Frame i of the incompressible stream: exercises the full 256-entry LUT."""
    return np.random.default_rng(1234 + i).integers(0, 256, (H, W, 3), dtype=np.uint8)


def structured_frame(i: int, H: int, W: int) -> np.ndarray:
    """Gradients + 4 flat colour bars + seeded +-8 noise, so that the P95/P98/P99
    order statistics of the UV mappers are non-degenerate."""
    rng = np.random.default_rng(4321 + i)
    y = np.linspace(0.0, 1.0, H, dtype=np.float32)[:, None]
    x = np.linspace(0.0, 1.0, W, dtype=np.float32)[None, :]
    img = np.empty((H, W, 3), np.float32)
    img[..., 0] = 255.0 * x
    img[..., 1] = 255.0 * y
    img[..., 2] = 255.0 * (0.5 + 0.5 * np.sin(6.2831853 * (x + y)))
    bars = np.array([[230, 40, 40], [40, 200, 60], [50, 70, 220], [240, 230, 90]], np.float32)
    bh = max(1, H // 6)
    for k in range(4):
        x0, x1 = (k * W) // 4, ((k + 1) * W) // 4
        img[H - bh :, x0:x1, :] = bars[k]
    img += rng.integers(-8, 9, (H, W, 3)).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


class SyntheticVideoSource:
    """`get_image()`-style source: n_frames frames of HxW, then None (end of stream).

    offset / stride: serve only the global frames offset, offset + stride, ... (< n_frames) -- a rank of a sharded stream
    touches only its own frames (frame i belongs to rank i mod world).  `index` is the global index of the
    frame last returned."""

    def __init__(self, H: int, W: int, n_frames: int, kind: str = "noise", pool: int = 8, offset: int = 0, stride: int = 1):
        if stride < 1 or not (0 <= offset < stride or stride == 1):
            raise ValueError(f"offset {offset} / stride {stride}")
        self.H, self.W, self.n_frames, self.kind = H, W, n_frames, kind
        self.offset, self.stride = int(offset), int(stride)
        gen = noise_frame if kind == "noise" else structured_frame
        # A small pool of distinct frames, cycled (global frame i is pool[i % pool]): generating 4K noise costs more
        # host time than the device pipeline takes to process it.  Only the pool entries this shard can reach are built.
        psize = min(pool, max(1, n_frames))
        reach = sorted({i % psize for i in range(self.offset, n_frames, self.stride)})
        self._psize = psize
        self._pool = {k: gen(k, H, W) for k in reach}
        self._i = self.offset
        self.index = -1

    def get_image(self):
        if self._i >= self.n_frames:
            return None
        f = self._pool[self._i % self._psize]
        self.index = self._i
        self._i += self.stride
        return f

    def reset(self):
        self._i = self.offset
        self.index = -1
