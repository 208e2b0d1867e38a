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
    """`get_image()`-style source: n_frames frames of HxW, then None (end of stream)."""

    def __init__(self, H: int, W: int, n_frames: int, kind: str = "noise", pool: int = 8):
        self.H, self.W, self.n_frames, self.kind = H, W, n_frames, kind
        gen = noise_frame if kind == "noise" else structured_frame
        # A small pool of distinct frames, cycled: generating 4K noise costs more
        # host time than the device pipeline takes to process it.
        self._pool = [gen(i, H, W) for i in range(min(pool, max(1, n_frames)))]
        self._i = 0

    def get_image(self):
        if self._i >= self.n_frames:
            return None
        f = self._pool[self._i % len(self._pool)]
        self._i += 1
        return f

    def reset(self):
        self._i = 0
