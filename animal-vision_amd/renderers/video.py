"""VideoRenderer -- the frame source/sink of the reference's renderers/video.py with its keyword surface
(`read_path, write_path, fps, window_name`, :31-38) and frame contract (RGB uint8 HxWx3 from get_image(),
None at end of stream, :82-96).

Codec I/O is out of scope (the reference leans on cv2.VideoCapture / VideoWriter 'mp4v', :68,:113; neither
cv2 nor ffmpeg exists here), so paths name what this box can read and write:
  read_path : "synthetic:<W>x<H>:<n>[:noise|structured]"  |  a .npy file (N,H,W,3 uint8)  |  a directory of images
  write_path: a directory (one PNG per frame, via Pillow)  |  a .npy file  |  None
No preview window (no GUI); `window_name` is accepted and ignored."""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from ..synthetic import SyntheticVideoSource
from .renderer import Renderer


def split_compose(original: np.ndarray, modified: np.ndarray, draw_seam: bool = True) -> np.ndarray:
    """Left half original, right half modified, 1-px white seam at W//2 (renderers/video.py:234-239).
    The Hershey-font corner labels (:242-244) are cv2 drawing code: not reproduced."""
    assert isinstance(original, np.ndarray) and original.ndim == 3 and original.shape[2] == 3, "original must be HxWx3 RGB"
    assert isinstance(modified, np.ndarray) and modified.ndim == 3 and modified.shape[2] == 3, "modified must be HxWx3 RGB"
    if modified.shape[:2] != original.shape[:2]:
        raise NotImplementedError("split compare of different sizes needs the INTER_AREA resize kernel (SURVEY 8f row 1)")
    out = original.copy()
    mid = original.shape[1] // 2
    out[:, mid:, :] = modified[:, mid:, :]
    if draw_seam:
        out[:, mid : mid + 1, :] = 255
    return out


class VideoRenderer(Renderer):
    def __init__(self, *, read_path: Optional[str] = None, write_path: Optional[str] = None, fps: Optional[int] = None,
                 window_name: str = "Video Analysis"):
        self.read_path, self.write_path = read_path, write_path
        self.fps = fps or 30
        self.window_name = window_name
        self._src = None
        self._frames: Optional[np.ndarray] = None
        self._files: List[str] = []
        self._i = 0
        self._written: List[np.ndarray] = []
        self.frames_written = 0

    def open(self) -> None:
        p = self.read_path
        if p:
            if p.startswith("synthetic:"):
                parts = p.split(":")
                w, h = (int(v) for v in parts[1].lower().split("x"))
                kind = parts[3] if len(parts) > 3 else "noise"
                self._src = SyntheticVideoSource(h, w, int(parts[2]), kind)
            elif p.endswith(".npy"):
                self._frames = np.load(p, mmap_mode="r")
            elif os.path.isdir(p):
                self._files = sorted(os.path.join(p, f) for f in os.listdir(p) if f.lower().endswith((".png", ".jpg", ".jpeg")))
            else:
                raise RuntimeError(f"Failed to open video for reading: {p} (no codec on this box: synthetic:, .npy or an image directory)")
        if self.write_path and not self.write_path.endswith(".npy"):
            os.makedirs(self.write_path, exist_ok=True)

    def get_image(self) -> Optional[np.ndarray]:
        if self._src is not None:
            return self._src.get_image()
        if self._frames is not None:
            if self._i >= len(self._frames):
                return None
            f = np.ascontiguousarray(self._frames[self._i])
            self._i += 1
            return f
        if self._files:
            if self._i >= len(self._files):
                return None
            from PIL import Image

            f = np.asarray(Image.open(self._files[self._i]).convert("RGB"))
            self._i += 1
            return f
        return None

    def render(self, frame: np.ndarray) -> None:
        if self.write_path:
            if self.write_path.endswith(".npy"):
                self._written.append(np.ascontiguousarray(frame))
            else:
                from PIL import Image

                Image.fromarray(frame).save(os.path.join(self.write_path, f"frame_{self.frames_written:06d}.png"))
        self.frames_written += 1

    def close(self) -> None:
        if self.write_path and self.write_path.endswith(".npy") and self._written:
            np.save(self.write_path, np.stack(self._written))
            self._written = []
        self._src = self._frames = None

    def make_split_frame(self, original: np.ndarray, modified: np.ndarray, *, left_label: str = "Original",
                         right_label: str = "Transformed", draw_seam: bool = True) -> np.ndarray:
        return split_compose(original, modified, draw_seam)

    def render_split_compare(self, original: np.ndarray, modified: np.ndarray, *, left_label: str = "Original",
                             right_label: str = "Transformed", draw_seam: bool = True) -> None:
        self.render(self.make_split_frame(original, modified, left_label=left_label, right_label=right_label, draw_seam=draw_seam))
