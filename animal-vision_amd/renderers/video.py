"""VideoRenderer -- the frame source/sink of the reference's renderers/video.py with its keyword surface
(`read_path, write_path, fps, window_name`, :31-38) and frame contract (RGB uint8 HxWx3 from get_image(),
None at end of stream, :82-96).

Codec I/O is out of scope (the reference leans on cv2.VideoCapture / VideoWriter 'mp4v', :68,:113; neither
cv2 nor ffmpeg exists here), so paths name what this box can read and write:
  read_path : "synthetic:<W>x<H>:<n>[:noise|structured]"  |  a .npy file (N,H,W,3 uint8)  |  a directory of images
  write_path: a directory (one PNG per frame, via Pillow)  |  a .npy file  |  None
No preview window (no GUI); `window_name` is accepted and ignored.

Sharded streams (two keywords the reference does not have, both defaulted): with `world` > 1 this renderer
serves and stores only the frames of rank `rank` (global frame i belongs to rank i mod world) -- every source here is
index-addressable, so the other ranks' frames are never generated, read or decoded.  Outputs carry the GLOBAL frame
index: PNG sinks are named frame_<global index>.png (ranks may share the directory); a .npy sink is written as one
memory-mapped shard per rank (streamed to disk frame by frame, never held in RAM) and rank 0's merge_shards() -- called
by pipeline.run_video after the ranks' closing collective -- interleaves them into the one ordered stream."""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from ..synthetic import SyntheticVideoSource
from .renderer import Renderer


def _resize_area_u8(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv2.resize(img, (W, H), interpolation=cv2.INTER_AREA) of a uint8 frame on the device (csrc/geom.hip)."""
    from ..geometry import INTER_AREA, resize

    return resize(np.ascontiguousarray(img), (W, H), INTER_AREA)


def split_compose(original: np.ndarray, modified: np.ndarray, draw_seam: bool = True, *, left_label: Optional[str] = None,
                  right_label: Optional[str] = None) -> np.ndarray:
    """renderers/video.py:225-245: `modified` resized to `original`'s size if needed (INTER_AREA, on the device), left half
    original, right half modified, 1-px white seam at W//2, then the two corner labels (None = no label)."""
    assert isinstance(original, np.ndarray) and original.ndim == 3 and original.shape[2] == 3, "original must be HxWx3 RGB"
    assert isinstance(modified, np.ndarray) and modified.ndim == 3 and modified.shape[2] == 3, "modified must be HxWx3 RGB"
    H, W, _ = original.shape
    if modified.shape[:2] != (H, W):
        if original.dtype != np.uint8 or modified.dtype != np.uint8:
            raise TypeError("split compare of frames of different sizes is implemented for uint8 frames (what get_image() yields)")
        modified = _resize_area_u8(modified, W, H)
    out = original.copy()
    mid = W // 2
    out[:, mid:, :] = modified[:, mid:, :]
    if draw_seam:
        out[:, mid : mid + 1, :] = 255
    if left_label is not None or right_label is not None:
        from .labels import draw_split_labels

        out = draw_split_labels(out, left_label, right_label)
    return out


class VideoRenderer(Renderer):
    def __init__(self, *, read_path: Optional[str] = None, write_path: Optional[str] = None, fps: Optional[int] = None,
                 window_name: str = "Video Analysis", rank: int = 0, world: int = 1):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.read_path, self.write_path = read_path, write_path
        self.fps = fps or 30
        self.window_name = window_name
        self.rank, self.world = int(rank), int(world)
        self._src = None
        self._frames: Optional[np.ndarray] = None
        self._files: List[str] = []
        self._i = self.rank          # global index of the next frame this rank reads
        self.last_index = -1         # global index of the frame get_image() returned last
        self.total_frames: Optional[int] = None
        self._sink = None            # memory-mapped .npy shard (streamed)
        self._sink_rows = 0
        self.frames_written = 0

    # ---- source ----------------------------------------------------------------------------------------
    def open(self) -> None:
        p = self.read_path
        if p:
            if p.startswith("synthetic:"):
                parts = p.split(":")
                w, h = (int(v) for v in parts[1].lower().split("x"))
                kind = parts[3] if len(parts) > 3 else "noise"
                self.total_frames = int(parts[2])
                self._src = SyntheticVideoSource(h, w, self.total_frames, kind, offset=self.rank, stride=self.world)
            elif p.endswith(".npy"):
                self._frames = np.load(p, mmap_mode="r")
                self.total_frames = len(self._frames)
            elif os.path.isdir(p):
                self._files = sorted(os.path.join(p, f) for f in os.listdir(p) if f.lower().endswith((".png", ".jpg", ".jpeg")))
                self.total_frames = len(self._files)
            else:
                raise RuntimeError(f"Failed to open video for reading: {p} (no codec on this box: synthetic:, .npy or an image directory)")
        if self.write_path and not self.write_path.endswith(".npy"):
            os.makedirs(self.write_path, exist_ok=True)

    def get_image(self) -> Optional[np.ndarray]:
        if self._src is not None:
            f = self._src.get_image()
            if f is not None:
                self.last_index = self._src.index
            return f
        n = len(self._frames) if self._frames is not None else len(self._files)
        if self._i >= n:
            return None
        if self._frames is not None:
            f = np.ascontiguousarray(self._frames[self._i])
        else:
            from PIL import Image

            f = np.asarray(Image.open(self._files[self._i]).convert("RGB"))
        self.last_index = self._i
        self._i += self.world
        return f

    # ---- sink ------------------------------------------------------------------------------------------
    def _shard_path(self, rank: int) -> str:
        return self.write_path if self.world == 1 else f"{self.write_path[:-4]}.rank{rank}of{self.world}.npy"

    def _own_count(self) -> Optional[int]:
        return None if self.total_frames is None else len(range(self.rank, self.total_frames, self.world))

    def render(self, frame: np.ndarray, *, index: Optional[int] = None) -> None:
        """renderers/video.py:118-142 (write the frame).  `index` = the frame's GLOBAL stream index; by default frames are
        taken to arrive in this rank's stream order (rank, rank + world, ...)."""
        if index is None:
            index = self.rank + self.frames_written * self.world
        if self.write_path:
            if self.write_path.endswith(".npy"):
                row = (index - self.rank) // self.world
                if (index - self.rank) % self.world or row < 0:
                    raise ValueError(f"frame {index} does not belong to rank {self.rank} of {self.world}")
                if self._sink is None or row >= self._sink_rows:
                    self._grow_sink(frame, row)
                self._sink[row] = frame
            else:
                from PIL import Image

                Image.fromarray(frame).save(os.path.join(self.write_path, f"frame_{index:06d}.png"))
        self.frames_written += 1

    def _grow_sink(self, frame: np.ndarray, row: int) -> None:
        """Create the shard on first use (sized from the source's frame count when it is known); a stream of unknown length
        doubles the mapping (old rows are copied once per doubling, on disk)."""
        want = self._own_count() or 0
        rows = max(want, row + 1, 2 * self._sink_rows, 1)
        path = self._shard_path(self.rank)
        new = np.lib.format.open_memmap(path + ".tmp" if self._sink is not None else path, mode="w+", dtype=frame.dtype, shape=(rows,) + frame.shape)
        if self._sink is not None:
            new[: self._sink_rows] = self._sink[: self._sink_rows]
            new.flush()
            del self._sink
            os.replace(path + ".tmp", path)
            new = np.load(path, mmap_mode="r+")
        self._sink, self._sink_rows = new, rows

    def flush(self) -> None:
        """Everything rendered so far is on disk as a well-formed .npy (a mapping sized for more frames than arrived is
        trimmed to what was written); rendering may continue afterwards."""
        if self._sink is None:
            return
        self._sink.flush()
        written, rows, path = self.frames_written, self._sink_rows, self._shard_path(self.rank)
        if written < rows:
            shape, dtype = self._sink.shape[1:], self._sink.dtype
            tmp = np.lib.format.open_memmap(path + ".tmp", mode="w+", dtype=dtype, shape=(written,) + shape)
            tmp[:] = self._sink[:written]
            tmp.flush()
            del tmp
            self._sink = None
            os.replace(path + ".tmp", path)
            self._sink, self._sink_rows = np.load(path, mmap_mode="r+"), written

    def close(self) -> None:
        self.flush()
        self._sink = None
        self._src = self._frames = None

    def merge_shards(self) -> Optional[str]:
        """Rank 0, after every rank has flushed/closed its shard (run_video calls it behind the closing collective): interleave
        <write_path>.rank<r>of<world>.npy into the one ordered stream <write_path>, frame i from shard i mod world."""
        if not (self.write_path and self.write_path.endswith(".npy")) or self.world == 1:
            return self.write_path
        shards = [np.load(self._shard_path(r), mmap_mode="r") if os.path.exists(self._shard_path(r)) else None for r in range(self.world)]
        have = [s for s in shards if s is not None]
        if not have:
            return None
        n = sum(len(s) for s in have)
        out = np.lib.format.open_memmap(self.write_path, mode="w+", dtype=have[0].dtype, shape=(n,) + have[0].shape[1:])
        for r, s in enumerate(shards):
            if s is not None:
                idx = np.arange(r, r + len(s) * self.world, self.world)
                if len(idx) and idx[-1] >= n:
                    raise ValueError("shards are not a round-robin partition of one stream")
                for k, i in enumerate(idx):  # frame by frame: bounded memory
                    out[i] = s[k]
        out.flush()
        del out, shards, have
        for r in range(self.world):
            if os.path.exists(self._shard_path(r)):
                os.remove(self._shard_path(r))
        return self.write_path

    # ---- split compare ---------------------------------------------------------------------------------
    def make_split_frame(self, original: np.ndarray, modified: np.ndarray, *, left_label: str = "Original",
                         right_label: str = "Transformed", draw_seam: bool = True) -> np.ndarray:
        return split_compose(original, modified, draw_seam, left_label=left_label, right_label=right_label)

    def render_split_compare(self, original: np.ndarray, modified: np.ndarray, *, left_label: str = "Original",
                             right_label: str = "Transformed", draw_seam: bool = True) -> None:
        self.render(self.make_split_frame(original, modified, left_label=left_label, right_label=right_label, draw_seam=draw_seam))
