"""The frame loop (reference: main.py:53-72 `video`, one frame in flight, synchronous) as a pipelined,
sharded device loop.

  * Sharding: frames are independent, so frame i belongs to rank i mod world (round-robin,
    BASELINE.json config 4).  One process per GPU; NO collective on the data path.  torch.distributed
    (RCCL over xGMI on the GPU box, gloo in CPU tests) carries only the start/stop barriers and the
    reduction of per-rank statistics.
  * Pipelining inside a rank: `depth` slots, each with its own HIP stream, pinned host buffers and HBM
    buffers; a frame's H2D copy, kernels and D2H copy are enqueued in order on its slot's stream, so the
    copies of frames i+1 / i-1 overlap the kernels of frame i (different streams), and the host only
    blocks on a slot when it comes around again."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np


# ---------------------------------------------------------------- sharding (pure host logic) --------
def shard_indices(n_frames: int, rank: int, world: int) -> List[int]:
    """Global frame indices owned by `rank`: i mod world == rank."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_frames, world))


def owner_of(frame_index: int, world: int) -> int:
    return frame_index % world


def merge_in_order(per_rank: List[List[Tuple[int, object]]]) -> List[object]:
    """Re-interleave per-rank (index, item) lists into stream order; checks every index appears once."""
    flat = sorted((i, item) for lst in per_rank for i, item in lst)
    idx = [i for i, _ in flat]
    if idx != list(range(len(idx))):
        raise ValueError("frame indices are not a permutation of 0..n-1")
    return [item for _, item in flat]


@dataclass
class StreamStats:
    frames: int = 0
    pixels: int = 0
    seconds: float = 0.0
    ranks: int = 1
    host_copy_seconds: float = 0.0  # this rank's (after reduce_stats: the slowest rank's) time in pageable <-> pinned frame copies

    @property
    def megapixels_per_second(self) -> float:
        return self.pixels / 1e6 / self.seconds if self.seconds > 0 else 0.0


def reduce_stats(local: StreamStats, dist=None) -> StreamStats:
    """Whole-job statistics: frames and pixels summed over ranks, wall time = max over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    s = torch.tensor([float(local.frames), float(local.pixels)], dtype=torch.float64, device=dev)
    t = torch.tensor([local.seconds, local.host_copy_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return StreamStats(int(s[0].item()), int(s[1].item()), float(t[0].item()), dist.get_world_size(), float(t[1].item()))


# ---------------------------------------------------------------- device pipeline -------------------
# Host copies of a frame (pageable <-> pinned, 25 MB at 4K) run on a few threads: one core moves ~10 GB/s and faults the fresh
# output array's pages in one by one, which made the host side of a dichromat stream (2 ms per 4K frame) the limiter, not the GPU.
_COPY_THREADS = 4
_copy_pool = None


def _pcopy(dst: np.ndarray, src: np.ndarray) -> None:
    """dst[...] = src, split by rows over a small thread pool for large frames (NumPy releases the GIL while copying)."""
    global _copy_pool
    if src.nbytes < (4 << 20) or src.shape[0] < _COPY_THREADS:
        dst[...] = src
        return
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="avx-copy")
    rows = src.shape[0]
    step = (rows + _COPY_THREADS - 1) // _COPY_THREADS

    def part(r0):
        dst[r0:r0 + step] = src[r0:r0 + step]

    list(_copy_pool.map(part, range(0, rows, step)))


@dataclass
class _Slot:
    stream: int
    h_in: object
    h_out: object
    d_in: object
    d_out: object
    index: int = -1
    busy: bool = False


class FramePipeline:
    """Runs `op.run_device(d_in, d_out, 1, H, W, stream=...)` (DichromatOp / HoneybeeOp) over a stream of
    uint8 frames with `depth` frames in flight."""

    def __init__(self, op, H: int, W: int, *, ctx=None, depth: int = 3, split_compare: bool = False, draw_seam: bool = True,
                 labels: Optional[Tuple[Optional[str], Optional[str]]] = ("Original", "Transformed")):
        """split_compare: emit make_split_frame(original, transformed) composed on the device (renderers/video.py:198-245:
        halves, seam, and the two corner labels -- `labels` = (left, right), None = none) instead of the transformed frame."""
        from .runtime import get_context

        self.op, self.H, self.W, self.depth = op, H, W, depth
        self.split_compare, self.draw_seam = bool(split_compare), bool(draw_seam)
        self.labels = tuple(labels) if labels else (None, None)
        self.ctx = ctx or getattr(op, "ctx", None) or get_context()
        if getattr(op, "ctx", None) is None:
            op.ctx = self.ctx
        nbytes = H * W * 3
        # ops that own their device frames (recorded species plans, animals/_uv_species.py::SpeciesStreamOp) lend them per slot
        lend = getattr(op, "slot_buffers", None)
        self._lent = lend is not None
        self.slots = []
        for k in range(depth):
            d_in, d_out = lend(k) if lend else (self.ctx.malloc(nbytes), self.ctx.malloc(nbytes))
            self.slots.append(_Slot(self.ctx.stream_create(), self.ctx.pinned((H, W, 3), np.uint8), self.ctx.pinned((H, W, 3), np.uint8), d_in, d_out))

    def close(self):
        for s in self.slots:
            self.ctx.sync(s.stream)
        release = getattr(self.op, "release_streams", None)  # ops that wrap the slot streams (ml/predict.py::MstHoneybeeStreamOp) drop the wrappers first
        if release is not None:
            release()
        for s in self.slots:
            self.ctx.stream_destroy(s.stream)
            s.h_in.free(); s.h_out.free()
            if not self._lent:
                s.d_in.free(); s.d_out.free()
        self.slots = []

    def _retire(self, s: _Slot, emit: Callable[[int, np.ndarray], None]):
        if s.busy:
            self.ctx.sync(s.stream)
            t0 = time.perf_counter()
            out = np.empty_like(s.h_out.array)
            _pcopy(out, s.h_out.array)
            self._copy_s += time.perf_counter() - t0
            emit(s.index, out)
            s.busy = False

    def run(self, frames: Iterator[Tuple[int, np.ndarray]], emit: Callable[[int, np.ndarray], None]) -> StreamStats:
        """frames: (global index, HxWx3 uint8) pairs owned by this rank; emit(index, out) in submission order."""
        from ._lib import lib

        ctx, n, t0 = self.ctx, 0, time.perf_counter()
        self._copy_s = 0.0
        nbytes = self.H * self.W * 3
        for k, (index, frame) in enumerate(frames):
            s = self.slots[k % self.depth]
            self._retire(s, emit)
            if frame.shape != (self.H, self.W, 3) or frame.dtype != np.uint8:
                raise ValueError(f"frame {index}: expected uint8 {(self.H, self.W, 3)}, got {frame.dtype} {frame.shape}")
            tc = time.perf_counter()
            _pcopy(s.h_in.array, frame)
            self._copy_s += time.perf_counter() - tc
            ctx._check(lib.avx_memcpy_h2d(ctx._h, s.d_in.ptr, s.h_in.ptr, nbytes, s.stream))
            self.op.run_device(s.d_in, s.d_out, 1, self.H, self.W, stream=s.stream)
            if self.split_compare:
                ctx._check(lib.avx_split_compose_u8(ctx._h, s.d_in.ptr, s.d_out.ptr, s.d_out.ptr, self.H, self.W, int(self.draw_seam), s.stream))
                if self.labels[0] is not None or self.labels[1] is not None:
                    from .renderers.labels import draw_split_labels_device

                    draw_split_labels_device(ctx, s.d_out, self.H, self.W, self.labels[0], self.labels[1], s.stream)
            ctx._check(lib.avx_memcpy_d2h(ctx._h, s.h_out.ptr, s.d_out.ptr, nbytes, s.stream))
            s.index, s.busy = index, True
            n += 1
        for j in range(self.depth):  # drain in submission order
            self._retire(self.slots[(n + j) % self.depth], emit)
        return StreamStats(n, n * self.H * self.W, time.perf_counter() - t0, 1, self._copy_s)


def run_video(animal_op, renderer, *, rank: int = 0, world: int = 1, depth: int = 3, split_compare: bool = False, dist=None,
              labels: Optional[Tuple[Optional[str], Optional[str]]] = ("Original", "Transformed")) -> StreamStats:
    """main.py:53-72 on the device: read -> visualize -> (split-compose + labels) -> render, this rank's shard only.

    A renderer that shards itself (renderers.VideoRenderer(rank=, world=): strided source, index-addressed sink) hands over
    only this rank's frames and its `last_index` names each one's place in the stream; any other get_image()/render() pair
    is read in full and filtered here (frame i belongs to rank i mod world).  Outputs go to the sink under their GLOBAL frame
    index; after the ranks' closing collective (the statistics reduction) rank 0 reassembles a sharded .npy sink into the one
    ordered stream ("host re-orders outputs by frame index before render()")."""
    self_sharding = getattr(renderer, "world", 1) == world and getattr(renderer, "rank", 0) == rank and hasattr(renderer, "last_index") and world > 1
    first = renderer.get_image()
    if first is None:
        stats = StreamStats()
        pipe = None
    else:
        H, W, _ = first.shape
        pipe = FramePipeline(animal_op, H, W, depth=depth, split_compare=split_compare, labels=labels)

    def frames():
        i, f = 0, first
        while f is not None:
            if self_sharding:
                yield renderer.last_index, f
            elif owner_of(i, world) == rank:
                yield i, f
            i += 1
            f = renderer.get_image()

    def emit(i, out):
        if self_sharding:
            renderer.render(out, index=i)  # already split-composed on the device when split_compare
        else:
            renderer.render(out)

    if pipe is not None:
        try:
            stats = pipe.run(frames(), emit)
        finally:
            pipe.close()
    if hasattr(renderer, "flush"):
        renderer.flush()
    total = reduce_stats(stats, dist)  # a collective: every rank's shard is flushed once it returns anywhere
    if world > 1 and rank == 0 and dist is not None and hasattr(renderer, "merge_shards"):
        renderer.close()
        renderer.merge_shards()
    return total
