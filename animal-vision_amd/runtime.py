"""Device runtime on top of the C ABI: one Context per process/GPU, device buffers, streams.

The reference has no device runtime (pure NumPy); this is the host plumbing that
lets `Animal.visualize(np.ndarray)` (animals/animal.py:5-7) keep its signature while the work runs
on the MI355X, and lets the frame loop keep frames resident in HBM between stages."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib
from ._lib import AvxError, lib


class DeviceBuffer:
    """A caller-owned allocation in HBM (freed on .free() or garbage collection)."""

    __slots__ = ("ctx", "ptr", "nbytes", "_owned")

    def __init__(self, ctx: "Context", ptr: int, nbytes: int, owned: bool = True):
        self.ctx, self.ptr, self.nbytes, self._owned = ctx, ptr, nbytes, owned

    def free(self):
        if self._owned and self.ptr and self.ctx._h:
            lib.avx_free(self.ctx._h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def view(self, offset: int, nbytes: int) -> "DeviceBuffer":
        assert 0 <= offset and offset + nbytes <= self.nbytes
        return DeviceBuffer(self.ctx, self.ptr + offset, nbytes, owned=False)


class PinnedArray:
    """Page-locked host array (H2D/D2H at full PCIe rate, truly asynchronous copies)."""

    def __init__(self, ctx: "Context", shape, dtype):
        self.ctx = ctx
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        ctx._check(lib.avx_host_alloc(ctx._h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value
        buf = (ctypes.c_ubyte * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def free(self):
        if self.ptr and self.ctx._h:
            self.array = None
            lib.avx_host_free(self.ctx._h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """avx_ctx wrapper: bound to one device, single-threaded (include/avx.h conventions)."""

    def __init__(self, device: int = 0):
        h = ctypes.c_void_p()
        rc = lib.avx_init(int(device), ctypes.byref(h))
        if rc != _lib.AVX_OK:
            raise AvxError(rc, lib.avx_last_error(None).decode())
        self._h = h.value
        self.device = int(device)
        # The context's own stream: what every op uses when the caller names none.  (In the C ABI NULL is
        # HIP's null stream, so that a torch default-stream handle of 0 means what torch means by it.)
        self.stream = self.stream_create()

    def _s(self, stream):
        return self.stream if stream is None else stream

    # -- errors --------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != _lib.AVX_OK:
            raise AvxError(rc, lib.avx_last_error(self._h).decode())

    def close(self):
        if self._h:
            if getattr(self, "stream", None):
                lib.avx_stream_destroy(self._h, self.stream)
                self.stream = None
            lib.avx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory --------------------------------------------------------------------------------
    def malloc(self, nbytes: int) -> DeviceBuffer:
        p = ctypes.c_void_p()
        self._check(lib.avx_malloc(self._h, int(nbytes), ctypes.byref(p)))
        return DeviceBuffer(self, p.value, int(nbytes))

    def pinned(self, shape, dtype) -> PinnedArray:
        return PinnedArray(self, shape, dtype)

    def upload(self, arr: np.ndarray, dst: Optional[DeviceBuffer] = None, stream=None) -> DeviceBuffer:
        a = np.ascontiguousarray(arr)
        if dst is None:
            dst = self.malloc(a.nbytes)
        assert dst.nbytes >= a.nbytes
        self._check(lib.avx_memcpy_h2d(self._h, dst.ptr, a.ctypes.data, a.nbytes, self._s(stream)))
        self.sync(stream)  # pageable source: do not let the caller free it under the copy
        return dst

    def download(self, src: DeviceBuffer, shape, dtype, out: Optional[np.ndarray] = None, stream=None, sync=True) -> np.ndarray:
        dtype = np.dtype(dtype)
        if out is None:
            out = np.empty(shape, dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        assert out.nbytes >= nbytes and out.flags["C_CONTIGUOUS"] and src.nbytes >= nbytes
        self._check(lib.avx_memcpy_d2h(self._h, out.ctypes.data, src.ptr, nbytes, self._s(stream)))
        if sync:
            self.sync(stream)
        return out

    def memset(self, buf: DeviceBuffer, value: int, nbytes: Optional[int] = None, stream=None):
        self._check(lib.avx_memset(self._h, buf.ptr, value, buf.nbytes if nbytes is None else nbytes, self._s(stream)))

    # -- streams / timing ----------------------------------------------------------------------
    def stream_create(self) -> int:
        s = ctypes.c_void_p()
        self._check(lib.avx_stream_create(self._h, ctypes.byref(s)))
        return s.value

    def stream_destroy(self, s):
        self._check(lib.avx_stream_destroy(self._h, s))

    def sync(self, stream=None):
        self._check(lib.avx_sync(self._h, self._s(stream)))

    def stream_wait(self, waiter, signaler):
        """`waiter` waits (on the device) for everything enqueued on `signaler` so far."""
        self._check(lib.avx_stream_wait(self._h, self._s(waiter), self._s(signaler)))

    def device_sync(self):
        self._check(lib.avx_device_sync(self._h))

    def timer_start(self, stream=None):
        self._check(lib.avx_timer_start(self._h, self._s(stream)))

    def timer_stop(self, stream=None) -> float:
        ms = ctypes.c_float()
        self._check(lib.avx_timer_stop(self._h, self._s(stream), ctypes.byref(ms)))
        return float(ms.value)


_default: Optional[Context] = None


def device_count() -> int:
    return int(lib.avx_device_count())


def get_context(device: Optional[int] = None) -> Context:
    """Process-wide default context (one process per GPU; LOCAL_RANK picks the device)."""
    global _default
    if _default is None:
        import os

        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, device_count())
        _default = Context(device)
    return _default


def get_table(which: int) -> np.ndarray:
    """The constant tables compiled into the library (0: decode LUT, 1: f32 thresholds, 2: f64 thresholds)."""
    size = lib.avx_get_table(which, None, 0)
    dt = np.float64 if which == 2 else np.float32
    out = np.empty(size // np.dtype(dt).itemsize, dt)
    lib.avx_get_table(which, out.ctypes.data, out.nbytes)
    return out
