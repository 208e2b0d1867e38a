"""Common machinery of the UV species written against the plane-program backend vocabulary (planevm.py).

A species implements `render(be, image)` once; `visualize` records it for the frame size on a DeviceBackend the
first time it sees that size and replays the recorded device calls afterwards (species are stateless between
frames, SURVEY 8b: only constructor constants and O(bands)/O(H)/O(W) tables enter the recording)."""
from typing import Dict, Optional, Tuple

import numpy as np

from .animal import Animal


class UVSpecies(Animal):
    """Base class: same `visualize(image) -> (baseline, out)` contract as the reference's species."""

    _MAX_PLANS = 4

    def render(self, be, image: np.ndarray) -> None:  # pragma: no cover - abstract
        """Species with a frame-dependent branch take a third argument: the value `variant()` returned."""
        raise NotImplementedError

    def variant(self, image: np.ndarray, probes, **kw):
        """A hashable that selects between recorded plans when the reference branches on frame content
        (rat_uv.py:100-105); `probes` supplies the frame statistics (device: planevm.DeviceProbes)."""
        return None

    def _plan(self, image: np.ndarray, variant=None):
        from ..planevm import DeviceBackend

        plans: Dict[tuple, object] = self.__dict__.setdefault("_plans", {})
        floats = image.dtype != np.uint8
        key = (image.shape[0], image.shape[1], variant, floats)
        be = plans.get(key)
        if be is None:
            if len(plans) >= self._MAX_PLANS:
                plans.pop(next(iter(plans))).close()
            be = DeviceBackend(image.shape[0], image.shape[1], float_frames=floats)
            if variant is None:
                self.render(be, image)
            else:
                self.render(be, image, variant)
            be.flush()
            plans[key] = be
        return be

    def visualize(self, image: np.ndarray, **kw) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        assert isinstance(image, np.ndarray), "Input must be a numpy ndarray."
        assert image.ndim == 3 and image.shape[2] == 3, "Input must be HxWx3 RGB."
        if image.dtype != np.uint8 and not np.issubdtype(image.dtype, np.floating):
            raise NotImplementedError(f"{type(self).__name__}: device path implemented for uint8 and float frames, got {image.dtype}")
        from ..planevm import DeviceProbes

        return self._plan(image, self.variant(image, DeviceProbes, **kw)).run(image)


class SpeciesStreamOp:
    """Adapter that lets pipeline.FramePipeline / run_video stream uint8 frames through a UV species: one recorded
    plan per pipeline slot (each owns its device frames, which the pipeline copies into / out of directly), replayed
    on the slot's stream.  `output`: "out" (the stylised frame) or "baseline".  Species whose plan depends on frame
    content (RatUV's mode="auto") need the branch fixed by `variant`."""

    def __init__(self, species: UVSpecies, H: int, W: int, *, depth: int = 3, variant=None, ctx=None):
        from ..planevm import DeviceBackend

        self.species, self.H, self.W, self.ctx = species, int(H), int(W), ctx
        probe = np.zeros((H, W, 3), np.uint8)
        self.plans = []
        for _ in range(depth):
            be = DeviceBackend(H, W, ctx)
            if variant is None:
                species.render(be, probe)
            else:
                species.render(be, probe, variant)
            be.flush()
            self.plans.append(be)
        self.ctx = self.plans[0].ctx
        self._by_in = {be.d_in.ptr: be for be in self.plans}

    def slot_buffers(self, k: int):
        be = self.plans[k]
        return be.d_in, be.d_out

    def run_device(self, d_in, d_out, n_frames: int, H: int, W: int, stream=None):
        assert n_frames == 1 and (H, W) == (self.H, self.W)
        self._by_in[d_in.ptr].run_device(stream)

    def close(self):
        for be in self.plans:
            be.close()
        self.plans = []


def default_lambdas(lambdas) -> np.ndarray:
    """The 300-700 nm / 81-band grid every UV species but HoneyBee defaults to (e.g. reindeer.py:56-58)."""
    lam = np.asarray(lambdas, dtype=np.float32) if lambdas is not None else np.linspace(300.0, 700.0, 81, dtype=np.float32)
    assert lam.ndim == 1 and lam.size >= 10, "lambdas must be a 1D vector of wavelengths (nm)."
    return lam


def snow_glare_tone_compress(be, x, *, strength: float, knee: float = 0.8):
    """uv_helpers.py:111-121 on one plane."""
    if strength <= 0.0:
        return x
    x = be.clip01(x)
    t = (x - knee) / (1.0 - knee)
    return be.where(x <= knee, x, knee + (1.0 - knee) * (t / (1.0 + strength * t)))


def radial_sigmoid(be, *, radius: float, softness: float):
    """The peripheral blend weight several species share (e.g. goldfish.py:165-170, mantis_shrimp.py:270-273)."""
    yy = be.row(np.linspace(-1.0, 1.0, be.H, dtype=np.float32))
    xx = be.col(np.linspace(-1.0, 1.0, be.W, dtype=np.float32))
    r = be.sqrt(xx * xx + yy * yy)
    return 1.0 / (1.0 + be.exp(-softness * (r - radius)))
