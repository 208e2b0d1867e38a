"""The reference's uv_helpers.py by name -- NumPy in / NumPy out with everything O(pixels) on the device -- for code written
against the reference (`from uv_helpers import to_float01, srgb_to_linear, gaussian_blur, ...`, e.g. animals/reindeer.py:6-24).

The species classes of this package do not come through here (their whole frame is one recorded plan, animals/_uv_species.py);
this module is the step-wise surface: same names, argument meaning, shapes, dtypes and errors as the reference's helpers.
Arithmetic is float32 in the reference's order with the device's transcendentals (powf / cosf: <= 2 ulp from NumPy's);
`cv2.GaussianBlur` / `cv2.resize` are the restated OpenCV algorithms of csrc/uv.hip and csrc/geom.hip (bit-exact with the
oracle; OpenCV itself is not available to pin against).  No CPU path: a missing library raises at import."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from .geometry import INTER_AREA, INTER_LINEAR, panorama_warp, resize_preserve_range  # noqa: F401  (uv_helpers.py:57-64, :84-99)
from .uv import EPS_DEFAULT, D65_like, bandpass_weights, gaussian_blur, integrate_band, safe_norm  # noqa: F401  (:47-53, :67-81, :125-152, :187-192)


def _rows(x: np.ndarray):
    a = np.ascontiguousarray(x, dtype=np.float32)
    shape = a.shape
    flat = a.reshape(1, -1) if a.ndim < 2 else a.reshape(shape[0], -1)
    return flat, lambda r: r.reshape(shape)


def _elementwise(x: np.ndarray, fn) -> np.ndarray:
    from .planevm import run_planes

    x = np.asarray(x)
    if x.size == 0:
        return x.astype(np.float32)
    flat, restore = _rows(x)
    return restore(run_planes([flat], lambda be, v: [fn(be, v[0])])[0])


# -------------------- dtype & colour transforms --------------------
def to_float01(x: np.ndarray) -> np.ndarray:
    """uv_helpers.py:15-23: uint8 -> / 255; floats: `/ 255` and clip only when the frame's maximum exceeds 1.001 (a frame-wide
    reduction, on the device).  Always float32."""
    from .planevm import run_planes

    x = np.asarray(x)
    if x.dtype == np.uint8:
        return _elementwise(x, lambda be, v: v / 255.0)
    flat, restore = _rows(x)

    def build(be, v):
        mx = be.max(v[0])
        return [be.where(mx > 1.001, be.clip01(v[0] / 255.0), v[0])]

    return restore(run_planes([flat], build)[0])


def from_float01(img01: np.ndarray, dtype) -> np.ndarray:
    """uv_helpers.py:26-30: integer dtypes get clip(x * 255 + 0.5, 0, 255) truncated; float dtypes a cast."""
    if np.issubdtype(dtype, np.integer):
        return _elementwise(img01, lambda be, v: be.clip(v * 255.0 + 0.5, 0.0, 255.0)).astype(dtype)
    return np.asarray(img01).astype(dtype)


def srgb_to_linear(s: np.ndarray) -> np.ndarray:
    """uv_helpers.py:33-37 (float32 result whatever the input dtype)."""
    a = 0.055
    return _elementwise(s, lambda be, v: be.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4))


def linear_to_srgb(l: np.ndarray) -> np.ndarray:  # noqa: E741  (the reference's argument name)
    """uv_helpers.py:40-44: the power branch sees clip(l, 0, None)."""
    a = 0.055
    return _elementwise(l, lambda be, v: be.where(v <= 0.0031308, v * 12.92, (1 + a) * be.power(be.maximum(v, 0.0), 1 / 2.4) - a))


# -------------------- optics-ish tweaks --------------------
def apply_scatter_and_blue_bias(img_lin: np.ndarray, *, sigma: float, blue_bias: float) -> np.ndarray:
    """uv_helpers.py:100-106: blur when sigma > 0.15, then B <- clip(B + blue_bias, 0, 1) (channel 2; the others pass through)."""
    out = np.array(img_lin, copy=True)
    if sigma > 0.15:
        out = gaussian_blur(out, sigma)
    bb = float(blue_bias)
    out[..., 2] = _elementwise(out[..., 2], lambda be, v: be.clip01(v + bb))
    return out


def snow_glare_tone_compress(img_lin: np.ndarray, *, strength: float, knee: float = 0.8) -> np.ndarray:
    """uv_helpers.py:109-119: soft-knee highlight compression of clip(x, 0, 1); strength <= 0 returns the argument itself."""
    if strength <= 0.0:
        return img_lin
    k, st = float(knee), float(strength)

    def fn(be, v):
        x = be.clip01(v)
        t = (x - k) / (1.0 - k)
        return be.where(x <= k, x, k + (1.0 - k) * (t / (1.0 + st * t)))

    out = _elementwise(img_lin, fn)
    return out.astype(img_lin.dtype, copy=False) if np.issubdtype(np.asarray(img_lin).dtype, np.floating) else out


# -------------------- spectral helpers --------------------
def integrate_uv(hsi: np.ndarray, lambdas: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """uv_helpers.py:148-152: integrate_band then safe_norm."""
    return safe_norm(integrate_band(hsi, lambdas, lo, hi))


def classic_rgb_to_hsi_scaled(rgb01: np.ndarray, *, wavelengths: np.ndarray, scale: float,
                              converter: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None) -> np.ndarray:
    """uv_helpers.py:155-183: INTER_AREA down to round(H s) x round(W s) -> RGB -> HSI -> INTER_LINEAR back up.
    converter: a `classic_rgb_to_hsi(rgb01, wavelengths)`-like callable; default = this package's device version of the
    reference's analytic branch (animal_vision_amd.uv.classic_rgb_to_hsi)."""
    assert 0.0 < scale <= 1.0, "scale must be (0,1]."
    H, W = rgb01.shape[:2]
    h_small = max(1, int(round(H * scale)))
    w_small = max(1, int(round(W * scale)))
    rgb_small = resize_preserve_range(rgb01, (h_small, w_small), interp=INTER_AREA)
    if converter is None:
        from .uv import classic_rgb_to_hsi

        hsi_small = classic_rgb_to_hsi(rgb_small, wavelengths=np.asarray(wavelengths, dtype=np.float32))
    else:
        hsi_small = converter(rgb_small, np.asarray(wavelengths, dtype=np.float32))
    return resize_preserve_range(hsi_small, (H, W), interp=INTER_LINEAR)


# -------------------- adaptation --------------------
def _von_kries(U, B, G, eps: float, kind: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    from .planevm import run_planes

    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in (U, B, G)]
    shape = planes[0].shape
    flat = [p.reshape(1, -1) if p.ndim < 2 else p.reshape(shape[0], -1) for p in planes]
    e = float(np.float32(eps))

    def build(be, v):
        return [c / be.maximum(be.max(c) if kind == "max" else be.mean(c), e) for c in v]

    out = run_planes(flat, build)
    return tuple(o.reshape(shape) for o in out)


def von_kries_white_patch(U: np.ndarray, B: np.ndarray, G: np.ndarray, eps: float = EPS_DEFAULT):
    """uv_helpers.py:195-199: each plane divided by max(its maximum, eps)."""
    return _von_kries(U, B, G, eps, "max")


def von_kries_gray_world(U: np.ndarray, B: np.ndarray, G: np.ndarray, eps: float = EPS_DEFAULT):
    """uv_helpers.py:202-206: each plane divided by max(its mean, eps)."""
    return _von_kries(U, B, G, eps, "mean")
