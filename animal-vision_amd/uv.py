"""Host logic + device ops of the UV / spectral path (reference: uv_helpers.py, uv_mappers.py,
ml/classic_rgb_to_hsi/classic_rgb_to_hsi.py, animals/honeybee.py).

Host side = O(bands) table construction in NumPy (wavelength grids, illuminant, receptor curves,
band-pass weights, folding them into K x B or K x 3 weight matrices).  Everything O(pixels) runs in
csrc/uv.hip through the C ABI."""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import HoneybeeDesc, lib
from .dichromat import gaussian_taps
from .runtime import Context, DeviceBuffer, get_context

EPS_DEFAULT: float = 1e-8


# ---------------------------------------------------------------- host tables (NumPy, O(bands)) ----
def D65_like(lambdas_nm: np.ndarray) -> np.ndarray:
    """Daylight-like illuminant of uv_helpers.py:187-192: a broad lobe at 560 nm plus 0.3 of a narrower one at 450 nm,
    scaled to unit mean over the grid (float32 in, float32 arithmetic, like the reference)."""
    grid = np.asarray(lambdas_nm)

    def lobe(centre_nm: float, width_nm: float) -> np.ndarray:
        z = (grid - centre_nm) / width_nm
        return np.exp(-0.5 * z**2)

    spd = lobe(560.0, 50.0) + 0.3 * lobe(450.0, 35.0)
    return (spd / spd.mean()).astype(np.float32)


def bandpass_weights(lambdas: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """Band weights of uv_helpers.py:125-139 over a wavelength grid: a raised-cosine (Hann) window across [lo, hi] scaled to
    unit sum.  Quirk Q7 made explicit: a band that contains no grid wavelength, or whose window sums to (numerically)
    nothing, gets the UNIFORM weights 1/B -- what a UV band on a 400-700 nm grid degenerates to in the reference."""
    grid = lambdas.astype(np.float32)
    uniform = np.full(grid.shape, 1.0, np.float32) / float(grid.size)
    inside = (grid >= lo) & (grid <= hi)
    if not inside.any():
        return uniform
    window = np.zeros_like(grid)
    phase = (grid[inside] - lo) / (hi - lo)
    window[inside] = 0.5 * (1.0 - np.cos(2.0 * np.pi * phase))
    total = float(window.sum())
    return window / total if total > 1e-12 else uniform


def lobe_tables(wavelengths: np.ndarray) -> Tuple[np.ndarray, float]:
    """classic_rgb_to_hsi.py:63-79: per-band lobe gains as a (B,3) table whose column j multiplies INPUT
    channel j (quirk Q5: channel 0 -> 460 nm lobe, channel 2 -> 610 nm lobe) and the scalar denominator."""
    wl = wavelengths.astype(np.float32)
    gR = np.exp(np.float32(-0.5) * ((wl - np.float32(610.0)) / np.float32(60.0)) ** 2)
    gG = np.exp(np.float32(-0.5) * ((wl - np.float32(545.0)) / np.float32(60.0)) ** 2)
    gB = np.exp(np.float32(-0.5) * ((wl - np.float32(460.0)) / np.float32(55.0)) ** 2)
    gB2 = np.exp(np.float32(-0.5) * ((wl - np.float32(460.0)) ** 2) / np.float32(55.0**2))
    denom = float(np.float32((gR + gG + gB2).mean()) + np.float32(1e-8))
    return np.ascontiguousarray(np.stack([gB, gG, gR], axis=1), dtype=np.float32), denom


def honeybee_cone_curves(lambdas: np.ndarray):
    """animals/honeybee.py:179-192 + sum-normalisation :89-93 -> (UV, Blue, Green) float32 curves."""

    def log_normal(lam, peak, sigma):
        return np.exp(-0.5 * ((lam - peak) / sigma) ** 2)

    curves = [log_normal(lambdas, 350.0, 25.0).astype(np.float32), log_normal(lambdas, 440.0, 30.0).astype(np.float32),
              log_normal(lambdas, 540.0, 35.0).astype(np.float32)]
    for v in curves:
        s = v.sum()
        if s > 0:
            v /= s
    return curves


def fold_rgb_matrix(weights_kb: np.ndarray, wavelengths: np.ndarray) -> np.ndarray:
    """K x 3 matrix M with plane_k = sum_j M[k,j] * linear_channel_j == integrate(lobes(rgb)) exactly in
    real arithmetic (the lobe cube is linear in the linear channels and never negative, so clamp_min(0) is
    the identity); evaluated in float64, consumed as float32: ~1e-7 relative to the band-by-band float32."""
    gains, denom = lobe_tables(wavelengths)
    return np.ascontiguousarray((weights_kb.astype(np.float64) @ gains.astype(np.float64)) / denom, dtype=np.float32)


def uv_blur_ksize(sigma: float) -> int:
    """uv_helpers.py:72."""
    return int(2 * np.ceil(3 * sigma) + 1)


# ---------------------------------------------------------------- device ops (NumPy in / out) --------
def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def percentile(x: np.ndarray, q: float, ctx: Optional[Context] = None) -> float:
    """np.percentile(x, q) (linear interpolation) computed on the device: exact order statistics."""
    ctx = ctx or get_context()
    a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    d = ctx.upload(a)
    out = ctypes.c_double()
    try:
        ctx._check(lib.avx_percentile(ctx._h, d.ptr, a.size, float(q), ctypes.byref(out), ctx.stream))
    finally:
        d.free()
    return float(out.value)


def spectral_integrate(hsi: np.ndarray, weights_kb: np.ndarray, *, layout: str = "nhwc", ctx: Optional[Context] = None,
                       return_stats: bool = False):
    """K x (H,W) planes = sum_b hsi[..., b] * weights[k, b] on the device (integrate_band, cone catches)."""
    ctx = ctx or get_context()
    w = np.ascontiguousarray(weights_kb, dtype=np.float32)
    K, B = w.shape
    cube = np.ascontiguousarray(hsi)
    if cube.dtype not in (np.float32, np.float16):
        cube = cube.astype(np.float32)
    if layout == "nhwc":
        H, W, Bc = cube.shape
    else:
        Bc, H, W = cube.shape
    assert Bc == B, "HSI bands must match the weight matrix"
    d_in = ctx.upload(cube)
    d_out = ctx.malloc(4 * K * H * W)
    stats = np.empty((K, 4), np.float32)
    try:
        ctx._check(lib.avx_spectral_integrate(ctx._h, d_in.ptr, 0 if layout == "nhwc" else 1, 0 if cube.dtype == np.float32 else 1,
                                              H, W, B, w.ctypes.data, K, d_out.ptr, stats.ctypes.data, ctx.stream))
        out = ctx.download(d_out, (K, H, W), np.float32)
    finally:
        d_in.free()
        d_out.free()
    return (out, stats) if return_stats else out


def integrate_band(hsi: np.ndarray, lambdas: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """uv_helpers.py:142-146 on the device."""
    return spectral_integrate(hsi, bandpass_weights(lambdas, lo, hi)[None, :])[0]


def planes_blur(planes: np.ndarray, ksize: int, sigma: float, *, scale: str = "none", eps: float = EPS_DEFAULT,
                ctx: Optional[Context] = None) -> np.ndarray:
    """K planes: optional per-plane rescale ("white_patch" | "gray_world" | "safe_norm"), then the
    cv2-semantics Gaussian (uv_helpers.gaussian_blur); ksize == 1 is the rescale alone."""
    ctx = ctx or get_context()
    a = np.ascontiguousarray(planes, dtype=np.float32)
    K, H, W = a.shape
    adapt = {"none": 0, "white_patch": 1, "gray_world": 2, "safe_norm": 3}[scale]
    d_in = ctx.upload(a)
    d_out = ctx.malloc(a.nbytes)
    taps = gaussian_taps(ksize, sigma) if ksize > 1 else np.ones(1)
    try:
        if adapt:
            ctx._check(lib.avx_plane_stats(ctx._h, d_in.ptr, K, H * W, adapt, float(eps), None, ctx.stream))
        ctx._check(lib.avx_planes_gaussian_blur(ctx._h, d_in.ptr, d_out.ptr, K, H, W, ksize, taps.ctypes.data,
                                                {0: 0, 1: 1, 2: 1, 3: 3}[adapt], ctx.stream))
        out = ctx.download(d_out, a.shape, np.float32)
    finally:
        d_in.free()
        d_out.free()
    return out


def gaussian_blur(img: np.ndarray, sigma: float) -> np.ndarray:
    """uv_helpers.py:67-73 (cv2 branch) on the device for float32 HxW or HxWxC images."""
    if sigma <= 0:
        return img
    k = uv_blur_ksize(sigma)
    a = np.asarray(img, dtype=np.float32)
    planes = a[None] if a.ndim == 2 else np.ascontiguousarray(a.transpose(2, 0, 1))
    out = planes_blur(planes, k, sigma)
    return out[0] if a.ndim == 2 else np.ascontiguousarray(out.transpose(1, 2, 0))


def safe_norm(x: np.ndarray) -> np.ndarray:
    """uv_helpers.py:47-53 on the device."""
    a = np.asarray(x, dtype=np.float32)
    return planes_blur(a[None], 1, 0.0, scale="safe_norm")[0]


def classic_rgb_to_hsi(frame: np.ndarray, *, wavelengths: np.ndarray = np.linspace(400.0, 700.0, 31, dtype=np.float32),
                       device: str = "cuda", ctx: Optional[Context] = None) -> np.ndarray:
    """ml/classic_rgb_to_hsi/classic_rgb_to_hsi.py:25-82, analytic branch (the one a GPU box executes, F5):
    HxWx3 frame -> HxWxB float32 cube.  uint8 frames go through the decode table (== /255 + sRGB EOTF)."""
    assert frame.ndim == 3 and frame.shape[2] == 3, "Input must be HxWx3."
    if wavelengths.size < 2:
        raise ValueError("Need at least two wavelengths.")
    step = float(wavelengths[1] - wavelengths[0])
    if not np.allclose(np.diff(wavelengths), step):
        raise ValueError("`wavelengths` must be uniformly spaced.")
    ctx = ctx or get_context()
    H, W, _ = frame.shape
    gains, denom = lobe_tables(wavelengths)
    B = gains.shape[0]
    is_u8 = frame.dtype == np.uint8
    src = np.ascontiguousarray(frame if is_u8 else frame.astype(np.float32))
    d_in = ctx.upload(src)
    d_out = ctx.malloc(4 * H * W * B)
    try:
        ctx._check(lib.avx_rgb_to_hsi_lobes(ctx._h, d_in.ptr, 1 if is_u8 else 0, H, W, B, gains.ctypes.data, float(denom), d_out.ptr, ctx.stream))
        out = ctx.download(d_out, (H, W, B), np.float32)
    finally:
        d_in.free()
        d_out.free()
    return out


class HoneybeeOp:
    """Configured device pipeline for HoneyBee.visualize steps 1-7 (animals/honeybee.py:99-175)."""

    def __init__(self, *, lambdas: np.ndarray, illuminant: Callable[[np.ndarray], np.ndarray], curves: Sequence[np.ndarray],
                 reflectance: bool, adaptation: Optional[str], mapping_mode: str, custom_matrix: Optional[np.ndarray],
                 blur_sigma_px: float, eps: float = EPS_DEFAULT, ctx: Optional[Context] = None):
        self.ctx = ctx
        E = illuminant(lambdas).astype(np.float32) if reflectance else np.ones_like(lambdas, dtype=np.float32)
        # weights[k][b] = E[b] * curve_k[b]  (radiance = hsi * E, then tensordot with the curve)
        self.weights = np.ascontiguousarray(np.stack([E * c for c in curves]), dtype=np.float32)
        self.rgb_matrix = fold_rgb_matrix(self.weights, lambdas)
        d = HoneybeeDesc()
        d.struct_size = ctypes.sizeof(HoneybeeDesc)
        d.source = 0
        d.rgb_matrix = (ctypes.c_float * 9)(*self.rgb_matrix.reshape(-1).tolist())
        d.bands = int(lambdas.size)
        d.weights_host = _fptr(self.weights)
        d.adaptation = {None: 0, "white_patch": 1, "gray_world": 2}[adaptation]
        d.eps = float(eps)
        self._taps = None
        if blur_sigma_px > 0:
            k = uv_blur_ksize(blur_sigma_px)
            self._taps = gaussian_taps(k, blur_sigma_px)
            d.blur_ksize = k
            d.blur_taps_host = self._taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        if mapping_mode not in _lib.AVX_MAP:
            raise ValueError(f"Unknown mapping_mode: {mapping_mode}")
        d.mapping = _lib.AVX_MAP[mapping_mode]
        if mapping_mode == "custom_matrix":
            assert custom_matrix is not None and np.shape(custom_matrix) == (3, 3), "Provide custom_matrix as 3x3 for 'custom_matrix' mode."
            d.custom_matrix = (ctypes.c_float * 9)(*np.asarray(custom_matrix, np.float32).reshape(-1).tolist())
        d.mixed_alpha = 0.45  # honeybee.py:162
        self.desc = d

    def _ctx(self) -> Context:
        if self.ctx is None:
            self.ctx = get_context()
        return self.ctx

    def padded_clone(self, bands: int) -> "HoneybeeOp":
        """The same pipeline for a cube stored with `bands` >= B channels per pixel, the extra ones being padding
        (weights zero): what MSTPlusPlus.forward_nhwc emits (31 bands in a 32-wide, 64-byte-aligned group)."""
        import copy

        op = copy.copy(self)
        K, B = self.weights.shape
        assert bands >= B
        op.weights = np.zeros((K, bands), np.float32)
        op.weights[:, :B] = self.weights
        d = HoneybeeDesc()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(self.desc), ctypes.sizeof(HoneybeeDesc))
        d.bands = bands
        d.weights_host = _fptr(op.weights)
        op.desc = d
        return op

    def run_device(self, d_in: Optional[DeviceBuffer], d_out: DeviceBuffer, n_frames: int, H: int, W: int, *, hsi_ptr: int = 0,
                   hsi_layout: int = 1, hsi_dtype: int = 0, debug: Optional[DeviceBuffer] = None, stream=None, catches=None):
        """uint8 frames (or an HSI cube at hsi_ptr, e.g. the MST++ output tensor; or catches = (planes_ptr, partials_ptr, n_partials): the three
        catch planes already formed with their statistics, ml/mst_plus_plus.py::conv3x3_lds_spectral) -> uint8 frames, all on device."""
        ctx = self._ctx()
        d = self.desc
        if catches is not None:
            d.source, d.hsi = 2, None
            d.catches, d.catch_partials, d.n_catch_partials = int(catches[0]), int(catches[1]), int(catches[2])
        elif hsi_ptr:
            d.source, d.hsi, d.hsi_layout, d.hsi_dtype = 1, hsi_ptr, hsi_layout, hsi_dtype
        else:
            d.source, d.hsi = 0, None
        ctx._check(lib.avx_honeybee_u8(ctx._h, d_in.ptr if d_in else None, d_out.ptr, n_frames, H, W, ctypes.byref(d),
                                       debug.ptr if debug else None, ctx._s(stream)))

    def __call__(self, image: np.ndarray, *, hsi: Optional[np.ndarray] = None, hsi_layout: str = "nhwc", return_planes: bool = False):
        if image.dtype != np.uint8:
            raise NotImplementedError("device honeybee path takes uint8 frames")
        H, W, _ = image.shape
        ctx = self._ctx()
        d_in = ctx.upload(image)
        d_out = ctx.malloc(image.nbytes)
        d_dbg = ctx.malloc(4 * 3 * H * W) if return_planes else None
        d_hsi = None
        try:
            if hsi is not None:
                cube = np.ascontiguousarray(hsi)
                d_hsi = ctx.upload(cube)
                self.run_device(d_in, d_out, 1, H, W, hsi_ptr=d_hsi.ptr, hsi_layout=0 if hsi_layout == "nhwc" else 1,
                                hsi_dtype=0 if cube.dtype == np.float32 else 1, debug=d_dbg)
            else:
                self.run_device(d_in, d_out, 1, H, W, debug=d_dbg)
            out = ctx.download(d_out, image.shape, np.uint8)
            planes = ctx.download(d_dbg, (3, H, W), np.float32) if return_planes else None
        finally:
            for b in (d_in, d_out, d_dbg, d_hsi):
                if b is not None:
                    b.free()
        return (out, planes) if return_planes else out
