"""Geometric resampling on the device (csrc/geom.hip) with the reference's helper names.

Reference: uv_helpers.py:57-64 (resize_preserve_range), :84-99 (panorama_warp);
animals/cat_widevision_utils.py:11-99 (center_zoom, zoom_scale_from_cat_ratio, animal_fov_binocular_warp);
animals/mantis_shrimp.py:122-131 (Sobel).  Host side: O(W+H) map/weight vectors in NumPy exactly as the
reference builds them; everything O(pixels) runs on the GPU."""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import numpy as np

from ._lib import lib
from .runtime import Context, DeviceBuffer, get_context

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3  # cv2's values


def resize_device(ctx: Context, d_src: DeviceBuffer, dtype, H: int, W: int, C: int, Hd: int, Wd: int, interp: int, d_dst: Optional[DeviceBuffer] = None,
                  stream=None) -> DeviceBuffer:
    item = np.dtype(dtype).itemsize
    if d_dst is None:
        d_dst = ctx.malloc(Hd * Wd * C * item)
    code = 2 if np.dtype(dtype) == np.uint8 else 0
    ctx._check(lib.avx_resize_hwc(ctx._h, d_src.ptr, code, H, W, C, d_dst.ptr, Hd, Wd, int(interp), ctx._s(stream)))
    return d_dst


def resize(img: np.ndarray, dsize: Tuple[int, int], interpolation: int = INTER_LINEAR, ctx: Optional[Context] = None) -> np.ndarray:
    """cv2.resize(img, (W_out, H_out), interpolation=...) for float32 (LINEAR/CUBIC/AREA) or uint8 (LINEAR) images."""
    ctx = ctx or get_context()
    Wd, Hd = int(dsize[0]), int(dsize[1])
    squeeze = img.ndim == 2
    a = np.ascontiguousarray(img[..., None] if squeeze else img)
    if a.dtype != np.uint8:
        a = a.astype(np.float32, copy=False)
    H, W, C = a.shape
    d_in = ctx.upload(a)
    d_out = None
    try:
        d_out = resize_device(ctx, d_in, a.dtype, H, W, C, Hd, Wd, interpolation)
        out = ctx.download(d_out, (Hd, Wd, C), a.dtype)
    finally:
        d_in.free()
        if d_out is not None:
            d_out.free()
    return out[..., 0] if squeeze else out


def resize_preserve_range(x: np.ndarray, out_hw: Tuple[int, int], *, interp: int) -> np.ndarray:
    """uv_helpers.py:57-64."""
    H_out, W_out = out_hw
    was_float = np.issubdtype(x.dtype, np.floating)
    y = resize(x.astype(np.float32, copy=False), (W_out, H_out), interp)
    return y.astype(x.dtype, copy=False) if not was_float else y


def panorama_warp(img_lin: np.ndarray, *, scale_x: float) -> np.ndarray:
    """uv_helpers.py:84-99: INTER_CUBIC widen, centre crop back to W."""
    if abs(scale_x - 1.0) < 1e-3:
        return img_lin
    H, W = img_lin.shape[:2]
    newW = max(2, int(round(W * scale_x)))
    widened = resize(img_lin, (newW, H), INTER_CUBIC)
    if newW == W:
        return widened
    start = (newW - W) // 2
    return widened[:, start : start + W, :]


def center_zoom(image: np.ndarray, scale: float) -> np.ndarray:
    """animals/cat_widevision_utils.py:11-29: crop the centre (W/scale, H/scale), INTER_LINEAR back to (W, H)."""
    assert image.ndim >= 2, "HxW or HxWxC"
    if scale <= 1.0:
        return image
    H, W = image.shape[:2]
    cw, ch = max(1, int(round(W / scale))), max(1, int(round(H / scale)))
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    return resize(np.ascontiguousarray(image[y0 : y0 + ch, x0 : x0 + cw]), (W, H), INTER_LINEAR)


def zoom_scale_from_cat_ratio(*, camera_hfov_deg: float, cat_per_eye_half_fov_deg: float, cat_to_human_ratio: float) -> float:
    """animals/cat_widevision_utils.py:31-44."""
    phi = float(cat_per_eye_half_fov_deg)
    eff_cat_hfov = min(float(camera_hfov_deg), 2.0 * phi)
    ratio = max(1.01, float(cat_to_human_ratio))
    cam = math.tan(math.radians(camera_hfov_deg) * 0.5)
    hum = math.tan(math.radians(eff_cat_hfov / ratio) * 0.5)
    return float(cam / max(hum, 1e-6))


def binocular_warp_tables(H_in: int, W_in: int, out_w: int, out_h: int, fov_in_deg: float, per_eye_half_fov_deg: float, overlap_deg: float):
    """Per-column maps/weights and the per-row y map of cat_widevision_utils.py:66-96.  The reference builds them
    as (out_h, out_w) grids with np.meshgrid, but U depends on the column only and ymap on the row only."""
    phi = np.deg2rad(per_eye_half_fov_deg)
    psi = np.deg2rad(fov_in_deg * 0.5)
    O = np.deg2rad(overlap_deg)
    alpha = max(0.0, phi - 0.5 * O)
    u = np.linspace(-1.0, 1.0, out_w, dtype=np.float32)
    thetaL = u * phi
    thetaR = u * phi
    gammaL, gammaR = thetaL - alpha, thetaR + alpha
    xL = ((gammaL / psi) * (W_in * 0.5) + (W_in * 0.5)).astype(np.float32)
    xR = ((gammaR / psi) * (W_in * 0.5) + (W_in * 0.5)).astype(np.float32)
    ymap = np.linspace(0, H_in - 1, out_h, dtype=np.float32)
    validL = (np.abs(gammaL) <= psi).astype(np.float32)
    validR = (np.abs(gammaR) <= psi).astype(np.float32)
    wL = (np.cos(0.5 * np.pi * (thetaL / phi)) ** 2).astype(np.float32) * validL
    wR = (np.cos(0.5 * np.pi * (thetaR / phi)) ** 2).astype(np.float32) * validR
    return [np.ascontiguousarray(v, np.float32) for v in (xL, xR, ymap, wL, wR)]


def binocular_warp_device(ctx: Context, d_in: DeviceBuffer, H: int, W: int, tables, out_h: int, out_w: int, d_out: DeviceBuffer, stream=None):
    xL, xR, ymap, wL, wR = tables
    ctx._check(lib.avx_binocular_warp_u8(ctx._h, d_in.ptr, H, W, xL.ctypes.data, xR.ctypes.data, ymap.ctypes.data, wL.ctypes.data, wR.ctypes.data,
                                         out_h, out_w, d_out.ptr, ctx._s(stream)))


def animal_fov_binocular_warp_u8(image: np.ndarray, *, fov_in_deg: float, per_eye_half_fov_deg: float, overlap_deg: float,
                                 out_size=None, ctx: Optional[Context] = None) -> np.ndarray:
    """get_normalized_image + animal_fov_binocular_warp (cat.py:82-92) for a uint8 frame -> float32 HxWx3 in [0,1]."""
    ctx = ctx or get_context()
    H, W, _ = image.shape
    out_w, out_h = (W, H) if out_size is None else out_size
    tables = binocular_warp_tables(H, W, out_w, out_h, fov_in_deg, per_eye_half_fov_deg, overlap_deg)
    d_in = ctx.upload(image)
    d_out = ctx.malloc(out_h * out_w * 3 * 4)
    try:
        binocular_warp_device(ctx, d_in, H, W, tables, out_h, out_w, d_out)
        return ctx.download(d_out, (out_h, out_w, 3), np.float32)
    finally:
        d_in.free()
        d_out.free()


def sobel3(plane: np.ndarray, ctx: Optional[Context] = None):
    """(gx, gy) = cv2.Sobel(plane, CV_32F, 1,0 / 0,1, ksize=3, BORDER_REFLECT101)."""
    ctx = ctx or get_context()
    a = np.ascontiguousarray(plane, np.float32)
    H, W = a.shape
    d_in = ctx.upload(a)
    d_g = ctx.malloc(2 * a.nbytes)
    try:
        ctx._check(lib.avx_sobel3_plane(ctx._h, d_in.ptr, H, W, d_g.ptr, d_g.ptr + a.nbytes, ctx.stream))
        g = ctx.download(d_g, (2, H, W), np.float32)
    finally:
        d_in.free()
        d_g.free()
    return g[0], g[1]
