"""Host logic of the dichromat path: 3x3 collapse matrix, OpenCV-style tap generation, and the
DichromatOp that drives the fused HIP kernel (csrc/dichromat.hip) through the C ABI.

Mirrors the per-species template of the reference (animals/dog.py:14-61 and 19 siblings):
validate -> normalise -> sRGB->linear -> collapse matrix -> post stage -> OETF -> dtype restore.
Everything per-pixel runs on the device; only 9-element matrices and <=33 taps are built here."""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import DichromatDesc, lib
from .runtime import Context, DeviceBuffer, get_context

# animals/animal_utils.py:56-63 (float32) and :70-76 (float64: the literal has no dtype)
M_RGB_TO_LMS = np.array(
    [[0.31399022, 0.63951294, 0.04649755], [0.15537241, 0.75789446, 0.08670142], [0.01775239, 0.10944209, 0.87256922]],
    dtype=np.float32,
)
M_LMS_TO_RGB = np.array(
    [[5.472213, -4.6419606, 0.16963711], [-1.125242, 2.2931712, -0.16789523], [0.02980164, -0.19318072, 1.1636479]]
)


def collapse_LMS_matrix(alpha: float, s_scale: float) -> np.ndarray:
    """animals/animal_utils.py:88-119.  T = float32(LMS(E) @ D.T @ M_lms_to_rgb.T); callers apply it
    as `pixels @ T.T`, i.e. out_i = sum_j T[i][j] * in_j -- exactly how the kernel consumes it (Q1)."""
    E = np.eye(3, dtype=np.float32)
    LMS = E @ M_RGB_TO_LMS.T
    D = np.array([[alpha, 1.0 - alpha, 0.0], [alpha, 1.0 - alpha, 0.0], [0.0, 0.0, s_scale]], dtype=np.float32)
    return ((LMS @ D.T) @ M_LMS_TO_RGB.T).astype(np.float32)


def cv_auto_ksize(sigma: float) -> int:
    """OpenCV createGaussianKernels for non-uint8 depth and ksize=(0,0): cvRound(8*sigma+1)|1
    (what cv2.GaussianBlur(img, (0,0), sigma) resolves to in animals/animal_utils.py:144)."""
    return int(round(sigma * 8 + 1)) | 1  # Python round == cvRound (half to even)


def gaussian_taps(ksize: int, sigma: float) -> np.ndarray:
    """cv::getGaussianKernel(ksize, sigma > 0) in double: exp(-x^2/(2 sigma^2)) / sum, summed left to right."""
    scale2x = -0.5 / (sigma * sigma)
    c = (ksize - 1) * 0.5
    t = [math.exp(scale2x * (i - c) * (i - c)) for i in range(ksize)]
    s = 0.0
    for v in t:
        s += v
    s = 1.0 / s
    return np.array([v * s for v in t], dtype=np.float64)


def s_cone_row_gain(H: int, s_top=1.0, s_bottom=0.6, *, power=1.0, extra_boost=0.0, band=None) -> np.ndarray:
    """Per-row gain vector of apply_s_cone_vertical_gain (animals/animal_utils.py:236-250); H floats."""
    w = np.linspace(s_top, s_bottom, H, dtype=np.float32)
    if power != 1.0:
        t = (w - s_bottom) / max(1e-8, (s_top - s_bottom))
        t = np.clip(t, 0.0, 1.0) ** power
        w = s_bottom + (s_top - s_bottom) * t
    if extra_boost != 0.0:
        w = 1.0 + extra_boost * (w - 1.0)
    if band is not None:
        y_center, sigma, peak = band
        yy = np.linspace(0.0, 1.0, H, dtype=np.float32)
        w = w * (1.0 + peak * np.exp(-0.5 * ((yy - y_center) / max(1e-8, sigma)) ** 2))
    return np.ascontiguousarray(w, dtype=np.float32)


STREAK_STRIDE = 48


def streak_row_tables(H: int, y_center: float, sigma_streak: float, sigma_far: float, falloff: float) -> np.ndarray:
    """Per-image-row kernels of apply_anisotropic_acuity_blur_with_streak (animals/animal_utils.py:156-171):
    sigma map :157-162 in float32 exactly as coded, then for each row the two cv2.GaussianBlur kernels
    (ksize (0,0) -> cvRound(8 sigma + 1)|1, getGaussianKernel taps cast to the image's float32).
    Layout per row: [k1, k2, 13 x taps(sigmaX), 33 x taps(sigmaY)]."""
    yy = np.linspace(0, 1, H, dtype=np.float32)[:, None]
    d = np.abs(yy - y_center)
    sigma_map = sigma_streak + (sigma_far - sigma_streak) * (1.0 - np.exp(-falloff * d**2))
    sigmaY = sigma_map
    sigmaX = np.maximum(0.4, 0.5 * sigma_map)
    out = np.zeros((H, STREAK_STRIDE), np.float32)
    cache = {}
    for y in range(H):
        sx, sy = float(sigmaX[y, 0]), float(sigmaY[y, 0])
        key = (sx, sy)
        row = cache.get(key)
        if row is None:
            k1, k2 = cv_auto_ksize(sx), cv_auto_ksize(sy)
            if k1 > 13 or k2 > 33:
                raise ValueError(f"streak sigma too large for the device tables (k1={k1}, k2={k2})")
            row = np.zeros(STREAK_STRIDE, np.float32)
            row[0], row[1] = k1, k2
            row[2 : 2 + k1] = gaussian_taps(k1, sx).astype(np.float32)
            row[15 : 15 + k2] = gaussian_taps(k2, sy).astype(np.float32)
            cache[key] = row
        out[y] = row
    return out


@dataclass
class DichromatSpec:
    """One dichromat species: colour stage + post stage."""

    name: str
    alpha: float
    s_scale: float = 1.0
    color: str = "collapse"      # "collapse": dog.py:46-47 | "cat_merge": cat.py:95-101 (float64 tail)
    post: str = "gauss"          # "gauss" | "scone" | "streak" | "none"
    sigma: float = 0.0
    scone: Tuple[float, float, float, float] = (1.3, 0.5, 1.4, 0.25)  # s_top, s_bottom, power, extra_boost
    streak: Tuple[float, float, float, float] = (0.5, 0.8, 2.2, 6.0)
    chroma: Optional[float] = None   # apply_chroma_compression strength whose result is USED
    

class DichromatOp:
    """Configured fused kernel launch for one species; reusable across frames (stateless between them)."""

    def __init__(self, spec: DichromatSpec, ctx: Optional[Context] = None):
        self.spec = spec
        self.ctx = ctx
        d = DichromatDesc()
        d.struct_size = ctypes.sizeof(DichromatDesc)
        if spec.color == "collapse":
            d.color_mode = _lib.AVX_COLOR_MATRIX
            T = collapse_LMS_matrix(spec.alpha, spec.s_scale)
            d.matrix = (ctypes.c_float * 9)(*T.reshape(-1).tolist())
        elif spec.color == "cat_merge":
            d.color_mode = _lib.AVX_COLOR_CAT_MERGE
            d.cat_alpha = float(np.float32(spec.alpha))
            d.cat_beta = float(np.float32(1.0 - spec.alpha))
        else:
            raise ValueError(f"unknown colour stage {spec.color!r}")
        self._taps = None
        self._gain = None
        if spec.post == "gauss":
            k = cv_auto_ksize(spec.sigma)
            if k > _lib.AVX_MAX_KSIZE:
                raise ValueError(f"sigma {spec.sigma} needs {k} taps > {_lib.AVX_MAX_KSIZE}")
            self._taps = gaussian_taps(k, spec.sigma)
            d.post_mode = _lib.AVX_POST_GAUSS
            d.ksize = k
            d.taps_host = self._taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        elif spec.post == "scone":
            d.post_mode = _lib.AVX_POST_ROWGAIN
            d.row_gain_clamp = 1
        elif spec.post == "streak":
            if spec.color != "collapse":
                raise ValueError("the streak blur runs on the float32 collapse-matrix colour stage")
            d.post_mode = _lib.AVX_POST_STREAK
            d.streak_stride = STREAK_STRIDE
        elif spec.post == "none":
            d.post_mode = _lib.AVX_POST_NONE
        else:
            raise ValueError(f"unknown post stage {spec.post!r}")
        self._streak = None
        if spec.chroma is not None:
            d.chroma_enable = 1
            d.chroma_keep = float(np.float32(1 - spec.chroma))
        self.desc = d

    def _ctx(self) -> Context:
        if self.ctx is None:
            self.ctx = get_context()
        return self.ctx

    def run_device(self, d_in: DeviceBuffer, d_out: DeviceBuffer, n_frames: int, H: int, W: int, stream=None):
        """N uint8 HWC frames resident in HBM -> N uint8 HWC frames; asynchronous on `stream`."""
        ctx = self._ctx()
        need = n_frames * H * W * 3
        if d_in.nbytes < need * (4 if self.desc.in_f32 else 1) or d_out.nbytes < need:
            raise ValueError("device buffers smaller than n_frames*H*W*3")
        if self.spec.post == "scone":
            if self._gain is None or self._gain.size != H:
                s_top, s_bottom, power, boost = self.spec.scone
                self._gain = s_cone_row_gain(H, s_top, s_bottom, power=power, extra_boost=boost)
            self.desc.row_gain_host = self._gain.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        if self.spec.post == "streak":
            if self._streak is None or self._streak.shape[0] != H:
                self._streak = streak_row_tables(H, *self.spec.streak)
            self.desc.streak_rows_host = self._streak.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        ctx._check(lib.avx_dichromat_u8(ctx._h, d_in.ptr, d_out.ptr, n_frames, H, W, ctypes.byref(self.desc), ctx._s(stream)))

    def __call__(self, image: np.ndarray) -> np.ndarray:
        """uint8 HxWx3 (or NxHxWx3) host frame(s) -> same shape uint8, via upload/kernel/download."""
        if image.dtype != np.uint8:
            raise NotImplementedError("device dichromat path takes uint8 frames (float frames: see animal_utils)")
        batch = image if image.ndim == 4 else image[None]
        n, H, W, _ = batch.shape
        ctx = self._ctx()
        d_in = ctx.upload(batch)
        d_out = ctx.malloc(batch.nbytes)
        try:
            self.run_device(d_in, d_out, n, H, W)
            out = ctx.download(d_out, batch.shape, np.uint8)
        finally:
            d_in.free()
            d_out.free()
        return out if image.ndim == 4 else out[0]
