"""oracle/cpu_ref.py -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

CPU restatement (NumPy + the small C++ library in oracle/avxref.cpp) of the
reference's per-frame hot path, SURVEY.md section 8(a).  Every function cites the
reference file:line it follows (paths relative to /root/reference).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Pinning status (tests/test_oracle_golden.py, fixtures made by
tools/make_goldens.py from the imported reference):
  * pure-NumPy stages (a2-a7, a10-a12, a15-a18, a20-a22): pinned bit-exact
    against the reference's own functions executed in the build container;
  * a13 analytic RGB->HSI lobes: pinned against the reference function executed
    with its torch calls redirected from "cuda" to CPU;
  * every cv2-backed stage (a8, a9, a19, Sobel): OpenCV is neither in
    /root/reference nor installed -> restated from OpenCV 4.x's published
    algorithm, PARITY UNPINNED.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libavxref.so")
_lib = None


def build_lib() -> str:
    """Compile oracle/avxref.cpp with g++ (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in ("avxref.cpp", "cvref.cpp")]
    if not os.path.exists(_LIB_PATH) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_lib())
        for name in (
            "avxref_gaussian_kernel",
            "avxref_sepfilter_f32",
            "avxref_sepfilter_f64",
            "avxref_matmul3_f32",
            "avxref_matmul3_f32_f64",
            "avxref_quantize_f32",
            "avxref_quantize_f64",
            "cvref_resize_linear_f32",
            "cvref_resize_linear_u8",
            "cvref_resize_cubic_f32",
            "cvref_resize_area_f32",
            "cvref_resize_nearest_f32",
            "cvref_remap_linear_f32",
            "cvref_sobel3_f32",
        ):
            getattr(_lib, name).restype = None
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# =============================================================================
# Dichromat path: animals/animal_utils.py
# =============================================================================
def srgb_to_linear(x: np.ndarray) -> np.ndarray:
    """animals/animal_utils.py:5-11 (dtype follows x: f32 in -> f32 out)."""
    a = 0.055
    return np.where(x <= 0.04045, x / 12.92, ((x + a) / (1 + a)) ** 2.4)


def linear_to_srgb(x: np.ndarray) -> np.ndarray:
    """animals/animal_utils.py:13-19."""
    a = 0.055
    return np.where(x <= 0.0031308, 12.92 * x, (1 + a) * (x ** (1 / 2.4)) - a)


def check_input_image(image) -> bool:
    """animals/animal_utils.py:21-39."""
    if not isinstance(image, np.ndarray):
        return False
    if image.ndim != 3 or image.shape[2] != 3:
        return False
    return bool(np.issubdtype(image.dtype, np.number))


def get_normalized_image(image: np.ndarray) -> np.ndarray:
    """animals/animal_utils.py:41-50: data-dependent /255 (only if max > 1.0)."""
    out = image.astype(np.float32)
    if out.max() > 1.0:
        out /= 255.0
    return np.clip(out, 0.0, 1.0)


M_RGB_TO_LMS = np.array(
    [
        [0.31399022, 0.63951294, 0.04649755],
        [0.15537241, 0.75789446, 0.08670142],
        [0.01775239, 0.10944209, 0.87256922],
    ],
    dtype=np.float32,
)  # animals/animal_utils.py:56-63 (float32)

M_LMS_TO_RGB = np.array(
    [
        [5.472213, -4.6419606, 0.16963711],
        [-1.125242, 2.2931712, -0.16789523],
        [0.02980164, -0.19318072, 1.1636479],
    ]
)  # animals/animal_utils.py:70-76 (no dtype => float64)


def matmul3(vec: np.ndarray, M: np.ndarray) -> np.ndarray:
    """`vec @ M.T` for (N,3) pixels, as an explicit FMA chain (see avxref.cpp).

    f32 x f32 -> f32 (animals/dog.py:47, animal_utils.py:64);
    f32 x f64 -> f64 (animal_utils.py:77)."""
    vec = np.ascontiguousarray(vec, dtype=np.float32)
    n = vec.shape[0]
    if M.dtype == np.float32:
        out = np.empty((n, 3), np.float32)
        Mc = np.ascontiguousarray(M, np.float32)
        lib().avxref_matmul3_f32(_p(vec), _p(out), ctypes.c_size_t(n), _p(Mc))
    else:
        out = np.empty((n, 3), np.float64)
        Mc = np.ascontiguousarray(M, np.float64)
        lib().avxref_matmul3_f32_f64(_p(vec), _p(out), ctypes.c_size_t(n), _p(Mc))
    return out


def collapse_LMS_matrix(alpha: float, s_scale: float) -> np.ndarray:
    """animals/animal_utils.py:88-119.  Quirk Q1: the returned matrix is used as
    `pixels @ T.T` by callers, i.e. out_i = sum_j T[i,j] in_j with T as returned."""
    E = np.eye(3, dtype=np.float32)
    LMS = E @ M_RGB_TO_LMS.T
    D = np.array([[alpha, 1.0 - alpha, 0.0], [alpha, 1.0 - alpha, 0.0], [0.0, 0.0, s_scale]], dtype=np.float32)
    LMS_collapsed = LMS @ D.T
    RGB_out = LMS_collapsed @ M_LMS_TO_RGB.T
    return RGB_out.astype(np.float32)


# ---- OpenCV GaussianBlur restatement (PARITY UNPINNED: no cv2 anywhere) ------
def cv_round(x: float) -> int:
    """cvRound: round half to even (lrint)."""
    return int(np.rint(x))


def cv_auto_ksize(sigma: float, depth_is_u8: bool = False) -> int:
    """createGaussianKernels: ksize = cvRound(sigma*(u8?3:4)*2+1)|1 when ksize==(0,0)."""
    return cv_round(sigma * (3 if depth_is_u8 else 4) * 2 + 1) | 1


def gaussian_kernel(n: int, sigma: float, dtype=np.float32) -> np.ndarray:
    """cv::getGaussianKernel(n, sigma>0, max(depth, CV_32F)): f64 taps cast to dtype."""
    out = np.empty(n, np.float64)
    lib().avxref_gaussian_kernel(ctypes.c_int(n), ctypes.c_double(float(sigma)), _p(out))
    return out.astype(dtype)


def sepfilter(img: np.ndarray, kx: np.ndarray, ky: np.ndarray) -> np.ndarray:
    """Separable filter, BORDER_REFLECT_101, row pass then column pass, f32 or f64."""
    squeeze = img.ndim == 2
    a = img[..., None] if squeeze else img
    H, W, C = a.shape
    if a.dtype == np.float64:
        a = np.ascontiguousarray(a)
        kx = np.ascontiguousarray(kx, np.float64)
        ky = np.ascontiguousarray(ky, np.float64)
        out = np.empty_like(a)
        lib().avxref_sepfilter_f64(_p(a), _p(out), H, W, C, _p(kx), len(kx), _p(ky), len(ky))
    else:
        a = np.ascontiguousarray(a, np.float32)
        kx = np.ascontiguousarray(kx, np.float32)
        ky = np.ascontiguousarray(ky, np.float32)
        out = np.empty_like(a)
        lib().avxref_sepfilter_f32(_p(a), _p(out), H, W, C, _p(kx), len(kx), _p(ky), len(ky))
    return out[..., 0] if squeeze else out


def cv_gaussian_blur(img: np.ndarray, ksize: Tuple[int, int], sigma_x: float, sigma_y: float = 0.0) -> np.ndarray:
    """cv2.GaussianBlur(img, ksize, sigmaX, sigmaY, BORDER_REFLECT_101) for f32/f64 images.

    ksize == (0,0) -> derived from sigma; sigmaY <= 0 -> sigmaX."""
    if sigma_y <= 0:
        sigma_y = sigma_x
    kw, kh = ksize
    if kw <= 0:
        kw = cv_auto_ksize(sigma_x)
    if kh <= 0:
        kh = cv_auto_ksize(sigma_y)
    if kw == 1 and kh == 1:
        return img.copy()
    dt = np.float64 if img.dtype == np.float64 else np.float32
    return sepfilter(img, gaussian_kernel(kw, sigma_x, dt), gaussian_kernel(kh, sigma_y, dt))


def apply_acuity_blur(image: np.ndarray, sigma: float = 1.5) -> np.ndarray:
    """animals/animal_utils.py:121-145 (float images keep their dtype: f64 stays f64)."""
    if image.ndim != 3 or image.shape[2] != 3:
        raise ValueError("Expected HxWx3 image")
    dtype = image.dtype
    image_f = image.astype(np.float32, copy=False) if np.issubdtype(dtype, np.integer) else image
    blurred = cv_gaussian_blur(image_f, (0, 0), sigma, sigma)
    return blurred.astype(dtype, copy=False)


def streak_sigma_maps(H: int, y_center: float, sigma_streak: float, sigma_far: float, falloff: float):
    """animals/animal_utils.py:157-162 -> (sigmaX[H], sigmaY[H]) float32."""
    yy = np.linspace(0, 1, H, dtype=np.float32)[:, None]
    d = np.abs(yy - y_center)
    sigma_map = sigma_streak + (sigma_far - sigma_streak) * (1.0 - np.exp(-falloff * d**2))
    sigmaY = sigma_map
    sigmaX = np.maximum(0.4, 0.5 * sigma_map)
    return sigmaX[:, 0], sigmaY[:, 0]


def apply_anisotropic_acuity_blur_with_streak(image, y_center=0.5, sigma_streak=0.8, sigma_far=2.2, falloff=6.0):
    """animals/animal_utils.py:147-172, quirks Q3/Q4 reproduced as coded.

    Each row `out[y]` of shape (W,3) is handed to cv2.GaussianBlur, which sees a
    W-row x 3-column single-channel image: pass 1 blurs across (R,G,B) and along
    the image row with sigmaX; pass 2 (sigmaX=1e-16 -> 1 tap) blurs along the
    image row again with sigmaY.  f32 input is mutated in place (copy=False)."""
    H = image.shape[0]
    sx, sy = streak_sigma_maps(H, y_center, sigma_streak, sigma_far, falloff)
    out = image.astype(np.float32, copy=False)
    tmp = np.empty_like(out)
    for y in range(H):
        tmp[y] = cv_gaussian_blur(out[y], (0, 0), float(sx[y]), 0.0)
    for y in range(H):
        out[y] = cv_gaussian_blur(tmp[y], (0, 0), 1e-16, float(sy[y]))
    return out.astype(image.dtype, copy=False)


def apply_chroma_compression(image: np.ndarray, strength: float = 0.4):
    """animals/animal_utils.py:174-181."""
    gray = image.mean(axis=2, keepdims=True)
    return gray + (image - gray) * (1 - strength)


def apply_tapetum_bloom(image: np.ndarray, strength: float = 0.12, sigma: float = 3.0) -> np.ndarray:
    """animals/animal_utils.py:183-204 (defined, referenced only from a commented block of cat.py): luminance-gated screen blend with a
    blurred copy.  cv2.GaussianBlur -> the shared Gaussian contract (cv_gaussian_blur, ksize from sigma)."""
    x = np.clip(image.astype(np.float32, copy=False), 0.0, 1.0)
    L = 0.2126 * x[..., 0] + 0.7152 * x[..., 1] + 0.0722 * x[..., 2]
    mask = np.clip((L - 0.4) / 0.6, 0.0, 1.0)
    mask = cv_gaussian_blur(mask, (0, 0), sigma, sigma)[..., None]
    blur = cv_gaussian_blur(x, (0, 0), sigma, sigma)
    screen = 1.0 - (1.0 - x) * (1.0 - blur)
    y = x + strength * mask * (screen - x)
    return np.clip(y, 0.0, 1.0).astype(image.dtype, copy=False)


def apply_rod_vision(image: np.ndarray, chroma_scale: float = 0.08, luminance_boost: float = 1.4, gamma: float = 0.8) -> np.ndarray:
    """animals/animal_utils.py:261-305 (defined, not called): scotopic luminance (blurred, sigma 1.2), heavy desaturation, boost, gamma."""
    x = np.clip(image.astype(np.float32), 0.0, 1.0)
    L = 0.1 * x[..., 0] + 0.8 * x[..., 1] + 0.1 * x[..., 2]
    L = cv_gaussian_blur(L, (0, 0), 1.2, 1.2)
    gray = L[..., None]
    x = gray * (1 - chroma_scale) + x * chroma_scale
    x = np.clip(x * luminance_boost, 0.0, 1.0)
    x = np.power(x, gamma)
    return x.astype(image.dtype, copy=False)


def apply_s_cone_vertical_gain(image_lin, s_top=1.0, s_bottom=0.6, *, power=1.0, extra_boost=0.0, band=None, clamp=True):
    """animals/animal_utils.py:206-259 (mutates an f32 input in place, like the reference)."""
    out = image_lin.astype(np.float32, copy=False)
    H, W = out.shape[:2]
    w = s_cone_row_gain(H, s_top, s_bottom, power=power, extra_boost=extra_boost, band=band)
    w2d = w[:, None]
    if clamp:
        out[..., 2] = np.clip(out[..., 2] * w2d, 0.0, 1.0)
    else:
        out[..., 2] = out[..., 2] * w2d
    return out


def s_cone_row_gain(H, s_top, s_bottom, *, power=1.0, extra_boost=0.0, band=None) -> np.ndarray:
    """Row gain vector of animals/animal_utils.py:236-250."""
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
        bump = 1.0 + peak * np.exp(-0.5 * ((yy - y_center) / max(1e-8, sigma)) ** 2)
        w = w * bump
    return w


def encode_u8(result_in_rgb: np.ndarray, orig_dtype=np.uint8) -> np.ndarray:
    """animals/dog.py:54-59: clip -> OETF -> clip -> (x*255+0.5).astype (truncation)."""
    s = np.clip(linear_to_srgb(np.clip(result_in_rgb, 0.0, 1.0)), 0.0, 1.0)
    if np.issubdtype(orig_dtype, np.integer):
        return (s * 255.0 + 0.5).astype(orig_dtype)
    return s.astype(orig_dtype)


@dataclass
class DichromatSpec:
    """One dichromat species = colour stage + post stage."""

    name: str
    alpha: float
    s_scale: float = 1.0
    color: str = "collapse"  # "collapse" (dog.py:46-47) | "cat_merge" (cat.py:95-101, float64 tail)
    post: str = "gauss"  # "gauss" | "streak" | "scone" | "none"
    sigma: float = 0.0
    streak: Tuple[float, float, float, float] = (0.5, 0.8, 2.2, 6.0)  # y_center, sigma_streak, sigma_far, falloff
    streak_result_discarded: bool = False  # pig.py:35 relies on the in-place side effect (Q4)
    chroma: Optional[float] = None  # apply_chroma_compression strength
    chroma_discarded: bool = False  # pig.py:38 discards the pure result
    scone: Tuple[float, float, float, float] = (1.3, 0.5, 1.4, 0.25)  # s_top, s_bottom, power, extra_boost


# SURVEY.md Appendix A (file:line per species there).
DICHROMATS = {
    s.name: s
    for s in [
        DichromatSpec("dog", 0.58, 0.65, sigma=3.5),
        DichromatSpec("cat", 0.5, 1.0, color="cat_merge", sigma=1.0),
        DichromatSpec("sheep", 0.74, 1.06, post="streak", streak=(0.48, 0.8, 2.2, 6.0)),
        DichromatSpec("pig", 0.89, 1.32, post="streak", streak=(0.5, 1.2, 2.5, 3.0), streak_result_discarded=True, chroma=0.55, chroma_discarded=True),
        DichromatSpec("cow", 0.84, 1.07, post="streak", streak=(0.5, 0.9, 2.3, 6.5)),
        DichromatSpec("goat", 0.75, 1.06, post="streak", streak=(0.5, 0.8, 2.4, 8.0)),
        DichromatSpec("horse", 0.30, 1.02, post="streak", streak=(0.5, 0.8, 2.2, 6.0)),
        DichromatSpec("rabbit", 0.20, 1.01, post="streak", streak=(0.52, 0.9, 2.5, 5.0), chroma=0.06),
        DichromatSpec("panda", 0.58, 0.74, post="streak", streak=(0.52, 1.0, 2.1, 4.5), chroma=0.06),
        DichromatSpec("deer", 0.60, 0.95, post="streak", streak=(0.5, 0.8, 2.6, 8.0)),
        DichromatSpec("kangaroo", 0.60, 0.98, post="streak", streak=(0.55, 0.8, 2.3, 8.0)),
        DichromatSpec("rat", 0.05, 0.86, post="scone"),
        DichromatSpec("squirrel", 0.55, 1.05, sigma=0.7),
        DichromatSpec("elephant", 0.60, 0.95, sigma=1.8),
        DichromatSpec("lion", 0.60, 0.95, sigma=1.2),
        DichromatSpec("tiger", 0.60, 0.95, sigma=1.2),
        DichromatSpec("bear", 0.60, 0.95, sigma=1.6),
        DichromatSpec("wolf", 0.65, 0.95, sigma=1.4),
        DichromatSpec("fox", 0.65, 0.98, sigma=1.3),
        DichromatSpec("raccoon", 0.60, 0.98, sigma=2.0),
    ]
}


def dichromat_color_stage(spec: DichromatSpec, lin: np.ndarray) -> np.ndarray:
    """Linear RGB (H,W,3 f32) -> collapsed linear RGB.  f32 for "collapse"
    (dog.py:43-48), f64 for "cat_merge" (cat.py:96-101)."""
    H, W, _ = lin.shape
    vec = lin.reshape(-1, 3)
    if spec.color == "collapse":
        T = collapse_LMS_matrix(spec.alpha, spec.s_scale)
        return matmul3(vec, T).reshape(H, W, 3)
    lms = matmul3(vec, M_RGB_TO_LMS)  # cat.py:97
    alpha = spec.alpha
    LM = alpha * lms[:, 0] + (1.0 - alpha) * lms[:, 1]  # cat.py:99 (f32, separate mul/mul/add)
    merged = np.stack([LM, LM, lms[:, 2]], axis=1)
    return matmul3(merged, M_LMS_TO_RGB).reshape(H, W, 3)  # cat.py:101 -> float64


def dichromat_visualize(spec: DichromatSpec, image: np.ndarray, *, blur=None) -> Tuple[np.ndarray, np.ndarray]:
    """Template of animals/dog.py:14-61 and its 19 siblings (cat.py:94-112 for the
    colour core of Cat; Cat's zoom/FOV warp are SURVEY 8f 'next' rows and not here).

    `blur(img, sigma)` overrides the Gaussian (goldens use identity + this oracle's)."""
    assert check_input_image(image)
    orig_dtype = image.dtype
    lin = srgb_to_linear(get_normalized_image(image))
    rgb = dichromat_color_stage(spec, lin)
    if spec.post == "gauss":
        rgb = (blur or apply_acuity_blur)(rgb, spec.sigma)
    elif spec.post == "streak":
        r = apply_anisotropic_acuity_blur_with_streak(rgb, *spec.streak)
        if not spec.streak_result_discarded:
            rgb = r  # (same values either way: f32 input is blurred in place, Q4)
    elif spec.post == "scone":
        s_top, s_bottom, power, boost = spec.scone
        rgb = apply_s_cone_vertical_gain(rgb, s_top=s_top, s_bottom=s_bottom, power=power, extra_boost=boost)
    if spec.chroma is not None:
        c = apply_chroma_compression(rgb, spec.chroma)
        if not spec.chroma_discarded:
            rgb = c
    return image, encode_u8(rgb, orig_dtype)


# ---- transfer tables (the data the device kernels consume) -------------------
def decode_lut_u8() -> np.ndarray:
    """256-entry table: u8 code -> linear f32, through a3+a4 exactly as the
    reference evaluates them on a u8 frame whose max is > 1."""
    codes = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    return srgb_to_linear(get_normalized_image(codes))[..., 0].reshape(256).astype(np.float32)


def _encode_scalar_fn(dtype):
    def f(x: np.ndarray) -> np.ndarray:
        return encode_u8(np.asarray(x, dtype=dtype), np.uint8)

    return f


def encode_thresholds(dtype=np.float32, encode=None) -> np.ndarray:
    """255 ascending thresholds t_k (k = 1..255): the smallest value of `dtype`
    in [0,1] whose encode is >= k.  Found by bisection over the ordered bit
    patterns of non-negative floats, using the reference arithmetic itself."""
    enc = encode or _encode_scalar_fn(dtype)
    it = np.uint32 if dtype == np.float32 else np.uint64
    one = np.array([1.0], dtype).view(it)[0]
    lo = np.zeros(255, it)  # encode(lo) < k  (invariant; lo=+0.0 encodes to 0)
    hi = np.full(255, one, it)  # encode(hi) >= k (1.0 encodes to 255)
    ks = np.arange(1, 256)
    while np.any(hi - lo > 1):
        mid = lo + (hi - lo) // it(2)
        e = enc(mid.view(dtype)).astype(np.int64)
        ge = e >= ks
        hi = np.where(ge, mid, hi)
        lo = np.where(ge, lo, mid)
    return hi.view(dtype).copy()


def quantize_with_thresholds(x: np.ndarray, thr: np.ndarray) -> np.ndarray:
    """out = #{k: thr[k] <= x}; equals encode_u8 for every x (verified exhaustively
    for f32 by tools/verify_thresholds.py)."""
    flat = np.ascontiguousarray(x).reshape(-1)
    out = np.empty(flat.shape, np.uint8)
    if flat.dtype == np.float64:
        lib().avxref_quantize_f64(_p(flat), _p(out), ctypes.c_size_t(flat.size), _p(np.ascontiguousarray(thr, np.float64)))
    else:
        flat = flat.astype(np.float32, copy=False)
        lib().avxref_quantize_f32(_p(flat), _p(out), ctypes.c_size_t(flat.size), _p(np.ascontiguousarray(thr, np.float32)))
    return out.reshape(x.shape)


# =============================================================================
# UV / spectral path: uv_helpers.py, uv_mappers.py, classic_rgb_to_hsi.py
# =============================================================================
EPS_DEFAULT = 1e-8


def to_float01(x: np.ndarray) -> np.ndarray:
    """uv_helpers.py:15-23."""
    if x.dtype == np.uint8:
        return x.astype(np.float32) / 255.0
    y = x.astype(np.float32)
    if y.max() > 1.001:
        y = np.clip(y / 255.0, 0.0, 1.0)
    return y


def from_float01(img01: np.ndarray, dtype) -> np.ndarray:
    """uv_helpers.py:26-30."""
    if np.issubdtype(dtype, np.integer):
        return np.clip(img01 * 255.0 + 0.5, 0.0, 255.0).astype(dtype)
    return img01.astype(dtype)


def uv_srgb_to_linear(s: np.ndarray) -> np.ndarray:
    """uv_helpers.py:33-37."""
    a = 0.055
    return np.where(s <= 0.04045, s / 12.92, ((s + a) / (1 + a)) ** 2.4).astype(np.float32)


def uv_linear_to_srgb(l: np.ndarray) -> np.ndarray:
    """uv_helpers.py:40-44."""
    a = 0.055
    srgb = np.where(l <= 0.0031308, l * 12.92, (1 + a) * np.power(np.clip(l, 0.0, None), 1 / 2.4) - a)
    return srgb.astype(np.float32)


def safe_norm(x: np.ndarray) -> np.ndarray:
    """uv_helpers.py:47-53."""
    x = x.astype(np.float32)
    mn, mx = float(np.min(x)), float(np.max(x))
    if mx - mn < 1e-9:
        return np.zeros_like(x, dtype=np.float32)
    return (x - mn) / (mx - mn)


def uv_blur_ksize(sigma: float) -> int:
    """uv_helpers.py:72."""
    return int(2 * np.ceil(3 * sigma) + 1)


def gaussian_blur(img: np.ndarray, sigma: float) -> np.ndarray:
    """uv_helpers.py:67-73, cv2 branch (the one a box with OpenCV executes)."""
    if sigma <= 0:
        return img
    k = uv_blur_ksize(sigma)
    return cv_gaussian_blur(img, (k, k), sigma, sigma)


def apply_scatter_and_blue_bias(img_lin: np.ndarray, *, sigma: float, blue_bias: float) -> np.ndarray:
    """uv_helpers.py:100-106."""
    out = img_lin.copy()
    if sigma > 0.15:
        out = gaussian_blur(out, sigma)
    out[..., 2] = np.clip(out[..., 2] + float(blue_bias), 0.0, 1.0)
    return out


def snow_glare_tone_compress(img_lin: np.ndarray, *, strength: float, knee: float = 0.8) -> np.ndarray:
    """uv_helpers.py:111-121."""
    if strength <= 0.0:
        return img_lin
    x = np.clip(img_lin, 0.0, 1.0)
    below = x <= knee
    y = np.empty_like(x)
    y[below] = x[below]
    t = (x[~below] - knee) / (1.0 - knee)
    y[~below] = knee + (1.0 - knee) * (t / (1.0 + strength * t))
    return y


def bandpass_weights(lambdas: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """uv_helpers.py:125-139, including quirk Q7 (uniform 1/B fallback)."""
    wl = lambdas.astype(np.float32)
    w = np.zeros_like(wl, dtype=np.float32)
    mask = (wl >= lo) & (wl <= hi)
    if not np.any(mask):
        return np.ones_like(wl, dtype=np.float32) / float(wl.size)
    x = (wl[mask] - lo) / (hi - lo)
    w[mask] = 0.5 * (1.0 - np.cos(2.0 * np.pi * x))
    s = float(np.sum(w))
    if s > 1e-12:
        w /= s
    else:
        w = np.ones_like(wl, dtype=np.float32) / float(wl.size)
    return w


def integrate_band(hsi: np.ndarray, lambdas: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """uv_helpers.py:142-146."""
    weight = bandpass_weights(lambdas, lo, hi)
    return np.tensordot(hsi, weight, axes=([2], [0])).astype(np.float32)


def integrate_uv(hsi, lambdas, lo, hi):
    """uv_helpers.py:149-152."""
    return safe_norm(integrate_band(hsi, lambdas, lo, hi))


def D65_like(lambdas_nm: np.ndarray) -> np.ndarray:
    """uv_helpers.py:187-192."""
    x = (lambdas_nm - 560.0) / 50.0
    base = np.exp(-0.5 * x**2) + 0.3 * np.exp(-0.5 * ((lambdas_nm - 450.0) / 35.0) ** 2)
    base /= base.mean()
    return base.astype(np.float32)


def von_kries_white_patch(U, B, G, eps: float = EPS_DEFAULT):
    """uv_helpers.py:195-199."""
    return U / max(U.max(), eps), B / max(B.max(), eps), G / max(G.max(), eps)


def von_kries_gray_world(U, B, G, eps: float = EPS_DEFAULT):
    """uv_helpers.py:202-206."""
    return U / max(U.mean(), eps), B / max(B.mean(), eps), G / max(G.mean(), eps)


def classic_rgb_to_hsi_lobes(frame: np.ndarray, wavelengths: np.ndarray) -> np.ndarray:
    """ml/classic_rgb_to_hsi/classic_rgb_to_hsi.py:47-82 (the branch a GPU box runs, F5).

    Quirk Q5 kept: channel 0 drives the 460 nm lobe, channel 2 the 610 nm lobe,
    whatever the caller's channel order.  Evaluated with torch CPU ops so that op
    order and dtypes follow the reference line by line."""
    import torch

    t = torch.as_tensor(np.ascontiguousarray(frame), dtype=torch.float32)
    a = 0.055
    t = torch.where(t <= 0.04045, t / 12.92, ((t + a) / (1.0 + a)) ** 2.4)
    Bc, G, R = t[..., 0], t[..., 1], t[..., 2]
    wl = torch.as_tensor(wavelengths.astype(np.float32)).view(-1, 1, 1)
    cR, cG, cB = 610.0, 545.0, 460.0
    sR, sG, sB = 60.0, 60.0, 55.0
    gR = torch.exp(-0.5 * ((wl - cR) / sR) ** 2)
    gG = torch.exp(-0.5 * ((wl - cG) / sG) ** 2)
    gB = torch.exp(-0.5 * ((wl - cB) / sB) ** 2)
    spec = gR * R.unsqueeze(0) + gG * G.unsqueeze(0) + gB * Bc.unsqueeze(0)
    w1 = wl.squeeze()
    denom = (
        torch.exp(-0.5 * ((w1 - cR) / sR) ** 2)
        + torch.exp(-0.5 * ((w1 - cG) / sG) ** 2)
        + torch.exp(-0.5 * ((w1 - cB) ** 2) / (sB**2))
    ).mean()
    spec = spec / (denom + 1e-8)
    spec = spec.clamp_min(0.0).permute(1, 2, 0).contiguous()
    return spec.numpy().astype(np.float32)


def lobe_tables(wavelengths: np.ndarray) -> Tuple[np.ndarray, float]:
    """(B,3) lobe gains [gB, gG, gR] per band (column j multiplies input channel j)
    and the scalar denominator, as classic_rgb_to_hsi.py:63-79 evaluates them."""
    import torch

    wl = torch.as_tensor(wavelengths.astype(np.float32))
    gR = torch.exp(-0.5 * ((wl - 610.0) / 60.0) ** 2)
    gG = torch.exp(-0.5 * ((wl - 545.0) / 60.0) ** 2)
    gB = torch.exp(-0.5 * ((wl - 460.0) / 55.0) ** 2)
    denom = (gR + gG + torch.exp(-0.5 * ((wl - 460.0) ** 2) / (55.0**2))).mean() + 1e-8
    return torch.stack([gB, gG, gR], dim=1).numpy(), float(denom)


# ---- uv_mappers.py -----------------------------------------------------------
def hsv_to_rgb(hsv: np.ndarray) -> np.ndarray:
    """uv_mappers.py:14-26."""
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    i = np.floor(h * 6.0).astype(np.int32)
    f = h * 6.0 - i
    p = v * (1.0 - s)
    q = v * (1.0 - f * s)
    t = v * (1.0 - (1.0 - f) * s)
    i_mod = i % 6
    conds = [i_mod == 0, i_mod == 1, i_mod == 2, i_mod == 3, i_mod == 4, i_mod == 5]
    r = np.select(conds, [v, q, p, p, t, v], default=0)
    g = np.select(conds, [t, v, v, q, p, p], default=0)
    b = np.select(conds, [p, p, t, v, v, q], default=0)
    return np.stack([r, g, b], axis=2)


def map_falsecolor(U, B, G, eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:29-42."""

    def norm95(x):
        s = np.percentile(x, 95.0)
        return x / max(float(s), eps)

    U_n, B_n, G_n = norm95(U), norm95(B), norm95(G)
    R = 0.85 * U_n + 0.10 * G_n
    Gc = 0.80 * G_n + 0.20 * B_n
    Bl = 0.70 * B_n + 0.40 * U_n
    return np.clip(np.stack([R, Gc, Bl], axis=2), 0.0, 1.0).astype(np.float32)


def map_linear_matrix(U, B, G, M) -> np.ndarray:
    """uv_mappers.py:45-50."""
    H, W = U.shape
    C = np.stack([U, B, G], axis=2).reshape(-1, 3)
    return (C @ M.T).reshape(H, W, 3).astype(np.float32)


def map_opponent(U, B, G, eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:53-64."""
    O1 = G - B
    O2 = B - U
    L = (U + B + G) / 3.0
    angle = np.arctan2(O2, O1)
    hue = (angle + np.pi) / (2 * np.pi)
    radius = np.sqrt(O1 * O1 + O2 * O2)
    sat = radius / (np.percentile(radius, 95.0) + eps)
    val = L / (np.percentile(L, 95.0) + eps)
    hsv = np.stack([hue, np.clip(sat, 0, 1), np.clip(val, 0, 1)], axis=2)
    return hsv_to_rgb(hsv).astype(np.float32)


def _s2l(v):
    a = 0.055
    return np.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4).astype(np.float32)


def map_uv_purple_yellow(U, eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:67-87."""
    if U.ndim == 3 and U.shape[2] == 1:
        U = U[..., 0]
    elif U.ndim != 2:
        raise ValueError(f"U must be HxW or HxWx1, got {U.shape}")
    denom = max(float(np.percentile(U, 99.0)), eps)
    u = (U.astype(np.float32) / denom).clip(0.0, 1.0) ** 0.85
    c0 = _s2l(np.array([128, 0, 150], np.float32) / 255.0)
    c1 = _s2l(np.array([255, 225, 60], np.float32) / 255.0)
    u3 = u[..., None]
    return np.clip((1.0 - u3) * c0 + u3 * c1, 0.0, 1.0).astype(np.float32)


def map_uv_purple_yellow_soft(U, *, u_gamma=0.90, accent_gamma=0.85, accent_strength=0.05, eps=EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:90-132."""
    if U.ndim == 3 and U.shape[2] == 1:
        U = U[..., 0]
    elif U.ndim != 2:
        raise ValueError(f"U must be HxW or HxWx1, got {U.shape}")
    denom = max(float(np.percentile(U, 98.0)), eps)
    u = (U.astype(np.float32) / denom).clip(0.0, 1.0) ** float(u_gamma)
    c0 = _s2l(np.array([176, 124, 232], np.float32) / 255.0)
    c1 = _s2l(np.array([255, 211, 138], np.float32) / 255.0)
    u3 = u[..., None]
    rgb_lin = (1.0 - u3) * c0 + u3 * c1
    gray = np.array([0.5, 0.5, 0.5], np.float32)
    purple_dir = c0 - gray
    a = float(accent_strength)
    if a > 0:
        w = (u ** float(accent_gamma))[..., None]
        rgb_lin = rgb_lin + a * w * purple_dir
    Y = (0.2126 * rgb_lin[..., 0] + 0.7152 * rgb_lin[..., 1] + 0.0722 * rgb_lin[..., 2]) + eps
    Y_target = np.clip(0.22 + 0.55 * u, 0.0, 1.0)
    gain = np.clip((Y_target / Y)[..., None], 0.6, 1.6)
    rgb_lin = rgb_lin * gain
    rgb_lin = rgb_lin / (1.0 + 0.6 * rgb_lin)
    return np.clip(rgb_lin, 0.0, 1.0).astype(np.float32)


def map_falsecolor_uv_mixed(U, B, G, alpha: float = 0.35) -> np.ndarray:
    """uv_mappers.py:135-144."""
    base = map_falsecolor(U, B, G)
    uv_tint = map_uv_purple_yellow_soft(U)
    alpha = float(np.clip(alpha, 0.0, 1.0))
    mixed = (1.0 - alpha) * base + alpha * uv_tint
    p99 = float(np.percentile(mixed, 99.0))
    if p99 > EPS_DEFAULT:
        mixed = mixed / max(1.0, p99)
    return np.clip(mixed.astype(np.float32), 0.0, 1.0)


# ---- animals/honeybee.py -------------------------------------------------------
def honeybee_cone_curves(lambdas: np.ndarray):
    """animals/honeybee.py:179-192 + the sum-normalisation of :89-93."""

    def log_normal(lam, peak, sigma):
        return np.exp(-0.5 * ((lam - peak) / sigma) ** 2)

    curves = [
        log_normal(lambdas, 350.0, 25.0).astype(np.float32),
        log_normal(lambdas, 440.0, 30.0).astype(np.float32),
        log_normal(lambdas, 540.0, 35.0).astype(np.float32),
    ]
    for v in curves:
        s = v.sum()
        if s > 0:
            v /= s
    return curves


def honeybee_catches(hsi: np.ndarray, lambdas: np.ndarray, reflectance: bool = True):
    """animals/honeybee.py:125-135: radiance = hsi * D65_like, then 3 tensordots."""
    UVc, Bc, Gc = honeybee_cone_curves(lambdas)
    radiance = hsi * D65_like(lambdas).astype(hsi.dtype)[None, None, :] if reflectance else hsi
    U = np.tensordot(radiance, UVc, axes=([2], [0]))
    B = np.tensordot(radiance, Bc, axes=([2], [0]))
    G = np.tensordot(radiance, Gc, axes=([2], [0]))
    return U, B, G


def honeybee_tail(U, B, G, orig_dtype=np.uint8, *, adaptation="white_patch", mapping_mode="opponent", blur_sigma_px=0.2, custom_matrix=None):
    """animals/honeybee.py:137-175 (adaptation -> blur -> map -> clip -> encode)."""
    eps = EPS_DEFAULT
    if adaptation == "white_patch":
        U, B, G = von_kries_white_patch(U, B, G, eps=eps)
    elif adaptation == "gray_world":
        U, B, G = von_kries_gray_world(U, B, G, eps=eps)
    if blur_sigma_px > 0:
        U, B, G = gaussian_blur(U, blur_sigma_px), gaussian_blur(B, blur_sigma_px), gaussian_blur(G, blur_sigma_px)
    if mapping_mode == "falsecolor":
        rgb_lin = map_falsecolor(U, B, G, eps=eps)
    elif mapping_mode == "custom_matrix":
        rgb_lin = map_linear_matrix(U, B, G, custom_matrix)
    elif mapping_mode == "opponent":
        rgb_lin = map_opponent(U, B, G, eps=eps)
    elif mapping_mode == "uv_purple_yellow":
        rgb_lin = map_uv_purple_yellow_soft(U)
    elif mapping_mode == "falsecolor_uv_mixed":
        rgb_lin = map_falsecolor_uv_mixed(U, B, G, alpha=0.45)
    else:
        raise ValueError(f"Unknown mapping_mode: {mapping_mode}")
    rgb_lin = np.clip(rgb_lin, 0.0, 1.0)
    out_srgb = uv_linear_to_srgb(rgb_lin)
    if np.issubdtype(orig_dtype, np.integer):
        return (out_srgb * 255.0 + 0.5).astype(orig_dtype), rgb_lin
    return out_srgb.astype(orig_dtype), rgb_lin


def honeybee_visualize(image: np.ndarray, *, hsi: Optional[np.ndarray] = None, lambdas=None, hsi_downsample: bool = False,
                       hsi_scale: float = 0.1, **kw):
    """animals/honeybee.py:99-175 with the as-coded HSI source (analytic lobes, F3/F5; at reduced size when
    `hsi_downsample`, :109-117), or with a caller-supplied cube (the MST++ route of the north star)."""
    assert isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] == 3
    lambdas = np.linspace(400.0, 700.0, 31, dtype=np.float32) if lambdas is None else np.asarray(lambdas, np.float32)
    img01 = to_float01(image)
    if hsi is None:
        if hsi_downsample and 0.05 <= hsi_scale < 1.0:
            hsi = classic_rgb_to_hsi_scaled(img01, wavelengths=lambdas, scale=hsi_scale)
        else:
            hsi = classic_rgb_to_hsi_lobes(img01, lambdas)
    U, B, G = honeybee_catches(hsi, lambdas)
    out, _ = honeybee_tail(U, B, G, image.dtype, **kw)
    return image, out


# ---- animals/mantis_shrimp.py: spectral stack + barcode (a17, a22) -------------
MANTIS_BANDS = (
    (320.0, 360.0), (360.0, 400.0), (400.0, 430.0), (430.0, 460.0), (460.0, 490.0),
    (490.0, 520.0), (520.0, 550.0), (550.0, 580.0), (580.0, 610.0), (610.0, 680.0),
)  # animals/mantis_shrimp.py:49-60


def mantis_band_stack(hsi: np.ndarray, lambdas: np.ndarray, bands: Sequence[Tuple[float, float]] = MANTIS_BANDS) -> np.ndarray:
    """animals/mantis_shrimp.py:167-172: N x safe_norm(integrate_band) stacked (H,W,N)."""
    return np.stack([safe_norm(integrate_band(hsi, lambdas, lo, hi)) for lo, hi in bands], axis=2).astype(np.float32)


def mantis_hue_lut(N: int) -> np.ndarray:
    """animals/mantis_shrimp.py:175-197."""
    idx = np.arange(N, dtype=np.float32)
    h = (idx / max(N, 1)).astype(np.float32)
    s = np.full_like(h, 0.95, np.float32)
    v = np.ones_like(h, np.float32)
    i = np.floor(h * 6.0).astype(np.int32)
    f = h * 6.0 - i
    p = v * (1.0 - s)
    q = v * (1.0 - f * s)
    t = v * (1.0 - (1.0 - f) * s)
    i = i % 6
    conds = [i == 0, i == 1, i == 2, i == 3, i == 4, i == 5]
    return np.stack(
        [
            np.select(conds, [v, q, p, p, t, v], default=v),
            np.select(conds, [t, v, v, q, p, p], default=v),
            np.select(conds, [p, p, t, v, v, q], default=v),
        ],
        axis=-1,
    ).astype(np.float32)


def mantis_barcode(S: np.ndarray, *, winner_take_most: float = 0.35, barcode_saturation: float = 0.40, _jit=None):
    """animals/mantis_shrimp.py:199-211 -> (barcode_rgb (H,W,3), S_norm (H,W,N))."""
    N = S.shape[2]
    lut = mantis_hue_lut(N)
    p95 = np.percentile(S, 95.0)
    S_norm = S / ((p95 if _jit is None else _jit(p95)) + 1e-8)
    S_norm = np.clip(S_norm, 0.0, 1.0)
    max_idx = np.argmax(S_norm, axis=2)
    weights = S_norm / (np.sum(S_norm, axis=2, keepdims=True) + 1e-8)
    soft_rgb = weights @ lut
    hard_rgb = lut[max_idx]
    barcode_rgb = (1.0 - winner_take_most) * soft_rgb + winner_take_most * hard_rgb
    Yb = (0.2126 * barcode_rgb[..., 0] + 0.7152 * barcode_rgb[..., 1] + 0.0722 * barcode_rgb[..., 2])[..., None]
    barcode_rgb = np.clip(Yb + (barcode_rgb - Yb) * (1.0 + barcode_saturation), 0.0, 1.0)
    return barcode_rgb, S_norm


# ---- renderers/video.py:198-245 (split compose without labels) -----------------
def relative_jitter(seed: int, amp: float = 2.0 ** -20):
    """The _jit callable of mantis_visualize: x -> x * (1 + amp * u), u uniform in [-1, 1] per element (tests only)."""
    rng = np.random.default_rng(seed)

    def jit(x):
        if isinstance(x, np.ndarray):
            return x * (1 + x.dtype.type(amp) * rng.uniform(-1.0, 1.0, x.shape).astype(x.dtype))
        return type(x)(x * (1.0 + amp * rng.uniform(-1.0, 1.0)))

    return jit


def make_split_frame_nolabel(original: np.ndarray, modified: np.ndarray, draw_seam: bool = True) -> np.ndarray:
    """renderers/video.py:234-239: left half original, right half modified, 1-px white seam.
    (Hershey-font labels of :242-244 are cv2 drawing code: out of scope)"""
    H, W = original.shape[:2]
    if modified.shape[:2] != (H, W):  # :228-231
        modified = cv_resize(np.ascontiguousarray(modified), (W, H), INTER_AREA)
    out = original.copy()
    mid = W // 2
    out[:, mid:, :] = modified[:, mid:, :]
    if draw_seam:
        out[:, mid : mid + 1, :] = 255
    return out


def draw_label_pixels(img: np.ndarray, box, segs: np.ndarray, outline_thickness: float, text_thickness: float) -> np.ndarray:
    """The pixel arithmetic of csrc/labels.hip::k_draw_label in NumPy float32, operation for operation (in place on a
    uint8 HxWx3 frame; returns it): 60 % black box (renderers/video.py:189-191: saturate_cast(0.4 * in) inside the
    inclusive box), then black strokes at outline_thickness and white strokes at text_thickness with analytic coverage
    clamp(t/2 + 0.5 - distance, 0, 1).  box = (x0, y0, x1, y1); segs = (n, 6) float32 {ax, ay, dx, dy, 1/len^2, 0}.
    Geometry and segments come from the caller (the reference's layout rules are checked separately, with literal numbers,
    in tests/test_labels.py).  OpenCV's LINE_AA rasteriser is NOT restated: label pixels are parity-unpinned."""
    H, W, _ = img.shape
    x0, y0, x1, y1 = (int(v) for v in box)
    ho, ht = np.float32(0.5 * outline_thickness), np.float32(0.5 * text_thickness)
    grow = int(float(ho) + 2.0)
    rx0, ry0, rx1, ry1 = max(x0 - grow, 0), max(y0 - grow, 0), min(x1 + grow, W - 1), min(y1 + grow, H - 1)
    if rx1 < rx0 or ry1 < ry0:
        return img
    ys, xs = np.mgrid[ry0 : ry1 + 1, rx0 : rx1 + 1]
    px, py = xs.astype(np.float32), ys.astype(np.float32)
    d2 = np.full(px.shape, np.float32(3.0e38), np.float32)
    for g in np.asarray(segs, np.float32).reshape(-1, 6):
        qx, qy = px - g[0], py - g[1]
        t = np.clip((qx * g[2] + qy * g[3]) * g[4], np.float32(0), np.float32(1))
        ex, ey = qx - t * g[2], qy - t * g[3]
        d2 = np.minimum(d2, ex * ex + ey * ey)
    d = np.sqrt(d2)
    co = np.clip(ho + np.float32(0.5) - d, np.float32(0), np.float32(1))
    ct = np.clip(ht + np.float32(0.5) - d, np.float32(0), np.float32(1))
    inbox = (xs >= x0) & (xs <= x1) & (ys >= y0) & (ys <= y1)
    v = img[ry0 : ry1 + 1, rx0 : rx1 + 1, :].astype(np.float32)
    v = np.where(inbox[..., None], np.rint(v * np.float32(0.4)), v)
    v = np.rint(v - v * co[..., None])
    v = np.rint(v + (np.float32(255.0) - v) * ct[..., None])
    img[ry0 : ry1 + 1, rx0 : rx1 + 1, :] = np.clip(v, 0, 255).astype(np.uint8)
    return img


# =============================================================================
# OpenCV resize / remap / Sobel restatements (oracle/cvref.cpp): PARITY UNPINNED
# =============================================================================
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3  # cv2 constants


def cv_resize(img: np.ndarray, dsize: Tuple[int, int], interpolation: int = INTER_LINEAR) -> np.ndarray:
    """cv2.resize(img, (W_out, H_out), interpolation=...) for HxW[xC] float32 (NEAREST/LINEAR/CUBIC/AREA) or uint8 (LINEAR)."""
    Wd, Hd = int(dsize[0]), int(dsize[1])
    squeeze = img.ndim == 2
    a = np.ascontiguousarray(img[..., None] if squeeze else img)
    H, W, C = a.shape
    if a.dtype == np.uint8 and interpolation == INTER_AREA and not (Wd > W or Hd > H):
        # cv::resize(8-bit, INTER_AREA), shrinking: resizeAreaFast_<uchar, int> / resizeArea_<uchar, float> run the float path's
        # arithmetic on integer samples (block sums are exact in float32 below 2^24) and store saturate_cast<uchar> (cvRound:
        # half to even); the 8-bit 2x2 special case (ResizeAreaFastVec) rounds (sum + 2) >> 2 instead.  PARITY UNPINNED (no OpenCV here).
        if W == 2 * Wd and H == 2 * Hd:
            s4 = a[0::2, 0::2].astype(np.int32) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]
            out = ((s4 + 2) >> 2).astype(np.uint8)
        else:
            f = np.empty((Hd, Wd, C), np.float32)
            lib().cvref_resize_area_f32(_p(a.astype(np.float32)), H, W, C, _p(f), Hd, Wd)
            out = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    elif a.dtype == np.uint8:
        assert interpolation in (INTER_LINEAR, INTER_AREA), "uint8: INTER_LINEAR and INTER_AREA are restated"
        out = np.empty((Hd, Wd, C), np.uint8)
        lib().cvref_resize_linear_u8(_p(a), H, W, C, _p(out), Hd, Wd)  # INTER_AREA when enlarging behaves like INTER_LINEAR
    else:
        a = a.astype(np.float32, copy=False)
        out = np.empty((Hd, Wd, C), np.float32)
        fn = {INTER_NEAREST: "cvref_resize_nearest_f32", INTER_LINEAR: "cvref_resize_linear_f32", INTER_CUBIC: "cvref_resize_cubic_f32",
              INTER_AREA: "cvref_resize_area_f32"}[interpolation]
        if interpolation == INTER_AREA and (Wd > W or Hd > H):
            fn = "cvref_resize_linear_f32"  # cv::resize: INTER_AREA when enlarging behaves like INTER_LINEAR
        getattr(lib(), fn)(_p(a), H, W, C, _p(out), Hd, Wd)
    return out[..., 0] if squeeze else out


def cv_remap_linear(img: np.ndarray, mapx: np.ndarray, mapy: np.ndarray, border_value: float = 0.0) -> np.ndarray:
    """cv2.remap(img, mapx, mapy, INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=...) for float32 HxWxC."""
    a = np.ascontiguousarray(img, np.float32)
    H, W, C = a.shape
    mx, my = np.ascontiguousarray(mapx, np.float32), np.ascontiguousarray(mapy, np.float32)
    Hd, Wd = mx.shape
    out = np.empty((Hd, Wd, C), np.float32)
    lib().cvref_remap_linear_f32(_p(a), H, W, C, _p(mx), _p(my), _p(out), Hd, Wd, ctypes.c_float(border_value))
    return out


def cv_sobel3(img: np.ndarray, dx: int, dy: int) -> np.ndarray:
    """cv2.Sobel(img, CV_32F, dx, dy, ksize=3, borderType=BORDER_REFLECT101) for (dx,dy) in {(1,0),(0,1)}."""
    assert (dx, dy) in ((1, 0), (0, 1))
    a = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(a)
    lib().cvref_sobel3_f32(_p(a), a.shape[0], a.shape[1], int(dx), _p(out))
    return out


# ---- geometry helpers of the reference on top of them ---------------------------------------------
def resize_preserve_range(x: np.ndarray, out_hw: Tuple[int, int], *, interp: int) -> np.ndarray:
    """uv_helpers.py:57-64."""
    H_out, W_out = out_hw
    was_float = np.issubdtype(x.dtype, np.floating)
    y = cv_resize(x.astype(np.float32, copy=False), (W_out, H_out), interp)
    return y.astype(x.dtype, copy=False) if not was_float else y


def panorama_warp(img_lin: np.ndarray, *, scale_x: float) -> np.ndarray:
    """uv_helpers.py:84-99 (cv2 branch)."""
    if abs(scale_x - 1.0) < 1e-3:
        return img_lin
    H, W = img_lin.shape[:2]
    newW = max(2, int(round(W * scale_x)))
    widened = cv_resize(img_lin, (newW, H), INTER_CUBIC)
    if newW == W:
        return widened
    start = (newW - W) // 2
    return widened[:, start : start + W, :]


def classic_rgb_to_hsi_scaled(rgb01: np.ndarray, *, wavelengths: np.ndarray, scale: float) -> np.ndarray:
    """uv_helpers.py:155-183: INTER_AREA down -> analytic lobes -> INTER_LINEAR up of the B-band cube."""
    assert 0.0 < scale <= 1.0, "scale must be (0,1]."
    H, W = rgb01.shape[:2]
    h_small, w_small = max(1, int(round(H * scale))), max(1, int(round(W * scale)))
    rgb_small = resize_preserve_range(rgb01, (h_small, w_small), interp=INTER_AREA)
    hsi_small = classic_rgb_to_hsi_lobes(rgb_small, wavelengths.astype(np.float32))
    return resize_preserve_range(hsi_small, (H, W), interp=INTER_LINEAR)


def center_zoom(image: np.ndarray, scale: float) -> np.ndarray:
    """animals/cat_widevision_utils.py:11-29 (cv2 branch)."""
    if scale <= 1.0:
        return image
    H, W = image.shape[:2]
    cw, ch = max(1, int(round(W / scale))), max(1, int(round(H / scale)))
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    return cv_resize(image[y0 : y0 + ch, x0 : x0 + cw], (W, H), INTER_LINEAR)


def zoom_scale_from_cat_ratio(*, camera_hfov_deg: float, cat_per_eye_half_fov_deg: float, cat_to_human_ratio: float) -> float:
    """animals/cat_widevision_utils.py:31-44."""
    import math

    phi = float(cat_per_eye_half_fov_deg)
    eff = min(float(camera_hfov_deg), 2.0 * phi)
    ratio = max(1.01, float(cat_to_human_ratio))
    cam = math.tan(math.radians(camera_hfov_deg) * 0.5)
    hum = math.tan(math.radians(eff / ratio) * 0.5)
    return float(cam / max(hum, 1e-6))


def binocular_warp_maps(H_in: int, W_in: int, out_w: int, out_h: int, fov_in_deg: float, per_eye_half_fov_deg: float, overlap_deg: float):
    """Maps and blend weights of animals/cat_widevision_utils.py:66-96 -> (xL, xR, ymap, wL, wR) float32."""
    phi = np.deg2rad(per_eye_half_fov_deg)
    psi = np.deg2rad(fov_in_deg * 0.5)
    O = np.deg2rad(overlap_deg)
    alpha = max(0.0, phi - 0.5 * O)
    u = np.linspace(-1.0, 1.0, out_w, dtype=np.float32)
    v = np.linspace(0.0, float(out_h - 1), out_h, dtype=np.float32)
    U, _ = np.meshgrid(u, v)
    thetaL = U * phi
    thetaR = U * phi
    gammaL, gammaR = thetaL - alpha, thetaR + alpha

    def yaw_to_xsrc(gamma):
        return ((gamma / psi) * (W_in * 0.5) + (W_in * 0.5)).astype(np.float32)

    xL, xR = yaw_to_xsrc(gammaL), yaw_to_xsrc(gammaR)
    ymap = np.repeat(np.linspace(0, H_in - 1, out_h, dtype=np.float32)[:, None], out_w, axis=1)
    validL = (np.abs(gammaL) <= psi).astype(np.float32)
    validR = (np.abs(gammaR) <= psi).astype(np.float32)
    wL = (np.cos(0.5 * np.pi * (thetaL / phi)) ** 2).astype(np.float32) * validL
    wR = (np.cos(0.5 * np.pi * (thetaR / phi)) ** 2).astype(np.float32) * validR
    return xL, xR, ymap, wL, wR


def animal_fov_binocular_warp(img01: np.ndarray, *, fov_in_deg, per_eye_half_fov_deg, overlap_deg, out_size=None, border_value=0.0):
    """animals/cat_widevision_utils.py:46-99."""
    H_in, W_in, _ = img01.shape
    out_w, out_h = (W_in, H_in) if out_size is None else out_size
    xL, xR, ymap, wL, wR = binocular_warp_maps(H_in, W_in, out_w, out_h, fov_in_deg, per_eye_half_fov_deg, overlap_deg)
    left = cv_remap_linear(img01, xL, ymap, border_value)
    right = cv_remap_linear(img01, xR, ymap, border_value)
    wsum = (wL + wR + 1e-8)[..., None]
    out = (left * wL[..., None] + right * wR[..., None]) / wsum
    return np.clip(out, 0.0, 1.0).astype(np.float32)


def cat_visualize(image: np.ndarray, *, enable_fov_warp: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """animals/cat.py:23-28,73-114 (Tina-animals side of the unresolved merge, quirk Q8): human-zoom baseline +
    wide-FOV cat view with the float64 colour tail."""
    assert isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] == 3
    orig_dtype = image.dtype
    H, W = image.shape[:2]
    scale = zoom_scale_from_cat_ratio(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, cat_to_human_ratio=1.30)
    human_zoomed = center_zoom(image, scale=scale)
    cat01 = get_normalized_image(image)
    if enable_fov_warp:
        cat01 = animal_fov_binocular_warp(cat01.astype(np.float32), fov_in_deg=100.0, per_eye_half_fov_deg=105.0, overlap_deg=40.0,
                                          out_size=(W, H), border_value=0.0)
    lin = srgb_to_linear(cat01)
    rgb = dichromat_color_stage(DICHROMATS["cat"], lin)
    rgb = apply_acuity_blur(rgb, sigma=1.0)
    cat_srgb = np.clip(linear_to_srgb(np.clip(rgb, 0.0, 1.0)), 0.0, 1.0)
    if np.issubdtype(orig_dtype, np.integer):
        human_out = human_zoomed if np.issubdtype(human_zoomed.dtype, np.integer) else (np.clip(human_zoomed, 0, 1) * 255.0 + 0.5).astype(orig_dtype)
        cat_out = (cat_srgb * 255.0 + 0.5).astype(orig_dtype)
    else:
        human_out, cat_out = human_zoomed.astype(orig_dtype), cat_srgb.astype(orig_dtype)
    return human_out, cat_out


def mantis_visualize(image: np.ndarray, *, hsi_scale: float = 0.25, panorama_scale: float = 1.12, lambdas=None,
                     bands=MANTIS_BANDS, red_kill=0.18, haze_strength=0.03, haze_tint=(0.92, 0.98, 1.00), pre_soft_sigma=0.25,
                     unsharp_sigma=1.0, unsharp_amount=0.32, evec_angle_deg=30.0, pol_linear_strength=0.55, pol_linear_gamma=1.2,
                     pol_circular_strength=0.35, orientation_mix=0.5, barcode_saturation=0.40, barcode_opacity=0.55,
                     winner_take_most=0.35, scan_row_freq=26.0, scan_row_gain=0.08, scan_soften=0.8, periph_blur_sigma=0.7,
                     periph_radius=0.80, periph_softness=7.0, _jit=None):
    """animals/mantis_shrimp.py:143-279 (cv2 branches), defaults of :42-86.
    _jit (tests only, tests/_sensitivity.py): a callable applied to the band stack, the P95, the gradient angle and the blurred
    planes -- relative float32-level jitter that shows where this arithmetic is unstable (argmax ties, arctan2 of ~0 gradients)."""
    jit = (lambda a: a) if _jit is None else _jit
    assert isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] == 3
    dtype = image.dtype
    lambdas = np.linspace(300.0, 700.0, 81, dtype=np.float32) if lambdas is None else np.asarray(lambdas, np.float32)
    haze_tint = np.array(haze_tint, dtype=np.float32)
    evec_angle = np.deg2rad(float(evec_angle_deg))
    orientation_mix = float(np.clip(orientation_mix, 0.0, 1.0))
    barcode_opacity = float(np.clip(barcode_opacity, 0.0, 1.0))
    winner_take_most = float(np.clip(winner_take_most, 0.0, 1.0))
    img_lin = uv_srgb_to_linear(to_float01(image))                                                     # :148-149
    baseline_lin = panorama_warp(img_lin, scale_x=panorama_scale) if panorama_scale != 1.0 else img_lin  # :152
    baseline_out = from_float01(uv_linear_to_srgb(np.clip(baseline_lin, 0, 1)), dtype)                 # :153
    if 0.0 < hsi_scale < 1.0:                                                                          # :156-163
        hsi = classic_rgb_to_hsi_scaled(baseline_lin, wavelengths=lambdas, scale=hsi_scale)
    else:
        hsi = classic_rgb_to_hsi_lobes(baseline_lin, lambdas)
    H, W = baseline_lin.shape[:2]
    S = jit(mantis_band_stack(hsi, lambdas, bands))                                                    # :167-172
    barcode_rgb, S_norm = mantis_barcode(S, winner_take_most=winner_take_most, barcode_saturation=barcode_saturation, _jit=_jit)  # :175-211
    render = baseline_lin.copy()                                                                       # :214-220
    render[..., 0] = np.clip(render[..., 0] * (1.0 - red_kill), 0.0, 1.0)
    if haze_strength > 0.0:
        a = float(np.clip(haze_strength, 0.0, 1.0))
        render = (1.0 - a) * render + a * haze_tint[None, None, :]
    if pre_soft_sigma > 0.0:
        render = jit(gaussian_blur(render, pre_soft_sigma))
    broad = np.mean(S_norm, axis=2).astype(np.float32)                                                 # :224-226
    gx, gy = jit(cv_sobel3(broad, 1, 0)), jit(cv_sobel3(broad, 0, 1))
    theta = jit(np.arctan2(gy, gx).astype(np.float32))
    cos2_local, sin2_local = np.cos(2.0 * theta), np.sin(2.0 * theta)                                  # :229-242
    cos2_global, sin2_global = float(np.cos(2.0 * evec_angle)), float(np.sin(2.0 * evec_angle))
    mix = orientation_mix
    cos2_mix = (1.0 - mix) * cos2_global + mix * cos2_local
    sin2_mix = (1.0 - mix) * sin2_global + mix * sin2_local
    align01 = np.clip(0.5 * (cos2_mix + 1.0), 0.0, 1.0) ** pol_linear_gamma
    align_circ = np.clip(0.5 * (sin2_mix + 1.0), 0.0, 1.0)
    pol_gain = 1.0 + pol_linear_strength * align01 + pol_circular_strength * align_circ
    if unsharp_sigma > 0.0 and unsharp_amount > 0.0:                                                   # :244-247
        blur = jit(gaussian_blur(render, unsharp_sigma))
        high = np.clip(render - blur, -1.0, 1.0)
        render = np.clip(render + (unsharp_amount * pol_gain[..., None]) * high, 0.0, 1.0)
    render = np.clip((1.0 - barcode_opacity) * render + barcode_opacity * barcode_rgb, 0.0, 1.0)       # :250
    if scan_row_gain != 0.0:                                                                           # :253-265
        y = np.linspace(0.0, 1.0, H, dtype=np.float32)[:, None]
        rows = 0.5 + 0.5 * np.sin(2.0 * np.pi * scan_row_freq * y)
        rows = rows * np.ones((1, W), dtype=np.float32)
        if scan_soften > 0.0:
            rows = gaussian_blur(rows, scan_soften)
        row_gain = 1.0 + scan_row_gain * (rows - 0.5)
        render = np.clip(render * row_gain[..., None], 0.0, 1.0)
    if periph_blur_sigma > 0.0:                                                                        # :268-275
        periph = jit(gaussian_blur(render, periph_blur_sigma))
        yy = (np.linspace(-1.0, 1.0, H, dtype=np.float32))[:, None]
        xx = (np.linspace(-1.0, 1.0, W, dtype=np.float32))[None, :]
        r = np.sqrt(xx * xx + yy * yy)
        t = 1.0 / (1.0 + np.exp(-periph_softness * (r - periph_radius)))
        t = t[..., None]
        render = (1.0 - t) * render + t * periph
    out = from_float01(uv_linear_to_srgb(np.clip(render, 0.0, 1.0)), dtype)                            # :278
    return baseline_out, out
