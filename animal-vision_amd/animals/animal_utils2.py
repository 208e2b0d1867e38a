"""The reference's animals/animal_utils2.py by name (its near-copy of cat_widevision_utils.py plus `human_zoom_and_cat_view`) --
NumPy in / NumPy out, everything O(pixels) on the device."""
from __future__ import annotations

import numpy as np

from .cat_widevision_utils import animal_fov_binocular_warp, center_zoom, zoom_scale_from_cat_ratio  # noqa: F401  (animal_utils2.py:12-105)


def _to_float01(x: np.ndarray) -> np.ndarray:
    """animal_utils2.py:107-114: integers -> / 255; floats -> / 255 only when the maximum exceeds 1.001; clip to [0, 1]; float32."""
    from ..planevm import run_planes

    a = np.ascontiguousarray(x, dtype=np.float32)
    flat = a.reshape(1, -1) if a.ndim < 2 else a.reshape(a.shape[0], -1)
    if np.issubdtype(np.asarray(x).dtype, np.integer):
        build = lambda be, v: [be.clip01(v[0] / 255.0)]  # noqa: E731
    else:
        def build(be, v):
            mx = be.max(v[0])
            return [be.clip01(be.where(mx > 1.001, v[0] / 255.0, v[0]))]
    return run_planes([flat], build)[0].reshape(a.shape)


def _from_float01(x: np.ndarray, dtype) -> np.ndarray:
    """animal_utils2.py:116-121: clip to [0, 1]; integer dtypes get (x * 255 + 0.5) truncated."""
    from ..planevm import run_planes

    a = np.ascontiguousarray(x, dtype=np.float32)
    flat = a.reshape(1, -1) if a.ndim < 2 else a.reshape(a.shape[0], -1)
    if np.issubdtype(dtype, np.integer):
        return run_planes([flat], lambda be, v: [be.clip01(v[0]) * 255.0 + 0.5])[0].reshape(a.shape).astype(dtype)
    return run_planes([flat], lambda be, v: [be.clip01(v[0])])[0].reshape(a.shape).astype(dtype)


def human_zoom_and_cat_view(image: np.ndarray, *, camera_hfov_deg: float, cat_per_eye_half_fov_deg: float, binocular_overlap_deg: float,
                            cat_to_human_ratio: float = 1.30, zoom_scale: float | None = None):
    """animal_utils2.py:123-165 -> (centre-zoomed "human" view, binocular wide-FOV "cat" view of the ORIGINAL frame), both in the input's dtype."""
    orig_dtype = image.dtype
    scale = (zoom_scale_from_cat_ratio(camera_hfov_deg=camera_hfov_deg, cat_per_eye_half_fov_deg=cat_per_eye_half_fov_deg, cat_to_human_ratio=cat_to_human_ratio)
             if zoom_scale is None else float(zoom_scale))
    zoomed = center_zoom(image, scale=scale)
    cat01 = animal_fov_binocular_warp(_to_float01(image), fov_in_deg=camera_hfov_deg, per_eye_half_fov_deg=cat_per_eye_half_fov_deg,
                                      overlap_deg=binocular_overlap_deg, out_size=(image.shape[1], image.shape[0]), border_mode=0, border_value=0.0)
    return _from_float01(_to_float01(zoomed), orig_dtype), _from_float01(cat01, orig_dtype)
