"""The reference's animals/cat_widevision_utils.py by name -- NumPy in / NumPy out, everything O(pixels) on the device.

`Cat.visualize` does not come through here (its uint8 route is csrc/geom.hip::avx_binocular_warp_u8 inside the fused frame, its
float route a recorded plane program, animals/_dichromats.py); this is the step-wise surface for code written against the
reference (`from animals.cat_widevision_utils import center_zoom, animal_fov_binocular_warp`, cat.py:10-14)."""
from __future__ import annotations

import numpy as np

from ..geometry import binocular_warp_tables, center_zoom, zoom_scale_from_cat_ratio  # noqa: F401  (cat_widevision_utils.py:11-44)


def animal_fov_binocular_warp(img_srgb_01: np.ndarray, *, fov_in_deg: float, per_eye_half_fov_deg: float, overlap_deg: float,
                              out_size: tuple | None = None, border_mode: int = 0, border_value: float | int = 0) -> np.ndarray:
    """cat_widevision_utils.py:46-99: two cv2.remap(INTER_LINEAR, BORDER_CONSTANT) views (left / right eye axes) of a float [0, 1]
    frame, blended with cos^2 weights over the valid yaw range; float32 HxWx3 in [0, 1].  The maps depend on the column (x) and on
    the row (y) only, so they travel as two W-vectors and one H-vector.  Same-size output and constant borders only (what cat.py uses)."""
    from ..planevm import run_planes

    assert img_srgb_01.ndim == 3 and img_srgb_01.shape[2] == 3
    H, W, _ = img_srgb_01.shape
    out_w, out_h = (W, H) if out_size is None else out_size
    if (out_w, out_h) != (W, H):
        raise NotImplementedError("animal_fov_binocular_warp: out_size other than the input size (the reference's cat.py never asks for one)")
    if border_mode != 0:
        raise NotImplementedError("animal_fov_binocular_warp: BORDER_CONSTANT only")
    xL, xR, ymap, wL, wR = binocular_warp_tables(H, W, W, H, fov_in_deg, per_eye_half_fov_deg, overlap_deg)
    planes = [np.ascontiguousarray(img_srgb_01[..., c], dtype=np.float32) for c in range(3)]
    bv = float(border_value)

    def build(be, x):
        my = be.row(ymap)
        left = be.remap(x, be.col(xL), my, bv)
        right = be.remap(x, be.col(xR), my, bv)
        cwl, cwr = be.col(wL), be.col(wR)
        wsum = cwl + cwr + 1e-8
        return [be.clip01((left[c] * cwl + right[c] * cwr) / wsum) for c in range(3)]

    return np.stack(run_planes(planes, build), axis=-1)
