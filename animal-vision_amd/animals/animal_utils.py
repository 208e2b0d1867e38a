"""Host-side mirror of the reference's animals/animal_utils.py for the names species modules use.

Per-pixel work lives in the fused device kernel (dichromat.DichromatOp); what remains here is the
O(1) host logic with the reference's names and argument meaning."""
from __future__ import annotations

import numpy as np

from ..dichromat import collapse_LMS_matrix, cv_auto_ksize, gaussian_taps, s_cone_row_gain  # noqa: F401


def check_input_image(image: np.ndarray) -> bool:
    """animals/animal_utils.py:21-39: ndarray, HxWx3, numeric."""
    if not isinstance(image, np.ndarray):
        return False
    if image.ndim != 3 or image.shape[2] != 3:
        return False
    return bool(np.issubdtype(image.dtype, np.number))


def apply_acuity_blur(image: np.ndarray, sigma: float = 1.5) -> np.ndarray:
    """animals/animal_utils.py:121-145 on the device: OpenCV GaussianBlur(img,(0,0),sigma) semantics
    for a float32 HxWx3 image (csrc/blur.hip)."""
    from ..ops import gaussian_blur_f32

    if image.ndim != 3 or image.shape[2] != 3:
        raise ValueError("Expected HxWx3 image")
    k = cv_auto_ksize(sigma)
    return gaussian_blur_f32(image, k, sigma).astype(image.dtype, copy=False)
