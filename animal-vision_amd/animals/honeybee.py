"""animals/honeybee.py of the reference, on the device.  Same constructor keywords and defaults
(honeybee.py:47-66), same visualize contract: (baseline, bee_render), baseline IS the input object."""
from typing import Callable, Literal, Optional, Tuple

import numpy as np

from ..uv import D65_like, EPS_DEFAULT, HoneybeeOp, honeybee_cone_curves
from .animal import Animal


class HoneyBee(Animal):
    def __init__(
        self,
        onnx_path: str = "./ml/MST_plus_plus/export/mst_plus_plus.onnx",
        hsi_band_centers_nm: Optional[np.ndarray] = None,
        illuminant: Optional[Callable[[np.ndarray], np.ndarray]] = None,
        adaptation: Optional[Literal["white_patch", "gray_world"]] = "white_patch",
        mapping_mode: Literal["falsecolor", "custom_matrix", "opponent", "uv_purple_yellow", "falsecolor_uv_mixed"] = "opponent",
        custom_matrix: Optional[np.ndarray] = None,
        blur_sigma_px: Optional[float] = 0.2,
        assume_hsi_is_reflectance: bool = True,
        *,
        hsi_downsample: bool = False,
        hsi_scale: float = 0.1,
        hsi_model=None,
    ):
        """`hsi_model` (keyword-only, NOT in the reference, default None = reference behaviour): an
        animal_vision_amd.ml.MSTPlusPlusPredictor; when given, the RGB->31-band cube comes from MST++ on
        the device (the route the reference stores `onnx_path` for but never wires)."""
        self.onnx_path = onnx_path  # stored and unused, like the reference (honeybee.py:67)
        self.adaptation = adaptation
        self.mapping_mode = mapping_mode
        self.custom_matrix = custom_matrix
        self.blur_sigma_px = float(blur_sigma_px or 0.0)
        self.assume_hsi_is_reflectance = assume_hsi_is_reflectance
        self.hsi_downsample = bool(hsi_downsample)
        self.hsi_scale = float(hsi_scale)
        self.lambdas = (np.linspace(400.0, 700.0, 31, dtype=np.float32) if hsi_band_centers_nm is None
                        else np.asarray(hsi_band_centers_nm, dtype=np.float32))
        self.E = illuminant if illuminant is not None else D65_like
        self.UV_curve, self.Blue_curve, self.Green_curve = honeybee_cone_curves(self.lambdas)
        self._eps = EPS_DEFAULT
        self.hsi_model = hsi_model
        if mapping_mode not in ("falsecolor", "custom_matrix", "opponent", "uv_purple_yellow", "falsecolor_uv_mixed"):
            pass  # the reference raises at visualize time (honeybee.py:164); so does the op below
        self._op = None

    def _operator(self) -> HoneybeeOp:
        if self._op is None:
            self._op = HoneybeeOp(lambdas=self.lambdas, illuminant=self.E, curves=(self.UV_curve, self.Blue_curve, self.Green_curve),
                                  reflectance=self.assume_hsi_is_reflectance, adaptation=self.adaptation, mapping_mode=self.mapping_mode,
                                  custom_matrix=self.custom_matrix, blur_sigma_px=self.blur_sigma_px, eps=self._eps)
        return self._op

    def visualize(self, image: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        assert isinstance(image, np.ndarray), "Input must be a numpy ndarray."
        assert image.ndim == 3 and image.shape[2] == 3, "Input must be HxWx3 RGB."
        if self.hsi_downsample and 0.05 <= self.hsi_scale < 1.0:
            raise NotImplementedError("hsi_downsample needs the INTER_AREA/INTER_LINEAR resampling kernels (SURVEY 8f row 1)")
        if image.dtype != np.uint8:
            raise NotImplementedError(f"HoneyBee: device path implemented for uint8 frames, got {image.dtype}")
        op = self._operator()
        if self.hsi_model is not None:
            out = self.hsi_model.honeybee(image, op)
        else:
            out = op(image)
        return image, out
