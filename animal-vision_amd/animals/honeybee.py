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
        if image.dtype != np.uint8:
            if not np.issubdtype(image.dtype, np.floating) or self.hsi_model is not None:
                raise NotImplementedError(f"HoneyBee: device path implemented for uint8 and float frames, got {image.dtype}")
            return image, self._visualize_staged(image, floats=True)
        op = self._operator()
        if self.hsi_model is not None:
            out = self.hsi_model.honeybee(image, op)
        elif self.hsi_downsample and 0.05 <= self.hsi_scale < 1.0:
            out = self._visualize_staged(image, floats=False)
        else:
            out = op(image)
        return image, out

    def _visualize_staged(self, image: np.ndarray, floats: bool) -> np.ndarray:
        """The routes the fused uint8 kernel does not take, as a recorded sequence of stage calls (planevm.py):
        * `hsi_downsample` (honeybee.py:109-117): classic_rgb_to_hsi_scaled = INTER_AREA down -> lobes -> INTER_LINEAR up
          of the cube.  The three cone catches are linear in the cube, so they are taken at the reduced size and the
          3-plane result is upsampled (instead of 31 bands);
        * float frames (to_float01's `max > 1.001` rule, float output = linear_to_srgb(clip(rgb_lin)).astype(dtype)).
        The tail (adaptation, blur, map, encode) is the same device pipeline, fed the catches as a 3-band cube with
        identity weights."""
        import ctypes

        from ..planevm import DeviceBackend, PlaneRef

        down = self.hsi_downsample and 0.05 <= self.hsi_scale < 1.0
        plans = self.__dict__.setdefault("_staged_plans", {})
        key = (image.shape[0], image.shape[1], floats, down)
        hit = plans.get(key)
        if hit is None:
            if len(plans) >= 4:
                plans.pop(next(iter(plans)))[0].close()
            H, W = key[:2]
            be = DeviceBackend(H, W, float_frames=floats)
            img01 = be._alloc(4 * 3 * be.n)
            if floats:  # uv_helpers.to_float01 (:15-23): /255 and clip only when the frame's maximum exceeds 1.001
                y = [be.load(PlaneRef(be.d_in, 4 * c, 3)) for c in range(3)]
                mx = be.max(be.maximum(be.maximum(y[0], y[1]), y[2]))
                for c in range(3):
                    be.store(be.where(mx > 1.001, be.clip01(y[c] / 255.0), y[c]), PlaneRef(img01, 4 * c, 3))
            else:       # byte / 255
                for c in range(3):
                    be.store(be.load(PlaneRef(be.d_in, c, 3, "u8")) / 255.0, PlaneRef(img01, 4 * c, 3))
            be.lin_hwc = img01
            op = self._operator()
            be.spectral_planes(self.lambdas, op.weights, self.hsi_scale, downsample=down)
            be.flush()
            tail = HoneybeeOp(lambdas=self.lambdas, illuminant=self.E, curves=(self.UV_curve, self.Blue_curve, self.Green_curve),
                              reflectance=self.assume_hsi_is_reflectance, adaptation=self.adaptation, mapping_mode=self.mapping_mode,
                              custom_matrix=self.custom_matrix, blur_sigma_px=self.blur_sigma_px, eps=self._eps, ctx=be.ctx)
            tail.weights = np.eye(3, dtype=np.float32)
            tail.desc.bands = 3
            tail.desc.weights_host = tail.weights.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            tail.desc.out_float = 1 if floats else 0
            hit = (be, tail)
            plans[key] = hit
        be, tail = hit
        ctx = be.ctx
        H, W = key[:2]
        dt = np.float32 if floats else np.uint8
        ctx.upload(np.ascontiguousarray(image, dtype=dt), be.d_in)  # float64 frames: x.astype(float32), the reference's first step
        be.run_device()
        tail.run_device(None, be.d_out, 1, H, W, hsi_ptr=be.last_stack.ptr, hsi_layout=0, hsi_dtype=0)
        out = ctx.download(be.d_out, image.shape, dt)
        return out.astype(image.dtype) if out.dtype != image.dtype else out
