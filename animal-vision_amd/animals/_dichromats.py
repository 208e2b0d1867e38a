"""The dichromat species (reference: animals/{dog,cat,...,tiger}.py), all on the fused HIP kernel.

Each class keeps the reference's surface: no-argument constructor, `visualize(image) ->
(baseline, out)` with `baseline is image` (dog.py:61) and out.dtype == image.dtype.
Parameters: SURVEY.md Appendix A, each from the cited species file."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ..dichromat import DichromatOp, DichromatSpec
from .animal import Animal
from .animal_utils import check_input_image


class _Dichromat(Animal):
    SPEC: DichromatSpec = None  # type: ignore

    def __init__(self):
        self._op = None

    def _operator(self) -> DichromatOp:
        if self._op is None:
            self._op = DichromatOp(self.SPEC)
        return self._op

    def visualize(self, image: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        assert check_input_image(image)  # dog.py:33
        if image.dtype != np.uint8:
            # The reference tolerates float / wider-int frames (animal_utils.py:45-48); its video, webcam and image
            # renderers only ever produce uint8 (video.py:95).  Float frames of the Gaussian / row-gain species run as a
            # plane program; anything else is not silently re-routed to a CPU path: say so.
            if np.issubdtype(image.dtype, np.floating) and self.SPEC.color == "collapse":
                return image, self._visualize_float(image)
            raise NotImplementedError(f"{type(self).__name__}: device path implemented for uint8 frames (and float frames of the "
                                      f"collapse-matrix species), got {image.dtype}")
        return image, self._operator()(image)

    def _visualize_float(self, image: np.ndarray) -> np.ndarray:
        """dog.py:37-59 for a float frame: get_normalized_image's `max > 1` rule (a frame-wide reduction), both transfer
        functions as expressions, `px @ T.T`, cv2.GaussianBlur((0,0), sigma) or the S-cone row gain; float32 sRGB out."""
        from ..dichromat import collapse_LMS_matrix, cv_auto_ksize, gaussian_taps, s_cone_row_gain
        from ..planevm import DeviceBackend, PlaneRef

        plans = self.__dict__.setdefault("_float_plans", {})
        key = image.shape[:2]
        be = plans.get(key)
        if be is None:
            if len(plans) >= 4:
                plans.pop(next(iter(plans))).close()
            H, W = key
            sp = self.SPEC
            be = DeviceBackend(H, W, float_frames=True)
            y = [be.load(PlaneRef(be.d_in, 4 * c, 3)) for c in range(3)]
            mx = be.max(be.maximum(be.maximum(y[0], y[1]), y[2]))
            a = 0.055
            lin = []
            for c in range(3):  # get_normalized_image (animal_utils.py:41-50) + srgb_to_linear (:5-11)
                x = be.clip01(be.where(mx > 1.0, y[c] / 255.0, y[c]))
                lin.append(be.where(x <= 0.04045, x / 12.92, ((x + a) / (1 + a)) ** 2.4))
            T = collapse_LMS_matrix(sp.alpha, sp.s_scale)  # float32 3x3; out_i = sum_j T[i][j] * in_j (dog.py:47)
            rgb = [lin[0] * float(T[i, 0]) + lin[1] * float(T[i, 1]) + lin[2] * float(T[i, 2]) for i in range(3)]
            if sp.post == "gauss":
                k = cv_auto_ksize(sp.sigma)
                rgb = be.blur_taps(rgb, k, gaussian_taps(k, sp.sigma))
            elif sp.post == "scone":
                s_top, s_bottom, power, boost = sp.scone
                rgb = [rgb[0], rgb[1], be.clip01(rgb[2] * be.row(s_cone_row_gain(H, s_top, s_bottom, power=power, extra_boost=boost)))]
            elif sp.post == "streak":  # animal_utils.py:147-172 as coded (Q3); the float32 frame is blurred in place (Q4)
                rgb = be.streak(rgb, sp.streak)
            if sp.chroma is not None:  # apply_chroma_compression (animal_utils.py:174-181), where its result is used
                gray = ((rgb[0] + rgb[1]) + rgb[2]) / 3.0
                keep = float(np.float32(1.0 - sp.chroma))
                rgb = [gray + (v - gray) * keep for v in rgb]
            for c in range(3):  # np.clip(linear_to_srgb(np.clip(x, 0, 1)), 0, 1).astype(dtype)  (dog.py:54-59)
                x = be.clip01(rgb[c])
                be.store(be.clip01(be.where(x <= 0.0031308, 12.92 * x, (1 + a) * be.power(x, 1 / 2.4) - a)), PlaneRef(be.d_out, 4 * c, 3))
            be.flush()
            plans[key] = be
        ctx = be.ctx
        ctx.upload(np.ascontiguousarray(image, dtype=np.float32), be.d_in)
        be.run_device()
        out = ctx.download(be.d_out, image.shape, np.float32)
        return out.astype(image.dtype, copy=False)


def _mk(cls_name: str, spec: DichromatSpec, doc: str):
    return type(cls_name, (_Dichromat,), {"SPEC": spec, "__doc__": doc})


Dog = _mk("Dog", DichromatSpec("dog", 0.58, 0.65, sigma=3.5), "animals/dog.py:46,51")
Squirrel = _mk("Squirrel", DichromatSpec("squirrel", 0.55, 1.05, sigma=0.7), "animals/squirrel.py:29,34")
Elephant = _mk("Elephant", DichromatSpec("elephant", 0.60, 0.95, sigma=1.8), "animals/elephant.py:29,34")
Lion = _mk("Lion", DichromatSpec("lion", 0.60, 0.95, sigma=1.2), "animals/lion.py:29,34")
Tiger = _mk("Tiger", DichromatSpec("tiger", 0.60, 0.95, sigma=1.2), "animals/tiger.py:29,34")
Bear = _mk("Bear", DichromatSpec("bear", 0.60, 0.95, sigma=1.6), "animals/bear.py:29,34")
Wolf = _mk("Wolf", DichromatSpec("wolf", 0.65, 0.95, sigma=1.4), "animals/wolf.py:29,34")
Fox = _mk("Fox", DichromatSpec("fox", 0.65, 0.98, sigma=1.3), "animals/fox.py:29,34")
Raccoon = _mk("Raccoon", DichromatSpec("raccoon", 0.60, 0.98, sigma=2.0), "animals/raccoon.py:29,34")
Rat = _mk("Rat", DichromatSpec("rat", 0.05, 0.86, post="scone", scone=(1.3, 0.5, 1.4, 0.25)), "animals/rat.py:29,34")
# Streak-blur species: apply_anisotropic_acuity_blur_with_streak as coded (quirks Q3/Q4), csrc/dichromat_streak.hip.
Sheep = _mk("Sheep", DichromatSpec("sheep", 0.74, 1.06, post="streak", streak=(0.48, 0.8, 2.2, 6.0)), "animals/sheep.py:30,35")
Pig = _mk("Pig", DichromatSpec("pig", 0.89, 1.32, post="streak", streak=(0.5, 1.2, 2.5, 3.0)), "animals/pig.py:30,35,38 (chroma result discarded, Q4)")
Cow = _mk("Cow", DichromatSpec("cow", 0.84, 1.07, post="streak", streak=(0.5, 0.9, 2.3, 6.5)), "animals/cow.py:29,34")
Goat = _mk("Goat", DichromatSpec("goat", 0.75, 1.06, post="streak", streak=(0.5, 0.8, 2.4, 8.0)), "animals/goat.py:29,34")
Horse = _mk("Horse", DichromatSpec("horse", 0.30, 1.02, post="streak", streak=(0.5, 0.8, 2.2, 6.0)), "animals/horse.py:29,34")
Rabbit = _mk("Rabbit", DichromatSpec("rabbit", 0.20, 1.01, post="streak", streak=(0.52, 0.9, 2.5, 5.0), chroma=0.06), "animals/rabbit.py:29,34,37")
Panda = _mk("Panda", DichromatSpec("panda", 0.58, 0.74, post="streak", streak=(0.52, 1.0, 2.1, 4.5), chroma=0.06), "animals/panda.py:29,34,37")
Deer = _mk("Deer", DichromatSpec("deer", 0.60, 0.95, post="streak", streak=(0.5, 0.8, 2.6, 8.0)), "animals/deer.py:29,34")
Kangaroo = _mk("Kangaroo", DichromatSpec("kangaroo", 0.60, 0.98, post="streak", streak=(0.55, 0.8, 2.3, 8.0)), "animals/kangaroo.py:29,34")


class Cat(_Dichromat):
    """animals/cat.py (Tina-animals side of the unresolved merge, quirk Q8): returns (human_zoomed, cat_wide).

    human_zoomed = centre zoom of the input (cat.py:74-79, uint8 INTER_LINEAR); cat_wide = binocular wide-FOV
    warp of the ORIGINAL (cat.py:82-92) -> float64 L/M-merge colour tail -> sigma 1.0 blur -> OETF (cat.py:95-109).
    With ENABLE_FOV_WARP the warp output is float, so its sRGB decode uses the device powf (final bytes within
    1 code of the reference); with it off the uint8 path is bit-exact."""

    SPEC = DichromatSpec("cat", 0.5, 1.0, color="cat_merge", sigma=1.0)
    CAMERA_HFOV_DEG = 100.0
    CAT_PER_EYE_HALF_FOV_DEG = 105.0
    CAT_OVERLAP_DEG = 40.0
    CAT_TO_HUMAN_RATIO = 1.30
    ENABLE_FOV_WARP = True

    def visualize(self, image: np.ndarray):
        assert isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] == 3, "HxWx3 RGB"
        if image.dtype != np.uint8:
            if np.issubdtype(image.dtype, np.floating):
                return self._visualize_float(image)
            raise NotImplementedError(f"Cat: device path implemented for uint8 and float frames, got {image.dtype}")
        from .. import geometry as G

        H, W = image.shape[:2]
        scale = G.zoom_scale_from_cat_ratio(camera_hfov_deg=self.CAMERA_HFOV_DEG, cat_per_eye_half_fov_deg=self.CAT_PER_EYE_HALF_FOV_DEG,
                                            cat_to_human_ratio=self.CAT_TO_HUMAN_RATIO)
        human_zoomed = G.center_zoom(image, scale=scale)
        op = self._operator()
        if not self.ENABLE_FOV_WARP:
            return human_zoomed, op(image)
        ctx = op._ctx()
        tables = G.binocular_warp_tables(H, W, W, H, self.CAMERA_HFOV_DEG, self.CAT_PER_EYE_HALF_FOV_DEG, self.CAT_OVERLAP_DEG)
        d_in = ctx.upload(image)
        d_warp = ctx.malloc(H * W * 3 * 4)
        d_out = ctx.malloc(H * W * 3)
        try:
            G.binocular_warp_device(ctx, d_in, H, W, tables, H, W, d_warp)
            op.desc.in_f32 = 1
            try:
                op.run_device(d_warp, d_out, 1, H, W)
            finally:
                op.desc.in_f32 = 0
            cat_out = ctx.download(d_out, image.shape, np.uint8)
        finally:
            d_in.free(); d_warp.free(); d_out.free()
        return human_zoomed, cat_out

    def _visualize_float(self, image: np.ndarray):
        """cat.py:73-112 for a float frame, as a plane program in float32 (the reference's float64 tail is held to the float
        pipeline's 1e-4, like every float frame): get_normalized_image's `max > 1` rule, the binocular warp as two
        cv2.remap's with per-column / per-row maps + the cos^2 blend, srgb_to_linear, RGB->LMS, L/M merge, LMS->RGB, sigma 1.0
        blur, OETF; returns (centre-zoomed input, cat view), both in the input's dtype."""
        from .. import geometry as G
        from ..dichromat import cv_auto_ksize, gaussian_taps
        from ..planevm import DeviceBackend, PlaneRef

        H, W = image.shape[:2]
        scale = G.zoom_scale_from_cat_ratio(camera_hfov_deg=self.CAMERA_HFOV_DEG, cat_per_eye_half_fov_deg=self.CAT_PER_EYE_HALF_FOV_DEG,
                                            cat_to_human_ratio=self.CAT_TO_HUMAN_RATIO)
        human_zoomed = G.center_zoom(image, scale=scale)
        plans = self.__dict__.setdefault("_float_plans", {})
        key = (H, W, bool(self.ENABLE_FOV_WARP))
        be = plans.get(key)
        if be is None:
            if len(plans) >= 4:
                plans.pop(next(iter(plans))).close()
            be = DeviceBackend(H, W, float_frames=True)
            y = [be.load(PlaneRef(be.d_in, 4 * c, 3)) for c in range(3)]
            mx = be.max(be.maximum(be.maximum(y[0], y[1]), y[2]))
            x = [be.clip01(be.where(mx > 1.0, v / 255.0, v)) for v in y]  # get_normalized_image (animal_utils.py:41-50)
            if self.ENABLE_FOV_WARP:  # cat_widevision_utils.py:46-99
                xL, xR, ymap, wL, wR = G.binocular_warp_tables(H, W, W, H, self.CAMERA_HFOV_DEG, self.CAT_PER_EYE_HALF_FOV_DEG, self.CAT_OVERLAP_DEG)
                my = be.row(ymap)
                left = be.remap(x, be.col(xL), my, 0.0)
                right = be.remap(x, be.col(xR), my, 0.0)
                cwl, cwr = be.col(wL), be.col(wR)
                wsum = cwl + cwr + 1e-8
                x = [be.clip01((left[c] * cwl + right[c] * cwr) / wsum) for c in range(3)]
            a = 0.055
            lin = [be.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4) for v in x]
            M = [[0.31399022, 0.63951294, 0.04649755], [0.15537241, 0.75789446, 0.08670142], [0.01775239, 0.10944209, 0.87256922]]
            Bk = [[5.472213, -4.6419606, 0.16963711], [-1.125242, 2.2931712, -0.16789523], [0.02980164, -0.19318072, 1.1636479]]
            lms = [lin[0] * M[i][0] + lin[1] * M[i][1] + lin[2] * M[i][2] for i in range(3)]       # animal_utils.py:52-63
            al = float(np.float32(self.SPEC.alpha))
            lm = lms[0] * al + lms[1] * float(np.float32(1.0 - self.SPEC.alpha))                    # cat.py:99
            rgb = [lm * Bk[i][0] + lm * Bk[i][1] + lms[2] * Bk[i][2] for i in range(3)]            # LMS_to_RGB (:65-77)
            k = cv_auto_ksize(self.SPEC.sigma)
            rgb = be.blur_taps(rgb, k, gaussian_taps(k, self.SPEC.sigma))
            for c in range(3):
                v = be.clip01(rgb[c])
                be.store(be.clip01(be.where(v <= 0.0031308, 12.92 * v, (1 + a) * be.power(v, 1 / 2.4) - a)), PlaneRef(be.d_out, 4 * c, 3))
            be.flush()
            plans[key] = be
        ctx = be.ctx
        ctx.upload(np.ascontiguousarray(image, dtype=np.float32), be.d_in)
        be.run_device()
        out = ctx.download(be.d_out, image.shape, np.float32)
        return human_zoomed.astype(image.dtype, copy=False), out.astype(image.dtype, copy=False)
