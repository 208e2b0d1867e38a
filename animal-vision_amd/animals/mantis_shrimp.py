"""animals/mantis_shrimp.py of the reference, on the device.  Same constructor keywords and defaults
(mantis_shrimp.py:42-119), same visualize contract: (baseline_rgb, mantis_rgb), both HxWx3 uint8.

Host side = the O(bands), O(H) and O(W) tables (band-pass windows folded with the analytic lobes, hue LUT,
scanline profile, radial coordinates, Gaussian taps); every O(pixels) pass is csrc/mantis.hip."""
import ctypes
from typing import Optional, Tuple

import numpy as np

from .. import _lib
from .._lib import MantisDesc, lib
from ..dichromat import gaussian_taps
from ..runtime import Context, DeviceBuffer, get_context
from ..uv import bandpass_weights, fold_rgb_matrix, lobe_tables, uv_blur_ksize
from .animal import Animal


def hue_lut(N: int) -> np.ndarray:
    """mantis_shrimp.py:175-197: N evenly spaced hues at s=0.95, v=1 (float32 throughout)."""
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
    return np.ascontiguousarray(np.stack([np.select(conds, [v, q, p, p, t, v], default=v), np.select(conds, [t, v, v, q, p, p], default=v),
                                          np.select(conds, [p, p, t, v, v, q], default=v)], axis=-1), dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class MantisShrimp(Animal):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        panorama_scale: float = 1.12,
        bands: Tuple[Tuple[float, float], ...] = (
            (320.0, 360.0), (360.0, 400.0), (400.0, 430.0), (430.0, 460.0), (460.0, 490.0),
            (490.0, 520.0), (520.0, 550.0), (550.0, 580.0), (580.0, 610.0), (610.0, 680.0),
        ),
        red_kill: float = 0.18,
        haze_strength: float = 0.03,
        haze_tint: Tuple[float, float, float] = (0.92, 0.98, 1.00),
        pre_soft_sigma: float = 0.25,
        unsharp_sigma: float = 1.0,
        unsharp_amount: float = 0.32,
        evec_angle_deg: float = 30.0,
        pol_linear_strength: float = 0.55,
        pol_linear_gamma: float = 1.2,
        pol_circular_strength: float = 0.35,
        orientation_mix: float = 0.5,
        barcode_saturation: float = 0.40,
        barcode_opacity: float = 0.55,
        winner_take_most: float = 0.35,
        scan_row_freq: float = 26.0,
        scan_row_gain: float = 0.08,
        scan_soften: float = 0.8,
        periph_blur_sigma: float = 0.7,
        periph_radius: float = 0.80,
        periph_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = np.asarray(lambdas, dtype=np.float32) if lambdas is not None else np.linspace(300.0, 700.0, 81, dtype=np.float32)
        assert self.lambdas.ndim == 1 and self.lambdas.size >= 10, "lambdas must be a 1D wavelength vector (nm)."
        self.panorama_scale = float(panorama_scale)
        self.bands = tuple((float(a), float(b)) for (a, b) in bands)
        self.red_kill = float(red_kill)
        self.haze_strength = float(haze_strength)
        self.haze_tint = np.array(haze_tint, dtype=np.float32)
        self.pre_soft_sigma = float(pre_soft_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)
        self.evec_angle = np.deg2rad(float(evec_angle_deg))
        self.pol_linear_strength = float(pol_linear_strength)
        self.pol_linear_gamma = float(pol_linear_gamma)
        self.pol_circular_strength = float(pol_circular_strength)
        self.orientation_mix = float(np.clip(orientation_mix, 0.0, 1.0))
        self.barcode_saturation = float(barcode_saturation)
        self.barcode_opacity = float(np.clip(barcode_opacity, 0.0, 1.0))
        self.winner_take_most = float(np.clip(winner_take_most, 0.0, 1.0))
        self.scan_row_freq = float(scan_row_freq)
        self.scan_row_gain = float(scan_row_gain)
        self.scan_soften = float(scan_soften)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)
        self.ctx: Optional[Context] = None
        self._desc_cache = {}

    # ---- host tables --------------------------------------------------------------------------------
    def _blur(self, sigma: float):
        """uv_helpers.gaussian_blur (:66-81): sigma <= 0 is the identity; ksize = 2*ceil(3 sigma)+1."""
        if sigma <= 0.0:
            return 0, None
        k = uv_blur_ksize(sigma)
        if k > _lib.AVX_MAX_KSIZE:
            raise ValueError(f"MantisShrimp: blur sigma {sigma} needs ksize {k} > {_lib.AVX_MAX_KSIZE}")
        return k, gaussian_taps(k, sigma)

    def _desc(self, H: int, W: int):
        key = (H, W)
        hit = self._desc_cache.get(key)
        if hit is not None:
            return hit
        keep = {}  # host arrays the descriptor points into
        d = MantisDesc()
        d.struct_size = ctypes.sizeof(MantisDesc)
        N = len(self.bands)
        wts = np.ascontiguousarray(np.stack([bandpass_weights(self.lambdas, lo, hi) for lo, hi in self.bands]), dtype=np.float32)
        gains, denom = lobe_tables(self.lambdas)
        keep["M"] = fold_rgb_matrix(wts, self.lambdas)
        keep["lut"], keep["wts"], keep["gains"] = hue_lut(N), wts, gains
        d.n_bands, d.band_matrix_host, d.band_lut_host = N, _fp(keep["M"]), _fp(keep["lut"])
        d.n_wavelengths, d.lobe_gains_host, d.lobe_denom, d.band_weights_host = int(self.lambdas.size), _fp(gains), denom, _fp(wts)
        # panorama_warp (uv_helpers.py:84-99): identity for scale <= 1 or when rounding leaves the width alone
        new_w = W
        if self.panorama_scale != 1.0 and abs(self.panorama_scale - 1.0) >= 1e-3:
            new_w = max(2, int(round(W * self.panorama_scale)))
            if new_w < W:  # the reference's centre crop of a NARROWER image yields a frame of the wrong shape
                raise ValueError("MantisShrimp: panorama_scale < 1 is not a usable configuration of the reference")
        d.pano_new_w = new_w if new_w > W else 0
        if 0.0 < self.hsi_scale < 1.0:  # :156-163, uv_helpers.py:155-183
            d.hsi_small_h, d.hsi_small_w = max(1, int(round(H * self.hsi_scale))), max(1, int(round(W * self.hsi_scale)))
            if (d.hsi_small_h, d.hsi_small_w) == (H, W):
                d.hsi_small_h = d.hsi_small_w = 0
        d.red_keep = float(np.float32(1.0 - self.red_kill))
        a = float(np.clip(self.haze_strength, 0.0, 1.0)) if self.haze_strength > 0.0 else 0.0
        d.haze, d.haze_keep = a, 1.0 - a
        d.haze_tint = (ctypes.c_float * 3)(*self.haze_tint.tolist())
        for name, sigma in (("pre_soft", self.pre_soft_sigma), ("unsharp", self.unsharp_sigma), ("scan", self.scan_soften),
                            ("periph", self.periph_blur_sigma)):
            k, taps = self._blur(sigma)
            keep[name] = taps
            setattr(d, f"{name}_ksize", k)
            if taps is not None:
                setattr(d, f"{name}_taps_host", _dp(taps))
        mix = self.orientation_mix
        d.cos2_global = (1.0 - mix) * float(np.cos(2.0 * self.evec_angle))
        d.sin2_global = (1.0 - mix) * float(np.sin(2.0 * self.evec_angle))
        d.orientation_mix = mix
        d.pol_linear_strength, d.pol_linear_gamma, d.pol_circular_strength = self.pol_linear_strength, self.pol_linear_gamma, self.pol_circular_strength
        d.unsharp_amount = self.unsharp_amount
        d.barcode_saturation, d.barcode_opacity, d.winner_take_most = self.barcode_saturation, self.barcode_opacity, self.winner_take_most
        y = np.linspace(0.0, 1.0, H, dtype=np.float32)
        keep["rows"] = np.ascontiguousarray(0.5 + 0.5 * np.sin(2.0 * np.pi * self.scan_row_freq * y), dtype=np.float32)  # :257-258
        keep["xx"] = np.linspace(-1.0, 1.0, W, dtype=np.float32)
        keep["yy"] = np.linspace(-1.0, 1.0, H, dtype=np.float32)
        d.rows_host, d.xx_host, d.yy_host = _fp(keep["rows"]), _fp(keep["xx"]), _fp(keep["yy"])
        d.scan_row_gain = self.scan_row_gain
        d.periph_radius, d.periph_softness = self.periph_radius, self.periph_softness
        self._desc_cache[key] = (d, keep)
        return d, keep

    # ---- device ------------------------------------------------------------------------------------
    def run_device(self, d_in: DeviceBuffer, d_base: DeviceBuffer, d_out: DeviceBuffer, H: int, W: int, stream=None):
        if self.ctx is None:
            self.ctx = get_context()
        ctx = self.ctx
        d, _keep = self._desc(H, W)
        ctx._check(lib.avx_mantis_u8(ctx._h, d_in.ptr, d_base.ptr, d_out.ptr, H, W, ctypes.byref(d), ctx._s(stream)))

    def run_device_batch(self, d_in: DeviceBuffer, d_base: DeviceBuffer, d_out: DeviceBuffer, n_frames: int, H: int, W: int, stream=None):
        """n_frames contiguous uint8 frames (baselines / outputs contiguous too): independent frames overlap on the library's lanes."""
        if self.ctx is None:
            self.ctx = get_context()
        ctx = self.ctx
        d, _keep = self._desc(H, W)
        ctx._check(lib.avx_mantis_u8_batch(ctx._h, d_in.ptr, d_base.ptr, d_out.ptr, n_frames, H, W, ctypes.byref(d), ctx._s(stream)))

    def _visualize_float(self, image: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Float frames (same-dtype contract): to_float01's `max > 1.001` rule and srgb_to_linear run as a plane program
        (a frame-wide reduction decides the /255), the fused stack takes the linear frame and writes float32 sRGB."""
        from ..planevm import DeviceBackend, PlaneRef

        plans = self.__dict__.setdefault("_float_plans", {})
        key = image.shape[:2]
        hit = plans.get(key)
        if hit is None:
            if len(plans) >= 4:
                plans.pop(next(iter(plans)))[0].close()
            H, W = key
            be = DeviceBackend(H, W, float_frames=True)
            lin = be._alloc(4 * 3 * be.n)
            y = [be.load(PlaneRef(be.d_in, 4 * c, 3)) for c in range(3)]
            mx = be.max(be.maximum(be.maximum(y[0], y[1]), y[2]))
            a = 0.055
            for c in range(3):
                v = be.where(mx > 1.001, be.clip01(y[c] / 255.0), y[c])
                be.store(be.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4), PlaneRef(lin, 4 * c, 3))
            be.flush()
            d0, keep = self._desc(H, W)
            d = MantisDesc()  # a byte copy shares the host tables `keep` holds
            ctypes.memmove(ctypes.byref(d), ctypes.byref(d0), ctypes.sizeof(MantisDesc))
            d.lin_hwc_in, d.out_float = lin.ptr, 1
            hit = (be, d, keep)
            plans[key] = hit
        be, d, _keep = hit
        ctx = be.ctx
        H, W = key
        ctx.upload(np.ascontiguousarray(image, dtype=np.float32), be.d_in)
        be.run_device()
        ctx._check(lib.avx_mantis_u8(ctx._h, None, be.d_base.ptr, be.d_out.ptr, H, W, ctypes.byref(d), ctx._s(None)))
        base = ctx.download(be.d_base, image.shape, np.float32, sync=False)
        out = ctx.download(be.d_out, image.shape, np.float32)
        return base.astype(image.dtype, copy=False), out.astype(image.dtype, copy=False)

    def visualize(self, image: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        assert isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] == 3
        if image.dtype != np.uint8:
            if not np.issubdtype(image.dtype, np.floating):
                raise NotImplementedError(f"MantisShrimp: device path implemented for uint8 and float frames, got {image.dtype}")
            return self._visualize_float(image)
        if self.ctx is None:
            self.ctx = get_context()
        ctx = self.ctx
        H, W, _ = image.shape
        d_in = ctx.upload(np.ascontiguousarray(image))
        d_base, d_out = ctx.malloc(image.nbytes), ctx.malloc(image.nbytes)
        try:
            self.run_device(d_in, d_base, d_out, H, W)
            base = ctx.download(d_base, image.shape, np.uint8)
            out = ctx.download(d_out, image.shape, np.uint8)
        finally:
            for b in (d_in, d_base, d_out):
                b.free()
        return base, out
