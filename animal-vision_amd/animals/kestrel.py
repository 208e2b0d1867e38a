"""animals/kestrel.py of the reference on the device: same constructor keywords and defaults (:36-92)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class Kestrel(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (440.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        panorama_scale: float = 1.10,
        sky_cool_tint: Tuple[float, float, float] = (0.95, 0.98, 1.03),
        sky_haze: float = 0.10,
        ground_warm_tint: Tuple[float, float, float] = (1.02, 1.01, 0.99),
        ground_contrast: float = 0.08,
        uv_overlay_strength: float = 0.55,
        uv_magenta: Tuple[float, float, float] = (0.60, 0.12, 0.70),
        ridge_sigma: float = 3,
        ridge_gain: float = 1.0,
        unsharp_sigma: float = 1.0,
        unsharp_amount: float = 0.30,
        periph_blur_sigma: float = 0.7,
        periph_radius: float = 0.82,
        periph_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.panorama_scale = float(panorama_scale)
        self.sky_cool_tint = np.array(sky_cool_tint, np.float32)
        self.sky_haze = float(sky_haze)
        self.ground_warm_tint = np.array(ground_warm_tint, np.float32)
        self.ground_contrast = float(ground_contrast)
        self.uv_overlay_strength = float(np.clip(uv_overlay_strength, 0.0, 1.0))
        self.uv_magenta = np.array(uv_magenta, np.float32)
        self.ridge_sigma = float(ridge_sigma)
        self.ridge_gain = float(ridge_gain)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)

    def _ridge_measure(self, be, u, sigma: float):
        """kestrel.py:106-129: structure-tensor coherence x gradient energy of the UV map."""
        gx, gy = be.sobel(u)
        gxx, gyy, gxy = be.blur([gx * gx, gy * gy, gx * gy], sigma)
        trace, diff = gxx + gyy, gxx - gyy
        root = be.sqrt(be.maximum((0.5 * diff) ** 2 + gxy * gxy, 0.0))
        lam1, lam2 = 0.5 * trace + root, 0.5 * trace - root
        coh = (lam1 - lam2) / (lam1 + lam2 + 1e-8)
        energy = be.maximum(trace, 0.0)
        energy = energy / (be.percentile(energy, 95.0) + 1e-8)
        return be.clip01(coh * energy)

    def render(self, be, image):
        """kestrel.py:132-234: sky/ground split, UV 'trailness' ridges on the ground, magenta UV overlay."""
        rgb = be.front(image, self.panorama_scale)                                                        # :136-140
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi)]
        u_raw, b_raw, g_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :142-154 (Rv is never used)
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bv, Gv = be.safe_norm(b_raw), be.safe_norm(g_raw)
        vert_prior = be.row(np.linspace(1.0, 0.0, be.H, dtype=np.float32))                                # :158-166
        blue_dom = be.clip01(Bv - 0.6 * Gv)
        sky_score = be.blur([0.6 * vert_prior + 0.4 * blue_dom], 3.0)[0]
        sky_score = be.clip01(sky_score / (be.percentile(sky_score, 98.0) + 1e-8))
        sky_w = be.mat(1.0 / (1.0 + be.exp(-6.0 * (sky_score - 0.45))))
        ground_w = 1.0 - sky_w
        ridge = self._ridge_measure(be, U, self.ridge_sigma)                                              # :171-172
        trailness = be.mat(be.clip01(self.ridge_gain * ridge * ground_w))
        tinted = [be.clip01(c * k) for c, k in zip(rgb, self.sky_cool_tint)]                              # :176-182
        if self.sky_haze > 0.0:
            a = float(np.clip(self.sky_haze, 0.0, 1.0))
            rgb = [sky_w * ((1.0 - a) * t + a * k) + ground_w * c for c, t, k in zip(rgb, tinted, np.array([0.90, 0.97, 1.00], np.float32))]
        else:
            rgb = [sky_w * t + ground_w * c for c, t in zip(rgb, tinted)]
        ground = [be.clip01(c * k) for c, k in zip(rgb, self.ground_warm_tint)]                           # :185-190
        if self.ground_contrast > 0.0:
            blurred = be.blur(ground, 1.2)
            ground = [be.clip01(g_ + self.ground_contrast * (g_ - b)) for g_, b in zip(ground, blurred)]
        rgb = [sky_w * c + ground_w * g_ for c, g_ in zip(rgb, ground)]
        U95 = be.clip01(U / (be.percentile(U, 95.0) + 1e-8))                                              # :193-199
        ow = self.uv_overlay_strength * ground_w
        rgb = [be.clip01((1.0 - ow) * c + ow * (U95 * k)) for c, k in zip(rgb, self.uv_magenta)]
        if self.unsharp_sigma > 0.0 and self.unsharp_amount > 0.0:                                        # :202-205
            blur = be.blur(rgb, self.unsharp_sigma)
            gain = self.unsharp_amount * trailness
            rgb = [be.clip01(c + gain * be.clip(c - b, -1.0, 1.0)) for c, b in zip(rgb, blur)]
        if self.periph_blur_sigma > 0.0:                                                                  # :208-214
            periph = be.blur(rgb, self.periph_blur_sigma)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)
