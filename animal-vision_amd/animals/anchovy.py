"""animals/anchovy.py of the reference on the device: same constructor keywords and defaults (:39-97)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class Anchovy(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (440.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        panorama_scale: float = 1.20,
        red_kill: float = 0.25,
        base_soft_sigma: float = 0.30,
        unsharp_sigma: float = 1.0,
        unsharp_amount: float = 0.35,
        haze_strength: float = 0.04,
        haze_tint: Tuple[float, float, float] = (0.90, 0.97, 1.00),
        evec_angle_deg: float = 0.0,
        pol_strength: float = 0.55,
        pol_gamma: float = 1.2,
        orientation_mix: float = 0.35,
        uv_gloss_gain: float = 0.28,
        blue_chroma_gain: float = 0.18,
        green_chroma_gain: float = 0.10,
        periph_blur_sigma: float = 0.6,
        periph_radius: float = 0.78,
        periph_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.panorama_scale = float(panorama_scale)
        self.red_kill = float(red_kill)
        self.base_soft_sigma = float(base_soft_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)
        self.haze_strength = float(haze_strength)
        self.haze_tint = np.array(haze_tint, dtype=np.float32)
        self.evec_angle = np.deg2rad(float(evec_angle_deg))
        self.pol_strength = float(pol_strength)
        self.pol_gamma = float(pol_gamma)
        self.orientation_mix = float(np.clip(orientation_mix, 0.0, 1.0))
        self.uv_gloss_gain = float(uv_gloss_gain)
        self.blue_chroma_gain = float(blue_chroma_gain)
        self.green_chroma_gain = float(green_chroma_gain)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)

    def render(self, be, image):
        """anchovy.py:119-253: UV-gradient orientation vs a global E-vector drives an unsharp gain and a gloss."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :135-146
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi)]
        u_raw, b_raw, g_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :148-167 (Rn is never used)
        Un = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bn, Gn = be.safe_norm(b_raw), be.safe_norm(g_raw)
        gx, gy = be.sobel(Un)                                                                             # :170-172
        theta = be.arctan2(gy, gx)
        cos2_local = be.cos(2.0 * theta)
        cos2_global = float(np.cos(2.0 * self.evec_angle))                                                # :176-184
        mix = float(self.orientation_mix)
        align = (1.0 - mix) * cos2_global + mix * cos2_local
        align01 = be.clip01(0.5 * (align + 1.0)) ** float(self.pol_gamma)
        uv_mag = be.sqrt(gx * gx + gy * gy)                                                               # :187-190
        uv_mag = be.clip01(uv_mag / (be.percentile(uv_mag, 95.0) + 1e-8))
        pol_gain = 1.0 + self.pol_strength * (align01 * Un * uv_mag)
        R = be.clip01(R * (1.0 - self.red_kill))                                                          # :194
        rgb = [R, G, B]
        if self.haze_strength > 0.0:                                                                      # :197-199
            a = float(np.clip(self.haze_strength, 0.0, 1.0))
            rgb = [(1.0 - a) * c + a * t for c, t in zip(rgb, self.haze_tint)]
        if self.base_soft_sigma > 0.0:                                                                    # :202-203
            rgb = be.blur(rgb, self.base_soft_sigma)
        if self.unsharp_sigma > 0.0 and self.unsharp_amount > 0.0:                                        # :205-208
            blurred = be.blur(rgb, self.unsharp_sigma)
            gain = self.unsharp_amount * pol_gain
            rgb = [be.clip01(c + gain * be.clip(c - bl, -1.0, 1.0)) for c, bl in zip(rgb, blurred)]
        R, G, B = rgb
        gloss = self.uv_gloss_gain * (align01 * Un)                                                       # :211-213
        B = be.clip01(B + 0.70 * gloss)
        G = be.clip01(G + 0.30 * gloss)
        B = be.clip01(B + self.blue_chroma_gain * (Bn * Un))                                              # :216-217
        G = be.clip01(G + self.green_chroma_gain * (Gn * Un))
        rgb = [R, G, B]
        if self.periph_blur_sigma > 0.0:                                                                  # :220-228
            periph = be.blur(rgb, self.periph_blur_sigma)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)
