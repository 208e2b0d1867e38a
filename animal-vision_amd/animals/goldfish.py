"""animals/goldfish.py of the reference on the device: same constructor keywords and defaults (:40-84)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class Goldfish(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (430.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        uv_boost: float = 3.0,
        panorama_scale: float = 1.45,
        haze_strength: float = 0.12,
        haze_tint: Tuple[float, float, float] = (0.78, 0.92, 1.0),
        red_kill: float = 0.55,
        green_lift: float = 0.12,
        blue_lift: float = 0.06,
        base_blur_sigma: float = 0.8,
        periph_blur_sigma: float = 1.8,
        periph_radius: float = 0.65,
        periph_softness: float = 6.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.uv_boost = float(uv_boost)
        self.panorama_scale = float(panorama_scale)
        self.haze_strength = float(haze_strength)
        self.haze_tint = np.array(haze_tint, dtype=np.float32)
        self.red_kill = float(red_kill)
        self.green_lift = float(green_lift)
        self.blue_lift = float(blue_lift)
        self.base_blur_sigma = float(base_blur_sigma)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)

    def render(self, be, image):
        """goldfish.py:86-180."""
        R, G, B = be.front(image, self.panorama_scale)                                                     # :103-113
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi), (self.red_lo, self.red_hi)]
        U, Bv, Gv, Rv = (be.safe_norm(b) for b in be.bands(self.lambdas, bands, self.hsi_scale))           # :116-129
        uv = be.safe_norm(U / (1e-6 + 0.45 * Gv + 0.35 * Bv + 0.15 * Rv))                                 # :132
        R = be.clip01(R * (1.0 - self.red_kill))                                                          # :138-140
        G = be.clip01(G + self.green_lift)
        B = be.clip01(B + self.blue_lift)
        rgb = [R, G, B]
        if self.haze_strength > 0.0:                                                                      # :143-145
            a = np.clip(self.haze_strength, 0.0, 1.0)  # an np.float64, as in the reference
            rgb = [(1.0 - a) * c + a * t for c, t in zip(rgb, self.haze_tint)]
        if self.base_blur_sigma > 0.0:                                                                    # :148-149
            rgb = be.blur(rgb, self.base_blur_sigma)
        R, G, B = rgb
        R = be.clip01(R + self.uv_boost * 0.42 * uv)                                                      # :153-156
        B = be.clip01(B + self.uv_boost * 0.35 * uv)
        G = be.clip01(G + self.uv_boost * 0.12 * uv)
        B = be.clip01(B + 0.22 * Bv)                                                                      # :159-160
        G = be.clip01(G + 0.30 * Gv)
        rgb = [R, G, B]
        if self.periph_blur_sigma > 0.0:                                                                  # :163-172
            periph = be.blur(rgb, self.periph_blur_sigma)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)                                                                                    # :175-176
