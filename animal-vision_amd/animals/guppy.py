"""animals/guppy.py of the reference on the device: same constructor keywords and defaults (:36-100)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


def luma(rgb):
    """0.2126 R + 0.7152 G + 0.0722 B (guppy.py:109-110 and siblings)."""
    return 0.2126 * rgb[0] + 0.7152 * rgb[1] + 0.0722 * rgb[2]


class Guppy(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (430.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        panorama_scale: float = 1.22,
        red_kill: float = 0.28,
        haze_strength: float = 0.06,
        haze_tint: Tuple[float, float, float] = (0.92, 0.98, 1.00),
        warm_tint: Tuple[float, float, float] = (1.03, 1.01, 0.99),
        base_soft_sigma: float = 0.35,
        unsharp_sigma: float = 0.9,
        unsharp_amount: float = 0.28,
        dog_small_sigma: float = 0.8,
        dog_large_sigma: float = 2.4,
        dog_gain: float = 0.85,
        uv_chroma_boost: float = 0.40,
        uv_blue_gain: float = 0.55,
        uv_green_gain: float = 0.35,
        uv_red_gain: float = 0.12,
        background_desat: float = 0.18,
        vignette_strength: float = 0.12,
        vignette_radius: float = 0.78,
        vignette_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.panorama_scale = float(panorama_scale)
        self.red_kill = float(red_kill)
        self.haze_strength = float(haze_strength)
        self.haze_tint = np.array(haze_tint, dtype=np.float32)
        self.warm_tint = np.array(warm_tint, dtype=np.float32)
        self.base_soft_sigma = float(base_soft_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)
        self.dog_small_sigma = float(dog_small_sigma)
        self.dog_large_sigma = float(dog_large_sigma)
        self.dog_gain = float(dog_gain)
        self.uv_chroma_boost = float(uv_chroma_boost)
        self.uv_blue_gain = float(uv_blue_gain)
        self.uv_green_gain = float(uv_green_gain)
        self.uv_red_gain = float(uv_red_gain)
        self.background_desat = float(background_desat)
        self.vignette_strength = float(vignette_strength)
        self.vignette_radius = float(vignette_radius)
        self.vignette_softness = float(vignette_softness)

    def render(self, be, image):
        """guppy.py:122-235: DoG of the UV map gates an unsharp mask, a chroma lift and a background desaturation."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :138-149
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi)]
        u_raw, b_raw, g_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :151-171 (Rn is never used)
        Un = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bn, Gn = be.safe_norm(b_raw), be.safe_norm(g_raw)
        rgb = [be.clip01(R * (1.0 - self.red_kill)), G, B]                                                # :174-175
        if self.haze_strength > 0.0:                                                                      # :177-179
            a = float(np.clip(self.haze_strength, 0.0, 1.0))
            rgb = [(1.0 - a) * c + a * t for c, t in zip(rgb, self.haze_tint)]
        rgb = [be.clip01(c * w) for c, w in zip(rgb, self.warm_tint)]                                     # :181
        if self.base_soft_sigma > 0.0:                                                                    # :183-184
            rgb = be.blur(rgb, self.base_soft_sigma)
        uv_small = be.blur([Un], self.dog_small_sigma)[0]                                                 # :187-191
        uv_large = be.blur([Un], self.dog_large_sigma)[0]
        uv_dog = be.clip01(uv_small - uv_large)
        uv_spot = be.clip01(uv_dog / (be.percentile(uv_dog, 95.0) + 1e-8))
        if self.unsharp_sigma > 0.0 and self.unsharp_amount > 0.0:                                        # :194-197
            blurred = be.blur(rgb, self.unsharp_sigma)
            gain = self.unsharp_amount * uv_spot
            rgb = [be.clip01(c + gain * be.clip(c - bl, -1.0, 1.0)) for c, bl in zip(rgb, blurred)]
        R, G, B = rgb
        lift = self.uv_chroma_boost * uv_spot                                                             # :200-203
        B = be.clip01(B + self.uv_blue_gain * lift * Bn)
        G = be.clip01(G + self.uv_green_gain * lift * Gn)
        R = be.clip01(R + self.uv_red_gain * lift * Un)
        rgb = [R, G, B]
        Y = luma(rgb)                                                                                     # :206-208, :112-120
        mean_chroma = (be.abs(R - Y) + be.abs(G - Y) + be.abs(B - Y)) / 3
        sat = mean_chroma / (be.percentile(mean_chroma, 95.0) + 1e-8)
        desat = 1.0 - self.background_desat * (1.0 - Un) * (1.0 - sat)
        rgb = [be.clip01(Y + (c - Y) * desat) for c in rgb]
        if self.vignette_strength > 0.0:                                                                  # :211-218
            t = radial_sigmoid(be, radius=self.vignette_radius, softness=self.vignette_softness)
            vign = 1.0 - self.vignette_strength * t
            rgb = [be.clip01(c * vign) for c in rgb]
        be.encode(rgb)
