"""animals/damselfish.py of the reference on the device: same constructor keywords and defaults (:37-82)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class Damselfish(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (440.0, 500.0),
        yellow_band: Tuple[float, float] = (560.0, 600.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        uv_edge_boost: float = 0.45,
        uv_gloss_boost: float = 0.30,
        blue_chroma_gain: float = 0.22,
        yellow_chroma_gain: float = 0.28,
        red_kill: float = 0.35,
        base_blur_sigma: float = 0.35,
        unsharp_sigma: float = 1.2,
        panorama_scale: float = 1.25,
        periph_radius: float = 0.70,
        periph_softness: float = 7.0,
        periph_extra_blur: float = 0.8,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.yellow_lo, self.yellow_hi = map(float, yellow_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.uv_edge_boost = float(uv_edge_boost)
        self.uv_gloss_boost = float(uv_gloss_boost)
        self.blue_chroma_gain = float(blue_chroma_gain)
        self.yellow_chroma_gain = float(yellow_chroma_gain)
        self.red_kill = float(red_kill)
        self.base_blur_sigma = float(base_blur_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.panorama_scale = float(panorama_scale)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)
        self.periph_extra_blur = float(periph_extra_blur)

    def render(self, be, image):
        """damselfish.py:85-181."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :100-111
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.yellow_lo, self.yellow_hi)]
        u_raw, b_raw, y_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :113-127
        Bn, Yn = be.safe_norm(b_raw), be.safe_norm(y_raw)                                                 # :130-133 (Rn is never used)
        Un = be.safe_norm(be.safe_norm(u_raw))                                                            # integrate_uv, then safe_norm again
        R = be.clip01(R * (1.0 - self.red_kill))                                                          # :137
        rgb = [R, G, B]
        if self.base_blur_sigma > 0.0:                                                                    # :140-141
            rgb = be.blur(rgb, self.base_blur_sigma)
        if self.unsharp_sigma > 0.0 and self.uv_edge_boost > 0.0:                                         # :143-148
            blurred = be.blur(rgb, self.unsharp_sigma)
            gain = 1.0 + self.uv_edge_boost * Un
            rgb = [be.clip01(c + gain * be.clip(c - bl, -1.0, 1.0)) for c, bl in zip(rgb, blurred)]
        R, G, B = rgb
        if self.uv_gloss_boost > 0.0:                                                                     # :151-156
            lift = self.uv_gloss_boost * Un
            B = be.clip01(B + 0.60 * lift)
            G = be.clip01(G + 0.30 * lift)
            R = be.clip01(R + 0.15 * lift)
        B = be.clip01(B + self.blue_chroma_gain * Bn)                                                     # :159-163
        y_boost = self.yellow_chroma_gain * Yn
        G = be.clip01(G + 0.65 * y_boost)
        R = be.clip01(R + 0.35 * y_boost)
        rgb = [R, G, B]
        if self.periph_extra_blur > 0.0:                                                                  # :166-174
            periph = be.blur(rgb, self.periph_extra_blur)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)                                                                                    # :177-178
