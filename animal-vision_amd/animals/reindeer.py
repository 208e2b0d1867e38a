"""animals/reindeer.py of the reference on the device: same constructor keywords and defaults (:41-68)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, snow_glare_tone_compress


class Reindeer(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (300.0, 410.0),
        uv_boost: float = 3.5,
        snow_glare_compression: float = 0.55,
        winter_mode: bool = True,
        scatter_sigma: float = 1.2,
        blue_bias: float = 0.08,
        panorama_scale: float = 1.3,
        return_uv_heatmap: bool = True,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = float(uv_band[0]), float(uv_band[1])
        self.uv_boost = float(uv_boost)
        self.snow_glare_compression = float(snow_glare_compression)
        self.winter_mode = bool(winter_mode)
        self.scatter_sigma = float(scatter_sigma)
        self.blue_bias = float(blue_bias)
        self.panorama_scale = float(panorama_scale)
        self.return_uv_heatmap = bool(return_uv_heatmap)

    def render(self, be, image):
        """reindeer.py:83-135."""
        R, G, B = be.front(image, self.panorama_scale)                                   # :86-98
        u_raw, v_raw = be.bands(self.lambdas, [(self.uv_lo, self.uv_hi), (420.0, 680.0)], self.hsi_scale)  # :100-113
        uv_map, vis_map = be.safe_norm(u_raw), be.safe_norm(v_raw)
        uv_saliency = be.safe_norm(uv_map / (1e-6 + 0.6 * vis_map))                    # :116
        B = be.clip01(B + self.uv_boost * 0.35 * uv_saliency)                           # :121-122
        G = be.clip01(G + self.uv_boost * 0.15 * uv_saliency)
        rgb = [snow_glare_tone_compress(be, c, strength=self.snow_glare_compression) for c in (R, G, B)]  # :125
        if self.winter_mode:                                                            # :128-129, uv_helpers.py:100-107
            if self.scatter_sigma > 0.15:
                rgb = be.blur(rgb, self.scatter_sigma)
            rgb[2] = be.clip01(rgb[2] + float(self.blue_bias))
        be.encode(rgb)                                                                  # :132-133
