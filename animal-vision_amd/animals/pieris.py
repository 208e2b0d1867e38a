"""animals/pieris.py of the reference on the device: same constructor keywords and defaults (:31-63)."""
from typing import Optional

import numpy as np

from ._uv_species import UVSpecies, radial_sigmoid


class Pieris(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band=(320.0, 400.0),
        blue_band=(430.0, 500.0),
        green_band=(500.0, 570.0),
        panorama_scale: float = 1.05,
        guide_sigma: float = 1.2,
        guide_gain: float = 0.75,
        foliage_opponent_gain: float = 0.25,
        petal_warmth: float = 0.08,
        clarity_unsharp_sigma: float = 0.8,
        clarity_amount: float = 0.22,
        center_bias: float = 0.12,
        bias_radius: float = 0.80,
        bias_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = np.asarray(lambdas, np.float32) if lambdas is not None else np.linspace(300, 700, 81)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.panorama_scale = float(panorama_scale)
        self.guide_sigma = float(guide_sigma)
        self.guide_gain = float(guide_gain)
        self.foliage_opponent_gain = float(foliage_opponent_gain)
        self.petal_warmth = float(petal_warmth)
        self.clarity_unsharp_sigma = float(clarity_unsharp_sigma)
        self.clarity_amount = float(clarity_amount)
        self.center_bias = float(center_bias)
        self.bias_radius = float(bias_radius)
        self.bias_softness = float(bias_softness)

    def render(self, be, image):
        """pieris.py:66-124: blurred UV 'nectar guides' brighten toward warm white; green-vs-(UV+blue) opponent lift."""
        rgb = be.front(image, self.panorama_scale)                                                        # :70-74
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi)]
        u_raw, b_raw, g_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :76-87
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bv, Gv = be.safe_norm(b_raw), be.safe_norm(g_raw)
        U_s = be.blur([U], self.guide_sigma)[0]                                                           # :92-95
        U_s = be.clip01(U_s / (be.percentile(U_s, 95.0) + 1e-8))
        guide_w = self.guide_gain * U_s
        tint = np.array([0.35, 0.35 + self.petal_warmth, 0.25], np.float32)                               # :96
        rgb = [be.clip01(c + guide_w * k) for c, k in zip(rgb, tint)]
        foliage = be.clip01(Gv - 0.5 * (U + Bv))                                                          # :99-100
        rgb[1] = be.clip01(rgb[1] + self.foliage_opponent_gain * foliage)
        if self.clarity_unsharp_sigma > 0.0 and self.clarity_amount > 0.0:                                # :103-105
            blur = be.blur(rgb, self.clarity_unsharp_sigma)
            rgb = [be.clip01(c + self.clarity_amount * (c - b)) for c, b in zip(rgb, blur)]
        t = radial_sigmoid(be, radius=self.bias_radius, softness=self.bias_softness)                      # :108-114
        att = 1.0 + self.center_bias * (1.0 - t)
        be.encode([be.clip01(c * att) for c in rgb])
