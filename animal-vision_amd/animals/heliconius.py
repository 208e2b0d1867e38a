"""animals/heliconius.py of the reference on the device: same constructor keywords and defaults (:31-65)."""
from typing import Optional

import numpy as np

from ._uv_species import UVSpecies
from .guppy import luma


def sat_apply(be, rgb, scale):
    """heliconius.py:70-72 `_sat_apply`: Y + (c - Y) * scale, clipped."""
    Y = luma(rgb)
    return [be.clip01(Y + (c - Y) * scale) for c in rgb]


class Heliconius(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band=(320.0, 400.0),
        red_band=(600.0, 680.0),
        green_band=(500.0, 570.0),
        panorama_scale: float = 1.05,
        conj_sigma_small: float = 0.8,
        conj_sigma_large: float = 2.2,
        conj_gain: float = 1.0,
        sat_boost: float = 0.45,
        red_gain: float = 0.40,
        bg_desat: float = 0.20,
        bg_cool: float = 0.04,
        base_soft_sigma: float = 0.30,
        unsharp_sigma: float = 1.0,
        unsharp_amount: float = 0.25,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = np.asarray(lambdas, np.float32) if lambdas is not None else np.linspace(300, 700, 81)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.panorama_scale = float(panorama_scale)
        self.conj_sigma_small = float(conj_sigma_small)
        self.conj_sigma_large = float(conj_sigma_large)
        self.conj_gain = float(conj_gain)
        self.sat_boost = float(sat_boost)
        self.red_gain = float(red_gain)
        self.bg_desat = float(bg_desat)
        self.bg_cool = float(bg_cool)
        self.base_soft_sigma = float(base_soft_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)

    def render(self, be, image):
        """heliconius.py:75-135: the conjunction of UV and red band-passed spots is the signal everything keys on."""
        rgb = be.front(image, self.panorama_scale)                                                        # :79-83
        u_raw, r_raw = be.bands(self.lambdas, [(self.uv_lo, self.uv_hi), (self.red_lo, self.red_hi)], self.hsi_scale)  # :85-96 (Gv unused)
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Rb = be.mat(be.safe_norm(r_raw))
        uv_small, r_small = be.blur([U, Rb], self.conj_sigma_small)                                       # :99-107
        uv_large, r_large = be.blur([U, Rb], self.conj_sigma_large)
        conj = be.clip01(uv_small - uv_large) * be.clip01(r_small - r_large)
        conj = be.clip01(conj / (be.percentile(conj, 95.0) + 1e-8))
        if self.base_soft_sigma > 0.0:                                                                    # :110-111
            rgb = be.blur(rgb, self.base_soft_sigma)
        bg_w = 1.0 - conj                                                                                 # :114-117
        rgb = [rgb[0], rgb[1], be.clip01(rgb[2] + self.bg_cool * bg_w)]
        rgb = sat_apply(be, rgb, 1.0 - self.bg_desat * bg_w)
        if self.unsharp_sigma > 0.0 and self.unsharp_amount > 0.0:                                        # :120-122
            blurred = be.blur(rgb, self.unsharp_sigma)
            gain = self.unsharp_amount * conj
            rgb = [be.clip01(c + gain * (c - b)) for c, b in zip(rgb, blurred)]
        rgb = [be.clip01(rgb[0] + self.red_gain * conj), rgb[1], rgb[2]]                                  # :125-126
        rgb = sat_apply(be, rgb, 1.0 + self.sat_boost * conj)
        be.encode(rgb)
