"""animals/hummingbird.py of the reference on the device: same constructor keywords and defaults (:38-103)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid
from .guppy import luma


def _s2l(rgb) -> np.ndarray:
    """hummingbird.py:84-87: sRGB byte triple -> linear float32."""
    v = np.array(rgb, np.float32) / 255.0
    a = 0.055
    return np.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4).astype(np.float32)


class Hummingbird(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (430.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        panorama_scale: float = 1.05,
        red_kill: float = 0.10,
        base_soft_sigma: float = 0.25,
        unsharp_sigma: float = 0.9,
        unsharp_amount: float = 0.24,
        combo_opacity: float = 0.55,
        combo_saturation: float = 0.45,
        combo_sheen: float = 0.28,
        tgt_uvb_srgb: Tuple[int, int, int] = (120, 150, 255),
        tgt_uvg_srgb: Tuple[int, int, int] = (110, 255, 170),
        tgt_uvr_srgb: Tuple[int, int, int] = (255, 110, 210),
        guide_sigma: float = 1.0,
        guide_gain: float = 0.25,
        periph_blur_sigma: float = 0.6,
        periph_radius: float = 0.82,
        periph_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.b_lo, self.b_hi = map(float, blue_band)
        self.g_lo, self.g_hi = map(float, green_band)
        self.r_lo, self.r_hi = map(float, red_band)
        self.panorama_scale = float(panorama_scale)
        self.red_kill = float(red_kill)
        self.base_soft_sigma = float(base_soft_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)
        self.combo_opacity = float(np.clip(combo_opacity, 0.0, 1.0))
        self.combo_saturation = float(combo_saturation)
        self.combo_sheen = float(combo_sheen)
        self.tgt_uvb_lin, self.tgt_uvg_lin, self.tgt_uvr_lin = _s2l(tgt_uvb_srgb), _s2l(tgt_uvg_srgb), _s2l(tgt_uvr_srgb)
        self.guide_sigma = float(guide_sigma)
        self.guide_gain = float(guide_gain)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)

    def render(self, be, image):
        """hummingbird.py:118-227: three UV x visible 'combo' maps, band-passed, tint the frame with synthetic hues."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :122-126
        bands = [(self.uv_lo, self.uv_hi), (self.b_lo, self.b_hi), (self.g_lo, self.g_hi), (self.r_lo, self.r_hi)]
        u_raw, b_raw, g_raw, r_raw = be.bands(self.lambdas, bands, self.hsi_scale)                        # :128-140
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bv, Gv, Rv = be.safe_norm(b_raw), be.safe_norm(g_raw), be.safe_norm(r_raw)
        combos = [be.safe_norm(U * Bv), be.safe_norm(U * Gv), be.safe_norm(U * Rv)]                       # :143-145
        small, large = be.blur(combos, 0.8), be.blur(combos, 2.0)                                         # :148-157 `bandpass`
        dogs = [be.clip01(m1 - m2) for m1, m2 in zip(small, large)]
        p95 = be.percentiles(dogs, 95.0)  # three independent order statistics: one set of radix passes
        UxB, UxG, UxR = [be.mat(be.clip01(d / (p + 1e-8))) for d, p in zip(dogs, p95)]
        rgb = [be.clip01(R * (1.0 - self.red_kill)), G, B]                                                # :160-163
        if self.base_soft_sigma > 0.0:
            rgb = be.blur(rgb, self.base_soft_sigma)
        combo_max = be.maximum(be.maximum(UxB, UxG), UxR)                                                 # :166-170
        if self.unsharp_sigma > 0.0 and self.unsharp_amount > 0.0:
            blurred = be.blur(rgb, self.unsharp_sigma)
            gain = self.unsharp_amount * combo_max
            rgb = [be.clip01(c + gain * be.clip(c - bl, -1.0, 1.0)) for c, bl in zip(rgb, blurred)]
        if self.combo_sheen > 0.0:                                                                        # :173-175
            sheen = 0.55 * UxB + 0.65 * UxG + 0.75 * UxR
            rgb = [be.clip01(c + self.combo_sheen * sheen) for c in rgb]
        w_sum = UxB + UxG + UxR + 1e-8                                                                    # :178-185
        wB, wG, wR = UxB / w_sum, UxG / w_sum, UxR / w_sum
        tint = [wB * tb + wG * tg + wR * tr for tb, tg, tr in zip(self.tgt_uvb_lin, self.tgt_uvg_lin, self.tgt_uvr_lin)]
        s = 1.0 + self.combo_saturation                                                                   # :186, :111-116
        if s != 1.0:
            Y = luma(tint)
            tint = [be.clip01(Y + (c - Y) * s) for c in tint]
        rgb = [be.clip01((1.0 - self.combo_opacity) * c + self.combo_opacity * t) for c, t in zip(rgb, tint)]  # :187
        if self.guide_gain > 0.0:                                                                         # :190-194
            U_s = be.blur([U], self.guide_sigma)[0]
            U_s = be.clip01(U_s / (be.percentile(U_s, 95.0) + 1e-8))
            g_ = self.guide_gain * U_s
            rgb = [be.clip01(c + g_ * k) for c, k in zip(rgb, np.array([0.20, 0.25, 0.10], np.float32))]
        if self.periph_blur_sigma > 0.0:                                                                  # :197-204
            periph = be.blur(rgb, self.periph_blur_sigma)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)
