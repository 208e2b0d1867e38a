"""animals/jumping_spider.py of the reference on the device: same constructor keywords and defaults (:36-101)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class JumpingSpider(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        blue_band: Tuple[float, float] = (430.0, 500.0),
        panorama_scale: float = 1.02,
        dog_small_sigma: float = 0.9,
        dog_large_sigma: float = 2.2,
        uv_patch_gain: float = 0.95,
        opponent_gain: float = 0.30,
        red_kill: float = 0.25,
        base_soft_sigma: float = 0.25,
        clarity_sigma: float = 0.9,
        clarity_amount: float = 0.24,
        fovea_radius: float = 0.38,
        fovea_softness: float = 10.0,
        periph_blur_sigma: float = 2.2,
        periph_vignette_strength: float = 0.22,
        scan_row_freq: float = 22.0,
        scan_row_gain: float = 0.08,
        scan_soften: float = 0.9,
        spots: Tuple[Tuple[float, float], ...] = ((0.50, 0.52), (0.57, 0.48)),
        spot_sigma: float = 0.08,
        spot_gain: float = 0.20,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.g_lo, self.g_hi = map(float, green_band)
        self.r_lo, self.r_hi = map(float, red_band)
        self.b_lo, self.b_hi = map(float, blue_band)
        self.panorama_scale = float(panorama_scale)
        self.dog_small_sigma = float(dog_small_sigma)
        self.dog_large_sigma = float(dog_large_sigma)
        self.uv_patch_gain = float(uv_patch_gain)
        self.opponent_gain = float(opponent_gain)
        self.red_kill = float(red_kill)
        self.base_soft_sigma = float(base_soft_sigma)
        self.clarity_sigma = float(clarity_sigma)
        self.clarity_amount = float(clarity_amount)
        self.fovea_radius = float(fovea_radius)
        self.fovea_softness = float(fovea_softness)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_vignette_strength = float(periph_vignette_strength)
        self.scan_row_freq = float(scan_row_freq)
        self.scan_row_gain = float(scan_row_gain)
        self.scan_soften = float(scan_soften)
        self.spots = tuple((float(y), float(x)) for (y, x) in spots)
        self.spot_sigma = float(spot_sigma)
        self.spot_gain = float(spot_gain)

    def _attention_spots(self, be):
        """jumping_spider.py:110-120: a few Gaussian spots near the centre, normalised by their own P95."""
        yy = be.row(np.linspace(0.0, 1.0, be.H, dtype=np.float32))
        xx = be.col(np.linspace(0.0, 1.0, be.W, dtype=np.float32))
        s2 = max(self.spot_sigma, 1e-4) ** 2
        mask = None
        for yc, xc in self.spots:
            e = be.exp(-((yy - yc) ** 2 + (xx - xc) ** 2) / (2.0 * s2))
            mask = e if mask is None else mask + e
        m95 = be.maximum(1e-8, be.percentile(mask, 95.0))
        return be.mat(be.clip01(mask / m95))

    def render(self, be, image):
        """jumping_spider.py:123-236: UV patches + green/UV opponency, scanlines, attention spots, strong foveation."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :127-131
        bands = [(self.uv_lo, self.uv_hi), (self.g_lo, self.g_hi), (self.b_lo, self.b_hi)]
        u_raw, g_raw, b_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :133-145 (Rv is never used)
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Gv, Bv = be.safe_norm(g_raw), be.safe_norm(b_raw)
        rgb = [be.clip01(R * (1.0 - self.red_kill)), G, B]                                                # :148-151
        if self.base_soft_sigma > 0.0:
            rgb = be.blur(rgb, self.base_soft_sigma)
        uv_small = be.blur([U], self.dog_small_sigma)[0]                                                  # :154-158
        uv_large = be.blur([U], self.dog_large_sigma)[0]
        uv_dog = be.clip01(uv_small - uv_large)
        uv_patch = be.clip01(uv_dog / (be.percentile(uv_dog, 95.0) + 1e-8))
        opp = Gv - U                                                                                      # :161-165
        opp = be.mat(be.clip(opp / (be.percentile(be.abs(opp), 95.0) + 1e-8), -1.0, 1.0))
        g_boost = be.clip01(opp) * self.opponent_gain
        u_boost = be.clip01(-opp) * self.opponent_gain
        R, G, B = rgb                                                                                     # :167-169
        G = be.clip01(G + 0.40 * g_boost)
        B = be.clip01(B + 0.30 * u_boost * Bv)
        R = be.clip01(R + 0.12 * u_boost * U)
        rgb = [R, G, B]
        if self.clarity_sigma > 0.0 and self.clarity_amount > 0.0:                                        # :172-175
            blurred = be.blur(rgb, self.clarity_sigma)
            gain = self.clarity_amount * self.uv_patch_gain * uv_patch
            rgb = [be.clip01(c + gain * be.clip(c - bl, -1.0, 1.0)) for c, bl in zip(rgb, blurred)]
        if self.scan_row_gain != 0.0:                                                                     # :178-186
            y = np.linspace(0.0, 1.0, be.H, dtype=np.float32)
            rows = be.row(0.5 + 0.5 * np.sin(2.0 * np.pi * self.scan_row_freq * y))
            if self.scan_soften > 0.0:
                rows = be.blur([rows * 1.0], self.scan_soften)[0]
            row_gain = 1.0 + self.scan_row_gain * (rows - 0.5)
            rgb = [be.clip01(c * row_gain) for c in rgb]
        if self.spot_gain > 0.0:                                                                          # :189-194
            spots = self._attention_spots(be)
            rgb = [be.clip01(c + self.spot_gain * spots) for c in rgb]
            blur = be.blur(rgb, 0.8)
            sharp = [be.clip01(c + 0.25 * be.clip(c - b, -1.0, 1.0)) for c, b in zip(rgb, blur)]
            rgb = [be.clip01((1.0 - 0.6 * spots) * c + (0.6 * spots) * s_) for c, s_ in zip(rgb, sharp)]
        if self.periph_blur_sigma > 0.0 or self.periph_vignette_strength > 0.0:                           # :197-210
            edge_w = radial_sigmoid(be, radius=self.fovea_radius, softness=self.fovea_softness)
            if self.periph_blur_sigma > 0.0:
                periph = be.blur(rgb, self.periph_blur_sigma)
                rgb = [(1.0 - edge_w) * c + edge_w * p for c, p in zip(rgb, periph)]
            if self.periph_vignette_strength > 0.0:
                vign = 1.0 - self.periph_vignette_strength * edge_w
                rgb = [be.clip01(c * vign) for c in rgb]
        be.encode(rgb)
