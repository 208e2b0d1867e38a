"""animals/anableps.py of the reference on the device: same constructor keywords and defaults (:38-104)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class Anableps(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (430.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        panorama_scale: float = 1.20,
        horizon_y: float = 0.44,
        seam_softness_px: float = 8.0,
        ripple_amp_px: float = 6.0,
        ripple_waves: float = 2.5,
        refract_push_px: float = 3.0,
        air_warmth: Tuple[float, float, float] = (1.06, 1.03, 0.99),
        air_clarity_unsharp: float = 0.35,
        air_unsharp_sigma: float = 1.0,
        red_kill: float = 0.55,
        blue_lift: float = 0.08,
        green_lift: float = 0.12,
        haze_strength: float = 0.10,
        haze_tint: Tuple[float, float, float] = (0.80, 0.92, 1.00),
        base_blur_sigma_water: float = 0.7,
        uv_boost: float = 3.4,
        uv_R_gain: float = 0.36,
        uv_G_gain: float = 0.18,
        uv_B_gain: float = 0.42,
        periph_blur_sigma: float = 1.2,
        periph_radius: float = 0.70,
        periph_softness: float = 6.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.red_lo, self.red_hi = map(float, red_band)
        self.panorama_scale = float(panorama_scale)
        self.horizon_y = float(horizon_y)
        self.seam_softness_px = float(seam_softness_px)
        self.ripple_amp_px = float(ripple_amp_px)
        self.ripple_waves = float(ripple_waves)
        self.refract_push_px = float(refract_push_px)
        self.air_warmth = np.array(air_warmth, dtype=np.float32)
        self.air_clarity_unsharp = float(air_clarity_unsharp)
        self.air_unsharp_sigma = float(air_unsharp_sigma)
        self.red_kill = float(red_kill)
        self.blue_lift = float(blue_lift)
        self.green_lift = float(green_lift)
        self.haze_strength = float(haze_strength)
        self.haze_tint = np.array(haze_tint, dtype=np.float32)
        self.base_blur_sigma_water = float(base_blur_sigma_water)
        self.uv_boost = float(uv_boost)
        self.uv_R_gain = float(uv_R_gain)
        self.uv_G_gain = float(uv_G_gain)
        self.uv_B_gain = float(uv_B_gain)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)

    def render(self, be, image):
        """anableps.py:113-255: air above a rippled horizon, water below, blended by a vertical sigmoid."""
        base = be.front(image, self.panorama_scale)                                                       # :129-140
        bands = [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi)]
        u_raw, b_raw, g_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :142-157 (Rv is never used)
        Bv, Gv = be.safe_norm(b_raw), be.safe_norm(g_raw)
        Un = be.safe_norm(be.safe_norm(u_raw))
        H, W = be.H, be.W
        y0 = int(np.clip(self.horizon_y * H, 0, H - 1))                                                   # :161-176
        if self.ripple_amp_px > 0.0:
            x = np.linspace(0, 2.0 * np.pi * self.ripple_waves, W, dtype=np.float32)
            ripple = (self.ripple_amp_px * np.sin(x)).astype(np.float32)
        else:
            ripple = np.zeros((W,), np.float32)
        yy = be.row(np.arange(H, dtype=np.float32))
        seam_soft = max(1.0, float(self.seam_softness_px))
        horizon = be.col(y0 + ripple)
        dist = yy - horizon
        air_w = 1.0 / (1.0 + be.exp(dist / seam_soft))
        water_w = 1.0 - air_w
        air = [be.clip01(c * w) for c, w in zip(base, self.air_warmth)]                                   # :181-183, :106-111
        if self.air_unsharp_sigma > 0.0 and self.air_clarity_unsharp > 0.0:
            blur = be.blur(air, self.air_unsharp_sigma)
            air = [be.clip01(c + self.air_clarity_unsharp * be.clip(c - b, -1.0, 1.0)) for c, b in zip(air, blur)]
        R, G, B = base                                                                                    # :186-189
        water = [be.clip01(R * (1.0 - self.red_kill)), be.clip01(G + self.green_lift), be.clip01(B + self.blue_lift)]
        if self.haze_strength > 0.0:                                                                      # :191-193
            a = np.clip(self.haze_strength, 0.0, 1.0)  # np.float64, as in the reference
            water = [(1.0 - a) * c + a * t for c, t in zip(water, self.haze_tint)]
        if self.base_blur_sigma_water > 0.0:                                                              # :195-196
            water = be.blur(water, self.base_blur_sigma_water)
        wR, wG, wB = water
        wR = be.clip01(wR + self.uv_boost * self.uv_R_gain * Un)                                          # :199-202
        wG = be.clip01(wG + self.uv_boost * self.uv_G_gain * Un)
        wB = be.clip01(wB + self.uv_boost * self.uv_B_gain * Un)
        wB = be.clip01(wB + 0.20 * Bv)                                                                    # :205-206
        wG = be.clip01(wG + 0.26 * Gv)
        water = [wR, wG, wB]
        if self.refract_push_px > 0.0:                                                                    # :209-226 (cv2 present)
            y_idx, x_idx = yy, be.col(np.arange(W, dtype=np.float32))
            below = be.maximum(y_idx - horizon, 0.0)
            push = self.refract_push_px * be.exp(-below / (2.5 * self.seam_softness_px))
            map_y = be.clip(y_idx + push, 0, H - 1)
            water = be.remap(water, x_idx, map_y)  # BORDER_REFLECT101 in the reference: unreachable, map_y is clipped to the frame
        rgb = [a_ * air_w + w_ * water_w for a_, w_ in zip(air, water)]                                   # :229
        if self.periph_blur_sigma > 0.0:                                                                  # :232-240
            periph = be.blur(rgb, self.periph_blur_sigma)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)
