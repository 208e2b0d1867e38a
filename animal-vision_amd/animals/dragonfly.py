"""animals/dragonfly.py of the reference on the device: same constructor keywords and defaults (:39-113)."""
from typing import Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, default_lambdas, radial_sigmoid


class Dragonfly(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band: Tuple[float, float] = (320.0, 400.0),
        blue_band: Tuple[float, float] = (440.0, 500.0),
        green_band: Tuple[float, float] = (500.0, 570.0),
        red_band: Tuple[float, float] = (600.0, 680.0),
        panorama_scale: float = 1.15,
        sky_prior_strength: float = 0.6,
        sky_blue_weight: float = 0.4,
        sky_sigmoid_mid: float = 0.46,
        sky_sigmoid_steepness: float = 6.0,
        sky_pol_strength: float = 0.65,
        sky_pol_gamma: float = 1.3,
        water_pol_strength: float = 0.55,
        water_pol_gamma: float = 1.2,
        sky_evec_base_deg: float = 90.0,
        sky_evec_sweep_deg: float = -45.0,
        red_kill: float = 0.22,
        sky_uv_blue_gain: Tuple[float, float] = (0.25, 0.20),
        water_uv_blue_gain: Tuple[float, float] = (0.30, 0.24),
        ventral_green_gain: float = 0.12,
        base_soft_sigma: float = 0.30,
        unsharp_sigma: float = 1.0,
        unsharp_amount: float = 0.30,
        highlight_knee: float = 0.85,
        highlight_strength: float = 0.35,
        periph_blur_sigma: float = 0.7,
        periph_radius: float = 0.80,
        periph_softness: float = 7.0,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = default_lambdas(lambdas)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.b_lo, self.b_hi = map(float, blue_band)
        self.g_lo, self.g_hi = map(float, green_band)
        self.r_lo, self.r_hi = map(float, red_band)
        self.panorama_scale = float(panorama_scale)
        self.sky_prior_strength = float(sky_prior_strength)
        self.sky_blue_weight = float(sky_blue_weight)
        self.sky_sigmoid_mid = float(sky_sigmoid_mid)
        self.sky_sigmoid_steepness = float(sky_sigmoid_steepness)
        self.sky_pol_strength = float(sky_pol_strength)
        self.sky_pol_gamma = float(sky_pol_gamma)
        self.water_pol_strength = float(water_pol_strength)
        self.water_pol_gamma = float(water_pol_gamma)
        self.sky_evec_base = np.deg2rad(float(sky_evec_base_deg))
        self.sky_evec_sweep = np.deg2rad(float(sky_evec_sweep_deg))
        self.red_kill = float(red_kill)
        self.sky_uv_blue_gain = tuple(map(float, sky_uv_blue_gain))
        self.water_uv_blue_gain = tuple(map(float, water_uv_blue_gain))
        self.ventral_green_gain = float(ventral_green_gain)
        self.base_soft_sigma = float(base_soft_sigma)
        self.unsharp_sigma = float(unsharp_sigma)
        self.unsharp_amount = float(unsharp_amount)
        self.highlight_knee = float(highlight_knee)
        self.highlight_strength = float(highlight_strength)
        self.periph_blur_sigma = float(periph_blur_sigma)
        self.periph_radius = float(periph_radius)
        self.periph_softness = float(periph_softness)

    def _soft_knee(self, be, x, knee: float, amount: float):
        """dragonfly.py:126-135."""
        if amount <= 0.0:
            return x
        x = be.clip01(x)
        t = (x - knee) / (1.0 - knee + 1e-8)
        return be.where(x <= knee, x, knee + (1.0 - knee) * (t / (1.0 + amount * t)))

    def render(self, be, image):
        """dragonfly.py:138-251: sky / water polarisation gains from the local orientation of a UV-blue map."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :142-146
        bands = [(self.uv_lo, self.uv_hi), (self.b_lo, self.b_hi), (self.g_lo, self.g_hi)]
        u_raw, b_raw, g_raw = be.bands(self.lambdas, bands, self.hsi_scale)                               # :148-160 (Rv is never used)
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bv, Gv = be.mat(be.safe_norm(b_raw)), be.mat(be.safe_norm(g_raw))
        H = be.H
        vert_prior = be.row(np.linspace(1.0, 0.0, H, dtype=np.float32))                                   # :164-172
        blue_dom = be.clip01(Bv - 0.6 * Gv)
        score = be.blur([self.sky_prior_strength * vert_prior + self.sky_blue_weight * blue_dom], 2.5)[0]
        score = score / (be.percentile(score, 98.0) + 1e-8)
        sky_w = be.mat(1.0 / (1.0 + be.exp(-self.sky_sigmoid_steepness * (score - self.sky_sigmoid_mid))))
        ground_w = 1.0 - sky_w
        gx, gy = be.sobel(0.6 * Bv + 0.4 * U)                                                             # :178-180
        theta = be.arctan2(gy, gx)
        y_norm = np.linspace(0.0, 1.0, H, dtype=np.float32)                                               # :183-184
        sky_evec = self.sky_evec_base + self.sky_evec_sweep * y_norm  # float64 rows, as in the reference
        cos2_local, sin2_local = be.mat(be.cos(2.0 * theta)), be.mat(be.sin(2.0 * theta))                 # :187-194
        cos2_sky, sin2_sky = be.row(np.cos(2.0 * sky_evec)), be.row(np.sin(2.0 * sky_evec))
        align_sky = cos2_local * cos2_sky + sin2_local * sin2_sky
        align_sky01 = be.mat(be.clip01(0.5 * (align_sky + 1.0)) ** self.sky_pol_gamma)
        align_water = cos2_local * 1.0 + sin2_local * 0.0                                                 # :197-199
        align_water01 = be.mat(be.clip01(0.5 * (align_water + 1.0)) ** self.water_pol_gamma)
        rgb = [be.clip01(R * (1.0 - self.red_kill)), G, B]                                                # :202-205
        if self.base_soft_sigma > 0.0:
            rgb = be.blur(rgb, self.base_soft_sigma)
        sky_gain = 1.0 + self.sky_pol_strength * (align_sky01 * sky_w)                                    # :208-212
        R, G, B = [be.clip01(c * (0.95 + 0.05 * sky_w)) for c in rgb]
        B = be.clip01(B + self.sky_uv_blue_gain[1] * (Bv * sky_w * align_sky01))
        G = be.clip01(G + 0.10 * (U * sky_w * align_sky01))
        R, G, B = [be.clip01(c * sky_gain) for c in (R, G, B)]
        water_gain = 1.0 + self.water_pol_strength * (align_water01 * ground_w)                            # :215-221
        B = be.clip01(B + self.water_uv_blue_gain[1] * (Bv * ground_w * align_water01))
        B = be.clip01(B + self.water_uv_blue_gain[0] * (U * ground_w * align_water01))
        G = be.clip01(G + self.ventral_green_gain * (Gv * ground_w))
        rgb = be.mat_all([be.clip01(c * water_gain) for c in (R, G, B)])
        if self.unsharp_sigma > 0.0 and self.unsharp_amount > 0.0:                                        # :224-227
            blur = be.blur(rgb, self.unsharp_sigma)
            rgb = [be.clip01(c + self.unsharp_amount * be.clip(c - b, -1.0, 1.0)) for c, b in zip(rgb, blur)]
        rgb = [self._soft_knee(be, c, self.highlight_knee, self.highlight_strength) for c in rgb]         # :230
        if self.periph_blur_sigma > 0.0:                                                                  # :233-239
            periph = be.blur(rgb, self.periph_blur_sigma)
            t = radial_sigmoid(be, radius=self.periph_radius, softness=self.periph_softness)
            rgb = [(1.0 - t) * c + t * p for c, p in zip(rgb, periph)]
        be.encode(rgb)
