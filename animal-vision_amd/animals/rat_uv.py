"""animals/rat_uv.py of the reference on the device: same constructor keywords and defaults (:57-97), and the
one `visualize(image, *, mode="auto")` outlier signature (:131)."""
from typing import Literal, Optional, Tuple

import numpy as np

from ._uv_species import UVSpecies, snow_glare_tone_compress

Mode = Literal["auto", "day", "night"]


class RatUV(UVSpecies):
    DEFAULT_LAMBDAS = np.linspace(320.0, 700.0, 129, dtype=np.float64)
    UV_BAND = (330.0, 400.0)
    B_BAND = (400.0, 500.0)
    G_BAND = (500.0, 600.0)

    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.55,
        panorama_scale: float = 1.45,
        uv_boost_alpha: float = 0.55,
        day_blur_sigma: float = 0.8,
        night_blur_sigma: float = 1.25,
        blue_bias_day: float = 0.03,
        blue_bias_night: float = 0.05,
        tone_knee: float = 0.82,
        tone_strength: float = 0.65,
        ground_vignette_day: float = 0.10,
        ground_vignette_night: float = 0.14,
    ):
        if lambdas is None:
            self.lambdas = self.DEFAULT_LAMBDAS.copy()
        else:
            wl = np.asarray(lambdas, dtype=np.float64).ravel()
            self.lambdas = np.linspace(float(wl[0]), float(wl[-1]), wl.size, dtype=np.float64)
        self.hsi_scale = float(hsi_scale)
        self.panorama_scale = float(panorama_scale)
        self.uv_boost_alpha = float(np.clip(uv_boost_alpha, 0.0, 1.0))
        self.day_blur_sigma = float(day_blur_sigma)
        self.night_blur_sigma = float(night_blur_sigma)
        self.blue_bias_day = float(blue_bias_day)
        self.blue_bias_night = float(blue_bias_night)
        self.tone_knee = float(tone_knee)
        self.tone_strength = float(tone_strength)
        self.ground_vignette_day = float(ground_vignette_day)
        self.ground_vignette_night = float(ground_vignette_night)
        self._uv_band, self._b_band, self._g_band = self.UV_BAND, self.B_BAND, self.G_BAND

    def variant(self, image, probes, mode: Mode = "auto"):
        """rat_uv.py:99-105 `_choose_mode`: the median luminance of the sRGB frame decides day / night."""
        if mode != "auto":
            return mode
        return "night" if probes.median_luma(image) < 0.12 else "day"

    def visualize(self, image: np.ndarray, *, mode: Mode = "auto") -> Optional[Tuple[np.ndarray, np.ndarray]]:
        return super().visualize(image, mode=mode)

    def render(self, be, image, mode_eff: str):
        """rat_uv.py:131-214."""
        base = be.front(image, self.panorama_scale)                                                     # :148-159
        # :163, _classic_rgb_to_hsi_scaled_nocast (:116-128): the analytic converter casts the wavelengths to float32 itself
        u_raw, b_raw, g_raw = be.bands(self.lambdas, [self._uv_band, self._b_band, self._g_band], self.hsi_scale)
        U = be.safe_norm(u_raw)                                                                         # :166-168

        def norm95(x):                                                                                  # :173-174
            return x / be.maximum(1e-8, be.percentile(x, 95.0))

        U_n, B_n, G_n = norm95(U), norm95(b_raw), norm95(g_raw)
        false_lin = [be.clip01(0.85 * U_n + 0.10 * G_n), be.clip01(0.80 * G_n + 0.20 * B_n), be.clip01(0.70 * B_n + 0.40 * U_n)]  # :177-184
        a = self.uv_boost_alpha
        rgb = [be.clip01((1.0 - a) * c + a * f) for c, f in zip(base, false_lin)]                       # :187-188
        night = mode_eff == "night"
        blur_sigma = self.night_blur_sigma if night else self.day_blur_sigma                            # :191-193
        blue_bias = self.blue_bias_night if night else self.blue_bias_day
        if blur_sigma > 0.15:                                                                           # uv_helpers.py:100-107
            rgb = be.blur(rgb, blur_sigma)
        rgb[2] = be.clip01(rgb[2] + float(blue_bias))
        if not night:                                                                                   # :199-205
            rgb = [snow_glare_tone_compress(be, c, strength=self.tone_strength, knee=self.tone_knee) for c in rgb]
        else:
            Y = 0.2126 * rgb[0] + 0.7152 * rgb[1] + 0.0722 * rgb[2]
            gain = (Y + 0.18) / (Y + 1e-6)
            rgb = [be.clip01(c * gain) for c in rgb]
        gv = self.ground_vignette_night if night else self.ground_vignette_day                          # :208-209, :107-113
        yy = np.linspace(0.0, 1.0, be.H, dtype=np.float32)
        mask = np.clip(1.0 - yy, 0.0, 1.0)
        gain_rows = be.row(1.0 - gv * (1.0 - mask))
        be.encode([be.clip01(c * gain_rows) for c in rgb])                                              # :212-213
