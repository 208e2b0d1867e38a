"""animals/morpho.py of the reference on the device: same constructor keywords and defaults (:36-63)."""
from typing import Optional

import numpy as np

from ..geometry import INTER_AREA, INTER_NEAREST
from ._uv_species import UVSpecies, radial_sigmoid


class Morpho(UVSpecies):
    def __init__(
        self,
        *,
        lambdas: Optional[np.ndarray] = None,
        hsi_scale: float = 0.25,
        uv_band=(320.0, 400.0),
        blue_band=(440.0, 500.0),
        green_band=(500.0, 570.0),
        panorama_scale: float = 1.05,
        sheen_strength: float = 0.55,
        hue_shift_strength: float = 0.45,
        gloss_sigma: float = 1.0,
        mosaic_downscale: float = 0.35,
        center_clarity: float = 0.25,
        vignette_softness: float = 7.0,
        vignette_radius: float = 0.82,
    ):
        self.hsi_scale = float(hsi_scale)
        self.lambdas = np.asarray(lambdas, np.float32) if lambdas is not None else np.linspace(300, 700, 81)
        self.uv_lo, self.uv_hi = map(float, uv_band)
        self.blue_lo, self.blue_hi = map(float, blue_band)
        self.green_lo, self.green_hi = map(float, green_band)
        self.panorama_scale = float(panorama_scale)
        self.sheen_strength = float(sheen_strength)
        self.hue_shift_strength = float(hue_shift_strength)
        self.gloss_sigma = float(gloss_sigma)
        self.mosaic_downscale = float(np.clip(mosaic_downscale, 0.15, 1.0))
        self.center_clarity = float(center_clarity)
        self.vignette_softness = float(vignette_softness)
        self.vignette_radius = float(vignette_radius)

    def render(self, be, image):
        """morpho.py:95-154: orientation-selected blue/cyan shift, UV gloss, 'ommatidial' mosaic, sharp centre."""
        R, G, B = be.front(image, self.panorama_scale)                                                    # :99-103
        u_raw, b_raw, g_raw = be.bands(self.lambdas, [(self.uv_lo, self.uv_hi), (self.blue_lo, self.blue_hi), (self.green_lo, self.green_hi)],
                                       self.hsi_scale)                                                   # :105-116 (Gv is never used)
        U = be.mat(be.safe_norm(be.safe_norm(u_raw)))
        Bv = be.mat(be.safe_norm(b_raw))
        gx, gy = be.sobel(Bv)                                                                             # :121-123
        ori = be.arctan2(gy, gx)
        align = 0.5 * (1.0 + be.cos(2.0 * ori))
        gloss = be.blur([U], self.gloss_sigma)[0]                                                         # :126-128
        gloss = be.clip01(gloss / (be.percentile(gloss, 95.0) + 1e-8))
        shift_cyan = self.hue_shift_strength * align                                                      # :131-134
        shift_deep = self.hue_shift_strength * (1.0 - align)
        B = be.clip01(B + 0.40 * shift_deep + 0.25 * shift_cyan)
        G = be.clip01(G + 0.35 * shift_cyan)
        sheen = self.sheen_strength * gloss                                                               # :137
        rgb = [be.clip01(c + sheen * k) for c, k in zip((R, G, B), np.array([0.10, 0.25, 0.45], np.float32))]
        if self.mosaic_downscale < 0.999:                                                                 # :140, :84-92
            h = max(1, int(round(be.H * self.mosaic_downscale)))
            w = max(1, int(round(be.W * self.mosaic_downscale)))
            rgb = be.down_up(rgb, h, w, INTER_AREA, INTER_NEAREST)
        t = radial_sigmoid(be, radius=self.vignette_radius, softness=self.vignette_softness)              # :143-151
        blur = be.blur(rgb, 1.0)
        rgb = [be.clip01((1.0 - t) * (c + 0.22 * (c - b)) + t * c) for c, b in zip(rgb, blur)]
        be.encode(rgb)
