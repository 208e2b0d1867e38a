"""The reference's uv_mappers.py by name -- (U, B, G) photoreceptor planes -> linear-RGB visualisations -- NumPy in /
NumPy out with everything O(pixels) on the device (plane programs of csrc/ew.hip + the exact radix-select percentile).

`HoneyBee.visualize` does not come through here: its fused route (csrc/uv.hip::k_map_encode) applies the same mappings
inside the frame's last launch.  This module is the step-wise surface for code written against the reference
(`from uv_mappers import map_opponent`, honeybee.py:11-19).  Same names, arguments, shapes, float32 results and errors;
arithmetic is float32 with the device's transcendentals (atan2f / powf / sqrtf: <= 2 ulp from NumPy's), percentiles
exact (np.percentile's linear interpolation, float32).  No CPU path."""
from __future__ import annotations

import numpy as np

from .uv import EPS_DEFAULT


def _run(planes, build):
    from .planevm import run_planes

    return np.stack(run_planes(planes, build), axis=2)


def _hsv_to_rgb(be, h, s, v):
    """uv_mappers.py:14-26 on Vals: sector i = floor(6 h) mod 6 picks (v, q, p, p, t, v) / (t, v, v, q, p, p) / (p, p, t, v, v, q)."""
    h6 = h * 6.0
    i = be.floor(h6)
    f = h6 - i
    p, q, t = v * (1.0 - s), v * (1.0 - f * s), v * (1.0 - (1.0 - f) * s)
    m = i - 6.0 * be.floor(i / 6.0)  # i % 6 (exact for the small integers involved)

    def pick(vals):  # nested selects over the sector number (integers: compare against half-way points)
        out = vals[5]
        for k in (4, 3, 2, 1, 0):
            out = be.where(m < k + 0.5, vals[k], out)
        return out

    return [pick([v, q, p, p, t, v]), pick([t, v, v, q, p, p]), pick([p, p, t, v, v, q])]


def hsv_to_rgb(hsv: np.ndarray) -> np.ndarray:
    """uv_mappers.py:14-26: HxWx3 HSV in [0, 1] -> HxWx3 RGB."""
    a = np.asarray(hsv)
    return _run([a[..., 0], a[..., 1], a[..., 2]], lambda be, c: _hsv_to_rgb(be, c[0], c[1], c[2])).astype(np.result_type(a.dtype, np.float32), copy=False)


def _check_planes(*planes):
    for p in planes:
        if p.ndim != 2 or p.shape != planes[0].shape:
            raise ValueError(f"planes must share one HxW shape, got {[q.shape for q in planes]}")


def _falsecolor(be, U, B, G, eps):
    def norm95(x):
        return x / be.maximum(be.percentile(x, 95.0), eps)

    Un, Bn, Gn = norm95(U), norm95(B), norm95(G)
    return [be.clip01(0.85 * Un + 0.10 * Gn), be.clip01(0.80 * Gn + 0.20 * Bn), be.clip01(0.70 * Bn + 0.40 * Un)]


def map_falsecolor(U: np.ndarray, B: np.ndarray, G: np.ndarray, eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:29-42: UV -> magenta bias, blue -> blue/cyan, green -> green/yellow; each plane / max(P95, eps)."""
    _check_planes(U, B, G)
    return _run([U, B, G], lambda be, c: _falsecolor(be, c[0], c[1], c[2], float(eps)))


def map_linear_matrix(U: np.ndarray, B: np.ndarray, G: np.ndarray, M: np.ndarray) -> np.ndarray:
    """uv_mappers.py:45-50: sRGB_linear = M . [U, B, G]^T (the FMA chain of the reference's `C @ M.T`)."""
    _check_planes(U, B, G)
    Mf = np.asarray(M, np.float32)
    if Mf.shape != (3, 3):
        raise ValueError("M must be 3x3")
    return _run([U, B, G], lambda be, c: [c[0] * float(Mf[i, 0]) + c[1] * float(Mf[i, 1]) + c[2] * float(Mf[i, 2]) for i in range(3)])


def map_opponent(U: np.ndarray, B: np.ndarray, G: np.ndarray, eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:53-64: hue = angle of (G - B, B - U), saturation = radius / (P95 + eps), value = mean / (P95 + eps)."""
    _check_planes(U, B, G)

    def build(be, c):
        Uv, Bv, Gv = c
        O1, O2 = be.mat(Gv - Bv), be.mat(Bv - Uv)
        L = be.mat(((Uv + Bv) + Gv) / 3.0)
        hue = (be.arctan2(O2, O1) + float(np.float32(np.pi))) / float(np.float32(2 * np.pi))
        radius = be.mat(be.sqrt(O1 * O1 + O2 * O2))
        sat = be.clip01(radius / (be.percentile(radius, 95.0) + float(eps)))
        val = be.clip01(L / (be.percentile(L, 95.0) + float(eps)))
        return _hsv_to_rgb(be, hue, sat, val)

    return _run([U, B, G], build)


def _s2l(rgb255) -> np.ndarray:
    v = np.array(rgb255, np.float32) / 255.0
    a = 0.055
    return np.where(v <= 0.04045, v / 12.92, ((v + a) / (1 + a)) ** 2.4).astype(np.float32)


def _plane_of(U: np.ndarray) -> np.ndarray:
    if U.ndim == 3 and U.shape[2] == 1:
        return U[..., 0]
    if U.ndim != 2:
        raise ValueError(f"U must be HxW or HxWx1, got {U.shape}")
    return U


def map_uv_purple_yellow(U: np.ndarray, eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:67-87: u = clip(U / max(P99, eps))^0.85 lerps purple (128, 0, 150) -> yellow (255, 225, 60), linear RGB."""
    U = _plane_of(U)
    c0, c1 = _s2l([128, 0, 150]), _s2l([255, 225, 60])

    def build(be, c):
        u = be.power(be.clip01(c[0] / be.maximum(be.percentile(c[0], 99.0), float(eps))), 0.85)
        return [be.clip01((1.0 - u) * float(c0[k]) + u * float(c1[k])) for k in range(3)]

    return _run([U], build)


def _purple_yellow_soft(be, Uv, u_gamma, accent_gamma, accent_strength, eps):
    c0, c1 = _s2l([176, 124, 232]), _s2l([255, 211, 138])
    u = be.power(be.clip01(Uv / be.maximum(be.percentile(Uv, 98.0), eps)), u_gamma)
    rgb = [(1.0 - u) * float(c0[k]) + u * float(c1[k]) for k in range(3)]
    if accent_strength > 0:
        w = be.power(u, accent_gamma)
        rgb = [rgb[k] + accent_strength * w * float(np.float32(c0[k] - np.float32(0.5))) for k in range(3)]
    Y = (0.2126 * rgb[0] + 0.7152 * rgb[1] + 0.0722 * rgb[2]) + eps
    gain = be.clip(be.clip01(0.22 + 0.55 * u) / Y, 0.6, 1.6)
    rgb = [v * gain for v in rgb]
    return [be.clip01(v / (1.0 + 0.6 * v)) for v in rgb]


def map_uv_purple_yellow_soft(U: np.ndarray, *, u_gamma: float = 0.90, accent_gamma: float = 0.85, accent_strength: float = 0.05,
                              eps: float = EPS_DEFAULT) -> np.ndarray:
    """uv_mappers.py:90-132: pastel purple -> warm lerp, purple accent, luminance retarget (gain clipped to [0.6, 1.6]), Reinhard x / (1 + 0.6 x)."""
    U = _plane_of(U)
    return _run([U], lambda be, c: _purple_yellow_soft(be, c[0], float(u_gamma), float(accent_gamma), float(accent_strength), float(eps)))


def map_falsecolor_uv_mixed(U: np.ndarray, B: np.ndarray, G: np.ndarray, alpha: float = 0.35) -> np.ndarray:
    """uv_mappers.py:135-144: (1 - alpha) falsecolor + alpha soft purple/yellow tint, divided by max(1, P99) when P99 > eps."""
    _check_planes(U, B, G)
    a = float(np.clip(alpha, 0.0, 1.0))

    def build(be, c):
        base = _falsecolor(be, c[0], c[1], c[2], EPS_DEFAULT)
        tint = _purple_yellow_soft(be, c[0], 0.90, 0.85, 0.05, EPS_DEFAULT)
        mixed = be.mat_all([(1.0 - a) * b + a * t for b, t in zip(base, tint)])
        p99 = be.percentile(mixed, 99.0)  # over the whole HxWx3 array
        scale = be.where(p99 > EPS_DEFAULT, be.maximum(p99, 1.0), 1.0)
        return [be.clip01(m / scale) for m in mixed]

    return _run([U, B, G], build)
