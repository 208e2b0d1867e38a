"""Shared criterion for the float32 UV pipelines whose last step is a uint8 encode (tests/test_uv_species_gpu.py,
tests/test_mantis_gpu.py, bench.py's parity_checked).

Contract: the baseline is bit-exact; the stylised frame is a float32 pipeline held to 1e-4
relative before the encode, i.e. codes within +-1 of the oracle's.  A sample further off than that is accepted ONLY where
the oracle's own arithmetic is unstable at float32 rounding level -- a categorical decision (mask, argmax, floor) within
rounding of its boundary, or a cancellation such as hummingbird.py:178-185's x / (x + y + z + 1e-8) with every term
at 1e-8.  That is established per frame, not assumed: the oracle is re-run `runs` times with every stage result,
reduction and transcendental jittered by a relative 2^-20 (oracle/np_backend.JitterBackend); the pixels whose output
code moves by >= 2 under that jitter, dilated by `dilate` pixels (a flipped pixel leaks into its neighbours through
the blurs that follow), are the only places where an outlier may sit.  On top of that the number of outlier pixels per
frame is capped in absolute terms."""
import numpy as np


def _dilate(mask: np.ndarray, r: int) -> np.ndarray:
    if r <= 0 or not mask.any():
        return mask
    out = mask.copy()
    for _ in range(r):  # r steps of the 3x3 structuring element
        m = out.copy()
        m[1:, :] |= out[:-1, :]
        m[:-1, :] |= out[1:, :]
        out = m.copy()
        m[:, 1:] |= out[:, :-1]
        m[:, :-1] |= out[:, 1:]
        out = m
    return out


def outlier_stats(got: np.ndarray, want: np.ndarray, rerun_jittered=None, *, runs: int = 6, dilate: int = 3) -> dict:
    """rerun_jittered(seed) -> the oracle's uint8 output under jitter (same shape as `want`)."""
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    px = diff.max(axis=2) if diff.ndim == 3 else diff
    out_px = px > 1
    st = {"frac_gt1": float((diff > 1).mean()), "frac_ne": float((diff > 0).mean()), "max": int(diff.max()), "outlier_px": int(out_px.sum()),
          "pixels": int(px.size)}
    if st["outlier_px"] and rerun_jittered is not None:
        unstable = np.zeros(px.shape, bool)
        for k in range(runs):
            j = rerun_jittered(k)
            dj = np.abs(j.astype(np.int16) - want.astype(np.int16))
            unstable |= (dj.max(axis=2) if dj.ndim == 3 else dj) >= 2
        cover = _dilate(unstable, dilate)
        st["unstable_px"] = int(unstable.sum())
        st["unexplained_px"] = int((out_px & ~cover).sum())
        if st["unexplained_px"]:
            ys, xs = np.nonzero(out_px & ~cover)
            st["unexplained_at"] = [(int(y), int(x), int(px[y, x])) for y, x in list(zip(ys, xs))[:8]]
    return st


def check_codes(got, want, what, rerun_jittered=None, *, frac_any: float = 0.05, frac_beyond: float = 2e-3, px_cap_abs: int = 16, px_cap_rel: float = 1e-3,
                runs: int = 6, dilate: int = 3):
    st = outlier_stats(got, want, rerun_jittered, runs=runs, dilate=dilate)
    assert st["frac_ne"] <= frac_any, (what, "fraction of samples != oracle:", st)
    assert st["frac_gt1"] <= frac_beyond, (what, "fraction of samples beyond +-1:", st)
    assert st["outlier_px"] <= px_cap_abs + px_cap_rel * st["pixels"], (what, "outlier pixels over the absolute cap:", st)
    if st["outlier_px"]:
        assert rerun_jittered is not None, (what, "samples beyond +-1 and no sensitivity probe given:", st)
        assert st["unexplained_px"] == 0, (what, "samples beyond +-1 where the oracle is stable under float32-level jitter:", st)
    return st
