"""CPU-only: host logic of the product vs golden vectors / oracle, and the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

import animal_vision_amd as av
from animal_vision_amd import dichromat as D


def test_library_exports_every_declared_symbol():
    """Every function include/avx.h declares is exported by libavx.so (no compute calls here)."""
    hdr = open(os.path.join(ROOT, "include", "avx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(avx_[a-z0-9_]+)\s*\(", hdr))
    assert {"avx_init", "avx_dichromat_u8", "avx_last_error"} <= names
    lib = ctypes.CDLL(av._lib.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, f"declared in include/avx.h but not exported: {missing}"
    assert lib.avx_abi_version() == 1


def test_no_gpu_fails_loudly():
    if av.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(av.AvxError) as e:
        av.Context(0)
    assert e.value.code == av._lib.AVX_ERR_NO_DEVICE
    from animal_vision_amd.animals import Dog

    with pytest.raises(av.AvxError):
        Dog().visualize(np.zeros((4, 4, 3), np.uint8))


def test_compiled_tables_are_the_reference_values():
    g = load_golden("srgb_tables")
    assert np.array_equal(av.get_table(0), g["decode_lut"])
    assert np.array_equal(av.get_table(1), g["enc_thr_f32"])
    assert np.array_equal(av.get_table(2), g["enc_thr_f64"])


def test_collapse_matrix_matches_reference():
    g = load_golden("collapse_matrices")
    for (a, s), T in zip(g["pairs"].tolist(), g["T"]):
        assert np.array_equal(D.collapse_LMS_matrix(a, s), T)


def test_tap_generation_matches_oracle(oracle):
    for sigma in (0.2, 0.7, 1.0, 1.2, 1.3, 1.4, 1.6, 1.8, 2.0, 3.5):
        k = D.cv_auto_ksize(sigma)
        assert k == oracle.cv_auto_ksize(sigma)
        assert np.array_equal(D.gaussian_taps(k, sigma), oracle.gaussian_kernel(k, sigma, np.float64))
    # sigma -> ksize table
    assert [D.cv_auto_ksize(s) for s in (3.5, 1.0, 1.2, 1.4, 2.0, 0.7)] == [29, 9, 11, 13, 17, 7]


def test_row_gain_matches_oracle(oracle):
    for H in (1, 2, 24, 1080):
        assert np.array_equal(D.s_cone_row_gain(H, 1.3, 0.5, power=1.4, extra_boost=0.25),
                              oracle.s_cone_row_gain(H, 1.3, 0.5, power=1.4, extra_boost=0.25))


def test_species_table_matches_oracle_table(oracle):
    """Product species parameters == the oracle's independent restatement of Appendix A."""
    from animal_vision_amd import animals as A

    for name, ospec in oracle.DICHROMATS.items():
        cls = getattr(A, name.capitalize())
        s = cls.SPEC
        assert (s.alpha, s.s_scale, s.post) == (ospec.alpha, ospec.s_scale, ospec.post), name
        if s.post == "gauss":
            assert s.sigma == ospec.sigma
        if s.post == "streak":
            assert s.streak == ospec.streak
        used_chroma = None if ospec.chroma_discarded else ospec.chroma
        assert s.chroma == used_chroma, name


def test_reference_module_names_importable():
    import importlib

    for n in ("dog", "cat", "sheep", "pig", "cow", "goat", "rat", "horse", "rabbit", "panda", "squirrel",
              "elephant", "lion", "wolf", "fox", "bear", "raccoon", "deer", "kangaroo", "tiger"):
        m = importlib.import_module(f"animal_vision_amd.animals.{n}")
        cls = getattr(m, n.capitalize())
        inst = cls()  # registry instantiates with no arguments (utils.py:91-130)
        assert hasattr(inst, "visualize")


def test_uv_species_modules_and_constructors_match_reference_registry():
    """Module and class names of the 16 UV species (animals/__init__.py:22-32 plus the modules utils.py imports), all
    constructible without arguments and exposing the reference's constructor attributes."""
    import importlib

    from animal_vision_amd import animals

    names = {"honeybee": "HoneyBee", "mantis_shrimp": "MantisShrimp", "reindeer": "Reindeer", "rat_uv": "RatUV", "goldfish": "Goldfish",
             "damselfish": "Damselfish", "anableps": "Anableps", "anchovy": "Anchovy", "guppy": "Guppy", "morpho": "Morpho",
             "heliconius": "Heliconius", "pieris": "Pieris", "hummingbird": "Hummingbird", "kestrel": "Kestrel",
             "jumping_spider": "JumpingSpider", "dragonfly": "Dragonfly"}
    for mod, cls in names.items():
        m = importlib.import_module(f"animal_vision_amd.animals.{mod}")
        inst = getattr(m, cls)()
        assert getattr(animals, cls) is getattr(m, cls) and hasattr(inst, "visualize")
        if mod not in ("honeybee",):
            assert hasattr(inst, "hsi_scale") and hasattr(inst, "panorama_scale") and inst.lambdas.ndim == 1
    assert set(animals.UV_CLASS) == set(names) - {"honeybee", "mantis_shrimp"}


# ---- UV / spectral host tables (product code in animal_vision_amd/uv.py) vs reference goldens ----------
def test_uv_host_tables_match_reference():
    from animal_vision_amd import uv

    g = load_golden("uv_helpers")
    assert np.array_equal(uv.D65_like(g["lam31"]), g["d65_31"])
    assert np.array_equal(uv.D65_like(g["lam81"]), g["d65_81"])
    with np.errstate(invalid="ignore", divide="ignore"):
        for i, (lo, hi) in enumerate(g["bp_bands"].tolist()):
            assert np.array_equal(uv.bandpass_weights(g["lam31"], lo, hi), g["bp_31"][i], equal_nan=True)
            assert np.array_equal(uv.bandpass_weights(g["lam81"], lo, hi), g["bp_81"][i], equal_nan=True)
    hb = load_golden("honeybee")
    lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    assert np.array_equal(np.stack(uv.honeybee_cone_curves(lam)), hb["curves"])
    assert [uv.uv_blur_ksize(s) for s in (0.2, 0.7, 1.0, 1.2, 3.0)] == [3, 7, 7, 9, 19]


def test_folded_rgb_matrix_reproduces_reference_catches(oracle):
    """K x 3 folded matrix (lobes x illuminant x cone curves) vs the reference's band-by-band catches."""
    from animal_vision_amd import uv

    hb = load_golden("honeybee")
    lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    E = uv.D65_like(lam)
    Wk = np.stack([E * c for c in uv.honeybee_cone_curves(lam)])
    M = uv.fold_rgb_matrix(Wk, lam)
    lin = oracle.uv_srgb_to_linear(oracle.to_float01(hb["in_s40"])).astype(np.float64)
    got = np.tensordot(lin, M.astype(np.float64), axes=([2], [1])).transpose(2, 0, 1)
    np.testing.assert_allclose(got, hb["catches_s40"], rtol=2e-6, atol=1e-9)  # tolerance: 1e-4 allowed, 2e-6 achieved
    gains, denom = uv.lobe_tables(lam)
    g2, d2 = oracle.lobe_tables(lam)
    np.testing.assert_allclose(gains, g2, rtol=2e-7)
    assert abs(denom - d2) < 3e-7


def test_honeybee_constructor_surface():
    from animal_vision_amd.animals import HoneyBee

    bee = HoneyBee()
    assert bee.mapping_mode == "opponent" and bee.adaptation == "white_patch" and bee.blur_sigma_px == 0.2
    assert bee.lambdas.shape == (31,) and bee.lambdas.dtype == np.float32
    assert bee.onnx_path.endswith("mst_plus_plus.onnx")
    with pytest.raises(AssertionError):
        bee.visualize(np.zeros((4, 4), np.uint8))


def test_gelu_tables_of_the_matrix_pipe_kernels_hold_their_stated_error():
    """csrc/mst_common.h: the coefficient tables of the clamped odd polynomial (gelu_poly2: degree 7 / 6 / 5, x clamped to +-A) and of its
    prescaled form (gelu_pre_multi: x / 4 enters, t = clamp01(x'^2), Phi = clamp01(x' R^(t) + 1/2), x' Phi = gelu(x) / 4 leaves), evaluated here
    in float32 NumPy step for step, against the exact erf GELU (nn.GELU() of MST_Plus_Plus.py:48-50) over [-9, 9]: max(|error| for x < 0,
    |error| / gelu for x > 0.02) must stay within what the header states -- a transcription slip in a table shows here, not in a tolerance test."""
    import math
    import os
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "animal-vision_amd", "csrc", "mst_common.h")).read()

    def table(macro, deg_macro, deg):
        m = re.search(r"#(?:el)?if %s == %d\n(?:#define AVX_GELU_A ([0-9.]+)f\n)?#define %s \{([^}]*)\}" % (deg_macro, deg, macro), src)
        assert m, (macro, deg)
        return [np.float32(float.fromhex(v.strip().rstrip("f"))) for v in m.group(2).split(",")], (float(m.group(1)) if m.group(1) else None)

    xs = np.linspace(-9.0, 9.0, 360001)
    ref = xs * 0.5 * (1.0 + np.vectorize(math.erf)(xs / math.sqrt(2.0)))

    def check(got, bound, what):
        err = np.abs(got.astype(np.float64) - ref)
        pos = xs > 0.02
        worst = max(float(err[~pos].max()), float((err[pos] / ref[pos]).max()))
        assert worst <= bound, (what, worst, bound)

    # the clamped form: Phi(x) - 1/2 = xc R(xc^2)
    for deg, bound in ((7, 1.4e-4), (6, 2.5e-4), (5, 7.0e-4)):
        R, A = table("AVX_GELU_COEFFS", "AVX_GELU_DEG", deg)
        x32 = xs.astype(np.float32)
        xc = np.clip(x32, -np.float32(A), np.float32(A))
        t = (xc * xc).astype(np.float32)
        r = np.full_like(t, R[-1])
        for k in range(deg - 1, -1, -1):
            r = (r * t + R[k]).astype(np.float32)
        check((x32 * (xc * r + np.float32(0.5)).astype(np.float32)).astype(np.float32), bound, ("clamped", deg))
    # the prescaled form (A = 4)
    for deg, bound in ((5, 1.15e-3), (6, 2.5e-4)):
        R, _ = table("AVX_GELU_PRE_COEFFS", "AVX_GELU_PRE_DEG", deg)
        assert abs(float(sum(np.float64(v) for v in R)) - 0.5) < 1e-6  # R^(1) = 1/2: beyond |x| = A the clamp pins Phi to 0 / 1
        xp = (xs / 4.0).astype(np.float32)
        t = np.clip((xp * xp).astype(np.float32), 0, 1)
        r = np.full_like(t, R[-1])
        for k in range(deg - 1, -1, -1):
            r = (r * t + R[k]).astype(np.float32)
        phi = np.clip((xp * r + np.float32(0.5)).astype(np.float32), 0, 1)
        check((xp * phi).astype(np.float32).astype(np.float64) * 4.0, bound, ("prescaled", deg))
        big = np.abs(xs) > 4.0
        assert np.array_equal((xp * phi)[big] * 4, np.where(xs[big] > 0, xp[big] * 4, 0.0 * xp[big]))  # exactly x or 0 outside the clamp


def test_fold_layernorm_is_the_affine_layernorm_followed_by_the_conv():
    """ml/mst_plus_plus.py::fold_layernorm (the weight contract of avx_mst_ffn_fused_mx, include/avx.h): with gamma scaling W1's rows and W1^T beta in
    row 31 (the padding channel, whose operand the kernel sets to 1), (normalised row with a 1 in slot 31) @ W' * s == (gamma * normalised + beta) @ W1,
    for one and for several 32-wide channel groups."""
    import torch

    from animal_vision_amd.ml.mst_plus_plus import fold_layernorm, pad_channels

    g = torch.Generator().manual_seed(5)
    for c in (32, 64, 128):
        real = c // 32 * 31
        w1 = pad_channels(torch.randn(real, 4 * real, generator=g, dtype=torch.float64), (0, 1))     # (c, 4c): K = input channels
        gam = pad_channels(torch.randn(real, generator=g, dtype=torch.float64), (0,))
        bet = pad_channels(torch.randn(real, generator=g, dtype=torch.float64), (0,))
        yn = pad_channels(torch.randn(7, real, generator=g, dtype=torch.float64), (1,))               # normalised rows, padding channels 0
        want = (yn * gam + bet) @ w1
        for s in (1.0, 4.0):
            wf = fold_layernorm(w1, gam, bet, s).double()
            op = yn.clone()
            op[:, 31] = 1.0                                                                            # what the kernel writes into the bias slot
            torch.testing.assert_close(op @ wf * s, want, rtol=2e-5, atol=2e-5)  # the fold itself runs in float32
