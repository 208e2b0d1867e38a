"""GPU: the reference's STEP-WISE helper surface by name (animals/animal_utils.py, uv_mappers.py), NumPy in / NumPy out on
the device -- what lets a species module written against the reference run unchanged (north_star) -- against the oracle's
restatements of the same functions and against the reference-generated goldens.

Tolerances: blurs bit-exact (shared Gaussian contract); transfer functions / matrix products / mappers float32 within 2e-6
relative (device powf / atan2f vs NumPy's); the uint8 end of the chained dog pipeline within 1 code on < 0.5 % of samples."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_dog_pipeline_chained_from_helpers_matches_reference_golden(oracle):
    """animals/dog.py:37-59, step by step with the drop-in helpers, in the reference's order."""
    from animal_vision_amd.animals import animal_utils as au

    g = load_golden("dichromat")
    for k in ("n48", "s48", "n120"):
        image = g[f"in_{k}"]
        assert au.check_input_image(image)
        img = au.get_normalized_image(image)
        assert img.dtype == np.float32 and img.shape == image.shape
        lin = au.srgb_to_linear(img)
        T = au.collapse_LMS_matrix(0.58, 0.65)
        out_lin = (lin.reshape(-1, 3) @ T.T).reshape(lin.shape)       # dog.py:43-48 (host matmul, as the species file does it)
        out_lin = au.apply_acuity_blur(out_lin, sigma=3.5)            # dog.py:51
        srgb = np.clip(au.linear_to_srgb(np.clip(out_lin, 0.0, 1.0)), 0.0, 1.0)
        out = (srgb * 255.0 + 0.5).astype(np.uint8)                   # dog.py:56-59
        want = g[f"dog_oracle_{k}"]
        d = np.abs(out.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 5e-3, (k, int(d.max()), float((d > 0).mean()))


def test_transfer_functions_and_normalise(oracle):
    from animal_vision_amd.animals import animal_utils as au

    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(5000, dtype=np.float32), np.array([0.0, 0.04045, 0.0404501, 0.0031308, 0.0031309, 1.0], np.float32)])
    np.testing.assert_allclose(au.srgb_to_linear(x), oracle.srgb_to_linear(x), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(au.linear_to_srgb(x), oracle.linear_to_srgb(x), rtol=2e-6, atol=1e-8)
    x3 = rng.random((7, 9, 3)).astype(np.float64)
    y = au.srgb_to_linear(x3)
    assert y.dtype == np.float64 and y.shape == x3.shape
    np.testing.assert_allclose(y, oracle.srgb_to_linear(x3), rtol=1e-6, atol=1e-9)
    u8 = rng.integers(0, 256, (33, 21, 3), dtype=np.uint8)
    assert np.array_equal(au.get_normalized_image(u8), oracle.get_normalized_image(u8))
    dark = (u8 // 255).astype(np.uint8)  # all <= 1: NOT divided (a3 quirk)
    assert np.array_equal(au.get_normalized_image(dark), oracle.get_normalized_image(dark))
    f255 = u8.astype(np.float32)
    assert np.array_equal(au.get_normalized_image(f255), oracle.get_normalized_image(f255))


def test_blurs_bit_exact_and_in_place_quirks(oracle):
    from animal_vision_amd.animals import animal_utils as au

    rng = np.random.default_rng(1)
    img = rng.random((50, 70, 3), dtype=np.float32)
    for sigma in (0.7, 1.0, 1.4, 3.5):
        assert np.array_equal(au.apply_acuity_blur(img, sigma), oracle.apply_acuity_blur(img, sigma)), sigma
    u8 = rng.integers(0, 256, (40, 32, 3), dtype=np.uint8)
    got = au.apply_acuity_blur(u8, 1.2)
    assert got.dtype == np.uint8 and np.array_equal(got, oracle.apply_acuity_blur(u8, 1.2))
    with pytest.raises(ValueError):
        au.apply_acuity_blur(img[..., 0], 1.0)
    # streak blur as coded (Q3), float32 argument blurred IN PLACE and returned (Q4)
    a, b = img.copy(), img.copy()
    ra = au.apply_anisotropic_acuity_blur_with_streak(a, 0.48, 0.8, 2.2, 6.0)
    rb = oracle.apply_anisotropic_acuity_blur_with_streak(b, 0.48, 0.8, 2.2, 6.0)
    assert ra is a and np.array_equal(a, rb)
    f64 = img.astype(np.float64)
    keep = f64.copy()
    r64 = au.apply_anisotropic_acuity_blur_with_streak(f64, 0.5, 1.2, 2.5, 3.0)
    assert r64.dtype == np.float64 and np.array_equal(f64, keep)  # not mutated: astype made a copy in the reference too
    np.testing.assert_allclose(r64, oracle.apply_anisotropic_acuity_blur_with_streak(keep.copy(), 0.5, 1.2, 2.5, 3.0), rtol=0, atol=1e-6)
    # chroma compression, S-cone row gain (in place on float32)
    np.testing.assert_allclose(au.apply_chroma_compression(img, 0.06), oracle.apply_chroma_compression(img, 0.06), rtol=2e-7, atol=1e-8)
    c, d = img.copy(), img.copy()
    rc = au.apply_s_cone_vertical_gain(c, 1.3, 0.5, power=1.4, extra_boost=0.25)
    rd = oracle.apply_s_cone_vertical_gain(d, 1.3, 0.5, power=1.4, extra_boost=0.25)
    assert rc is c and np.array_equal(c[..., :2], img[..., :2])
    np.testing.assert_allclose(c, rd, rtol=2e-7, atol=1e-8)
    e = au.apply_s_cone_vertical_gain(img.copy(), 1.5, 0.9, band=(0.4, 0.2, 0.5), clamp=False)
    np.testing.assert_allclose(e, oracle.apply_s_cone_vertical_gain(img.copy(), 1.5, 0.9, band=(0.4, 0.2, 0.5), clamp=False), rtol=2e-7, atol=1e-8)


def test_lms_matrices(oracle):
    from animal_vision_amd.animals import animal_utils as au

    g = load_golden("collapse_matrices")
    E = np.eye(3, dtype=np.float32)
    np.testing.assert_allclose(au.sRGB_to_LMS(E), g["lms_of_eye"], rtol=1e-7, atol=1e-9)
    r = au.LMS_to_RGB(E)
    assert r.dtype == np.float64
    np.testing.assert_allclose(r, g["rgb_of_eye"], rtol=1e-6, atol=1e-7)
    v = np.random.default_rng(2).random((1000, 3), dtype=np.float32)
    M = np.array([[0.31399022, 0.63951294, 0.04649755], [0.15537241, 0.75789446, 0.08670142], [0.01775239, 0.10944209, 0.87256922]], np.float32)
    np.testing.assert_allclose(au.sRGB_to_LMS(v), v @ M.T, rtol=2e-7, atol=1e-8)
    m = au.merge_L_M(v, 0.58)
    assert m.shape == (1000, 3) and np.array_equal(m[:, 0], m[:, 1]) and np.array_equal(m[:, 2], v[:, 2])
    np.testing.assert_allclose(m[:, 0], 0.58 * v[:, 0] + (1.0 - 0.58) * v[:, 1], rtol=3e-7, atol=1e-8)
    for i, (a, s) in enumerate(g["pairs"].tolist()):
        assert np.array_equal(au.collapse_LMS_matrix(a, s), g["T"][i])


def _frac_close(got, want, rtol, atol, frac):
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert bad.mean() <= frac, (float(bad.mean()), float(np.abs(got - want).max()))


def test_uv_mappers_by_name_vs_oracle(oracle):
    from animal_vision_amd import uv_mappers as um

    g = load_golden("uv_helpers")
    U, B, G = g["U"], g["B"], g["G"]
    for name, args in (("map_falsecolor", (U, B, G)), ("map_uv_purple_yellow", (U,)), ("map_uv_purple_yellow_soft", (U,)),
                       ("map_falsecolor_uv_mixed", (U, B, G)), ("map_opponent", (U, B, G))):
        got, want = getattr(um, name)(*args), getattr(oracle, name)(*args)
        assert got.shape == want.shape == U.shape + (3,) and got.dtype == np.float32 == want.dtype, name
        # opponent: floor(6 hue) is categorical -- a sample whose hue sits within rounding of a sector boundary may flip
        _frac_close(got, want, 2e-5, 2e-6, 2e-3 if name == "map_opponent" else 0.0)
    # the reference's own outputs on the same planes (tools/make_goldens.py ran uv_mappers.py itself)
    for name, key, args in (("map_falsecolor", "map_falsecolor", (U, B, G)), ("map_opponent", "map_opponent", (U, B, G)), ("map_uv_purple_yellow", "map_upy", (U,)),
                            ("map_uv_purple_yellow_soft", "map_upy_soft", (U,)), ("map_falsecolor_uv_mixed", "map_mixed_035", (U, B, G))):
        _frac_close(getattr(um, name)(*args), g[key], 2e-5, 2e-6, 2e-3 if name == "map_opponent" else 0.0)
    _frac_close(um.map_falsecolor_uv_mixed(U, B, G, alpha=0.45), g["map_mixed_045"], 2e-5, 2e-6, 0.0)
    np.testing.assert_allclose(um.map_linear_matrix(U, B, G, g["M"]), g["map_matrix"], rtol=3e-7, atol=1e-8)
    _frac_close(um.hsv_to_rgb(g["hsv"]), g["hsv_rgb"].astype(np.float32), 2e-6, 1e-6, 1e-3)
    M = np.array([[0.9, 0.1, 0.0], [0.05, 0.8, 0.2], [0.4, 0.0, 0.7]], np.float32)
    np.testing.assert_allclose(um.map_linear_matrix(U, B, G, M), oracle.map_linear_matrix(U, B, G, M), rtol=3e-7, atol=1e-8)
    hsv = np.random.default_rng(3).random((20, 30, 3), dtype=np.float32)
    _frac_close(um.hsv_to_rgb(hsv), oracle.hsv_to_rgb(hsv).astype(np.float32), 2e-6, 1e-6, 1e-3)
    assert um.map_uv_purple_yellow(U[..., None]).shape == U.shape + (3,)
    with pytest.raises(ValueError):
        um.map_uv_purple_yellow(np.zeros((2, 3, 4), np.float32))
    with pytest.raises(ValueError):
        um.map_falsecolor(U, B[:-1], G)


def test_tapetum_bloom_and_rod_vision_vs_oracle(oracle):
    """animal_utils.py:183-204, :261-305 -- defined by the reference, used by none of its species (cat.py:50-59 names them in a commented
    block): no reference outputs exist, so the pin is the oracle's restatement (blur = the shared Gaussian contract, bit-exact; the
    elementwise parts float32, `power` within the device's powf)."""
    from animal_vision_amd.animals import animal_utils as au

    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:97, 0:131]
    img = (0.5 + 0.45 * np.sin(xx / 7.0)[..., None] * np.cos(yy[..., None] / 5.0 + np.arange(3)) + 0.05 * rng.standard_normal((97, 131, 3))).astype(np.float32)
    for frame in (img, (img * 1.3 - 0.1).astype(np.float32), img.astype(np.float64)):  # the second leaves [0, 1]: both helpers clip first
        for kw in ({}, dict(strength=0.3, sigma=1.5)):
            got, want = au.apply_tapetum_bloom(frame, **kw), oracle.apply_tapetum_bloom(frame, **kw)
            assert got.dtype == frame.dtype and got.shape == frame.shape
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-7)
        for kw in ({}, dict(chroma_scale=0.15, luminance_boost=1.1, gamma=0.6)):
            got, want = au.apply_rod_vision(frame, **kw), oracle.apply_rod_vision(frame, **kw)
            assert got.dtype == frame.dtype and got.shape == frame.shape
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7)


def test_tapetum_bloom_and_rod_vision_vs_reference_golden():
    """The same two helpers on the device against the REFERENCE's outputs (tests/golden/bloom_rod.npz, tools/make_goldens.py::g_bloom_rod:
    animal_utils.py:183-204, :261-305 run with the shared Gaussian injected): blur bit-exact by the shared contract, elementwise float32,
    `power` within the device's powf."""
    from animal_vision_amd.animals import animal_utils as au
    from conftest import load_golden
    from test_oracle_golden import _bloom_rod_cases

    for what, f, kw, want in _bloom_rod_cases(load_golden("bloom_rod")):
        got = (au.apply_tapetum_bloom if what == "bloom" else au.apply_rod_vision)(f.copy(), **kw)
        assert got.dtype == want.dtype and got.shape == want.shape
        if what == "bloom":
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-7)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7)


def test_uv_helpers_by_name_vs_oracle(oracle):
    """animal_vision_amd.uv_helpers: every function name of the reference's uv_helpers.py, NumPy in / NumPy out on the device, against
    the oracle's restatements (blurs and resizes bit-exact, elementwise float32 within the device's powf / cosf)."""
    from animal_vision_amd import uv_helpers as uh

    for name in ("to_float01", "from_float01", "srgb_to_linear", "linear_to_srgb", "safe_norm", "resize_preserve_range", "gaussian_blur", "panorama_warp",
                 "apply_scatter_and_blue_bias", "snow_glare_tone_compress", "bandpass_weights", "integrate_band", "integrate_uv", "classic_rgb_to_hsi_scaled",
                 "D65_like", "von_kries_white_patch", "von_kries_gray_world", "EPS_DEFAULT"):
        assert hasattr(uh, name), name
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    f01 = rng.random((37, 53, 3), dtype=np.float32)
    f255 = (f01 * 255).astype(np.float32)
    for x in (u8, f01, f255, f01.astype(np.float64)):
        got, want = uh.to_float01(x), oracle.to_float01(x)
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, want, rtol=0, atol=6e-8)
    assert np.array_equal(uh.from_float01(f01, np.uint8), oracle.from_float01(f01, np.uint8))
    assert uh.from_float01(f01, np.float64).dtype == np.float64
    wide = np.concatenate([f01.ravel(), np.array([0.0, 0.04045, 0.0404501, 0.0031308, 0.0031309, 1.0, -0.2, 1.7], np.float32)])
    with np.errstate(invalid="ignore"):  # NumPy evaluates both np.where branches: the power of a negative base is discarded, not used
        np.testing.assert_allclose(uh.srgb_to_linear(wide), oracle.uv_srgb_to_linear(wide), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(uh.linear_to_srgb(wide), oracle.uv_linear_to_srgb(wide), rtol=2e-6, atol=1e-7)
    for kw in (dict(sigma=0.1, blue_bias=0.05), dict(sigma=1.2, blue_bias=-0.02)):
        np.testing.assert_allclose(uh.apply_scatter_and_blue_bias(f01, **kw), oracle.apply_scatter_and_blue_bias(f01, **kw), rtol=0, atol=6e-8)
    hot = (f01 * 1.2).astype(np.float32)
    np.testing.assert_allclose(uh.snow_glare_tone_compress(hot, strength=0.6), oracle.snow_glare_tone_compress(hot, strength=0.6), rtol=2e-7, atol=6e-8)
    assert uh.snow_glare_tone_compress(hot, strength=0.0) is hot
    lam = np.linspace(300.0, 700.0, 81, dtype=np.float32)
    hsi = rng.random((20, 30, 81), dtype=np.float32)
    np.testing.assert_allclose(uh.integrate_uv(hsi, lam, 320.0, 400.0), oracle.integrate_uv(hsi, lam, 320.0, 400.0), rtol=0, atol=2e-6)
    U, B, G = (rng.random((25, 31), dtype=np.float32) * s for s in (0.3, 1.4, 1e-12))
    for mine, ref in ((uh.von_kries_white_patch, oracle.von_kries_white_patch), (uh.von_kries_gray_world, oracle.von_kries_gray_world)):
        for a, b in zip(mine(U, B, G), ref(U, B, G)):
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=0)
    lam31 = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    got = uh.classic_rgb_to_hsi_scaled(f01[:36, :52], wavelengths=lam31, scale=0.25)
    want = oracle.classic_rgb_to_hsi_scaled(f01[:36, :52], wavelengths=lam31, scale=0.25)
    assert got.shape == (36, 52, 31)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7)


def test_cat_widevision_utils_by_name_vs_oracle(oracle):
    """animals/cat_widevision_utils.py by name: center_zoom (uint8 and float32 INTER_LINEAR), zoom_scale_from_cat_ratio, and the
    binocular wide-FOV warp of a float [0, 1] frame (two remaps + cos^2 blend) against the oracle's restatement."""
    from animal_vision_amd.animals import cat_widevision_utils as cw

    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    f01 = rng.random((90, 130, 3), dtype=np.float32)
    kw = dict(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, cat_to_human_ratio=1.25)
    s = cw.zoom_scale_from_cat_ratio(**kw)
    assert s == oracle.zoom_scale_from_cat_ratio(**kw) and s > 1.0
    assert np.array_equal(cw.center_zoom(u8, s), oracle.center_zoom(u8, s))
    assert cw.center_zoom(u8, 1.0) is u8
    np.testing.assert_allclose(cw.center_zoom(f01, s), oracle.center_zoom(f01, s), rtol=0, atol=1e-7)
    wk = dict(fov_in_deg=100.0, per_eye_half_fov_deg=105.0, overlap_deg=40.0)
    got, want = cw.animal_fov_binocular_warp(f01, **wk), oracle.animal_fov_binocular_warp(f01, **wk)
    assert got.dtype == np.float32 and got.shape == f01.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    with pytest.raises(NotImplementedError):
        cw.animal_fov_binocular_warp(f01, out_size=(64, 48), **wk)


def test_animal_utils2_human_zoom_and_cat_view_vs_oracle(oracle):
    """animals/animal_utils2.py:123-165 by name, composed here from the oracle's restatements of its parts."""
    from animal_vision_amd.animals import animal_utils2 as au2

    rng = np.random.default_rng(9)
    kw = dict(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, binocular_overlap_deg=40.0)
    for image in (rng.integers(0, 256, (84, 126, 3), dtype=np.uint8), rng.random((84, 126, 3), dtype=np.float32)):
        zoomed, cat = au2.human_zoom_and_cat_view(image, **kw)
        assert zoomed.dtype == image.dtype and cat.dtype == image.dtype and zoomed.shape == cat.shape == image.shape
        scale = oracle.zoom_scale_from_cat_ratio(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, cat_to_human_ratio=1.30)
        want_zoom = oracle.center_zoom(image, scale)
        img01 = image.astype(np.float32) / 255.0 if image.dtype == np.uint8 else image
        want_cat01 = oracle.animal_fov_binocular_warp(np.clip(img01, 0, 1).astype(np.float32), fov_in_deg=100.0, per_eye_half_fov_deg=105.0, overlap_deg=40.0)
        if image.dtype == np.uint8:
            assert np.array_equal(zoomed, want_zoom)
            want_cat = (np.clip(want_cat01, 0, 1) * 255.0 + 0.5).astype(np.uint8)
            d = np.abs(cat.astype(np.int16) - want_cat.astype(np.int16))
            assert d.max() <= 1 and (d > 0).mean() < 1e-3
        else:
            np.testing.assert_allclose(zoomed, np.clip(want_zoom, 0, 1), rtol=0, atol=1e-7)
            np.testing.assert_allclose(cat, np.clip(want_cat01, 0, 1), rtol=0, atol=2e-6)
