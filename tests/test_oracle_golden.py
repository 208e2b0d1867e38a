"""Pins the oracle (oracle/cpu_ref.py) against golden vectors produced by the
imported reference (tools/make_goldens.py).  CPU only.

Bit-exact unless a tolerance is written next to the assert."""
import numpy as np
import pytest

from conftest import load_golden


def test_decode_lut_bit_exact(oracle):
    g = load_golden("srgb_tables")
    assert np.array_equal(oracle.decode_lut_u8(), g["decode_lut"])
    # UV route (to_float01 + uv_helpers.srgb_to_linear) builds the same table
    codes = np.arange(256, dtype=np.uint8).reshape(16, 16, 1)
    assert np.array_equal(oracle.uv_srgb_to_linear(oracle.to_float01(codes)).reshape(256), g["decode_lut_uv"])
    assert np.array_equal(g["decode_lut"], g["decode_lut_uv"])


def test_encode_formula_and_thresholds_bit_exact(oracle):
    g = load_golden("srgb_tables")
    assert np.array_equal(oracle.encode_u8(g["ramp_f32"]), g["ramp_f32_u8"])
    assert np.array_equal(oracle.encode_u8(g["ramp_f64"]), g["ramp_f64_u8"])
    assert np.array_equal(oracle.encode_thresholds(np.float32), g["enc_thr_f32"])
    assert np.array_equal(oracle.encode_thresholds(np.float64), g["enc_thr_f64"])
    # the three f32 quantisers of the reference (dog.py:54-57, honeybee.py:166-171,
    # from_float01) share one threshold table
    assert np.array_equal(g["enc_thr_f32"], g["enc_thr_bee_f32"])
    assert np.array_equal(g["enc_thr_f32"], g["enc_thr_ff01_f32"])
    # threshold quantiser == formula on the ramps (which include thr-1ulp, thr, thr+1ulp)
    assert np.array_equal(oracle.quantize_with_thresholds(g["ramp_f32"], g["enc_thr_f32"]), g["ramp_f32_u8"])
    assert np.array_equal(oracle.quantize_with_thresholds(g["ramp_f32"], g["enc_thr_f32"]), g["ramp_f32_bee_u8"])
    assert np.array_equal(oracle.quantize_with_thresholds(g["ramp_f32"], g["enc_thr_f32"]), g["ramp_f32_ff01_u8"])
    assert np.array_equal(oracle.quantize_with_thresholds(g["ramp_f64"], g["enc_thr_f64"]), g["ramp_f64_u8"])


def test_collapse_matrices_bit_exact(oracle):
    g = load_golden("collapse_matrices")
    for (a, s), T in zip(g["pairs"], g["T"]):
        assert np.array_equal(oracle.collapse_LMS_matrix(float(a), float(s)), T), (a, s)
    # SURVEY 8a row a6 quotes dog's matrix
    dogT = oracle.collapse_LMS_matrix(0.58, 0.65)
    np.testing.assert_allclose(dogT[0], [0.2073376, 0.28697416, -0.026987808], rtol=1e-6)
    E = np.eye(3, dtype=np.float32)
    assert np.array_equal(oracle.matmul3(E, oracle.M_RGB_TO_LMS), g["lms_of_eye"])
    assert np.array_equal(oracle.matmul3(E, oracle.M_LMS_TO_RGB), g["rgb_of_eye"])


def test_color_stage_fma_chain_bit_exact(oracle):
    g = load_golden("dichromat")
    lin = oracle.srgb_to_linear(oracle.get_normalized_image(g["in_n48"]))
    got = oracle.dichromat_color_stage(oracle.DICHROMATS["dog"], lin)
    assert got.dtype == np.float32
    assert np.array_equal(got, g["dog_colorstage_n48"])


@pytest.mark.parametrize("name", sorted(__import__("oracle.cpu_ref", fromlist=["x"]).DICHROMATS))
def test_dichromat_species_bit_exact(oracle, name):
    """<Species>.visualize of the reference vs the oracle template, blur = identity
    (pins every non-cv2 stage) and blur = the shared OpenCV restatement."""
    g = load_golden("dichromat")
    spec = oracle.DICHROMATS[name]
    for mode in ("identity", "oracle"):
        for k in ("n48", "s48", "n120", "dark"):
            key = f"{name}_{mode}_{k}"
            if key not in g.files:
                continue
            if mode == "identity" and spec.post == "streak":
                # identity must replace cv2.GaussianBlur inside the streak loop too
                import oracle.cpu_ref as O

                saved = O.cv_gaussian_blur
                O.cv_gaussian_blur = lambda img, *a, **kw: img.copy()
                try:
                    base, out = oracle.dichromat_visualize(spec, g[f"in_{k}"].copy())
                finally:
                    O.cv_gaussian_blur = saved
            else:
                blur = (lambda img, s: img.copy()) if mode == "identity" else None
                base, out = oracle.dichromat_visualize(spec, g[f"in_{k}"].copy(), blur=blur)
            assert out.dtype == np.uint8
            assert np.array_equal(out, g[key]), key


def test_dichromat_float_input_contract(oracle):
    g = load_golden("dichromat")
    base, out = oracle.dichromat_visualize(oracle.DICHROMATS["dog"], g["in_f32"].copy())
    assert out.dtype == np.float32
    assert np.array_equal(out, g["dog_oracle_f32"])


def test_a10_helpers_bit_exact(oracle):
    g = load_golden("dichromat")
    x = g["helper_in"]
    assert np.array_equal(oracle.apply_chroma_compression(x.copy(), 0.4), g["chroma_0p4"])
    assert np.array_equal(oracle.apply_s_cone_vertical_gain(x.copy(), s_top=1.3, s_bottom=0.5, power=1.4, extra_boost=0.25), g["scone_rat"])
    assert np.array_equal(oracle.apply_s_cone_vertical_gain(x.copy(), 1.0, 0.6, band=(0.4, 0.2, 0.5), clamp=False), g["scone_band"])


def _bloom_rod_cases(g):
    a = g["in_a"]
    frames = {"a": a, "b": (a * 1.3 - 0.1).astype(np.float32), "c": a.astype(np.float64)}
    for k, f in frames.items():
        for j, kw in enumerate(({}, dict(strength=0.3, sigma=1.5))):
            yield "bloom", f, kw, g[f"bloom_{k}_{j}"]
        for j, kw in enumerate(({}, dict(chroma_scale=0.15, luminance_boost=1.1, gamma=0.6))):
            yield "rod", f, kw, g[f"rod_{k}_{j}"]


def test_a10_bloom_and_rod_vision_vs_reference(oracle):
    """apply_tapetum_bloom / apply_rod_vision (animal_utils.py:183-204, :261-305): the oracle's restatement against the reference's own
    lines run by tools/make_goldens.py::g_bloom_rod with the shared Gaussian injected (cv2 itself: parity unpinned).  Bit-exact except
    rod vision's np.power, whose float32 SIMD implementation is host dependent: 1 ulp."""
    g = load_golden("bloom_rod")
    for what, f, kw, want in _bloom_rod_cases(g):
        got = (oracle.apply_tapetum_bloom if what == "bloom" else oracle.apply_rod_vision)(f.copy(), **kw)
        assert got.dtype == want.dtype and got.shape == want.shape
        if what == "bloom":
            assert np.array_equal(got, want), (what, kw, f.dtype)
        else:
            np.testing.assert_allclose(got, want, rtol=2.5e-7, atol=0, err_msg=str((what, kw, f.dtype)))


def test_uv_helpers_bit_exact(oracle):
    g = load_golden("uv_helpers")
    lam31, lam81 = g["lam31"], g["lam81"]
    assert np.array_equal(oracle.D65_like(lam31), g["d65_31"])
    assert np.array_equal(oracle.D65_like(lam81), g["d65_81"])
    with np.errstate(invalid="ignore", divide="ignore"):
        for i, (lo, hi) in enumerate(g["bp_bands"].tolist()):  # python floats, as species pass them
            assert np.array_equal(oracle.bandpass_weights(lam31, lo, hi), g["bp_31"][i], equal_nan=True), (lo, hi)
            assert np.array_equal(oracle.bandpass_weights(lam81, lo, hi), g["bp_81"][i], equal_nan=True), (lo, hi)
    # quirk Q7: a UV band on the 400-700 grid degenerates to uniform 1/B weights
    assert np.array_equal(g["bp_31"][0], np.full(31, 1 / 31.0, np.float32))
    cube = g["cube31"]
    for i, (lo, hi) in enumerate(g["bp_bands"][:5].tolist()):
        assert np.array_equal(oracle.integrate_band(cube, lam31, lo, hi), g["ib_31"][i])
    assert np.array_equal(oracle.integrate_uv(cube, lam31, 400.0, 460.0), g["iuv_31"])
    U, B, G = g["U"], g["B"], g["G"]
    assert np.array_equal(oracle.safe_norm(U), g["safe_norm_U"])
    assert np.array_equal(oracle.safe_norm(np.full((4, 5), 0.25, np.float32)), g["safe_norm_const"])
    assert np.array_equal(np.stack(oracle.von_kries_white_patch(U, B, G)), g["vk_wp"])
    assert np.array_equal(np.stack(oracle.von_kries_gray_world(U, B, G)), g["vk_gw"])
    assert np.array_equal(oracle.to_float01(g["tf01_u8_in"]), g["tf01_u8"])
    assert np.array_equal(oracle.to_float01(g["tf01_u8_in"].astype(np.float32)), g["tf01_f255"])
    assert np.array_equal(oracle.from_float01(oracle.to_float01(g["tf01_u8_in"]) * 0.9, np.uint8), g["ff01_u8"])
    assert np.array_equal(oracle.snow_glare_tone_compress(g["glare_in"], strength=0.7), g["glare_out"])


def test_uv_mappers_bit_exact(oracle):
    g = load_golden("uv_helpers")
    U, B, G = g["U"], g["B"], g["G"]
    assert np.array_equal(oracle.hsv_to_rgb(g["hsv"]), g["hsv_rgb"])
    assert np.array_equal(oracle.map_falsecolor(U, B, G), g["map_falsecolor"])
    assert np.array_equal(oracle.map_opponent(U, B, G), g["map_opponent"])
    assert np.array_equal(oracle.map_uv_purple_yellow(U), g["map_upy"])
    assert np.array_equal(oracle.map_uv_purple_yellow_soft(U), g["map_upy_soft"])
    assert np.array_equal(oracle.map_falsecolor_uv_mixed(U, B, G), g["map_mixed_035"])
    assert np.array_equal(oracle.map_falsecolor_uv_mixed(U, B, G, alpha=0.45), g["map_mixed_045"])
    assert np.array_equal(oracle.map_linear_matrix(U, B, G, g["M"]), g["map_matrix"])


def test_lobes_bit_exact(oracle):
    """a13 against the reference's own lines 47-82 executed on torch-CPU."""
    g = load_golden("lobes")
    for nm in ("31", "81", "129"):
        got = oracle.classic_rgb_to_hsi_lobes(g["img"], g[f"lam{nm}"])
        assert got.shape == g[f"hsi{nm}"].shape and got.dtype == np.float32
        assert np.array_equal(got, g[f"hsi{nm}"]), nm
        # factored tables reproduce the cube to float rounding (tolerance: 2e-6 relative)
        gains, denom = oracle.lobe_tables(g[f"lam{nm}"])
        lin = oracle.uv_srgb_to_linear(g["img"])
        cube = (lin @ gains.T.astype(np.float32)) / np.float32(denom)
        np.testing.assert_allclose(cube, g[f"hsi{nm}"], rtol=2e-6, atol=1e-7)


def test_honeybee_end_to_end_bit_exact(oracle):
    g = load_golden("honeybee")
    lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    assert np.array_equal(np.stack(oracle.honeybee_cone_curves(lam)), g["curves"])
    img01 = oracle.to_float01(g["in_s40"])
    hsi = oracle.classic_rgb_to_hsi_lobes(img01, lam)
    assert np.array_equal(np.stack(oracle.honeybee_catches(hsi, lam)), g["catches_s40"])
    for key in g.files:
        parts = key.rsplit("_", 1)
        if parts[-1] not in ("s40", "n40") or key.startswith(("in_", "catches")):
            continue
        frame = g[f"in_{parts[-1]}"]
        head = parts[0]
        if head == "opponent_noblur":
            kw = dict(blur_sigma_px=0.0)
        elif head.startswith("downsample"):
            kw = dict(hsi_downsample=True, hsi_scale=int(head[len("downsample"):]) / 100.0)
        else:
            for adapt in ("white_patch", "gray_world"):
                if head.endswith("_" + adapt):
                    mode = head[: -len(adapt) - 1]
                    kw = dict(mapping_mode=mode, adaptation=adapt)
            if kw["mapping_mode"] == "custom_matrix":
                kw["custom_matrix"] = g["custom_matrix"]
        base, out = oracle.honeybee_visualize(frame, **kw)
        assert base is frame
        assert np.array_equal(out, g[key]), key


def test_geometry_helpers_bit_exact(oracle):
    """panorama_warp / classic_rgb_to_hsi_scaled / cat FOV helpers of the reference, driven with the shared
    resize/remap restatements (cv2 itself: parity unpinned), vs the oracle's restatement of the glue."""
    g = load_golden("geometry")
    lin = g["lin"]
    for sname, sc in (("112", 1.12), ("145", 1.45), ("105", 1.05)):
        assert np.array_equal(oracle.panorama_warp(lin, scale_x=sc), g[f"pano_{sname}"])
    lam81 = np.linspace(300.0, 700.0, 81, dtype=np.float32)
    assert np.array_equal(oracle.classic_rgb_to_hsi_scaled(lin, wavelengths=lam81, scale=0.25), g["hsi_scaled_025"])
    assert np.array_equal(oracle.classic_rgb_to_hsi_scaled(g["lin2"], wavelengths=lam81, scale=0.25), g["hsi_scaled_025_ragged"])
    assert oracle.zoom_scale_from_cat_ratio(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, cat_to_human_ratio=1.30) == float(g["zoom_scale"])
    for k in ("n48", "s60"):
        f = g[f"in_{k}"]
        H, W = f.shape[:2]
        warp = oracle.animal_fov_binocular_warp(oracle.get_normalized_image(f).astype(np.float32), fov_in_deg=100.0,
                                                per_eye_half_fov_deg=105.0, overlap_deg=40.0, out_size=(W, H))
        assert np.array_equal(warp, g[f"cat_warp01_{k}"])
        human, cat = oracle.cat_visualize(f)
        assert np.array_equal(human, g[f"cat_human_{k}"])
        assert np.array_equal(cat, g[f"cat_out_{k}"])


def test_mantis_end_to_end_bit_exact(oracle):
    g = load_golden("mantis")
    for tag, kw in (("default", {}), ("noresample", dict(hsi_scale=1.0, panorama_scale=1.0))):
        for k in ("s64", "n50"):
            base, out = oracle.mantis_visualize(g[f"in_{k}"], **kw)
            assert np.array_equal(base, g[f"{tag}_base_{k}"]), (tag, k)
            assert np.array_equal(out, g[f"{tag}_out_{k}"]), (tag, k)
