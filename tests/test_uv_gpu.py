"""GPU parity of the UV / spectral path (csrc/uv.hip through the C ABI) vs the oracle and golden vectors.

Tolerances (north_star): float spectral math within 1e-4 relative; order statistics exact; the shared
Gaussian contract bit-exact; final uint8 within 1 code (device atan2f/powf differ from NumPy's by ulps,
which can move a value across a quantiser threshold) on a small fraction of bytes."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uv():
    import animal_vision_amd as av
    from animal_vision_amd import uv

    assert av.device_count() > 0
    return uv


def _u8_close(got, want, max_frac=2e-3):
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"max |diff| = {d.max()}"
    frac = float((d > 0).mean())
    assert frac <= max_frac, f"{frac:.2e} of bytes differ by 1"


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 65537, 1 << 20])
@pytest.mark.parametrize("q", [0.0, 50.0, 95.0, 98.0, 99.0, 100.0])
def test_percentile_matches_numpy(uv, n, q):
    rng = np.random.default_rng(n + int(q))
    x = (rng.random(n, dtype=np.float32) ** 3) * 4.0 - 0.5  # mixed signs, skewed
    got = uv.percentile(x, q)
    want = float(np.percentile(x, q))
    assert got == want, (n, q, got, want)


def test_percentile_one_launch_equals_numpy_and_the_three_launch_form(uv, monkeypatch):
    """Round 3: the three radix passes can run in one launch (csrc/uv.hip::k_sel_all: register-resident share of the plane, ticket + generation
    counter between the passes; opt-in, AVX_SEL_ONE=1: several such launches on different streams can starve each other).  Planes smaller than a grid, a 1080p plane (exactly the register share), a 4K plane (the re-read loop),
    unaligned starts; each against np.percentile and against the three-launch form (AVX_SEL_ONE=0), repeated (the histograms and the
    ticket must be left clean)."""
    rng = np.random.default_rng(77)
    for n in (5, 4099, 1920 * 1080, 3840 * 2160 + 3):
        x = (rng.random(n + 1, dtype=np.float32) ** 2) * 3.0 - 0.25
        for off in (0, 1):
            for q in (95.0, 50.0, 99.9):
                want = float(np.percentile(x[off:off + n], q))
                assert uv.percentile(x[off:off + n], q) == want, (n, off, q, "three launches (the default)")
                monkeypatch.setenv("AVX_SEL_ONE", "1")
                for rep in range(2):
                    assert uv.percentile(x[off:off + n], q) == want, (n, off, q, rep)
                monkeypatch.delenv("AVX_SEL_ONE")


def test_percentile_duplicates_and_constants(uv):
    x = np.zeros(10000, np.float32)
    x[9000:] = 1.0
    for q in (50.0, 89.99, 90.0, 95.0):
        assert uv.percentile(x, q) == float(np.percentile(x, q))
    x = np.full(333, 0.25, np.float32)
    assert uv.percentile(x, 95.0) == 0.25
    x = np.round(np.random.default_rng(1).random(50000, dtype=np.float32) * 16) / 16  # heavy ties
    for q in (5.0, 95.0, 99.0):
        assert uv.percentile(x, q) == float(np.percentile(x, q))


def test_spectral_integrate_vs_oracle(uv, oracle):
    g = load_golden("uv_helpers")
    cube, lam = g["cube31"], g["lam31"]
    for i, (lo, hi) in enumerate(g["bp_bands"][:5].tolist()):
        got = uv.integrate_band(cube, lam, lo, hi)
        np.testing.assert_allclose(got, g["ib_31"][i], rtol=1e-5, atol=1e-7)
    # K = 10 mantis bands on an 81-band cube, NHWC float32 and NCHW float16
    rng = np.random.default_rng(3)
    lam81 = np.linspace(300.0, 700.0, 81, dtype=np.float32)
    cube81 = rng.random((40, 52, 81), dtype=np.float32)
    Wk = np.stack([uv.bandpass_weights(lam81, lo, hi) for lo, hi in oracle.MANTIS_BANDS])
    want = np.stack([oracle.integrate_band(cube81, lam81, lo, hi) for lo, hi in oracle.MANTIS_BANDS])
    got, stats = uv.spectral_integrate(cube81, Wk, return_stats=True)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(stats[:, 0], want.reshape(10, -1).min(1), rtol=1e-5)
    np.testing.assert_allclose(stats[:, 1], want.reshape(10, -1).max(1), rtol=1e-5)
    np.testing.assert_allclose(stats[:, 2], want.reshape(10, -1).mean(1), rtol=1e-5)
    chw16 = np.ascontiguousarray(cube81.transpose(2, 0, 1)).astype(np.float16)
    got16 = uv.spectral_integrate(chw16, Wk, layout="nchw")
    want16 = np.tensordot(chw16.astype(np.float32), Wk, axes=([0], [1])).transpose(2, 0, 1)
    np.testing.assert_allclose(got16, want16, rtol=1e-5, atol=1e-7)


def _band_matrix(uv, K, B):
    lam = np.linspace(300.0 if B == 81 else 400.0, 700.0, B, dtype=np.float32)
    edges = np.linspace(float(lam[0]), float(lam[-1]), K + 1)
    return np.ascontiguousarray(np.stack([uv.bandpass_weights(lam, float(lo), float(hi)) for lo, hi in zip(edges[:-1], edges[1:])]), dtype=np.float32)


@pytest.mark.parametrize("K,B", [(12, 31), (10, 81), (3, 32)])
@pytest.mark.parametrize("shape", [(1, 7), (9, 23), (16, 16), (37, 53), (64, 260)])
def test_spectral_integrate_fp16_nhwc_vs_tensordot(uv, K, B, shape):
    """BASELINE config 5 and the MST++ hand-off: float16 NHWC cube -> K float32 planes (csrc/uv.hip::k_spectral_nhwc_h, the
    LDS-staged kernel).  (12,31) = the config's mantis stack, (10,81) = the reference MantisShrimp default, (3,32) = the
    32-wide padded MST++ cube.  Pixel counts below, at and across the 256-pixel tile, not multiples of it; float32
    np.tensordot of the same float16 values is the reference arithmetic (north_star: 1e-4 relative; held to 1e-5)."""
    rng = np.random.default_rng(K * 100 + B + shape[1])
    H, W = shape
    cube = rng.random((H, W, B), dtype=np.float32).astype(np.float16)
    Wk = _band_matrix(uv, K, B) if B != 32 else rng.random((K, B), dtype=np.float32)
    got, stats = uv.spectral_integrate(cube, Wk, return_stats=True)
    want = np.tensordot(cube.astype(np.float32), Wk, axes=([2], [1])).transpose(2, 0, 1)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(stats[:, 0], want.reshape(K, -1).min(1), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(stats[:, 1], want.reshape(K, -1).max(1), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(stats[:, 2], want.reshape(K, -1).mean(1), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("K,B", [(12, 31), (10, 81)])
def test_spectral_integrate_fp16_nhwc_4k_properties(uv, K, B):
    """The same kernel at BASELINE's full size (3840x2160) through size-independent properties: (1) linearity -- planes of
    a cube built from two pixel values tiled over the frame equal the K x 2 table computed on the CPU, at every pixel;
    (2) the device statistics agree with the planes that came back; (3) changing the weights between calls is honoured
    (the device-side weight table is cached against a host mirror)."""
    H, W = 2160, 3840
    rng = np.random.default_rng(B)
    px = rng.random((2, B), dtype=np.float32).astype(np.float16)
    which = (np.add.outer(np.arange(H), np.arange(W)) % 3 == 0).astype(np.int8)  # pattern not aligned to the 256-pixel tiles
    cube = px[which]
    for scale in (1.0, 0.5):
        Wk = _band_matrix(uv, K, B) * np.float32(scale)
        table = (px.astype(np.float32) @ Wk.T.astype(np.float32))  # (2, K) float32, same sum order is not needed at 1e-5
        got, stats = uv.spectral_integrate(cube, Wk, return_stats=True)
        want = table[which].transpose(2, 0, 1)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(stats[:, 0], got.reshape(K, -1).min(1), rtol=1e-6)
        np.testing.assert_allclose(stats[:, 1], got.reshape(K, -1).max(1), rtol=1e-6)
        np.testing.assert_allclose(stats[:, 2], got.reshape(K, -1).mean(1, dtype=np.float64), rtol=1e-5)


@pytest.mark.parametrize("sigma", [0.2, 0.7, 1.0, 1.2, 3.0])
@pytest.mark.parametrize("shape", [(36, 44), (5, 3), (70, 130, 3)])
def test_gaussian_blur_bit_exact_with_shared_contract(uv, oracle, sigma, shape):
    img = np.random.default_rng(int(sigma * 10) + len(shape)).random(shape, dtype=np.float32)
    assert np.array_equal(uv.gaussian_blur(img, sigma), oracle.gaussian_blur(img, sigma))


def test_safe_norm_and_von_kries_scaling(uv, oracle):
    g = load_golden("uv_helpers")
    U, B, G = g["U"], g["B"], g["G"]
    assert np.array_equal(uv.safe_norm(U), g["safe_norm_U"])
    assert np.array_equal(uv.safe_norm(np.full((4, 5), 0.25, np.float32)), g["safe_norm_const"])
    planes = np.stack([U, B, G])
    assert np.array_equal(uv.planes_blur(planes, 1, 0.0, scale="white_patch"), g["vk_wp"])
    np.testing.assert_allclose(uv.planes_blur(planes, 1, 0.0, scale="gray_world"), g["vk_gw"], rtol=2e-6)


def test_classic_rgb_to_hsi_vs_reference_golden(uv):
    g = load_golden("lobes")
    for nm in ("31", "81", "129"):
        got = uv.classic_rgb_to_hsi(g["img"], wavelengths=g[f"lam{nm}"])
        assert got.shape == g[f"hsi{nm}"].shape and got.dtype == np.float32
        np.testing.assert_allclose(got, g[f"hsi{nm}"], rtol=1e-5, atol=1e-7)


def test_honeybee_vs_reference_golden(uv):
    """Outputs of the reference HoneyBee class (every mapping mode x adaptation)."""
    from animal_vision_amd.animals import HoneyBee

    g = load_golden("honeybee")
    checked = 0
    for key in g.files:
        tail = key.rsplit("_", 1)[-1]
        if tail not in ("s40", "n40") or key.startswith(("in_", "catches")):
            continue
        head = key[: -len(tail) - 1]
        if head == "opponent_noblur":
            bee = HoneyBee(blur_sigma_px=0.0)
        elif head.startswith("downsample"):  # a14: hsi_downsample route (area down, lobes, linear up)
            bee = HoneyBee(hsi_downsample=True, hsi_scale=int(head[len("downsample"):]) / 100.0)
        else:
            adapt = "white_patch" if head.endswith("white_patch") else "gray_world"
            mode = head[: -len(adapt) - 1]
            bee = HoneyBee(mapping_mode=mode, adaptation=adapt, custom_matrix=g["custom_matrix"] if mode == "custom_matrix" else None)
        frame = g[f"in_{tail}"]
        base, out = bee.visualize(frame)
        assert base is frame and out.dtype == np.uint8 and out.shape == frame.shape
        # 40x56 frames: allow a handful of +-1 codes
        _u8_close(out, g[key], max_frac=5e-3)
        checked += 1
    assert checked >= 20


def test_honeybee_downsample_vs_oracle_other_size(uv, oracle):
    from animal_vision_amd.animals import HoneyBee
    from animal_vision_amd.synthetic import structured_frame

    frame = structured_frame(2, 270, 481)
    bee = HoneyBee(hsi_downsample=True, hsi_scale=0.1)
    for f in (frame, np.ascontiguousarray(frame[::-1])):  # second frame replays the recorded plan
        base, out = bee.visualize(f)
        assert base is f
        _u8_close(out, oracle.honeybee_visualize(f, hsi_downsample=True, hsi_scale=0.1)[1], max_frac=5e-3)


def test_honeybee_float_frames_vs_oracle(uv, oracle):
    """Same-dtype contract: float32 in [0,1], float32 in [0,255], float64 frames."""
    from animal_vision_amd.animals import HoneyBee
    from animal_vision_amd.synthetic import structured_frame

    u8 = structured_frame(1, 90, 121)
    for mode in ("opponent", "falsecolor"):
        bee = HoneyBee(mapping_mode=mode)
        for frame in ((u8 / 255.0).astype(np.float32), u8.astype(np.float32), u8 / 255.0):
            base, out = bee.visualize(frame)
            want = oracle.honeybee_visualize(frame, mapping_mode=mode)[1]
            assert base is frame and out.dtype == frame.dtype == want.dtype and out.shape == frame.shape
            d = np.abs(out.astype(np.float64) - want.astype(np.float64))
            assert float((d > 4e-3).mean()) <= 2e-3 and float(np.median(d)) < 2e-5, (mode, frame.dtype, float(d.max()))


def test_honeybee_planes_and_1080p_vs_oracle(uv, oracle):
    from animal_vision_amd.animals import HoneyBee
    from animal_vision_amd.synthetic import structured_frame

    frame = structured_frame(0, 1080, 1920)
    bee = HoneyBee()
    out, planes = bee._operator()(frame, return_planes=True)
    lam = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    hsi = oracle.classic_rgb_to_hsi_lobes(oracle.to_float01(frame), lam)
    U, B, G = oracle.honeybee_catches(hsi, lam)
    U, B, G = oracle.von_kries_white_patch(U, B, G)
    want_planes = np.stack([oracle.gaussian_blur(p, 0.2) for p in (U, B, G)])
    np.testing.assert_allclose(planes, want_planes, rtol=1e-4, atol=1e-6)
    want, _ = oracle.honeybee_tail(*oracle.honeybee_catches(hsi, lam), np.uint8)
    _u8_close(out, want)
    # the HSI-cube source (what the MST++ route feeds) agrees with the folded-RGB source
    out2 = bee._operator()(frame, hsi=hsi, hsi_layout="nhwc")
    _u8_close(out2, want)


@pytest.mark.parametrize("mapping", ["opponent", "falsecolor", "custom_matrix", "uv_purple_yellow"])
@pytest.mark.parametrize("adaptation", ["white_patch", "gray_world", None])
def test_honeybee_recompute_form_equals_plane_form(uv, oracle, mapping, adaptation, monkeypatch):
    """csrc/uv.hip has two schedules of the honeybee tail: planes streamed between stage kernels, and the recompute form
    (k_bee_tile: every pass re-derives the blurred catches from the uint8 frame, the frames of a batch share each launch).
    They run the same arithmetic, so a batch through both must agree byte for byte -- frames whose sides are not multiples
    of the 32-pixel tile, a batch of three, and a no-blur variant -- and both agree with the oracle to +-1 code."""
    from animal_vision_amd.animals import HoneyBee
    from animal_vision_amd.synthetic import structured_frame

    M = np.array([[0.9, 0.1, 0.0], [0.05, 0.8, 0.2], [0.4, 0.0, 0.7]], np.float32)
    for (H, W), sigma in (((70, 101), 0.2), ((33, 31), 0.2), ((64, 64), 0.0)):
        frames = np.stack([structured_frame(k, H, W) for k in range(3)])
        outs = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("AVX_BEE_FUSED", fused)
            op = HoneyBee(adaptation=adaptation, mapping_mode=mapping, custom_matrix=M, blur_sigma_px=sigma)._operator()
            ctx = op._ctx()
            d_in, d_out = ctx.upload(frames), ctx.malloc(frames.nbytes)
            op.run_device(d_in, d_out, 3, H, W)
            outs[fused] = ctx.download(d_out, frames.shape, np.uint8)
            d_in.free(); d_out.free()
        assert np.array_equal(outs["1"], outs["0"]), (mapping, adaptation, (H, W), int(np.abs(outs["1"].astype(int) - outs["0"].astype(int)).max()))
        for k in range(3):
            _, want = oracle.honeybee_visualize(frames[k], adaptation=adaptation, mapping_mode=mapping, custom_matrix=M, blur_sigma_px=sigma)
            _u8_close(outs["1"][k], want, max_frac=5e-3)
