"""GPU: MantisShrimp.visualize on the device (csrc/mantis.hip) vs the reference's re-enacted outputs
(tests/golden/mantis.npz, produced by the imported reference class) and vs the oracle on other shapes/parameters.

Contract: the baseline (decode -> cubic panorama warp -> encode) is bit-exact; the stylised frame is a
float pipeline held to 1e-4 relative before the uint8 encode, i.e. codes within +-1, except where the categorical
`argmax` band of mantis_shrimp.py:202 ties to within rounding (hard tint flips: a bounded handful of pixels)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _check(got, want, what, frame, kw):
    """+-1 code everywhere except where the oracle itself is unstable under float32-level jitter (argmax ties of
    mantis_shrimp.py:202, arctan2 of ~0 gradients at :226), with an absolute cap (tests/_sensitivity.py)."""
    from _sensitivity import check_codes
    from oracle import cpu_ref

    return check_codes(got, want, what, lambda seed: cpu_ref.mantis_visualize(frame, _jit=cpu_ref.relative_jitter(seed), **kw)[1])


@pytest.mark.parametrize("tag,kw", [("default", {}), ("noresample", dict(hsi_scale=1.0, panorama_scale=1.0))])
def test_mantis_vs_reference_golden(tag, kw):
    from animal_vision_amd.animals import MantisShrimp

    g = load_golden("mantis")
    m = MantisShrimp(**kw)
    for k in ("s64", "n50"):
        base, out = m.visualize(g[f"in_{k}"])
        assert base.dtype == np.uint8 and out.dtype == np.uint8 and base.shape == out.shape == g[f"in_{k}"].shape
        assert np.array_equal(base, g[f"{tag}_base_{k}"]), (tag, k, "baseline")
        _check(out, g[f"{tag}_out_{k}"], (tag, k), g[f"in_{k}"], kw)


@pytest.mark.parametrize("shape,kw", [
    ((96, 128, 3), {}),
    ((90, 121, 3), dict(hsi_scale=0.5, panorama_scale=1.3, winner_take_most=1.0, orientation_mix=1.0)),
    ((72, 100, 3), dict(pre_soft_sigma=0.0, unsharp_amount=0.0, scan_row_gain=0.0, periph_blur_sigma=0.0, haze_strength=0.0)),
    ((64, 64, 3), dict(bands=((400.0, 500.0), (500.0, 600.0), (600.0, 700.0)), lambdas=np.linspace(400.0, 700.0, 31), scan_soften=0.0)),
    ((270, 480, 3), {}),
])
def test_mantis_vs_oracle(oracle, shape, kw):
    from animal_vision_amd.animals import MantisShrimp

    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    yy, xx = np.mgrid[0 : shape[0], 0 : shape[1]]
    smooth = (127 + 100 * np.sin(xx / 9.0)[..., None] * np.cos(yy[..., None] / 7.0 + np.arange(3))).clip(0, 255)
    frame = (0.7 * smooth + 0.3 * rng.integers(0, 256, shape)).astype(np.uint8)
    base, out = MantisShrimp(**kw).visualize(frame)
    wbase, wout = oracle.mantis_visualize(frame, **kw)
    assert np.array_equal(base, wbase)
    _check(out, wout, (shape, kw), frame, kw)


def test_mantis_float_frames_vs_oracle(oracle):
    """Same-dtype contract: float32 in [0,1], float32 in [0,255], float64 frames."""
    from animal_vision_amd.animals import MantisShrimp

    u8 = load_golden("mantis")["in_s64"]
    sp = MantisShrimp()
    for frame in ((u8 / 255.0).astype(np.float32), u8.astype(np.float32), u8 / 255.0):
        base, out = sp.visualize(frame)
        wbase, wout = oracle.mantis_visualize(frame)
        assert base.dtype == out.dtype == frame.dtype == wout.dtype and out.shape == frame.shape
        np.testing.assert_allclose(base, wbase, rtol=0, atol=2e-5)
        d = np.abs(out.astype(np.float64) - wout.astype(np.float64))
        assert float((d > 4e-3).mean()) <= 2e-3 and float(np.median(d)) < 2e-5, (frame.dtype, float(d.max()))


def test_mantis_rejects_non_numeric_and_bad_panorama():
    from animal_vision_amd.animals import MantisShrimp

    with pytest.raises(NotImplementedError):
        MantisShrimp().visualize(np.zeros((8, 8, 3), np.int32))
    with pytest.raises(ValueError):
        MantisShrimp(panorama_scale=0.8).visualize(np.zeros((16, 16, 3), np.uint8))


def test_mantis_reduced_stack_read_through_its_resize_equals_the_materialised_stack(monkeypatch):
    """hsi_scale < 1: min/max, the 95th percentile and the barcode recompute each pixel's K band values from the small stack
    (csrc/stack_up.h) instead of reading a resized H x W x K copy -- same arithmetic, so the frames must be identical."""
    from animal_vision_amd.animals import MantisShrimp
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    for frame, kw in ((structured_frame(3, 270, 484), {}), (noise_frame(4, 133, 201), dict(hsi_scale=0.5, panorama_scale=1.2)), (structured_frame(5, 1080, 1920), {})):
        m = MantisShrimp(**kw)
        monkeypatch.setenv("AVX_MANTIS_UP", "0")
        base0, out0 = m.visualize(frame)
        monkeypatch.setenv("AVX_MANTIS_UP", "1")
        base1, out1 = m.visualize(frame)
        assert np.array_equal(base0, base1) and np.array_equal(out0, out1), (frame.shape, kw, int(np.abs(out0.astype(int) - out1.astype(int)).max()))


def test_mantis_fused_finishing_stages_equal_the_seven_launch_route(monkeypatch):
    """Round 3: prep_render | blur | polgain | blur | unsharp_blend | blur | finish run as three launches, each blur carrying its neighbours
    as prologue / epilogue (csrc/mantis.hip::k_mantis_fused_blur).  Same expressions in the same order, so uint8 and float frames must be
    identical to the unfused route -- border tiles (reflection), frames smaller than a tile, pre-soften off, other radii."""
    from animal_vision_amd.animals import MantisShrimp
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    cases = ((structured_frame(3, 270, 484), {}), (noise_frame(4, 133, 201), dict(hsi_scale=0.5, panorama_scale=1.2)), (structured_frame(5, 1080, 1920), {}),
             (noise_frame(6, 40, 50), {}), (structured_frame(7, 97, 130), dict(pre_soft_sigma=0.0)), (noise_frame(8, 200, 333), dict(unsharp_sigma=1.6, periph_blur_sigma=0.4, pre_soft_sigma=0.6)),
             (structured_frame(9, 64, 96).astype(np.float32) / 255.0, {}))
    for frame, kw in cases:
        m = MantisShrimp(**kw)
        monkeypatch.setenv("AVX_MANTIS_FUSE", "0")
        base0, out0 = m.visualize(frame)
        monkeypatch.delenv("AVX_MANTIS_FUSE")
        base1, out1 = m.visualize(frame)
        assert out0.dtype == out1.dtype == frame.dtype
        assert np.array_equal(base0, base1) and np.array_equal(out0, out1), (frame.shape, kw, float(np.abs(out0.astype(np.float64) - out1.astype(np.float64)).max()))


def test_mantis_percentile_candidates_equal_three_full_passes(monkeypatch):
    """Round 3: the second radix pass of the P95 over the (virtual) band stack copies the values of the first pass's bin out, the third pass
    scans those candidates only (csrc/uv.hip::k_sel_cand).  Exact by construction: the frames must equal the three-full-pass route
    (AVX_MANTIS_CAND=0), also where nearly every value shares a bin (flat frames: the candidate list is the whole stack)."""
    from animal_vision_amd.animals import MantisShrimp
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    flat = np.full((120, 200, 3), 97, np.uint8)
    flat[40:80, 50:150] = (180, 60, 20)
    for frame, kw in ((structured_frame(3, 270, 484), {}), (noise_frame(4, 133, 201), dict(hsi_scale=0.5, panorama_scale=1.2)), (structured_frame(5, 1080, 1920), {}), (flat, {}),
                      (noise_frame(6, 720, 1280), {})):
        m = MantisShrimp(**kw)
        monkeypatch.setenv("AVX_MANTIS_CAND", "0")
        base0, out0 = m.visualize(frame)
        monkeypatch.delenv("AVX_MANTIS_CAND")
        base1, out1 = m.visualize(frame)
        base2, out2 = m.visualize(frame)  # again: the candidate counter must have been left clean
        assert np.array_equal(base0, base1) and np.array_equal(out0, out1) and np.array_equal(out0, out2), (frame.shape, kw)


def test_mantis_batch_on_lanes_equals_frame_by_frame(monkeypatch):
    """avx_mantis_u8_batch: the frames of a batch run on up to four internal streams with their own workspaces; every frame must
    come out exactly as from the single-frame entry point, with one lane and with four."""
    from animal_vision_amd import get_context
    from animal_vision_amd.animals import MantisShrimp
    from animal_vision_amd.synthetic import noise_frame, structured_frame

    H, W, B = 136, 200, 5
    frames = np.stack([structured_frame(i, H, W) if i % 2 else noise_frame(i, H, W) for i in range(B)])
    m = MantisShrimp()
    want = [m.visualize(f) for f in frames]
    ctx = get_context()
    m.ctx = ctx
    d_in, d_base, d_out = ctx.malloc(frames.nbytes), ctx.malloc(frames.nbytes), ctx.malloc(frames.nbytes)
    try:
        ctx.upload(frames, d_in)
        for lanes in ("1", "4"):
            monkeypatch.setenv("AVX_MANTIS_LANES", lanes)
            ctx.memset(d_out, 0)
            ctx.memset(d_base, 0)
            m.run_device_batch(d_in, d_base, d_out, B, H, W)
            base = ctx.download(d_base, frames.shape, np.uint8)
            out = ctx.download(d_out, frames.shape, np.uint8)
            for i in range(B):
                assert np.array_equal(base[i], want[i][0]) and np.array_equal(out[i], want[i][1]), (lanes, i)
    finally:
        for b in (d_in, d_base, d_out):
            b.free()
