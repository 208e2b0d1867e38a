"""GPU: the UV species on the device (plane programs, csrc/ew.hip + the stage kernels) vs the reference's
re-enacted outputs (tests/golden/uv_species.npz) and vs the same species run with the oracle's NumPy backend
on other frame sizes.

Contract: baseline (decode -> cubic panorama warp -> encode) bit-exact; the stylised frame is a float32
pipeline held to 1e-4 relative before the uint8 encode: codes within +-1, allowing a small fraction of samples
beyond that ONLY where the oracle's own arithmetic is unstable under float32-level jitter (a categorical decision within
rounding of its boundary, a 0/0-like cancellation) -- established per frame by tests/_sensitivity.py, with an absolute cap."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


# kept in step with animal_vision_amd.animals.UV_CLASS (asserted below): collection must not need the library
SPECIES = ["anableps", "anchovy", "damselfish", "dragonfly", "goldfish", "guppy", "heliconius", "hummingbird", "jumping_spider", "kestrel", "morpho",
           "pieris", "rat_uv", "reindeer"]


def test_species_list_complete():
    from animal_vision_amd import animals

    assert sorted(animals.UV_CLASS) == SPECIES


def _check(got, want, what, sp=None, frame=None, **kw):
    """+-1 code everywhere except where the oracle itself is unstable under float32-level jitter, with an absolute cap on
    the number of such pixels (tests/_sensitivity.py)."""
    from _sensitivity import check_codes
    from oracle import np_backend

    rerun = None if sp is None else (lambda seed: np_backend.run_jittered(sp, frame, seed, **kw)[1])
    return check_codes(got, want, what, rerun)


@pytest.mark.parametrize("mod", SPECIES)
def test_species_vs_reference_golden(mod):
    from animal_vision_amd import animals

    g = load_golden("uv_species")
    sp = getattr(animals, animals.UV_CLASS[mod])()
    for k in ("s64", "n50"):
        base, out = sp.visualize(g[f"in_{k}"])
        assert base.dtype == np.uint8 and out.dtype == np.uint8
        assert np.array_equal(base, g[f"{mod}_base_{k}"]), (mod, k, "baseline")
        _check(out, g[f"{mod}_out_{k}"], (mod, k), sp, g[f"in_{k}"])


@pytest.mark.parametrize("mod", SPECIES)
def test_species_vs_oracle_other_sizes(mod):
    from animal_vision_amd import animals
    from oracle import np_backend

    sp = getattr(animals, animals.UV_CLASS[mod])()
    for shape in ((96, 128, 3), (135, 241, 3)):
        rng = np.random.default_rng(shape[0] + shape[1])
        yy, xx = np.mgrid[0 : shape[0], 0 : shape[1]]
        smooth = (127 + 100 * np.sin(xx / 9.0)[..., None] * np.cos(yy[..., None] / 7.0 + np.arange(3))).clip(0, 255)
        frame = (0.75 * smooth + 0.25 * rng.integers(0, 256, shape)).astype(np.uint8)
        base, out = sp.visualize(frame)
        wbase, wout = np_backend.run(sp, frame)
        assert np.array_equal(base, wbase), (mod, shape)
        _check(out, wout, (mod, shape), sp, frame)
    # second frame of an already-recorded size replays the plan
    frame2 = np.ascontiguousarray(frame[::-1])
    base, out = sp.visualize(frame2)
    wbase, wout = np_backend.run(sp, frame2)
    assert np.array_equal(base, wbase)
    _check(out, wout, (mod, "replay"), sp, frame2)


@pytest.mark.parametrize("mod", ["reindeer", "rat_uv", "kestrel", "anableps"])
def test_species_float_frames_vs_oracle(mod):
    """Animal.visualize keeps the input dtype: float32 in [0,1], float32 in [0,255], float64."""
    from animal_vision_amd import animals
    from oracle import np_backend

    g = load_golden("uv_species")
    sp = getattr(animals, animals.UV_CLASS[mod])()
    u8 = g["in_s64"]
    for name, frame in (("f32_01", (u8 / 255.0).astype(np.float32)), ("f32_255", u8.astype(np.float32)), ("f64_01", u8 / 255.0)):
        base, out = sp.visualize(frame)
        wbase, wout = np_backend.run(sp, frame)
        assert base.dtype == frame.dtype == wbase.dtype and out.dtype == frame.dtype and out.shape == frame.shape
        np.testing.assert_allclose(base, wbase, rtol=0, atol=2e-5, err_msg=f"{mod} {name} baseline")
        d = np.abs(out.astype(np.float64) - wout.astype(np.float64))
        assert float((d > 4e-3).mean()) <= 2e-3 and float(np.median(d)) < 1e-5, (mod, name, float(d.max()), float((d > 4e-3).mean()))


def test_rat_uv_night_branch_and_auto_mode():
    from animal_vision_amd.animals import RatUV
    from oracle import np_backend

    g = load_golden("uv_species")
    sp = RatUV()
    for k in ("s64", "n50"):
        _check(sp.visualize(g[f"in_{k}"], mode="night")[1], g[f"rat_uv_night_out_{k}"], ("night", k), sp, g[f"in_{k}"], mode="night")
    dark = (g["in_s64"] // 6).astype(np.uint8)  # median luminance < 0.12: `auto` takes the night branch
    assert sp.variant(dark, np_backend.NumpyProbes) == "night"
    from animal_vision_amd.planevm import DeviceProbes

    assert abs(DeviceProbes.median_luma(dark) - np_backend.NumpyProbes.median_luma(dark)) < 1e-6
    base, out = sp.visualize(dark)
    wbase, wout = np_backend.run(sp, dark)
    assert np.array_equal(base, wbase)
    _check(out, wout, "auto->night", sp, dark)


def test_ew_program_basic_ops_and_reductions():
    """The interpreter itself: arithmetic, masks, row/column broadcasts, reductions, vs NumPy float32."""
    from animal_vision_amd.planevm import DeviceBackend, PlaneRef

    H, W = 37, 53
    be = DeviceBackend(H, W)
    rng = np.random.default_rng(5)
    a, b = rng.random((H, W), dtype=np.float32), rng.random((H, W), dtype=np.float32) + 0.1
    pa, pb = be.new_planes(2)
    be.ctx.upload(a, pa.buf.view(pa.offset, a.nbytes))
    be.ctx.upload(b, pb.buf.view(pb.offset, b.nbytes))
    A, B = be.load(pa), be.load(pb)
    col, row = np.linspace(-1, 1, W, dtype=np.float32), np.linspace(0, 2, H, dtype=np.float32)
    expr = be.where(A > 0.5, be.sqrt(A * B) + be.col(col), be.exp(-A) / B - be.row(row)) ** 2
    mn, mx, mean = be.min(expr), be.max(expr), be.mean(expr)
    norm = be.mat((expr - mn) / (mx - mn) + 0.0 * mean)
    p50 = be.percentile(norm, 50.0)
    res = be.mat(be.clip(norm - p50, -0.25, 0.25) + be.arctan2(A, B) * be.cos(A) - be.abs(be.sin(B)) + be.power(B, 1.7))
    be.flush()
    for fn in be.plan:
        fn(be.ctx.stream)
    got = be.ctx.download(res.imm.buf.view(res.imm.offset, 4 * H * W), (H, W), np.float32)
    e = np.where(a > 0.5, np.sqrt(a * b) + col[None, :], np.exp(-a) / b - row[:, None]) ** 2
    n = (e - e.min()) / (e.max() - e.min()) + np.float32(0.0) * e.mean()
    want = np.clip(n - np.percentile(n, 50.0), -0.25, 0.25) + np.arctan2(a, b) * np.cos(a) - np.abs(np.sin(b)) + np.power(b, np.float32(1.7))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    be.close()


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 4), (9, 16), (16, 7)])
def test_tiny_frames_all_uv_paths(shape):
    """Degenerate sizes: every resampling / blur / Sobel / percentile stage must stay in bounds and agree with the oracle
    (reflect-101 over several periods, 1-pixel reduced HSI grids, panorama of a 1-pixel-wide frame)."""
    from animal_vision_amd import animals
    from oracle import cpu_ref, np_backend

    H, W = shape
    frame = np.random.default_rng(H * 100 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    for mod in ("reindeer", "anableps", "kestrel", "morpho", "jumping_spider"):
        sp = getattr(animals, animals.UV_CLASS[mod])()
        base, out = sp.visualize(frame)
        wbase, wout = np_backend.run(sp, frame)
        assert np.array_equal(base, wbase), (mod, shape)
        d = np.abs(out.astype(np.int16) - wout.astype(np.int16))
        assert d.max() <= 1 or (d > 1).mean() <= 0.1, (mod, shape, int(d.max()))  # a handful of samples: one flip is a large fraction
    base, out = animals.MantisShrimp().visualize(frame)
    wbase, wout = cpu_ref.mantis_visualize(frame)
    assert np.array_equal(base, wbase)
    d = np.abs(out.astype(np.int16) - wout.astype(np.int16))
    assert d.max() <= 1 or (d > 1).mean() <= 0.1, ("mantis", shape, int(d.max()))
    _, bee = animals.HoneyBee().visualize(frame)
    _, wbee = cpu_ref.honeybee_visualize(frame)
    d = np.abs(bee.astype(np.int16) - wbee.astype(np.int16))
    assert d.max() <= 1 or (d > 1).mean() <= 0.1, ("honeybee", shape, int(d.max()))
