"""GPU parity: fused HIP dichromat path (through the C ABI) vs golden vectors and the oracle.
uint8 in, uint8 out: BIT-EXACT everywhere."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

GAUSS = ["dog", "cat", "squirrel", "elephant", "lion", "tiger", "bear", "wolf", "fox", "raccoon", "rat",
         "sheep", "pig", "cow", "goat", "horse", "rabbit", "panda", "deer", "kangaroo"]  # all 20 dichromats


@pytest.fixture(scope="module")
def av():
    import animal_vision_amd as av

    assert av.device_count() > 0, "no GPU visible"
    return av


def _species(av, name):
    from animal_vision_amd import animals

    sp = getattr(animals, name.capitalize())()
    if name == "cat":
        sp.ENABLE_FOV_WARP = False  # these tests pin the colour core (cat.py:95-109); the warp is in test_geometry_gpu.py
    return sp


@pytest.mark.parametrize("name", GAUSS)
def test_species_vs_golden(av, name):
    """Outputs of the reference's own <Species>.visualize (blur = shared OpenCV restatement)."""
    g = load_golden("dichromat")
    sp = _species(av, name)
    n = 0
    for k in ("n48", "s48", "n120", "dark"):
        key = f"{name}_oracle_{k}"
        if key not in g.files:
            continue
        frame = g[f"in_{k}"]
        base, out = sp.visualize(frame)
        assert (base is frame or name == "cat") and out.dtype == np.uint8 and out.shape == frame.shape
        assert np.array_equal(out, g[key]), f"{key}: {int((out != g[key]).sum())} bytes differ"
        n += 1
    assert n >= 3


@pytest.mark.parametrize("name", GAUSS)
@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (5, 4), (37, 91), (64, 64), (65, 129), (240, 320)])
def test_species_vs_oracle_shapes(av, oracle, name, shape):
    """Ragged tiles, frames smaller than the blur radius (multiple reflections), tile multiples."""
    H, W = shape
    frame = np.random.default_rng(1234 + H * 1000 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    _, want = oracle.dichromat_visualize(oracle.DICHROMATS[name], frame)
    _, got = _species(av, name).visualize(frame)
    assert np.array_equal(got, want), f"{name} {shape}: {int((got != want).sum())} bytes differ"


def test_dark_frame_and_mixed_batch(av, oracle):
    """a3 quirk: a frame whose bytes are all <= 1 is NOT divided by 255; decided per frame in a batch."""
    rng = np.random.default_rng(5)
    dark = rng.integers(0, 2, (33, 47, 3), dtype=np.uint8)
    zeros = np.zeros((33, 47, 3), np.uint8)
    normal = rng.integers(0, 256, (33, 47, 3), dtype=np.uint8)
    almost = dark.copy()
    almost[20, 30, 1] = 2  # one byte > 1 flips the whole frame to the /255 branch
    batch = np.stack([normal, dark, zeros, almost, dark])
    from animal_vision_amd.dichromat import DichromatOp
    from animal_vision_amd.animals import Dog, Cat

    for cls, nm in ((Dog, "dog"), (Cat, "cat")):
        got = DichromatOp(cls.SPEC)(batch)
        for i in range(batch.shape[0]):
            _, want = oracle.dichromat_visualize(oracle.DICHROMATS[nm], batch[i])
            assert np.array_equal(got[i], want), (nm, i)


def test_full_1080p_dog_and_cat_bit_exact(av, oracle):
    """BASELINE config sizes: one 1080p noise frame, bit-exact against the oracle."""
    from animal_vision_amd.synthetic import noise_frame

    frame = noise_frame(0, 1080, 1920)
    for nm in ("dog", "cat"):
        _, want = oracle.dichromat_visualize(oracle.DICHROMATS[nm], frame)
        _, got = _species(av, nm).visualize(frame)
        assert np.array_equal(got, want), f"{nm}: {int((got != want).sum())} bytes differ"


def test_constant_frame_is_preserved_in_shape_and_flat(av):
    """Size-independent property at 4K: a flat frame stays flat (blur of a constant; reflect-101 border)."""
    frame = np.full((2160, 3840, 3), (200, 120, 40), np.uint8)
    _, out = _species(av, "dog").visualize(frame)
    assert out.shape == frame.shape
    assert (out == out[0, 0]).all()


@pytest.mark.parametrize("name", ["dog", "cat", "wolf"])
def test_tuned_kernel_equals_reference_kernel(av, name):
    """A/B inside the library: every kernel variant produces the same bytes."""
    from animal_vision_amd import animals
    from animal_vision_amd.dichromat import DichromatOp

    spec = getattr(animals, name.capitalize()).SPEC
    rng = np.random.default_rng(77)
    batch = rng.integers(0, 256, (3, 200, 328, 3), dtype=np.uint8)
    batch[1] = rng.integers(0, 2, (200, 328, 3), dtype=np.uint8)  # an all-<=1 frame inside the batch
    outs = []
    for variant in (1, 2, 3, 0):  # reference, 2-D tiled, marching strip, auto
        op = DichromatOp(spec)
        op.desc.variant = variant
        outs.append(op(batch))
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


def test_unaligned_widths_and_device_offsets(av, oracle):
    """Widths with W*3 % 4 != 0 (byte-granular store path) and W % 16 != 0 (dword path)."""
    for (H, W) in [(70, 66), (70, 68), (33, 131), (129, 63)]:
        frame = np.random.default_rng(H * W).integers(0, 256, (H, W, 3), dtype=np.uint8)
        for nm in ("dog", "cat", "squirrel"):
            _, want = oracle.dichromat_visualize(oracle.DICHROMATS[nm], frame)
            _, got = _species(av, nm).visualize(frame)
            assert np.array_equal(got, want), (nm, H, W)


def test_bad_arguments_raise(av):
    from animal_vision_amd.animals import Dog

    with pytest.raises(AssertionError):
        Dog().visualize(np.zeros((4, 4), np.uint8))
    with pytest.raises(AssertionError):
        Dog().visualize(np.zeros((4, 4, 4), np.uint8))
    with pytest.raises(NotImplementedError):  # wider integers: not a device path (the reference's renderers never produce them)
        Dog().visualize(np.zeros((4, 4, 3), np.int32))
    from animal_vision_amd.animals import Cat

    with pytest.raises(NotImplementedError):
        Cat().visualize(np.zeros((4, 4, 3), np.int32))


@pytest.mark.parametrize("name", ["dog", "wolf", "rat", "squirrel", "sheep", "pig", "rabbit", "panda", "kangaroo"])
def test_float_frames_collapse_matrix_species(av, oracle, name):
    """Same-dtype contract for float frames: float32 in [0,1], float32 in [0,255], float64 (float pipeline:
    1e-4, the transfer functions are evaluated with the device's powf).  Gaussian, row-gain and streak species."""
    from animal_vision_amd import animals
    from animal_vision_amd.synthetic import structured_frame

    u8 = structured_frame(3, 72, 100)
    sp = getattr(animals, name.capitalize())()
    for frame in ((u8 / 255.0).astype(np.float32), u8.astype(np.float32), u8 / 255.0):
        base, out = sp.visualize(frame)
        _, want = oracle.dichromat_visualize(oracle.DICHROMATS[name], frame)
        assert base is frame and out.dtype == frame.dtype == want.dtype and out.shape == frame.shape
        np.testing.assert_allclose(out, want, rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", ["cat", "dog", "wolf", "squirrel"])
def test_marching_kernel_ragged_strips_and_misaligned_frames(av, oracle, name):
    """Shapes that stay on the marching-strip kernel (batch bytes a multiple of 4, frames taller than 2 (R + 4) rows) but stress
    its addressing: row lengths that are not multiples of 4 bytes (per-row dword misalignment), several ragged strips, and a
    batch whose individual frames have an odd byte size (every second frame base is misaligned)."""
    from animal_vision_amd import animals
    from animal_vision_amd.dichromat import DichromatOp

    spec = getattr(animals, name.capitalize()).SPEC
    for (n, H, W) in [(1, 96, 250), (1, 52, 121 * 4), (4, 45, 33), (2, 40, 1001 * 2), (1, 1080 // 4, 1922)]:
        assert (n * H * W * 3) % 4 == 0
        batch = np.random.default_rng(n * H + W).integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        op = DichromatOp(spec)
        op.desc.variant = 3
        got = op(batch)
        for i in range(n):
            _, want = oracle.dichromat_visualize(oracle.DICHROMATS[name], batch[i])
            assert np.array_equal(got[i], want), f"{name} frame {i} of {(n, H, W)}: {int((got[i] != want).sum())} bytes differ"


@pytest.mark.parametrize("warp", [False, True])
def test_cat_float_frames(av, oracle, warp):
    """Same-dtype contract for Cat on float frames: float32 in [0,1] and [0,255], float64; zoomed baseline + cat
    view (with and without the binocular FOV warp) against the oracle's cat_visualize, 1e-4 (float pipeline)."""
    from animal_vision_amd.animals import Cat
    from animal_vision_amd.synthetic import structured_frame

    u8 = structured_frame(5, 72, 100)
    cat = Cat()
    cat.ENABLE_FOV_WARP = warp
    for frame in ((u8 / 255.0).astype(np.float32), u8.astype(np.float32), u8 / 255.0):
        base, out = cat.visualize(frame)
        want_base, want = oracle.cat_visualize(frame, enable_fov_warp=warp)
        assert out.dtype == frame.dtype == want.dtype and out.shape == frame.shape and base.dtype == frame.dtype
        np.testing.assert_allclose(out, want, rtol=0, atol=1e-4)
        np.testing.assert_allclose(base, want_base, rtol=0, atol=1e-4 * max(1.0, float(frame.max())))
