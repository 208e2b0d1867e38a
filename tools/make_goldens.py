#!/usr/bin/env python3
"""tools/make_goldens.py -- generate tests/golden/*.npz by running the REFERENCE.

Runs only in the build container (needs /root/reference; never on the GPU box).
The reference's source files are imported from where they lie and executed; only
their inputs and outputs are written out (small fixtures).  Harness tricks
(SURVEY.md 8c):
  * `cv2`, `colour` are not installed -> empty placeholder modules in sys.modules;
    cv2.GaussianBlur / cv2.Sobel are injected where a fixture needs the blur
    (identity, or this repo's OpenCV-semantics restatement: PARITY UNPINNED);
  * `animals/__init__.py` cannot be imported (animals/cat.py holds merge-conflict
    markers) -> a synthetic `animals` package with the real __path__;
  * classic_rgb_to_hsi's analytic branch is guarded by torch.cuda.is_available()
    -> the module's `torch` global is swapped for a proxy that reports CUDA and
    maps device="cuda" to CPU tensors, so the reference's own lines 47-82 run.

Usage: python tools/make_goldens.py [--only NAME] (writes tests/golden/).
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import cpu_ref as O  # noqa: E402  (only for the injected OpenCV-semantics blur)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_synth = _load("_avx_synth", os.path.join(ROOT, "animal-vision_amd", "synthetic.py"))
noise_frame, structured_frame = _synth.noise_frame, _synth.structured_frame

# ---- placeholder third-party modules ---------------------------------------
cv2 = types.ModuleType("cv2")
cv2.BORDER_REFLECT101 = 4
cv2.BORDER_REFLECT_101 = 4
cv2.BORDER_DEFAULT = 4
cv2.CV_32F = 5
BLUR_MODE = {"mode": "oracle"}


def _GaussianBlur(src, ksize, sigmaX, sigmaY=0.0, borderType=4, dst=None):
    if BLUR_MODE["mode"] == "identity":
        return src.copy()
    return O.cv_gaussian_blur(src, tuple(ksize), float(sigmaX), float(sigmaY))


cv2.GaussianBlur = _GaussianBlur
# geometry: this repo's OpenCV-semantics restatements, PARITY UNPINNED
cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA = 0, 1, 2, 3
cv2.BORDER_CONSTANT = 0
cv2.resize = lambda src, dsize, interpolation=1, **kw: O.cv_resize(src, dsize, interpolation)
cv2.remap = lambda src, m1, m2, interpolation=1, borderMode=0, borderValue=0, **kw: O.cv_remap_linear(src, m1, m2, float(borderValue))
cv2.Sobel = lambda src, ddepth, dx, dy, ksize=3, scale=1.0, delta=0.0, borderType=4: O.cv_sobel3(src, dx, dy)
sys.modules["cv2"] = cv2
sys.modules["colour"] = types.ModuleType("colour")

sys.path.insert(0, REF)
animals_pkg = types.ModuleType("animals")
animals_pkg.__path__ = [os.path.join(REF, "animals")]
sys.modules["animals"] = animals_pkg

import importlib  # noqa: E402

ref_au = importlib.import_module("animals.animal_utils")
ref_uvh = importlib.import_module("uv_helpers")
ref_uvm = importlib.import_module("uv_mappers")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)/1024:.1f} KiB)")


# =============================================================================
def g_srgb_tables():
    """a3+a4 decode LUT; a11 encode thresholds (f32 dichromat, f64 cat tail, f32 UV)."""
    codes = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    lut = ref_au.srgb_to_linear(ref_au.get_normalized_image(codes))[..., 0].reshape(256)
    assert lut.dtype == np.float32
    # to_float01 + uv_helpers.srgb_to_linear (mantis/reindeer route) on u8 codes
    lut_uv = ref_uvh.srgb_to_linear(ref_uvh.to_float01(codes))[..., 0].reshape(256)

    def enc_dog(x):  # animals/dog.py:54-57
        s = np.clip(ref_au.linear_to_srgb(np.clip(x, 0.0, 1.0)), 0.0, 1.0)
        return (s * 255.0 + 0.5).astype(np.uint8)

    def enc_bee(x):  # animals/honeybee.py:166-171
        s = ref_uvh.linear_to_srgb(np.clip(x, 0.0, 1.0))
        return (s * 255.0 + 0.5).astype(np.uint8)

    def enc_ff01(x):  # animals/mantis_shrimp.py:278 (from_float01)
        return ref_uvh.from_float01(ref_uvh.linear_to_srgb(np.clip(x, 0.0, 1.0)), np.uint8)

    thr32 = O.encode_thresholds(np.float32, lambda x: enc_dog(np.asarray(x, np.float32)))
    thr64 = O.encode_thresholds(np.float64, lambda x: enc_dog(np.asarray(x, np.float64)))
    thr_bee = O.encode_thresholds(np.float32, lambda x: enc_bee(np.asarray(x, np.float32)))
    thr_ff = O.encode_thresholds(np.float32, lambda x: enc_ff01(np.asarray(x, np.float32)))
    rng = np.random.default_rng(7)
    ramp = np.concatenate([rng.random(1 << 16, dtype=np.float32), np.linspace(0, 1, 4097, dtype=np.float32),
                           np.array([-0.5, 0.0, 1.0, 1.5, 0.0031308, 0.0031309], np.float32)])
    near = np.concatenate([np.nextafter(thr32, np.float32(0)), thr32, np.nextafter(thr32, np.float32(2))]).astype(np.float32)
    ramp = np.concatenate([ramp, near])
    ramp64 = np.concatenate([ramp.astype(np.float64), rng.random(1 << 14), np.nextafter(thr64, 0.0), thr64])
    save("srgb_tables", decode_lut=lut, decode_lut_uv=lut_uv, enc_thr_f32=thr32, enc_thr_f64=thr64,
         enc_thr_bee_f32=thr_bee, enc_thr_ff01_f32=thr_ff,
         ramp_f32=ramp, ramp_f32_u8=enc_dog(ramp), ramp_f32_bee_u8=enc_bee(ramp), ramp_f32_ff01_u8=enc_ff01(ramp),
         ramp_f64=ramp64, ramp_f64_u8=enc_dog(ramp64))
    print("   thresholds equal (dog f32 == bee f32):", np.array_equal(thr32, thr_bee), " (== from_float01):", np.array_equal(thr32, thr_ff))


def g_matrices():
    """a5/a6: collapse_LMS_matrix for every (alpha, s) of SURVEY Appendix A."""
    pairs = sorted({(s.alpha, s.s_scale) for s in O.DICHROMATS.values()} | {(0.45, 0.80)})
    mats = np.stack([ref_au.collapse_LMS_matrix(a, s) for a, s in pairs])
    E = np.eye(3, dtype=np.float32)
    save("collapse_matrices", pairs=np.array(pairs, np.float64), T=mats,
         lms_of_eye=ref_au.sRGB_to_LMS(E), rgb_of_eye=ref_au.LMS_to_RGB(E))


def _species(name):
    mod = importlib.import_module(f"animals.{name}")
    cls = [v for k, v in vars(mod).items() if isinstance(v, type) and k.lower() == name.replace("_", "")][0]
    return cls()


def g_dichromat():
    """a1-a11: <Species>.visualize of the reference on seeded frames, with the blur
    (cv2.GaussianBlur) injected as identity and as this repo's OpenCV restatement."""
    frames = {"n48": noise_frame(0, 48, 64), "s48": structured_frame(0, 48, 64), "n120": noise_frame(1, 120, 160),
              "dark": (noise_frame(2, 24, 32) // 255).astype(np.uint8)}  # all <= 1: the a3 "no /255" branch
    out = {f"in_{k}": v for k, v in frames.items()}
    for name in sorted(O.DICHROMATS):
        if name == "cat":
            continue  # animals/cat.py does not parse (F2); its colour core is re-enacted below
        sp = _species(name)
        for mode in ("identity", "oracle"):
            BLUR_MODE["mode"] = mode
            for k, f in frames.items():
                if k == "n120" and name not in ("dog", "wolf", "sheep"):
                    continue  # keep the fixture small: the big frame only for three species
                base, res = sp.visualize(f.copy())
                assert base.dtype == f.dtype and res.dtype == f.dtype
                out[f"{name}_{mode}_{k}"] = res
    # Cat colour core, cat.py:95-103,109 (Tina-animals side, Q8) re-enacted with the
    # reference's own helpers; ENABLE_FOV_WARP path excluded.
    for mode in ("identity", "oracle"):
        BLUR_MODE["mode"] = mode
        for k, f in frames.items():
            H, W = f.shape[:2]
            cat01 = ref_au.get_normalized_image(f)
            lin = ref_au.srgb_to_linear(cat01)
            vec = lin.reshape(-1, 3)
            lms = ref_au.sRGB_to_LMS(vec)
            alpha = 0.5
            LM = alpha * lms[:, 0] + (1.0 - alpha) * lms[:, 1]
            merged = np.stack([LM, LM, lms[:, 2]], axis=1)
            lin_rgb = ref_au.LMS_to_RGB(merged).reshape(H, W, 3)
            assert lin_rgb.dtype == np.float64
            lin_rgb = ref_au.apply_acuity_blur(lin_rgb, sigma=1.0)
            cat_srgb = np.clip(ref_au.linear_to_srgb(np.clip(lin_rgb, 0.0, 1.0)), 0.0, 1.0)
            out[f"cat_{mode}_{k}"] = (cat_srgb * 255.0 + 0.5).astype(np.uint8)
    # float-input contract (dtype == input dtype): dog on a float32 [0,1] frame
    BLUR_MODE["mode"] = "oracle"
    f32 = (frames["n48"].astype(np.float32) / 255.0)
    out["in_f32"] = f32
    out["dog_oracle_f32"] = _species("dog").visualize(f32.copy())[1]
    # intermediate: linear colour stage of dog on n48 (pins the FMA-chain matmul)
    lin = ref_au.srgb_to_linear(ref_au.get_normalized_image(frames["n48"]))
    out["dog_colorstage_n48"] = (lin.reshape(-1, 3) @ ref_au.collapse_LMS_matrix(0.58, 0.65).T).reshape(lin.shape)
    # a10 helpers on a seeded linear image
    x = np.random.default_rng(3).random((20, 24, 3), dtype=np.float32)
    out["helper_in"] = x
    out["chroma_0p4"] = ref_au.apply_chroma_compression(x.copy(), 0.4)
    out["scone_rat"] = ref_au.apply_s_cone_vertical_gain(x.copy(), s_top=1.3, s_bottom=0.5, power=1.4, extra_boost=0.25)
    out["scone_band"] = ref_au.apply_s_cone_vertical_gain(x.copy(), 1.0, 0.6, band=(0.4, 0.2, 0.5), clamp=False)
    save("dichromat", **out)


def g_bloom_rod():
    """a10's two helpers that no species calls (animal_utils.py:183-204 apply_tapetum_bloom, :261-305 apply_rod_vision; cat.py:50-59
    names them in a commented block): the reference's own lines on seeded linear images, cv2.GaussianBlur injected as this repo's
    OpenCV restatement (PARITY UNPINNED for the blur itself, as everywhere).  Inputs inside and outside [0, 1], float32 and float64."""
    BLUR_MODE["mode"] = "oracle"
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:49, 0:67]
    img = (0.5 + 0.45 * np.sin(xx / 7.0)[..., None] * np.cos(yy[..., None] / 5.0 + np.arange(3)) + 0.05 * rng.standard_normal((49, 67, 3))).astype(np.float32)
    frames = {"a": img, "b": (img * 1.3 - 0.1).astype(np.float32), "c": img.astype(np.float64)}  # the test derives b and c from in_a the same way
    out = {"in_a": img}
    for k, f in frames.items():
        for j, kw in enumerate(({}, dict(strength=0.3, sigma=1.5))):
            out[f"bloom_{k}_{j}"] = ref_au.apply_tapetum_bloom(f.copy(), **kw)
        for j, kw in enumerate(({}, dict(chroma_scale=0.15, luminance_boost=1.1, gamma=0.6))):
            out[f"rod_{k}_{j}"] = ref_au.apply_rod_vision(f.copy(), **kw)
        assert out[f"bloom_{k}_0"].dtype == f.dtype and out[f"rod_{k}_0"].dtype == f.dtype
    save("bloom_rod", **out)


def g_uv():
    """a12, a15-a18, a20, a21 + hsv/snow-glare on seeded planes (pure NumPy reference code)."""
    rng = np.random.default_rng(11)
    out = {}
    lam31 = np.linspace(400.0, 700.0, 31, dtype=np.float32)
    lam81 = np.linspace(300.0, 700.0, 81, dtype=np.float32)
    out["lam31"], out["lam81"] = lam31, lam81
    out["d65_31"], out["d65_81"] = ref_uvh.D65_like(lam31), ref_uvh.D65_like(lam81)
    bands = [(320.0, 360.0), (360.0, 400.0), (400.0, 430.0), (610.0, 680.0), (300.0, 400.0), (100.0, 200.0), (405.0, 406.0), (400.0, 400.0)]
    out["bp_bands"] = np.array(bands)
    out["bp_31"] = np.stack([ref_uvh.bandpass_weights(lam31, lo, hi) for lo, hi in bands])
    out["bp_81"] = np.stack([ref_uvh.bandpass_weights(lam81, lo, hi) for lo, hi in bands])
    cube = rng.random((18, 22, 31), dtype=np.float32)
    out["cube31"] = cube
    out["ib_31"] = np.stack([ref_uvh.integrate_band(cube, lam31, lo, hi) for lo, hi in bands[:5]])
    out["iuv_31"] = ref_uvh.integrate_uv(cube, lam31, 400.0, 460.0)
    U, B, G = (rng.random((36, 44), dtype=np.float32) ** 2 for _ in range(3))
    U = U * 0.3
    out["U"], out["B"], out["G"] = U, B, G
    out["safe_norm_U"] = ref_uvh.safe_norm(U)
    out["safe_norm_const"] = ref_uvh.safe_norm(np.full((4, 5), 0.25, np.float32))
    for nm, fn in (("wp", ref_uvh.von_kries_white_patch), ("gw", ref_uvh.von_kries_gray_world)):
        r = fn(U, B, G)
        out[f"vk_{nm}"] = np.stack(r)
    out["map_falsecolor"] = ref_uvm.map_falsecolor(U, B, G)
    out["map_opponent"] = ref_uvm.map_opponent(U, B, G)
    out["map_upy"] = ref_uvm.map_uv_purple_yellow(U)
    out["map_upy_soft"] = ref_uvm.map_uv_purple_yellow_soft(U)
    out["map_mixed_035"] = ref_uvm.map_falsecolor_uv_mixed(U, B, G)
    out["map_mixed_045"] = ref_uvm.map_falsecolor_uv_mixed(U, B, G, alpha=0.45)
    M = rng.random((3, 3)).astype(np.float32)
    out["M"], out["map_matrix"] = M, ref_uvm.map_linear_matrix(U, B, G, M)
    hsv = rng.random((16, 16, 3), dtype=np.float32)
    hsv[0, 0] = (1.0, 0.5, 0.5)  # floor(6h) == 6 -> i_mod == 0
    out["hsv"], out["hsv_rgb"] = hsv, ref_uvm.hsv_to_rgb(hsv)
    x = rng.random((10, 12, 3), dtype=np.float32) * 1.2
    out["glare_in"], out["glare_out"] = x, ref_uvh.snow_glare_tone_compress(x, strength=0.7)
    f = noise_frame(5, 8, 9)
    out["tf01_u8_in"], out["tf01_u8"] = f, ref_uvh.to_float01(f)
    f255 = f.astype(np.float32)
    out["tf01_f255"] = ref_uvh.to_float01(f255)
    out["ff01_u8"] = ref_uvh.from_float01(ref_uvh.to_float01(f) * 0.9, np.uint8)
    save("uv_helpers", **out)


class _TorchCudaAsCpu:
    """Proxy for the `torch` global of classic_rgb_to_hsi.py: CUDA 'available', tensors on CPU."""

    def __init__(self, real):
        self._r = real
        self.cuda = types.SimpleNamespace(is_available=lambda: True)

    def __getattr__(self, n):
        return getattr(self._r, n)

    def as_tensor(self, data, dtype=None, device=None):
        return self._r.as_tensor(data, dtype=dtype)


def _ref_classic():
    import torch

    mod = importlib.import_module("ml.classic_rgb_to_hsi.classic_rgb_to_hsi")
    mod.torch = _TorchCudaAsCpu(torch)
    return mod.classic_rgb_to_hsi


def g_lobes():
    """a13: the reference's analytic branch (classic_rgb_to_hsi.py:47-82) on float frames."""
    conv = _ref_classic()
    rng = np.random.default_rng(21)
    img = rng.random((12, 14, 3), dtype=np.float32)
    img[0, 0] = (0.0, 0.04045, 1.0)
    out = {"img": img}
    for nm, lam in (("31", np.linspace(400.0, 700.0, 31, dtype=np.float32)),
                    ("81", np.linspace(300.0, 700.0, 81, dtype=np.float32)),
                    ("129", np.linspace(320.0, 700.0, 129))):
        out[f"lam{nm}"] = lam
        out[f"hsi{nm}"] = conv(img, wavelengths=lam)
    save("lobes", **out)


def g_honeybee():
    """a12-a21 end to end: the reference HoneyBee class, every mapping mode."""
    _ref_classic()
    hb = importlib.import_module("animals.honeybee")
    hb.classic_rgb_to_hsi = sys.modules["ml.classic_rgb_to_hsi.classic_rgb_to_hsi"].classic_rgb_to_hsi
    ref_uvh.cv2 = cv2  # uv_helpers.gaussian_blur takes its cv2 branch (injected blur)
    frames = {"s40": structured_frame(0, 40, 56), "n40": noise_frame(3, 40, 56)}
    out = {f"in_{k}": v for k, v in frames.items()}
    BLUR_MODE["mode"] = "oracle"
    M = np.array([[0.9, 0.1, 0.0], [0.1, 0.2, 0.7], [0.3, 0.6, 0.1]], np.float32)
    out["custom_matrix"] = M
    for mode in ("opponent", "falsecolor", "uv_purple_yellow", "falsecolor_uv_mixed", "custom_matrix"):
        for adapt in ("white_patch", "gray_world"):
            bee = hb.HoneyBee(mapping_mode=mode, adaptation=adapt, custom_matrix=M if mode == "custom_matrix" else None)
            for k, f in frames.items():
                base, res = bee.visualize(f)
                assert base is f and res.dtype == np.uint8
                out[f"{mode}_{adapt}_{k}"] = res
    bee = hb.HoneyBee(blur_sigma_px=0.0)
    out["opponent_noblur_s40"] = bee.visualize(frames["s40"])[1]
    # a14: the hsi_downsample route (INTER_AREA down, lobes, INTER_LINEAR up of the 31-band cube)
    for sc_name, sc in (("025", 0.25), ("060", 0.6)):
        bee = hb.HoneyBee(hsi_downsample=True, hsi_scale=sc)
        for k, f in frames.items():
            out[f"downsample{sc_name}_{k}"] = bee.visualize(f)[1]
    # intermediates for the default species on s40: catches after adaptation + blur
    bee = hb.HoneyBee()
    img01 = ref_uvh.to_float01(frames["s40"])
    hsi = hb.classic_rgb_to_hsi(img01, wavelengths=bee.lambdas)
    rad = hsi * bee.E(bee.lambdas).astype(hsi.dtype)[None, None, :]
    U = np.tensordot(rad, bee.UV_curve, axes=([2], [0]))
    B = np.tensordot(rad, bee.Blue_curve, axes=([2], [0]))
    G = np.tensordot(rad, bee.Green_curve, axes=([2], [0]))
    out["catches_s40"] = np.stack([U, B, G])
    out["curves"] = np.stack([bee.UV_curve, bee.Blue_curve, bee.Green_curve])
    ref_uvh.cv2 = None
    save("honeybee", **out)


def g_mstpp():
    """a25: reference MST_Plus_Plus with torch.manual_seed(0) weights (rounded to fp16 and
    stored), outputs on two small frames; a26: pad/crop/tile index math of predict_torch.py."""
    import torch

    arch = _load("_ref_mstpp", os.path.join(REF, "ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py"))
    torch.manual_seed(0)
    model = arch.MST_Plus_Plus().eval()
    sd = {k: v.half().float() for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    rng = np.random.default_rng(31)
    out = {}
    for nm, (h, w) in (("60x70", (60, 70)), ("64x64", (64, 64))):
        x = rng.random((1, 3, h, w), dtype=np.float32)
        with torch.no_grad():
            y = model(torch.from_numpy(x)).numpy()
        out[f"x_{nm}"], out[f"y_{nm}"] = x, y
    save("mstpp_io", **out)
    save("mstpp_weights_fp16", **{k: v.numpy().astype(np.float16) for k, v in sd.items()})
    print("   params:", sum(v.numel() for v in sd.values()), "tensors:", len(sd))
    # predict_torch.py helpers (pure NumPy); stub its architecture import
    stub = types.ModuleType("ml.MST_plus_plus.predict_code.architecture")
    stub.model_generator = lambda *a, **k: None
    for n in ("ml.MST_plus_plus", "ml.MST_plus_plus.predict_code"):
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = [os.path.join(REF, *n.split(".")[0:1], *n.split(".")[1:])]
            sys.modules[n] = m
    sys.modules["ml.MST_plus_plus.predict_code.architecture"] = stub
    pt = _load("_ref_predict_torch", os.path.join(REF, "ml/MST_plus_plus/predict_code/predict_torch.py"))
    o2 = {}
    img = rng.random((37, 45, 3), dtype=np.float32)
    pad, pads = pt._pad_to_multiple_reflect(img, 16)
    o2["img"], o2["pad16"], o2["pads16"] = img, pad, np.array(pads)
    o2["crop16"] = pt._crop_pads(pad, pads)
    o2["tiles_300_500_256_64"] = np.array(pt._tile_coords(300, 500, 256, 64))
    o2["tiles_200_1100_256_64"] = np.array(pt._tile_coords(200, 1100, 256, 64))
    o2["hann_8_6"] = pt._hann2d(8, 6)
    o2["tf01_u8"] = pt._to_float01(noise_frame(9, 5, 6))
    save("predict_torch_helpers", **o2)


def g_mstpp_large():
    """a25 above toy sizes: the reference module (same seeded fp16-rounded weights as g_mstpp) on structured frames of
    256x256, 512x512 and 1920x1080 -- the normalise and k @ q^T of MS_MSA (:127-129) reduce over EVERY pixel, so the
    Gram/norm accumulations are only exercised at size.  Stored: the whole 256x256 output (float16 container), three
    64x64 crops of the 512x512 and 1080p outputs (float32), every block's attention matrix (the softmax output of
    :131, captured by wrapping Tensor.softmax while the reference runs), and -- where this torch accepts CPU float16
    autocast, the precision predict_torch.py:109 runs at -- the error of that autocast output against the float32 one
    (the yardstick the GPU float16 path is held to).  Inputs are synthetic.structured_frame(seed, H, W) / 255 (not stored)."""
    import time

    import torch

    arch = _load("_ref_mstpp", os.path.join(REF, "ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py"))
    torch.manual_seed(0)
    model = arch.MST_Plus_Plus().eval()
    sd = {k: v.half().float() for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    out = {}
    real_softmax = torch.Tensor.softmax
    for nm, seed, (h, w), crops in (("256", 5, (256, 256), None), ("512", 6, (512, 512), ((0, 0), (200, 300), (448, 448))),
                                    ("1080p", 7, (1080, 1920), ((0, 0), (500, 900), (1016, 1856)))):
        x = (structured_frame(seed, h, w).astype(np.float32) / 255.0).transpose(2, 0, 1)[None]
        attn = []

        def spy(self, *a, **k):
            r = real_softmax(self, *a, **k)
            attn.append(r.detach().float().numpy().copy())
            return r

        torch.Tensor.softmax = spy
        t0 = time.time()
        try:
            with torch.no_grad():
                y = model(torch.from_numpy(x)).numpy()[0]  # (31, h, w)
        finally:
            torch.Tensor.softmax = real_softmax
        print(f"   {nm}: reference float32 forward {time.time() - t0:.1f} s, |y| mean {np.abs(y).mean():.4f} max {np.abs(y).max():.4f}")
        assert len(attn) == 15
        out[f"seed_{nm}"] = np.array([seed, h, w])
        for i, a in enumerate(attn):
            out[f"attn_{nm}_{i}"] = a[0]  # (heads, 31, 31)
        if crops is None:
            out[f"y_{nm}"] = y.astype(np.float16)
        else:
            out[f"crops_{nm}"] = np.array(crops)
            for j, (cy, cx) in enumerate(crops):
                out[f"y_{nm}_crop{j}"] = y[:, cy:cy + 64, cx:cx + 64].copy()
        out[f"ymeanabs_{nm}"] = np.array([np.abs(y).mean(), np.sqrt((y.astype(np.float64) ** 2).mean())])
        if nm != "1080p":
            try:
                with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
                    ya = model(torch.from_numpy(x)).float().numpy()[0]
                d = np.abs(ya - y)
                out[f"autocast_err_{nm}"] = np.array([d.max(), d.mean(), np.sqrt((d.astype(np.float64) ** 2).mean())])
                print(f"   {nm}: CPU float16 autocast vs float32: max {d.max():.3e} mean {d.mean():.3e}")
            except Exception as e:  # noqa: BLE001
                print(f"   {nm}: CPU float16 autocast not available here: {type(e).__name__}: {e}")
    save("mstpp_large", **out)


def g_mstpp_4k():
    """a25 at the HEADLINE size: the reference module (same seeded fp16-rounded weights as g_mstpp) on ONE structured
    3840x2160 frame in float32 on the CPU.  Stored: three 64x64 crops of the output (float32), the 15 attention matrices
    and the output's mean |y| / rms -- the Gram / norm sums of MS_MSA (:127-129) run over 8.3 M pixels here, four times
    longer than anything g_mstpp_large pins.  Separate fixture (mstpp_4k.npz) so the other sizes need not be re-run."""
    import resource
    import time

    import torch

    arch = _load("_ref_mstpp", os.path.join(REF, "ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py"))
    torch.manual_seed(0)
    model = arch.MST_Plus_Plus().eval()
    sd = {k: v.half().float() for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    out = {}
    real_softmax = torch.Tensor.softmax
    nm, seed, (h, w), crops = "4k", 8, (2160, 3840), ((0, 0), (1000, 1900), (2096, 3776))
    x = (structured_frame(seed, h, w).astype(np.float32) / 255.0).transpose(2, 0, 1)[None]
    attn = []

    def spy(self, *a, **k):
        r = real_softmax(self, *a, **k)
        attn.append(r.detach().float().numpy().copy())
        return r

    torch.Tensor.softmax = spy
    t0 = time.time()
    try:
        with torch.no_grad():
            y = model(torch.from_numpy(x)).numpy()[0]  # (31, h, w)
    finally:
        torch.Tensor.softmax = real_softmax
    print(f"   {nm}: reference float32 forward {time.time() - t0:.1f} s, |y| mean {np.abs(y).mean():.4f} max {np.abs(y).max():.4f}, "
          f"peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.1f} GiB")
    assert len(attn) == 15
    out[f"seed_{nm}"] = np.array([seed, h, w])
    for i, a in enumerate(attn):
        out[f"attn_{nm}_{i}"] = a[0]
    out[f"crops_{nm}"] = np.array(crops)
    for j, (cy, cx) in enumerate(crops):
        out[f"y_{nm}_crop{j}"] = y[:, cy:cy + 64, cx:cx + 64].copy()
    out[f"ymeanabs_{nm}"] = np.array([np.abs(y).mean(), np.sqrt((y.astype(np.float64) ** 2).mean())])
    save("mstpp_4k", **out)


def g_geometry():
    """8f row 1 helpers of the reference (uv_helpers.panorama_warp, classic_rgb_to_hsi_scaled, cat FOV helpers)
    driven with the injected resize/remap, plus the full Cat (cat.py:73-112 re-enacted: the file does not parse)."""
    conv = _ref_classic()
    ref_uvh.cv2 = cv2
    cw = importlib.import_module("animals.cat_widevision_utils")
    assert cw._HAS_CV2
    rng = np.random.default_rng(41)
    out = {}
    lin = rng.random((40, 56, 3), dtype=np.float32)
    out["lin"] = lin
    for sname, sc in (("112", 1.12), ("145", 1.45), ("105", 1.05)):
        out[f"pano_{sname}"] = ref_uvh.panorama_warp(lin, scale_x=sc)
    lam81 = np.linspace(300.0, 700.0, 81, dtype=np.float32)
    out["hsi_scaled_025"] = ref_uvh.classic_rgb_to_hsi_scaled(lin, wavelengths=lam81, scale=0.25, converter=lambda a, w: conv(a, wavelengths=w))
    lin2 = rng.random((50, 70, 3), dtype=np.float32)
    out["lin2"] = lin2
    out["hsi_scaled_025_ragged"] = ref_uvh.classic_rgb_to_hsi_scaled(lin2, wavelengths=lam81, scale=0.25, converter=lambda a, w: conv(a, wavelengths=w))
    out["zoom_scale"] = np.array(cw.zoom_scale_from_cat_ratio(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, cat_to_human_ratio=1.30))
    frames = {"n48": noise_frame(0, 48, 64), "s60": structured_frame(2, 60, 84)}
    for k, f in frames.items():
        out[f"in_{k}"] = f
        H, W = f.shape[:2]
        scale = cw.zoom_scale_from_cat_ratio(camera_hfov_deg=100.0, cat_per_eye_half_fov_deg=105.0, cat_to_human_ratio=1.30)
        human_zoomed = cw.center_zoom(f, scale=scale)
        cat01 = ref_au.get_normalized_image(f)
        cat01 = cw.animal_fov_binocular_warp(cat01.astype(np.float32), fov_in_deg=100.0, per_eye_half_fov_deg=105.0, overlap_deg=40.0,
                                             out_size=(W, H), border_mode=0, border_value=0.0)
        out[f"cat_warp01_{k}"] = cat01
        BLUR_MODE["mode"] = "oracle"
        lin_c = ref_au.srgb_to_linear(cat01)
        lms = ref_au.sRGB_to_LMS(lin_c.reshape(-1, 3))
        LM = 0.5 * lms[:, 0] + (1.0 - 0.5) * lms[:, 1]
        lin_rgb = ref_au.LMS_to_RGB(np.stack([LM, LM, lms[:, 2]], axis=1)).reshape(H, W, 3)
        lin_rgb = ref_au.apply_acuity_blur(lin_rgb, sigma=1.0)
        cat_srgb = np.clip(ref_au.linear_to_srgb(np.clip(lin_rgb, 0.0, 1.0)), 0.0, 1.0)
        out[f"cat_human_{k}"] = human_zoomed
        out[f"cat_out_{k}"] = (cat_srgb * 255.0 + 0.5).astype(np.uint8)
    ref_uvh.cv2 = None
    save("geometry", **out)


def g_mantis():
    """a22/a23 + geometry: the reference MantisShrimp class end to end (default parameters and a no-resample variant)."""
    _ref_classic()
    ref_uvh.cv2 = cv2
    BLUR_MODE["mode"] = "oracle"
    ms = importlib.import_module("animals.mantis_shrimp")
    ms.classic_rgb_to_hsi = sys.modules["ml.classic_rgb_to_hsi.classic_rgb_to_hsi"].classic_rgb_to_hsi
    ms.cv2 = cv2
    frames = {"s64": structured_frame(3, 64, 80), "n50": noise_frame(7, 50, 70)}
    out = {f"in_{k}": v for k, v in frames.items()}
    for tag, kw in (("default", {}), ("noresample", dict(hsi_scale=1.0, panorama_scale=1.0))):
        m = ms.MantisShrimp(**kw)
        for k, f in frames.items():
            base, res = m.visualize(f)
            assert base.dtype == np.uint8 and res.dtype == np.uint8
            out[f"{tag}_base_{k}"], out[f"{tag}_out_{k}"] = base, res
    ref_uvh.cv2 = None
    save("mantis", **out)


UV_SPECIES = (("reindeer", "Reindeer"), ("rat_uv", "RatUV"), ("goldfish", "Goldfish"), ("damselfish", "Damselfish"), ("anableps", "Anableps"),
              ("anchovy", "Anchovy"), ("guppy", "Guppy"), ("morpho", "Morpho"), ("heliconius", "Heliconius"), ("pieris", "Pieris"),
              ("hummingbird", "Hummingbird"), ("kestrel", "Kestrel"), ("jumping_spider", "JumpingSpider"), ("dragonfly", "Dragonfly"))


def g_uv_species():
    """8f row 3: the 14 remaining UV species of the reference, default parameters, end to end on two small frames
    (cv2 = the oracle's OpenCV restatements, classic_rgb_to_hsi = the reference's analytic branch on CPU)."""
    conv_mod = None
    _ref_classic()
    conv_mod = sys.modules["ml.classic_rgb_to_hsi.classic_rgb_to_hsi"]
    ref_uvh.cv2 = cv2
    BLUR_MODE["mode"] = "oracle"
    frames = {"s64": structured_frame(3, 64, 80), "n50": noise_frame(7, 50, 70)}
    out = {f"in_{k}": v for k, v in frames.items()}
    for mod, cls in UV_SPECIES:
        try:
            m = importlib.import_module(f"animals.{mod}")
        except Exception as e:  # noqa: BLE001
            print(f"   {mod}: import failed: {type(e).__name__}: {e}")
            continue
        if hasattr(m, "classic_rgb_to_hsi"):
            m.classic_rgb_to_hsi = conv_mod.classic_rgb_to_hsi
        for attr in ("cv2", "cv"):
            if hasattr(m, attr):
                setattr(m, attr, cv2)
        for flag in ("_HAS_CV2", "HAS_CV2", "_HAS_CV"):
            if hasattr(m, flag):
                setattr(m, flag, True)
        try:
            sp = getattr(m, cls)()
            for k, f in frames.items():
                base, res = sp.visualize(f)
                assert base.dtype == np.uint8 and res.dtype == np.uint8, (mod, base.dtype, res.dtype)
                out[f"{mod}_base_{k}"], out[f"{mod}_out_{k}"] = base, res
                if mod == "rat_uv":  # the branch `mode="auto"` does not take on these frames
                    out[f"{mod}_night_out_{k}"] = sp.visualize(f, mode="night")[1]
            print(f"   {mod}: ok")
        except Exception as e:  # noqa: BLE001
            import traceback
            print(f"   {mod}: FAILED {type(e).__name__}: {e}")
            traceback.print_exc(limit=3)
    ref_uvh.cv2 = None
    save("uv_species", **out)


GENERATORS = {"srgb_tables": g_srgb_tables, "matrices": g_matrices, "dichromat": g_dichromat, "bloom_rod": g_bloom_rod, "uv": g_uv,
              "lobes": g_lobes, "honeybee": g_honeybee, "mstpp": g_mstpp, "mstpp_large": g_mstpp_large, "mstpp_4k": g_mstpp_4k, "geometry": g_geometry, "mantis": g_mantis, "uv_species": g_uv_species}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for name, fn in GENERATORS.items():
        if a.only and a.only != name:
            continue
        print(f"[{name}]")
        fn()
