"""ctypes binding of libavx.so (C ABI: include/avx.h).  No torch types cross this boundary.

The library is the product: if it is missing this module raises at import -- there is no CPU path."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libavx.so")

AVX_OK = 0
AVX_ERR_INVALID, AVX_ERR_NO_DEVICE, AVX_ERR_HIP, AVX_ERR_UNSUPPORTED, AVX_ERR_NOMEM = -1, -2, -3, -4, -5
AVX_COLOR_MATRIX, AVX_COLOR_CAT_MERGE = 0, 1
AVX_POST_NONE, AVX_POST_GAUSS, AVX_POST_ROWGAIN, AVX_POST_STREAK = 0, 1, 2, 3
AVX_MAX_KSIZE = 33


class AvxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libavx error {code}: {msg}")
        self.code = code


class DichromatDesc(ctypes.Structure):
    """avx_dichromat_desc (include/avx.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("color_mode", ctypes.c_int32),
        ("matrix", ctypes.c_float * 9),
        ("cat_alpha", ctypes.c_float),
        ("cat_beta", ctypes.c_float),
        ("post_mode", ctypes.c_int32),
        ("ksize", ctypes.c_int32),
        ("taps_host", ctypes.POINTER(ctypes.c_double)),
        ("row_gain_host", ctypes.POINTER(ctypes.c_float)),
        ("row_gain_clamp", ctypes.c_int32),
        ("chroma_enable", ctypes.c_int32),
        ("chroma_keep", ctypes.c_float),
        ("variant", ctypes.c_int32),
        ("streak_rows_host", ctypes.POINTER(ctypes.c_float)),
        ("streak_stride", ctypes.c_int32),
        ("in_f32", ctypes.c_int32),
    ]


class HoneybeeDesc(ctypes.Structure):
    """avx_honeybee_desc (include/avx.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("source", ctypes.c_int32),
        ("rgb_matrix", ctypes.c_float * 9),
        ("hsi", ctypes.c_void_p),
        ("hsi_layout", ctypes.c_int32),
        ("hsi_dtype", ctypes.c_int32),
        ("bands", ctypes.c_int32),
        ("weights_host", ctypes.POINTER(ctypes.c_float)),
        ("adaptation", ctypes.c_int32),
        ("eps", ctypes.c_float),
        ("blur_ksize", ctypes.c_int32),
        ("blur_taps_host", ctypes.POINTER(ctypes.c_double)),
        ("mapping", ctypes.c_int32),
        ("custom_matrix", ctypes.c_float * 9),
        ("mixed_alpha", ctypes.c_float),
        ("out_float", ctypes.c_int32),
        ("catches", ctypes.c_void_p),
        ("catch_partials", ctypes.c_void_p),
        ("n_catch_partials", ctypes.c_int32),
    ]


_fp, _dp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)


class MantisDesc(ctypes.Structure):
    """avx_mantis_desc (include/avx.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_bands", ctypes.c_int32),
        ("band_matrix_host", _fp),
        ("band_lut_host", _fp),
        ("n_wavelengths", ctypes.c_int32),
        ("lobe_gains_host", _fp),
        ("lobe_denom", ctypes.c_float),
        ("band_weights_host", _fp),
        ("pano_new_w", ctypes.c_int32),
        ("hsi_small_h", ctypes.c_int32),
        ("hsi_small_w", ctypes.c_int32),
        ("red_keep", ctypes.c_float),
        ("haze", ctypes.c_float),
        ("haze_keep", ctypes.c_float),
        ("haze_tint", ctypes.c_float * 3),
        ("pre_soft_ksize", ctypes.c_int32),
        ("pre_soft_taps_host", _dp),
        ("cos2_global", ctypes.c_float),
        ("sin2_global", ctypes.c_float),
        ("orientation_mix", ctypes.c_float),
        ("pol_linear_strength", ctypes.c_float),
        ("pol_linear_gamma", ctypes.c_float),
        ("pol_circular_strength", ctypes.c_float),
        ("unsharp_ksize", ctypes.c_int32),
        ("unsharp_taps_host", _dp),
        ("unsharp_amount", ctypes.c_float),
        ("barcode_saturation", ctypes.c_float),
        ("barcode_opacity", ctypes.c_float),
        ("winner_take_most", ctypes.c_float),
        ("rows_host", _fp),
        ("scan_row_gain", ctypes.c_float),
        ("scan_ksize", ctypes.c_int32),
        ("scan_taps_host", _dp),
        ("periph_ksize", ctypes.c_int32),
        ("periph_taps_host", _dp),
        ("xx_host", _fp),
        ("yy_host", _fp),
        ("periph_radius", ctypes.c_float),
        ("periph_softness", ctypes.c_float),
        ("lin_hwc_in", ctypes.c_void_p),
        ("out_float", ctypes.c_int32),
    ]


class BandStackDesc(ctypes.Structure):
    """avx_band_stack_desc (include/avx.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_bands", ctypes.c_int32),
        ("band_matrix_host", _fp),
        ("n_wavelengths", ctypes.c_int32),
        ("lobe_gains_host", _fp),
        ("lobe_denom", ctypes.c_float),
        ("band_weights_host", _fp),
        ("small_h", ctypes.c_int32),
        ("small_w", ctypes.c_int32),
    ]


class EwInsn(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint8), ("dst", ctypes.c_uint8), ("a", ctypes.c_uint8), ("b", ctypes.c_uint8), ("imm", ctypes.c_uint32)]


class EwPlane(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("stride", ctypes.c_int32), ("kind", ctypes.c_int32)]


class EwProgram(ctypes.Structure):
    """avx_ew_program (include/avx.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("n_insn", ctypes.c_int32),
        ("insn_host", ctypes.POINTER(EwInsn)),
        ("n_planes", ctypes.c_int32),
        ("planes_host", ctypes.POINTER(EwPlane)),
        ("n_acc", ctypes.c_int32),
        ("acc_host", ctypes.POINTER(ctypes.c_int32)),
        ("scalars_dev", ctypes.c_void_p),
        ("n_scalars", ctypes.c_int32),
    ]


AVX_EW_MAX_INSN, AVX_EW_MAX_PLANES, AVX_EW_MAX_REGS, AVX_EW_MAX_ACC = 384, 24, 32, 16
_EW_OPS = ("CONST SCALAR LOAD STORE ADD SUB MUL DIV MIN MAX POW ATAN2 NEG ABS SQRT EXP LOG SIN COS FLOOR CEIL CLIP01 TANH "
           "LT LE GT GE EQ AND OR NOT SELECT ACCMIN ACCMAX ACCSUM").split()
EW = {name: i + 1 for i, name in enumerate(_EW_OPS)}  # enum in include/avx.h starts at AVX_EW_CONST = 1
EW_PLANE = {"f32": 0, "u8": 1, "u8_lut": 2, "col": 3, "row": 4, "u8_enc": 5}
EW_ACC = {"min": 0, "max": 1, "sum": 2, "mean": 3}

AVX_MAP = {"falsecolor": 0, "custom_matrix": 1, "opponent": 2, "uv_purple_yellow": 3, "falsecolor_uv_mixed": 4}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C animal-vision_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback."
    )


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as the one in
    /opt/rocm that libavx.so is linked against); whichever loads first wins the SONAME, and if it is the
    system copy, torch afterwards initialises a second runtime and reports "No HIP GPUs are available".
    Loading torch's copy first (without importing torch) makes libavx.so and torch share it."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


_preload_torch_hip_runtime()
lib = ctypes.CDLL(LIB_PATH)

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_SIGS = {
    "avx_abi_version": (_i, []),
    "avx_device_count": (_i, []),
    "avx_init": (_i, [_i, ctypes.POINTER(_vp)]),
    "avx_destroy": (None, [_vp]),
    "avx_last_error": (ctypes.c_char_p, [_vp]),
    "avx_malloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "avx_free": (_i, [_vp, _vp]),
    "avx_host_alloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "avx_host_free": (_i, [_vp, _vp]),
    "avx_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "avx_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "avx_memset": (_i, [_vp, _vp, _i, _sz, _vp]),
    "avx_stream_create": (_i, [_vp, ctypes.POINTER(_vp)]),
    "avx_stream_destroy": (_i, [_vp, _vp]),
    "avx_sync": (_i, [_vp, _vp]),
    "avx_stream_wait": (_i, [_vp, _vp, _vp]),
    "avx_device_sync": (_i, [_vp]),
    "avx_timer_start": (_i, [_vp, _vp]),
    "avx_timer_stop": (_i, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    "avx_dichromat_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, ctypes.POINTER(DichromatDesc), _vp]),
    "avx_get_table": (_i, [_i, _vp, _sz]),
    "avx_percentile": (_i, [_vp, _vp, _sz, ctypes.c_double, ctypes.POINTER(ctypes.c_double), _vp]),
    "avx_spectral_integrate": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "avx_plane_stats": (_i, [_vp, _vp, _i, _sz, _i, ctypes.c_float, _vp, _vp]),
    "avx_planes_gaussian_blur": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "avx_streak_planes_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "avx_rgb_to_hsi_lobes": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_float, _vp, _vp]),
    "avx_honeybee_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, ctypes.POINTER(HoneybeeDesc), _vp, _vp]),
    "avx_uv_front_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "avx_panorama_warp_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "avx_band_stack": (_i, [_vp, _vp, _i, _i, ctypes.POINTER(BandStackDesc), _vp, _vp]),
    "avx_percentile_dev": (_i, [_vp, _vp, _sz, ctypes.c_double, _vp, _vp]),
    "avx_percentiles_dev": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(ctypes.c_void_p), _vp]),
    "avx_ew_run": (_i, [_vp, ctypes.POINTER(EwProgram), _vp]),
    "avx_ew_spec_stats": (_i, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]),
    "avx_mantis_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, ctypes.POINTER(MantisDesc), _vp]),
    "avx_mantis_u8_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, ctypes.POINTER(MantisDesc), _vp]),
    "avx_resize_hwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "avx_binocular_warp_u8": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "avx_split_compose_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "avx_draw_label_u8": (_i, [_vp, _vp, _i, _i, ctypes.POINTER(ctypes.c_int), _fp, _i, ctypes.c_float, ctypes.c_float, _i, _vp]),
    "avx_remap_linear_planes": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, ctypes.c_float, _vp]),
    "avx_sobel3_plane": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "avx_mst_gram": (_i, [_vp, _vp, _i, _sz, _i, _i, _vp, _vp, _vp, _vp]),
    "avx_mst_qkv_gram": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "avx_mst_qkv_gram16": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp, _vp, _vp, _vp]),
    "avx_mst_ln_gemm_gelu": (_i, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _sz, _i, _vp, _vp]),
    "avx_mst_convt2x2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_convt2x2_fuse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_posemb": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_conv3x3_add": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_attn_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "avx_mst_rowgemm_add": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "avx_mst_conv_in_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "avx_mst_conv3x3_lds": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_down4x4": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_conv3x3_lds_gram": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "avx_mst_convt2x2_fuse_gram": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "avx_mst_attn_pack16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "avx_mst_attn_pack_mx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "avx_mst_attn_tail_mx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_attn_tail": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_attn_tail_x": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_ffn_fused": (_i, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_conv3x3_lds_spectral": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _fp, _i, _i, _i, _i, _vp, _vp, ctypes.POINTER(ctypes.c_int), _vp]),
    "avx_mst_ffn_fused_mx": (_i, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_mst_gelu_prescale": (ctypes.c_float, []),
    "avx_mst_dw_gemm_add": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "avx_dwconv3x3_nhwc_add": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "avx_dwconv3x3_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "avx_layernorm_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _i, ctypes.c_float, _vp]),
    "avx_layernorm_rows_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _i, ctypes.c_float, _i, _i, _vp]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = stale libavx.so: rebuild
    _fn.restype = _res
    _fn.argtypes = _args

if lib.avx_abi_version() != 1:
    raise ImportError(f"libavx.so ABI {lib.avx_abi_version()} != 1 expected by this package: rebuild")
