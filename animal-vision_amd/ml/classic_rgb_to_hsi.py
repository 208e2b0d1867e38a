"""The reference's ml/classic_rgb_to_hsi/classic_rgb_to_hsi.py by name: `classic_rgb_to_hsi(frame, wavelengths=...)` -> HxWxB float32
cube, the analytic-lobe branch (the one the reference runs on a GPU box, :47-82) on the device (csrc/uv.hip::avx_rgb_to_hsi_lobes).
The `colour`-library CPU branch (:92-111, Mallett 2019) is not restated: its dependency is absent here and unpinned in the reference."""
from ..uv import classic_rgb_to_hsi  # noqa: F401
