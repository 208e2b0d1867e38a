"""animal-vision on MI355X: the per-frame `Animal.visualize` hot path as hand-written HIP kernels
for gfx950 behind the reference's plugin surface (animals/animal.py, renderers/renderer.py).

Import name: `animal_vision_amd` (the directory is `animal-vision_amd/`; the repo-root shim
`animal_vision_amd.py` registers it).  Importing loads libavx.so and fails loudly if it is missing;
creating a Context fails loudly if there is no GPU.  There is no CPU fallback anywhere."""
from . import _lib  # noqa: F401  (raises ImportError when libavx.so is absent)
from ._lib import AvxError  # noqa: F401
from .runtime import Context, DeviceBuffer, device_count, get_context, get_table  # noqa: F401
from .dichromat import DichromatOp, DichromatSpec, collapse_LMS_matrix, cv_auto_ksize, gaussian_taps  # noqa: F401

from . import uv  # noqa: F401

__version__ = "0.1.0"
