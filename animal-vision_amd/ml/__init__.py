"""RGB -> hyperspectral stage: MST++ on PyTorch-ROCm (the only torch-backed compute in the package)."""
from .mst_plus_plus import MSTPlusPlus, reference_key_names  # noqa: F401
from .predict import MSTPlusPlusPredictor, MstHoneybeeStreamOp, pad_to_multiple_reflect, crop_pads, to_float01  # noqa: F401
