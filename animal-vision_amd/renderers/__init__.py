from .renderer import Renderer  # noqa: F401
from .video import VideoRenderer, split_compose  # noqa: F401
from .image import ImageRenderer  # noqa: F401
