"""ImageRenderer -- still-image source/sink (reference: renderers/image.py:65,115,216) on Pillow, since cv2 is
absent: get_image() -> RGB uint8 HxWx3, render() saves to `save_to`."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .renderer import Renderer
from .video import split_compose


class ImageRenderer(Renderer):
    def __init__(self, path: Optional[str] = None, *, show_window: bool = False, save_to: Optional[str] = None, wait_key: int = 0,
                 window_name: str = "Image"):
        self.path, self.save_to = path, save_to
        self.show_window, self.wait_key, self.window_name = show_window, wait_key, window_name

    def get_image(self) -> Optional[np.ndarray]:
        if not self.path:
            return None
        from PIL import Image

        return np.asarray(Image.open(self.path).convert("RGB"))

    def render(self, frame: np.ndarray) -> None:
        if self.save_to:
            from PIL import Image

            Image.fromarray(frame).save(self.save_to)

    def render_split_compare(self, original, modified, *, left_label="Original", right_label="Transformed", draw_seam=True):
        self.render(split_compose(original, modified, draw_seam))
