"""ImageRenderer -- still-image source/sink (reference: renderers/image.py:28-260) on Pillow, since cv2 is absent:
get_image() -> RGB uint8 HxWx3, render() saves to `save_to` and remembers the frame (`visualized_image`), render_split_compare()
composes the half-and-half frame with its two corner labels (renderers/video.py's composer: halves, seam, Hershey-simplex labels on
the device).  The GUI preview (`show_window`, cv2.imshow) is out of scope: asking for it raises."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .renderer import Renderer
from .video import split_compose


class ImageRenderer(Renderer):
    def __init__(self, path: Optional[str] = None, *, show_window: bool = False, save_to: Optional[str] = None, wait_key: int = 0,
                 window_name: str = "Image"):
        self.path, self.save_to = path, save_to
        self.show_window, self.wait_key, self.window_name = bool(show_window), wait_key, window_name
        self.visualized_image: Optional[np.ndarray] = None  # the last frame render() saw (image.py:135)

    def get_image(self) -> Optional[np.ndarray]:
        if not self.path:
            return None
        from PIL import Image

        return np.asarray(Image.open(self.path).convert("RGB"))

    def open(self) -> None:
        """image.py:102-113 prepares the preview window; there is none here."""
        if self.show_window:
            raise NotImplementedError("ImageRenderer(show_window=True): the GUI preview (cv2.imshow) is out of scope; use save_to")

    def render(self, frame: np.ndarray) -> None:
        """image.py:115-147 without the window: remember the frame, save it when `save_to` is set."""
        if self.show_window:
            raise NotImplementedError("ImageRenderer(show_window=True): the GUI preview (cv2.imshow) is out of scope; use save_to")
        self.visualized_image = frame
        if self.save_to:
            from PIL import Image

            Image.fromarray(frame).save(self.save_to)

    def close(self) -> None:
        """image.py:149-160: nothing to tear down."""

    def send_image(self, image: np.ndarray) -> None:
        """image.py:163-172: alias of render()."""
        self.render(image)

    def render_split_compare(self, original, modified, *, left_label="Original", right_label="Transformed", draw_seam=True):
        """image.py:216-260: left half original, right half modified (resized when the sizes differ), seam, the two labels; then render()."""
        assert isinstance(original, np.ndarray) and original.ndim == 3 and original.shape[2] == 3, "original must be an HxWx3 RGB image"
        assert isinstance(modified, np.ndarray) and modified.ndim == 3 and modified.shape[2] == 3, "modified must be an HxWx3 RGB image"
        self.render(split_compose(original, modified, draw_seam, left_label=left_label, right_label=right_label))
