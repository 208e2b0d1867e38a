"""Renderer ABC -- same surface as the reference's renderers/renderer.py:6-29."""
from abc import ABC, abstractmethod

import numpy as np


class Renderer(ABC):
    def open(self) -> None:
        """Optional: allocate resources."""
        pass

    @abstractmethod
    def render(self, frame: np.ndarray) -> None:
        """Display or output one frame."""
        ...

    def render_split_compare(self, original: np.ndarray, modified: np.ndarray, *, left_label: str = "Original",
                             right_label: str = "Transformed", draw_seam: bool = True):
        pass

    def close(self) -> None:
        """Optional: release resources."""
        pass
