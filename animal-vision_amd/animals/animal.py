"""Plugin base class -- signature kept verbatim from the reference (animals/animal.py:5-7)."""
from typing import Optional, Tuple

import numpy as np


class Animal:
    def visualize(self, image: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        pass
