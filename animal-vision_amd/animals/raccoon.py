"""animals/raccoon.py of the reference: `from animals.raccoon import Raccoon` keeps working."""
from ._dichromats import Raccoon  # noqa: F401
