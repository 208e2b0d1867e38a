"""animals/goat.py of the reference: `from animals.goat import Goat` keeps working."""
from ._dichromats import Goat  # noqa: F401
