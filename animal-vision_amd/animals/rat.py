"""animals/rat.py of the reference: `from animals.rat import Rat` keeps working."""
from ._dichromats import Rat  # noqa: F401
