"""animals/pig.py of the reference: `from animals.pig import Pig` keeps working."""
from ._dichromats import Pig  # noqa: F401
