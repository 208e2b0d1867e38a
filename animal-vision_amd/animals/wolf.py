"""animals/wolf.py of the reference: `from animals.wolf import Wolf` keeps working."""
from ._dichromats import Wolf  # noqa: F401
