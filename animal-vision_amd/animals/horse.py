"""animals/horse.py of the reference: `from animals.horse import Horse` keeps working."""
from ._dichromats import Horse  # noqa: F401
