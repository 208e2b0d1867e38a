"""animals/bear.py of the reference: `from animals.bear import Bear` keeps working."""
from ._dichromats import Bear  # noqa: F401
