"""animals/cat.py of the reference: `from animals.cat import Cat` keeps working."""
from ._dichromats import Cat  # noqa: F401
