"""animals/tiger.py of the reference: `from animals.tiger import Tiger` keeps working."""
from ._dichromats import Tiger  # noqa: F401
