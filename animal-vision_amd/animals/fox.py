"""animals/fox.py of the reference: `from animals.fox import Fox` keeps working."""
from ._dichromats import Fox  # noqa: F401
