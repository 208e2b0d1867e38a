"""animals/deer.py of the reference: `from animals.deer import Deer` keeps working."""
from ._dichromats import Deer  # noqa: F401
