"""animals/panda.py of the reference: `from animals.panda import Panda` keeps working."""
from ._dichromats import Panda  # noqa: F401
