"""animals/kangaroo.py of the reference: `from animals.kangaroo import Kangaroo` keeps working."""
from ._dichromats import Kangaroo  # noqa: F401
