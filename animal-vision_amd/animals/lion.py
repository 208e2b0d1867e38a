"""animals/lion.py of the reference: `from animals.lion import Lion` keeps working."""
from ._dichromats import Lion  # noqa: F401
