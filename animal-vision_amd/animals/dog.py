"""animals/dog.py of the reference: `from animals.dog import Dog` keeps working."""
from ._dichromats import Dog  # noqa: F401
