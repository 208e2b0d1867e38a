"""animals/cow.py of the reference: `from animals.cow import Cow` keeps working."""
from ._dichromats import Cow  # noqa: F401
