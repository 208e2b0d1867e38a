"""animals/squirrel.py of the reference: `from animals.squirrel import Squirrel` keeps working."""
from ._dichromats import Squirrel  # noqa: F401
