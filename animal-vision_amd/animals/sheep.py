"""animals/sheep.py of the reference: `from animals.sheep import Sheep` keeps working."""
from ._dichromats import Sheep  # noqa: F401
