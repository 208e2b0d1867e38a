"""animals/elephant.py of the reference: `from animals.elephant import Elephant` keeps working."""
from ._dichromats import Elephant  # noqa: F401
