"""animals/rabbit.py of the reference: `from animals.rabbit import Rabbit` keeps working."""
from ._dichromats import Rabbit  # noqa: F401
