"""Species registry; same export names as the reference's animals/__init__.py:1-32."""
from .animal import Animal  # noqa: F401
from ._dichromats import (  # noqa: F401
    Bear, Cat, Cow, Deer, Dog, Elephant, Fox, Goat, Horse, Kangaroo, Lion, Panda, Pig, Rabbit, Raccoon,
    Rat, Sheep, Squirrel, Tiger, Wolf,
)
from .honeybee import HoneyBee  # noqa: F401
from .mantis_shrimp import MantisShrimp  # noqa: F401
from .reindeer import Reindeer  # noqa: F401
from .goldfish import Goldfish  # noqa: F401

# module name -> class name of the UV species written against the plane-program backend (planevm.py)
from .damselfish import Damselfish  # noqa: F401
from .rat_uv import RatUV  # noqa: F401
from .anableps import Anableps  # noqa: F401
from .anchovy import Anchovy  # noqa: F401
from .guppy import Guppy  # noqa: F401
from .morpho import Morpho  # noqa: F401
from .heliconius import Heliconius  # noqa: F401
from .pieris import Pieris  # noqa: F401
from .hummingbird import Hummingbird  # noqa: F401
from .kestrel import Kestrel  # noqa: F401
from .jumping_spider import JumpingSpider  # noqa: F401
from .dragonfly import Dragonfly  # noqa: F401

UV_CLASS = {"reindeer": "Reindeer", "goldfish": "Goldfish", "damselfish": "Damselfish", "rat_uv": "RatUV", "anableps": "Anableps",
            "anchovy": "Anchovy", "guppy": "Guppy", "morpho": "Morpho",
            "heliconius": "Heliconius", "pieris": "Pieris", "hummingbird": "Hummingbird",
            "kestrel": "Kestrel", "jumping_spider": "JumpingSpider", "dragonfly": "Dragonfly"}
