"""Importing the package registers the in-scope models, like the reference's ``models/__init__.py``."""
from .base import BaseModel  # noqa: F401
from .dcmht import DCMHT  # noqa: F401
from .dsph import DSPH  # noqa: F401
from .mith import MITH  # noqa: F401
from .twdh import TwDH  # noqa: F401
