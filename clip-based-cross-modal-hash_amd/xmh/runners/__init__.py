"""Importing the package registers the in-scope runners, like the reference's ``runners/__init__.py``."""
from .base import BaseTrainer  # noqa: F401
from .methods import DCMHTTrainer, DSPHTrainer, MITHTrainer, TwDHTrainer  # noqa: F401
