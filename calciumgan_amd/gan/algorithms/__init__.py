"""Mirror of gan/algorithms/__init__.py (registers 'gan' and 'wgan-gp')."""
from .registry import get_algorithm, register

__all__ = ['get_algorithm', 'register']

from . import gan  # noqa: E402,F401
from . import wgan_gp  # noqa: E402,F401
