"""Mirror of gan/models/__init__.py: importing the package registers the
models.  (The reference also imports conv1d/conv2d/rnn modules that do not
exist in its tree -- SURVEY Appendix D.1; only the 1-D calciumgan is in scope.)"""
from .registry import get_models, register

__all__ = ['get_models', 'register']

from . import calciumgan  # noqa: E402,F401  (registers 'calciumgan')
