"""Importable alias of the ``ipc-filecoin-proofs_amd/`` package directory.

Python cannot import a directory whose name contains a hyphen, so this shim
points its ``__path__`` at the real package directory next to it and re-exports
its public names.  There is no code here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ipc-filecoin-proofs_amd")
__path__.insert(0, _real)

from .binding import *  # noqa: E402,F401,F403
from .binding import __all__ as _all  # noqa: E402

__all__ = list(_all)
