"""Importable name of the package: the layout prescribes the directory `distributed-drl_amd/`, which is not a Python identifier, so this
package's search path points there — `distributed_drl_amd.replay` IS `distributed-drl_amd/replay.py` — and it carries the same export table."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "distributed-drl_amd")]

from ._exports import __version__, make_getattr  # noqa: E402,F401
from . import _lib  # noqa: E402,F401
from . import remote  # noqa: E402,F401

__getattr__ = make_getattr(__name__)
