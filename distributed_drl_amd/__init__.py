"""Import alias: the package directory is `distributed-drl_amd/` (not an importable name);
this shim makes `import distributed_drl_amd` resolve to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "distributed-drl_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f
