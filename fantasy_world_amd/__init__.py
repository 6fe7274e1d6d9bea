"""Import shim: `import fantasy_world_amd` -> the package that lives in ../fantasy-world_amd/ (hyphenated dir)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fantasy-world_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
