"""Import shim: the product package lives in the directory `ot-gan_amd/` (the name the
project layout prescribes), which is not a valid Python identifier.  This stub makes it
importable as `otgan_amd` by pointing the package search path at that directory and
executing its __init__.py in this namespace."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ot-gan_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
