"""Import alias: the product package lives in the directory `csm-hf_amd/` (the name the build contract
asks for), which is not a valid Python identifier.  This shim makes it importable as `csm_hf_amd` by
pointing the package search path at that directory and running its `__init__`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "csm-hf_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
