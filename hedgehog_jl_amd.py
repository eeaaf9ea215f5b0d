"""Import shim: the package directory is `hedgehog.jl_amd/` (a dot cannot appear in a Python
module name), so `import hedgehog_jl_amd` resolves here and is redirected to that directory."""
import os as _os

__package__ = __name__
__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "hedgehog.jl_amd")]
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
