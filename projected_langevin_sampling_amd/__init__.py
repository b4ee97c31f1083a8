"""Import alias: ``import projected_langevin_sampling_amd`` loads the package that lives in the
directory ``projected-langevin-sampling_amd/`` (a hyphen cannot appear in a Python module name)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "projected-langevin-sampling_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
