"""Import alias: ``import bathymetric_gnn_amd`` -> the package in ``bathymetric-gnn_amd/``.

The package directory carries the project's name (with a hyphen, which Python's ``import``
statement cannot spell); this module gives it an importable name by adopting that
directory as its ``__path__`` and executing the package's ``__init__.py`` in place.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "bathymetric-gnn_amd")]
__package__ = "bathymetric_gnn_amd"
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init, "r") as _f:
    exec(compile(_f.read(), _init, "exec"), globals())
del _f, _init
