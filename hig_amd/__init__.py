"""Importable alias of the `human-interaction-generation_amd/` package (a hyphen cannot appear
in a Python module name).  All code lives there; this shim only redirects the package path."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "human-interaction-generation_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
