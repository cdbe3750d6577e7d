"""Importable alias for the ``ecg-multimodal-model_amd/`` package directory (a hyphenated directory
name cannot be imported directly): ``import ecgmm.multimodal`` resolves to
``ecg-multimodal-model_amd/multimodal.py``."""
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ecg-multimodal-model_amd")
__path__.insert(0, _pkg)
with open(_os.path.join(_pkg, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg, "__init__.py"), "exec"))
