"""Importable alias for the package directory `sdxl-training-improvements_amd/`."""
import importlib
import sys
from pathlib import Path

_root = str(Path(__file__).resolve().parent)
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("sdxl-training-improvements_amd")
sys.modules[__name__] = _pkg
