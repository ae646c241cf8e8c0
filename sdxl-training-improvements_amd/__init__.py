"""MI355X-native SDXL training step (gfx950 HIP kernels behind the reference's trainer-plugin surface).

The directory name is not a valid Python identifier; import it with
    importlib.import_module("sdxl-training-improvements_amd")      or      import sdxl_amd   (alias at repo root)
"""
from . import lib  # noqa: F401
from . import unet  # noqa: F401
from . import synth  # noqa: F401
from . import distributed  # noqa: F401

__all__ = ["lib", "unet", "synth", "distributed"]
