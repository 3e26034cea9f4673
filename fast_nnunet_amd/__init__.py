"""Importable alias of the ``fast-nnunet_amd/`` source directory.

The package lives in ``fast-nnunet_amd/`` (the name the project layout asks for);
a hyphen is not a valid Python identifier, so this stub makes the same modules
importable as ``fast_nnunet_amd.<module>``.
"""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'fast-nnunet_amd')
if not _os.path.isdir(_src):
    raise ImportError(f'{_src} is missing')
__path__.append(_src)

from .predictor import nnUNetPredictor  # noqa: E402,F401
from .sliding_window import compute_gaussian, compute_steps_for_sliding_window  # noqa: E402,F401
