"""laed_amd -- MI355X-native DI-VAE / DI-VST train-step engine behind the LAED model API.

Module layout mirrors the reference package `laed` (models.sent_models, enc2dec.{encoders,decoders},
nn_lib, criterions, utils, main) so `from laed...` imports port by renaming the package.
All arithmetic runs in hand-written HIP kernels (csrc/, gfx950) through the C ABI in include/laed_hip.h.
"""
from . import _lib  # noqa: F401  (fails loudly when liblaed_hip.so is missing)

__version__ = "0.1.0"
