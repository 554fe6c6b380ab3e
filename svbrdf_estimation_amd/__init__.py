"""svbrdf_estimation_amd -- MI355X-native SVBRDF rendering-loss engine.

Drop-in replacement for ONE hot path of mworchel/svbrdf-estimation
(development/multiImage_pytorch): ``renderers.LocalRenderer.render`` and
``losses.RenderingLoss.forward`` (+ backward), computed by hand-written gfx950 HIP
kernels behind the C ABI in ``include/svbrdf_hip.h``.  Module names mirror the
reference's flat modules so a training script only changes its imports:

    from svbrdf_estimation_amd import renderers, losses, environment, utils

(The directory is spelled with an underscore because ``svbrdf-estimation_amd`` is
not an importable Python identifier.)

There is no CPU implementation in this package: every compute entry point raises if
the HIP extension is missing or the tensors are not on a ROCm device.
"""
from . import distributed, environment, losses, renderers, synthesis, utils  # noqa: F401
from ._native import NativeLibraryError, library_path  # noqa: F401

__all__ = ["environment", "losses", "renderers", "synthesis", "utils", "distributed", "NativeLibraryError",
           "library_path"]
__version__ = "0.1.0"
