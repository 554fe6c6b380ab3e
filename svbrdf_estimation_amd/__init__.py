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
the HIP extension is missing or no ROCm device is there to compute on.  (A HOST tensor handed
to ``LocalRenderer.render`` -- the reference dataloader's call -- is staged to the GPU and
rendered there; the losses take device tensors only.)
"""
from . import distributed, environment, losses, renderers, synthesis, utils  # noqa: F401
from ._native import NativeLibraryError, library_path  # noqa: F401

__all__ = ["environment", "losses", "renderers", "synthesis", "utils", "distributed", "NativeLibraryError",
           "library_path", "install"]

# hot-path classes the reference's scripts reach by name (main.py:8,12 `from losses import MixedLoss`,
# `from renderers import LocalRenderer, RednerRenderer`; dataset.py:206 `renderers.LocalRenderer()`)
_PATCHED = {"renderers": ("LocalRenderer",), "losses": ("SVBRDFL1Loss", "RenderingLoss", "MixedLoss")}


def install(modules=None, patch_renderer=True):
    """make the REFERENCE'S OWN flat modules hand out this engine's hot-path classes,
    without editing its files.  Call it once, before the training script's ``from losses import MixedLoss`` /
    ``from renderers import LocalRenderer`` run (top of main.py, or a sitecustomize):

        import svbrdf_estimation_amd; svbrdf_estimation_amd.install()

    It imports the reference's ``renderers`` and ``losses`` (they must be importable, i.e. the reference's directory
    is on sys.path) and rebinds ``LocalRenderer`` / ``SVBRDFL1Loss`` / ``RenderingLoss`` / ``MixedLoss`` in them;
    everything else in those modules (RednerRenderer, OrthoToPerspectiveMapping, ...) and the reference's
    ``environment`` / ``utils`` / ``dataset`` / ``models`` stay the reference's.  ``modules`` (for tests): a dict
    {"renderers": module, "losses": module} to patch instead of importing by name.  Returns what it replaced.

    The default serves ``main.py`` unchanged: the losses take the fused kernel, and the reference's dataloader call
    ``LocalRenderer().render(scene, svbrdf.unsqueeze(0))`` with HOST tensors in the main process (dataset.py:94-98,
    :206-212; main.py:63 uses num_workers=0) is staged to the GPU, rendered by K1 and handed back as a CPU tensor
    (renderers._HostStaging).  ``patch_renderer=False`` leaves ``renderers.LocalRenderer`` the reference's -- for a machine
    without a GPU in the data process, or forked DataLoader workers (a forked child cannot use the GPU; this engine has
    no CPU path) -- while the patched ``RenderingLoss`` / ``MixedLoss`` still take the fused kernel: they recognise the
    reference's ``LocalRenderer`` object as the renderer the kernels replace (``losses.RenderingLoss.uses_fused_kernel``)."""
    import importlib
    mine = {"renderers": renderers, "losses": losses}
    replaced = {}
    for mname, names in _PATCHED.items():
        if mname == "renderers" and not patch_renderer:
            continue
        target = modules[mname] if modules is not None else importlib.import_module(mname)
        if target is mine[mname]:
            continue
        for name in names:
            replaced["%s.%s" % (mname, name)] = getattr(target, name, None)
            setattr(target, name, getattr(mine[mname], name))
    return replaced


__version__ = "0.1.0"
