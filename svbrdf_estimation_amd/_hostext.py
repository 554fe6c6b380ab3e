"""Loader of the optional native host path (csrc/host_ext.cpp).

The extension moves the per-call host work of the fused loss -- scene sampling, table upload,
kernel launch, autograd node -- out of the Python interpreter; it launches exactly the same
HIP kernels through the same C ABI as the ctypes binding, so results are identical.  It is
built in-tree by ``__graft_entry__.build()``; when it is absent (or ``SVBRDF_NO_HOST_EXT=1``)
the ctypes path in ``_native.py`` is used.  Neither path computes anything on the CPU.
"""
import importlib.machinery
import importlib.util
import os

import torch  # noqa: F401  (the extension links against libtorch)

from . import _native

NAME = "svbrdf_host_ext"
BUILD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "host_ext")
# SVBRDF_HOST_EXT_SO: load another build of the same extension (the AddressSanitizer / UBSan build of tests/test_sanitizers.py)
_SO = os.environ.get("SVBRDF_HOST_EXT_SO") or os.path.join(BUILD_DIR, NAME + ".so")
_mod = None
_tried = False


def build(verbose=False):
    """compile csrc/host_ext.cpp with torch.utils.cpp_extension into lib/host_ext/ (in-tree)"""
    from torch.utils import cpp_extension
    os.makedirs(BUILD_DIR, exist_ok=True)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "host_ext.cpp")
    cpp_extension.load(name=NAME, sources=[src], build_directory=BUILD_DIR, extra_cflags=["-O2", "-std=c++17", "-ffp-contract=off"],
                       extra_ldflags=["-ldl"], verbose=verbose)
    return _SO


def build_sanitized(build_dir, verbose=False):
    """the same source under -fsanitize=address,undefined into `build_dir` (CPU test infrastructure: the build is loaded
    into a child interpreter that has the ASan runtime preloaded, never on a GPU box).  Compiled with the command
    torch.utils.cpp_extension generates for the product build (lib/host_ext/build.ninja) plus the sanitizer flags, but by
    a plain compiler call: cpp_extension.load would also dlopen the result into THIS process, which has no ASan runtime."""
    import subprocess
    import sysconfig
    from torch.utils import cpp_extension
    os.makedirs(build_dir, exist_ok=True)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "host_ext.cpp")
    out = os.path.join(build_dir, NAME + ".so")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["c++", "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H"]
    for inc in cpp_extension.include_paths() + [sysconfig.get_paths()["include"]]:
        cmd += ["-isystem", inc]
    cmd += ["-fPIC", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fno-omit-frame-pointer",
            "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
            "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2", "-shared", src, "-o", out, "-ldl"]
    for lib in cpp_extension.library_paths():
        cmd += ["-L" + lib, "-Wl,-rpath," + lib]
    cmd += ["-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return out


_disabled = False


def set_enabled(flag):
    """switch between the native host path and the ctypes path at run time (tests compare the two)"""
    global _disabled
    _disabled = not flag


def module():
    """the bound extension module, or None if it is not built / disabled"""
    global _mod, _tried
    if _disabled:
        return None
    if _mod is not None or _tried:
        return _mod
    _tried = True
    if os.environ.get("SVBRDF_NO_HOST_EXT") or not os.path.exists(_SO):
        return None
    loader = importlib.machinery.ExtensionFileLoader(NAME, _SO)
    spec = importlib.util.spec_from_loader(NAME, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    _native._load()                       # make sure libsvbrdf_hip.so (and torch's HIP runtime) are in the process
    mod.bind(_native.library_path())
    mod.set_second_order_hooks(_loss_second_order, _render_second_order)
    _mod = mod
    return _mod


# what the extension's autograd nodes hand backward(create_graph=True) to (imported at call time: losses imports this module)
def _loss_second_order(*args):
    from . import losses
    return losses.differentiable_loss_backward(*args)


def _render_second_order(maps, scenes, grad_out):
    from . import renderers
    return renderers.differentiable_render_backward(maps, scenes, grad_out)
