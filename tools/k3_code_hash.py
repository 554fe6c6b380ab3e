#!/usr/bin/env python3
"""sha256 of a kernel's machine code inside a built libsvbrdf_hip.so (svbrdf_estimation_amd/_codehash.py): what
profiles/k3_hbm_traffic.json is keyed to.  No GPU needed.
    python tools/k3_code_hash.py [library.so] [name substring ...]      default: the shipped library, bench.py's headline kernel
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("_codehash", os.path.join(ROOT, "svbrdf_estimation_amd", "_codehash.py"))
_codehash = importlib.util.module_from_spec(spec)
spec.loader.exec_module(_codehash)           # (without importing the package: no torch needed)

if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "svbrdf_estimation_amd", "lib", "libsvbrdf_hip.so")
    names = tuple(sys.argv[2:]) or _codehash.K3_HEADLINE
    print(json.dumps(_codehash.kernel_code_sha256(so, names)))
