#!/usr/bin/env python3
"""A/B of the copy kernel behind ``svbrdf_debug_copy`` (the measured-copy peak of SURVEY 8d): unroll 1/2/4/8 x non-temporal
or plain accesses x buffer size, each in a child process (the variant is read from the environment when the kernel is
launched), next to hipMemcpyAsync device-to-device (``Tensor.copy_``).  Prints one JSON line per case.
    python tools/copy_peak.py            # the grid
    python tools/copy_peak.py --one      # the shipped variant only"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(gib, reps=20):
    import torch
    from svbrdf_estimation_amd import _native
    dev = torch.device("cuda:0")
    n = int(gib * 2 ** 30) // 4
    src = torch.empty(n, device=dev).uniform_(-1, 1)
    dst = torch.empty_like(src)
    res = {}
    for name, fn in (("kernel", lambda: _native.debug_copy(dst, src)), ("memcpy_d2d", lambda: dst.copy_(src))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            best = us if best is None else min(best, us)
        res[name] = {"us": best, "GBps": 8.0 * n / (best * 1e-6) / 1e9}
    assert torch.equal(src, dst)
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(json.dumps(measure(float(sys.argv[2]))), flush=True)
        sys.exit(0)
    grid = [(1, 1, 1.0)] if "--one" in sys.argv else [(u, nt, g) for g in (1.0, 4.0) for nt in (1, 0) for u in (1, 2, 4, 8)]
    for unroll, nt, gib in grid:
        env = dict(os.environ, SVBRDF_COPY_UNROLL=str(unroll), SVBRDF_COPY_NT=str(nt))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(gib)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"unroll": unroll, "nontemporal": nt, "GiB_each_way": gib,
                          "result": json.loads(line[0]) if line else r.stderr[-300:]}), flush=True)
