#!/usr/bin/env python3
"""Shader clock over time, from idle into a run of fused-loss launches and out again (round 5).

Question: bench.py (rounds 1-4) read the clock "under load" with ONE 3 ms probe launched on an idle GPU just before the
step loop started; tests/test_gpu_perf_guard.py launches its probe in the middle of a deep queue of launches.  The two
disagree by 10-13 % on one box.  This prints a time series: back-to-back 0.2 ms probes (svbrdf_debug_clock_probe: one
wave, s_memtime / s_memrealtime) on a stream of their own, while another stream idles for ~2 ms, runs N launches of K3
(config 2) and idles again.  Columns: probe start (ms, 100 MHz counter), clock GHz, and whether K3 was running.

    python3 tools/clock_timeline.py [launches=150]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_gpu_perf_guard import Harness  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    dev = torch.device("cuda:0")
    h = Harness(dev)
    enqueue, keep, _ = h.k3_calls(8, 256, 3, 6)          # enqueue(first, n): n launches from C (tests/c_host/launch_loop.c)
    enqueue(0, 40)
    torch.cuda.synchronize(dev)
    for mode in ("idle_start", "busy_start", "sustained"):
        n_probe = 60
        outs = torch.zeros(n_probe, 2, dtype=torch.int64, device=dev)
        if mode == "busy_start":            # the GPU has been under this load for 0.3 s when the series starts
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                enqueue(0, 32)
                torch.cuda.synchronize(dev)
        elif mode == "sustained":           # 0.6 s of launches with NO synchronisation in between (a training loop's load)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.6:
                enqueue(0, 32)
        else:
            time.sleep(0.5)
        e_start, e_k0, e_k1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e_start.record(h.sb)
        for i in range(n_probe):            # 60 x 0.2 ms = 12 ms of clock readings
            assert h.probe(outs[i].data_ptr(), 20000, ctypes.c_void_p(h.sb.cuda_stream)) == 0
        if mode == "idle_start":
            time.sleep(0.002)
        e_k0.record(h.sa)
        enqueue(0, n)
        e_k1.record(h.sa)
        torch.cuda.synchronize(dev)
        k0, k1 = e_start.elapsed_time(e_k0), e_start.elapsed_time(e_k1)
        print("# %s: K3 x %d ran from %.2f to %.2f ms after the first probe (%.2f us per launch)" % (mode, n, k0, k1, 1e3 * (k1 - k0) / n))
        t = 0.0
        for i, (cyc, ticks) in enumerate(outs.tolist()):
            ms = ticks * 1e-5
            print("%s probe %2d  t=%6.2f ms  %.3f GHz  %s" % (mode, i, t, cyc / ticks * 0.1, "K3" if (t + ms > k0 and t < k1) else ""))
            t += ms + 0.004


if __name__ == "__main__":
    main()
