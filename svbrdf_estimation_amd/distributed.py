"""Multi-GPU sharding of the rendering loss: one process per GPU, batch split by rank.

The hot path has NO exchange step: every (batch item, scene, pixel) is independent and the
loss of a batch is the mean of its items' losses, so each rank computes the loss and
d loss/d input of ITS shard with the same kernels and no data-path collective.  With equal
shards, DDP's gradient averaging of the U-Net (stock PyTorch-ROCm, RCCL over xGMI -- not
part of this package) turns the per-rank means into the global-batch mean.  The only
collective offered here is the one-scalar all-reduce for REPORTING a global loss.

Per-rank scene RNG: the reference samples scenes from the global CPU generator
(losses.py:35); ranks must not draw identical scenes, hence ``rank_seed``.
"""
import torch


def rank_seed(base_seed, rank):
    return int(base_seed) + int(rank)


def shard_bounds(global_batch, rank, world_size):
    """contiguous, equal shards; the global batch must divide evenly (as DistributedSampler pads to)."""
    if global_batch % world_size != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def shard(tensor, rank, world_size):
    lo, hi = shard_bounds(tensor.shape[0], rank, world_size)
    return tensor[lo:hi]


class ContiguousShardSampler:
    """Sampler for verification runs: global batch k is the index block [k*G, (k+1)*G), G = per_rank_batch * world,
    and rank r reads its contiguous slice of it, unshuffled -- so that N ranks together see exactly the batches one
    process with batch G sees, in the same item order (DistributedSampler interleaves the ranks' items)."""

    def __init__(self, length, per_rank_batch, rank, world_size):
        self.per, self.rank, self.world = int(per_rank_batch), int(rank), int(world_size)
        self.batches = int(length) // (self.per * self.world)

    def __len__(self):
        return self.batches * self.per

    def __iter__(self):
        G = self.per * self.world
        for k in range(self.batches):
            lo = k * G + self.rank * self.per
            yield from range(lo, lo + self.per)


def global_mean(local_mean):
    """mean over ranks of a per-rank mean (equal shards) -- for logging only."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_mean
    t = local_mean.detach().clone().reshape(1)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).reshape(())


# ---------------------------------------------------------------------------------------------------------------------
# bring-up of the process group, fail-fast.  Until an 8-GPU node has run this code, RCCL with more than one device has
# executed nowhere: if the first real multi-GPU run cannot rendezvous, cannot build its
# communicator or hangs in its first collective, it must cost one minute and leave a diagnosis -- not sit in
# init_process_group's 30-minute default until the driver's limit kills it.
# ---------------------------------------------------------------------------------------------------------------------
BRINGUP_EXIT_CODE = 3


class ProcessGroupBringupError(RuntimeError):
    """the process group did not come up (rendezvous, communicator set-up or the first collective raised, or the group is
    not the size the launcher said); the message is the multi-line diagnosis of ``bringup_diagnosis``"""


def bringup_facts(device):
    """the facts of the diagnosis that need a call into torch, collected BEFORE anything can hang: the watchdog thread must
    not call into the GPU runtime while the main thread may be stuck inside it holding its lock"""
    try:
        n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU runtime on this image
    except Exception as e:  # pragma: no cover
        n_dev = "? (%r)" % (e,)
    return {"n_dev": n_dev, "device": str(device)}


def bringup_diagnosis(backend, device, phase, waited_s, facts=None):
    """what a maintainer needs to see when the process group does not come up (one line per fact, for stderr)"""
    import os
    env = os.environ
    facts = facts if facts is not None else bringup_facts(device)
    return "\n".join([
        "[bring-up] rank %s of %s: process group did not come up -- stuck in %s for %.0f s" % (
            env.get("RANK", "?"), env.get("WORLD_SIZE", "?"), phase, waited_s),
        "[bring-up]   backend                      %s%s" % (backend, " (= RCCL on ROCm)" if backend == "nccl" else ""),
        "[bring-up]   this rank's device           %s   (LOCAL_RANK=%s, LOCAL_WORLD_SIZE=%s)" % (
            facts["device"], env.get("LOCAL_RANK", "?"), env.get("LOCAL_WORLD_SIZE", "?")),
        "[bring-up]   devices visible to torch     %s   (ROCR_VISIBLE_DEVICES=%s HIP_VISIBLE_DEVICES=%s CUDA_VISIBLE_DEVICES=%s)" % (
            facts["n_dev"], env.get("ROCR_VISIBLE_DEVICES"), env.get("HIP_VISIBLE_DEVICES"), env.get("CUDA_VISIBLE_DEVICES")),
        "[bring-up]   HSA_ENABLE_IPC_MODE_LEGACY   %s   (hosts whose driver only does dmabuf IPC need 0: RCCL otherwise fails "
        "with hipIpcGetMemHandle: invalid argument, or hangs)" % env.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        "[bring-up]   rendezvous                   %s:%s" % (env.get("MASTER_ADDR"), env.get("MASTER_PORT")),
        "[bring-up]   next steps: every rank started? (one rank per GPU, WORLD_SIZE of them); NCCL_DEBUG=INFO for RCCL's own "
        "account; --backend gloo separates a rendezvous problem from an RCCL one; a multi-node job or a cold start that "
        "legitimately needs longer: raise --bringup-timeout (the limit also covers the rendezvous)",
    ])


class _BringupWatchdog:
    """Exits the process with BRINGUP_EXIT_CODE and the diagnosis if not cancelled within `timeout_s`.  Two plain Python
    threads (torch's blocking calls release the GIL): the first prints the diagnosis -- every fact that needs torch was
    collected before the clock started, so it touches nothing the hung main thread may hold -- and every thread's stack
    (WHERE it hangs), then exits; the second, `GRACE_S` later, only calls ``os._exit``: the exit code is BRINGUP_EXIT_CODE
    even if writing the diagnosis itself gets stuck.  faulthandler's process-wide ``dump_traceback_later`` timer is NOT used:
    it has one owner per process (pytest's faulthandler_timeout, an application's own hang dump), and it is not ours."""
    GRACE_S = 5.0

    def __init__(self, timeout_s, backend, device):
        import threading
        self.timeout_s, self.backend, self.device, self.phase = float(timeout_s), backend, device, "start"
        self.facts = bringup_facts(device)
        self._cancel = threading.Event()
        self._thread = threading.Thread(target=self._run, name="svbrdf-bringup-watchdog", daemon=True)
        self._backstop = threading.Thread(target=self._exit_only, name="svbrdf-bringup-backstop", daemon=True)

    def start(self):
        self._thread.start()
        self._backstop.start()
        return self

    def _run(self):
        import os
        import sys
        if self._cancel.wait(self.timeout_s):
            return
        try:
            sys.stderr.write(bringup_diagnosis(self.backend, self.device, self.phase, self.timeout_s, self.facts) + "\n")
            sys.stderr.flush()
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)      # a direct dump: no timer is armed or cleared
            sys.stderr.flush()
        finally:
            os._exit(BRINGUP_EXIT_CODE)

    def _exit_only(self):
        import os
        if not self._cancel.wait(self.timeout_s + self.GRACE_S):
            os._exit(BRINGUP_EXIT_CODE)

    def cancel(self):
        self._cancel.set()


def init_process_group_checked(backend, device=None, timeout_s=60.0, exit_on_failure=True):
    """``dist.init_process_group`` + ONE one-element all-reduce (the first collective is where RCCL builds its
    communicator, opens its IPC handles and its xGMI rings), under a watchdog.  Returns ``ranks_seen`` = the sum of ones
    over the group -- counted by the collective itself, not read from the environment.

    `timeout_s` covers the rendezvous too: a multi-node job, or a cold start in which ranks page in their images at very
    different speeds, legitimately needs more than the single-node default -- raise it (--bringup-timeout) there.

    Failure modes.  (a) Nothing completes within `timeout_s` seconds (a peer that never arrives, RCCL wedged in its
    set-up): the thread that called this is blocked inside torch and cannot be interrupted, so the watchdog prints the
    diagnosis and ends the PROCESS with BRINGUP_EXIT_CODE -- always; launch.spawn_ranks / torchrun then stop the other
    ranks.  (b) Something raises, or the group is not the size the launcher said: with `exit_on_failure` (what bench.py
    and train.py use: a rank's half-built communicator must not get a chance to hang in a destructor) the diagnosis is
    printed and the process exits with BRINGUP_EXIT_CODE; otherwise the default group, if it got as far as existing, is
    destroyed again and ``ProcessGroupBringupError`` carries the diagnosis to the caller, who may retry.  `device`: this rank's torch.device for backend "nccl" (passed as device_id: eager communicator on that
    device), ignored for gloo."""
    import os
    import sys
    import torch.distributed as dist
    nccl = backend == "nccl"

    def failed(text, cause=None):
        if exit_on_failure:
            if cause is not None:
                import traceback
                traceback.print_exception(type(cause), cause, cause.__traceback__)
            sys.stderr.write(text + "\n")
            sys.stderr.flush()
            os._exit(BRINGUP_EXIT_CODE)         # not sys.exit: no destructor of a half-built communicator gets to hang
        # a library caller that catches this must be able to try again: leave no half-built default group behind (the next
        # init_process_group would fail with "initialised twice", and a dead communicator would stay alive)
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # pragma: no cover  (tearing down what never fully came up)
            pass
        raise ProcessGroupBringupError(text) from cause

    wd = _BringupWatchdog(timeout_s, backend, device).start()
    try:
        wd.phase = "the rendezvous / communicator set-up (init_process_group)"
        dist.init_process_group(backend=backend, **({"device_id": device} if nccl else {}))
        wd.phase = "the first collective (all-reduce of one element)"
        one = torch.ones(1, dtype=torch.float32, device=device if nccl else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        seen = int(round(float(one.item())))            # .item(): the reduction has really finished
    except (KeyboardInterrupt, SystemExit):
        wd.cancel()
        raise
    except BaseException as e:
        wd.cancel()
        failed(bringup_diagnosis(backend, device, wd.phase + " -- raised %r" % (e,), 0.0, wd.facts), e)
    wd.cancel()
    if seen != dist.get_world_size():
        failed(bringup_diagnosis(backend, device, "the first collective -- it summed %d ones over a group of %d"
                                 % (seen, dist.get_world_size()), 0.0, wd.facts))
    return seen


# ---------------------------------------------------------------------------------------------------------------------
# first contact with a multi-GPU node (``bench.py --gpus N --selftest``, tools/scale_first_contact.md): after the bring-up
# above, three questions whose answers the first 8-GPU run should not have to discover one failure at a time --
#   (a) does every rank really sit on its own GPU?            all-gather of PCI bus ids, N distinct expected
#   (b) does the all-reduce DDP will issue move data at a sane rate?   8 MB and 320 MB (the U-Net's fp32 gradients)
#   (c) does the fused loss compute the reference's numbers on THIS device?   a committed tiny fixture of the reference
# Everything runs under one deadline with the bring-up's discipline: on a hang or a failed check the rank prints what it
# knows and the process exits with BRINGUP_EXIT_CODE.
# ---------------------------------------------------------------------------------------------------------------------
def pci_bus_id(device):
    """'0000:c1:00' of a torch device from the runtime's device properties (what HIP and RCCL see; the same fields
    launch.crosscheck_placement compares with sysfs); the device name and index where a build does not expose them"""
    p = torch.cuda.get_device_properties(device)
    try:
        return "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except AttributeError:  # pragma: no cover
        return "%s#%s" % (p.name, device.index)


def allreduce_rate(nbytes, device, nccl, reps=5):
    """bus bandwidth of an fp32 SUM all-reduce of `nbytes` in GB/s as the ring formula counts it, 2 (N-1)/N x bytes / time
    (world size 1: bytes / time of the call, which moves nothing -- reported, not judged); best of `reps` after one warm-up"""
    import time
    import torch.distributed as dist
    n = dist.get_world_size()
    buf = torch.ones(max(1, nbytes // 4), dtype=torch.float32, device=device if nccl else "cpu")
    best = None
    for r in range(reps + 1):
        buf.fill_(1.0)
        if nccl:
            torch.cuda.synchronize(device)
        dist.barrier(**({"device_ids": [device.index]} if nccl else {}))
        t0 = time.perf_counter()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if nccl:
            torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        if r > 0:
            best = dt if best is None else min(best, dt)
    ok = bool(buf[0].item() == float(n) and buf[-1].item() == float(n))
    factor = 2.0 * (n - 1) / n if n > 1 else 1.0
    return {"bytes": int(buf.numel() * 4), "seconds": best, "GBps": factor * buf.numel() * 4 / best / 1e9, "sum_correct": ok}


def fixture_parity(device, fixture_path):
    """the fused loss on `device` against a committed fixture of the reference (tests/golden/g3_*.npz: input, target,
    captured scene table, the reference's loss and gradient): loss within 2e-6 relative, gradient within
    1e-4 |ref| + 1e-5 max|ref| at every element but the few tie / highlight pixels the GPU suite ledgers (<= 0.1 %)"""
    import numpy as np
    from . import _native
    g = np.load(fixture_path)
    inp = torch.from_numpy(np.ascontiguousarray(g["input"], dtype=np.float32)).to(device)
    tgt = torch.from_numpy(np.ascontiguousarray(g["target"], dtype=np.float32)).to(device)
    table = torch.from_numpy(np.ascontiguousarray(g["scenes"], dtype=np.float32))
    loss, grad = _native.rendering_loss(inp, tgt, table, 0.1, want_grad=True)
    torch.cuda.synchronize(device)
    ref_loss, ref_grad = float(g["loss"]), np.asarray(g["grad_input"], dtype=np.float32)
    err = np.abs(grad.cpu().numpy() - ref_grad)
    outside = int((err > 1e-4 * np.abs(ref_grad) + 1e-5 * np.abs(ref_grad).max()).sum())
    rel = abs(float(loss.item()) - ref_loss) / abs(ref_loss)
    return {"loss": float(loss.item()), "reference_loss": ref_loss, "loss_rel_err": rel, "grad_elements": int(err.size),
            "grad_outside_tolerance": outside, "ok": bool(rel <= 2e-6 and outside <= max(8, err.size // 1000))}


def first_contact_selftest(device, nccl, share_device, fixture_path, timeout_s=60.0):
    """-> dict for the JSON line (ranks_seen, distinct_devices, allreduce_GBps per rank, parity per rank, ok); exits the
    process with BRINGUP_EXIT_CODE (diagnosis on stderr) when a check fails or the whole thing exceeds `timeout_s`.
    `device` None (bench.py --plumbing-only, the CPU suite): the collective part only -- gathers and all-reduce rates over
    gloo -- with the device checks (a) and (c) reported as skipped."""
    import os
    import sys
    import torch.distributed as dist
    wd = _BringupWatchdog(timeout_s, "nccl" if nccl else "gloo", device).start()
    world, rank = dist.get_world_size(), dist.get_rank()

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    try:
        wd.phase = "selftest (a): all-gather of PCI bus ids"
        ids = gather(pci_bus_id(device) if device is not None else "no device (plumbing only), rank %d" % rank)
        wd.phase = "selftest (b): 8 MB all-reduce"
        small = allreduce_rate(8 << 20, device, nccl)
        wd.phase = "selftest (b): 320 MB all-reduce (the U-Net's fp32 gradients)"
        large = allreduce_rate(320 << 20, device, nccl, reps=3)
        wd.phase = "selftest (c): fused loss against the committed reference fixture"
        parity = fixture_parity(device, fixture_path) if device is not None else {"ok": True, "skipped": "no device (plumbing only)"}
        wd.phase = "selftest: gathering the ranks' results"
        every = gather({"allreduce_8MB": small, "allreduce_320MB": large, "parity": parity})
    except BaseException as e:
        wd.cancel()
        import traceback
        traceback.print_exception(type(e), e, e.__traceback__)
        sys.stderr.write(bringup_diagnosis("nccl" if nccl else "gloo", device, wd.phase + " -- raised %r" % (e,), 0.0, wd.facts) + "\n")
        sys.stderr.flush()
        os._exit(BRINGUP_EXIT_CODE)
    wd.cancel()
    distinct = len(set(ids))
    want_distinct = 1 if (share_device and device is not None) else world
    res = {"ranks_seen": world, "pci_bus_ids": ids, "distinct_devices": distinct, "distinct_devices_expected": want_distinct,
           "allreduce_GBps": {"8MB": [r["allreduce_8MB"]["GBps"] for r in every],
                              "320MB": [r["allreduce_320MB"]["GBps"] for r in every]},
           "allreduce_seconds": {"8MB": [r["allreduce_8MB"]["seconds"] for r in every],
                                 "320MB": [r["allreduce_320MB"]["seconds"] for r in every]},
           "allreduce_sums_correct": all(r["allreduce_8MB"]["sum_correct"] and r["allreduce_320MB"]["sum_correct"] for r in every),
           "allreduce_formula": "2 (N-1)/N x bytes / time (ring bus bandwidth); N = 1: bytes / time of a call that moves nothing",
           "parity": [r["parity"] for r in every],
           "transport": "RCCL" if nccl else "gloo (CPU transport: plumbing only, rates are host memory's)"}
    res["ok"] = bool(distinct == want_distinct and res["allreduce_sums_correct"] and all(p["ok"] for p in res["parity"]))
    if not res["ok"]:
        if rank == 0:
            import json
            sys.stderr.write("[selftest] FAILED: %s\n" % json.dumps(res))
            if distinct != want_distinct:
                sys.stderr.write("[selftest]   %d ranks sit on %d distinct GPUs (%s): check LOCAL_RANK -> device mapping and "
                                 "ROCR/HIP_VISIBLE_DEVICES\n" % (world, distinct, ids))
            sys.stderr.flush()
        try:        # every rank has printed what it knows before any of them goes (rank 0's stderr is the one that is read)
            dist.barrier(**({"device_ids": [device.index]} if nccl else {}))
        finally:
            os._exit(BRINGUP_EXIT_CODE)
    return res


def require_distinct_devices(device, share_device, rank):
    """After the bring-up, before anything is timed or trained: every rank's GPU as the runtime reports it (PCI address),
    gathered; `world` distinct ones or the job stops with exit code 3 and the reason -- two ranks on one device is a launcher
    mistake (LOCAL_RANK -> device mapping, ROCR/HIP_VISIBLE_DEVICES) that would otherwise only show as a bad scaling number.
    `share_device`: the plumbing tests' announced exception.  -> the list of addresses (rank order)."""
    import sys
    import torch.distributed as dist
    pcis = [None] * dist.get_world_size()
    dist.all_gather_object(pcis, pci_bus_id(device))
    if not share_device and len(set(pcis)) != dist.get_world_size():
        if rank == 0:
            sys.stderr.write("[bring-up] %d ranks sit on %d distinct GPUs (%s): LOCAL_RANK -> device mapping or ROCR/HIP_VISIBLE_DEVICES "
                             "is wrong (one rank per GPU; --share-device is for plumbing tests only)\n" % (len(pcis), len(set(pcis)), pcis))
            sys.stderr.flush()
        dist.destroy_process_group()
        sys.exit(BRINGUP_EXIT_CODE)
    return pcis
