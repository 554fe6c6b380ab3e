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


def bringup_diagnosis(backend, device, phase, waited_s):
    """what a maintainer needs to see when the process group does not come up (one line per fact, for stderr)"""
    import os
    env = os.environ
    try:
        n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU runtime on this image
    except Exception as e:  # pragma: no cover
        n_dev = "? (%r)" % (e,)
    return "\n".join([
        "[bring-up] rank %s of %s: process group did not come up -- stuck in %s for %.0f s" % (
            env.get("RANK", "?"), env.get("WORLD_SIZE", "?"), phase, waited_s),
        "[bring-up]   backend                      %s%s" % (backend, " (= RCCL on ROCm)" if backend == "nccl" else ""),
        "[bring-up]   this rank's device           %s   (LOCAL_RANK=%s, LOCAL_WORLD_SIZE=%s)" % (
            device, env.get("LOCAL_RANK", "?"), env.get("LOCAL_WORLD_SIZE", "?")),
        "[bring-up]   devices visible to torch     %s   (ROCR_VISIBLE_DEVICES=%s HIP_VISIBLE_DEVICES=%s CUDA_VISIBLE_DEVICES=%s)" % (
            n_dev, env.get("ROCR_VISIBLE_DEVICES"), env.get("HIP_VISIBLE_DEVICES"), env.get("CUDA_VISIBLE_DEVICES")),
        "[bring-up]   HSA_ENABLE_IPC_MODE_LEGACY   %s   (hosts whose driver only does dmabuf IPC need 0: RCCL otherwise fails "
        "with hipIpcGetMemHandle: invalid argument, or hangs)" % env.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        "[bring-up]   rendezvous                   %s:%s" % (env.get("MASTER_ADDR"), env.get("MASTER_PORT")),
        "[bring-up]   next steps: every rank started? (one rank per GPU, WORLD_SIZE of them); NCCL_DEBUG=INFO for RCCL's own "
        "account; --backend gloo separates a rendezvous problem from an RCCL one",
    ])


class _BringupWatchdog:
    """Exits the process with BRINGUP_EXIT_CODE and the diagnosis if not cancelled within `timeout_s`.  A Python thread
    (needs the GIL for a moment: torch's blocking calls release it), backed by faulthandler's C-level timer 15 s later,
    which needs no GIL, dumps every thread's stack -- showing WHERE it hangs -- and exits."""

    def __init__(self, timeout_s, backend, device):
        import threading
        self.timeout_s, self.backend, self.device, self.phase = float(timeout_s), backend, device, "start"
        self._cancel = threading.Event()
        self._thread = threading.Thread(target=self._run, name="svbrdf-bringup-watchdog", daemon=True)

    def start(self):
        import faulthandler
        import sys
        self._thread.start()
        try:
            faulthandler.dump_traceback_later(self.timeout_s + 15.0, exit=True, file=sys.stderr)
        except Exception:  # pragma: no cover  (stderr without a file descriptor)
            pass
        return self

    def _run(self):
        import os
        import sys
        if self._cancel.wait(self.timeout_s):
            return
        try:
            sys.stderr.write(bringup_diagnosis(self.backend, self.device, self.phase, self.timeout_s) + "\n")
            sys.stderr.flush()
        finally:
            os._exit(BRINGUP_EXIT_CODE)

    def cancel(self):
        import faulthandler
        self._cancel.set()
        try:
            faulthandler.cancel_dump_traceback_later()
        except Exception:  # pragma: no cover
            pass


def init_process_group_checked(backend, device=None, timeout_s=60.0, exit_on_failure=True):
    """``dist.init_process_group`` + ONE one-element all-reduce (the first collective is where RCCL builds its
    communicator, opens its IPC handles and its xGMI rings), under a watchdog.  Returns ``ranks_seen`` = the sum of ones
    over the group -- counted by the collective itself, not read from the environment.

    Failure modes.  (a) Nothing completes within `timeout_s` seconds (a peer that never arrives, RCCL wedged in its
    set-up): the thread that called this is blocked inside torch and cannot be interrupted, so the watchdog prints the
    diagnosis and ends the PROCESS with BRINGUP_EXIT_CODE -- always; launch.spawn_ranks / torchrun then stop the other
    ranks.  (b) Something raises, or the group is not the size the launcher said: with `exit_on_failure` (what bench.py
    and train.py use: a rank's half-built communicator must not get a chance to hang in a destructor) the diagnosis is
    printed and the process exits with BRINGUP_EXIT_CODE; otherwise ``ProcessGroupBringupError`` carries the diagnosis to
    the caller.  `device`: this rank's torch.device for backend "nccl" (passed as device_id: eager communicator on that
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
        failed(bringup_diagnosis(backend, device, wd.phase + " -- raised %r" % (e,), 0.0), e)
    wd.cancel()
    if seen != dist.get_world_size():
        failed(bringup_diagnosis(backend, device, "the first collective -- it summed %d ones over a group of %d"
                                 % (seen, dist.get_world_size()), 0.0))
    return seen
