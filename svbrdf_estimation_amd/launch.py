"""Self-launch of the one-process-per-GPU layout: ``script --gpus N`` started as ONE plain process
(no torchrun) becomes N fresh rank processes.

The reference is single-device (development/multiImage_pytorch/main.py:33-36); the multi-GPU layout
of this engine is one process per GPU, batch sharded by rank (distributed.py).  ``bench.py`` and
``train.py`` accept being started either way:

  * under ``python -m torch.distributed.run --nproc-per-node N ...``: RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* are in the environment, the script is a rank and just runs;
  * as ``python script.py --gpus N`` with no rank environment: the script calls ``spawn_ranks`` FIRST,
    before anything initialises the GPU runtime.  The parent never touches the GPU, never ``exec``s and
    never re-launches itself after a GPU call -- it only starts N children with ``subprocess`` (each a
    fresh interpreter with its own rank environment), relays rank 0's stdout, and exits non-zero if
    any child does.

Nothing here imports torch.
"""
import os
import socket
import subprocess
import sys
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def launched_as_rank(environ=None):
    """True when a launcher (torchrun or spawn_ranks) already gave this process a rank."""
    environ = os.environ if environ is None else environ
    return "WORLD_SIZE" in environ and "RANK" in environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environment(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # the host driver only supports dmabuf IPC: RCCL / cross-process device memory needs this
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(script, argv, world, timeout_s=None, poll_s=0.05, grace_s=10.0):
    """Start `world` rank processes of ``python script argv...``; relay rank 0's stdout to ours (the
    other ranks' stdout goes to stderr), wait for all of them and return the largest exit code.
    When one rank fails the others are terminated (exact PIDs) instead of being left in a barrier; a rank
    that has not exited `grace_s` seconds after its SIGTERM (wedged in a collective, say) is killed."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()       # closed again before the ranks bind it: another process could take it in between, in which
    procs = []               # case rank 0's rendezvous fails loudly and the parent exits non-zero (no silent retry)
    for rank in range(world):
        out = None if rank == 0 else sys.stderr     # rank 0 inherits our stdout: its JSON line is ours
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=rank_environment(rank, world, port),
                                      stdout=out, stderr=None))
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    kill_at = None                                   # set when the ranks have been sent SIGTERM
    worst = 0
    try:
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and kill_at is None:
                    worst = max(worst, rc if rc > 0 else 1)
                    print("[launch] rank %d exited with code %d; stopping the other ranks" % (r, rc),
                          file=sys.stderr, flush=True)
                    for o in sorted(pending):
                        procs[o].terminate()
                    kill_at = time.monotonic() + grace_s
            if deadline is not None and time.monotonic() > deadline and pending:
                print("[launch] timeout after %.0f s; stopping ranks %s" % (timeout_s, sorted(pending)),
                      file=sys.stderr, flush=True)
                for o in sorted(pending):
                    procs[o].terminate()
                worst = max(worst, 124)
                deadline = None
                kill_at = time.monotonic() + grace_s
            if kill_at is not None and time.monotonic() > kill_at and pending:
                print("[launch] ranks %s ignored SIGTERM for %.0f s; killing them" % (sorted(pending), grace_s),
                      file=sys.stderr, flush=True)
                for o in sorted(pending):
                    procs[o].kill()
                worst = max(worst, 1)
                kill_at = None
            if pending:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return worst
