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


# ---------------------------------------------------------------------------------------------------------------------
# rank -> CPU placement.  A rank drives its GPU with ~20 us of host work per 40 us loss step and feeds it through
# DataLoader workers; on a two-socket node the GPU hangs off ONE socket's PCIe root, and a rank (or its workers, or its
# pinned staging buffers) scheduled on the other socket pays the inter-socket hop on every launch and every upload.
# Each rank therefore pins itself -- first thing, before any GPU call, no exec and no wrapper process -- to the CPUs of
# its GPU's NUMA node; ranks whose GPUs share a node split that node's cores among themselves (SMT siblings stay
# together), so that their worker pools do not compete.  Everything comes from sysfs; nothing here imports torch or
# touches the GPU runtime.  When the topology cannot be read (no /sys/class/kfd, opaque *_VISIBLE_DEVICES values, a
# device without a NUMA node) the rank is left where it is and says so.
# ---------------------------------------------------------------------------------------------------------------------

def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def _read(path):
    with open(path) as f:
        return f.read()


def _visible_indices(value, n):
    """an *_VISIBLE_DEVICES value as a list of indices into a list of n devices; None = not an index list"""
    if value is None:
        return list(range(n))
    out = []
    for tok in value.split(","):
        tok = tok.strip()
        if tok == "":
            continue
        if not tok.isdigit():
            return None                      # UUID form (GPU-xxxx): not mapped here
        if int(tok) >= n:
            break                            # the runtimes stop at the first invalid index
        out.append(int(tok))
    return out


def gpu_render_minors(sysfs="/sys", environ=None, dev="default"):
    """DRM render minors of the GPUs in the order the HIP runtime numbers them, or None.  KFD topology nodes with
    simd_count > 0 whose render node this process may OPEN -- sysfs lists every GPU of the host even inside a container
    that was given only some /dev/dri/renderD* (docker --device without a device cgroup on sysfs), while ROCr enumerates
    only the accessible ones: counting the others would shift every HIP index to a neighbour's render minor and pin the
    rank to the wrong NUMA node -- in node order = ROCr's agent order; ROCR_VISIBLE_DEVICES then HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES select and reorder by index.  `dev`: the /dev directory to check the render nodes in ("default":
    /dev for the real sysfs, no check for a test tree; None: no check).  Anything unreadable or malformed -> None (the
    rank is left unbound), never an exception."""
    environ = os.environ if environ is None else environ
    if dev == "default":
        dev = "/dev" if sysfs == "/sys" else None
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        names = sorted((n for n in os.listdir(base) if n.isdigit()), key=int)
    except OSError:
        return None
    minors = []
    for n in names:
        try:
            props = dict(line.split(None, 1) for line in _read(os.path.join(base, n, "properties")).splitlines() if " " in line)
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            pci = (int(props.get("domain", "0")), int(props.get("location_id", "0")))
        except ValueError:
            return None                      # a properties file this code does not understand: bind nothing
        if dev is not None and not os.access(os.path.join(dev, "dri", "renderD%d" % minor), os.R_OK | os.W_OK):
            continue                         # not ours to open: the runtime does not enumerate it either
        minors.append(minor)
        _PCI_OF_MINOR[minor] = pci
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES" if "HIP_VISIBLE_DEVICES" in environ else "CUDA_VISIBLE_DEVICES"):
        idx = _visible_indices(environ.get(var), len(minors))
        if idx is None:
            return None
        minors = [minors[i] for i in idx]
    return minors


_PCI_OF_MINOR = {}      # render minor -> (PCI domain, KFD location_id = bus << 8 | devfn), filled by gpu_render_minors


def _numa_node_of_minor(minor, sysfs):
    """NUMA node of a GPU: from its DRM render node, or -- for the render nodes of compute partitions, which are platform
    devices without one -- from the PCI function the KFD topology names (domain, location_id)."""
    paths = [os.path.join(sysfs, "class", "drm", "renderD%d" % minor, "device", "numa_node")]
    if minor in _PCI_OF_MINOR:
        domain, loc = _PCI_OF_MINOR[minor]
        paths.append(os.path.join(sysfs, "bus", "pci", "devices",
                                  "%04x:%02x:%02x.%d" % (domain, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7), "numa_node"))
    for path in paths:
        try:
            node = int(_read(path).strip())
        except (OSError, ValueError):
            continue
        if node >= 0:
            return node
    return None


def pci_address_of_minor(minor):
    """"dddd:bb:dd.f" of a render minor seen by gpu_render_minors (KFD domain / location_id), or None"""
    if minor not in _PCI_OF_MINOR:
        return None
    domain, loc = _PCI_OF_MINOR[minor]
    return "%04x:%02x:%02x.%d" % (domain, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)


def _cores(cpus, sysfs):
    """[[cpu, sibling, ...], ...]: the CPUs grouped by physical core, cores in ascending order"""
    seen, cores = set(), []
    for c in sorted(cpus):
        if c in seen:
            continue
        try:
            sib = [s for s in parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology",
                                                                "thread_siblings_list"))) if s in cpus]
        except (OSError, ValueError):
            sib = [c]
        sib = sorted(set(sib) | {c})
        seen.update(sib)
        cores.append(sib)
    return cores


def rank_cpu_placement(local_rank, local_world, share_device=False, sysfs="/sys", environ=None, allowed=None):
    """Where local rank `local_rank` of `local_world` should run: {"cpus": [...], "numa_node": int or None,
    "render_minor": int or None, "ranks_on_node": int, "source": str}.  `allowed` = the CPUs the process may use
    (default: its current affinity mask)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    unbound = {"cpus": allowed, "numa_node": None, "render_minor": None, "ranks_on_node": local_world}
    minors = gpu_render_minors(sysfs, environ)
    if not minors:
        return dict(unbound, source="unbound: GPU topology not readable from sysfs")
    device_of = (lambda r: 0) if share_device else (lambda r: r)
    if device_of(local_world - 1) >= len(minors) or device_of(local_rank) >= len(minors):
        return dict(unbound, source="unbound: %d local ranks but %d GPU(s) in the topology" % (local_world, len(minors)))
    nodes = [_numa_node_of_minor(minors[device_of(r)], sysfs) for r in range(local_world)]
    mine = nodes[local_rank]
    if mine is None:
        return dict(unbound, render_minor=minors[device_of(local_rank)], source="unbound: the GPU reports no NUMA node")
    try:
        node_cpus = set(parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % mine, "cpulist"))))
    except (OSError, ValueError):
        return dict(unbound, render_minor=minors[device_of(local_rank)], source="unbound: node%d has no cpulist" % mine)
    usable = node_cpus & set(allowed)
    if not usable:
        return dict(unbound, render_minor=minors[device_of(local_rank)],
                    source="unbound: none of node%d's CPUs is in this process's affinity mask" % mine)
    sharers = [r for r in range(local_world) if nodes[r] == mine]          # ranks whose GPU hangs off the same node
    cores = _cores(usable, sysfs)
    k, n = sharers.index(local_rank), len(sharers)
    if len(cores) >= n:
        lo, hi = k * len(cores) // n, (k + 1) * len(cores) // n
        cpus = sorted(c for core in cores[lo:hi] for c in core)
    else:                                                                  # more ranks than cores: share the node
        cpus = sorted(usable)
    return {"cpus": cpus, "numa_node": mine, "render_minor": minors[device_of(local_rank)], "ranks_on_node": n,
            "pci": pci_address_of_minor(minors[device_of(local_rank)]),
            "source": "sysfs: renderD%d -> numa node %d, core slice %d of %d" % (minors[device_of(local_rank)], mine, k + 1, n)}


def crosscheck_placement(placement, device_index, restore_cpus=None):
    """After torch.cuda.set_device: is the GPU the runtime gave this rank the one its CPUs were chosen for?  Compares the
    PCI address recorded from sysfs (placement["pci"]) with torch.cuda.get_device_properties(device_index); on a mismatch
    the binding is undone (affinity back to `restore_cpus`) rather than kept on another GPU's socket.  Records the verdict
    in placement["pci_crosscheck"] ("match", "mismatch: ...", or why it could not be made) and returns the placement."""
    want = placement.get("pci")
    if not placement.get("bound") or want is None:
        placement["pci_crosscheck"] = "not applicable: unbound" if not placement.get("bound") else "not applicable: no PCI address in sysfs"
        return placement
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        got = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception as e:          # a torch build without the PCI fields: nothing to compare
        placement["pci_crosscheck"] = "not checked: %r" % (e,)
        return placement
    if want.rsplit(".", 1)[0] == got:
        placement["pci_crosscheck"] = "match"
        return placement
    placement["pci_crosscheck"] = "mismatch: sysfs says %s, the runtime's device %d is %s" % (want, device_index, got)
    if restore_cpus:
        try:
            os.sched_setaffinity(0, restore_cpus)
            placement.update(bound=False, cpus=format_cpulist(sorted(restore_cpus)), n_cpus=len(restore_cpus), numa_node=None,
                             source=placement["source"] + " (undone: PCI mismatch)")
            placement["pci_crosscheck"] += " -- binding undone"
        except OSError as e:
            placement["pci_crosscheck"] += " -- binding kept (sched_setaffinity failed: %s)" % e
    else:
        placement["pci_crosscheck"] += " -- binding kept (no CPU set to restore)"
    return placement


def bind_rank_to_gpu_numa(local_rank, local_world, share_device=False):
    """Pin the calling process (call it first thing in a rank, before any GPU call) as rank_cpu_placement says; returns
    that record with "cpus" as a compact range string and "bound": whether the affinity mask was changed.
    SVBRDF_NO_CPU_BINDING=1 leaves the process alone."""
    if os.environ.get("SVBRDF_NO_CPU_BINDING"):
        place = {"cpus": sorted(os.sched_getaffinity(0)), "numa_node": None, "render_minor": None,
                 "ranks_on_node": local_world, "source": "unbound: SVBRDF_NO_CPU_BINDING is set"}
    else:
        place = rank_cpu_placement(local_rank, local_world, share_device)
    bound = False
    if place["numa_node"] is not None:
        try:
            os.sched_setaffinity(0, place["cpus"])
            bound = True
        except OSError as e:
            place["source"] += " (sched_setaffinity failed: %s)" % e
    place["n_cpus"] = len(place["cpus"])
    place["cpus"] = format_cpulist(place["cpus"])
    place["bound"] = bound
    return place
