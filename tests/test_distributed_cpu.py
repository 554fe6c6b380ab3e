"""CPU, world_size 2 over gloo: the multi-GPU story of the path.  The loss shards by batch with
no data-path collective; the only collective is the scalar all-reduce used for reporting.
The per-rank loss values come from the C oracle here (no GPU in this container) -- the
sharding logic under test (shard bounds, per-rank scene RNG, mean of shard means) is the
product's svbrdf_estimation_amd.distributed + RenderingLoss.sample_scene_table."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import c_oracle
    from svbrdf_estimation_amd import distributed as D
    from svbrdf_estimation_amd import losses, renderers

    c_oracle.set_threads(2)
    G, H = 4, 16
    inp, tgt = synth.make_maps(301, G, H), synth.make_maps(302, G, H)
    lo, hi = D.shard_bounds(G, rank, world)
    fn = losses.RenderingLoss(renderers.LocalRenderer())
    torch.manual_seed(D.rank_seed(313, rank))                 # per-rank scene RNG
    table = fn.sample_scene_table(hi - lo).numpy()
    local, grad = c_oracle.rendering_loss(inp[lo:hi], tgt[lo:hi], table)
    glob = D.global_mean(torch.tensor(local, dtype=torch.float64))
    # gather every rank's table/loss on rank 0 to check against the unsharded computation
    tables = [None] * world
    dist.all_gather_object(tables, (table, local, grad))
    if rank == 0:
        full_table = np.concatenate([t[0] for t in tables], axis=0)
        full_loss, full_grad = c_oracle.rendering_loss(inp, tgt, full_table)
        shard_grads = np.concatenate([t[2] for t in tables], axis=0) / world   # what DDP's averaging yields
        np.save(out_path, np.array([glob.item(), full_loss, np.abs(shard_grads - full_grad).max(),
                                    np.abs(full_grad).max(),
                                    float(np.array_equal(tables[0][0], tables[1][0]))]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded(tmp_path):
    port, out = _free_port(), str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    glob, full, gerr, gmax, same_tables = np.load(out)
    assert abs(glob - full) <= 1e-12 * abs(full)          # mean of equal shard means == global mean
    assert gerr <= 1e-7 * gmax                            # averaged shard gradients == global-batch gradient
    assert same_tables == 0.0                             # ranks drew different scenes


def _train_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(3)
    import train
    args = train.parse_args(["--device", "cpu", "--loss", "l1", "--steps", "2", "--warmup", "0", "--batch", "1",
                             "--workers", "0", "--lr", "1e-4", "--save", os.path.join(out_dir, "ckpt.tar")])
    res = train.run(args)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([res["loss_first_quarter"], res["loss_last_quarter"]]))


def test_ddp_training_harness_two_ranks_cpu(tmp_path):
    """row f4: train.py under gloo, world size 2 -- DDP wiring, DistributedSampler shards, per-rank seeds,
    global loss reporting, checkpoint.  (The rendering loss has no CPU path, so the plumbing run uses the
    stock SVBRDFL1Loss; the GPU test runs the fused losses.)"""
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(str(tmp_path / "r0.npy")), np.load(str(tmp_path / "r1.npy"))
    assert np.allclose(a, b) and np.isfinite(a).all()          # the reported loss is the global mean on every rank
    ck = torch.load(str(tmp_path / "ckpt.tar"), map_location="cpu", weights_only=False)
    assert ck["model_type"] == "single" and len(ck["model_state_dict"]) == 98 and ck["epoch"] == 0
    # the reference's parameter names (persistence.py / models.py), so its Checkpoint.restore_model_state can load it
    assert "generator.enc1.conv.conv.weight" in ck["model_state_dict"] and "optimizer_state_dict" not in ck
    from svbrdf_estimation_amd.training import models
    net = models.SingleViewModel()
    net.load_state_dict(models.convert_reference_state_dict(ck["model_state_dict"]))


# ---------------------------------------------------------------- row e: `--gpus N` starts its own N ranks
def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _json_lines(text):
    import json
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            out.append(json.loads(line))
    return out


def test_bench_gpus_n_spawns_n_ranks_without_a_launcher():
    """`python bench.py --gpus 2` exactly as the driver calls it (one plain process, no torchrun, no rank
    environment): the parent must start 2 fresh rank processes, relay rank 0's single JSON line and report what
    torch.distributed saw.  --plumbing-only keeps the GPU out of it (gloo rendezvous, barrier, MAX over ranks)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--plumbing-only"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["launch"] == "self-spawned"
    assert lines[0]["elapsed_max_over_ranks_s"] >= 0.02           # MAX over ranks: rank 1 sleeps 20 ms
    # every rank reports where it runs (launch.bind_rank_to_gpu_numa; no GPU topology in this container: left unbound)
    pr = lines[0]["per_rank"]
    assert len(pr["cpus"]) == 2 and all(isinstance(c, str) and c for c in pr["cpus"]) and len(pr["cpu_binding"]) == 2


def test_bench_gpus_8_plumbing_as_the_scaling_run_starts_it():
    """the driver's scaling run ends at `bench.py --gpus 8`: eight self-spawned ranks, one rendezvous, one JSON line with
    eight per-rank entries (GPU work left out: --plumbing-only)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo",
                        "--plumbing-only"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["ranks_seen"] == 8, r.stdout
    assert len(lines[0]["per_rank"]["cpus"]) == 8 and lines[0]["elapsed_max_over_ranks_s"] >= 0.08    # rank 7 sleeps 80 ms


def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """the contract's N>1 form, word for word: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W`: the ranks come from the launcher's
    environment (no self-spawn), rank 0 alone prints the line"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--backend", "gloo", "--plumbing-only"], env=_clean_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["launch"] == "external launcher"
    assert len(lines[0]["per_rank"]["cpus"]) == 2


def test_bring_up_fails_fast_with_a_diagnosis_when_a_rank_never_arrives():
    """the first real multi-GPU run must not sit in init_process_group's 30-minute default.
    Rank 0 of a world of two is started and rank 1 never is (a stalled / crashed peer): within --bringup-timeout seconds
    rank 0 prints what a maintainer needs (backend, devices, HSA_ENABLE_IPC_MODE_LEGACY, its device, the rendezvous) and
    exits with code 3 -- both entry points, bench.py and train.py."""
    import subprocess
    import time
    for script, extra in (("bench.py", ["--gpus", "2", "--plumbing-only", "--backend", "gloo"]),
                          ("train.py", ["--gpus", "2", "--device", "cpu", "--loss", "l1", "--steps", "1", "--workers", "0"])):
        env = _clean_env()
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        t0 = time.monotonic()
        r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + extra + ["--bringup-timeout", "4"], env=env,
                           capture_output=True, text=True, timeout=120)
        took = time.monotonic() - t0
        assert r.returncode == 3, (script, r.returncode, r.stderr[-1500:])
        assert took < 60, took
        err = r.stderr
        assert "[bring-up] rank 0 of 2" in err and "init_process_group" in err, err[-1500:]
        assert "backend                      gloo" in err and "HSA_ENABLE_IPC_MODE_LEGACY   0" in err
        assert "devices visible to torch" in err and "127.0.0.1:%s" % env["MASTER_PORT"] in err
        assert not _json_lines(r.stdout)                                    # no result line from a run that never ran


def test_bring_up_error_is_a_diagnosis_not_a_bare_traceback():
    """a rendezvous address that does not resolve: torch's store client keeps retrying (minutes), so it is again the
    watchdog that ends the rank after --bringup-timeout -- same diagnosis, code 3"""
    import subprocess
    env = _clean_env()
    env.update(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="no-such-host.invalid",
               MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--backend", "gloo",
                        "--bringup-timeout", "20"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert "[bring-up] rank 1 of 2" in r.stderr and "no-such-host.invalid:29999" in r.stderr


def test_bring_up_error_can_be_an_exception_for_library_callers():
    """exit_on_failure=False: a caller that is not a rank script gets ProcessGroupBringupError with the diagnosis and the
    original exception as its cause (here: a backend torch does not know -- raised at once, inside the watchdog's window)"""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29997')\n"
            "from svbrdf_estimation_amd import distributed as D\n"
            "try:\n"
            "    D.init_process_group_checked('no-such-backend', None, 30.0, exit_on_failure=False)\n"
            "except D.ProcessGroupBringupError as e:\n"
            "    assert '127.0.0.1:29997' in str(e) and 'no-such-backend' in str(e) and e.__cause__ is not None\n"
            "    print('RAISED')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "RAISED" in r.stdout, (r.returncode, r.stdout, r.stderr[-1500:])
    # ... and the rank scripts' mode: same failure, diagnosis on stderr, exit code 3
    r = subprocess.run([sys.executable, "-c", code.replace("exit_on_failure=False", "exit_on_failure=True")], env=_clean_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "[bring-up] rank 0 of 1" in r.stderr and "RAISED" not in r.stdout, (r.returncode, r.stderr[-800:])


def test_bring_up_error_after_a_successful_init_leaves_no_group_behind_and_can_be_retried():
    """exit_on_failure=False when the group DID come up but the first collective counts the wrong number of ranks: the
    default group is destroyed again before the error is raised, so the caller can retry (a second init_process_group
    would otherwise fail with "initialised twice" and the half-checked communicator would stay alive); the watchdog leaves
    faulthandler's process-wide timer to whoever armed it."""
    import subprocess
    code = ("import os, sys, faulthandler; sys.path.insert(0, %r)\n"
            "from svbrdf_estimation_amd import launch\n"
            "os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(launch.free_port()))\n"
            "import torch.distributed as dist\n"
            "from svbrdf_estimation_amd import distributed as D\n"
            "faulthandler.dump_traceback_later(3.0, exit=True)          # the application's own hang dump: must survive\n"
            "real = dist.all_reduce\n"
            "def doubled(t, op=None, **kw):\n"
            "    real(t, op=op, **kw); t.mul_(2)                        # the collective 'sees' two ranks in a world of one\n"
            "dist.all_reduce = doubled\n"
            "try:\n"
            "    D.init_process_group_checked('gloo', None, 30.0, exit_on_failure=False)\n"
            "    raise SystemExit('no error raised')\n"
            "except D.ProcessGroupBringupError as e:\n"
            "    assert 'summed 2 ones over a group of 1' in str(e), e\n"
            "assert not dist.is_initialized()\n"
            "dist.all_reduce = real\n"
            "assert D.init_process_group_checked('gloo', None, 30.0, exit_on_failure=False) == 1   # the retry\n"
            "dist.destroy_process_group()\n"
            "print('RETRIED', flush=True)\n"
            "import time; time.sleep(5)                                  # the timer armed above was not cancelled by the watchdog\n"
            "print('TIMER-LOST')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert "RETRIED" in r.stdout and "TIMER-LOST" not in r.stdout and r.returncode == 1, (r.returncode, r.stdout, r.stderr[-1500:])
    assert "Timeout (0:00:03)!" in r.stderr


def test_selftest_collective_part_world_of_two_on_cpu():
    """`bench.py --gpus 2 --selftest --plumbing-only` (self-spawned, gloo, no GPU): the collective part of the first-contact
    self-test -- the gathers, the 8 MB and 320 MB all-reduces with their rates, sums checked, the verdict -- with the
    device checks reported as skipped.  (With devices it runs in the GPU suite: tests/test_gpu_multirank.py.)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest", "--plumbing-only"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = _json_lines(r.stdout)
    st = line["selftest"]
    assert line["ranks_seen"] == 2 and st["ok"] is True and st["ranks_seen"] == 2 and st["distinct_devices"] == 2
    assert st["allreduce_sums_correct"] is True and st["transport"].startswith("gloo")
    for size in ("8MB", "320MB"):
        assert len(st["allreduce_GBps"][size]) == 2 and all(v > 0 for v in st["allreduce_GBps"][size])
    assert all(p.get("skipped") for p in st["parity"])


def test_bench_refuses_a_world_that_is_not_gpus():
    """a launcher environment whose WORLD_SIZE contradicts --gpus must be an error, not a silent 1-rank run"""
    import subprocess
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not _json_lines(r.stdout)


def test_spawned_rank_failure_stops_the_others_and_fails_the_parent(tmp_path):
    from svbrdf_estimation_amd import launch
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
                      "time.sleep(60)\n")
    import time
    t0 = time.monotonic()
    rc = launch.spawn_ranks(str(script), [], 3)
    assert rc == 7 and time.monotonic() - t0 < 30          # ranks 0 and 2 were terminated, not waited for


def test_train_gpus_n_spawns_n_ranks_cpu(tmp_path):
    """train.py --gpus 2 as one plain process (CPU plumbing configuration: gloo, stock L1 loss)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--gpus", "2", "--device", "cpu", "--loss", "l1",
                        "--steps", "1", "--warmup", "0", "--batch", "1", "--workers", "0", "--ddp-probe"], env=_clean_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2
    # the DDP probe steps after the timed region: a step inside model.no_sync() (forward AND backward: DDP decides at the
    # forward) must leave the gradients rank-local, a synchronised one equal on both ranks -- the probe checks what it times
    probe = lines[0]["ddp_backward_probe"]
    assert probe["samples_each"] == 3 and probe["backward_ms_no_sync"] > 0 and probe["backward_ms_with_allreduce"] > 0
    assert probe["grad_checksum_spread_over_ranks_synced"] <= 1e-9 < probe["grad_checksum_spread_over_ranks_no_sync"]
    assert probe["no_sync_left_gradients_rank_local"] is True


def test_verify_global_batch_two_ranks_equal_one_cpu(tmp_path):
    """train.py --verify-global-batch on the CPU plumbing configuration (gloo, stock L1 loss): two self-spawned ranks
    with one item each against one process with both items -- contiguous shards, dropout off, DDP's averaged gradient
    is the global-batch gradient.  (tests/test_gpu_multirank.py runs the same check on the GPU with the fused loss.)"""
    import subprocess
    common = ["--device", "cpu", "--loss", "l1", "--steps", "1", "--warmup", "0", "--workers", "0", "--samples", "2"]
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    for extra in (["--gpus", "1", "--batch", "2", "--verify-global-batch", one],
                  ["--gpus", "2", "--batch", "1", "--verify-global-batch", two]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py")] + extra + common, env=_clean_env(),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert int(b["world"]) == 2 and a["grad"].shape == b["grad"].shape and a["grad"].size > 79_000_000
    assert abs(float(a["loss"]) - float(b["loss"])) <= 1e-6 * abs(float(a["loss"]))
    assert np.abs(a["grad"] - b["grad"]).max() <= 1e-5 * np.abs(a["grad"]).max()


def test_contiguous_shard_sampler_tiles_the_global_batches():
    from svbrdf_estimation_amd import distributed
    per, world, n = 3, 4, 31
    seen = [list(distributed.ContiguousShardSampler(n, per, r, world)) for r in range(world)]
    assert all(len(s) == 2 * per for s in seen)                       # two whole global batches of 12, the tail dropped
    for k in range(2):
        block = sum((s[k * per:(k + 1) * per] for s in seen), [])
        assert block == list(range(k * 12, (k + 1) * 12))


def test_resume_accepts_both_checkpoint_layouts_and_counts_steps(tmp_path):
    """train.py --resume: a checkpoint in the reference's parameter names (what --save writes and what the reference's
    persistence.py writes) and one in this package's own names (written by --save before it switched layouts) both
    load; the step count carries over into the next --save."""
    import train
    from svbrdf_estimation_amd.training import models
    torch.manual_seed(0)
    net = models.SingleViewModel()
    own, ref = str(tmp_path / "own.tar"), str(tmp_path / "ref.tar")
    torch.save({"model_state_dict": net.state_dict(), "steps": 7}, own)
    torch.save({"model_state_dict": models.convert_to_reference_state_dict(net.state_dict()), "steps": 5}, ref)
    torch.set_num_threads(4)
    for src, before in ((own, 7), (ref, 5)):
        out = str(tmp_path / "out.tar")
        train.run(train.parse_args(["--device", "cpu", "--loss", "l1", "--steps", "1", "--warmup", "0", "--batch", "1",
                                    "--workers", "0", "--resume", src, "--save", out, "--lr", "0"]))
        ck = torch.load(out, map_location="cpu", weights_only=False)
        assert ck["steps"] == before + 1
        back = models.convert_reference_state_dict(ck["model_state_dict"])
        assert all(torch.equal(back[k], v) for k, v in net.state_dict().items())       # lr 0: the weights are the loaded ones


def _fake_sysfs(root, gpus, nodes, readable=None):
    """a sysfs tree shaped like the MI355X node the GPU boxes are slots of: KFD topology nodes 0..1 = CPU sockets, then one
    node per GPU (drm_render_minor 128 + 8 i); `gpus` = the NUMA node of each; `nodes` = {numa node: cpulist};
    `readable` = indices of the GPUs whose KFD properties this process may read (a device cgroup hides the others)."""
    def put(rel, text):
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    for n in range(2):
        put("class/kfd/kfd/topology/nodes/%d/properties" % n, "cpu_cores_count 128\nsimd_count 0\ndrm_render_minor 0\n")
    for i, numa in enumerate(gpus):
        minor = 128 + 8 * i
        if readable is None or i in readable:
            put("class/kfd/kfd/topology/nodes/%d/properties" % (2 + i),
                "cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\nunique_id %d\n" % (minor, 1000 + i))
        else:
            os.makedirs(os.path.join(root, "class/kfd/kfd/topology/nodes/%d" % (2 + i)), exist_ok=True)   # no properties
        put("class/drm/renderD%d/device/numa_node" % minor, "%d\n" % numa)
    from svbrdf_estimation_amd import launch
    for numa, cpulist in nodes.items():
        put("devices/system/node/node%d/cpulist" % numa, cpulist + "\n")
        cpus = launch.parse_cpulist(cpulist)
        half = len(cpus) // 2
        for k, c in enumerate(cpus):                    # SMT: the second half of a node's list are the siblings of the first
            sib = sorted((c, cpus[(k + half) % len(cpus)]))
            put("devices/system/cpu/cpu%d/topology/thread_siblings_list" % c, "%d,%d\n" % tuple(sib))


def test_rank_cpu_placement_follows_the_gpu_numa_node(tmp_path):
    """launch.rank_cpu_placement on a fake sysfs of the two-socket, eight-GPU node (GPUs 0-3 on node 0, 4-7 on node 1;
    256 CPUs, SMT siblings c and c + 128): every rank lands on ITS GPU's socket, ranks of one socket get disjoint
    16-core slices with both SMT siblings, together exactly the socket; *_VISIBLE_DEVICES and a device cgroup that
    hides the other GPUs are honoured; opaque settings leave the rank unbound rather than misplaced."""
    from svbrdf_estimation_amd import launch
    root = str(tmp_path / "sys")
    nodes = {0: "0-63,128-191", 1: "64-127,192-255"}
    _fake_sysfs(root, [0, 0, 0, 0, 1, 1, 1, 1], nodes)
    allowed = list(range(256))
    assert launch.gpu_render_minors(root, {}) == [128 + 8 * i for i in range(8)]
    places = [launch.rank_cpu_placement(r, 8, sysfs=root, environ={}, allowed=allowed) for r in range(8)]
    assert [p["numa_node"] for p in places] == [0] * 4 + [1] * 4 and all(p["ranks_on_node"] == 4 for p in places)
    for node, ranks in ((0, range(0, 4)), (1, range(4, 8))):
        sets = [set(places[r]["cpus"]) for r in ranks]
        assert all(len(s) == 32 for s in sets)                                   # 16 cores x 2 threads
        assert all(a.isdisjoint(b) for i, a in enumerate(sets) for b in sets[i + 1:])
        assert set().union(*sets) == set(launch.parse_cpulist(nodes[node]))
        assert all((c + 128 in s) == (c < 128) or c >= 128 for s in sets for c in s if c < 128)   # siblings stay together
    assert places[0]["cpus"][:3] == [0, 1, 2] and 128 in places[0]["cpus"] and places[5]["render_minor"] == 168
    # one rank alone on its socket gets the whole socket; two ranks sharing ONE device split its socket
    assert len(launch.rank_cpu_placement(0, 1, sysfs=root, environ={}, allowed=allowed)["cpus"]) == 128
    a = launch.rank_cpu_placement(0, 2, share_device=True, sysfs=root, environ={}, allowed=allowed)
    b = launch.rank_cpu_placement(1, 2, share_device=True, sysfs=root, environ={}, allowed=allowed)
    assert a["numa_node"] == b["numa_node"] == 0 and set(a["cpus"]).isdisjoint(b["cpus"]) and len(a["cpus"]) == len(b["cpus"]) == 64
    # ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES select and reorder by index (HIP indexes ROCr's visible list)
    assert launch.gpu_render_minors(root, {"ROCR_VISIBLE_DEVICES": "5,1"}) == [168, 136]
    assert launch.gpu_render_minors(root, {"ROCR_VISIBLE_DEVICES": "5,1", "HIP_VISIBLE_DEVICES": "1"}) == [136]
    assert launch.rank_cpu_placement(0, 1, sysfs=root, environ={"ROCR_VISIBLE_DEVICES": "6"}, allowed=allowed)["numa_node"] == 1
    # the GPU boxes: only the slot's GPU is readable in the KFD topology, and it is device 0 whatever its position
    root2 = str(tmp_path / "sys2")
    _fake_sysfs(root2, [0, 0, 0, 0, 1, 1, 1, 1], nodes, readable=[7])
    p = launch.rank_cpu_placement(0, 1, sysfs=root2, environ={"ROCR_VISIBLE_DEVICES": "0", "HIP_VISIBLE_DEVICES": "0"}, allowed=allowed)
    assert p["render_minor"] == 184 and p["numa_node"] == 1 and launch.format_cpulist(p["cpus"]) == "64-127,192-255"
    # an affinity mask narrower than the socket is respected; UUIDs, missing topology, more ranks than GPUs: unbound
    narrow = launch.rank_cpu_placement(0, 1, sysfs=root, environ={}, allowed=range(0, 8))
    assert narrow["cpus"] == list(range(8)) and narrow["numa_node"] == 0
    for env, world in (({"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}, 1), ({}, 9)):
        u = launch.rank_cpu_placement(0, world, sysfs=root, environ=env, allowed=allowed)
        assert u["numa_node"] is None and u["cpus"] == allowed and u["source"].startswith("unbound")
    assert launch.rank_cpu_placement(0, 1, sysfs=str(tmp_path / "nothing"), environ={}, allowed=allowed)["numa_node"] is None
    # a compute-partition render node has no numa_node of its own: the PCI function the KFD topology names is asked
    root3 = str(tmp_path / "sys3")
    _fake_sysfs(root3, [1], nodes)
    os.remove(os.path.join(root3, "class/drm/renderD128/device/numa_node"))
    with open(os.path.join(root3, "class/kfd/kfd/topology/nodes/2/properties"), "a") as f:
        f.write("domain 0\nlocation_id 41984\n")                              # bus 0xa4, device 0, function 0
    os.makedirs(os.path.join(root3, "bus/pci/devices/0000:a4:00.0"))
    with open(os.path.join(root3, "bus/pci/devices/0000:a4:00.0/numa_node"), "w") as f:
        f.write("1\n")
    p3 = launch.rank_cpu_placement(0, 1, sysfs=root3, environ={}, allowed=allowed)
    assert p3["numa_node"] == 1 and p3["pci"] == "0000:a4:00.0"
    # a container that was given only SOME /dev/dri/renderD* (sysfs still lists every GPU of the host): only the render
    # nodes this process may open count -- the runtime enumerates exactly those -- so HIP index 0 is GPU 5 here, not GPU 0
    fake_dev = tmp_path / "dev"
    (fake_dev / "dri").mkdir(parents=True)
    for m in (168, 176):
        (fake_dev / "dri" / ("renderD%d" % m)).write_text("")
    assert launch.gpu_render_minors(root, {}, dev=str(fake_dev)) == [168, 176]
    assert launch.gpu_render_minors(root, {}, dev=None) == [128 + 8 * i for i in range(8)]
    os.chmod(str(fake_dev / "dri" / "renderD176"), 0o000)
    if os.geteuid() != 0:                                                      # root opens anything: nothing to test then
        assert launch.gpu_render_minors(root, {}, dev=str(fake_dev)) == [168]
    # a properties file this code does not understand leaves the rank unbound instead of raising at start-up
    root4 = str(tmp_path / "sys4")
    _fake_sysfs(root4, [0, 1], nodes)
    with open(os.path.join(root4, "class/kfd/kfd/topology/nodes/2/properties"), "a") as f:
        f.write("simd_count many\n")
    assert launch.gpu_render_minors(root4, {}) is None
    u = launch.rank_cpu_placement(0, 1, sysfs=root4, environ={}, allowed=allowed)
    assert u["numa_node"] is None and u["source"].startswith("unbound")


def test_placement_crosscheck_against_the_runtime_pci_address(monkeypatch):
    """launch.crosscheck_placement: the PCI address read from sysfs for the rank's GPU against the one the runtime reports
    for the device the rank got; a mismatch undoes the CPU binding (it would sit on another GPU's socket)"""
    import types
    from svbrdf_estimation_amd import launch
    before = os.sched_getaffinity(0)
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xa4, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: props)
    ok = launch.crosscheck_placement({"bound": True, "pci": "0000:a4:00.0", "cpus": "0-1", "n_cpus": 2, "numa_node": 1, "source": "sysfs"}, 0, before)
    assert ok["pci_crosscheck"] == "match" and ok["bound"]
    bad = launch.crosscheck_placement({"bound": True, "pci": "0000:25:00.0", "cpus": "0-1", "n_cpus": 2, "numa_node": 0, "source": "sysfs"}, 0, before)
    assert bad["pci_crosscheck"].startswith("mismatch") and not bad["bound"] and bad["numa_node"] is None
    assert bad["n_cpus"] == len(before) and os.sched_getaffinity(0) == before
    assert launch.crosscheck_placement({"bound": False, "pci": None}, 0)["pci_crosscheck"].startswith("not applicable")


def test_bind_rank_pins_the_process_and_reports_a_cpulist(tmp_path):
    """bind_rank_to_gpu_numa in a child process (so that this suite's own affinity is untouched): with a readable
    topology the affinity mask becomes the slice; without one it stays; both report compact cpulists."""
    import json
    import subprocess
    root = str(tmp_path / "sys")
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        pytest.skip("needs two CPUs")
    _fake_sysfs(root, [0, 0], {0: "%d-%d" % (have[0], have[-1])})
    code = ("import json, os, sys; sys.path.insert(0, %r)\n"
            "from svbrdf_estimation_amd import launch\n"
            "real = launch.rank_cpu_placement\n"
            "launch.rank_cpu_placement = lambda r, w, s=False: real(r, w, s, sysfs=%r, environ={})\n"
            "p = launch.bind_rank_to_gpu_numa(1, 2)\n"
            "print(json.dumps({'p': p, 'now': sorted(os.sched_getaffinity(0))}))\n" % (ROOT, root))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    from svbrdf_estimation_amd import launch
    assert out["p"]["bound"] and out["p"]["numa_node"] == 0 and launch.parse_cpulist(out["p"]["cpus"]) == out["now"]
    assert 0 < len(out["now"]) < len(have) and set(out["now"]) <= set(have) and out["p"]["n_cpus"] == len(out["now"])
    env = dict(os.environ, SVBRDF_NO_CPU_BINDING="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert not out["p"]["bound"] and out["now"] == have
