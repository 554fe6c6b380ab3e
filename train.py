#!/usr/bin/env python3
"""train.py -- data-parallel training harness around the rendering-loss engine.

One process per GPU (started by torchrun, or by this script itself: ``--gpus N`` without a rank environment
spawns N fresh rank processes before anything touches a GPU, svbrdf_estimation_amd/launch.py), batch sharded by rank with a DistributedSampler, stock
DistributedDataParallel for the U-Net: its ~320 MB of fp32 gradients per step are all-reduced by
RCCL over xGMI in buckets overlapped with the backward pass (backend "nccl" is RCCL on ROCm).  The
rendering / mixed loss itself needs no collective: each rank evaluates it on its shard with its own
scene RNG stream (seed + rank) and DDP's gradient averaging makes the result the global-batch
gradient.  What the reference's main.py:56-150 does on one GPU, in the reference's order:
batch -> (synthesise missing photos) -> model -> MixedLoss -> backward -> Adam(lr 1e-5).

  python train.py --steps 20                                   # 1 GPU, synthetic SVBRDFs, single-view
  python train.py --gpus 8 --batch 8 --steps 100               # 8 GPUs: spawns its own 8 ranks (config 3: global batch 64)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --gpus 8 --batch 8 --steps 100
  python train.py --model multi --views 5 --batch 16           # BASELINE config 4
  python train.py --size 512 --random-scenes 11 --specular-scenes 21   # BASELINE config 5 (per GPU)
  python train.py --data /path/to/tiled_pngs --image-count 10  # Deschaintre tiled-PNG samples

  python train.py --gpus 1 --force-dist                        # one GPU, but every multi-rank branch (RCCL process group, DDP)
  python train.py --gpus 2 --backend gloo --share-device       # two ranks on one GPU (control flow of an N-GPU run)

Convolutions: the U-Net runs on stock PyTorch-ROCm (MIOpen).  ``--conv-mode auto`` (default) picks, per direction, between
the reference's cudnn flags and MIOpen's immediate-mode choice (a four-step calibration of the forward; backward always
immediate: the reference's ``deterministic=True`` pins MIOpen to a backward 5-20x slower) -- DESIGN.md section 10.

Prints one JSON line per run on rank 0 (end-to-end patches/s, mean loss of the first/last steps).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0,
                    help="ranks (one per GPU).  0: whatever the launcher's WORLD_SIZE says (1 without a launcher)")
    ap.add_argument("--data", default="synthetic",
                    help="'synthetic': random SVBRDF maps drawn ON THE DEVICE, one batch per step, per-rank generator (no "
                         "dataloader: one CPU worker produces ~60 synthetic samples/s, less than a training step consumes); "
                         "'synthetic-cpu': the same statistics from a torch Dataset through the DataLoader / "
                         "DistributedSampler (what --device cpu and --verify-global-batch use); or a directory of tiled "
                         "PNG samples (the reference's SvbrdfDataset format)")
    ap.add_argument("--image-count", type=int, default=10, help="photos stored per tiled PNG")
    ap.add_argument("--scale-mode", choices=("crop", "resize"), default="crop")      # cli.py scale mode, dataset.py:58-89
    ap.add_argument("--random-crop", action="store_true")
    ap.add_argument("--mix-materials", action="store_true",
                    help="material-mixing augmentation (dataset.py:52-56, 142-160; main.py:49 enables it for training): "
                         "partner and weight drawn in the dataloader, blend on the GPU (kernel K4); needs --image-count 0")
    ap.add_argument("--linear-input", action="store_true")
    ap.add_argument("--model", choices=("single", "multi"), default="single")
    ap.add_argument("--views", type=int, default=1, help="input photos per sample (multi-view: N)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lr", type=float, default=1e-5)                    # main.py:74
    ap.add_argument("--l1-weight", type=float, default=0.1)              # losses.py:55
    ap.add_argument("--random-scenes", type=int, default=3)              # losses.py:26
    ap.add_argument("--specular-scenes", type=int, default=6)            # losses.py:27
    ap.add_argument("--loss", choices=("mixed", "rendering", "l1"), default="mixed")
    ap.add_argument("--fused-head", action="store_true", help="model returns 9 channels, head decoded in the loss kernel")
    ap.add_argument("--no-coords", action="store_true")
    ap.add_argument("--samples", type=int, default=0, help="synthetic dataset length (default: enough for --steps)")
    ap.add_argument("--workers", type=int, default=-1,
                    help="DataLoader worker processes per rank; -1 = from the host: (cpus / ranks on this node) - 1, at "
                         "least 2, at most 16 -- the tiled-PNG reader delivers ~45-60 samples/s per worker "
                         "(profiles/r03_loader_rate.json), a training step consumes hundreds per GPU")
    ap.add_argument("--float-transport", action="store_true",
                    help="tiled-PNG samples: decode to float32 in the DataLoader workers (the reference's way) instead of "
                         "shipping the cropped 8-bit pixels and decoding them on the device with lookup tables (same bits, a "
                         "quarter of the bytes; 'crop' scale mode only)")
    ap.add_argument("--no-pin", action="store_true", help="DataLoader without the pinned-memory staging thread")
    ap.add_argument("--prefetch", type=int, default=4, help="batches each DataLoader worker keeps ready (prefetch_factor)")
    ap.add_argument("--seed", type=int, default=313)                     # utils.py:7
    ap.add_argument("--device", default="cuda", choices=("cuda", "cpu"))
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl on cuda, gloo on cpu)")
    ap.add_argument("--save", default=None, help="write a checkpoint (model + optimizer state) here at the end")
    ap.add_argument("--resume", default=None,
                    help="checkpoint to start from: one written by --save, or a checkpoint.tar of the reference "
                         "(persistence.py:52-69; model weights only)")
    ap.add_argument("--conv-mode", choices=("auto", "hybrid", "reference", "fast", "autotune"), default="auto",
                    help="how stock PyTorch-ROCm picks the U-Net's MIOpen convolution algorithms (DESIGN.md section 10).  "
                         "reference: cudnn.deterministic=True, benchmark=False as utils.py:11-12 sets them -- on ROCm a fine "
                         "forward and a pathological backward (2.7 s per step at configs[3]); fast: deterministic=False, "
                         "benchmark=False -- MIOpen's immediate-mode choice, no search: good backward; its forward is 5x "
                         "slower than the reference-flag forward on a box without stored find results and 30 %% faster with "
                         "them; hybrid: the reference's flags while the network runs forward, the immediate-mode choice while "
                         "autograd runs backward (the flags are read when a convolution executes); auto (default): hybrid, "
                         "plus a calibration during the first four steps that times the forward under both settings and "
                         "keeps the faster one -- no search, right on a fresh box and on one with a tuned cache; autotune: "
                         "benchmark=True -- MIOpen's find step compiles and times every candidate at the first call of "
                         "each shape (7 min at config 2; > 44 min at configs[3]); its results land in MIOpen's user cache")
    ap.add_argument("--autotune", action="store_true", help="same as --conv-mode autotune")
    ap.add_argument("--channels-last", action="store_true", help="NHWC activations/weights for the U-Net")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (with --backend gloo: the N-rank control flow, DDP included, on a box with "
                         "fewer GPUs than ranks; tests/test_gpu_multirank.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one rank: still create the process group (world size 1), wrap the network in DDP and take "
                         "every multi-rank branch, so that the RCCL code an N-GPU run executes runs on a one-GPU box")
    ap.add_argument("--verify-global-batch", default=None, metavar="NPZ",
                    help="verification run: makes an N-rank run and a one-rank run of the same GLOBAL batch comparable -- "
                         "contiguous unshuffled shards, dropout off, the network's input photo taken from the diffuse map, "
                         "and ONE scene stream (every rank seeds alike and skips the draws that belong to the other ranks' "
                         "items) -- and writes rank 0's gradient after the first backward (DDP-averaged) and the global "
                         "loss to this file")
    ap.add_argument("--bringup-timeout", type=float, default=60.0,
                    help="multi-rank: seconds the rendezvous + communicator set-up + first all-reduce may take before the "
                         "rank prints a diagnosis and exits with code 3 (the limit covers the rendezvous: raise it for multi-node "
                         "jobs and slow cold starts)")
    ap.add_argument("--no-ddp-probe", action="store_true",
                    help="skip the eight untimed steps after the timed region that time a whole step's forward + backward "
                         "with DDP's all-reduce and under model.no_sync() (multi-rank GPU runs; CPU runs with --ddp-probe)")
    ap.add_argument("--ddp-probe", action="store_true",
                    help="run the DDP probe steps on --device cpu too (host clock instead of HIP events): the gloo tests of "
                         "the probe's control flow")
    ap.add_argument("--phase-times", action="store_true",
                    help="bracket the phases of every timed step (dataloader wait on the host clock; upload + input "
                         "synthesis, network forward, loss, backward, optimizer with HIP events) and report their means: "
                         "costs one device synchronisation per step, so `value` of such a run is not a throughput figure")
    return ap.parse_args(argv)


def run(args):
    from svbrdf_estimation_amd import distributed, losses, renderers, training, utils
    from svbrdf_estimation_amd.training import data, models
    import torch.distributed as dist
    miopen_cache = training.use_in_tree_miopen_cache()      # before the first convolution of the process

    # this pool's host driver only supports dmabuf IPC: RCCL between rank processes needs it (read when the HSA runtime
    # starts, i.e. at the first GPU call; the self-spawning parent sets it for its children too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))      # ranks on THIS node (multi-node: world > local_world)
    if args.gpus and args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (or run train.py without a rank "
                         "environment and let it spawn them)" % (args.gpus, world))
    on_gpu = args.device == "cuda"
    # first thing in a rank, before any GPU call: pin it (and the DataLoader workers it will fork) to the CPUs next to its
    # GPU, ranks on one socket splitting that socket's cores (launch.py)
    from svbrdf_estimation_amd import launch
    host_cpus = os.sched_getaffinity(0)
    placement = launch.bind_rank_to_gpu_numa(local_rank, local_world, args.share_device) if on_gpu else \
        {"cpus": launch.format_cpulist(os.sched_getaffinity(0)), "n_cpus": len(os.sched_getaffinity(0)), "numa_node": None,
         "source": "unbound: --device cpu", "bound": False}
    if on_gpu:
        assert torch.cuda.is_available(), "no ROCm device visible"
        if args.share_device:
            local_rank = 0
            if world > 1 and (args.backend or "nccl") == "nccl":
                # RCCL refuses two ranks on one device (and would hang in init or the first collective before saying so)
                if args.backend == "nccl":
                    raise SystemExit("--share-device puts every rank on cuda:0, which RCCL does not support: use --backend gloo")
                args.backend = "gloo"
        elif torch.cuda.device_count() <= local_rank:
            raise SystemExit("local rank %d but only %d device(s) visible on this node" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        placement = launch.crosscheck_placement(placement, local_rank, host_cpus)   # PCI address: sysfs vs the runtime
    else:
        if args.loss != "l1":
            raise SystemExit("the rendering loss has no CPU path; --device cpu only supports --loss l1 (plumbing tests)")
        if args.mix_materials:
            raise SystemExit("--mix-materials blends on the GPU (kernel K4); there is no CPU path for it")
        dev = torch.device("cpu")
    ranks_seen = 1
    grouped = world > 1 or args.force_dist
    backend = (args.backend or ("nccl" if on_gpu else "gloo")) if grouped else None
    nccl = backend == "nccl"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                          # --force-dist without a launcher: a rendezvous of one
            from svbrdf_estimation_amd import launch
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # rendezvous + communicator + one all-reduce under a watchdog (distributed.init_process_group_checked): exit code 3
        # and a diagnosis after --bringup-timeout seconds instead of the launcher's limit
        ranks_seen = distributed.init_process_group_checked(backend, dev if nccl else None, args.bringup_timeout)
        if on_gpu:      # one rank per GPU, or the job stops here with the reason (--share-device: the plumbing tests' exception)
            distributed.require_distinct_devices(dev, args.share_device, rank)

    def barrier():
        if nccl:
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()

    verify = args.verify_global_batch
    # per-rank scene RNG -- or, for --verify-global-batch, one stream shared by all ranks (each skips the others' draws)
    utils.enable_deterministic_random_engine(args.seed if verify else distributed.rank_seed(args.seed, rank))
    if on_gpu:      # the rank's own CPU work is tiny (scene sampler, collation): a big intra-op pool only spins (bench.py main)
        torch.set_num_threads(max(1, min(8, placement["n_cpus"])))
    decode = not args.fused_head
    if args.model == "multi":
        net = models.MultiViewModel(use_coords=not args.no_coords, decode=decode)
    else:
        net = models.SingleViewModel(use_coords=not args.no_coords, decode=decode)
    # identical initial weights on every rank: DDP broadcasts rank 0's parameters at construction
    conv_mode = "autotune" if args.autotune else args.conv_mode
    if conv_mode in ("fast", "autotune"):       # enable_deterministic_random_engine set the reference's flags above
        torch.backends.cudnn.deterministic = False
        torch.backends.cudnn.benchmark = conv_mode == "autotune"
    hybrid = conv_mode in ("hybrid", "auto") and on_gpu
    auto = conv_mode == "auto" and on_gpu
    # auto: forward under the reference's flags at steps 0-1, under the immediate-mode choice at steps 2-3 (the first of
    # each pair compiles, the second is timed with HIP events); from step 4 on the faster of the two.  The four
    # calibration steps are EXTRA untimed steps in front of the warm-up (they carry two device synchronisations and two
    # steps under the slower setting: never part of `value`, whatever --warmup says), and the ranks agree on the choice:
    # the timings are MAX-reduced over the ranks before they are compared.
    forward_flag, calib = True, {}
    calib_steps = 4 if auto else 0
    steps_done = 0
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        state = ck["model_state_dict"]
        own = set(net.state_dict().keys())
        # a checkpoint in the reference's parameter names (what --save writes, and what the reference writes), or one in
        # this package's own names (written by --save before it switched to the reference's layout)
        net.load_state_dict(state if set(state.keys()) <= own else models.convert_reference_state_dict(state))
        steps_done = int(ck.get("steps", 0))
    net = net.to(dev).train()
    if verify:
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
    # gradient_as_bucket_view: the parameters' .grad ARE views of the all-reduce buckets (no 320 MB copy per step)
    model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank] if on_gpu else None,
                                                      gradient_as_bucket_view=True) if grouped else net
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)
    if args.resume and "optimizer_state_dict_amd" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict_amd"])

    if args.loss == "l1":
        loss_fn = losses.SVBRDFL1Loss()
        if args.fused_head:
            raise SystemExit("--fused-head needs the rendering or mixed loss")
    else:
        weight = args.l1_weight if args.loss == "mixed" else 0.0
        loss_fn = (losses.FusedHeadLoss if args.fused_head else losses.MixedLoss)(renderers.LocalRenderer(), weight)
        loss_fn.rendering_loss.random_configuration_count = args.random_scenes
        loss_fn.rendering_loss.specular_configuration_count = args.specular_scenes

    total_steps = calib_steps + args.warmup + args.steps
    if args.workers < 0:        # the rank's CPU set is its share of the node already (placement), else split the host
        mine = placement["n_cpus"] if placement.get("bound") else (os.cpu_count() or 2) // max(1, local_world)
        args.workers = max(2, min(16, mine - 1))
    device_source = args.data == "synthetic" and on_gpu and not verify
    if device_source:
        dataset, args.workers = None, 0
    elif args.data in ("synthetic", "synthetic-cpu"):
        n = args.samples or total_steps * args.batch * world
        dataset = data.SyntheticSvbrdfDataset(n, image_size=args.size, seed=args.seed)
    else:
        dataset = data.TiledPngDataset(args.data, image_size=args.size, image_count=args.image_count,
                                       used_image_count=args.views, is_linear=args.linear_input,
                                       scale_mode=args.scale_mode, random_crop=args.random_crop,
                                       mix_materials=args.mix_materials, uint8_transport=on_gpu and not args.float_transport)
    if device_source:
        sampler = loader = None
    elif verify:
        sampler = distributed.ContiguousShardSampler(len(dataset), args.batch, rank, world)
    else:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True,
                                                                  seed=args.seed, drop_last=True) if world > 1 else None
    if not device_source:
        loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch, sampler=sampler, shuffle=sampler is None,
                                             num_workers=args.workers, pin_memory=on_gpu and not args.no_pin, drop_last=True,
                                             persistent_workers=args.workers > 0,
                                             **({"prefetch_factor": args.prefetch} if args.workers > 0 else {}))

    def batches():
        if device_source:
            gen = torch.Generator(device=dev).manual_seed(distributed.rank_seed(args.seed * 7919, rank))
            empty = torch.zeros(args.batch, 0, 3, args.size, args.size)
            while True:
                yield {"inputs": empty, "svbrdf": data.synthetic_svbrdf_batch(args.batch, args.size, dev, gen)}
        epoch = 0
        while True:
            if sampler is not None and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            for b in loader:
                yield b
            epoch += 1

    phases = ("data_wait", "upload_synthesis", "forward", "loss", "backward", "optimizer")
    phase_ms = {k: [] for k in phases}
    timing = args.phase_times and on_gpu

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    # after the timed region, DDP runs only: the same step WITH the bucketed all-reduce (DDP's hooks launch a bucket's
    # all-reduce on RCCL's stream as soon as its gradients exist, overlapping the rest of the backward) and WITHOUT it,
    # alternating, HIP events around loss.backward() -- what the all-reduce of ~320 MB costs a step beyond the backward it
    # hides behind.  DDP decides whether a backward synchronises when the FORWARD runs (require_backward_grad_sync is read
    # in DistributedDataParallel.forward), so a no_sync probe step has its forward, its loss and its backward inside
    # model.no_sync() (round 4 entered it around the backward only and measured the all-reduce twice).  Each probe step
    # also checks what it claims: a checksum of the first parameter's gradient is gathered from every rank -- equal on all
    # ranks after a synchronised step, rank-local after a no_sync step.  No optimizer step in these.
    probe_steps = 8 if (grouped and (on_gpu or args.ddp_probe) and not verify and not args.no_ddp_probe) else 0
    end_timed = total_steps
    total_steps += probe_steps
    probe_ms = {True: [], False: []}
    probe_sums = {True: [], False: []}          # per probe step: the gradient checksum of every rank
    import contextlib
    losses_seen, t0, elapsed, it = [], None, None, batches()
    for step in range(total_steps):
        if step == calib_steps + args.warmup:
            if on_gpu:
                torch.cuda.synchronize(dev)
            if grouped:
                barrier()
            t0 = time.perf_counter()
        if step == end_timed:
            if on_gpu:
                torch.cuda.synchronize(dev)
            if grouped:
                barrier()
            elapsed = time.perf_counter() - t0
        probing = step >= end_timed
        synced = not (probing and (step - end_timed) % 2 == 1)
        t_wait = time.perf_counter()
        batch = next(it)
        t_wait = time.perf_counter() - t_wait
        marks = [mark()] if timing else None
        batch = data.decode_uint8_batch(batch, dev, is_linear=args.linear_input)    # 8-bit transport: decoded on the device
        svbrdf = batch["svbrdf"].to(dev, non_blocking=True)
        stored = batch["inputs"].to(dev, non_blocking=True)
        if on_gpu and not verify:
            svbrdf = data.apply_mixing(svbrdf, batch)                            # K4: material mixing, whole batch
            photos = data.complete_inputs(stored, svbrdf, args.views)           # K1: missing photos, whole batch
        else:                                                                    # CPU plumbing runs: constant photos
            photos = torch.cat((stored, svbrdf[:, None, 3:6].expand(-1, max(args.views - stored.shape[1], 0), -1, -1, -1)), 1)
        net_in = photos if args.model == "multi" else photos[:, 0]
        optimizer.zero_grad(set_to_none=True)
        if timing:
            marks.append(mark())
        if auto:
            if step < 4:
                forward_flag = step < 2
            elif step == 4 and len(calib) == 2:
                if grouped:     # one choice for the whole job: the slowest rank's timing of each setting decides
                    both = torch.tensor([calib[True], calib[False]], dtype=torch.float64, device=dev if nccl else "cpu")
                    dist.all_reduce(both, op=dist.ReduceOp.MAX)
                    calib = {True: float(both[0]), False: float(both[1])}
                forward_flag = calib[True] <= calib[False]
            probe = (mark(), None) if step in (1, 3) else None
        if hybrid:
            torch.backends.cudnn.deterministic = forward_flag
        with (contextlib.nullcontext() if synced else model.no_sync()):      # forward AND backward: see the probe note above
            out = model(net_in)
            if auto and probe is not None:
                probe = (probe[0], mark())
                torch.cuda.synchronize(dev)
                calib[forward_flag] = probe[0].elapsed_time(probe[1])
            if hybrid:      # backward always takes MIOpen's immediate-mode choice
                torch.backends.cudnn.deterministic = False
            if timing:
                marks.append(mark())
            if verify and args.loss != "l1":                                         # the scenes of the lower ranks' items
                if rank > 0:
                    loss_fn.rendering_loss.sample_scene_table(rank * args.batch)
            loss = loss_fn(out, svbrdf)
            if verify and args.loss != "l1":                                         # ... and of the higher ranks' items
                if rank < world - 1:
                    loss_fn.rendering_loss.sample_scene_table((world - 1 - rank) * args.batch)
            if timing:
                marks.append(mark())
            if probing:
                bw0 = mark() if on_gpu else time.perf_counter()
            loss.backward()
        if probing:
            if on_gpu:
                bw1 = mark()
                torch.cuda.synchronize(dev)
                ms = bw0.elapsed_time(bw1)
            else:
                ms = 1e3 * (time.perf_counter() - bw0)
            g0 = next(p.grad for p in net.parameters() if p.grad is not None)
            mine = g0.detach().double().abs().sum().reshape(1)
            if on_gpu and not nccl:                      # gloo moves host tensors
                mine = mine.cpu()
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            if step - end_timed >= 2:                    # the first pair warms the no_sync path up
                probe_ms[synced].append(ms)
                probe_sums[synced].append([float(e.item()) for e in every])
            optimizer.zero_grad(set_to_none=True)
            continue
        if verify and step == 0:
            g_loss = distributed.global_mean(loss.detach() if (nccl or not grouped) else loss.detach().cpu()).item()
            if rank == 0:
                import numpy as np
                flat = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])
                np.savez(verify, grad=flat.cpu().numpy(), loss=np.float64(g_loss), world=world, per_rank_batch=args.batch)
        if timing:
            marks.append(mark())
        optimizer.step()
        if hybrid:
            torch.backends.cudnn.deterministic = True
        losses_seen.append(loss.detach())
        if timing:
            marks.append(mark())
            torch.cuda.synchronize(dev)
            if step >= calib_steps + args.warmup:
                phase_ms["data_wait"].append(1e3 * t_wait)
                for name, a, b in zip(phases[1:], marks[:-1], marks[1:]):
                    phase_ms[name].append(a.elapsed_time(b))
    if on_gpu:
        torch.cuda.synchronize(dev)
    if grouped:
        barrier()
    if elapsed is None:
        elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    per_rank = {"elapsed_s": [own_elapsed], "cpus": [placement["cpus"]], "numa_node": [placement["numa_node"]],
                "cpu_binding": [placement["source"]], "pci_crosscheck": [placement.get("pci_crosscheck")]}
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, {"elapsed_s": own_elapsed, "cpus": placement["cpus"], "numa_node": placement["numa_node"],
                                       "cpu_binding": placement["source"], "pci_crosscheck": placement.get("pci_crosscheck"),
                                       "backward_ms": {("with_allreduce" if k else "no_sync"): (sum(v) / len(v) if v else None)
                                                       for k, v in probe_ms.items()}})
        per_rank = {k: [e[k] for e in every] for k in every[0]}
    per_rank["ms_per_step"] = [1e3 * e / args.steps for e in per_rank["elapsed_s"]]
    per_rank["patches_per_s"] = [args.batch * args.steps / e for e in per_rank["elapsed_s"]]
    vals = torch.stack(losses_seen).float()
    if grouped and not nccl:
        vals = vals.cpu()
    first, last = vals[: max(1, len(vals) // 4)].mean(), vals[-max(1, len(vals) // 4):].mean()
    first, last = distributed.global_mean(first).item(), distributed.global_mean(last).item()
    result = {"metric": "end-to-end training patches/s (U-Net + %s loss)" % args.loss,
              "value": world * args.batch * args.steps / elapsed, "unit": "patches/s", "n_gpus": world,
              "per_gpu_value": args.batch * args.steps / elapsed,
              "ranks_seen": ranks_seen, "per_rank": per_rank,
              "process_group": ("%s, world size %d, DistributedDataParallel" % (backend, dist.get_world_size())) if grouped else None,
              "ms_per_step": 1e3 * elapsed / args.steps, "steps": args.steps, "warmup": args.warmup,
              "loss_first_quarter": first, "loss_last_quarter": last,
              "config": {"model": args.model, "views": args.views, "size": args.size, "per_gpu_batch": args.batch,
                         "scenes": args.random_scenes + args.specular_scenes, "fused_head": bool(args.fused_head),
                         "data": ("synthetic (device)" if device_source else args.data) if args.data.startswith("synthetic") else "tiled-png",
                         "workers": args.workers, "conv_mode": conv_mode, "channels_last": bool(args.channels_last),
                         "miopen_cache": training.miopen_cache_identity(miopen_cache)}}
    if len(vals) <= 64:         # short runs (tests): every step's own loss, calibration and warm-up steps included
        result["loss_per_step"] = [float(v) for v in vals.tolist()]
        result["untimed_leading_steps"] = calib_steps + args.warmup
    if probe_steps:
        w, n = probe_ms[True], probe_ms[False]

        def spread(sums):       # largest relative difference between the ranks' gradient checksums of one step
            return max((max(v) - min(v)) / max(abs(max(v)), 1e-300) for v in sums) if sums else None
        result["ddp_backward_probe"] = {
            "backward_ms_with_allreduce": sum(w) / len(w), "backward_ms_no_sync": sum(n) / len(n), "samples_each": len(w),
            "grad_checksum_spread_over_ranks_synced": spread(probe_sums[True]),
            "grad_checksum_spread_over_ranks_no_sync": spread(probe_sums[False]),
            "no_sync_left_gradients_rank_local": (spread(probe_sums[False]) > 1e-9 and spread(probe_sums[True]) <= 1e-9)
                                                 if world > 1 else None,
            "note": "rank 0, untimed steps after the timed region, %s around loss.backward(): DDP's bucketed "
                    "all-reduce (gradient_as_bucket_view, 25 MB buckets) against a step whose forward and backward both ran "
                    "inside model.no_sync(); the difference is what the all-reduce of the U-Net's gradients costs a step "
                    "beyond the backward it overlaps with.  grad_checksum_spread_*: sum |grad| of the first parameter, "
                    "gathered from every rank per probe step -- 0 after a synchronised step, > 0 after a no_sync step (the "
                    "ranks hold different shards), which is the evidence that the second figure is a backward WITHOUT the "
                    "all-reduce (with one rank there is nothing to compare)" % ("HIP events" if on_gpu else "the host clock")}
    if auto:
        result["config"]["conv_forward"] = {"chosen": "reference flags" if forward_flag else "immediate mode",
                                            "calibration_ms": {("reference flags" if k else "immediate mode"): v for k, v in calib.items()}}
    if timing:
        result["phase_ms_mean"] = {k: sum(v) / max(1, len(v)) for k, v in phase_ms.items()}
        result["phase_note"] = "per-step means over the timed steps; one device synchronisation per step (not a throughput run)"
    if rank == 0:
        if args.save:
            # the reference's checkpoint.tar layout (persistence.py:52-69) with the REFERENCE's parameter names, so that
            # its Checkpoint.restore_model_state loads the weights into its own SingleViewModel / MultiViewModel.  The
            # optimizer state indexes parameters by position, which differs between the two module trees: it is kept
            # under its own key instead of one the reference would load into the wrong slots.
            torch.save({"model_type": args.model, "use_coords": not args.no_coords, "epoch": 0,
                        "model_state_dict": models.convert_to_reference_state_dict(net.state_dict()),
                        "optimizer_state_dict_amd": optimizer.state_dict(), "steps": steps_done + total_steps}, args.save)
        print(json.dumps(result), flush=True)
    if grouped:
        dist.destroy_process_group()
    return result


def main(argv=None):
    args = parse_args(argv)
    from svbrdf_estimation_amd import launch
    if args.gpus > 1 and not launch.launched_as_rank():
        # one plain process asked for N GPUs: become the parent of N fresh rank processes (no GPU call has
        # happened in this process and none will)
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else argv, args.gpus))
    return run(args)


if __name__ == "__main__":
    main()
