#!/usr/bin/env python3
"""bench.py -- points/sec fwd+bwd of pointnet2_sem_seg on 4096-point blocks (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training step (forward, nll_loss, backward, gradient all-reduce when N>1,
Adam) on a per-GPU batch of 16 synthetic 4096x9 blocks already resident in HBM (BASELINE
configs[1]).  Rank 0 prints ONE JSON line; `roofline` prices the query_ball_point+group kernel
of SA1 (the kernel the north_star names) against the HBM roof; `cpu_baseline` is the CPU oracle
timed on the host cores on a bounded sample (N=1 only)."""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
BLOCK_POINTS = 4096
PER_GPU_BATCH = 16
CHANNELS = 9
NUM_CLASSES = 18


def ball_group_algorithmic_bytes(B, N, S, K, D):
    """read xyz + new_xyz + feats once, write int64 idx + fp32 grouped once."""
    return B * (N * 3 * 4 + S * 3 * 4 + N * D * 4 + S * K * 8 + S * K * (3 + D) * 4)


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=14.0):
    """The CPU oracle (oracle/: C index ops + torch CPU network), fwd+bwd+Adam on the same synthetic blocks, timed as
    SURVEY.md 8(d) asks: one warm-up, then the MEDIAN of the timed repetitions -- at least 3 full B=16 steps (more while
    the budget lasts, at most 5), 5 B=1 steps (BASELINE configs[0]), 7 runs each of FPS alone and of
    query_ball_point + group alone; thread count and CPU model stated."""
    from khairil_tum_facade_semantic_segmentation_amd import synth
    from oracle import pn2_oracle as orc
    orc.build()
    threads = torch.get_num_threads()
    orc.set_num_threads(threads)
    filled = synth.fill_state_dict(orc.state_shapes(NUM_CLASSES, CHANNELS - 6))
    net = orc.OracleNet(filled)
    opt = orc.make_adam(net.parameters())
    blocks, labels, starts, _ = synth.draw_case(synth.BENCH_SEED, PER_GPU_BATCH, BLOCK_POINTS, CHANNELS, "cube",
                                                NUM_CLASSES)
    cf = np.ascontiguousarray(blocks.transpose(0, 2, 1))

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    net.train_step(cf, labels, starts, opt)                                    # warm-up at the timed size
    t_start = time.perf_counter()
    steps = []
    while len(steps) < 3 or (len(steps) < 5 and time.perf_counter() - t_start < seconds_budget):
        steps.append(timed(lambda: net.train_step(cf, labels, starts, opt)))
    dt = float(np.median(steps))
    out = {"value": PER_GPU_BATCH * BLOCK_POINTS / dt, "unit": "points/s", "cores": threads, "kind": "port",
           "cpu_model": host_cpu_model(),
           "sample": "median of %d fwd+bwd+Adam steps of %dx%dx%d cube blocks after 1 warm-up (%.2f s each, min %.2f max %.2f); the "
                     "builder's oracle: loop-form C index ops under OpenMP (FPS, ball query with early exit, grouping, 3-NN) + torch CPU "
                     "convolutions / autograd / Adam -- NOT the reference's sort-based torch composition, whose ball query alone took "
                     "0.4-1.0 s per call in the survey container (SURVEY.md 6): a stronger CPU baseline than the reference's own path"
                     % (len(steps), PER_GPU_BATCH, BLOCK_POINTS, CHANNELS, dt, min(steps), max(steps)),
           "step_seconds": steps}
    # configs[0]: one 4096x9 block, one fwd+bwd(+Adam) step
    one = [s[:1] for s in starts]
    net.train_step(cf[:1], labels[:1], one, opt)
    b1 = [timed(lambda: net.train_step(cf[:1], labels[:1], one, opt)) for _ in range(5)]
    out["b1_step_points_per_s"] = BLOCK_POINTS / float(np.median(b1))
    # the hot path alone, B = 16: FPS 4096 -> 1024, then query_ball_point + grouping (r = 0.1, K = 32, D = 9)
    xyz = np.ascontiguousarray(blocks[:, :, :3])
    fps = orc.farthest_point_sample(xyz, 1024, starts[0])                      # warm-up (threads, page faults)
    t_fps = [timed(lambda: orc.farthest_point_sample(xyz, 1024, starts[0])) for _ in range(7)]
    cxyz = orc.index_points(xyz, fps)

    def ball():
        idx = orc.query_ball_point(0.1, 32, xyz, cxyz)
        orc.group_points(xyz, cxyz, blocks, idx)
    ball()
    t_ball = [timed(ball) for _ in range(7)]
    out["fps_input_points_per_s"] = PER_GPU_BATCH * BLOCK_POINTS / float(np.median(t_fps))
    out["ball_query_group_input_points_per_s"] = PER_GPU_BATCH * BLOCK_POINTS / float(np.median(t_ball))
    out["fps_seconds_min_median_max"] = [min(t_fps), float(np.median(t_fps)), max(t_fps)]
    out["ball_seconds_min_median_max"] = [min(t_ball), float(np.median(t_ball)), max(t_ball)]
    return out


def active_switches():
    """PN2_* environment switches in effect (A/B and lab switches change what a line measures: they travel with it)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("PN2_") and k != "PN2_BENCH_DRY_LAUNCH"}


def kernel_source_hash(names=("pn2_ball_grid.hip", "pn2_common.h")):
    """sha256 over the translation unit of the kernel the roofline prices (pn2_ball_grid.hip and the one package header it
    includes): a PMC traffic figure is only quoted while it was measured on exactly these sources."""
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(REPO, "khairil_tum-facade_semantic_segmentation_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def measured_traffic():
    """HBM bytes per launch of the operator-level kernel from the newest committed PMC record (profiles/rNN/ball_query_pmc.json:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, FETCH doubled per the gfx950 correction) whose
    source hash equals the hash of the kernel's sources in this tree; None when no record matches."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "ball_query_pmc.json")), reverse=True):
        try:
            with open(path) as fh:
                pmc = json.load(fh)
        except (OSError, ValueError):
            continue
        if pmc.get("source_sha256") == kernel_source_hash():
            return pmc.get("traffic_bytes_per_launch"), os.path.relpath(path, REPO)
    return None, None


def run_control(args, dev):
    """BASELINE configs[4]: pointnet_sem_seg (plain PointNet, no set abstraction) on the same 16 x 4096 x 9 blocks --
    the pointwise-MLP-only control: fwd + backward + Adam per step, priced against the fp32 MFMA peak (its GEMMs)
    and against the HBM roof (its activation traffic)."""
    from khairil_tum_facade_semantic_segmentation_amd import synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet_sem_seg as P
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, PER_GPU_BATCH, BLOCK_POINTS, CHANNELS, args.kind, NUM_CLASSES)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(labels).to(dev).view(-1)
    model = P.get_model(NUM_CLASSES, CHANNELS - 6)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev).train()
    crit = P.get_loss()
    cw = torch.ones(NUM_CLASSES, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        logp, tf = model(x)
        loss = crit(logp.reshape(-1, NUM_CLASSES), y, tf, cw)
        loss.backward()
        opt.step()
        return loss
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss).item()
    pts = PER_GPU_BATCH * BLOCK_POINTS * args.steps
    macs = P.macs_per_point(CHANNELS, NUM_CLASSES)
    flops = 3 * 2 * macs * pts                                       # forward + two backward products per layer
    # activation traffic: every layer's raw output (fp32) is written once, read by the next layer and by the backward,
    # and its gradient written and read once: 5 passes over sum(Co) floats per point
    widths = (64 + 128 + 1024) * 2 + 64 + 128 + 1024 + 512 + 256 + 128 + NUM_CLASSES
    act_bytes = 5 * 4 * widths * pts
    return {"metric": "points/sec fwd+bwd, 4096-pt blocks, pointnet_sem_seg (pointwise-MLP control)", "value": pts / dt,
            "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pointnet_sem_seg fwd+bwd+Adam, batch=16x4096x9 synthetic %s blocks (BASELINE configs[4])" % args.kind,
                       "global_batch": PER_GPU_BATCH, "points_per_block": BLOCK_POINTS, "parallelism": "dp1"},
            "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": flops / dt / 1e12 / 157.3, "traffic": None, "macs_per_point": macs,
                         "activation_gbs": act_bytes / dt / 1e9, "activation_frac_of_hbm": act_bytes / dt / 1e9 / HBM_PEAK_GBS}}


def run_drop_in(args, dev):
    """BASELINE configs[1] through the reference's unchanged wiring on the drop-in operator surface (one GPU)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from dropin_wiring import build, loss_fn
    from khairil_tum_facade_semantic_segmentation_amd import ops, synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, PER_GPU_BATCH, BLOCK_POINTS, CHANNELS, args.kind, NUM_CLASSES)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(labels).to(dev).view(-1)
    model = build(U, NUM_CLASSES, CHANNELS - 6)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev).train()
    cw = torch.ones(NUM_CLASSES, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)

    def step():
        opt.zero_grad()
        pred, _ = model(x)
        loss = loss_fn(pred.contiguous().view(-1, NUM_CLASSES), y, cw)
        loss.backward()
        opt.step()
        return loss
    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    ops.check_errors()
    assert torch.isfinite(loss).item()
    pts = PER_GPU_BATCH * BLOCK_POINTS * args.steps
    from khairil_tum_facade_semantic_segmentation_amd import graphed
    return {"metric": "points/sec fwd+bwd, 4096-pt blocks, pointnet2_sem_seg (drop-in mode: the reference's wiring on the HIP pointnet2_utils)",
            "module_graphs": dict(graphed.stats, enabled=graphed.ENABLED),
            "value": pts / dt, "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "switches": active_switches(),
            "config": {"workload": "pointnet2_sem_seg fwd+bwd+Adam through tests/dropin_wiring.py (channel-first module calls, torch head / "
                                   "nll_loss / Adam; the modules replay their own captured graphs unless PN2_MODULE_GRAPHS=0), batch=16x4096x%d synthetic %s blocks (BASELINE configs[1])" % (CHANNELS, args.kind),
                       "global_batch": PER_GPU_BATCH, "points_per_block": BLOCK_POINTS, "parallelism": "dp1"}}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run as
    fresh child processes -- this parent has not touched the GPU -- and pass their output and exit code through."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--kind", default="cube", choices=("cube", "facade"))
    ap.add_argument("--model", default="pointnet2_sem_seg", choices=("pointnet2_sem_seg", "pointnet_sem_seg"),
                    help="pointnet_sem_seg = the plain-PointNet control of BASELINE configs[4] (pointwise MLPs only, one GPU)")
    ap.add_argument("--rgb-off", action="store_true",
                    help="BASELINE configs[3]: the geometry-only 4096x6 blocks of the reference's --RGB_OFF (steps only; the roofline leg keeps the north_star shape D = 9)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--drop-in", action="store_true",
                    help="time the DROP-IN mode: the reference's own wiring (tests/dropin_wiring.py: channel-first module calls, torch "
                         "head, F.nll_loss, torch.optim.Adam, eager launches) on the package's models.pointnet2_utils -- what a maintainer "
                         "gets who swaps only pointnet2_utils.py; none of the package's own fast-path wiring")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="after the K timed steps, replay the step for about this many seconds and report sustained_ms_per_step (0: skip)")
    ap.add_argument("--batch-sweep", action="store_true",
                    help="secondary record: the step at 32 and 64 blocks per GPU as well (the headline stays 16, BASELINE configs[1])")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="compute each batch's FPS/ball-query/3-NN pyramid inside its own step instead of one step ahead on a side stream")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                       # before anything here touches the GPU
    if world != args.gpus:
        args.gpus = world
    if os.environ.get("PN2_BENCH_DRY_LAUNCH") == "1":     # launcher rehearsal (tests/test_bench_launcher_cpu.py): no GPU work
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world": world}), flush=True)
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # lazy communicator (no device_id): it must not exist before the graphs are captured (train.prepare)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from khairil_tum_facade_semantic_segmentation_amd import _lib, ops, synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    _lib.load()
    if args.model == "pointnet_sem_seg":
        if rank == 0:
            print(json.dumps(run_control(args, dev)), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    if args.drop_in:
        if rank == 0:
            print(json.dumps(run_drop_in(args, dev)), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    # synthetic blocks of this rank (weak scaling: 16 blocks per GPU), resident in HBM
    step_channels = 6 if args.rgb_off else CHANNELS
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED + rank, PER_GPU_BATCH, BLOCK_POINTS, step_channels, args.kind,
                                           NUM_CLASSES)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)     # [B,C,N]
    y = torch.from_numpy(labels).to(dev)
    model = M.get_model(NUM_CLASSES, step_channels - 6)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev)
    if os.environ.get("PN2_LAB_MAIN_HIGH_PRIORITY", "0") == "1":      # lab: the step on a high-priority HIP stream
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    trainer = SemSegTrainer(model, class_weight=torch.ones(NUM_CLASSES, device=dev), graphs=not args.no_graphs,
                            prefetch_geometry=not args.no_prefetch)
    if not args.no_graphs:
        # capture now (setup, not a step): before the first collective creates the RCCL communicator, and so that the warm-up
        # steps below are REPLAYS -- the first launch of a graph uploads it (0.3-0.5 ms once per graph: the two alternating step
        # graphs and their geometry graphs would otherwise see their first launch inside the K timed steps)
        trainer.prepare(x, y)
    trainer.broadcast_parameters()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 4 if not args.no_graphs else 0)):   # every graph replayed at least twice before timing
        trainer.step(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x, y)
    barrier()
    dt = time.perf_counter() - t0
    ops.check_errors()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item(), "training step produced a non-finite loss"
    # sustained leg (outside the driver's K timed steps): the same step replayed for >= args.sustain seconds -- clocks and
    # power settle, and a busy sampler sees the GPU working; max over ranks like the headline
    sustained = None
    if args.sustain > 0:
        n_sus = max(int(args.sustain / max(dt / args.steps, 1e-4)), args.steps)
        barrier()
        ts = time.perf_counter()
        for _ in range(n_sus):
            loss = trainer.step(x, y)
        barrier()
        dts = time.perf_counter() - ts
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sustained = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3,
                     "points_per_s": world * PER_GPU_BATCH * BLOCK_POINTS * n_sus / dts}
        assert torch.isfinite(loss).item()
    # evidence that the collective ran over N ranks: backend, world size and the measured all-reduce of the packed gradient
    dp_info = None
    if use_dist:
        flat = torch.zeros(trainer.grads.numel, dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_reduce(flat)
        torch.cuda.synchronize(dev)
        ta = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat)
        torch.cuda.synchronize(dev)
        ar_us = (time.perf_counter() - ta) / 20 * 1e6
        t = torch.tensor([ar_us], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dp_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "gradient_bytes": trainer.grads.numel * 4,
                   "all_reduce_us": float(t.item()), "collectives_per_step": 1}

    # secondary record: larger per-GPU batches (fresh models and trainers; the 288 GB of HBM hold ~100x the headline batch)
    sweep = None
    if args.batch_sweep and world == 1 and not args.no_graphs:
        sweep = {}
        for bsz in (32, 64):
            blk, lab, _, _ = synth.draw_case(synth.BENCH_SEED + rank, bsz, BLOCK_POINTS, step_channels, args.kind, NUM_CLASSES)
            xb = torch.from_numpy(np.ascontiguousarray(blk.transpose(0, 2, 1))).to(dev)
            yb = torch.from_numpy(lab).to(dev)
            mb = M.get_model(NUM_CLASSES, step_channels - 6)
            mb.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
            tb = SemSegTrainer(mb.to(dev), class_weight=torch.ones(NUM_CLASSES, device=dev), graphs=True, prefetch_geometry=not args.no_prefetch)
            for _ in range(6):
                tb.step(xb, yb)
            torch.cuda.synchronize(dev)
            tsw = time.perf_counter()
            nsw = 30
            for _ in range(nsw):
                tb.step(xb, yb)
            torch.cuda.synchronize(dev)
            dsw = time.perf_counter() - tsw
            sweep[str(bsz)] = {"ms_per_step": dsw / nsw * 1e3, "points_per_s": bsz * BLOCK_POINTS * nsw / dsw}
            del tb, mb, xb, yb
    # roofline of the north_star kernel: SA1 query_ball_point+group at B=16, N=4096, S=1024, K=32, D=9, same resident
    # inputs as the timed steps, priced at OPERATOR level: from (xyz, new_xyz, feats) to (idx, grouped), everything a
    # caller with nothing prepared must launch = ONE call of pn2_ball_query_group (one launch of the cell-pruned
    # kernel; what ops.ball_query_group / models.pointnet2_utils.sample_and_group run).  `kernel_ms` = its average
    # duration over `reps` calls running back to back in a captured graph, between HIP events on the launch stream.
    # Beside it: the planned pair (pn2_ball_plan + pn2_ball_query_group_planned) whose query launch alone is the
    # fastest kernel of the library but whose plan a single-use caller pays in full, on both synthetic distributions.
    rblocks, _, _, _ = synth.draw_case(synth.BENCH_SEED + rank, PER_GPU_BATCH, BLOCK_POINTS, CHANNELS, args.kind, NUM_CLASSES)
    pts = torch.from_numpy(rblocks).to(dev)
    xyz = pts[:, :, :3].contiguous()
    start = torch.zeros(PER_GPU_BATCH, dtype=torch.long, device=dev)
    _, new_xyz, plan = ops.farthest_point_sample_plan(xyz, 1024, 0.1, CHANNELS, start)
    plan.pack_rows(xyz, pts)
    lib = _lib.load()
    idx_buf = torch.empty((PER_GPU_BATCH, 1024, 32), dtype=torch.int64, device=dev)
    grp_buf = torch.empty((PER_GPU_BATCH, 1024, 32, 3 + CHANNELS), dtype=torch.float32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)

    def planned():
        rc = lib.pn2_ball_query_group_planned(0.1, 32, plan.buf.data_ptr(), xyz.data_ptr(), new_xyz.data_ptr(), pts.data_ptr(),
                                              PER_GPU_BATCH, BLOCK_POINTS, 1024, CHANNELS, idx_buf.data_ptr(), grp_buf.data_ptr(),
                                              0, err.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0, rc

    def selfcontained():
        rc = lib.pn2_ball_query_group(0.1, 32, xyz.data_ptr(), new_xyz.data_ptr(), pts.data_ptr(), PER_GPU_BATCH, BLOCK_POINTS,
                                      1024, CHANNELS, idx_buf.data_ptr(), grp_buf.data_ptr(), 0, err.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0, rc

    def producers():
        rc = lib.pn2_ball_plan(0.1, xyz.data_ptr(), new_xyz.data_ptr(), pts.data_ptr(), PER_GPU_BATCH, BLOCK_POINTS, 1024,
                               CHANNELS, plan.buf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0, rc

    def fps(with_plan):
        o = torch.empty((PER_GPU_BATCH, 1024), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        if with_plan:
            rc = lib.pn2_farthest_point_sample_plan(xyz.data_ptr(), PER_GPU_BATCH, BLOCK_POINTS, 1024, start.data_ptr(),
                                                    o.data_ptr(), new_xyz.data_ptr(), 0.1, CHANNELS, plan.buf.data_ptr(),
                                                    err.data_ptr(), st)
        else:
            rc = lib.pn2_farthest_point_sample(xyz.data_ptr(), PER_GPU_BATCH, BLOCK_POINTS, 1024, start.data_ptr(), o.data_ptr(),
                                               new_xyz.data_ptr(), err.data_ptr(), st)
        assert rc == 0, rc

    def back_to_back_ms(fn, reps):
        """average duration of `reps` calls of fn running back to back (one captured graph, so that the host's
        submission rate does not matter), median of 5 replays: what rocprofv3 --kernel-trace reports per launch"""
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph.replay()
        torch.cuda.synchronize(dev)
        times = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            graph.replay()
            b.record()
            torch.cuda.synchronize(dev)
            times.append(a.elapsed_time(b) / reps)
        del graph
        return float(np.median(times))

    reps = 50
    algo = ball_group_algorithmic_bytes(PER_GPU_BATCH, BLOCK_POINTS, 1024, 32, CHANNELS)

    def measure():
        """-> operator ms (pn2_ball_query_group), planned query ms, plan ms (stand-alone producers)"""
        return back_to_back_ms(selfcontained, reps), back_to_back_ms(planned, reps), back_to_back_ms(producers, reps)

    op_ms, k_ms, prod_ms = measure()
    # one event pair per call: includes the launch latency of an idle queue
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        selfcontained()
        b.record()
    torch.cuda.synchronize(dev)
    op_ms_single = float(np.median([a.elapsed_time(b) for a, b in ev]))
    fps_plain_ms = back_to_back_ms(lambda: fps(False), 10)
    # the same on the other synthetic distribution (dense facade slab: most balls truncated at nsample; sparse cube: most
    # balls scanned to the end), beside the figures for the distribution the steps were timed on
    other_kind = "facade" if args.kind == "cube" else "cube"
    oblocks, _, _, _ = synth.draw_case(synth.BENCH_SEED + rank, PER_GPU_BATCH, BLOCK_POINTS, CHANNELS, other_kind, NUM_CLASSES)
    pts = torch.from_numpy(oblocks).to(dev)
    xyz = pts[:, :, :3].contiguous()
    _, new_xyz, plan = ops.farthest_point_sample_plan(xyz, 1024, 0.1, CHANNELS, start)
    plan.pack_rows(xyz, pts)
    op_ms_o, k_ms_o, prod_ms_o = measure()
    assert int(err.item()) == 0

    def frac(ms):
        return algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    achieved = algo / (op_ms * 1e-3) / 1e9
    # HBM traffic of that launch cannot be read from inside this process: it is the rocprofv3 --pmc measurement
    # committed under profiles/ (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), quoted only while the
    # kernel sources are the ones it was measured on
    traffic, traffic_record = measured_traffic()

    if rank == 0:
        total_points = world * PER_GPU_BATCH * BLOCK_POINTS * args.steps
        out = {
            "metric": "points/sec fwd+bwd, 4096-pt blocks, pointnet2_sem_seg",
            "value": total_points / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "switches": active_switches(),
            "config": {"workload": "pointnet2_sem_seg fwd+bwd+Adam, batch=16x4096x%d synthetic %s blocks per GPU, "
                                   "npoint=[1024,256,64,16] nsample=32, 18 classes (BASELINE %s)"
                                   % (step_channels, args.kind, "configs[3], --RGB_OFF" if args.rgb_off else "configs[1]"),
                       "global_batch": world * PER_GPU_BATCH, "points_per_block": BLOCK_POINTS,
                       "parallelism": "dp%d" % world, "graphs": not args.no_graphs, "prefetch_geometry": not args.no_prefetch},
            "roofline": {"bound": "hbm",
                         "kernel": "operator level: pn2_ball_query_group, (xyz, new_xyz, feats) -> (idx, grouped) in one launch of "
                                   "ball_query_group_grid_kernel (SA1: N=4096,S=1024,K=32,D=9,B=16, %s)" % args.kind,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_record": traffic_record, "algorithmic_bytes": algo, "kernel_ms": op_ms, "kernel_ms_single_launch": op_ms_single,
                         "other_distribution": {"kind": other_kind, "kernel_ms": op_ms_o, "frac": frac(op_ms_o)},
                         "planned_pair": {
                             "what": "pn2_ball_plan (binning + row packing, one launch) + pn2_ball_query_group_planned "
                                     "(ball_query_binned_kernel): the query launch alone is the library's fastest ball kernel, a "
                                     "caller that uses a plan once pays both",
                             "query_ms": k_ms, "query_frac": frac(k_ms), "plan_ms": prod_ms, "pair_ms": k_ms + prod_ms,
                             "pair_frac": frac(k_ms + prod_ms),
                             "other_distribution": {"kind": other_kind, "query_ms": k_ms_o, "query_frac": frac(k_ms_o),
                                                    "plan_ms": prod_ms_o, "pair_frac": frac(k_ms_o + prod_ms_o)}},
                         "fps_kernel_ms": fps_plain_ms},
        }
        if sustained is not None:
            out["sustained_ms_per_step"] = sustained["ms_per_step"]
            out["sustained"] = sustained
        if sweep is not None:
            out["batch_sweep"] = sweep
        if dp_info is not None:
            out["data_parallel"] = dp_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
