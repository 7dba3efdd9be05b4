#!/usr/bin/env python3
"""Map-fusion throughput on MI355X: frames/s of the whole per-step fusion path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          (starts N ranks itself, one per GPU, RCCL between them)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of stages 1-5 (softmax/entropy, back-projection, tallest-per-cell,
rule update, found decision) over one batch of E frames per GPU, inputs resident in HBM.
Default workload = BASELINE.json configs[1]: height_map_log_odds, 16 envs per GPU,
640x480 frames, 40 classes, logits in (T = 3.2), 480x480 maps.  Environments are sharded by
GPU with no collective in the inner loop (weak scaling); RCCL carries only the timing
max and the final metric sum.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (prob_aggregate, C, T, map bounds, attrs)
    "log_odds_c40": dict(rule="height_map_log_odds", C=40, T=3.2, span=14.39,
                         attrs=dict(entropy_threshold=0.0171)),
    "avg_prob_c40": dict(rule="height_map_avg_prob", C=40, T=3.2, span=14.39,
                         attrs=dict(entropy_threshold=0.43759)),
    "w2_avg_prob_c150": dict(rule="height_map_w2_avg_prob", C=150, T=2.5, span=14.39,
                             attrs=dict(entropy_threshold=0.4820)),
    "goal_decay_c40": dict(rule="height_map_goal_decay", C=40, T=3.2, span=14.39,
                           filter_goal_prob=0.38158, attrs=dict(goal_decay=0.8931, goal_mark=1.5408)),
    "prob_hv_c40": dict(rule="height_map_prob_hv", C=40, T=3.2, span=28.79,
                        attrs=dict(hv_ratio=0.887, hv_min_views=5, hv_limit_obj_view_dist=4.13219)),
}
H, W = 480, 640
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
# a class count without a streaming kernel of its own: it runs the kernel of the next size up (24 class registers)
# other heads of the reference's EMSANet wrapper (infer.py: 37 classes SUNRGB-D; 81 as the widest of them)
WORKLOADS["log_odds_c37"] = dict(rule="height_map_log_odds", C=37, T=3.2, span=14.39, attrs=dict(entropy_threshold=0.0171))
WORKLOADS["w2_avg_prob_c81"] = dict(rule="height_map_w2_avg_prob", C=81, T=2.5, span=14.39, attrs=dict(entropy_threshold=0.4820))
WORKLOADS["avg_prob_c21"] = dict(rule="height_map_avg_prob", C=21, T=3.2, span=14.39,
                                 attrs=dict(entropy_threshold=0.43759))


def bind_device(local_rank, world, backend):
    """This rank's GPU, made current BEFORE the first call that touches a device (counting devices
    does not): LOCAL_RANK -> cuda:LOCAL_RANK, one rank per GPU over RCCL like the reference's one
    worker per device (eval_objnav.py:100).  With gloo (rehearsals) ranks may share a device."""
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        sys.exit(f"bench.py: {world} ranks but {n_dev} GPUs visible")
    device = torch.device("cuda", local_rank % max(n_dev, 1))
    torch.cuda.set_device(device)
    return device


def synth_frames(n_envs, n_classes, seed0, device):
    """BASELINE frame recipe, one numpy generator per env (seed = global env index)."""
    depth = np.empty((n_envs, H, W), np.float32)
    gt = np.empty((n_envs, H, W), np.int32)
    logits = torch.empty((n_envs, n_classes, H, W), dtype=torch.float32, device=device)
    for e in range(n_envs):
        rng = np.random.default_rng(seed0 + e)
        d = rng.uniform(0.6, 4.9, (H, W)).astype(np.float32)
        bad = rng.random((H, W)) < 0.02
        which = rng.random((H, W)) < 0.5
        d[bad & which] = np.float32(0.5)
        d[bad & ~which] = np.float32(5.0)
        depth[e] = d
        gt[e] = rng.integers(0, 9, (H, W))
        lg = (3.0 * rng.standard_normal((n_classes, H, W), dtype=np.float32)).astype(np.float32)
        logits[e] = torch.from_numpy(lg).to(device)
    return (torch.from_numpy(depth).to(device), logits, torch.from_numpy(gt).to(device))


def make_poses(n_envs, center, rng):
    from tests import scenes
    ag, cam = [], []
    for _ in range(n_envs):
        a, c = scenes.make_pose(center, 0.0, center, rng.uniform(0.0, 2 * np.pi))
        ag.append(a)
        cam.append(c)
    return np.stack(ag), np.stack(cam)


def algorithmic_bytes_per_frame(n_classes, k_updates, prob_rule):
    """Compulsory traffic: every input element once, every touched state element once."""
    n = H * W
    geometry = 4 * n + 2 * 56
    stage1 = 4 * n_classes * n + 3 * 4 * n
    if prob_rule:
        state = 2 * 4 * k_updates * n_classes + 2 * 4 * k_updates * 4
    else:
        state = 2 * 4 * k_updates * 4
    return geometry, stage1, state


def cpu_baseline(workload, n_frames, seed, budget_s=25.0):
    """The CPU oracle (numpy port of the reference path) on the same workload, one core."""
    from oracle import mapper_oracle as mo
    from oracle import stage1_oracle as s1
    from tests import scenes
    torch.set_num_threads(1)
    wl = WORKLOADS[workload]
    lut = scenes.schema_lut(wl["C"])
    m = mo.OracleMapper(num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate=wl["rule"],
                        seg_model_n_classes=wl["C"], filter_goal_prob=wl.get("filter_goal_prob"),
                        schema_lut=lut)
    for k, v in wl["attrs"].items():
        setattr(m, k, v)
    m.reset(np.array([0, 0, 0, wl["span"], 3.0, wl["span"]]))
    rng = np.random.default_rng(seed)
    fr = scenes.uniform_frame(rng, H, W, wl["C"], wl["T"], lut=lut, with_probs=False)
    prng = np.random.default_rng(seed + 1)
    elapsed, done = 0.0, 0
    for i in range(n_frames + 1):
        ag, cam = make_poses(1, wl["span"] / 2, prng)
        depth = fr["depth"].copy()
        t0 = time.perf_counter()
        out = s1.predict_tail(fr["logits"], wl["T"], lut)
        obs = {"depth": depth, "semantic_gt": fr["semantic_gt"], "objectgoal": np.array([0]),
               "agent_position_and_rotation": ag[0], "camera_position_and_rotation": cam[0],
               "semantic": out["semantic"], "semantic_prob": out["semantic_prob"],
               "semantic_unc": out["semantic_unc"], "semantic_max_prob": out["semantic_max_prob"]}
        m.update(obs)
        dt = time.perf_counter() - t0
        if i > 0:   # first frame is warm-up
            elapsed += dt
            done += 1
        if elapsed > budget_s:
            break
    return done / elapsed, done


def _cpu_worker(job):
    workload, n_frames, seed = job
    return cpu_baseline(workload, n_frames, seed)


def cpu_baseline_processes(workload, n_procs, n_frames):
    """The reference's own scaling model on a CPU: one mapper per OS process (eval_objnav.py:144).
    n_procs independent processes run the oracle at the same time; returns the summed frames/s.
    Must run before this process touches the GPU (the workers are forked)."""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(n_procs) as pool:
        res = pool.map(_cpu_worker, [(workload, n_frames, 100 + i) for i in range(n_procs)])
    return sum(r[0] for r in res), sum(r[1] for r in res)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process (which never touches a GPU)
    starts N ranks of itself, one per device like the reference's one worker per env shard
    (eval_objnav.py:100, 144), and passes rank 0's line through."""
    import socket
    import subprocess
    backend = os.environ.get("SSMAP_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if backend == "nccl" and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked for, {have} visible: one rank per GPU over RCCL "
              "needs that many devices", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    return rc


def rehearse(args, rank, world):
    """The multi-rank skeleton of main() without the engine: rendezvous, barrier, max and sum
    reductions, one line from rank 0."""
    import torch.distributed as dist
    backend = os.environ.get("SSMAP_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        sys.exit("bench.py --rehearse runs without GPUs: set SSMAP_BENCH_BACKEND=gloo")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    metric = torch.tensor([float(args.steps * args.envs_per_gpu), 1.0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(metric, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "ranks": int(metric[1].item()),
                          "frames": float(metric[0].item()), "slowest_rank_seconds": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_registers(name):
    """VGPRs / waves per SIMD of a kernel from the committed compiler report."""
    import csv
    for tag in ("r3", "r2"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.csv")) as fh:
                for row in csv.DictReader(fh):
                    if row["kernel"] == name:
                        return {"vgpr": int(row["VGPRs"]), "sgpr": int(row["TotalSGPRs"]),
                                "waves_per_simd": int(row["Occupancy [waves/SIMD]"]),
                                "lds_bytes_per_block": int(row["LDS Size [bytes/block]"]),
                                "source": f"profiles/{tag}_kernel_resources.csv"}
        except (OSError, KeyError, ValueError):
            pass
    return None


def kernel_trace_us(workload, E, kernel_prefix):
    """Average duration of a kernel in the committed rocprofv3 --kernel-trace --stats summary of this
    command (profiles/r3_bench_<workload>_E<E>_kernel_stats.csv), or None."""
    import csv
    for tag in ("r3", "r2"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_bench_{workload}_E{E}_kernel_stats.csv")) as fh:
                for row in csv.DictReader(fh):
                    if row["Name"].replace("void ", "").replace("ssmap::", "").startswith(kernel_prefix):
                        return float(row["AverageNs"]) / 1e3, f"profiles/{tag}_bench_{workload}_E{E}_kernel_stats.csv"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def measure(workload, E, args, rank, world, dist, device, backend):
    """K steps of one workload between barriers (repeated to --min-seconds), then the same K steps
    again with per-stage events.  Returns the numbers of this rank's view; the sums over ranks
    are already reduced."""
    import semantic_search_amd as ssa
    from tests import scenes

    wl = WORKLOADS[workload]
    C = wl["C"]
    span = wl["span"]
    map_hw = ssa.map_size([0, 0, 0, span, 3.0, span], 0.03)
    eng = ssa.MapEngine(E, (H, W), num_semantic_classes=9, depth_hfov=59.0, depth_range=(0.5, 5.0),
                        probablistic_mapping=True, prob_aggregate=wl["rule"], seg_model_n_classes=C,
                        max_map_hw=map_hw, filter_goal_prob=wl.get("filter_goal_prob"), device=device)
    for k, v in wl["attrs"].items():
        setattr(eng.params, k, v)
    eng.set_schema(scenes.schema_lut(C))
    for e in range(E):
        eng.reset_env(e, [0, 0, 0, span, 3.0, span])
    # one generator per env, seeded with the env's index in the whole job (rank * E + e)
    depth, logits, gt = synth_frames(E, C, rank * E, device)
    sem = torch.empty((E, H, W), dtype=torch.int32, device=device)
    unc = torch.empty((E, H, W), dtype=torch.float32, device=device)
    maxp = torch.empty((E, H, W), dtype=torch.float32, device=device)
    goals = np.full(E, 3, dtype=np.int32)
    prng = np.random.default_rng(1000 + rank)
    poses = [make_poses(E, span / 2, prng) for _ in range(256)]   # a new random yaw every step
    counter = [0]

    def one_step():
        ag, cam = poses[counter[0] % len(poses)]
        counter[0] += 1
        eng.step(depth, ag, cam, goals, logits=logits, temperature=wl["T"], semantic=sem,
                 semantic_unc=unc, semantic_max_prob=maxp, semantic_gt=gt)

    red_dev = device if backend == "nccl" else torch.device("cpu")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_block():
        """Exactly --steps steps between two barriers; the slowest rank's time."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        one_step()
    blocks = [timed_block()]
    # the same block again until min-seconds have been measured (every rank sees the same
    # reduced times, so every rank runs the same number of blocks)
    while sum(blocks) < args.min_seconds and len(blocks) < 200:
        blocks.append(timed_block())
    elapsed = float(np.median(blocks))
    found, status, n_upd = eng.results()
    assert not status.any(), "engine reported an index error on a benchmark frame"

    # one more pass of the same K steps with per-stage HIP events on the launch stream
    eng.profile_begin(args.steps)
    for _ in range(args.steps):
        one_step()
    stage_ms, n_prof = eng.profile_read()
    found, status, n_upd = eng.results()

    metric = torch.tensor([float(args.steps * E), float(found.sum()), float(n_upd.sum()), 1.0],
                          dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(metric, op=dist.ReduceOp.SUM)   # the path's only exchange: final metric sum
    out = dict(wl=wl, C=C, E=E, map_hw=map_hw, elapsed=elapsed, blocks=blocks,
               stage_ms=[m / max(n_prof, 1) for m in stage_ms], k_mean=float(n_upd.mean()),
               frames_total=float(metric[0].item()), ranks_seen=int(metric[3].item()))
    eng.close()
    del depth, logits, gt, sem, unc, maxp
    torch.cuda.empty_cache()
    return out


def dominant_kernel(workload, m):
    """Name, algorithmic bytes per launch and event-timed duration of the step's dominant kernel: the
    streaming pass k_stream_* (one launch per step, stage group [2])."""
    wl, C, E = m["wl"], m["C"], m["E"]
    prob_rule = any(t in wl["rule"] for t in ("avg_prob", "log_odds"))
    geo_b, s1_b, state_b = algorithmic_bytes_per_frame(C, m["k_mean"], prob_rule)
    dom_bytes = s1_b * E
    mode = (2 if "log_odds" in wl["rule"] else 1) if prob_rule else 0
    # probability rules: the streaming kernel also updates the owners' map cells (modes 3, 4), so its
    # compulsory bytes include the state traffic and no update kernel runs (SSMAP_DIRECT=0: staged
    # rows and k_update_prob instead)
    direct = prob_rule and os.environ.get("SSMAP_DIRECT", "1") != "0"
    if direct:
        mode += 2
        # per update cell: the fused row read and written (2 * 4 C), label (4) and colour (3) written, and for
        # the average family weight read and written (8) and uncertainty written (4); height map and
        # ground-truth map of the cell are the winner kernel's
        per_cell = 8 * C + 7 + (12 if "avg_prob" in wl["rule"] else 0)
        dom_bytes += per_cell * m["k_mean"] * E
    # third template argument: how the pooled goal confidence is staged (0: every goal pools one
    # class, as the 40-class table does; 2: the SegFormer lists, known at compile time).  Class counts
    # of the reference's heads have a kernel of their own, the others run the next size up.
    exact = C in (8, 33, 37, 40, 53, 81, 150)
    if exact:
        name = f"k_stream_reg<{C}, {mode}, {2 if C == 150 else 0}>"
    elif mode in (0, 3, 4):
        size = next(b for b in (16, 24, 32, 48, 64, 96, 128, 155) if C <= b)
        name = f"k_stream_reg_upto<{size}, {mode}, 0>"
    else:
        name = f"k_stream_lds<{mode}>"
    return name, direct, dom_bytes, (geo_b, s1_b, state_b)


def workload_summary(workload, m, world):
    """The short form of a measurement, for the `extras` block."""
    name, direct, dom_bytes, (geo_b, s1_b, state_b) = dominant_kernel(workload, m)
    fps = m["frames_total"] / m["elapsed"]
    dom_ms = m["stage_ms"][2]
    return {"workload": f"{m['wl']['rule']}, {m['E']} envs per GPU, {m['C']} classes, {m['map_hw'][0]}x{m['map_hw'][1]} map",
            "value": fps, "unit": "frames/s", "ms_per_step": 1e3 * m["elapsed"] / m["steps"],
            "algorithmic_MB_per_frame": (geo_b + s1_b + state_b) / 1e6,
            "whole_path_frac_of_hbm_peak": (geo_b + s1_b + state_b) * (fps / world) / 1e9 / HBM_PEAK_GBS,
            "dominant_kernel": name, "dominant_kernel_event_ms": dom_ms,
            "dominant_kernel_frac_of_hbm_peak": dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_ms > 0 else None,
            "stage_ms_per_step": dict(zip(STAGE_NAMES, m["stage_ms"]))}


STAGE_NAMES = ("geometry", "winners", "stream_stage1", "rule_update", "found")
EXTRA_WORKLOADS = ("avg_prob_c40", "w2_avg_prob_c150", "goal_decay_c40", "prob_hv_c40", "avg_prob_c21")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="log_odds_c40", choices=sorted(WORKLOADS) + ["all"],
                    help="'all' = the default workload as the headline plus every other one in `extras`")
    ap.add_argument("--envs-per-gpu", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extras` block (the other workloads at the same batch size, a short run each)")
    ap.add_argument("--cpu-frames", type=int, default=60)
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the block of --steps steps is repeated until this much time has been measured")
    ap.add_argument("--rehearse", action="store_true",
                    help="no GPU work: ranks only meet, reduce a dummy metric and rank 0 prints a stub line "
                         "(a CPU check of the launch path; needs SSMAP_BENCH_BACKEND=gloo)")
    args = ap.parse_args()
    want_extras = args.workload == "all" or (args.workload == "log_odds_c40" and not args.no_extras
                                             and not args.no_cpu_baseline and args.gpus == 1)
    if args.workload == "all":
        args.workload = "log_odds_c40"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    if args.rehearse:
        return rehearse(args, rank, world)
    cpu_many = None
    if world == 1 and not args.no_cpu_baseline:
        # forked workers: before anything here initialises the GPU
        n_procs = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "8")), os.cpu_count() or 1, 16))
        cpu_many = (n_procs,) + cpu_baseline_processes(args.workload, n_procs, 8)
    dist = None
    # RCCL between the ranks of a node; SSMAP_BENCH_BACKEND=gloo lets several ranks share one GPU
    # (a rehearsal of the multi-rank path on a one-GPU box: RCCL refuses two ranks on one device)
    backend = os.environ.get("SSMAP_BENCH_BACKEND", "nccl")
    device = bind_device(local_rank, world, backend)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    E = args.envs_per_gpu
    m = measure(args.workload, E, args, rank, world, dist, device, backend)
    m["steps"] = args.steps
    extras = {}
    if want_extras and world == 1:
        short = argparse.Namespace(**vars(args))
        short.steps, short.warmup, short.min_seconds = min(args.steps, 100), min(args.warmup, 10), 0.25
        for name in EXTRA_WORKLOADS:
            mx = measure(name, E, short, rank, world, None, device, backend)
            mx["steps"] = short.steps
            extras[name] = workload_summary(name, mx, world)

    if rank == 0:
        wl, C, map_hw = m["wl"], m["C"], m["map_hw"]
        elapsed, blocks, frames_total = m["elapsed"], m["blocks"], m["frames_total"]
        dom_name, direct, dom_bytes, (geo_b, s1_b, state_b) = dominant_kernel(args.workload, m)
        # event interval around the kernel's launch on its stream: the kernel plus the ~3-5 us of one
        # launch boundary and two event records; the committed kernel trace has the kernel alone
        dom_ms = m["stage_ms"][2]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        trace_us, trace_src = kernel_trace_us(args.workload, E, dom_name.split("(")[0])
        traffic = None
        try:   # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                traffic = json.load(fh).get(args.workload, {}).get(f"E{E}", {}).get("k_stream_bytes_per_launch")
        except OSError:
            pass
        line = {
            "metric": "map-fusion frames/sec (1 GPU and 8-GPU node) + % HBM peak, 640x480x40cls",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{wl['rule']}, {E} envs per GPU, 640x480 frames, {C} classes, "
                                   f"logits in (T={wl['T']}), {map_hw[0]}x{map_hw[1]} map",
                       "envs_per_gpu": E, "frames_per_step": E * world,
                       "ranks": m["ranks_seen"], "backend": backend if world > 1 else None,
                       "mean_update_cells_per_frame": m["k_mean"],
                       "algorithmic_MB_per_frame": (geo_b + s1_b + state_b) / 1e6,
                       "whole_path_frac_of_hbm_peak":
                           (geo_b + s1_b + state_b) * (frames_total / world / elapsed) / 1e9 / HBM_PEAK_GBS},
            "timing": {"blocks_of_steps": len(blocks), "seconds_measured": float(sum(blocks)),
                       "ms_per_step_median": 1e3 * elapsed / args.steps,
                       "ms_per_step_min": 1e3 * min(blocks) / args.steps,
                       "ms_per_step_max": 1e3 * max(blocks) / args.steps,
                       "value_from": "median block, slowest rank"},
            "roofline": {"bound": "hbm", "kernel": dom_name + (" (stage 1 softmax/entropy pass fused with the rule update of the owners' map cells)" if direct else " (stage 1 fused softmax/entropy pass)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command, committed (not collected in this run)" if traffic else None,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                         "avg_launch_ms_from": "HIP events recorded on the launch stream around the kernel, live in this run "
                                               "(kernel + one launch boundary + the event records); `achieved` and `frac` use it",
                         "kernel_trace_avg_launch_ms": trace_us / 1e3 if trace_us else None,
                         "kernel_trace_frac": (dom_bytes / (trace_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if trace_us else None,
                         "kernel_trace_source": trace_src,
                         "registers": kernel_registers(dom_name)},
            "stage_ms_per_step": dict(zip(STAGE_NAMES, m["stage_ms"])),
            "stage_ms_note": "a second pass of the same K steps with six event records per step on the launch stream: "
                             "every interval carries one launch boundary and an event record (about 3-5 us), so the "
                             "five intervals add up to more than ms_per_step, which is measured without events",
        }
        if extras:
            line["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            fps, n_done = cpu_baseline(args.workload, args.cpu_frames, seed=0, budget_s=12.0)
            line["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": f"{n_done} frames of the same workload (1 env, stage 1 on "
                                              "torch CPU + numpy oracle), 1 thread",
                                    "all_cores": {"value": cpu_many[1], "unit": "frames/s", "cores": cpu_many[0],
                                                  "sample": f"{cpu_many[2]} frames over {cpu_many[0]} independent "
                                                            "processes, one env each, run at the same time"}}
            if args.workload != "avg_prob_c40":
                # BASELINE configs[0]: height_map_avg_prob, one env, the reference's CPU-runnable case
                fps0, n0 = cpu_baseline("avg_prob_c40", args.cpu_frames, seed=0, budget_s=12.0)
                line["cpu_baseline"]["config0_avg_prob_1env"] = {
                    "value": fps0, "unit": "frames/s", "cores": 1, "kind": "port",
                    "sample": f"{n0} frames of height_map_avg_prob, 1 env, 640x480x40, 480x480 map, 1 thread"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
