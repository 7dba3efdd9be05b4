"""Env sharding and the final metric reduce on two gloo ranks (the N>1 path of bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from semantic_search_amd import sharding


def _episodes_for(env):
    rng = np.random.default_rng(env)
    return [dict(success=float(rng.random() < 0.7), spl=float(rng.random()), length=float(rng.integers(10, 500)),
                 fp_found=float(rng.random() < 0.1), fn_found=float(rng.random() < 0.1),
                 fp_detections=float(rng.integers(0, 4)), fn_detections=float(rng.integers(0, 2)),
                 goal=int(rng.integers(0, 6))) for _ in range(3)]


def _worker(rank, world, port, n_envs, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.shard_envs(n_envs, rank, world)
    eps = [ep for env in mine for ep in _episodes_for(env)]
    res = sharding.reduce_metrics(sharding.pack_metrics(eps))
    slowest = sharding.max_over_ranks(1.0 + rank)
    torch.save({"res": res, "slowest": slowest, "mine": mine}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shards_cover_every_env_once():
    for n, w in ((16, 1), (16, 2), (17, 4), (1024, 8), (3, 8)):
        seen = [e for r in range(w) for e in sharding.shard_envs(n, r, w)]
        assert seen == list(range(n))
        sizes = [len(sharding.shard_envs(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_reduce_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_envs, world = 7, 2
    mp.spawn(_worker, args=(world, port, n_envs, str(tmp_path)), nprocs=world, join=True)
    everything = [ep for env in range(n_envs) for ep in _episodes_for(env)]
    want = sharding.reduce_metrics(sharding.pack_metrics(everything))
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=False)
        assert got["slowest"] == 2.0
        assert got["res"]["n_episodes"] == want["n_episodes"] == 21.0
        for k in sharding.METRIC_FIELDS[1:]:
            assert abs(got["res"][k] - want[k]) < 1e-12
        assert np.allclose(got["res"]["per_goal_success"], want["per_goal_success"], equal_nan=True)


def _bench(*argv, **env):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    full.update(env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=full,
                          capture_output=True, text=True, timeout=300)


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks, they meet over
    gloo, reduce, and rank 0 prints one line that says two ranks took part."""
    import json
    res = _bench("--gpus", "2", "--rehearse", "--steps", "5", "--envs-per-gpu", "3", SSMAP_BENCH_BACKEND="gloo")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["frames"] == 2 * 5 * 3
    assert out["slowest_rank_seconds"] == 2.0


def test_bench_refuses_more_ranks_than_gpus():
    """RCCL needs one device per rank: asking for more GPUs than are visible fails loudly instead
    of printing a one-GPU number."""
    import torch
    n = torch.cuda.device_count()
    res = _bench("--gpus", str(n + 2), "--no-cpu-baseline")
    assert res.returncode != 0
    assert "visible" in res.stderr and not any(ln.startswith("{") for ln in res.stdout.splitlines())


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_binds_local_rank_to_its_device_before_anything_touches_a_gpu(monkeypatch):
    """One rank per GPU: rank r of a node must make cuda:r current before the process group, the
    engine or any tensor exists (a rank that initialises device 0 first leaves a context there and
    RCCL then sees two ranks on one device)."""
    import inspect
    bench = _load_bench()
    calls = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("set_device", str(d))))
    for lr in (0, 5, 7):
        dev = bench.bind_device(lr, 8, "nccl")
        assert str(dev) == f"cuda:{lr}" and calls[-1] == ("set_device", f"cuda:{lr}")
    # more ranks than devices is refused for RCCL, allowed (shared devices) for a gloo rehearsal
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    try:
        bench.bind_device(3, 4, "nccl")
        assert False, "four RCCL ranks on two GPUs must be refused"
    except SystemExit:
        pass
    assert str(bench.bind_device(3, 4, "gloo")) == "cuda:1"
    # order inside main(): device binding, then the process group, then the engine
    src = inspect.getsource(bench.main)
    at = [src.index(tok) for tok in ("bind_device(", "init_process_group(", "measure(")]
    assert at == sorted(at)
    assert "device_id=device" in src, "the RCCL process group is bound to this rank's device"
    # measure() seeds the envs of rank r with r * E + e and builds everything on `device`
    msrc = inspect.getsource(bench.measure)
    assert "synth_frames(E, C, rank * E, device)" in msrc and "device=device" in msrc


def test_bench_frames_depend_only_on_the_global_env_index():
    """Env e of rank r is frame number r * E + e of the job whatever the rank count: two ranks of
    two envs see the frames one rank of four envs sees."""
    bench = _load_bench()
    cpu = torch.device("cpu")
    whole = bench.synth_frames(3, 1, 0, cpu)
    part = bench.synth_frames(1, 1, 2, cpu)     # rank 2 of a job with one env per rank
    for a, b in zip(whole, part):
        assert torch.equal(a[2], b[0])
    assert not torch.equal(whole[0][0], whole[0][1])
