"""The N>1 path on CPU: gloo, world_size 2 (the collective is the start-up
model broadcast; streams are independent replicas)."""

import hashlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import M
from joshupscale_amd import dist as jdist


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    assert jdist.env_world() == (rank, rank, world)
    jdist.init("gloo")
    cfg = M.ModelConfig(frame_height=30, frame_width=48, gen_blocks=1)
    blob = M.serialize(cfg, M.make_seeded_weights(cfg)) if rank == 0 else None
    got = jdist.broadcast_model(blob, torch.device("cpu"))
    cfg2, w2 = M.deserialize(got)           # every rank can parse what it received
    t = jdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    jdist.barrier()
    q.put((rank, hashlib.sha256(got).hexdigest(), cfg2.frame_width, t))
    dist.destroy_process_group()


def test_model_broadcast_and_max_time_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]           # identical bytes on both ranks
    assert res[0][2] == res[1][2] == 48
    assert res[0][3] == res[1][3] == 2.0    # max over ranks


def test_stream_assignment_is_round_robin():
    assert jdist.assign_streams(8, 8) == [[i] for i in range(8)]
    assert jdist.assign_streams(5, 2) == [[0, 2, 4], [1, 3]]


def test_single_process_passthrough():
    assert jdist.broadcast_model(b"abc", torch.device("cpu")) == b"abc"
    assert jdist.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def test_bench_rank_pinning_splits_the_allowed_cores(monkeypatch):
    """bench.py pins each rank to its own slice of the cores BEFORE anything touches the GPU
    (round 3: multi-GPU readiness).  Without the amdgpu sysfs tree (this container) the allowed
    cores are split evenly by local rank: disjoint slices that cover them; a single bare
    process is left unpinned."""
    import ast
    import os
    src = open(os.path.join(os.path.dirname(__file__), "..", "bench.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "pin_rank_to_cpus")
    ns = {"os": os}
    exec(compile(ast.Module([fn], []), "bench.py", "exec"), ns)
    allowed = sorted(os.sched_getaffinity(0))
    try:
        monkeypatch.delenv("RANK", raising=False)
        assert ns["pin_rank_to_cpus"](0, 1)["how"].startswith("unpinned")
        assert sorted(os.sched_getaffinity(0)) == allowed
        monkeypatch.setenv("RANK", "0")
        seen = []
        world = min(4, len(allowed))
        for r in range(world):
            os.sched_setaffinity(0, allowed)
            info = ns["pin_rank_to_cpus"](r, world)
            mine = sorted(os.sched_getaffinity(0))
            assert info["cpus"] == len(mine) >= 1 and info["first_cpu"] == mine[0]
            seen.append(mine)
        flat = [c for s in seen for c in s]
        assert len(flat) == len(set(flat)) and set(flat) <= set(allowed)
        assert len(flat) >= len(allowed) - world
    finally:
        os.sched_setaffinity(0, allowed)
