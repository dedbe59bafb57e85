"""bench.py's multi-GPU plumbing on CPU: the launcher refuses to run on fewer GPUs than asked, and bench's OWN
sharded step (ShardPlan -> per-rank matcher -> one all-gather -> permutation to the original order,
bench.run_sharded) is driven over gloo at world size 2 with an injected matcher (the CPU oracle: tests only)."""
import math
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from nautilus_amd import sharding, synth  # noqa: E402
from nautilus_amd.csm import MATCH_DTYPE, pack_scans  # noqa: E402
from oracle import oracle as O  # noqa: E402


def test_gpus_flag_fails_loudly_without_enough_gpus():
    """No GPU here: `bench.py --gpus 2` must exit non-zero with a message, never fall back to one rank."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert b"--gpus 2" in p.stderr and b"visible" in p.stderr
    assert b'"metric"' not in p.stdout


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr


class _OracleMatcher:
    """Stands in for bench.HipMatcher on CPU: same step() contract (returns this rank's (n_local, 4) int32 records)."""

    def __init__(self, xy, off, shard, gs, ss):
        idx, src, tgt, th0, ids, slot = shard
        self.args = (xy, off, ids, src, slot, th0, gs, ss)
        self.calls = 0

    def step(self):
        xy, off, ids, src, slot, th0, gs, ss = self.args
        self.calls += 1
        out = np.zeros(len(src), dtype=MATCH_DTYPE)
        if len(src):
            grids = O.grid_build_batch(xy, off, ids, gs, 1)
            m = O.csm_match_batch(xy, off, grids, gs, src, slot, th0, ss, None, 1)
            for f in ("itheta", "ix", "iy"):
                out[f] = m[f]
            out["score"] = m["score"].astype(np.float32)
        return torch.from_numpy(out.view(np.int32).reshape(-1, 4).copy())


def _workload():
    bag = synth.SynthBag(20)
    xy, off = pack_scans(bag.scans)
    src, tgt, th0 = bag.sample_pairs(per_target=3, targets=[1, 4, 8, 12, 17], max_dist=3.5, min_sep=1)
    perm = np.random.default_rng(5).permutation(len(src))  # the global list need not be sorted by target
    return xy, off, src[perm], tgt[perm], th0[perm]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _weights(src, tgt):
    """A cost estimate that makes the pairs of the first two targets heavy (as sharding.predicted_pair_cost would for
    distant pairs)."""
    return np.where(tgt <= 4, 6.0, 1.0)


def _worker(rank, world, port, q, weighted=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xy, off, src, tgt, th0 = _workload()
        gs, ss = O.grid_spec(30.0, 0.05, 2.0, 1e-10), O.search_spec(5, 11, 11, math.radians(2))
        plan = sharding.ShardPlan(src, tgt, th0, world, _weights(src, tgt) if weighted else None)
        m = _OracleMatcher(xy, off, plan.shard(rank), gs, ss)
        elapsed, full = bench.run_sharded(plan, rank, world, "cpu", m, steps=2, warmup=1, dist=dist)
        # the N = 1 point of the same list, timed inside the N-rank run (what bench.py's N > 1 line quotes its speed-up against)
        one = bench.one_gpu_same_workload(src, tgt, th0, _weights(src, tgt) if weighted else None,
                                          lambda sh: _OracleMatcher(xy, off, sh, gs, ss), "cpu", rank, world, dist, full)
        fields = bench.scaling_fields(len(src) * 2 / elapsed, one, world, dist.get_world_size())
        # the N > 1 bench line as bench.worker assembles it: per-rank table over the communicator, config block, scaling
        # fields, then the compact stdout line (what the driver parses)
        cost = float(plan.rank_weight[rank]) if plan.rank_weight is not None else float(len(plan.shard(rank)[1]))
        per_rank = bench.gather_per_rank(torch, dist, world, "cpu", len(plan.shard(rank)[1]), len(plan.shard(rank)[4]),
                                         1e3 * elapsed / 2, 0.1, cost)
        out = {"metric": "loop-closure candidate pairs/sec (1081-beam)", "value": len(src) * 2 / elapsed, "unit": "pairs/s",
               "n_gpus": world, "steps": 2, "warmup": 1, "ms_per_step": 1e3 * elapsed / 2, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
               "config": bench.config_block("test list", "config4", len(src), 20, 1200, 2, dist.get_world_size(), "gloo", world,
                                            per_rank, "predicted cost" if weighted else "pair count"),
               "roofline": {"bound": "onchip-model", "kernel": "k", "achieved": 1.0, "peak": None, "unit": "pairs/s", "frac": None,
                            "traffic": None, "avg_launch_ms": 1.0, "launches": 2, "ideal_ms": None, "stale": True}}
        out.update(fields)
        line = bench.compact_line(out, "bench_details.json") if rank == 0 else None
        q.put((rank, full.numpy().tobytes(), m.calls, elapsed, len(plan.shard(rank)[1]), fields, line))
    finally:
        dist.destroy_process_group()


def test_bench_sharded_step_over_gloo_world_2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len({r[1] for r in res}) == 1, "ranks hold different tables"
    assert all(r[2] == 3 for r in res), "warm-up + steps = 3 matcher calls per rank"
    assert res[0][3] == res[1][3] > 0, "elapsed is the max over ranks, identical everywhere"
    assert res[0][4] + res[1][4] == 15 and min(res[0][4], res[1][4]) >= 6, "pairs are split between the ranks by target"
    # every rank reports the same one-GPU figure of the same list, measured by rank 0, and the ratio to it
    f0, f1 = res[0][5], res[1][5]
    assert f0 == f1 and f0["rccl_world_size"] == 2 and f0["one_gpu_records_equal_sharded_table"]
    assert f0["one_gpu_same_workload_pairs_per_s"] > 0
    assert abs(f0["speedup_vs_one_gpu"] - (15 * 2 / res[0][3]) / f0["one_gpu_same_workload_pairs_per_s"]) < 1e-9
    # equal to the unsharded computation in the ORIGINAL (unsorted) pair order
    xy, off, src, tgt, th0 = _workload()
    gs, ss = O.grid_spec(30.0, 0.05, 2.0, 1e-10), O.search_spec(5, 11, 11, math.radians(2))
    one = sharding.ShardPlan(src, tgt, th0, 1)
    want = one.all_gather(_OracleMatcher(xy, off, one.shard(0), gs, ss).step(), 0)
    assert res[0][1] == want.numpy().tobytes()
    got = np.frombuffer(res[0][1], dtype=MATCH_DTYPE)
    ids = np.unique(tgt)
    direct = O.csm_match_batch(xy, off, O.grid_build_batch(xy, off, ids, gs), gs, src, np.searchsorted(ids, tgt), th0, ss)
    assert np.array_equal(got["ix"], direct["ix"]) and np.array_equal(got["itheta"], direct["itheta"])


def test_cost_aware_split_over_gloo_world_2():
    """The same sharded step with a per-pair cost estimate in the plan (bench.py --mode config4 builds it from the
    odometry poses): the ranks' blocks balance the weight (3 heavy pairs | 3 heavy + 9 light = 18 | 27, where the split
    by count would give 39 | 6), every rank still ends with the same table, equal to the unsharded result in the
    original pair order."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    xy, off, src, tgt, th0 = _workload()
    w = _weights(src, tgt)
    plan = sharding.ShardPlan(src, tgt, th0, world, w)
    assert [res[0][4], res[1][4]] == [int(c) for c in plan.counts] == [3, 12]
    assert list(plan.rank_weight) == [18.0, 27.0]   # within one target's weight (18) of each other
    plain = sharding.ShardPlan(src, tgt, th0, world)
    assert [w[plain.shard(r)[0]].sum() for r in range(world)] == [39.0, 6.0]
    assert len({r[1] for r in res}) == 1, "ranks hold different tables"
    gs, ss = O.grid_spec(30.0, 0.05, 2.0, 1e-10), O.search_spec(5, 11, 11, math.radians(2))
    one = sharding.ShardPlan(src, tgt, th0, 1)
    want = one.all_gather(_OracleMatcher(xy, off, one.shard(0), gs, ss).step(), 0)
    assert res[0][1] == want.numpy().tobytes()


@pytest.mark.parametrize("world", [4, 8])
def test_n_gpu_bench_line_over_gloo(world):
    """World sizes 4 and 8 (the launch path of `bench.py --gpus N` above two ranks; at 8 the list's five targets leave three
    ranks with EMPTY shards -- they still take part in the one all-gather): every rank ends with the same table, and rank
    0's compact line -- the one the driver parses -- is strict JSON below 4 KB that carries the communicator's size, the
    same-workload one-GPU point with the ratio to it, and the ranks' load balance."""
    import json
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert len({r[1] for r in res}) == 1, "ranks hold different tables"
    assert sum(r[4] for r in res) == 15
    line = res[0][6]
    assert line and "\n" not in line and len(line) < 4096 and all(r[6] is None for r in res[1:])
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert d["n_gpus"] == world and d["rccl_world_size"] == world and d["config"]["rccl_world_size"] == world
    assert d["config"]["collective"] == "all_gather 16 B/pair" and d["scaling"] == "strong"
    assert d["one_gpu_same_workload_pairs_per_s"] > 0 and d["one_gpu_records_equal_sharded_table"] is True
    assert abs(d["speedup_vs_one_gpu"] - d["value"] / d["one_gpu_same_workload_pairs_per_s"]) < 1e-4 * d["speedup_vs_one_gpu"]
    xy, off, src, tgt, th0 = _workload()
    plan = sharding.ShardPlan(src, tgt, th0, world, _weights(src, tgt))
    want = float(np.max(plan.rank_weight) / np.mean(plan.rank_weight))
    assert abs(d["config"]["shard_balance"]["max_over_mean_predicted_cost"] - want) < 1e-5 * want
    assert "per_rank" not in d["config"], "the per-rank table belongs to the details file"
