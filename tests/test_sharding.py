"""Multi-GPU path on CPU: pair sharding by target + the single all-gather of 16-byte records,
exercised with world_size 2 (and 3) over gloo.  The per-rank matcher is injected; here it is the
CPU oracle (allowed in tests only) -- on the GPU box it is the HIP path."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nautilus_amd import sharding, synth
from nautilus_amd.csm import MATCH_DTYPE, pack_scans
from oracle import oracle as O


def test_partition_keeps_targets_whole_and_balanced():
    rng = np.random.default_rng(0)
    tgt = rng.integers(0, 50, 1000)
    for w in (1, 2, 3, 8):
        order, bounds = sharding.partition_by_target(tgt, w)
        assert bounds[0] == 0 and bounds[-1] == 1000 and np.all(np.diff(bounds) >= 0)
        assert sorted(order.tolist()) == list(range(1000))
        owners = {}
        for r in range(w):
            for t in set(tgt[order[bounds[r]:bounds[r + 1]]].tolist()):
                assert owners.setdefault(t, r) == r, "target %d straddles ranks" % t
        assert np.diff(bounds).max() <= 1000 / w + 40  # balanced up to one target's pairs
    # degenerate: fewer targets than ranks, empty input
    order, bounds = sharding.partition_by_target(np.zeros(5, int), 4)
    assert np.diff(bounds).sum() == 5
    order, bounds = sharding.partition_by_target(np.zeros(0, int), 2)
    assert list(bounds) == [0, 0, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_matcher(xy, off, gs, ss):
    def fn(src, slot, th0, ids):
        grids = O.grid_build_batch(xy, off, ids, gs, 1)
        m = O.csm_match_batch(xy, off, grids, gs, src, slot, th0, ss, None, 1)
        out = np.zeros(len(src), dtype=MATCH_DTYPE)
        for f in ("itheta", "ix", "iy"):
            out[f] = m[f]
        out["score"] = m["score"].astype(np.float32)
        return out
    return fn


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bag = synth.SynthBag(16)
        xy, off = pack_scans(bag.scans)
        src, tgt, th0 = bag.sample_pairs(per_target=3, targets=[2, 5, 9, 13], min_sep=1)
        gs, ss = O.grid_spec(30.0, 0.05, 2.0, 1e-10), O.search_spec(5, 11, 11, math.radians(2))
        got = sharding.distributed_match(_oracle_matcher(xy, off, gs, ss), src, tgt, th0, rank, world)
        q.put((rank, got.tobytes(), src.tobytes(), tgt.tobytes(), th0.tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_match_equals_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every rank holds the same full result ...
    assert len({r[1] for r in res}) == 1
    got = np.frombuffer(res[0][1], dtype=MATCH_DTYPE)
    src = np.frombuffer(res[0][2], dtype=np.int32)
    tgt = np.frombuffer(res[0][3], dtype=np.int32)
    th0 = np.frombuffer(res[0][4], dtype=np.float64)
    # ... equal to the unsharded computation, in the original pair order
    bag = synth.SynthBag(16)
    xy, off = pack_scans(bag.scans)
    gs, ss = O.grid_spec(30.0, 0.05, 2.0, 1e-10), O.search_spec(5, 11, 11, math.radians(2))
    ids = np.unique(tgt)
    want = _oracle_matcher(xy, off, gs, ss)(src, np.searchsorted(ids, tgt).astype(np.int32), th0, ids)
    assert got.tobytes() == want.tobytes()


def test_all_gather_single_rank_is_identity():
    rec = torch.arange(40, dtype=torch.int32).reshape(10, 4)
    tgt = np.array([3, 1, 2, 1, 3, 3, 0, 2, 1, 0])
    order, bounds = sharding.partition_by_target(tgt, 1)
    out = sharding.all_gather_matches(rec[torch.from_numpy(order)], tgt, 0, 1)
    assert torch.equal(out, rec)


def test_cost_aware_partition_balances_weight_not_count():
    """BASELINE configs[3] is strong scaling: the run ends when the slowest rank does, and the matcher's time per pair
    spans two orders of magnitude.  With a cost estimate per pair the contiguous by-target blocks balance its SUM;
    targets still never straddle ranks, every pair is owned once, and weights=None keeps the count-balanced split."""
    rng = np.random.default_rng(11)
    n_t, per = 400, 10
    tgt = np.repeat(np.arange(n_t), per).astype(np.int32)
    perm = rng.permutation(len(tgt))
    tgt = tgt[perm]
    src = rng.integers(0, n_t, len(tgt)).astype(np.int32)
    th0 = rng.uniform(-1, 1, len(tgt))
    w = np.where(tgt < 80, 9.0, 1.0) * rng.uniform(0.8, 1.2, len(tgt))   # the first fifth of the targets is 9x as heavy
    world = 8
    plain = sharding.ShardPlan(src, tgt, th0, world)
    plan = sharding.ShardPlan(src, tgt, th0, world, w)
    assert list(plain.counts) == [500] * 8
    tot = w.sum()
    assert np.allclose(plan.rank_weight.sum(), tot)
    assert plan.rank_weight.max() < 1.12 * tot / world, plan.rank_weight          # within a target's weight of the ideal
    plain_w = np.array([w[plain.order[plain.bounds[r]:plain.bounds[r + 1]]].sum() for r in range(world)])
    assert plain_w.max() > 2.5 * tot / world                                        # what the count split would have cost
    owned = np.concatenate([plan.shard(r)[0] for r in range(world)])
    assert np.array_equal(np.sort(owned), np.arange(len(tgt)))
    tsets = [set(plan.shard(r)[2].tolist()) for r in range(world)]
    assert all(not (tsets[a] & tsets[b]) for a in range(world) for b in range(a + 1, world)), "a target straddles ranks"
    for r in range(world):
        assert np.array_equal(plan.shard_weights(r), w[plan.shard(r)[0]])
    # heavy-first launch order inside a shard: XCD run x holds weight ranks x, x + 8, ...: every run starts with its heaviest
    sw = plan.shard_weights(0)
    order, inv = sharding.pair_launch_order(sw)
    assert np.array_equal(order[inv], np.arange(len(sw))) and np.array_equal(np.sort(order), np.arange(len(sw)))
    runs = np.array_split(order, 8) if len(sw) % 8 == 0 else None
    if runs is not None:
        for run in runs:
            assert np.all(np.diff(sw[run]) <= 1e-12), "a run must be in descending weight"
        assert max(sw[run].sum() for run in runs) < 1.05 * min(sw[run].sum() for run in runs)


def test_predicted_pair_cost_grows_with_the_predicted_offset():
    poses = np.zeros((6, 3))
    poses[:, 0] = [0.0, 0.5, 1.5, 2.5, 3.5, 3.5]
    poses[5, 2] = 1.0   # heading does not enter
    c = sharding.predicted_pair_cost(poses, [1, 2, 3, 4, 5], [0, 0, 0, 0, 0])
    assert np.all(np.diff(c[:4]) > 0) and c[3] == c[4] and 1.0 <= c[0] < 1.1 and 3.5 < c[3] < 6.0


def test_rank_weight_with_more_ranks_than_targets():
    """three targets over six ranks, weights [1, ..., 1, 5] -- the last pair's weight must stay with the
    last non-empty rank (np.add.reduceat with clipped bounds gave it to an empty trailing rank, times 0)."""
    tgt = np.array([0, 0, 0, 1, 1, 1, 1, 2, 2, 2], dtype=np.int32)
    w = np.ones(len(tgt))
    w[-1] = 5.0
    plan = sharding.ShardPlan(np.zeros(len(tgt), np.int32), tgt, np.zeros(len(tgt)), 6, w)
    assert np.isclose(plan.rank_weight.sum(), w.sum())
    for r in range(6):
        assert np.isclose(plan.rank_weight[r], plan.shard_weights(r).sum()), (r, plan.rank_weight)
    assert (plan.rank_weight[plan.counts == 0] == 0).all()


def _config4_plan(world):
    """BASELINE configs[3]'s pair list (10,000 scans, 1,000,000 pairs, 100 per target; bench.Workload("config4")) from the
    bag's poses alone, and its cost-balanced plan."""
    bag = synth.SynthBag(10000, dense=True, poses_only=True)
    ids = np.arange(10000, dtype=np.int32)
    src, tgt, th0 = bag.sample_pairs(per_target=100, targets=ids, max_dist=3.5, min_sep=20)
    w = sharding.predicted_pair_cost(bag.odom, src, tgt)
    return src, tgt, th0, w, sharding.ShardPlan(src, tgt, th0, world, w)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config4_plan_balances_predicted_cost(world):
    """The 1,000,000-pair list of configs[3] over 2 / 4 / 8 ranks: the plan balances PREDICTED COST (max / mean <= 1.03)
    and that -- not drift -- is why the ranks' pair counts differ (8 ranks: 118,400 .. 127,600 pairs)."""
    src, tgt, th0, w, plan = _config4_plan(world)
    assert plan.n_pairs == 1000000
    rw = plan.rank_weight
    assert np.isclose(rw.sum(), w.sum()) and rw.max() / rw.mean() <= 1.03, rw / rw.mean()
    plain = sharding.ShardPlan(src, tgt, th0, world)                      # the split by pair count, for comparison
    assert plain.counts.max() - plain.counts.min() <= 100                  # (one target's pairs)
    plain_w = np.array([w[plain.order[plain.bounds[r]:plain.bounds[r + 1]]].sum() for r in range(world)])
    assert rw.max() / rw.mean() <= plain_w.max() / plain_w.mean() + 1e-12  # never worse than the count split, by its own measure
    # the count spread IS the cost model's doing: ranks with more pairs hold cheaper pairs
    mean_cost = rw / plan.counts
    if world > 2:
        assert np.corrcoef(plan.counts, mean_cost)[0, 1] < -0.9
    owned = np.concatenate([plan.shard(r)[0] for r in range(world)])
    assert len(owned) == 1000000 and np.array_equal(np.sort(owned), np.arange(1000000))


def _config4_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, tgt, th0, w, plan = _config4_plan(world)
        idx = plan.shard(rank)[0]
        # stand-in records that name their pair: (global index, src, tgt, rank)
        rec = np.stack([idx.astype(np.int32), src[idx], tgt[idx], np.full(len(idx), rank, np.int32)], axis=1)
        full = plan.all_gather(torch.from_numpy(np.ascontiguousarray(rec)), rank).numpy()
        ok = (np.array_equal(full[:, 0], np.arange(plan.n_pairs)) and np.array_equal(full[:, 1], src)
              and np.array_equal(full[:, 2], tgt))
        owner = np.empty(plan.n_pairs, np.int32)
        for r in range(world):
            owner[plan.shard(r)[0]] = r
        ok = ok and np.array_equal(full[:, 3], owner)
        q.put((rank, bool(ok), int(plan.counts[rank]), float(plan.rank_weight[rank]), float(plan.rank_weight.mean())))
    finally:
        dist.destroy_process_group()


def test_config4_all_gather_over_gloo_world_4():
    """configs[3] at full size through the ONE collective of the multi-GPU path, four ranks over gloo: every rank ends
    with the 1,000,000 records in the original pair order, each contributed by the rank the plan gave it to."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert sum(r[2] for r in res) == 1000000
    assert max(r[3] for r in res) / res[0][4] <= 1.03
