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
