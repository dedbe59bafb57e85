"""Sharding of loop-closure candidate pairs across the GPUs of one node.

The unit of work is the candidate-pair list Solver::SolveAutoLC builds
(/root/reference/src/optimization/solver.cc:676-700).  Pairs are independent (each reads one
source scan and one target grid, writes one 16-byte record), so the only exchange is ONE
all-gather of the per-rank best-pose records (torch.distributed; backend "nccl" is RCCL over xGMI
on ROCm, "gloo" in CPU tests).  Partitioning is by TARGET scan in contiguous blocks, so every likelihood
grid is built on exactly one GPU; the blocks are balanced by pair count or, when the caller has one, by a per-pair
cost estimate: the branch-and-bound matcher's time per pair spans two orders of magnitude (a pair whose true offset
lies outside the search window has a flat score landscape and thousands of candidate blocks), and a strong-scaling
run ends when its slowest rank does.
"""
import numpy as np


def predicted_pair_cost(poses, pair_src, pair_tgt, window_m=2.0):
    """A cost estimate per candidate pair from what the caller already holds: the pose estimates of the two nodes
    (Solver::SolveAutoLC gates its pairs on them, lc_matcher.cc:48-57).  The matcher's bounds phase costs the same for
    every pair; its candidate phase grows steeply once the predicted offset approaches the edge of the search window
    (measured on BASELINE configs[1], profiles/r03_bnb_probe.json: 0.9k / 1.2k / 3.4k / 9.5k evaluated sub-blocks per
    pair at predicted offsets of 0-1 / 1-2 / 2-3 / 3-4 m).  Unit: the cost of a pair at zero offset."""
    P = np.asarray(poses, dtype=np.float64)
    d = np.linalg.norm(P[np.asarray(pair_src), :2] - P[np.asarray(pair_tgt), :2], axis=1)
    return 1.0 + 0.25 * (d / window_m) ** 2 + 1.5 * np.clip(d / window_m - 0.75, 0.0, None) ** 2 * (d / window_m)


def partition_by_target(pair_tgt, world_size, weights=None):
    """Returns (order, bounds): `order` sorts pairs by target (stable); rank r owns
    order[bounds[r]:bounds[r+1]].  Boundaries fall between targets and balance the pair count, or the sum of
    `weights` (one non-negative cost per pair) when given."""
    pair_tgt = np.asarray(pair_tgt)
    n = len(pair_tgt)
    order = np.argsort(pair_tgt, kind="stable")
    bounds = [0]
    st = pair_tgt[order]
    if weights is not None:
        w = np.asarray(weights, dtype=np.float64)[order]
        assert len(w) == n and (n == 0 or w.min() >= 0.0), "one non-negative weight per pair"
        cum = np.concatenate([[0.0], np.cumsum(w)])
    for r in range(1, world_size):
        if weights is None or cum[-1] <= 0.0:
            ideal = (n * r) // world_size
        else:
            # first position whose preceding weight reaches r / world of the total
            ideal = int(np.searchsorted(cum, cum[-1] * r / world_size, side="left"))
        ideal = min(max(ideal, bounds[-1]), n)
        # the cut goes to the nearer change of target (never before the previous cut): a target never straddles ranks
        fwd = ideal
        while 0 < fwd < n and st[fwd] == st[fwd - 1]:
            fwd += 1
        cut = fwd
        if weights is not None:
            back = ideal
            while 0 < back < n and st[back] == st[back - 1]:
                back -= 1
            if back > bounds[-1] and cum[ideal] - cum[back] < cum[fwd] - cum[ideal]:
                cut = back
        bounds.append(min(cut, n))
    bounds.append(n)
    return order, np.asarray(bounds, dtype=np.int64)


class ShardPlan:
    """The partition of one global pair list, computed once (the sort of a million pairs is host
    work that must stay out of a timed step) and shared by every rank: which pairs a rank matches,
    which grids it builds, and the index tables that put the all-gathered blocks back into the
    original pair order."""

    def __init__(self, pair_src, pair_tgt, theta0, world_size, weights=None):
        """weights: optional cost estimate per pair (predicted_pair_cost, or the matcher's own per-pair counts from
        an earlier pass): the ranks' blocks then balance its sum instead of the pair count, and shard() orders each
        rank's pairs for a heavy-first launch (pair_launch_order)."""
        self.pair_src = np.ascontiguousarray(pair_src, dtype=np.int32)
        self.pair_tgt = np.ascontiguousarray(pair_tgt, dtype=np.int32)
        self.theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        self.world = int(world_size)
        self.n_pairs = len(self.pair_src)
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.order, self.bounds = partition_by_target(self.pair_tgt, self.world, self.weights)
        self.counts = np.diff(self.bounds)
        # (from the cumulative sum: np.add.reduceat with clipped bounds handed the last pair of the last non-empty rank
        #  to an empty trailing rank, where it was then multiplied by 0)
        self.rank_weight = None
        if self.weights is not None and self.n_pairs:
            cum = np.concatenate([[0.0], np.cumsum(self.weights[self.order])])
            self.rank_weight = cum[self.bounds[1:]] - cum[self.bounds[:-1]]
        self.width = int(self.counts.max()) if self.n_pairs else 0  # padded block length of the all-gather
        # row of the gathered (world * width) table that holds the pair at position i of `order`
        self.gathered_rows = np.concatenate(
            [r * self.width + np.arange(self.counts[r], dtype=np.int64) for r in range(self.world)]
        ) if self.world else np.zeros(0, np.int64)
        self._dev_tables = {}

    def shard(self, rank):
        """This rank's pairs: (global indices, src, tgt, theta0, target ids, grid slot per pair)."""
        idx = self.order[self.bounds[rank]:self.bounds[rank + 1]]
        src, tgt, th = self.pair_src[idx], self.pair_tgt[idx], self.theta0[idx]
        ids = np.unique(tgt)
        slot = np.searchsorted(ids, tgt).astype(np.int32)
        return idx, src, tgt, th, ids.astype(np.int32), slot

    def shard_weights(self, rank):
        """The cost estimates of this rank's pairs, in shard order (None without weights)."""
        if self.weights is None:
            return None
        return self.weights[self.order[self.bounds[rank]:self.bounds[rank + 1]]]

    def _tables(self, device):
        import torch
        key = str(device)
        if key not in self._dev_tables:
            self._dev_tables[key] = (torch.from_numpy(self.gathered_rows).to(device),
                                     torch.from_numpy(self.order.astype(np.int64)).to(device))
        return self._dev_tables[key]

    def new_buffers(self, device):
        """(padded send block, gathered table, result in original order): allocate once, reuse per step."""
        import torch
        z = lambda n: torch.zeros((n, 4), dtype=torch.int32, device=device)
        return z(self.width), z(self.world * self.width), z(self.n_pairs)

    def all_gather(self, local_records, rank, buffers=None, group=None):
        """local_records: (counts[rank], 4) int32 tensor = nhip_match_t rows of this rank's shard, in
        shard order.  ONE all-gather of equal-sized (padded) blocks, then a device-side permutation;
        returns the (n_pairs, 4) table in the ORIGINAL pair order, identical on every rank."""
        import torch.distributed as dist
        dev = local_records.device
        block, gathered, out = buffers if buffers is not None else self.new_buffers(dev)
        n_local = int(self.counts[rank])
        assert local_records.shape[0] == n_local, (tuple(local_records.shape), n_local)
        block[:n_local].copy_(local_records)
        if self.world > 1:
            dist.all_gather_into_tensor(gathered, block, group=group)
        else:
            gathered.copy_(block)
        rows, order = self._tables(dev)
        out.index_copy_(0, order, gathered.index_select(0, rows))
        return out


def pair_launch_order(weights):
    """Order in which a rank hands its pairs to the matcher: heaviest first.  One workgroup works one pair and the
    dispatcher starts workgroups in index order, so a pair that takes milliseconds must not be among the last to
    start (longest-processing-time-first; ties keep the shard's by-target order, which keeps a target's pairs
    together).  Returns (perm, inverse): matcher position i holds shard pair perm[i]."""
    w = np.asarray(weights, dtype=np.float64)
    by_weight = np.argsort(-w, kind="stable")
    # The matcher deals its batch to the 8 XCDs as 8 contiguous runs of the pair list (pair = xcd * ceil(n / 8) + i,
    # workgroup 8 i + xcd: a target's pairs, consecutive in a by-target list, then share an XCD's L2) and the
    # dispatcher starts workgroups in index order: weight rank r goes to run r % 8, position r // 8, so that the
    # heaviest pairs start first on EVERY XCD (sorted as one run, the first XCD would get all the heavy ones).
    n = len(w)
    per = (n + 7) // 8
    r = np.arange(n)
    pos = (r % 8) * per + r // 8
    perm = by_weight[np.argsort(pos, kind="stable")]
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    return perm, inv


def local_shard(pair_src, pair_tgt, theta0, rank, world_size):
    """This rank's pairs: (global indices, src, tgt, theta0, target ids, grid slot per pair)."""
    return ShardPlan(pair_src, pair_tgt, theta0, world_size).shard(rank)


def all_gather_matches(local_records, pair_tgt, rank, world_size, group=None):
    """One-shot form of ShardPlan.all_gather (builds the plan from pair_tgt)."""
    n = len(pair_tgt)
    plan = ShardPlan(np.zeros(n, np.int32), pair_tgt, np.zeros(n), world_size)
    return plan.all_gather(local_records, rank, group=group)


def distributed_match(match_fn, pair_src, pair_tgt, theta0, rank, world_size, device="cpu", group=None):
    """match_fn(src, slot, theta0, target_ids) -> structured MATCH_DTYPE array for this rank's shard
    (on the GPU box: LikelihoodGrids + match_pairs).  Returns all matches in original order."""
    import torch
    from .csm import MATCH_DTYPE
    plan = ShardPlan(pair_src, pair_tgt, theta0, world_size)
    idx, src, tgt, th, ids, slot = plan.shard(rank)
    rec = match_fn(src, slot, th, ids)
    rec = np.ascontiguousarray(rec, dtype=MATCH_DTYPE)
    local = torch.from_numpy(rec.view(np.int32).reshape(-1, 4).copy()).to(device)
    full = plan.all_gather(local, rank, group=group)
    return full.cpu().numpy().reshape(-1).view(MATCH_DTYPE)
