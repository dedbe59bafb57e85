"""Sharding of loop-closure candidate pairs across the GPUs of one node.

The unit of work is the candidate-pair list Solver::SolveAutoLC builds
(/root/reference/src/optimization/solver.cc:676-700).  Pairs are independent (each reads one
source scan and one target grid, writes one 16-byte record), so the only exchange is ONE
all-gather of the per-rank best-pose records (torch.distributed; backend "nccl" is RCCL over xGMI
on ROCm, "gloo" in CPU tests).  Partitioning is by TARGET scan in contiguous blocks balanced by
pair count, so every likelihood grid is built on exactly one GPU.
"""
import numpy as np


def partition_by_target(pair_tgt, world_size):
    """Returns (order, bounds): `order` sorts pairs by target (stable); rank r owns
    order[bounds[r]:bounds[r+1]].  Boundaries fall between targets and balance pair counts."""
    pair_tgt = np.asarray(pair_tgt)
    n = len(pair_tgt)
    order = np.argsort(pair_tgt, kind="stable")
    bounds = [0]
    st = pair_tgt[order]
    for r in range(1, world_size):
        ideal = (n * r) // world_size
        ideal = max(ideal, bounds[-1])
        # move the cut forward to the next change of target so a target never straddles ranks
        cut = ideal
        while 0 < cut < n and st[cut] == st[cut - 1]:
            cut += 1
        bounds.append(min(cut, n))
    bounds.append(n)
    return order, np.asarray(bounds, dtype=np.int64)


class ShardPlan:
    """The partition of one global pair list, computed once (the sort of a million pairs is host
    work that must stay out of a timed step) and shared by every rank: which pairs a rank matches,
    which grids it builds, and the index tables that put the all-gathered blocks back into the
    original pair order."""

    def __init__(self, pair_src, pair_tgt, theta0, world_size):
        self.pair_src = np.ascontiguousarray(pair_src, dtype=np.int32)
        self.pair_tgt = np.ascontiguousarray(pair_tgt, dtype=np.int32)
        self.theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        self.world = int(world_size)
        self.n_pairs = len(self.pair_src)
        self.order, self.bounds = partition_by_target(self.pair_tgt, self.world)
        self.counts = np.diff(self.bounds)
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
