"""Sharding of loop-closure candidate pairs across the GPUs of one node.

Pairs are independent (each reads one source scan and one target grid, writes one 16-byte
record), so the only exchange is ONE all-gather of the per-rank best-pose records
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).
Partitioning is by TARGET scan in contiguous blocks balanced by pair count, so every likelihood
grid is built on exactly one GPU.
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


def local_shard(pair_src, pair_tgt, theta0, rank, world_size):
    """This rank's pairs: (global indices, src, tgt, theta0, target ids, grid slot per pair)."""
    order, bounds = partition_by_target(pair_tgt, world_size)
    idx = order[bounds[rank]:bounds[rank + 1]]
    src, tgt, th = np.asarray(pair_src)[idx], np.asarray(pair_tgt)[idx], np.asarray(theta0)[idx]
    ids = np.unique(tgt)
    slot = np.searchsorted(ids, tgt).astype(np.int32)
    return idx, src.astype(np.int32), tgt.astype(np.int32), th.astype(np.float64), ids.astype(np.int32), slot


def all_gather_matches(local_records, pair_tgt, rank, world_size, group=None):
    """local_records: torch tensor (n_local, 4) int32 = nhip_match_t rows of this rank's shard,
    in shard order.  One all-gather of equal-sized (padded) blocks; returns a (n_pairs, 4) int32
    tensor in the ORIGINAL pair order, identical on every rank."""
    import torch
    import torch.distributed as dist
    order, bounds = partition_by_target(pair_tgt, world_size)
    counts = np.diff(bounds)
    width = int(counts.max()) if len(counts) else 0
    dev = local_records.device
    block = torch.zeros((width, 4), dtype=torch.int32, device=dev)
    n_local = int(counts[rank])
    assert local_records.shape[0] == n_local, (local_records.shape, n_local)
    block[:n_local] = local_records
    gathered = torch.empty((world_size * width, 4), dtype=torch.int32, device=dev)
    if world_size > 1:
        dist.all_gather_into_tensor(gathered, block, group=group)
    else:
        gathered.copy_(block)
    out = torch.empty((len(order), 4), dtype=torch.int32, device=dev)
    gathered = gathered.view(world_size, width, 4)
    for r in range(world_size):
        idx = torch.from_numpy(order[bounds[r]:bounds[r + 1]]).to(dev)
        out[idx] = gathered[r, :int(counts[r])]
    return out


def distributed_match(match_fn, pair_src, pair_tgt, theta0, rank, world_size, device="cpu", group=None):
    """match_fn(src, slot, theta0, target_ids) -> structured MATCH_DTYPE array for this rank's shard
    (on the GPU box: LikelihoodGrids + match_pairs).  Returns all matches in original order."""
    import torch
    from .csm import MATCH_DTYPE
    idx, src, tgt, th, ids, slot = local_shard(pair_src, pair_tgt, theta0, rank, world_size)
    rec = match_fn(src, slot, th, ids)
    rec = np.ascontiguousarray(rec, dtype=MATCH_DTYPE)
    local = torch.from_numpy(rec.view(np.int32).reshape(-1, 4).copy()).to(device)
    full = all_gather_matches(local, pair_tgt, rank, world_size, group)
    return full.cpu().numpy().reshape(-1).view(MATCH_DTYPE)
