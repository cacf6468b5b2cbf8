"""Instance-level sharding of independent MSM / NTT jobs over the GPUs of one node.

A batch of proofs yields independent MSM instances (one per committed polynomial, reference fan-out:
polycommit/sonic_pc/mod.rs:186-245).  They are partitioned round-robin over the ranks (one process per GPU); every
rank holds its own registered copy of the static SRS bases, so the data path needs NO collective.  The only exchange
is the final gather of the 144-byte Jacobian results (replaces the host-side channel + `dadd` of
algorithms/cuda/cuda/snarkvm.cu:287-295): one `all_gather` of a few KB over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).

A single large MSM can instead be split by POINT RANGE (`split_range` / `msm_split`): rank d sums its n / world pairs with
the full window set and the partial results - one Jacobian point per rank - are all-gathered and added on every rank,
exactly the exchange of the reference's multi-GPU MSM (snarkvm.cu:254-295: per-device slices, host `dadd` of the
results).  One latency-bound collective of world * 144 bytes; link bandwidth is irrelevant.
"""
import numpy as np

RESULT_BYTES = 144  # G1Projective


def assigned(n_instances, world, rank):
    """Indices of the instances this rank computes (round-robin keeps per-rank work within one instance)."""
    return list(range(rank, n_instances, world))


def run_sharded(instances, compute, group=None, device=None):
    """Compute `compute(instance)` -> 144-byte result (any buffer) for this rank's share and return the full,
    ordered list of results on every rank.

    instances: list (same length and order on every rank); compute: callable; group: torch.distributed process group
    (None = default group, or single-process when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist

    n = len(instances)
    if not (dist.is_available() and dist.is_initialized()):
        return [np.frombuffer(bytes(memoryview(np.ascontiguousarray(compute(x))).cast("B")), dtype=np.uint8).copy() for x in instances]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = assigned(n, world, rank)
    per_rank = (n + world - 1) // world
    local = torch.zeros((per_rank, RESULT_BYTES), dtype=torch.uint8)
    for slot, idx in enumerate(mine):
        buf = np.frombuffer(bytes(memoryview(np.ascontiguousarray(compute(instances[idx]))).cast("B")), dtype=np.uint8)
        assert buf.size == RESULT_BYTES
        local[slot] = torch.from_numpy(buf.copy())
    if device is not None:
        local = local.to(device)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    out = [None] * n
    for r in range(world):
        g = gathered[r].cpu().numpy()
        for slot, idx in enumerate(assigned(n, world, r)):
            out[idx] = g[slot].copy()
    return out


def split_range(n, world, rank):
    """[lo, hi) of the points rank `rank` sums in a point-range-split MSM (contiguous, sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def msm_split(n, compute_range, combine, group=None, device=None):
    """One MSM of n pairs split by point range over the ranks.

    compute_range(lo, hi) -> 144-byte partial result of this rank's slice; combine(parts) -> 144-byte sum of a
    (world, 144) uint8 array (snarkvm_amd.msm.g1_sum on a GPU node).  Every rank returns the same total."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return np.frombuffer(bytes(memoryview(np.ascontiguousarray(compute_range(0, n))).cast("B")), dtype=np.uint8).copy()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = split_range(n, world, rank)
    part = np.frombuffer(bytes(memoryview(np.ascontiguousarray(compute_range(lo, hi))).cast("B")), dtype=np.uint8)
    assert part.size == RESULT_BYTES
    local = torch.from_numpy(part.copy())
    if device is not None:
        local = local.to(device)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    parts = np.stack([g.cpu().numpy() for g in gathered])
    return np.frombuffer(bytes(memoryview(np.ascontiguousarray(combine(parts))).cast("B")), dtype=np.uint8).copy()
