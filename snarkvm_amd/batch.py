"""Instance-level sharding of independent MSM / NTT jobs over the GPUs of one node.

A batch of proofs yields independent MSM instances (one per committed polynomial, reference fan-out:
polycommit/sonic_pc/mod.rs:186-245).  They are partitioned round-robin over the ranks (one process per GPU); every
rank holds its own registered copy of the static SRS bases, so the data path needs NO collective.  The only exchange
is the final gather of the 144-byte Jacobian results (replaces the host-side channel + `dadd` of
algorithms/cuda/cuda/snarkvm.cu:287-295): one `all_gather` of a few KB over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).
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
