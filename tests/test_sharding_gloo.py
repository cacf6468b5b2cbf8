"""N > 1 path on CPU: two gloo processes shard a batch of independent MSM instances round-robin and all_gather the
144-byte results.  The per-instance compute is injected (the oracle here; the HIP backend on a GPU node)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, util.ROOT)
    import torch.distributed as dist

    from oracle import cpu as oracle
    from snarkvm_amd import batch, synthetic

    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = util.g1_generator_affine()
    bases = oracle.g1_gen_bases(g, 1, 64)
    instances = [synthetic.random_fr_integers(64, 900 + i) for i in range(5)]  # 5 instances over 2 ranks: ragged
    calls = []

    def compute(sc):
        calls.append(1)
        return oracle.g1_msm(bases, sc, oracle.MSM_BATCHED)

    res = batch.run_sharded(instances, compute)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, len(calls), [r.tobytes() for r in res]))


def test_two_rank_sharded_batch_matches_single_process():
    import torch.multiprocessing as mp

    from oracle import cpu as oracle
    from snarkvm_amd import batch, synthetic

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = util.g1_generator_affine()
    bases = oracle.g1_gen_bases(g, 1, 64)
    want = [oracle.g1_msm(bases, synthetic.random_fr_integers(64, 900 + i), oracle.MSM_BATCHED).tobytes() for i in range(5)]
    calls = {}
    for rank, ncalls, res in got:
        assert res == want  # every rank holds the complete, ordered result list
        calls[rank] = ncalls
    assert calls == {0: 3, 1: 2}  # round-robin: rank 0 -> instances 0,2,4; rank 1 -> 1,3
    assert batch.assigned(5, 2, 0) == [0, 2, 4] and batch.assigned(5, 2, 1) == [1, 3]


def _split_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, util.ROOT)
    import torch.distributed as dist

    from oracle import cpu as oracle
    from snarkvm_amd import batch, synthetic

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 101  # odd: the two ranges differ by one point
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 1, n)
    scalars = synthetic.random_fr_integers(n, 4321)
    seen = []

    def compute_range(lo, hi):
        seen.append((lo, hi))
        return oracle.g1_msm(bases[lo:hi], scalars[lo:hi], oracle.MSM_BATCHED)

    def combine(parts):
        from snarkvm_amd.layout import G1_PROJECTIVE

        acc = np.frombuffer(parts[0].tobytes(), dtype=G1_PROJECTIVE)
        for p in parts[1:]:
            acc = oracle.g1_add(acc, np.frombuffer(p.tobytes(), dtype=G1_PROJECTIVE))
        return acc

    total = batch.msm_split(n, compute_range, combine)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, seen, total.tobytes()))


def test_two_rank_point_range_split_msm():
    """One MSM split by point range over two gloo ranks (second bullet): per-rank partial sums,
    all_gather of one Jacobian point per rank, local combination; every rank ends with the full result."""
    import torch.multiprocessing as mp

    from oracle import cpu as oracle
    from snarkvm_amd import batch, synthetic
    from snarkvm_amd.layout import G1_PROJECTIVE

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 101
    bases = oracle.g1_gen_bases(util.g1_generator_affine(), 1, n)
    want = oracle.g1_to_affine(oracle.g1_msm(bases, synthetic.random_fr_integers(n, 4321), oracle.MSM_BATCHED))
    ranges = {}
    for rank, seen, total in got:
        assert util.affine_equal(oracle.g1_to_affine(np.frombuffer(total, dtype=G1_PROJECTIVE)), want)
        ranges[rank] = seen
    assert ranges == {0: [(0, 51)], 1: [(51, 101)]}
    assert [batch.split_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert batch.split_range(2, 4, 3) == (2, 2)  # more ranks than points: empty slices
