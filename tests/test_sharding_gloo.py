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
