"""TEST INFRASTRUCTURE (like everything under oracle/): the expected results of ONE replayed proof, computed with the CPU oracle.

`snarkvm_amd/proofs.py::replay` issues the hot-path calls of a Varuna proof (credits.aleo transfer_private shapes) on
device-resident random data and returns its 14 G1 commitments and one G2 MSM result.  This module restates the same data flow on
the CPU oracle (oracle/cpu.py: fft_in_place, the pointwise passes, Polynomial::divide_with_q_and_r, batched::msm, standard::msm)
so that bench.py --workload proofs64 and tests/test_gpu_proofs.py can check a whole replayed proof - all 15 results - against
the reference algorithms, not against another replay of the same library.  Never imported by the product.

Reference call sites of the steps: first.rs:127-160, second.rs:104-170, third.rs:158-317, fourth.rs:174-231, fifth.rs:50-66,
sonic_pc/mod.rs:177-257, 316-337, kzg10/mod.rs:117-149, 213-236.
"""
import numpy as np

from . import cpu as oracle


def _fr_vec(pool, shift, n):
    return np.ascontiguousarray(pool[shift : shift + n]).reshape(-1, 4)


def _pad(v, n):
    out = np.zeros((n, 4), dtype=np.uint64)
    out[: v.shape[0]] = v
    return out


def expected_results(pool_host, g1_host, g2_host, point, lg_r, lg_k, lg_g2, nmax, salt):
    """[14 G1 affine records (oracle.G1_AFFINE, shape (1,)), then the G2 affine record or None] for the proof `salt`.

    pool_host: (m, 4) u64 Fr Montgomery images; g1_host: registered G1 bases (powers at [0, nmax), hiding bases at [nmax, ...));
    g2_host: registered G2 bases or None; point: (1, 4) the opening point."""
    nR, nK = 1 << lg_r, 1 << lg_k
    pool = np.ascontiguousarray(pool_host, dtype=np.uint64).reshape(-1, 4)
    INV, FWD, COSET = oracle.INVERSE, oracle.FORWARD, oracle.COSET
    out = []

    def load(n, shift, size=None):
        v = _fr_vec(pool, shift + salt, n)
        return _pad(v, size) if size else v.copy()

    def ntt(v, direction, kind=oracle.STANDARD):
        return oracle.ntt(v, oracle.ORDER_NN, direction, kind)

    def product(x, y, lg):  # the replay's product(): forward transforms of the zero-padded vectors, pointwise product, inverse transform
        fx, fy = ntt(_pad(x, 1 << lg), FWD), ntt(_pad(y, 1 << lg), FWD)
        return ntt(oracle.fr_vec_op("mul", fx, fy), INV)

    def commit(vec, n, hiding):
        """KZG10::commit of the first n + hiding entries of `vec`: plaintext MSM over powers[0, n) + hiding MSM over the bases at nmax."""
        sc = oracle.fr_op("to_bigint", np.ascontiguousarray(vec[: n + hiding]))
        acc = oracle.g1_msm(g1_host[:n], sc[:n], oracle.MSM_BATCHED)
        if hiding:
            acc = oracle.g1_add(acc, oracle.g1_msm(g1_host[nmax : nmax + hiding], sc[n : n + hiding], oracle.MSM_BATCHED))
        out.append(oracle.g1_to_affine(acc))

    # the work vectors hold nmax elements; a load zero-fills the rest, transforms only touch their domain
    a = ntt(load(nR, 1), INV)                                                                # round 1
    ntt(load(nR, 2), FWD)
    commit(_pad(a, nmax), nR - 2, 2)
    za, zb, zc = (ntt(load(nR, 10 + i), INV) for i in range(3))                              # round 2
    prod = product(za, zb, lg_r + 1)
    row = oracle.fr_vec_op("sub", prod, _pad(zc, 2 * nR))
    q, _ = oracle.poly_divide(row, [(0, oracle.fr_op("neg", oracle.fr_op("from_bigint", np.array([[1, 0, 0, 0]], dtype=np.uint64)))[0]),
                                    (nR, oracle.fr_op("from_bigint", np.array([[1, 0, 0, 0]], dtype=np.uint64))[0])])
    commit(_pad(q, nmax), nR, 0)
    a_vec = b_vec = None                                                                     # round 3
    for m in range(3):
        t = ntt(load(nR, 20 + m), INV)
        b_in = load(nR, 30 + m)
        a_vec = product(t, b_in, lg_r + 1)                                                   # result in `a` (2 nR elements)
        b_vec = ntt(_pad(b_in, 2 * nR), FWD)                                                 # `b` is left holding its forward transform
    commit(_pad(a_vec, nmax), nR - 1, 2)
    commit(_pad(b_vec, nmax), nR, 0)
    r4 = []                                                                                  # round 4
    for m in range(3):
        v = ntt(load(nK, 40 + m), INV)
        ntt(load(nK, 50 + m), INV)
        ntt(load(nK, 60 + m), INV, COSET)
        if m == 0:
            v = product(v, load(nK, 70), lg_k + 1)
        r4.append(v)
    for v in r4:
        commit(_pad(v, nmax), nK - 1, 0)
    for o, n in ((3, nK - 2), (5, nK), (9, nR), (11, nK)):                                   # round 5: straight from the pool
        commit(_fr_vec(pool, o + salt, n), n, 0)
    z = np.ascontiguousarray(point, dtype=np.uint64).reshape(1, 4)
    one = oracle.fr_op("from_bigint", np.array([[1, 0, 0, 0]], dtype=np.uint64))
    for s, n in ((13, nK), (17, nR), (19, nK)):                                              # openings: p / (X - z)
        p = load(n, s)
        q, _ = oracle.poly_divide(p, [(0, oracle.fr_op("neg", z)[0]), (1, one[0])])
        commit(_pad(q, nmax), n - 1, 0)
    if g2_host is not None and lg_g2:
        n2 = 1 << lg_g2
        sc = _fr_vec(pool, 23 + salt, n2)                                                    # read as canonical integers (scalars_montgomery = 0)
        out.append(oracle.g2_to_affine(oracle.g2_msm(g2_host[:n2], sc)))
    else:
        out.append(None)
    return out
