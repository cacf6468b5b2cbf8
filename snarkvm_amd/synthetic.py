"""Deterministic synthetic inputs for the MSM / NTT configs of BASELINE.json (host side, numpy only).

The reference's generators are OS-seeded (utilities/src/rand.rs:42-48), so the workloads are pinned
here instead: a SplitMix64 stream per seed; field elements are sampled the way
the reference samples `Fr::rand` (fields/src/macros.rs:40-57: 4 random u64 limbs, clear the top
REPR_SHAVE_BITS = 3 bits, reject values >= r).
"""
import numpy as np

R_LIMBS = np.array([725501752471715841, 6461107452199829505, 6968279316240510977, 1345280370688173398], dtype=np.uint64)
R_MOD = sum(int(l) << (64 * i) for i, l in enumerate(R_LIMBS))

SEED_MSM_2_16 = 0x5EED0001
SEED_MSM_LARGE = 0x5EED0002
SEED_NTT = 0x5EED0003


def splitmix64(seed, n, offset=0):
    """n outputs of SplitMix64(seed), starting at stream position `offset`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _lt_modulus(c):
    """c: (m,4) u64 candidates -> boolean mask c < r (lexicographic from the top limb)."""
    lt = np.zeros(c.shape[0], dtype=bool)
    eq = np.ones(c.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (c[:, i] < R_LIMBS[i])
        eq &= c[:, i] == R_LIMBS[i]
    return lt


def random_fr_integers(n, seed):
    """(n,4) u64: uniform canonical integers in [0, r) by rejection sampling, deterministic in (n, seed)."""
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    offset = 0
    while filled < n:
        m = max(1024, int((n - filled) * 1.8))
        c = splitmix64(seed, 4 * m, offset).reshape(m, 4).copy()
        offset += 4 * m
        c[:, 3] &= np.uint64((1 << 61) - 1)
        c = c[_lt_modulus(c)]
        take = min(n - filled, c.shape[0])
        out[filled : filled + take] = c[:take]
        filled += take
    return out


def witness_like_scalars(n, seed):
    """'Witness-like' distribution: 50 % zero, 25 % < 2^16, 25 % uniform."""
    s = random_fr_integers(n, seed)
    sel = splitmix64(seed ^ 0xA5A5, n) % np.uint64(4)
    s[sel < 2] = 0
    small = sel == 2
    s[small, 1:] = 0
    s[small, 0] &= np.uint64(0xFFFF)
    return s
