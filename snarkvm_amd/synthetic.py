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


def witness_like_from(scalars, seed):
    """The witness-like distribution laid over an existing uniform vector (same shape): entry i keeps its value, becomes zero or keeps only its low 16 bits -
    the selector of `witness_like_scalars`.  For vectors that already exist on the host (bench.py derives its second distribution from the seeded one)."""
    s = np.array(scalars, dtype=np.uint64, copy=True).reshape(-1, 4)
    sel = splitmix64(seed ^ 0xA5A5, s.shape[0]) % np.uint64(4)
    s[sel < 2] = 0
    small = sel == 2
    s[small, 1:] = 0
    s[small, 0] &= np.uint64(0xFFFF)
    return s


def witness_like_fr_montgomery(n, seed):
    """(n,4) u64 `Fr` memory images (Montgomery form, R = 2^256) whose VALUES follow the witness-like distribution: what a prover's witness-shaped
    coefficient vector looks like to a commitment that fuses `to_bigint` (scalars_montgomery = 1).  Python integers: meant for pools of <= 2^20 elements."""
    w = witness_like_scalars(n, seed)
    out = np.zeros((n, 4), dtype=np.uint64)
    nz = np.flatnonzero(w.any(axis=1))
    R = (1 << 256) % R_MOD
    mask = (1 << 64) - 1
    for i in nz:
        v = (int(w[i, 0]) | (int(w[i, 1]) << 64) | (int(w[i, 2]) << 128) | (int(w[i, 3]) << 192)) * R % R_MOD
        out[i] = (v & mask, (v >> 64) & mask, (v >> 128) & mask, v >> 192)
    return out


# ------------------------------------------------------------------------------------------ G2 base sets
Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
# G2 generator, canonical coordinates (x = x0 + x1 u, y = y0 + y1 u over Fq[u]/(u^2 + 5); curves/src/bls12_377/g2.rs:237-315)
G2_GEN = ((170590608266080109581922461902299092015242589883741236963254737235977648828052995125541529645051927918098146183295,
           83407003718128594709087171351153471074446327721872642659202721143408712182996929763094113874399921859453255070254),
          (1843833842842620867708835993770650838640642469700861403869757682057607397502738488921663703124647238454792872005,
           33145532013610981697337930729788870077912093258611421158732879580766461459275194744385880708057348608045241477209))


def _fq2_mul(a, b):
    return ((a[0] * b[0] - 5 * a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def _fq2_inv(a):
    ninv = pow((a[0] * a[0] + 5 * a[1] * a[1]) % Q_MOD, Q_MOD - 2, Q_MOD)
    return (a[0] * ninv % Q_MOD, (-a[1]) * ninv % Q_MOD)


def _g2_add_affine(p, q):
    """p + q for affine G2 points with p != -q (a = 0 curve; plain chord / tangent formulas over Python integers)."""
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        num = _fq2_mul((3, 0), _fq2_mul(x1, x1))
        den = ((2 * y1[0]) % Q_MOD, (2 * y1[1]) % Q_MOD)
    else:
        num = ((y2[0] - y1[0]) % Q_MOD, (y2[1] - y1[1]) % Q_MOD)
        den = ((x2[0] - x1[0]) % Q_MOD, (x2[1] - x1[1]) % Q_MOD)
    lam = _fq2_mul(num, _fq2_inv(den))
    l2 = _fq2_mul(lam, lam)
    x3 = ((l2[0] - x1[0] - x2[0]) % Q_MOD, (l2[1] - x1[1] - x2[1]) % Q_MOD)
    t = _fq2_mul(lam, ((x1[0] - x3[0]) % Q_MOD, (x1[1] - x3[1]) % Q_MOD))
    return (x3, ((t[0] - y1[0]) % Q_MOD, (t[1] - y1[1]) % Q_MOD))


def g2_points(n, distinct=512):
    """n G2 bases as a Rust `[G2Affine]` array (200 B stride, Montgomery coordinates): `distinct` consecutive multiples of the
    generator tiled to n - the shape of the reference's MSM benches, which tile a small set of points
    (algorithms/benches/msm/variable_base.rs:29-32)."""
    from .layout import G2_AFFINE

    distinct = max(1, min(distinct, n))
    R384 = 1 << 384
    pts = [G2_GEN]
    for _ in range(distinct - 1):
        pts.append(_g2_add_affine(pts[-1], G2_GEN))
    out = np.zeros(distinct, dtype=G2_AFFINE)
    for i, (x, y) in enumerate(pts):
        limbs = []
        for v in (x[0], x[1], y[0], y[1]):
            m = v * R384 % Q_MOD
            limbs.append([(m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(6)])
        out[i]["x"] = limbs[0] + limbs[1]
        out[i]["y"] = limbs[2] + limbs[3]
    reps = (n + distinct - 1) // distinct
    return np.tile(out, reps)[:n].copy()
