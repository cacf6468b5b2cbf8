"""Tier-0 oracle: BLS12-377 MSM / NTT in plain Python big integers.

TEST INFRASTRUCTURE ONLY. Nothing under snarkvm_amd/ may import this module; it exists so
that tests/ can check (a) the C++ restatement in oracle/cpu_oracle.cpp and (b) the HIP path
against an implementation that shares no code (and no limb arithmetic) with either.

Every function states the reference behaviour it restates (paths relative to the snarkVM
checkout, v1.0.0):
  * field constants ............ curves/src/bls12_377/{fr,fq}.rs (values re-derived here and
                                 compared with tests/golden/constants.json by tests/test_oracle.py)
  * affine add / double ........ curves/src/templates/short_weierstrass_jacobian/affine.rs:224-273
  * mul_bits ................... .../affine.rs:173-182 (MSB-first double-and-add)
  * msm_naive .................. algorithms/src/msm/variable_base/mod.rs:52-58
  * ntt / intt / coset ......... algorithms/src/fft/domain.rs:169-221, 403-443 (definitional DFT)
  * polymul .................... algorithms/src/fft/polynomial/multiplier.rs:70-134 (schoolbook,
                                 = DensePolynomial::naive_mul, fft/polynomial/dense.rs:136-149)
Only small sizes are practical (pure-Python loops).
"""

# ---------------------------------------------------------------------------------------------
# Constants (curves/src/bls12_377/fr.rs:138-145, fq.rs:112-121)
# ---------------------------------------------------------------------------------------------
R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041  # |Fr|
Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177  # |Fq|
FR_MONT_R = (1 << 256) % R_MOD
FQ_MONT_R = (1 << 384) % Q_MOD
FR_TWO_ADICITY = 47
FR_TWO_ADIC_ROOT = 8065159656716812877374967518403273466521432693661810619979959746626482506078  # fr.rs:109-120
FR_GENERATOR = 22  # fr.rs:126-135
G1_B = 1  # y^2 = x^3 + 1 (g1.rs:78-91)
G1_GEN = (
    89363714989903307245735717098563574705733591463163614225748337416674727625843187853442697973404985688481508350822,
    3702177272937190650578065972808860481433820514072818216637796320125658674906330993856598323293086021583822603349,
)  # g1.rs:219-253
# Fq2 = Fq[u]/(u^2 + 5) (fq2.rs:58-69); G2: y^2 = x^3 + b' (g2.rs:92-113)
FQ2_NONRESIDUE = Q_MOD - 5
G2_B = (0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906)


# ---------------------------------------------------------------------------------------------
# limb helpers (utilities/src/biginteger/bigint_256.rs:36, bigint_384.rs:36: little-endian u64)
# ---------------------------------------------------------------------------------------------
def to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def fr_to_mont(v):
    return (v * FR_MONT_R) % R_MOD


def fr_from_mont(v):
    return (v * pow(FR_MONT_R, -1, R_MOD)) % R_MOD


def fq_to_mont(v):
    return (v * FQ_MONT_R) % Q_MOD


def fq_from_mont(v):
    return (v * pow(FQ_MONT_R, -1, Q_MOD)) % Q_MOD


# ---------------------------------------------------------------------------------------------
# G1 (affine, None == infinity)
# ---------------------------------------------------------------------------------------------
def g1_is_on_curve(p):
    if p is None:
        return True
    x, y = p
    return (y * y - (x * x * x + G1_B)) % Q_MOD == 0


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q_MOD)


def g1_add(p, q):
    """Complete affine addition (cases of affine.rs:224-273: inf, P+P, P+(-P), generic)."""
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q_MOD == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, Q_MOD) % Q_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q_MOD) % Q_MOD
    x3 = (lam * lam - x1 - x2) % Q_MOD
    y3 = (lam * (x1 - x3) - y1) % Q_MOD
    return (x3, y3)


def g1_mul(p, k):
    """MSB-first double-and-add == AffineCurve::mul_bits (affine.rs:173-182)."""
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_add(acc, acc)
        if bit == "1":
            acc = g1_add(acc, p)
    return acc


def msm_naive(bases, scalars):
    """sum_i scalars[i] * bases[i] over zip(bases, scalars) (variable_base/mod.rs:52-58,
    msm/tests.rs:27-37).  Extra bases are ignored (Appendix A.1 of SURVEY.md)."""
    acc = None
    for b, s in zip(bases, scalars):
        acc = g1_add(acc, g1_mul(b, s))
    return acc


# ---------------------------------------------------------------------------------------------
# G2 over Fq2 (elements are (c0, c1))
# ---------------------------------------------------------------------------------------------
def fq2_add(a, b):
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def fq2_mul(a, b):
    # fields/src/fp2.rs:404-410: c0 = a0 b0 + nr a1 b1 ; c1 = a0 b1 + a1 b0
    return ((a[0] * b[0] + FQ2_NONRESIDUE * a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def fq2_inv(a):
    # fields/src/fp2.rs:167-184
    n = (a[0] * a[0] - FQ2_NONRESIDUE * a[1] * a[1]) % Q_MOD
    ni = pow(n, -1, Q_MOD)
    return (a[0] * ni % Q_MOD, (-a[1]) * ni % Q_MOD)


def g2_is_on_curve(p):
    if p is None:
        return True
    x, y = p
    return fq2_sub(fq2_mul(y, y), fq2_add(fq2_mul(fq2_mul(x, x), x), G2_B)) == (0, 0)


def g2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if fq2_add(y1, y2) == (0, 0):
            return None
        xx = fq2_mul(x1, x1)
        lam = fq2_mul(fq2_add(fq2_add(xx, xx), xx), fq2_inv(fq2_add(y1, y1)))
    else:
        lam = fq2_mul(fq2_sub(y2, y1), fq2_inv(fq2_sub(x2, x1)))
    x3 = fq2_sub(fq2_sub(fq2_mul(lam, lam), x1), x2)
    y3 = fq2_sub(fq2_mul(lam, fq2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(p, k):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, p)
    return acc


def msm_naive_g2(bases, scalars):
    acc = None
    for b, s in zip(bases, scalars):
        acc = g2_add(acc, g2_mul(b, s))
    return acc


# ---------------------------------------------------------------------------------------------
# Fr domain / NTT (definitional; O(n^2))
# ---------------------------------------------------------------------------------------------
def domain_group_gen(lg_n):
    """EvaluationDomain::new -> F::get_root_of_unity (fields/src/traits/fft_field.rs:75-85):
    omega = TWO_ADIC_ROOT ^ (2^(47 - lg_n))."""
    assert 0 <= lg_n <= FR_TWO_ADICITY
    w = FR_TWO_ADIC_ROOT
    for _ in range(FR_TWO_ADICITY - lg_n):
        w = w * w % R_MOD
    return w


def ntt(x, inverse=False, coset=False):
    """NN-order transform of canonical integers, semantics of domain.rs:169-221,403-443:
      forward        X[k] = sum_j x[j] w^(jk)            (coset: x[j] *= g^j first, g = 22)
      inverse        x[j] = n^-1 sum_k X[k] w^(-jk)      (coset: then x[j] *= g^-j)"""
    n = len(x)
    lg = n.bit_length() - 1
    assert 1 << lg == n
    w = domain_group_gen(lg)
    x = [v % R_MOD for v in x]
    if not inverse:
        if coset:
            x = [v * pow(FR_GENERATOR, j, R_MOD) % R_MOD for j, v in enumerate(x)]
        return [sum(x[j] * pow(w, j * k, R_MOD) for j in range(n)) % R_MOD for k in range(n)]
    wi = pow(w, -1, R_MOD)
    ni = pow(n, -1, R_MOD)
    out = [sum(x[k] * pow(wi, j * k, R_MOD) for k in range(n)) * ni % R_MOD for j in range(n)]
    if coset:
        gi = pow(FR_GENERATOR, -1, R_MOD)
        out = [v * pow(gi, j, R_MOD) % R_MOD for j, v in enumerate(out)]
    return out


def bitrev(i, lg):
    return int(bin(i)[2:].zfill(lg)[::-1], 2) if lg else 0


def bitrev_permute(x):
    n = len(x)
    lg = n.bit_length() - 1
    return [x[bitrev(i, lg)] for i in range(n)]


def poly_mul_naive(a, b):
    """DensePolynomial::naive_mul (fft/polynomial/dense.rs:136-149)."""
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            out[i + j] = (out[i + j] + ai * bj) % R_MOD
    return out


def horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


# ---------------------------------------------------------------------------------------------
# Canonical serialisation of G1 points (curves/src/templates/macros.rs:66-140,
# fields/src/macros.rs:187-282, utilities/src/serialize/flags.rs:72-99)
# ---------------------------------------------------------------------------------------------
class SerializationError(ValueError):
    pass


def fq_sqrt(a):
    """A square root of a in Fq or None (Tonelli-Shanks; the reference's sqrt_impl, fields/src/macros.rs:85-180, may
    return the other root - callers select by comparison, affine.rs:144-148)."""
    a %= Q_MOD
    if a == 0:
        return 0
    if pow(a, (Q_MOD - 1) // 2, Q_MOD) != 1:
        return None
    s, t = 0, Q_MOD - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (Q_MOD - 1) // 2, Q_MOD) != Q_MOD - 1:
        z += 1
    c, x, b, m = pow(z, t, Q_MOD), pow(a, (t + 1) // 2, Q_MOD), pow(a, t, Q_MOD), s
    while b != 1:
        k, b2 = 0, b
        while b2 != 1:
            b2, k = b2 * b2 % Q_MOD, k + 1
        w = pow(c, 1 << (m - k - 1), Q_MOD)
        c, x, b, m = w * w % Q_MOD, x * w % Q_MOD, b * w * w % Q_MOD, k
    return x


def g1_serialize(p, compressed):
    """p = (x, y) canonical ints or None (infinity).  macros.rs:66-97."""
    if compressed:
        if p is None:
            x, flags = 0, 1 << 6
        else:
            x, flags = p[0], (1 << 7) if p[1] > (-p[1]) % Q_MOD else 0  # SWFlags::from_y_sign(y > -y)
        b = bytearray(x.to_bytes(48, "little"))
        b[47] |= flags
        return bytes(b)
    x, y, flags = (0, 1, 1 << 6) if p is None else (p[0], p[1], 0)  # Affine::zero() = (0, 1, infinity)
    b = bytearray(x.to_bytes(48, "little") + y.to_bytes(48, "little"))
    b[95] |= flags
    return bytes(b)


def _read_fq_with_flags(b):
    """deserialize_with_flags::<SWFlags> (fields/src/macros.rs:255-281, flags.rs:87-98): (value, positive, infinity)."""
    b = bytearray(b)
    pos, inf = (b[47] >> 7) & 1, (b[47] >> 6) & 1
    if pos and inf:
        raise SerializationError("UnexpectedFlags")
    b[47] &= 0x3F
    v = int.from_bytes(b, "little")
    if v >= Q_MOD:
        raise SerializationError("coordinate >= q")
    return v, bool(pos), bool(inf)


def g1_deserialize(b, compressed, validate=False):
    """macros.rs:115-140.  Returns (x, y) canonical ints or None for infinity."""
    if compressed:
        x, pos, inf = _read_fq_with_flags(b[:48])
        if inf:
            return None
        y = fq_sqrt((x * x * x + G1_B) % Q_MOD)
        if y is None:
            raise SerializationError("InvalidData")
        ny = (-y) % Q_MOD
        y = y if (y < ny) != pos else ny  # affine.rs:147
        p = (x, y)
    else:
        x = int.from_bytes(b[:48], "little")
        if x >= Q_MOD:
            raise SerializationError("coordinate >= q")
        y, _, inf = _read_fq_with_flags(b[48:96])
        if inf:
            return None
        p = (x, y)
    if validate and not (g1_is_on_curve(p) and g1_mul(p, R_MOD) is None):
        raise SerializationError("InvalidData")
    return p


def g2_serialize(p):
    """Uncompressed G2 point ((x0, x1), (y0, y1)) or None: c0 plain, c1 with the SWFlags on the last coordinate
    (fields/src/fp2.rs:425-455; curves/src/templates/macros.rs:86-95)."""
    (x0, x1), (y0, y1), flags = (((0, 0), (1, 0), 1 << 6) if p is None else (p[0], p[1], 0))
    b = bytearray(b"".join(v.to_bytes(48, "little") for v in (x0, x1, y0, y1)))
    b[191] |= flags
    return bytes(b)


def g2_deserialize(b, validate=False):
    vals = []
    for k in range(3):
        v = int.from_bytes(b[48 * k : 48 * k + 48], "little")
        if v >= Q_MOD:
            raise SerializationError("coordinate >= q")
        vals.append(v)
    y1, _, inf = _read_fq_with_flags(b[144:192])
    if inf:
        return None
    p = ((vals[0], vals[1]), (vals[2], y1))
    if validate and not (g2_is_on_curve(p) and g2_mul(p, R_MOD) is None):
        raise SerializationError("InvalidData")
    return p


# ---- Fq2 square root and the compressed G2 encoding (test oracle for serde.hip.h's compressed G2 path)
def fq2_sqrt(a):
    """`Fp2::sqrt` (fields/src/fp2.rs:208-230): complex method (eprint 2012/685, algorithm 8); None where the reference
    returns None - including a base-field element that is a non-residue in Fq (fp2.rs:210-212 only tries `c0.sqrt()`)."""
    c0, c1 = a[0] % Q_MOD, a[1] % Q_MOD
    if c1 == 0:
        r = fq_sqrt(c0)
        return None if r is None else (r, 0)
    norm = (c0 * c0 + 5 * c1 * c1) % Q_MOD  # c0^2 - nonresidue c1^2, nonresidue = -5 (fq2.rs:58-69)
    alpha = fq_sqrt(norm)
    if alpha is None:  # legendre(norm) = QNR
        return None
    two_inv = pow(2, Q_MOD - 2, Q_MOD)
    delta = (alpha + c0) * two_inv % Q_MOD
    if pow(delta, (Q_MOD - 1) // 2, Q_MOD) == Q_MOD - 1:  # delta.legendre().is_qnr()
        delta = (delta - alpha) % Q_MOD
    r0 = fq_sqrt(delta)
    if r0 is None or r0 == 0:
        return None
    return (r0, c1 * two_inv % Q_MOD * pow(r0, Q_MOD - 2, Q_MOD) % Q_MOD)


def fq2_gt(a, b):
    """`Ord for Fp2` (fp2.rs:240-250): lexicographic, c1 first."""
    return (a[1], a[0]) > (b[1], b[0])


def g2_serialize_compressed(p):
    """x.c0 plain, x.c1 with the SWFlags (macros.rs:66-85; `from_y_sign(y > -y)` in the order of fp2.rs:240-250)."""
    if p is None:
        x, flags = (0, 0), 1 << 6
    else:
        ny = ((-p[1][0]) % Q_MOD, (-p[1][1]) % Q_MOD)
        x, flags = p[0], (1 << 7) if fq2_gt(p[1], ny) else 0
    b = bytearray(x[0].to_bytes(48, "little") + x[1].to_bytes(48, "little"))
    b[95] |= flags
    return bytes(b)


def g2_deserialize_compressed(b, validate=False):
    """macros.rs:115-127 + affine.rs:140-150 over Fq2."""
    x0 = int.from_bytes(b[:48], "little")
    if x0 >= Q_MOD:
        raise SerializationError("coordinate >= q")
    x1, pos, inf = _read_fq_with_flags(b[48:96])
    if inf:
        return None
    x = (x0, x1)
    y = fq2_sqrt(fq2_add(fq2_mul(fq2_mul(x, x), x), G2_B))
    if y is None:
        raise SerializationError("InvalidData")
    ny = ((-y[0]) % Q_MOD, (-y[1]) % Q_MOD)
    y = y if (fq2_gt(ny, y) != pos) else ny  # `if (y < negy) ^ greatest { y } else { negy }`
    p = (x, y)
    if validate and not (g2_is_on_curve(p) and g2_mul(p, R_MOD) is None):
        raise SerializationError("InvalidData")
    return p
