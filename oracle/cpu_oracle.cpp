// =====================================================================================
// oracle/cpu_oracle.cpp -- CPU restatement of snarkVM's MSM / NTT hot path (BLS12-377)
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load the
// library built from this file.  Nothing under snarkvm_amd/ links, imports or calls it.
//
// It is a C++ restatement (64-bit limbs, unsigned __int128, OpenMP standing in for
// rayon) of the reference's *CPU* algorithms, written from the reference's behaviour; each
// function cites the file:line (relative to the snarkVM v1.0.0 checkout) it follows.
// The Rust toolchain is absent from the build image, so this restatement doubles as the
// "reference rayon CPU path" for timing (bench.py labels it kind="port").
//
// Parity pinning (tests/test_oracle.py): field constants, 2-adic roots table, the size-8
// Varuna domain, KAT-iNTT8 (z_lde), KAT-polymul16 (h_0), generator / SRS points -- all from
// tests/golden/ (extracted from the reference by tests/golden/make_golden.py) -- plus the
// reference's own differential properties (MSM == sum mul_bits, NTT == Horner) and an
// independent Python big-int implementation (oracle/pyref.py).
// =====================================================================================
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

// -------------------------------------------------------------------------------------
// Prime fields in Montgomery form.
//   Fp256 <- fields/src/fp_256.rs   (Fr; params curves/src/bls12_377/fr.rs:56-193)
//   Fp384 <- fields/src/fp_384.rs   (Fq; params curves/src/bls12_377/fq.rs:33-177)
// Elements are always fully reduced (< p): fp_256.rs:61-65 `reduce`.
// -------------------------------------------------------------------------------------
struct FrParams {
    static constexpr int N = 4;
    static constexpr uint64_t MODULUS[4] = {725501752471715841ull, 6461107452199829505ull, 6968279316240510977ull,
                                            1345280370688173398ull};                       // fr.rs:138-145
    static constexpr uint64_t R[4] = {9015221291577245683ull, 8239323489949974514ull, 1646089257421115374ull,
                                      958099254763297437ull};                              // fr.rs:158-163
    static constexpr uint64_t R2[4] = {2726216793283724667ull, 14712177743343147295ull, 12091039717619697043ull,
                                       81024008013859129ull};                              // fr.rs:164-170
    static constexpr uint64_t INV = 725501752471715839ull;                                  // fr.rs:137
};
struct FqParams {
    static constexpr int N = 6;
    static constexpr uint64_t MODULUS[6] = {0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull,
                                            0x1a22d9f300f5138full, 0xc63b05c06ca1493bull,
                                            0x1ae3a4617c510eaull};                         // fq.rs:112-121
    static constexpr uint64_t R[6] = {202099033278250856ull, 5854854902718660529ull, 11492539364873682930ull,
                                      8885205928937022213ull, 5545221690922665192ull,
                                      39800542322357402ull};                               // fq.rs:134-141
    static constexpr uint64_t R2[6] = {0xb786686c9400cd22ull, 0x329fcaab00431b1ull, 0x22a5f11162d6b46dull,
                                       0xbfdf7d03827dc3acull, 0x837e92f041790bf9ull,
                                       0x6dfccb1e914b88ull};                               // fq.rs:142-150
    static constexpr uint64_t INV = 9586122913090633727ull;                                 // fq.rs:111
};
constexpr uint64_t FrParams::MODULUS[4], FrParams::R[4], FrParams::R2[4];
constexpr uint64_t FqParams::MODULUS[6], FqParams::R[6], FqParams::R2[6];

template <int N>
static inline bool big_geq(const uint64_t* a, const uint64_t* b) {
    for (int i = N - 1; i >= 0; i--) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
template <int N>
static inline uint64_t big_add(uint64_t* a, const uint64_t* b) {  // a += b, returns carry
    u128 c = 0;
    for (int i = 0; i < N; i++) {
        c += (u128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
template <int N>
static inline uint64_t big_sub(uint64_t* a, const uint64_t* b) {  // a -= b, returns borrow
    uint64_t br = 0;
    for (int i = 0; i < N; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        a[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
template <int N>
static inline void big_div2(uint64_t* a) {
    for (int i = 0; i < N - 1; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
    a[N - 1] >>= 1;
}
template <int N>
static inline bool big_is_one(const uint64_t* a) {
    if (a[0] != 1) return false;
    for (int i = 1; i < N; i++)
        if (a[i]) return false;
    return true;
}

template <class P>
struct Fp {
    static constexpr int N = P::N;
    uint64_t l[N];

    static Fp zero() {
        Fp r;
        memset(r.l, 0, sizeof r.l);
        return r;
    }
    static Fp one() {
        Fp r;
        memcpy(r.l, P::R, sizeof r.l);
        return r;
    }
    bool is_zero() const {
        uint64_t o = 0;
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    bool is_one() const { return memcmp(l, P::R, sizeof l) == 0; }
    bool operator==(const Fp& o) const { return memcmp(l, o.l, sizeof l) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }

    // fp_256.rs:61-65 / fp_384.rs:61-65
    void reduce() {
        if (big_geq<N>(l, P::MODULUS)) big_sub<N>(l, P::MODULUS);
    }
    // fp_256.rs:730-737 (add_assign): add_nocarry then reduce (modulus has spare bits)
    Fp operator+(const Fp& o) const {
        Fp r = *this;
        big_add<N>(r.l, o.l);
        r.reduce();
        return r;
    }
    // fp_256.rs:740-750 (sub_assign): if other > self add modulus first
    Fp operator-(const Fp& o) const {
        Fp r = *this;
        if (big_sub<N>(r.l, o.l)) big_add<N>(r.l, P::MODULUS);
        return r;
    }
    Fp& operator+=(const Fp& o) { return *this = *this + o; }
    Fp& operator-=(const Fp& o) { return *this = *this - o; }
    // fp_256.rs:670-684 (neg)
    Fp neg() const {
        if (is_zero()) return *this;
        Fp r;
        memcpy(r.l, P::MODULUS, sizeof r.l);
        big_sub<N>(r.l, l);
        return r;
    }
    // fp_256.rs:232-237 (double_in_place): mul2 then reduce
    Fp dbl() const { return *this + *this; }

    // fp_256.rs:752-818 / fp_384.rs:769-899 (mul_assign): CIOS Montgomery multiplication,
    // result fully reduced.  Interleaved multiply/reduce per limb of `b`.
    Fp operator*(const Fp& b) const {
        uint64_t t[N + 2];
        memset(t, 0, sizeof t);
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) {
                c += (u128)l[j] * b.l[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (uint64_t)c;
            t[N + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * P::INV;
            c = (u128)m * P::MODULUS[0] + t[0];
            c >>= 64;
            for (int j = 1; j < N; j++) {
                c += (u128)m * P::MODULUS[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        Fp r;
        memcpy(r.l, t, sizeof r.l);
        if (t[N] || big_geq<N>(r.l, P::MODULUS)) big_sub<N>(r.l, P::MODULUS);
        return r;
    }
    Fp& operator*=(const Fp& o) { return *this = *this * o; }
    // fp_256.rs:252-287 (square_in_place) -- same value as self*self
    Fp sqr() const { return *this * *this; }

    // fp_256.rs:362-377 (from_bigint): multiply by R2; fp_256.rs:380-413 (to_bigint): mont_reduce
    static Fp from_bigint(const uint64_t* v) {
        Fp a, r2;
        memcpy(a.l, v, sizeof a.l);
        memcpy(r2.l, P::R2, sizeof r2.l);
        return a * r2;
    }
    void to_bigint(uint64_t* out) const {
        Fp o;
        memset(o.l, 0, sizeof o.l);
        o.l[0] = 1;
        Fp r = *this * o;
        memcpy(out, r.l, sizeof r.l);
    }
    static Fp from_u64(uint64_t v) {
        uint64_t t[N] = {0};
        t[0] = v;
        return from_bigint(t);
    }

    // fp_256.rs:290-340 (inverse): binary extended Euclid (Guajardo et al. Alg. 16),
    // started from b = R2 so the result is already in Montgomery form.
    Fp inverse() const {
        assert(!is_zero());
        uint64_t u[N], v[N];
        memcpy(u, l, sizeof u);
        memcpy(v, P::MODULUS, sizeof v);
        Fp b, c = zero();
        memcpy(b.l, P::R2, sizeof b.l);
        while (!big_is_one<N>(u) && !big_is_one<N>(v)) {
            while ((u[0] & 1) == 0) {
                big_div2<N>(u);
                if (b.l[0] & 1) big_add<N>(b.l, P::MODULUS);
                big_div2<N>(b.l);
            }
            while ((v[0] & 1) == 0) {
                big_div2<N>(v);
                if (c.l[0] & 1) big_add<N>(c.l, P::MODULUS);
                big_div2<N>(c.l);
            }
            if (!big_geq<N>(v, u)) {  // v < u
                big_sub<N>(u, v);
                b -= c;
            } else {
                big_sub<N>(v, u);
                c -= b;
            }
        }
        return big_is_one<N>(u) ? b : c;
    }
    // Field::pow, little-endian u64 exponent words (fields/src/traits/field.rs `pow`)
    Fp pow(const uint64_t* e, int words) const {
        Fp res = one();
        bool started = false;
        for (int w = words - 1; w >= 0; w--)
            for (int b = 63; b >= 0; b--) {
                if (started) res = res.sqr();
                if ((e[w] >> b) & 1) {
                    res *= *this;
                    started = true;
                }
            }
        return res;
    }
    // Field::half (fp_256.rs:158-165): (p+1)/2 as a field element
    static Fp half() {
        uint64_t t[N];
        memcpy(t, P::MODULUS, sizeof t);
        uint64_t one1[N] = {1};
        big_add<N>(t, one1);
        big_div2<N>(t);
        return from_bigint(t);
    }
};

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

// fields/src/fp2.rs:57-60 ; Fq2 = Fq[u]/(u^2 - NONRESIDUE), NONRESIDUE = -5 (fq2.rs:58-69)
struct Fq2 {
    Fq c0, c1;
    static Fq nonresidue() {
        static const Fq nr = Fq::from_u64(5).neg();
        return nr;
    }
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool is_one() const { return c0.is_one() && c1.is_zero(); }
    bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fq2& o) const { return !(*this == o); }
    Fq2 operator+(const Fq2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2& operator+=(const Fq2& o) { return *this = *this + o; }
    Fq2& operator-=(const Fq2& o) { return *this = *this - o; }
    Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    // fp2.rs:404-410 (mul_assign via sum_of_products): c0 = a0 b0 + nr a1 b1, c1 = a0 b1 + a1 b0
    Fq2 operator*(const Fq2& o) const {
        return {c0 * o.c0 + nonresidue() * (c1 * o.c1), c0 * o.c1 + c1 * o.c0};
    }
    Fq2& operator*=(const Fq2& o) { return *this = *this * o; }
    Fq2 sqr() const { return *this * *this; }  // fp2.rs:149-165, same value
    // fp2.rs:167-184 (inverse): (c0 - c1 u) / (c0^2 - nr c1^2)
    Fq2 inverse() const {
        Fq n = c0.sqr() - nonresidue() * c1.sqr();
        Fq ni = n.inverse();
        return {c0 * ni, (c1 * ni).neg()};
    }
    static Fq2 half() { return {Fq::half(), Fq::zero()}; }
};

// -------------------------------------------------------------------------------------
// Short Weierstrass curves, a = 0.
//   Affine     <- curves/src/templates/short_weierstrass_jacobian/affine.rs:42-46
//   Projective <- .../projective.rs:37-41 (Jacobian; infinity <=> z == 0, projective.rs:52-60)
// In-memory layout equals the Rust one: Affine<G1> 104 B, Projective<G1> 144 B,
// Affine<G2> 200 B, Projective<G2> 288 B.
// -------------------------------------------------------------------------------------
template <class F>
struct Affine {
    F x, y;
    uint8_t infinity;
    uint8_t pad[7];
    static Affine zero() {  // affine.rs:55-60
        Affine a;
        a.x = F::zero();
        a.y = F::one();
        a.infinity = 1;
        memset(a.pad, 0, sizeof a.pad);
        return a;
    }
    bool is_zero() const { return infinity != 0; }
};
template <class F>
struct Projective {
    F x, y, z;
    static Projective zero() { return {F::zero(), F::one(), F::zero()}; }  // projective.rs:49-54
    bool is_zero() const { return z.is_zero(); }

    // projective.rs:302-339, a == 0 branch
    void double_in_place() {
        if (is_zero()) return;
        F a = x.sqr();
        F b = y.sqr();
        F c = b.sqr();
        F d = ((x + b).sqr() - a - c).dbl();
        F e = a + a.dbl();
        F f = e.sqr();
        z = (z * y).dbl();
        x = f - d.dbl();
        y = (d - x) * e - c.dbl().dbl().dbl();
    }
    // projective.rs:222-291 (madd-2007-bl, with the equal-point fallback to doubling)
    void add_assign_mixed(const Affine<F>& o) {
        if (o.is_zero()) return;
        if (is_zero()) {
            x = o.x;
            y = o.y;
            z = F::one();
            return;
        }
        F z1z1 = z.sqr();
        F u2 = o.x * z1z1;
        F s2 = (o.y * z) * z1z1;
        if (x == u2 && y == s2) {
            double_in_place();
            return;
        }
        F h = u2 - x;
        F hh = h.sqr();
        F i = hh.dbl().dbl();
        F j = h * i;
        F r = (s2 - y).dbl();
        F v = x * i;
        F x3 = r.sqr() - j - v.dbl();
        // Y3 = r*(V-X3) - 2*Y1*J   (sum_of_products, same value)
        F y3 = r * (v - x3) - y.dbl() * j;
        F z3 = (z + h).sqr() - z1z1 - hh;
        x = x3;
        y = y3;
        z = z3;
    }
    // projective.rs:407-468 (add-2007-bl)
    void add_assign(const Projective& o) {
        if (is_zero()) {
            *this = o;
            return;
        }
        if (o.is_zero()) return;
        F z1z1 = z.sqr();
        F z2z2 = o.z.sqr();
        F u1 = x * z2z2;
        F u2 = o.x * z1z1;
        F s1 = y * o.z * z2z2;
        F s2 = o.y * z * z1z1;
        if (u1 == u2 && s1 == s2) {
            double_in_place();
            return;
        }
        F h = u2 - u1;
        F i = h.dbl().sqr();
        F j = h * i;
        F r = (s2 - s1).dbl();
        F v = u1 * i;
        F x3 = r.sqr() - j - v.dbl();
        F y3 = r * (v - x3) - s1.dbl() * j;
        F z3 = ((z + o.z).sqr() - z1z1 - z2z2) * h;
        x = x3;
        y = y3;
        z = z3;
    }
    // affine.rs:331-353 (From<Projective> for Affine)
    Affine<F> to_affine() const {
        if (is_zero()) return Affine<F>::zero();
        Affine<F> a = Affine<F>::zero();
        a.infinity = 0;
        if (z.is_one()) {
            a.x = x;
            a.y = y;
            return a;
        }
        F zinv = z.inverse();
        F zinv2 = zinv.sqr();
        a.x = x * zinv2;
        a.y = y * (zinv2 * zinv);
        return a;
    }
};

// affine.rs:173-182 (mul_bits over BitIteratorBE of a 256-bit integer, leading zeros skipped)
template <class F>
static Projective<F> mul_bits(const Affine<F>& base, const uint64_t* scalar, int words = 4) {
    Projective<F> out = Projective<F>::zero();
    bool started = false;
    for (int w = words - 1; w >= 0; w--)
        for (int b = 63; b >= 0; b--) {
            bool bit = (scalar[w] >> b) & 1;
            if (!started && !bit) continue;
            started = true;
            out.double_in_place();
            if (bit) out.add_assign_mixed(base);
        }
    return out;
}

// affine.rs:224-254 (batch_add_loop_1)
template <class F>
static inline void batch_add_loop_1(Affine<F>& a, Affine<F>& b, const F& half, F& inversion_tmp) {
    if (a.is_zero() || b.is_zero()) {
    } else if (a.x == b.x) {
        if (a.y == b.y) {
            F x_sq = b.x.sqr();
            b.x -= b.y;                     // x - y
            a.x = b.y.dbl();                // denominator = 2y
            a.y = x_sq.dbl() + x_sq;        // numerator = 3x^2 (+ a, a == 0)
            b.y -= a.y * half;              // y - (3x^2)/2
            a.y *= inversion_tmp;           // (3x^2) * tmp
            inversion_tmp *= a.x;           // update tmp
        } else {
            a.infinity = 1;
            b.infinity = 1;
        }
    } else {
        a.x -= b.x;  // denominator = x1 - x2
        a.y -= b.y;  // numerator = y1 - y2
        a.y *= inversion_tmp;
        inversion_tmp *= a.x;
    }
}
// affine.rs:259-273 (batch_add_loop_2)
template <class F>
static inline void batch_add_loop_2(Affine<F>& a, const Affine<F>& b, F& inversion_tmp) {
    if (a.is_zero()) {
        a = b;
    } else if (!b.is_zero()) {
        F lambda = a.y * inversion_tmp;
        inversion_tmp *= a.x;
        a.x += b.x.dbl();
        a.x = lambda.sqr() - a.x;
        a.y = lambda * (b.x - a.x) - b.y;
    }
}

// -------------------------------------------------------------------------------------
// MSM
// -------------------------------------------------------------------------------------
// fft/domain.rs:64-72 (log2 = ceiling log2), msm/mod.rs:29-32 (ln_without_floats)
static inline uint32_t ceil_log2(size_t x) {
    if (x == 0) return 0;
    uint32_t lz = __builtin_clzll((unsigned long long)x);
    if ((x & (x - 1)) == 0) return 63 - lz;
    return 64 - lz;
}
static inline size_t ln_without_floats(size_t a) { return (size_t)(ceil_log2(a) * 69 / 100); }

// digit extraction: batched.rs:341-348 / standard.rs:29-33 -- divn(w_start) then limb0 % 2^c
static inline uint64_t window_digit(const uint64_t* s, size_t w_start, size_t c) {
    size_t limb = w_start / 64, sh = w_start % 64;
    uint64_t v = 0;
    if (limb < 4) {
        v = s[limb] >> sh;
        if (sh && limb + 1 < 4) v |= s[limb + 1] << (64 - sh);
    }
    return v & ((1ull << c) - 1);
}

struct BucketPosition {  // batched.rs:27-30
    uint32_t bucket_index, scalar_index;
};
static inline size_t batch_size_for(size_t msm_size) { return msm_size < 500000 ? 300 : 3000; }  // batched.rs:56-73 (x86_64)

// batched.rs:131-172 (batch_add_write)
template <class F>
static void batch_add_write(const Affine<F>* bases, const std::vector<std::pair<uint32_t, uint32_t>>& index,
                            std::vector<Affine<F>>& addition_result, std::vector<Affine<F>>& scratch,
                            std::vector<uint8_t>& scratch_some) {
    F inversion_tmp = F::one();
    const F half = F::half();
    size_t first = addition_result.size();
    for (auto& in : index) {
        if (in.second == ~0u) {
            addition_result.push_back(bases[in.first]);
            scratch.push_back(Affine<F>::zero());
            scratch_some.push_back(0);
        } else {
            Affine<F> a = bases[in.first], b = bases[in.second];
            batch_add_loop_1(a, b, half, inversion_tmp);
            addition_result.push_back(a);
            scratch.push_back(b);
            scratch_some.push_back(1);
        }
    }
    inversion_tmp = inversion_tmp.inverse();
    for (size_t k = index.size(); k-- > 0;) {
        if (scratch_some[k]) batch_add_loop_2(addition_result[first + k], scratch[k], inversion_tmp);
    }
    scratch.clear();
    scratch_some.clear();
}
// batched.rs:78-122 (batch_add_in_place_same_slice)
template <class F>
static void batch_add_in_place_same_slice(std::vector<Affine<F>>& bases,
                                          const std::vector<std::pair<uint32_t, uint32_t>>& index) {
    F inversion_tmp = F::one();
    const F half = F::half();
    for (auto& in : index) batch_add_loop_1(bases[in.first], bases[in.second], half, inversion_tmp);
    inversion_tmp = inversion_tmp.inverse();
    for (size_t k = index.size(); k-- > 0;) {
        Affine<F> b = bases[index[k].second];
        batch_add_loop_2(bases[index[k].first], b, inversion_tmp);
    }
}

// batched.rs:175-325 (batch_add)
template <class F>
static std::vector<Affine<F>> batch_add(size_t num_buckets, const Affine<F>* bases, size_t nbases,
                                        std::vector<BucketPosition>& bp) {
    assert(nbases >= bp.size());
    assert(nbases > 0);
    const size_t batch_size = batch_size_for(nbases);
    std::sort(bp.begin(), bp.end(),
              [](const BucketPosition& a, const BucketPosition& b) { return a.bucket_index < b.bucket_index; });

    size_t num_scalars = bp.size();
    bool all_ones = true;
    size_t new_scalar_length = 0, global_counter = 0, local_counter = 1, number_of_bases_in_batch = 0;
    std::vector<std::pair<uint32_t, uint32_t>> instr;
    instr.reserve(batch_size);
    std::vector<Affine<F>> new_bases;
    new_bases.reserve(nbases);
    std::vector<Affine<F>> scratch;
    std::vector<uint8_t> scratch_some;

    // first pass: results of the first addition tree level are written to new_bases (:196-257)
    while (global_counter < num_scalars) {
        uint32_t current_bucket = bp[global_counter].bucket_index;
        while (global_counter + 1 < num_scalars && bp[global_counter + 1].bucket_index == current_bucket) {
            global_counter++;
            local_counter++;
        }
        if (current_bucket >= (uint32_t)num_buckets) {
            local_counter = 1;
        } else if (local_counter > 1) {
            if (local_counter > 2) all_ones = false;
            bool is_odd = local_counter % 2 == 1;
            size_t half = local_counter / 2;
            for (size_t i = 0; i < half; i++) {
                instr.push_back({bp[global_counter - (local_counter - 1) + 2 * i].scalar_index,
                                 bp[global_counter - (local_counter - 1) + 2 * i + 1].scalar_index});
                bp[new_scalar_length + i] = {current_bucket, (uint32_t)(new_scalar_length + i)};
            }
            if (is_odd) {
                instr.push_back({bp[global_counter].scalar_index, ~0u});
                bp[new_scalar_length + half] = {current_bucket, (uint32_t)(new_scalar_length + half)};
            }
            new_scalar_length += half + (local_counter % 2);
            number_of_bases_in_batch += half;
            local_counter = 1;
            if (number_of_bases_in_batch >= batch_size / 2) {
                batch_add_write(bases, instr, new_bases, scratch, scratch_some);
                instr.clear();
                number_of_bases_in_batch = 0;
            }
        } else {
            instr.push_back({bp[global_counter].scalar_index, ~0u});
            bp[new_scalar_length] = {current_bucket, (uint32_t)new_scalar_length};
            new_scalar_length += 1;
        }
        global_counter++;
    }
    if (!instr.empty()) {
        batch_add_write(bases, instr, new_bases, scratch, scratch_some);
        instr.clear();
    }
    global_counter = 0;
    number_of_bases_in_batch = 0;
    local_counter = 1;
    num_scalars = new_scalar_length;
    new_scalar_length = 0;

    // remaining levels in place (:260-318)
    while (!all_ones) {
        all_ones = true;
        while (global_counter < num_scalars) {
            uint32_t current_bucket = bp[global_counter].bucket_index;
            while (global_counter + 1 < num_scalars && bp[global_counter + 1].bucket_index == current_bucket) {
                global_counter++;
                local_counter++;
            }
            if (current_bucket >= (uint32_t)num_buckets) {
                local_counter = 1;
            } else if (local_counter > 1) {
                if (local_counter != 2) all_ones = false;
                bool is_odd = local_counter % 2 == 1;
                size_t half = local_counter / 2;
                for (size_t i = 0; i < half; i++) {
                    instr.push_back({bp[global_counter - (local_counter - 1) + 2 * i].scalar_index,
                                     bp[global_counter - (local_counter - 1) + 2 * i + 1].scalar_index});
                    bp[new_scalar_length + i] = bp[global_counter - (local_counter - 1) + 2 * i];
                }
                if (is_odd) bp[new_scalar_length + half] = bp[global_counter];
                new_scalar_length += half + (local_counter % 2);
                number_of_bases_in_batch += half;
                local_counter = 1;
                if (number_of_bases_in_batch >= batch_size / 2) {
                    batch_add_in_place_same_slice(new_bases, instr);
                    instr.clear();
                    number_of_bases_in_batch = 0;
                }
            } else {
                bp[new_scalar_length] = bp[global_counter];
                new_scalar_length += 1;
            }
            global_counter++;
        }
        if (!instr.empty()) {
            batch_add_in_place_same_slice(new_bases, instr);
            instr.clear();
        }
        global_counter = 0;
        number_of_bases_in_batch = 0;
        local_counter = 1;
        num_scalars = new_scalar_length;
        new_scalar_length = 0;
    }

    std::vector<Affine<F>> res(num_buckets, Affine<F>::zero());
    for (size_t i = 0; i < num_scalars; i++) res[bp[i].bucket_index] = new_bases[bp[i].scalar_index];
    return res;
}

// batched.rs:328-364 (batched_window)
template <class F>
static Projective<F> batched_window(const Affine<F>* bases, size_t nbases, const uint64_t* scalars, size_t nscalars,
                                    size_t w_start, size_t c) {
    size_t window_size = (w_start % c) != 0 ? (w_start % c) : c;
    size_t num_buckets = ((size_t)1 << window_size) - 1;
    std::vector<BucketPosition> bp(nscalars);
    for (size_t i = 0; i < nscalars; i++) {
        int32_t d = (int32_t)window_digit(scalars + 4 * i, w_start, c);
        bp[i] = {(uint32_t)(d - 1), (uint32_t)i};  // digit 0 wraps to u32::MAX and is skipped
    }
    std::vector<Affine<F>> buckets = batch_add<F>(num_buckets, bases, nbases, bp);
    Projective<F> res = Projective<F>::zero(), running = Projective<F>::zero();
    for (size_t k = buckets.size(); k-- > 0;) {
        running.add_assign_mixed(buckets[k]);
        res.add_assign(running);
    }
    return res;
}

static const size_t FR_SIZE_IN_BITS = 253;  // fr.rs:146-147 MODULUS_BITS

// msm/variable_base/batched.rs:366-415 (msm); one rayon task per window -> one OpenMP task per window
template <class F>
static Projective<F> batched_msm(const Affine<F>* bases, size_t nbases, const uint64_t* scalars, size_t nscalars) {
    if (nbases < 15) {  // :367-387, shared double-and-add; zip truncates to the shorter of the two
        size_t n = std::min(nbases, nscalars);
        Projective<F> sum = Projective<F>::zero();
        bool encountered_one = false;
        for (int bit = (int)FR_SIZE_IN_BITS - 1; bit >= 0; bit--) {
            if (encountered_one) sum.double_in_place();
            for (size_t i = 0; i < n; i++) {
                if ((scalars[4 * i + bit / 64] >> (bit % 64)) & 1) {
                    sum.add_assign_mixed(bases[i]);
                    encountered_one = true;
                }
            }
        }
        return sum;
    }
    size_t c = nscalars < 32 ? 1 : ln_without_floats(nscalars) + 2;
    std::vector<size_t> starts;
    for (size_t w = 0; w < FR_SIZE_IN_BITS; w += c) starts.push_back(w);
    std::vector<Projective<F>> window_sums(starts.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < starts.size(); k++)
        window_sums[k] = batched_window<F>(bases, nbases, scalars, nscalars, starts[k], c);
    // :404-413: fold from the highest window; `c` doublings after each add (window_size == c always)
    Projective<F> total = Projective<F>::zero();
    for (size_t k = window_sums.size(); k-- > 1;) {
        total.add_assign(window_sums[k]);
        for (size_t d = 0; d < c; d++) total.double_in_place();
    }
    total.add_assign(window_sums[0]);
    return total;
}

// msm/variable_base/standard.rs:43-77 (standard_window) + :23-41 (update_buckets)
static inline bool big256_is_one(const uint64_t* s) { return s[0] == 1 && !s[1] && !s[2] && !s[3]; }
static inline bool big256_gt_one(const uint64_t* s) { return s[1] || s[2] || s[3] || s[0] > 1; }
template <class F>
static Projective<F> standard_window(const Affine<F>* bases, const uint64_t* scalars, size_t n, size_t w_start,
                                     size_t c) {
    Projective<F> res = Projective<F>::zero();
    if (w_start == 0)
        for (size_t i = 0; i < n; i++)
            if (big256_is_one(scalars + 4 * i)) res.add_assign_mixed(bases[i]);
    size_t window_size = (w_start % c) != 0 ? (w_start % c) : c;
    std::vector<Projective<F>> buckets(((size_t)1 << window_size) - 1, Projective<F>::zero());
    for (size_t i = 0; i < n; i++) {
        if (!big256_gt_one(scalars + 4 * i)) continue;
        uint64_t d = window_digit(scalars + 4 * i, w_start, c);
        if (d != 0) buckets[d - 1].add_assign_mixed(bases[i]);
    }
    Projective<F> running = Projective<F>::zero();
    for (size_t k = buckets.size(); k-- > 0;) {
        running.add_assign(buckets[k]);
        res.add_assign(running);
    }
    return res;
}
// standard.rs:79-105 (msm)
template <class F>
static Projective<F> standard_msm(const Affine<F>* bases, size_t nbases, const uint64_t* scalars, size_t nscalars) {
    size_t n = std::min(nbases, nscalars);  // `zip`
    size_t c = nscalars < 32 ? 1 : ln_without_floats(nscalars) + 2;
    std::vector<size_t> starts;
    for (size_t w = 0; w < FR_SIZE_IN_BITS; w += c) starts.push_back(w);
    std::vector<Projective<F>> window_sums(starts.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < starts.size(); k++) window_sums[k] = standard_window<F>(bases, scalars, n, starts[k], c);
    Projective<F> total = Projective<F>::zero();
    for (size_t k = window_sums.size(); k-- > 1;) {
        total.add_assign(window_sums[k]);
        for (size_t d = 0; d < c; d++) total.double_in_place();
    }
    total.add_assign(window_sums[0]);
    return total;
}
// variable_base/mod.rs:52-58 (msm_naive) / msm/tests.rs:27-37
template <class F>
static Projective<F> naive_msm(const Affine<F>* bases, const uint64_t* scalars, size_t n) {
    Projective<F> acc = Projective<F>::zero();
    for (size_t i = 0; i < n; i++) acc.add_assign(mul_bits<F>(bases[i], scalars + 4 * i));
    return acc;
}

// -------------------------------------------------------------------------------------
// NTT over Fr  (algorithms/src/fft/domain.rs)
// -------------------------------------------------------------------------------------
static const uint32_t FR_TWO_ADICITY = 47;  // fr.rs:109
static Fr fr_two_adic_root() {              // fr.rs:115-120 (Montgomery limbs)
    Fr w;
    const uint64_t v[4] = {12646347781564978760ull, 6783048705277173164ull, 268534165941069093ull,
                           1121515446318641358ull};
    memcpy(w.l, v, sizeof v);
    return w;
}
static Fr fr_generator() {  // fr.rs:126-135, GENERATOR = 22
    Fr g;
    const uint64_t v[4] = {2984901390528151251ull, 10561528701063790279ull, 5476750214495080041ull,
                           898978044469942640ull};
    memcpy(g.l, v, sizeof v);
    return g;
}

struct EvaluationDomain {  // domain.rs:83-98
    uint64_t size;
    uint32_t log_size_of_group;
    Fr size_as_field_element, size_inv, group_gen, group_gen_inv, generator_inv;
    // domain.rs:118-147 (new) ; group_gen via fields/src/traits/fft_field.rs:75-85
    static bool make(size_t num_coeffs, EvaluationDomain& d) {
        uint64_t size = 1;
        while (size < num_coeffs) size <<= 1;
        uint32_t lg = __builtin_ctzll(size);
        if (lg > FR_TWO_ADICITY) return false;
        Fr omega = fr_two_adic_root();
        for (uint32_t i = lg; i < FR_TWO_ADICITY; i++) omega = omega.sqr();
        d.size = size;
        d.log_size_of_group = lg;
        d.size_as_field_element = Fr::from_u64(size);
        d.size_inv = d.size_as_field_element.inverse();
        d.group_gen = omega;
        d.group_gen_inv = omega.inverse();
        d.generator_inv = fr_generator().inverse();
        return true;
    }
};

// domain.rs:594-648 (roots_of_unity): first size/2 powers of `root` (any schedule gives identical values)
static std::vector<Fr> roots_of_unity(uint64_t size, const Fr& root) {
    size_t half = size / 2;
    std::vector<Fr> r(half ? half : 0);
    if (!half) return r;
    const size_t CH = 1 << 10;
    size_t nch = (half + CH - 1) / CH;
#pragma omp parallel for schedule(static)
    for (size_t ci = 0; ci < nch; ci++) {
        uint64_t e[1] = {(uint64_t)(ci * CH)};
        Fr cur = root.pow(e, 1);
        size_t end = std::min(half, (ci + 1) * CH);
        for (size_t i = ci * CH; i < end; i++) {
            r[i] = cur;
            cur *= root;
        }
    }
    return r;
}

// domain.rs:789-804 (bitrev, derange_helper) -- serial in the reference
static inline uint64_t bitrev64(uint64_t a, uint32_t log_len) {
    if (log_len == 0) return 0;
    uint64_t r = 0;
    for (uint32_t i = 0; i < 64; i++) r |= ((a >> i) & 1) << (63 - i);
    return r >> (64 - log_len);
}
static void derange(Fr* xi, size_t n, uint32_t log_len) {
    if (n < 3) return;
    for (uint64_t idx = 1; idx < n - 1; idx++) {
        uint64_t r = bitrev64(idx, log_len);
        if (idx < r) std::swap(xi[idx], xi[r]);
    }
}

static const size_t MIN_NUM_CHUNKS_FOR_COMPACTION = 1 << 7;     // domain.rs:778
static const size_t MIN_GAP_SIZE_FOR_PARALLELISATION = 1 << 10;  // domain.rs:782

// domain.rs:651-656 (butterfly_fn_io) / :659-664 (butterfly_fn_oi) / :667-688 (apply_butterfly)
template <bool IO>
static void apply_butterfly(Fr* xi, size_t n, const Fr* roots, size_t step, size_t chunk_size, size_t num_chunks,
                            size_t max_threads, size_t gap) {
    bool inner_par = gap > MIN_GAP_SIZE_FOR_PARALLELISATION && num_chunks < max_threads;
    if (inner_par) {
        for (size_t c0 = 0; c0 < n; c0 += chunk_size) {
            Fr* lo = xi + c0;
            Fr* hi = lo + gap;
#pragma omp parallel for schedule(static)
            for (size_t k = 0; k < gap; k++) {
                const Fr& w = roots[k * step];
                if (IO) {
                    Fr neg = lo[k] - hi[k];
                    lo[k] += hi[k];
                    hi[k] = neg * w;
                } else {
                    hi[k] *= w;
                    Fr neg = lo[k] - hi[k];
                    lo[k] += hi[k];
                    hi[k] = neg;
                }
            }
        }
    } else {
#pragma omp parallel for schedule(static)
        for (size_t ci = 0; ci < num_chunks; ci++) {
            Fr* lo = xi + ci * chunk_size;
            Fr* hi = lo + gap;
            for (size_t k = 0; k < gap; k++) {
                const Fr& w = roots[k * step];
                if (IO) {
                    Fr neg = lo[k] - hi[k];
                    lo[k] += hi[k];
                    hi[k] = neg * w;
                } else {
                    hi[k] *= w;
                    Fr neg = lo[k] - hi[k];
                    lo[k] += hi[k];
                    hi[k] = neg;
                }
            }
        }
    }
}
static size_t max_threads() {
#ifdef _OPENMP
    return (size_t)omp_get_max_threads();
#else
    return 1;
#endif
}
// domain.rs:691-735 (io_helper_with_roots): DIF, natural in -> bit-reversed out
static void io_helper_with_roots(Fr* xi, size_t n, const std::vector<Fr>& roots_in) {
    std::vector<Fr> owned;
    const Fr* roots = roots_in.data();
    size_t roots_len = roots_in.size();
    size_t step = 1;
    bool first = true;
    size_t gap = n / 2;
    while (gap > 0) {
        size_t chunk_size = 2 * gap, num_chunks = n / chunk_size;
        if (num_chunks >= MIN_NUM_CHUNKS_FOR_COMPACTION) {
            if (!first) {
                std::vector<Fr> comp((roots_len + step * 2 - 1) / (step * 2));
                for (size_t i = 0; i < comp.size(); i++) comp[i] = roots[i * step * 2];
                owned.swap(comp);
                roots = owned.data();
                roots_len = owned.size();
            }
            step = 1;
        } else {
            step = num_chunks;
        }
        first = false;
        apply_butterfly<true>(xi, n, roots, step, chunk_size, num_chunks, max_threads(), gap);
        gap /= 2;
    }
}
// domain.rs:737-773 (oi_helper_with_roots): DIT, bit-reversed in -> natural out
static void oi_helper_with_roots(Fr* xi, size_t n, const std::vector<Fr>& roots_cache) {
    size_t compaction_max = std::min(roots_cache.size() / 2, roots_cache.size() / MIN_NUM_CHUNKS_FOR_COMPACTION);
    std::vector<Fr> compacted(compaction_max);
    size_t gap = 1;
    while (gap < n) {
        size_t chunk_size = 2 * gap, num_chunks = n / chunk_size;
        const Fr* roots;
        size_t step;
        if (num_chunks >= MIN_NUM_CHUNKS_FOR_COMPACTION && gap < n / 2) {
            for (size_t i = 0; i < gap; i++) compacted[i] = roots_cache[i * num_chunks];
            roots = compacted.data();
            step = 1;
        } else {
            roots = roots_cache.data();
            step = num_chunks;
        }
        apply_butterfly<false>(xi, n, roots, step, chunk_size, num_chunks, max_threads(), gap);
        gap *= 2;
    }
}
// domain.rs:241-254 (distribute_powers_and_mul_by_const)
static void distribute_powers_and_mul_by_const(Fr* x, size_t n, const Fr& g, const Fr& c) {
    const size_t CH = 1024;
    size_t nch = (n + CH - 1) / CH;
#pragma omp parallel for schedule(static)
    for (size_t ci = 0; ci < nch; ci++) {
        uint64_t e[1] = {(uint64_t)(ci * CH)};
        Fr pw = c * g.pow(e, 1);
        size_t end = std::min(n, (ci + 1) * CH);
        for (size_t i = ci * CH; i < end; i++) {
            x[i] *= pw;
            pw *= g;
        }
    }
}

enum { ORDER_NN = 0, ORDER_NR = 1, ORDER_RN = 2, ORDER_RR = 3 };  // algorithms/cuda/src/lib.rs:22-28
enum { DIR_FORWARD = 0, DIR_INVERSE = 1 };                         // lib.rs:30-34
enum { TYPE_STANDARD = 0, TYPE_COSET = 1 };                        // lib.rs:36-40

// Semantics of the FFI call `snarkvm_ntt` (algorithms/cuda/src/lib.rs:77-97), computed with the
// reference's CPU transforms:
//   NN fwd  = in_order_fft_in_place  (domain.rs:374-392: precompute_fft each call, DIF + derange)
//   NN inv  = in_order_ifft_in_place (domain.rs:403-422: derange + DIT, * size_inv)
//   coset fwd = distribute_powers(g) then fft (domain.rs:201-206)
//   coset inv = ifft helper then distribute_powers_and_mul_by_const(g^-1, size_inv) (domain.rs:424-444)
//   NR fwd  = out_order_fft  (IO, domain.rs:470-476) ; RN inv = out_order_ifft (OI, domain.rs:502-509)
//   other order combinations: explicit bit-reversal permutation before / after.
static int ntt_ref(Fr* x, uint32_t lg, int order, int dir, int type) {
    EvaluationDomain d;
    if (!EvaluationDomain::make((size_t)1 << lg, d)) return 1;
    size_t n = (size_t)1 << lg;
    bool in_rev = (order == ORDER_RN || order == ORDER_RR);
    bool out_rev = (order == ORDER_NR || order == ORDER_RR);
    if (dir == DIR_FORWARD) {
        if (in_rev) derange(x, n, lg);
        if (type == TYPE_COSET) distribute_powers_and_mul_by_const(x, n, fr_generator(), Fr::one());
        std::vector<Fr> roots = roots_of_unity(d.size, d.group_gen);  // precompute_fft (domain.rs:360-365)
        io_helper_with_roots(x, n, roots);
        if (!out_rev) derange(x, n, lg);
    } else {
        if (!in_rev) derange(x, n, lg);
        std::vector<Fr> roots = roots_of_unity(d.size, d.group_gen_inv);  // precompute_ifft (:367-372)
        oi_helper_with_roots(x, n, roots);
        if (type == TYPE_COSET) {
            distribute_powers_and_mul_by_const(x, n, d.generator_inv, d.size_inv);
        } else {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) x[i] *= d.size_inv;
        }
        if (out_rev) derange(x, n, lg);
    }
    return 0;
}

// fft/polynomial/multiplier.rs:70-134 (PolyMultiplier::multiply, CPU path) with the FFI corner
// cases of algorithms/cuda/cuda/snarkvm.cu:196-210.  `out` holds 2^lg elements.
static int polymul_ref(Fr* out, size_t pcount, const Fr* const* polys, const size_t* plens, size_t ecount,
                       const Fr* const* evals, const size_t* elens, uint32_t lg) {
    size_t n = (size_t)1 << lg;
    if (pcount + ecount == 0) return 0;
    if (pcount + ecount == 1) {
        if (pcount == 1) {
            memcpy(out, polys[0], sizeof(Fr) * plens[0]);
            return 0;
        }
        memset(out, 0, sizeof(Fr) * n);
        memcpy(out, evals[0], sizeof(Fr) * elens[0]);
        return ntt_ref(out, lg, ORDER_NN, DIR_INVERSE, TYPE_STANDARD);
    }
    EvaluationDomain d;
    if (!EvaluationDomain::make(n, d)) return 1;
    std::vector<Fr> roots = roots_of_unity(d.size, d.group_gen);
    std::vector<Fr> acc, tmp(n);
    for (size_t k = 0; k < pcount + ecount; k++) {
        std::fill(tmp.begin(), tmp.end(), Fr::zero());
        if (k < pcount) {
            if (plens[k] > n) return 2;
            memcpy(tmp.data(), polys[k], sizeof(Fr) * plens[k]);
            io_helper_with_roots(tmp.data(), n, roots);  // out_order_fft_in_place_with_pc
        } else {
            if (elens[k - pcount] != n) return 2;
            memcpy(tmp.data(), evals[k - pcount], sizeof(Fr) * n);
            derange(tmp.data(), n, lg);
        }
        if (acc.empty())
            acc = tmp;
        else {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) acc[i] *= tmp[i];
        }
    }
    std::vector<Fr> iroots = roots_of_unity(d.size, d.group_gen_inv);
    oi_helper_with_roots(acc.data(), n, iroots);  // out_order_ifft_in_place_with_pc
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) acc[i] *= d.size_inv;
    memcpy(out, acc.data(), sizeof(Fr) * n);
    return 0;
}

// -------------------------------------------------------------------------------------
// C exports (ctypes)
// -------------------------------------------------------------------------------------
typedef Affine<Fq> G1Affine;
typedef Projective<Fq> G1Projective;
typedef Affine<Fq2> G2Affine;
typedef Projective<Fq2> G2Projective;
static_assert(sizeof(Fr) == 32 && sizeof(Fq) == 48, "field sizes");
static_assert(sizeof(G1Affine) == 104 && sizeof(G1Projective) == 144, "G1 layout (SURVEY.md App. B)");
static_assert(sizeof(G2Affine) == 200 && sizeof(G2Projective) == 288, "G2 layout (SURVEY.md App. B)");

static void set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}


// ---------------------------------------------------------------------------------------------
// Prover-round polynomial helpers around the NTTs (SURVEY.md §8 N2 / a15 `open`)
// ---------------------------------------------------------------------------------------------
static void trim(std::vector<Fr>& v) {  // DensePolynomial::from_coefficients_vec (dense.rs:76-86)
    while (!v.empty() && v.back().is_zero()) v.pop_back();
}
// Polynomial::divide_with_q_and_r (fft/polynomial/mod.rs:222-256): schoolbook long division.  The divisor is a list
// of (degree, coefficient) terms sorted by degree (SparsePolynomial) - a dense divisor is the same list with every
// degree present.  Returns false for a zero divisor.
static bool divide_with_q_and_r(std::vector<Fr> self, const std::vector<std::pair<size_t, Fr>>& divisor, std::vector<Fr>& quotient,
                                std::vector<Fr>& remainder) {
    trim(self);
    if (divisor.empty()) return false;
    const size_t ddeg = divisor.back().first;
    quotient.clear();
    if (self.empty()) {
        remainder.clear();
        return true;
    }
    if (self.size() - 1 < ddeg) {
        remainder = self;
        return true;
    }
    quotient.assign(self.size() - 1 - ddeg + 1, Fr::zero());
    remainder = self;
    const Fr lead_inv = divisor.back().second.inverse();
    while (!remainder.empty() && remainder.size() - 1 >= ddeg) {
        const Fr cur_q = remainder.back() * lead_inv;
        const size_t cur_deg = remainder.size() - 1 - ddeg;
        quotient[cur_deg] = cur_q;
        for (auto& t : divisor) remainder[cur_deg + t.first] -= cur_q * t.second;
        trim(remainder);
    }
    trim(quotient);
    return true;
}
// DensePolynomial::evaluate (dense.rs:98-114): sum of coeff_i * point^i
static Fr poly_evaluate(const Fr* c, size_t n, const Fr& point) {
    std::vector<Fr> v(c, c + n);
    trim(v);
    if (v.empty()) return Fr::zero();
    if (point.is_zero()) return v[0];
    Fr acc = Fr::zero(), pw = Fr::one();
    for (size_t i = 0; i < v.size(); i++) {
        acc += pw * v[i];
        pw *= point;
    }
    return acc;
}
// fields/src/lib.rs:99-129 (serial_batch_inversion_and_mul): zeros are skipped and stay zero.  The rayon version
// (:77-93) applies the same function per chunk; every non-zero element becomes coeff / v_i either way.
static void batch_inversion_and_mul(Fr* v, size_t n, const Fr& coeff) {
    std::vector<Fr> prod;
    prod.reserve(n);
    Fr tmp = Fr::one();
    for (size_t i = 0; i < n; i++)
        if (!v[i].is_zero()) {
            tmp *= v[i];
            prod.push_back(tmp);
        }
    if (prod.empty()) return;
    tmp = tmp.inverse();
    tmp *= coeff;
    size_t k = prod.size();
    for (size_t i = n; i-- > 0;) {
        if (v[i].is_zero()) continue;
        k--;
        const Fr s = k ? prod[k - 1] : Fr::one();
        const Fr new_tmp = tmp * v[i];
        v[i] = tmp * s;
        tmp = new_tmp;
    }
}
// EvaluationDomain::evaluate_all_lagrange_coefficients (fft/domain.rs:258-292)
static void evaluate_all_lagrange_coefficients(const EvaluationDomain& d, const Fr& tau, Fr* u) {
    const size_t size = d.size;
    uint64_t e[1] = {d.size};
    const Fr t_size = tau.pow(e, 1);
    const Fr one = Fr::one();
    if (t_size.is_one()) {
        for (size_t i = 0; i < size; i++) u[i] = Fr::zero();
        Fr omega_i = one;
        for (size_t i = 0; i < size; i++) {
            if (omega_i == tau) {
                u[i] = one;
                break;
            }
            omega_i *= d.group_gen;
        }
    } else {
        Fr l = (t_size - one) * d.size_inv;
        Fr r = one;
        std::vector<Fr> ls(size);
        for (size_t i = 0; i < size; i++) {
            u[i] = tau - r;
            ls[i] = l;
            l *= d.group_gen;
            r *= d.group_gen;
        }
        batch_inversion_and_mul(u, size, one);
        for (size_t i = 0; i < size; i++) u[i] = ls[i] * u[i];
    }
}


// ---------------------------------------------------------------------------------------------
// Setup-time group operations
// ---------------------------------------------------------------------------------------------
typedef Projective<Fq> G1P;
static constexpr size_t FR_MODULUS_BITS = 253;  // fr.rs MODULUS_BITS
// `ProjectiveCurve::mul` by a 256-bit integer: MSB-first double-and-add (projective.rs `mul_bits` over BitIteratorBE)
static G1P proj_mul(const G1P& base, const uint64_t* k) {
    G1P out = G1P::zero();
    bool started = false;
    for (int w = 3; w >= 0; w--)
        for (int b = 63; b >= 0; b--) {
            const bool bit = (k[w] >> b) & 1;
            if (!started && !bit) continue;
            started = true;
            out.double_in_place();
            if (bit) out.add_assign(base);
        }
    return out;
}
// FixedBase::get_window_table (msm/fixed_base.rs:33-68)
static std::vector<std::vector<G1P>> fixed_window_table(size_t scalar_size, size_t window, G1P g) {
    const size_t in_window = (size_t)1 << window;
    const size_t outerc = (scalar_size + window - 1) / window;
    const size_t last_in_window = (size_t)1 << (scalar_size - (outerc - 1) * window);
    std::vector<std::vector<G1P>> multiples(outerc, std::vector<G1P>(in_window, G1P::zero()));
    std::vector<G1P> g_outers;
    G1P g_outer = g;
    for (size_t o = 0; o < outerc; o++) {
        g_outers.push_back(g_outer);
        for (size_t i = 0; i < window; i++) g_outer.double_in_place();
    }
#pragma omp parallel for schedule(dynamic)
    for (size_t outer = 0; outer < outerc; outer++) {
        const size_t cur = (outer == outerc - 1) ? last_in_window : in_window;
        G1P g_inner = G1P::zero();
        for (size_t inner = 0; inner < cur; inner++) {
            multiples[outer][inner] = g_inner;
            g_inner.add_assign(g_outers[outer]);
        }
    }
    return multiples;
}
// FixedBase::windowed_mul (fixed_base.rs:70-90): bits of scalar.to_bigint(), one table entry per window, + table[0][0]
static G1P fixed_windowed_mul(size_t outerc, size_t window, const std::vector<std::vector<G1P>>& table, const Fr& scalar) {
    uint64_t k[4];
    scalar.to_bigint(k);
    G1P sum = G1P::zero();
    for (size_t outer = 0; outer < outerc; outer++) {
        size_t inner = 0;
        for (size_t i = 0; i < window; i++) {
            const size_t bit = outer * window + i;
            if (bit < FR_MODULUS_BITS && ((k[bit >> 6] >> (bit & 63)) & 1)) inner |= (size_t)1 << i;
        }
        sum.add_assign(table[outer][inner]);
    }
    sum.add_assign(table[0][0]);
    return sum;
}
// EvaluationDomain::ifft over group elements (fft/domain.rs:177-192 with T = G1Projective): the same linear map as for
// field elements - bit-reverse, radix-2 decimation-in-time butterflies with powers of group_gen_inv, then * size_inv.
static void group_ifft(G1P* x, uint32_t lg, bool inverse) {
    EvaluationDomain d;
    EvaluationDomain::make((size_t)1 << lg, d);
    const size_t n = (size_t)1 << lg;
    for (size_t i = 0; i < n; i++) {
        const size_t r = lg ? (size_t)bitrev64(i, lg) : 0;
        if (i < r) std::swap(x[i], x[r]);
    }
    const Fr root = inverse ? d.group_gen_inv : d.group_gen;
    for (size_t len = 2; len <= n; len <<= 1) {
        uint64_t e[1] = {(uint64_t)(n / len)};
        const Fr wlen = root.pow(e, 1);
        const size_t half = len / 2;
        std::vector<Fr> tw(half);
        Fr w = Fr::one();
        for (size_t j = 0; j < half; j++) {
            tw[j] = w;
            w *= wlen;
        }
#pragma omp parallel for schedule(static)
        for (size_t t = 0; t < n / 2; t++) {
            const size_t blk = t / half, j = t % half;
            const size_t ia = blk * len + j, ib = ia + half;
            uint64_t k[4];
            tw[j].to_bigint(k);
            const G1P v = proj_mul(x[ib], k);
            G1P u = x[ia];
            G1P nv = v;
            nv.y = nv.y.neg();
            x[ia].add_assign(v);
            u.add_assign(nv);
            x[ib] = u;
        }
    }
    if (inverse) {
        uint64_t k[4];
        d.size_inv.to_bigint(k);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) x[i] = proj_mul(x[i], k);
    }
}

extern "C" {
int oracle_max_threads() { return (int)max_threads(); }
void oracle_set_threads(int n) { set_threads(n); }

// op: 0 add, 1 sub, 2 mul, 3 inverse(a), 4 from_bigint(a), 5 to_bigint(a), 6 neg(a), 7 sqr(a)
void oracle_fr_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    const Fr* A = (const Fr*)a;
    const Fr* B = (const Fr*)b;
    Fr* O = (Fr*)out;
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: O[i] = A[i] + B[i]; break;
            case 1: O[i] = A[i] - B[i]; break;
            case 2: O[i] = A[i] * B[i]; break;
            case 3: O[i] = A[i].inverse(); break;
            case 4: O[i] = Fr::from_bigint(A[i].l); break;
            case 5: A[i].to_bigint(O[i].l); break;
            case 6: O[i] = A[i].neg(); break;
            case 7: O[i] = A[i].sqr(); break;
        }
    }
}
void oracle_fq_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    const Fq* A = (const Fq*)a;
    const Fq* B = (const Fq*)b;
    Fq* O = (Fq*)out;
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: O[i] = A[i] + B[i]; break;
            case 1: O[i] = A[i] - B[i]; break;
            case 2: O[i] = A[i] * B[i]; break;
            case 3: O[i] = A[i].inverse(); break;
            case 4: O[i] = Fq::from_bigint(A[i].l); break;
            case 5: A[i].to_bigint(O[i].l); break;
            case 6: O[i] = A[i].neg(); break;
            case 7: O[i] = A[i].sqr(); break;
        }
    }
}
// domain constants: out = [group_gen, group_gen_inv, size_inv, generator_inv, size_as_field_element] (Montgomery)
int oracle_domain(uint32_t lg, uint64_t* out) {
    EvaluationDomain d;
    if (!EvaluationDomain::make((size_t)1 << lg, d)) return 1;
    Fr v[5] = {d.group_gen, d.group_gen_inv, d.size_inv, d.generator_inv, d.size_as_field_element};
    memcpy(out, v, sizeof v);
    return 0;
}
int oracle_ntt(uint64_t* inout, uint32_t lg, int order, int dir, int type) {
    return ntt_ref((Fr*)inout, lg, order, dir, type);
}
int oracle_polymul(uint64_t* out, size_t pcount, const uint64_t* const* polys, const size_t* plens, size_t ecount,
                   const uint64_t* const* evals, const size_t* elens, uint32_t lg) {
    return polymul_ref((Fr*)out, pcount, (const Fr* const*)polys, plens, ecount, (const Fr* const*)evals, elens, lg);
}

// ---- prover-round polynomial helpers ----
// op: 0 a+b, 1 a-b, 2 a*b, 3 a*b-c, 4 a*b[0], 5 a-b[0], 6 a+b*c[0]
void oracle_fr_vec_op(int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n) {
    const Fr *A = (const Fr*)a, *B = (const Fr*)b, *C = (const Fr*)c;
    Fr* O = (Fr*)out;
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: O[i] = A[i] + B[i]; break;
            case 1: O[i] = A[i] - B[i]; break;
            case 2: O[i] = A[i] * B[i]; break;
            case 3: O[i] = A[i] * B[i] - C[i]; break;
            case 4: O[i] = A[i] * B[0]; break;
            case 5: O[i] = A[i] - B[0]; break;
            case 6: O[i] = A[i] + B[i] * C[0]; break;
        }
    }
}
// Long division by a divisor given as `terms` (degree, coefficient) pairs.  quot / rem must hold n elements; their
// trimmed lengths are returned through qlen / rlen.  Returns 1 for a zero divisor.
int oracle_fr_poly_divide(const uint64_t* poly, size_t n, const size_t* div_deg, const uint64_t* div_coeff, size_t terms, uint64_t* quot,
                          size_t* qlen, uint64_t* rem, size_t* rlen) {
    std::vector<Fr> self((const Fr*)poly, (const Fr*)poly + n), q, r;
    std::vector<std::pair<size_t, Fr>> d;
    for (size_t i = 0; i < terms; i++) {
        const Fr cf = ((const Fr*)div_coeff)[i];
        if (!cf.is_zero()) d.push_back({div_deg[i], cf});
    }
    if (!divide_with_q_and_r(self, d, q, r)) return 1;
    if (!q.empty()) memcpy(quot, q.data(), q.size() * sizeof(Fr));
    if (!r.empty()) memcpy(rem, r.data(), r.size() * sizeof(Fr));
    *qlen = q.size();
    *rlen = r.size();
    return 0;
}
void oracle_fr_evaluate(const uint64_t* poly, size_t n, const uint64_t* point, uint64_t* out) {
    const Fr v = poly_evaluate((const Fr*)poly, n, *(const Fr*)point);
    memcpy(out, &v, sizeof v);
}
void oracle_fr_batch_inversion_and_mul(uint64_t* v, size_t n, const uint64_t* coeff) {
    batch_inversion_and_mul((Fr*)v, n, *(const Fr*)coeff);
}
void oracle_fr_distribute_powers(uint64_t* v, size_t n, const uint64_t* g, const uint64_t* c) {
    distribute_powers_and_mul_by_const((Fr*)v, n, *(const Fr*)g, *(const Fr*)c);
}
int oracle_fr_lagrange_coefficients(uint32_t lg, const uint64_t* tau, uint64_t* out) {
    EvaluationDomain d;
    if (!EvaluationDomain::make((size_t)1 << lg, d)) return 1;
    evaluate_all_lagrange_coefficients(d, *(const Fr*)tau, (Fr*)out);
    return 0;
}
// DensePolynomial::mul_by_vanishing_poly (dense.rs:153-159): out has len + domain elements (untrimmed)
void oracle_fr_mul_by_vanishing(const uint64_t* poly, size_t len, size_t domain, uint64_t* out) {
    const Fr* p = (const Fr*)poly;
    Fr* o = (Fr*)out;
    for (size_t i = 0; i < domain; i++) o[i] = Fr::zero();
    for (size_t i = 0; i < len; i++) o[domain + i] = p[i];
    for (size_t i = 0; i < len; i++) o[i] -= p[i];
}

// ---- setup-time group operations ----
// FixedBase::msm (fixed_base.rs:87-97): out[i] = v[i] * g, window table built with get_window_table(scalar_size, window, g)
void oracle_g1_fixed_base_msm(const void* g_affine, size_t scalar_size, size_t window, const uint64_t* scalars, size_t n, void* out_proj) {
    const Affine<Fq>& g = *(const Affine<Fq>*)g_affine;
    G1P gp = G1P::zero();
    if (!g.is_zero()) gp = {g.x, g.y, Fq::one()};
    const auto table = fixed_window_table(scalar_size, window, gp);
    const size_t outerc = (scalar_size + window - 1) / window;
    G1P* O = (G1P*)out_proj;
    const Fr* V = (const Fr*)scalars;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) O[i] = fixed_windowed_mul(outerc, window, table, V[i]);
}
void oracle_g1_group_ntt(void* inout_proj, uint32_t lg, int inverse) { group_ifft((G1P*)inout_proj, lg, inverse != 0); }

// ---- G1 ----
// kind: 0 = batched::msm, 1 = standard::msm, 2 = naive (sum of mul_bits)
void oracle_g1_msm(int kind, const void* bases, size_t nbases, const uint64_t* scalars, size_t nscalars, void* out) {
    const G1Affine* B = (const G1Affine*)bases;
    G1Projective r;
    if (kind == 0)
        r = batched_msm<Fq>(B, nbases, scalars, nscalars);
    else if (kind == 1)
        r = standard_msm<Fq>(B, nbases, scalars, nscalars);
    else
        r = naive_msm<Fq>(B, scalars, std::min(nbases, nscalars));
    memcpy(out, &r, sizeof r);
}
void oracle_g1_to_affine(const void* proj, void* out_affine, size_t n) {
    const G1Projective* P = (const G1Projective*)proj;
    G1Affine* A = (G1Affine*)out_affine;
    for (size_t i = 0; i < n; i++) A[i] = P[i].to_affine();
}
int oracle_g1_is_on_curve(const void* aff) {
    const G1Affine* a = (const G1Affine*)aff;
    if (a->is_zero()) return 1;
    return a->y.sqr() == (a->x.sqr() * a->x + Fq::one());  // affine.rs:211-221, b = 1
}
void oracle_g1_add(const void* p1, const void* p2, void* out) {  // projective + projective
    G1Projective a = *(const G1Projective*)p1;
    a.add_assign(*(const G1Projective*)p2);
    memcpy(out, &a, sizeof a);
}
void oracle_g1_mul(const void* base, const uint64_t* scalar, void* out_proj) {
    G1Projective r = mul_bits<Fq>(*(const G1Affine*)base, scalar);
    memcpy(out_proj, &r, sizeof r);
}
// bases[i] = (start + i) * G for i in [0, n) as Rust-layout affines (deterministic synthetic base set,
// BASELINE.md section 3): running mixed addition + batch normalisation (projective.rs:172-219).
void oracle_g1_gen_bases(const void* gen_affine, uint64_t start, size_t n, void* out) {
    const G1Affine g = *(const G1Affine*)gen_affine;
    G1Affine* O = (G1Affine*)out;
    const size_t CH = 4096;
    size_t nch = (n + CH - 1) / CH;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t ci = 0; ci < nch; ci++) {
        size_t lo = ci * CH, hi = std::min(n, lo + CH);
        uint64_t s[4] = {start + lo, 0, 0, 0};
        G1Projective cur = mul_bits<Fq>(g, s);
        std::vector<G1Projective> v(hi - lo);
        for (size_t i = lo; i < hi; i++) {
            v[i - lo] = cur;
            cur.add_assign_mixed(g);
        }
        // Montgomery's trick over z (batch_normalization); (start+i)*G is never infinity for i < r
        std::vector<Fq> prod(v.size());
        Fq acc = Fq::one();
        for (size_t i = 0; i < v.size(); i++) {
            acc *= v[i].z;
            prod[i] = acc;
        }
        Fq inv = acc.inverse();
        for (size_t i = v.size(); i-- > 0;) {
            Fq zinv = (i == 0) ? inv : inv * prod[i - 1];
            inv *= v[i].z;
            Fq z2 = zinv.sqr();
            G1Affine a = G1Affine::zero();
            a.infinity = 0;
            a.x = v[i].x * z2;
            a.y = v[i].y * (z2 * zinv);
            O[lo + i] = a;
        }
    }
}

// ---- G2 ----
void oracle_g2_msm(int kind, const void* bases, size_t nbases, const uint64_t* scalars, size_t nscalars, void* out) {
    const G2Affine* B = (const G2Affine*)bases;
    G2Projective r;
    if (kind == 0)
        r = batched_msm<Fq2>(B, nbases, scalars, nscalars);
    else if (kind == 1)
        r = standard_msm<Fq2>(B, nbases, scalars, nscalars);
    else
        r = naive_msm<Fq2>(B, scalars, std::min(nbases, nscalars));
    memcpy(out, &r, sizeof r);
}
void oracle_g2_to_affine(const void* proj, void* out_affine, size_t n) {
    const G2Projective* P = (const G2Projective*)proj;
    G2Affine* A = (G2Affine*)out_affine;
    for (size_t i = 0; i < n; i++) A[i] = P[i].to_affine();
}
void oracle_g2_mul(const void* base, const uint64_t* scalar, void* out_proj) {
    G2Projective r = mul_bits<Fq2>(*(const G2Affine*)base, scalar);
    memcpy(out_proj, &r, sizeof r);
}
}  // extern "C"
