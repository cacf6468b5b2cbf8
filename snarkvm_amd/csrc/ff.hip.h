// ff.hip.h - BLS12-377 prime-field arithmetic for gfx950 (CDNA4).
//
// Representation (chosen from measurements on MI355X, profiles/r01_microbench_instruction_rates.txt):
// gfx950 has no 64-bit multiplier but v_mad_u64_u32 (32x32+64 -> 64) issues at ~57 % of the plain
// VALU rate, while every carry hand-off through VCC costs two mandatory wait states on this
// target.  So field elements are held in registers as N limbs of 29 bits (Fq: 13 limbs = 377 bits,
// Fr: 9 limbs = 261 bits): a whole product column (<= 13 a_i*b_j plus <= 13 m_i*p_j, each < 2^58)
// accumulates into ONE 64-bit register pair with one v_mad_u64_u32 per partial product and no
// carry instruction at all.  Measured: 68 G Fq-mul/s vs 33 G (32-bit limbs, CIOS) and 50 G
// (32-bit limbs, hand-placed v_mad_u64_u32 + v_addc_co_u32).
//
// Montgomery radix.  The reference keeps Fq/Fr in Montgomery form with R = 2^384 / 2^256
// (fields/src/fp_384.rs, fp_256.rs).  The 29-bit column reduction divides by 2^(29 N) = 2^377 /
// 2^261 instead, so the *internal* Montgomery form of a is a * 2^(29N) mod p:
//   - Fq (MSM): bases are converted once when they enter the device (one multiplication by the
//     constant 2^370, see from_mem_mont); the three result coordinates are converted back
//     (to_mem_mont).  Everything in between stays internal.
//   - Fr (NTT): the transform is linear, so the memory word a*2^256 is simply *read as* the
//     internal form of a*2^-5; twiddles are stored internally (w*2^261) and every product
//     mont(x, w) = x*w*2^-261 keeps the data in memory form.  No conversion at all.
// Every operation returns the canonical representative (< p, limbs < 2^29), like the reference
// (fp_256.rs:61-65), which is what makes limb-exact parity meaningful.
//
// The same source compiles for the host, so the arithmetic is unit-tested without a GPU through
// the snarkvm_hip_selftest_* entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SV_HD __host__ __device__ __forceinline__
// Exceptional paths (the doubling inside the addition laws, quad_add's plain fallback).  -DSV_COLD_OOL (`python -m snarkvm_amd.build
// --ool`, an A/B switch) puts ALL of them out of line; measured on one box against the inlined shape the tail kernels of a 2^16 G1 MSM
// then run 10 % slower (0.184 vs 0.167 ms: scratch for the callee's operand copies) and the G2 accumulate 5 % - so the product
// build inlines them, with one exception that costs nothing measurable and cut the build from 14 to 4.6 minutes: the general
// addition over Fq2 (ec.hip.h COLD_DBL_OOL, msm.hip.h quad_add_plain).
#if defined(SV_COLD_OOL)
#define SV_COLD __host__ __device__ __noinline__
#else
#define SV_COLD __host__ __device__ __forceinline__
#endif

namespace sv {

static constexpr uint32_t LIMB_BITS = 29;
static constexpr uint32_t LIMB_MASK = (1u << 29) - 1;

// ------------------------------------------------------------------------------------------
// Field parameters: 29-bit limbs of the constants in curves/src/bls12_377/{fr,fq}.rs
// (tests/test_host_arith.py re-derives every table from tests/golden/constants.json)
// ------------------------------------------------------------------------------------------
struct FrP {  // scalar field r, 253 bits (fr.rs:137-170)
    static constexpr int N = 9;       // 29-bit limbs
    static constexpr int WORDS = 8;   // 32-bit words in memory
    static constexpr int MEM_R_BITS = 256;
    static constexpr uint32_t MOD[9] = {0x00000001u, 0x108c0000u, 0x00000042u, 0x14edfda0u, 0x1b00159au,
                                        0x068f2e1bu, 0x155982d1u, 0x0bd34594u, 0x0012ab65u};
    static constexpr uint32_t ONE[9] = {0x1ffffe4au, 0x1077ffffu, 0x1fff8e31u, 0x10d0103fu, 0x0ddb0965u,
                                        0x07071c5cu, 0x18da2e10u, 0x0486f3a3u, 0x000ec090u};  // 2^261 mod r
    static constexpr uint32_t R2[9] = {0x0615ebc3u, 0x1d9b570cu, 0x191f15fbu, 0x00f6d3e2u, 0x15ede530u,
                                       0x0fadbc68u, 0x0428942fu, 0x06a15030u, 0x000c9478u};   // 2^522 mod r
    static constexpr uint32_t MEM2INT[9] = {0x1fffc927u, 0x1153ffffu, 0x1ff1bfb1u, 0x0ec4435fu, 0x185f1096u,
                                            0x1ce80ad5u, 0x0587fb98u, 0x093ca8f4u, 0x0005551eu};  // 2^266 mod r
    static constexpr uint32_t INT2MEM[9] = {0x1ffffff3u, 0x08e3ffffu, 0x1ffffc9fu, 0x0fea1edfu, 0x00fee725u,
                                            0x0abaa896u, 0x0a745b60u, 0x06457773u, 0x000d4bdau};  // 2^256 mod r
};
struct FqP {  // base field q, 377 bits (fq.rs:111-150)
    static constexpr int N = 13;
    static constexpr int WORDS = 12;
    static constexpr int MEM_R_BITS = 384;
    static constexpr uint32_t MOD[13] = {0x00000001u, 0x08460000u, 0x00000021u, 0x16ba8860u, 0x14800170u,
                                         0x1117dd04u, 0x0e3c7bcdu, 0x1e601ea2u, 0x1b1a22d9u, 0x03650a49u,
                                         0x118ec170u, 0x0f8a21d5u, 0x1ae3a461u};
    static constexpr uint32_t ONE[13] = {0x1fffffffu, 0x17b9ffffu, 0x1fffffdeu, 0x0945779fu, 0x0b7ffe8fu,
                                         0x0ee822fbu, 0x11c38432u, 0x019fe15du, 0x04e5dd26u, 0x1c9af5b6u,
                                         0x0e713e8fu, 0x1075de2au, 0x051c5b9eu};  // 2^377 mod q
    static constexpr uint32_t R2[13] = {0x01b25004u, 0x0419539du, 0x0c50044cu, 0x09b9e179u, 0x01bcca40u,
                                        0x05c4d195u, 0x0d6add48u, 0x14f9f71au, 0x106154e2u, 0x04a0beb4u,
                                        0x11d7f1cdu, 0x11c1c61eu, 0x155f398du};   // 2^754 mod q
    static constexpr uint32_t MEM2INT[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x00400000u};  // 2^370
    static constexpr uint32_t INT2MEM[13] = {0x1fffff68u, 0x166fffffu, 0x1fffec40u, 0x013f06ffu, 0x13ff2514u,
                                             0x19d4c53eu, 0x0c167df6u, 0x16edcf8cu, 0x087b4e97u, 0x1c01e427u,
                                             0x133d256fu, 0x05fbe934u, 0x08d6661eu};  // 2^384 mod q
};

// memory image of a field element as the Rust side sees it (little-endian 32-bit words)
template <int W>
struct alignas(16) mem_words_t {
    uint32_t w[W];
};

// ------------------------------------------------------------------------------------------
// Fp<P>: N limbs of 29 bits, canonical, internal Montgomery form a * 2^(29N)
// ------------------------------------------------------------------------------------------
template <class P>
struct Fp {
    static constexpr int N = P::N;
    static constexpr int WORDS = P::WORDS;
    static constexpr int MEM_WORDS = P::WORDS;
    typedef mem_words_t<P::WORDS> mem_t;
    uint32_t v[N];

    SV_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    SV_HD static Fp from_table(const uint32_t* t) {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = t[i];
        return r;
    }
    SV_HD static Fp one() { return from_table(P::ONE); }
    SV_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= v[i];
        return o == 0;
    }
    SV_HD bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    SV_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // ---- memory <-> limbs (pure bit repacking; values < p < 2^(32 WORDS))
    SV_HD static Fp unpack(const uint32_t* w) {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int bit = 29 * i, wi = bit / 32, sh = bit % 32;
            uint32_t lo = (wi < WORDS) ? w[wi] : 0u;
            uint32_t hi = (wi + 1 < WORDS) ? w[wi + 1] : 0u;
            uint32_t x = (sh == 0) ? lo : ((lo >> sh) | (hi << (32 - sh)));
            r.v[i] = x & LIMB_MASK;
        }
        return r;
    }
    SV_HD void pack(uint32_t* w) const {
#pragma unroll
        for (int j = 0; j < WORDS; j++) {
            uint32_t x = 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int lo_bit = 29 * i - 32 * j;  // position of limb i's bit 0 inside word j
                if (lo_bit > -29 && lo_bit < 32) x |= (lo_bit >= 0) ? (v[i] << lo_bit) : (v[i] >> (-lo_bit));
            }
            w[j] = x;
        }
    }
    SV_HD static Fp load(const void* p) {  // 16-byte aligned memory image -> limbs
        uint32_t w[WORDS];
        const uint4* q = (const uint4*)p;
#pragma unroll
        for (int i = 0; i < WORDS / 4; i++) {
            uint4 t = q[i];
            w[4 * i] = t.x, w[4 * i + 1] = t.y, w[4 * i + 2] = t.z, w[4 * i + 3] = t.w;
        }
        return unpack(w);
    }
    SV_HD void store(void* p) const {
        uint32_t w[WORDS];
        pack(w);
        uint4* q = (uint4*)p;
#pragma unroll
        for (int i = 0; i < WORDS / 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }

    // ---- additive group.  fp_256.rs:730-750 (add_assign / sub_assign), 670-684 (neg), 232-237 (double)
    SV_HD Fp operator+(const Fp& b) const {
        uint32_t t[N], d[N], u[N];
        int32_t c = 0;
        uint32_t cc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            t[i] = v[i] + b.v[i];  // < 2^30
            int32_t x = (int32_t)(t[i] - P::MOD[i]) + c;
            d[i] = (uint32_t)x & LIMB_MASK;
            c = x >> 29;
            uint32_t y = t[i] + cc;
            u[i] = y & LIMB_MASK;
            cc = y >> 29;
        }
        Fp r;  // t < p  <=>  final borrow
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = (c < 0) ? u[i] : d[i];
        return r;
    }
    SV_HD Fp operator-(const Fp& b) const {
        uint32_t d[N];
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t x = (int32_t)(v[i] - b.v[i]) + c;
            d[i] = (uint32_t)x & LIMB_MASK;
            c = x >> 29;
        }
        const uint32_t mask = (uint32_t)c;  // all ones if a < b: add p back
        uint32_t cc = 0;
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t y = d[i] + (P::MOD[i] & mask) + cc;
            r.v[i] = y & LIMB_MASK;
            cc = y >> 29;
        }
        return r;
    }
    SV_HD Fp neg() const { return zero() - *this; }
    SV_HD Fp dbl() const { return *this + *this; }

    // ---- Montgomery product (fp_256.rs:752-818 / fp_384.rs:769-899 compute the same residue for their
    // radix).  Product scanning in radix 2^29: column k = sum a_i b_(k-i) + sum m_i p_(k-i); p_0 = 1 and
    // -p^-1 = -1 mod 2^29, so m_k = -column mod 2^29 and m_k * p_0 is an addition.
    SV_HD Fp operator*(const Fp& b) const {
        uint32_t m[N], t[N];
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * N; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) acc += (uint64_t)v[i] * b.v[j];
            }
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * P::MOD[j];
            }
            if (k < N) {
                m[k] = (0u - (uint32_t)acc) & LIMB_MASK;
                acc += m[k];
            } else {
                // the result is < 2p, which for Fq needs one bit more than 29 N: keep the top limb unmasked
                t[k - N] = (k == 2 * N - 1) ? (uint32_t)acc : ((uint32_t)acc & LIMB_MASK);
            }
            acc >>= 29;
        }
        return cond_sub(t);
    }
    // a*b - c*d with ONE Montgomery reduction (the XYZZ addition's Y3 = R*(Q - X3) - Y1*PPP): the column accumulator
    // is signed (v_mad_i64_i32); partial sums stay inside [-13*2^58, 26*2^58] for 13 limbs.  Result canonical.
    SV_HD static Fp diff_of_products(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
        uint32_t m[N];
        int32_t nc[N], t[N];
#pragma unroll
        for (int i = 0; i < N; i++) nc[i] = -(int32_t)c.v[i];
        int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * N; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) {
                    acc += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[j];
                    acc += (int64_t)nc[i] * (int32_t)d.v[j];
                }
            }
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc += (int64_t)(int32_t)m[i] * (int32_t)P::MOD[j];
            }
            if (k < N) {
                m[k] = (0u - (uint32_t)acc) & LIMB_MASK;
                acc += m[k];
            } else {
                // value in (-p, 2p): every limb but the top one is normalised, the top one carries the sign
                t[k - N] = (k == 2 * N - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & LIMB_MASK);
            }
            acc >>= 29;  // arithmetic
        }
        // three-way correction: T < 0 -> T + p;  T >= p -> T - p
        uint32_t up[N], dn[N];
        int32_t cu = 0, cd = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t x = t[i] + (int32_t)P::MOD[i] + cu;
            up[i] = (uint32_t)x & LIMB_MASK;
            cu = x >> 29;
            int32_t y = t[i] - (int32_t)P::MOD[i] + cd;
            dn[i] = (uint32_t)y & LIMB_MASK;
            cd = y >> 29;
        }
        const bool neg = t[N - 1] < 0;
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = neg ? up[i] : (cd < 0 ? (uint32_t)t[i] : dn[i]);
        return r;
    }
    // dedicated squaring: off-diagonal products once, against the doubled operand
    SV_HD Fp sqr() const {
        uint32_t m[N], t[N], v2[N];
#pragma unroll
        for (int i = 0; i < N; i++) v2[i] = v[i] << 1;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * N; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j > i && j < N) acc += (uint64_t)v[i] * v2[j];
                if (j == i) acc += (uint64_t)v[i] * v[i];
            }
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * P::MOD[j];
            }
            if (k < N) {
                m[k] = (0u - (uint32_t)acc) & LIMB_MASK;
                acc += m[k];
            } else {
                // the result is < 2p, which for Fq needs one bit more than 29 N: keep the top limb unmasked
                t[k - N] = (k == 2 * N - 1) ? (uint32_t)acc : ((uint32_t)acc & LIMB_MASK);
            }
            acc >>= 29;
        }
        return cond_sub(t);
    }
    // t < 2p with normalised limbs -> canonical
    SV_HD static Fp cond_sub(const uint32_t* t) {
        uint32_t d[N];
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t x = (int32_t)(t[i] - P::MOD[i]) + c;
            d[i] = (uint32_t)x & LIMB_MASK;
            c = x >> 29;
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = (c < 0) ? t[i] : d[i];
        return r;
    }

    // raw words of the reference's Montgomery memory form <-> internal form
    SV_HD static Fp from_raw_words(const uint32_t* w) { return unpack(w).from_mem_mont(); }
    SV_HD void to_raw_words(uint32_t* w) const { to_mem_mont().pack(w); }

    // ---- lazy arithmetic (NTT butterflies).  Valid only for a field with spare bits in 29N (Fr: 261 - 253 = 8):
    // operands are arbitrary integers < 2^(29N) with normalised limbs, results likewise; nothing is reduced mod p
    // until reduce_lazy().  Bounds are tracked by the caller (ntt.hip.h).
    SV_HD static Fp add_lazy(const Fp& a, const Fp& b) {  // a + b  (must stay < 2^(29N))
        Fp r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t t = a.v[i] + b.v[i] + c;
            r.v[i] = (i == N - 1) ? t : (t & LIMB_MASK);
            c = t >> 29;
        }
        return r;
    }
    SV_HD static Fp sub_lazy(const Fp& a, const Fp& b, const uint32_t* kp) {  // a - b + kp, kp = k*p >= b
        Fp r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int32_t x = (int32_t)(a.v[i] - b.v[i] + kp[i]) + c;
            r.v[i] = (i == N - 1) ? (uint32_t)x : ((uint32_t)x & LIMB_MASK);
            c = x >> 29;
        }
        return r;
    }
    // kp = 2^s * p as normalised limbs (s < 29, 2^s * p < 2^(29N)); s is wave-uniform, so this is scalar-unit work
    SV_HD static void mod_shl(uint32_t* kp, int s) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t below = (i > 0) ? (P::MOD[i > 0 ? i - 1 : 0] >> (29 - s)) : 0u;
            kp[i] = ((P::MOD[i] << s) | below) & LIMB_MASK;
        }
    }
    // Montgomery product without the final conditional subtraction: for a < 2^(29N), b < p the result is < 2p
    SV_HD Fp mul_lazy(const Fp& b) const {
        uint32_t m[N];
        Fp r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * N; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) acc += (uint64_t)v[i] * b.v[j];
            }
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * P::MOD[j];
            }
            if (k < N) {
                m[k] = (0u - (uint32_t)acc) & LIMB_MASK;
                acc += m[k];
            } else {
                r.v[k - N] = (k == 2 * N - 1) ? (uint32_t)acc : ((uint32_t)acc & LIMB_MASK);
            }
            acc >>= 29;
        }
        return r;
    }
    // Montgomery reduction alone: x * 2^(-29N) for any x < 2^(29N) with normalised limbs, result < 2p.  Half the
    // multiply-adds of mul_lazy(one()) (the a*b columns are just the limbs of x); the caller has folded the missing factor
    // 2^(29N) into an earlier constant (ntt.hip.h: closing twiddles of the pass before the last).
    SV_HD Fp mont_reduce_lazy() const {
        uint32_t m[N];
        Fp r;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * N; k++) {
            if (k < N) acc += v[k];
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * P::MOD[j];
            }
            if (k < N) {
                m[k] = (0u - (uint32_t)acc) & LIMB_MASK;
                acc += m[k];
            } else {
                r.v[k - N] = (k == 2 * N - 1) ? (uint32_t)acc : ((uint32_t)acc & LIMB_MASK);
            }
            acc >>= 29;
        }
        return r;
    }
    // value < 2p with normalised limbs -> canonical
    SV_HD Fp reduce_lazy() const { return cond_sub(v); }

    // ---- conversions between representations
    // canonical integer -> internal Montgomery, and back
    SV_HD Fp int_to_mont() const { return *this * from_table(P::R2); }
    SV_HD Fp mont_to_int() const {
        Fp o = zero();
        o.v[0] = 1;
        return *this * o;
    }
    // reference memory form (a * 2^MEM_R_BITS) <-> internal form (a * 2^(29N))
    SV_HD Fp from_mem_mont() const { return *this * from_table(P::MEM2INT); }
    SV_HD Fp to_mem_mont() const { return *this * from_table(P::INT2MEM); }
    SV_HD static Fp from_u32(uint32_t x) {  // small integer -> internal Montgomery
        Fp a = zero();
        a.v[0] = x & LIMB_MASK;
        a.v[1] = x >> 29;
        return a.int_to_mont();
    }

    // x^e, e given as 32-bit words LSB first (plain square-and-multiply; exponents are public)
    SV_HD Fp pow_words(const uint32_t* e, int nwords) const {
        Fp res = one();
        for (int w = nwords - 1; w >= 0; w--)
            for (int bit = 31; bit >= 0; bit--) {
                res = res.sqr();
                if ((e[w] >> bit) & 1) res = res * *this;
            }
        return res;
    }
    SV_HD Fp pow_u64(uint64_t e) const {  // starts at the top set bit: small exponents cost ~log2(e) squarings
        if (e == 0) return one();
        int top = 63;
        while (!((e >> top) & 1)) top--;
        Fp res = *this;
        for (int bit = top - 1; bit >= 0; bit--) {
            res = res.sqr();
            if ((e >> bit) & 1) res = res * *this;
        }
        return res;
    }
    // Fermat inverse a^(p-2): the same residue as the reference's binary EEA (fp_256.rs:290-340), a != 0
    SV_HD Fp inverse() const {
        uint32_t pw[WORDS];
        from_table(P::MOD).pack(pw);
        uint32_t borrow = 2;  // p - 2 (the low word of both moduli is 1, so the borrow ripples)
        for (int i = 0; i < WORDS; i++) {
            uint32_t x = pw[i] - borrow;
            borrow = (pw[i] < borrow) ? 1u : 0u;
            pw[i] = x;
        }
        return pow_words(pw, WORDS);
    }
};

typedef Fp<FrP> fr_t;
typedef Fp<FqP> fq_t;
typedef fr_t::mem_t fr_mem_t;
typedef fq_t::mem_t fq_mem_t;
static_assert(sizeof(fr_mem_t) == 32 && sizeof(fq_mem_t) == 48, "field element sizes must match the Rust layout");

// ------------------------------------------------------------------------------------------
// Fq2 = Fq[u] / (u^2 + 5)   (fields/src/fp2.rs:57-60, curves/src/bls12_377/fq2.rs:58-69: NONRESIDUE = -5)
// Same interface as Fp so that ec.hip.h / msm.hip.h work for G2.
// ------------------------------------------------------------------------------------------
struct fq2_t {
    fq_t c0, c1;
    static constexpr int MEM_WORDS = 24;
    struct alignas(16) mem_t {
        fq_mem_t c0, c1;
    };
    SV_HD static fq2_t zero() { return {fq_t::zero(), fq_t::zero()}; }
    SV_HD static fq2_t one() { return {fq_t::one(), fq_t::zero()}; }
    SV_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    SV_HD bool operator==(const fq2_t& b) const { return c0 == b.c0 && c1 == b.c1; }
    SV_HD bool operator!=(const fq2_t& b) const { return !(*this == b); }
    SV_HD fq2_t operator+(const fq2_t& b) const { return {c0 + b.c0, c1 + b.c1}; }
    SV_HD fq2_t operator-(const fq2_t& b) const { return {c0 - b.c0, c1 - b.c1}; }
    SV_HD fq2_t neg() const { return {c0.neg(), c1.neg()}; }
    SV_HD fq2_t dbl() const { return {c0.dbl(), c1.dbl()}; }
    SV_HD static fq_t mul5(const fq_t& x) {
        fq_t x4 = x.dbl().dbl();
        return x4 + x;
    }
    // fp2.rs:404-410: c0 = a0 b0 + nr a1 b1, c1 = a0 b1 + a1 b0   (nr = -5)
    // Schoolbook with one Montgomery reduction per component (two-product columns on the signed accumulator of
    // Fp::diff_of_products): 4 x 169 product multiply-adds + 2 x 169 reduction ones = the 1 014 of Karatsuba's three full
    // products, without Karatsuba's five canonical additions / subtractions and with 18 fewer live limbs (no v0, v1, s).
    SV_HD fq2_t operator*(const fq2_t& b) const {
#if defined(SV_FQ2_KARATSUBA)
        fq_t v0 = c0 * b.c0, v1 = c1 * b.c1;
        fq_t s = (c0 + c1) * (b.c0 + b.c1);  // Karatsuba cross term
        return {v0 - mul5(v1), s - v0 - v1};
#else
        return {fq_t::diff_of_products(c0, b.c0, mul5(c1), b.c1), fq_t::diff_of_products(c0, b.c1, c1.neg(), b.c0)};
#endif
    }
    // fp2.rs:149-165 (same value)
    SV_HD fq2_t sqr() const {
        fq_t a = c0.sqr(), b = c1.sqr(), m = c0 * c1;
        return {a - mul5(b), m.dbl()};
    }
    SV_HD static fq2_t diff_of_products(const fq2_t& a, const fq2_t& b, const fq2_t& c, const fq2_t& d) { return a * b - c * d; }
    // fp2.rs:167-184: (c0 - c1 u) / (c0^2 + 5 c1^2) since u^2 = -5; a != 0
    SV_HD fq2_t inverse() const {
        const fq_t ninv = (c0.sqr() + mul5(c1.sqr())).inverse();
        return {c0 * ninv, (c1 * ninv).neg()};
    }
    SV_HD static fq2_t load(const void* p) {
        const uint8_t* q = (const uint8_t*)p;
        return {fq_t::load(q), fq_t::load(q + 48)};
    }
    SV_HD void store(void* p) const {
        uint8_t* q = (uint8_t*)p;
        c0.store(q);
        c1.store(q + 48);
    }
    SV_HD static fq2_t from_raw_words(const uint32_t* w) { return {fq_t::from_raw_words(w), fq_t::from_raw_words(w + 12)}; }
    SV_HD void to_raw_words(uint32_t* w) const {
        c0.to_raw_words(w);
        c1.to_raw_words(w + 12);
    }
};
static_assert(sizeof(fq2_t::mem_t) == 96, "Fq2 memory image");

}  // namespace sv
