// serde.hip.h - the reference's canonical wire / on-disk encoding of G1 points, decoded and encoded on the device
// (SURVEY.md §8f N3), so that an SRS file or a serialised committer key goes from its bytes straight into the MSM
// engine's base slots without a CPU pass.
//
// Encoding (curves/src/templates/macros.rs:66-140, fields/src/macros.rs:187-282, utilities/src/serialize/flags.rs:72-99):
//   uncompressed (96 B): x as 48 little-endian bytes of its CANONICAL integer (not Montgomery), then y likewise with the
//                        SWFlags in the two top bits of the last byte: bit 6 = infinity, bit 7 = "y is the larger root";
//                        Compress::No writes infinity or the default (no bit) and the reader only looks at the infinity bit.
//   compressed   (48 B): x with the same two flag bits; y is recovered as the square root of x^3 + 1 selected by bit 7
//                        (affine.rs:140-150: `y = if (y < -y) ^ greatest { y } else { -y }`).
//   Both bits set is rejected (flags.rs:90-93); coordinates >= q are rejected (read_le -> from_bigint == None).
// `.usrs` files (parameters/src/mainnet/resources) are a u64 count followed by uncompressed points.
#pragma once
#include "ec.hip.h"
#include "ff.hip.h"

namespace sv {

enum { SERDE_BAD_FLAGS = 1, SERDE_NOT_CANONICAL = 2, SERDE_NOT_ON_CURVE = 4, SERDE_NOT_IN_SUBGROUP = 8 };

// (T - 1) / 2 with q - 1 = 2^46 T (fq.rs T_MINUS_ONE_DIV_TWO), the 2^46-th root of unity (fq.rs TWO_ADIC_ROOT_OF_UNITY,
// canonical integer) and r (fr.rs MODULUS); 32-bit words, little endian.  Re-derived from tests/golden/constants.json by
// tests/test_host_arith.py.
__device__ static const uint32_t FQ_T_MINUS_ONE_DIV_TWO[12] = {0x00010a11u, 0xba886000u, 0x90002e16u, 0xc45f7412u, 0x271e3de6u, 0xb3e601eau,
                                                              0x92763445u, 0x0b80d942u, 0x21d58c76u, 0x748c2f8au, 0x0000035cu, 0x00000000u};
__device__ static const uint32_t FQ_TWO_ADIC_ROOT_INT[12] = {0x94ff4419u, 0xca9d610du, 0x6386ae79u, 0xf86b201bu, 0x63e6bd3bu, 0x5808bf73u,
                                                            0x90d30280u, 0x48d30f28u, 0xc20ffe30u, 0x365126d9u, 0x7df7548du, 0x01760d08u};
__device__ static const uint32_t FR_MODULUS_WORDS[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu,
                                                        0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
static constexpr int FQ_TWO_ADICITY = 46;  // fq.rs TWO_ADICITY

// 12 words (48 bytes, any alignment) -> canonical-integer limbs; false if the integer is >= q
__device__ __forceinline__ bool fq_from_le_bytes(const uint8_t* b, uint32_t top_mask, fq_t& out_int) {
    uint32_t w[12];
#pragma unroll
    for (int k = 0; k < 12; k++) w[k] = (uint32_t)b[4 * k] | ((uint32_t)b[4 * k + 1] << 8) | ((uint32_t)b[4 * k + 2] << 16) | ((uint32_t)b[4 * k + 3] << 24);
    w[11] &= top_mask;
    uint32_t m[12];
    fq_t::from_table(FqP::MOD).pack(m);
    bool lt = false;  // w < q ?
#pragma unroll
    for (int k = 11; k >= 0; k--) {
        if (w[k] != m[k]) {
            lt = w[k] < m[k];
            break;
        }
    }
    out_int = fq_t::unpack(w);
    return lt;
}
__device__ __forceinline__ void fq_to_le_bytes(const fq_t& canonical_int, uint8_t flags, uint8_t* b) {
    uint32_t w[12];
    canonical_int.pack(w);
#pragma unroll
    for (int k = 0; k < 12; k++) {
        b[4 * k] = (uint8_t)w[k];
        b[4 * k + 1] = (uint8_t)(w[k] >> 8);
        b[4 * k + 2] = (uint8_t)(w[k] >> 16);
        b[4 * k + 3] = (uint8_t)(w[k] >> 24);
    }
    b[47] |= flags;
}
// a > b as canonical integers (`Ord for Fp384`, fp_384.rs:565-570)
__device__ __forceinline__ bool fq_int_gt(const fq_t& a, const fq_t& b) {
#pragma unroll
    for (int i = fq_t::N - 1; i >= 0; i--)
        if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
    return false;
}
// Tonelli-Shanks square root in Fq (internal Montgomery form); false for a non-residue.  Either root may come back -
// the caller picks by comparison, exactly like the reference does after its own sqrt (fields/src/macros.rs:85-180).
__device__ inline bool fq_sqrt(const fq_t& a, fq_t& root) {
    if (a.is_zero()) {
        root = a;
        return true;
    }
    uint32_t e[12];
#pragma unroll
    for (int k = 0; k < 12; k++) e[k] = FQ_T_MINUS_ONE_DIV_TWO[k];
    const fq_t w0 = a.pow_words(e, 12);  // a^((T-1)/2)
    fq_t x = a * w0;                     // a^((T+1)/2)
    fq_t b = x * w0;                     // a^T
    uint32_t zw[12];
#pragma unroll
    for (int k = 0; k < 12; k++) zw[k] = FQ_TWO_ADIC_ROOT_INT[k];
    fq_t z = fq_t::unpack(zw).int_to_mont();
    const fq_t one = fq_t::one();
    int v = FQ_TWO_ADICITY;
    while (b != one) {
        int k = 0;
        fq_t b2k = b;
        while (b2k != one) {
            b2k = b2k.sqr();
            k++;
            if (k == v) return false;  // order of b does not divide 2^(v-1): a is a non-residue
        }
        fq_t w = z;
        for (int j = 0; j < v - k - 1; j++) w = w.sqr();
        z = w.sqr();
        b = b * z;
        x = x * w;
        v = k;
    }
    root = x;
    return true;
}
__device__ inline bool g1_is_on_curve(const g1_aff_t& p) { return p.y.sqr() == p.x.sqr() * p.x + fq_t::one(); }
// r * P == infinity (the definition the reference's endomorphism shortcut is tested against, bls12_377/g1.rs:271-279)
__device__ inline bool g1_is_in_subgroup(const g1_aff_t& p) {
    g1_xyzz_t acc = g1_xyzz_t::inf();
    for (int w = 7; w >= 0; w--) {
        const uint32_t word = FR_MODULUS_WORDS[w];
        for (int bit = 31; bit >= 0; bit--) {
            acc = acc.dbl();
            if ((word >> bit) & 1) acc.add_affine(p);
        }
    }
    return acc.is_inf();
}

// One thread per point.  out_native (MSM base slots, infinity = all zero) and out_rust (Rust `G1Affine` memory image,
// 104-byte stride) are both optional.  status accumulates SERDE_* bits over all points.
static __global__ void g1_deserialize_kernel(const uint8_t* __restrict__ bytes, size_t n, int compressed, int validate, g1_aff_mem_t* out_native,
                                      uint8_t* out_rust, uint32_t* status) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t psz = compressed ? 48 : 96;
    const uint8_t* src = bytes + i * psz;
    const uint8_t fb = src[psz - 1];
    const bool f_pos = (fb >> 7) & 1, f_inf = (fb >> 6) & 1;
    uint32_t st = 0;
    if (f_pos && f_inf) st |= SERDE_BAD_FLAGS;
    fq_t xi, yi = fq_t::zero();
    g1_aff_t p;
    bool inf = f_inf;
    if (compressed) {
        if (!fq_from_le_bytes(src, 0x3fffffffu, xi)) st |= SERDE_NOT_CANONICAL;
        if (inf) {
            p = {fq_t::zero(), fq_t::one()};  // Affine::zero() = (0, 1, infinity) (affine.rs:57-59)
        } else {
            p.x = xi.int_to_mont();
            fq_t y;
            if (!fq_sqrt(p.x.sqr() * p.x + fq_t::one(), y)) {
                st |= SERDE_NOT_ON_CURVE;  // from_x_coordinate == None -> InvalidData
                y = fq_t::zero();
            }
            const fq_t ny = y.neg();
            const bool y_lt_ny = fq_int_gt(ny.mont_to_int(), y.mont_to_int());
            p.y = (y_lt_ny != f_pos) ? y : ny;
        }
    } else {
        if (!fq_from_le_bytes(src, 0xffffffffu, xi)) st |= SERDE_NOT_CANONICAL;
        if (!fq_from_le_bytes(src + 48, 0x3fffffffu, yi)) st |= SERDE_NOT_CANONICAL;
        p = {xi.int_to_mont(), yi.int_to_mont()};  // Affine::new(x, y, flags.is_infinity()) keeps x, y as read
    }
    if (validate && !inf && !st) {
        if (!g1_is_on_curve(p))
            st |= SERDE_NOT_ON_CURVE;
        else if (!g1_is_in_subgroup(p))
            st |= SERDE_NOT_IN_SUBGROUP;
    }
    if (st) atomicOr(status, st);
    if (out_native) store_aff<fq_t>(&out_native[i], inf ? g1_aff_t::inf() : p);
    if (out_rust) {
        uint32_t* dst = (uint32_t*)(out_rust + i * 104);
        uint32_t w[12];
        p.x.to_raw_words(w);
#pragma unroll
        for (int k = 0; k < 12; k++) dst[k] = w[k];
        p.y.to_raw_words(w);
#pragma unroll
        for (int k = 0; k < 12; k++) dst[12 + k] = w[k];
        dst[24] = inf ? 1u : 0u;
        dst[25] = 0;
    }
}

// Rust `G1Affine` records (stride bytes) -> canonical encoding
static __global__ void g1_serialize_kernel(const uint8_t* __restrict__ affine, size_t stride, size_t n, int compressed, uint8_t* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = (const uint32_t*)(affine + i * stride);
    uint32_t xw[12], yw[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        xw[k] = src[k];
        yw[k] = src[12 + k];
    }
    const bool inf = (src[24] & 0xffu) != 0;
    const fq_t x = fq_t::from_raw_words(xw), y = fq_t::from_raw_words(yw);
    const fq_t xi = x.mont_to_int(), yi = y.mont_to_int();
    if (compressed) {
        uint8_t* dst = out + i * 48;
        if (inf) {
            fq_to_le_bytes(fq_t::zero(), 1u << 6, dst);
        } else {
            const bool pos = fq_int_gt(yi, y.neg().mont_to_int());  // SWFlags::from_y_sign(y > -y)
            fq_to_le_bytes(xi, pos ? (uint8_t)(1u << 7) : (uint8_t)0, dst);
        }
    } else {
        uint8_t* dst = out + i * 96;
        fq_to_le_bytes(xi, 0, dst);
        fq_to_le_bytes(yi, inf ? (uint8_t)(1u << 6) : (uint8_t)0, dst + 48);
    }
}

// ---- G2: x.c0, x.c1, y.c0, y.c1 as 48-byte little-endian canonical integers, SWFlags in the top bits of the last one
// (fields/src/fp2.rs:425-455: c0 plain, c1 with the flags); 192 bytes uncompressed (e.g. `beta-h.usrs`), 96 bytes compressed
// (x only; y = the Fq2 square root of x^3 + b' selected by the sign flag, affine.rs:140-150 with `Ord for Fp2`, fp2.rs:240-250).
// G2 curve constant b' = (0, b1) (curves/src/bls12_377/g2.rs:92-113), canonical integer words of b1
__device__ static const uint32_t G2_B_C1_INT[12] = {0x9999999au, 0x1c9ed999u, 0x1ccccccdu, 0x0dd39e5cu, 0x3c6bf800u, 0x129207b6u,
                                                   0xcd5fd889u, 0xdc7b4f91u, 0x7460c589u, 0x43bd0373u, 0xdb0fd6f3u, 0x010222f6u};
// `Fp2::sqrt` (fields/src/fp2.rs:208-230; complex method, eprint 2012/685 algorithm 8).  false where the reference returns None
// - including an element of the base field that is a non-residue THERE (fp2.rs:210-212 only tries `c0.sqrt()`).
__device__ inline bool fq2_sqrt(const fq2_t& a, fq2_t& root) {
    if (a.c1.is_zero()) {
        fq_t r;
        if (!fq_sqrt(a.c0, r)) return false;
        root = {r, fq_t::zero()};
        return true;
    }
    const fq_t norm = a.c0.sqr() + fq2_t::mul5(a.c1.sqr());  // c0^2 - nonresidue * c1^2, nonresidue = -5
    fq_t alpha;
    if (!fq_sqrt(norm, alpha)) return false;  // legendre(norm) == QNR
    const fq_t two_inv = fq_t::from_u32(2).inverse();
    fq_t delta = (alpha + a.c0) * two_inv;
    fq_t c0;
    if (!fq_sqrt(delta, c0)) {  // delta is a non-residue: the other choice of alpha's sign
        delta = delta - alpha;
        if (!fq_sqrt(delta, c0)) return false;
    }
    if (c0.is_zero()) return false;  // cannot happen for c1 != 0
    root = {c0, a.c1 * two_inv * c0.inverse()};
    return true;
}
// a > b in the reference's order on Fp2 (fp2.rs:240-250): lexicographic, c1 first, canonical integers
__device__ __forceinline__ bool fq2_int_gt(const fq2_t& a, const fq2_t& b) {
    const fq_t a1 = a.c1.mont_to_int(), b1 = b.c1.mont_to_int();
    if (a1 != b1) return fq_int_gt(a1, b1);
    return fq_int_gt(a.c0.mont_to_int(), b.c0.mont_to_int());
}
__device__ inline fq2_t g2_curve_b() {
    uint32_t bw[12];
#pragma unroll
    for (int k = 0; k < 12; k++) bw[k] = G2_B_C1_INT[k];
    return {fq_t::zero(), fq_t::unpack(bw).int_to_mont()};
}
__device__ inline bool g2_is_on_curve(const aff_t<fq2_t>& p) {
    return p.y.sqr() == p.x.sqr() * p.x + g2_curve_b();
}
__device__ inline bool g2_is_in_subgroup(const aff_t<fq2_t>& p) {
    xyzz_t<fq2_t> acc = xyzz_t<fq2_t>::inf();
    for (int w = 7; w >= 0; w--) {
        const uint32_t word = FR_MODULUS_WORDS[w];
        for (int bit = 31; bit >= 0; bit--) {
            acc = acc.dbl();
            if ((word >> bit) & 1) acc.add_affine(p);
        }
    }
    return acc.is_inf();
}
static __global__ void g2_deserialize_kernel(const uint8_t* __restrict__ bytes, size_t n, int compressed, int validate, uint8_t* __restrict__ out_rust,
                                      uint32_t* status) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t psz = compressed ? 96 : 192;
    const uint8_t* src = bytes + i * psz;
    const uint8_t fb = src[psz - 1];
    const bool f_pos = (fb >> 7) & 1, f_inf = (fb >> 6) & 1;
    uint32_t st = 0;
    if (f_pos && f_inf) st |= SERDE_BAD_FLAGS;
    fq_t c[4];
    const int ncoord = compressed ? 2 : 4;
    for (int k = 0; k < ncoord; k++)
        if (!fq_from_le_bytes(src + 48 * k, k == ncoord - 1 ? 0x3fffffffu : 0xffffffffu, c[k])) st |= SERDE_NOT_CANONICAL;
    aff_t<fq2_t> p;
    if (compressed) {
        if (f_inf) {
            p = {fq2_t::zero(), fq2_t::one()};  // Affine::zero() = (0, 1, infinity)
        } else {
            p.x = {c[0].int_to_mont(), c[1].int_to_mont()};
            fq2_t y;
            if (!fq2_sqrt(p.x.sqr() * p.x + g2_curve_b(), y)) {
                st |= SERDE_NOT_ON_CURVE;  // from_x_coordinate == None -> InvalidData
                y = fq2_t::zero();
            }
            const fq2_t ny = y.neg();
            const bool y_lt_ny = fq2_int_gt(ny, y);
            p.y = (y_lt_ny != f_pos) ? y : ny;  // affine.rs:147
        }
    } else {
        p = {{c[0].int_to_mont(), c[1].int_to_mont()}, {c[2].int_to_mont(), c[3].int_to_mont()}};
    }
    if (validate && !f_inf && !st) {
        if (!g2_is_on_curve(p))
            st |= SERDE_NOT_ON_CURVE;
        else if (!g2_is_in_subgroup(p))
            st |= SERDE_NOT_IN_SUBGROUP;
    }
    if (st) atomicOr(status, st);
    uint32_t* dst = (uint32_t*)(out_rust + i * 200);  // Rust G2Affine: x (c0, c1), y (c0, c1), infinity, pad
    p.x.c0.to_raw_words(dst);
    p.x.c1.to_raw_words(dst + 12);
    p.y.c0.to_raw_words(dst + 24);
    p.y.c1.to_raw_words(dst + 36);
    dst[48] = f_inf ? 1u : 0u;
    dst[49] = 0;
}
static __global__ void g2_serialize_kernel(const uint8_t* __restrict__ affine, size_t stride, size_t n, int compressed, uint8_t* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = (const uint32_t*)(affine + i * stride);
    const bool inf = (src[48] & 0xffu) != 0;
    fq_t c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t w[12];
#pragma unroll
        for (int j = 0; j < 12; j++) w[j] = src[12 * k + j];
        c[k] = fq_t::from_raw_words(w);
    }
    if (compressed) {
        uint8_t* dst = out + i * 96;
        if (inf) {
            fq_to_le_bytes(fq_t::zero(), 0, dst);
            fq_to_le_bytes(fq_t::zero(), 1u << 6, dst + 48);
        } else {
            const fq2_t y = {c[2], c[3]};
            const bool pos = fq2_int_gt(y, y.neg());  // SWFlags::from_y_sign(y > -y)
            fq_to_le_bytes(c[0].mont_to_int(), 0, dst);
            fq_to_le_bytes(c[1].mont_to_int(), pos ? (uint8_t)(1u << 7) : (uint8_t)0, dst + 48);
        }
    } else {
        uint8_t* dst = out + i * 192;
#pragma unroll
        for (int k = 0; k < 4; k++) fq_to_le_bytes(c[k].mont_to_int(), (k == 3 && inf) ? (uint8_t)(1u << 6) : (uint8_t)0, dst + 48 * k);
    }
}

}  // namespace sv
