// ntt.hip.h - radix-2 NTT / iNTT over BLS12-377 Fr for gfx950.
//
// Replaces (behaviour, not code): sppark's NTT::Base / NTT_internal / bit_rev as called from
// algorithms/cuda/cuda/snarkvm.cu:154-186 and polynomial.cuh:104-266, and the CPU transforms of
// algorithms/src/fft/domain.rs:374-443 (in_order_fft / ifft / coset_ifft), :691-773 (io/oi helpers).
//
// Structure (MI355X-first): a 2^lg transform is split into at most three passes of radix <= 2^8
// ("four-step" decomposition applied recursively).  One workgroup stages a [2^a rows] x [T columns]
// tile of 32-byte elements in LDS (<= 64 KiB), runs the a radix-2 DIF stages there with __syncthreads
// between stages, and writes the tile back:
//   * non-last pass: in place (same addresses), each element multiplied by the inter-pass twiddle
//     w_L^(inner*k) looked up from a two-level power table (2 x 4096 entries, L2-resident);
//   * last pass: rows are contiguous; the output index is digit-reversed so the result lands in
//     natural (NN) order.  This pass is out-of-place (another workgroup still needs the slots it
//     would overwrite), hence the ping-pong with a scratch buffer in ntt_run().
// Every pass reads and writes each element exactly once in >= 256-byte contiguous runs, so HBM traffic is
// passes * 2 * 32 * n bytes (algorithmic minimum 2 * 32 * n: SURVEY.md 8d).  Twiddles are never
// streamed from HBM: per-stage twiddles come from a 256-entry table of w_512 powers staged in LDS.
//
// Data stays in the reference's memory form (Montgomery, R = 2^256) throughout; see ff.hip.h for why the
// 29-bit-limb arithmetic needs no conversion on this (linear) path.
#pragma once
#include <mutex>
#include <type_traits>
#include <vector>

#include "ff.hip.h"
#include "frs.hip.h"
#include "tuning.hip.h"

namespace sv {

static constexpr int NTT_LG_MAX = 26;     // two-level tables cover exponents < 2^26 (2 GiB vectors; larger domains: the caller's CPU path)
static constexpr int NTT_TW_BITS = 13;    // w^e = hi[e >> 13] * lo[e & 8191]
static constexpr int NTT_TW_SIZE = 1 << NTT_TW_BITS;
static constexpr int NTT_MAX_RADIX_LG = 9;  // passes of radix <= 2^8 up to 2^24; 2^25 / 2^26 use radix-2^9 passes on narrower tiles
static constexpr int NTT_LOCAL = 1 << (NTT_MAX_RADIX_LG - 1);  // per-stage twiddles: powers of w_512 staged in LDS

// order / direction / type enums: algorithms/cuda/src/lib.rs:22-40
enum { NTT_NN = 0, NTT_NR = 1, NTT_RN = 2, NTT_RR = 3 };
enum { NTT_FORWARD = 0, NTT_INVERSE = 1 };
enum { NTT_STANDARD = 0, NTT_COSET = 1 };

// Device-resident tables (internal Montgomery form, packed 32 B per entry).
struct ntt_tables_t {
    fr_mem_t* pow_lo[2];   // [dir][4096]  W^(+-i)          W = primitive 2^24-th root of unity
    fr_mem_t* pow_hi[2];   // [dir][4096]  W^(+-4096 i)
    fr_mem_t* local[2];    // [dir][NTT_LOCAL]   w_512^(+-i)
    fr_mem_t* g_lo[2];     // [0]: g^i  [1]: g^-i          g = 22 (fr.rs:126-135)
    fr_mem_t* g_hi[2];     //      g^(+-4096 i)
    fr_mem_t* size_inv;    // [25]  (2^lg)^-1
    fr_mem_t* consts;      // scratch for the set-up kernels
};

// TWO_ADIC_ROOT_OF_UNITY (fr.rs:115-120), memory form (a * 2^256), 32-bit words
__device__ static const uint32_t FR_TWO_ADIC_ROOT_MEM[8] = {0xda3ad648u, 0xaf80da4du, 0xfc381dacu, 0x5e223adbu,
                                                             0xb2f92525u, 0x03ba0666u, 0x3befb0ceu, 0x0f906c5bu};

// consts[0] = W, [1] = W^-1, [2] = g, [3] = g^-1, then size_inv[0..24]
static __global__ void ntt_setup_consts(ntt_tables_t t) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t w8[8];
    for (int i = 0; i < 8; i++) w8[i] = FR_TWO_ADIC_ROOT_MEM[i];
    fr_t w = fr_t::unpack(w8).from_mem_mont();          // 2^47-th root, internal form
    for (int i = 0; i < 47 - NTT_LG_MAX; i++) w = w.sqr();  // -> primitive 2^NTT_LG_MAX-th root
    fr_t g = fr_t::from_u32(22);
    w.store(&t.consts[0]);
    w.inverse().store(&t.consts[1]);
    g.store(&t.consts[2]);
    g.inverse().store(&t.consts[3]);
    fr_t half = fr_t::from_u32(2).inverse();
    fr_t cur = fr_t::one();
    for (int lg = 0; lg <= NTT_LG_MAX; lg++) {
        cur.store(&t.size_inv[lg]);
        cur = cur * half;
    }
}
// lo[i] = b^i, hi[i] = b^(4096 i) for the four bases; local[dir][i] = (W^+-1)^(65536 i)
static __global__ void ntt_fill_tables(ntt_tables_t t) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NTT_TW_SIZE) return;
    for (int which = 0; which < 4; which++) {
        fr_t b = fr_t::load(&t.consts[which]);
        fr_mem_t* lo = which < 2 ? t.pow_lo[which] : t.g_lo[which - 2];
        fr_mem_t* hi = which < 2 ? t.pow_hi[which] : t.g_hi[which - 2];
        b.pow_u64((uint64_t)i).store(&lo[i]);
        b.pow_u64((uint64_t)i << NTT_TW_BITS).store(&hi[i]);
        if (which < 2 && i < NTT_LOCAL) b.pow_u64((uint64_t)i << (NTT_LG_MAX - NTT_MAX_RADIX_LG)).store(&t.local[which][i]);
    }
}

struct ntt_pass_t {
    const fr_mem_t* in;
    fr_mem_t* out;
    int lg_n;        // transform size
    int a;           // log2 radix of this pass
    int s;           // log2 inner stride (non-last pass); 0 for the last pass
    int lgT;         // log2 tile width
    int last;        // 1: last pass
    int a1;          // last pass: log2 size of the leading digit (tile dimension); 0 for a single-pass transform
    int lg_mid;      // last pass: log2 size of the middle digit (three-pass transforms), else 0
    int dir;         // NTT_FORWARD / NTT_INVERSE
    int coset_pre;   // multiply input j by g^j        (forward coset, first pass)
    int scale_post;  // last pass: 0 none, 1 * n^-1, 2 * g^-k n^-1   (inverse / coset inverse)
    int tw_shift;    // non-last: twiddle exponent = (inner * k) << tw_shift
    const fr_mem_t* tw_full;  // non-last, optional: the closing twiddles of this pass, tw_full[(k << s) + inner] (one product instead of two)
    int reduce_only; // last pass: the previous pass' closing table already carries 2^261 (and n^-1 when inverse): the final
                     // multiplication by one / n^-1 shrinks to a bare Montgomery reduction
};

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

__device__ __forceinline__ fr_t tw_lookup(const fr_mem_t* lo, const fr_mem_t* hi, uint32_t e) {
    fr_t a = fr_t::load(&lo[e & (NTT_TW_SIZE - 1)]);
    uint32_t h = e >> NTT_TW_BITS;
    if (h == 0) return a;
    return a * fr_t::load(&hi[h]);
}

// ------------------------------------------------------------------------------------------
// v2 pass kernel: radix-4 butterflies in registers + lazy reduction.
//   * two DIF stages per LDS round trip (a = 8: 4 groups, 3 round trips instead of 8), the first group reads
//     straight from global memory and the last one writes straight back;
//   * the tile lives in LDS as 29-bit limbs, 9 consecutive words per element (no pack / unpack between stages);
//   * butterflies use lazy arithmetic (ff.hip.h): sums are only carry-normalised, differences add 2^s * r, products
//     skip the conditional subtraction.  Bound: entering local stage s every value is < 2^s * r (inputs canonical),
//     so after a <= 8 stages values are < 256 r < 2^261 (9 limbs).  The closing multiplication (inter-pass twiddle,
//     1/n scaling, or the constant one) brings the value below 2r and one conditional subtraction makes it
//     canonical again before it is stored.  (Radix-2^9 passes, used from 2^25: the top limb holds the 262nd bit and the
//     closing product is < 2.17 r - two conditional subtractions.)
// ------------------------------------------------------------------------------------------
struct ntt_lds_t {
    uint32_t* data;  // E elements of 9 limbs each (stride 9 words is coprime to the bank count; the limb offsets are immediates)
    uint32_t* tw;    // 9 planes of NL = 2^(a-1) limbs: the powers of this pass' own root w_(2^a) (internal form)
    int NL;
    int E;
    template <class E>  // fr_t (unsigned limbs) or frs_t (signed limbs): nine 32-bit words either way
    __device__ __forceinline__ E get(int e) const {
        E x;
#pragma unroll
        for (int l = 0; l < 9; l++) x.v[l] = static_cast<typename std::remove_reference<decltype(x.v[0])>::type>(data[e * 9 + l]);
        return x;
    }
    template <class E>
    __device__ __forceinline__ void put(int e, const E& x) const {
#pragma unroll
        for (int l = 0; l < 9; l++) data[e * 9 + l] = (uint32_t)x.v[l];
    }
    __device__ __forceinline__ fr_t twiddle(int idx) const {
        fr_t x;
#pragma unroll
        for (int l = 0; l < 9; l++) x.v[l] = tw[l * NL + idx];
        return x;
    }
};

// One DIF stage inside a register group: x[lo_m], x[lo_m | bit] -> (u + v, (u - v + 2^s r) * w)
__device__ __forceinline__ void lazy_butterfly(fr_t& u, fr_t& v, int s, int tw_idx, const ntt_lds_t& L) {
    uint32_t kp[9];
    fr_t::mod_shl(kp, s);  // 2^s * r
    const fr_t sum = fr_t::add_lazy(u, v);
    fr_t dif = fr_t::sub_lazy(u, v, kp);
    if (tw_idx != 0) dif = dif.mul_lazy(L.twiddle(tw_idx));
    u = sum;
    v = dif;
}

// The arithmetic of a pass as a policy: `ntt_arith_u` = the unsigned lazy routines of round 3 (tuning ntt_signed=0), `ntt_arith_s` =
// the signed limbs of frs.hip.h (default).  tw(): a word of the ff.hip.h twiddle tables (w * 2^261) in the form mul() wants.
struct ntt_arith_u {
    typedef fr_t elem;
    __device__ __forceinline__ static elem from_canonical(const fr_t& x) { return x; }
    __device__ __forceinline__ static void butterfly(elem& u, elem& v, int s, int tw_idx, const ntt_lds_t& L) { lazy_butterfly(u, v, s, tw_idx, L); }
    __device__ __forceinline__ static fr_t tw(const fr_t& w_int) { return w_int; }
    __device__ __forceinline__ static fr_t one() { return fr_t::one(); }
    __device__ __forceinline__ static elem mul(const elem& x, const fr_t& w) { return x.mul_lazy(w); }
    __device__ __forceinline__ static elem reduce_only(const elem& x) { return x.mont_reduce_lazy(); }
    // a radix-2^9 pass ends with values < 2^9 r = 1.17 * 2^261: the closing product is < 2.17 r, one more conditional subtraction
    __device__ __forceinline__ static fr_t finish(const elem& y, int a) {
        fr_t z = y.reduce_lazy();
        if (a > 8) z = z.reduce_lazy();
        return z;
    }
};
struct ntt_arith_s {
    typedef frs_t elem;
    __device__ __forceinline__ static elem from_canonical(const fr_t& x) { return frs_t::from_canonical(x); }
    // (u, v) -> (u + v, (u - v) w): the sum carry-normalised, the difference raw into the product (frs.hip.h); w = 1: the difference
    // is only normalised
    __device__ __forceinline__ static void butterfly(elem& u, elem& v, int s, int tw_idx, const ntt_lds_t& L) {
        (void)s;
        const elem sum = frs_t::add_norm(u, v);
        if (tw_idx != 0) {
            v = frs_t::mul(frs_t::sub_raw(u, v), L.twiddle(tw_idx));
        } else {
            elem neg;
#pragma unroll
            for (int i = 0; i < 9; i++) neg.v[i] = -v.v[i];
            v = frs_t::add_norm(u, neg);
        }
        u = sum;
    }
    __device__ __forceinline__ static fr_t tw(const fr_t& w_int) { return frs_t::twiddle_form(w_int); }
    __device__ __forceinline__ static fr_t one() { return fr_t::from_table(FrS::C290); }
    // the closing product of a pass: a normalised value against a table word (frs.hip.h: hide_range)
    __device__ __forceinline__ static elem mul(const elem& x, const fr_t& w) { return frs_t::mul(x.hide_range(), w); }
    __device__ __forceinline__ static elem reduce_only(const elem& x) { return frs_t::reduce_only(x); }
    __device__ __forceinline__ static fr_t finish(const elem& y, int a) {
        (void)a;
        return y.to_canonical();
    }
};

__device__ __forceinline__ fr_t load_fr_global(const fr_mem_t* p) {
    const uint4* src = (const uint4*)p;
    const uint4 x0 = src[0], x1 = src[1];
    const uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    return fr_t::unpack(w);
}
__device__ __forceinline__ void store_fr_global(fr_mem_t* p, const fr_t& x) {
    uint32_t w[8];
    x.pack(w);
    uint4* dst = (uint4*)p;
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// K stages (1 or 2) on the 2^K rows {base_row + m * row_step}; s = index of the first of them within the pass.
template <int K, class A>
__device__ __forceinline__ void dif_group(typename A::elem* x, int s, int a, int lo, const ntt_lds_t& L) {
    // rows of the group differ in the K bits just below bit (a - s); `lo` = the row bits below them
    const int lo_bits = a - s - K;
#pragma unroll
    for (int t = 0; t < K; t++) {
        const int bit = 1 << (K - 1 - t);  // group-local index bit paired at stage s + t
#pragma unroll
        for (int m = 0; m < (1 << K); m++) {
            if (m & bit) continue;
            // pos = row mod half, half = 2^(a - 1 - (s + t)): the group-local bits below `bit`, then `lo`
            const int pos = ((m & (bit - 1)) << lo_bits) | lo;
            const int tw_idx = pos << (s + t);  // exponent of w_(2^a)
            A::butterfly(x[m], x[m | bit], s + t, tw_idx, L);
        }
    }
}

// Batched launches (snarkvm_hip_ntt_device_batch: the independent transforms of a prover round, or of many proofs in lock step):
// blockIdx.y selects the vector.  Vector y lives at v[y]; its private scratch copy at scratch + (y << lg_n).  The pointers travel
// as kernel arguments (no device-side table to keep alive while calls are in flight); more than NTT_BATCH_MAX vectors = several launches.
static constexpr int NTT_BATCH_MAX = 48;
struct ntt_batch_t {
    fr_mem_t* v[NTT_BATCH_MAX];
    fr_mem_t* scratch;
    int in_scratch, out_scratch;
};
struct ntt_no_batch_t {};
template <bool BATCH, class A>
__global__ void __launch_bounds__(512) ntt_pass_kernel_v2(ntt_pass_t p, ntt_tables_t tb, typename std::conditional<BATCH, ntt_batch_t, ntt_no_batch_t>::type bt) {
    typedef typename A::elem elem;
    extern __shared__ uint32_t lds32[];
    if constexpr (BATCH) {
        fr_mem_t* vec = bt.v[blockIdx.y];
        fr_mem_t* scr = bt.scratch + ((size_t)blockIdx.y << p.lg_n);
        p.in = bt.in_scratch ? scr : vec;
        p.out = bt.out_scratch ? scr : vec;
    }
    const int R = 1 << p.a, T = 1 << p.lgT, E = R << p.lgT;
    ntt_lds_t L;
    L.data = lds32;
    L.tw = lds32 + 9 * E;
    L.NL = p.a ? (1 << (p.a - 1)) : 1;
    L.E = E;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t tile = blockIdx.x;

    // ---- addressing (as in v1)
    size_t in_base, in_rho_stride, in_col_stride;
    size_t inner0 = 0, d1_0 = 0, mid = 0;
    if (!p.last) {
        const size_t tiles_per_outer = (size_t)1 << (p.s - p.lgT);
        const size_t outer = tile / tiles_per_outer;
        inner0 = (tile % tiles_per_outer) << p.lgT;
        in_base = (outer << (p.a + p.s)) + inner0;
        in_rho_stride = (size_t)1 << p.s;
        in_col_stride = 1;
    } else {
        const size_t tiles_per_mid = (size_t)1 << (p.a1 - p.lgT);
        mid = tile / tiles_per_mid;
        d1_0 = (tile % tiles_per_mid) << p.lgT;
        in_base = (d1_0 << (p.lg_n - p.a1)) + (mid << p.a);
        in_rho_stride = 1;
        in_col_stride = (size_t)1 << (p.lg_n - p.a1);
    }
    // local twiddles -> LDS planes
    for (int i = tid; i < L.NL; i += nthr) {  // w_(2^a)^i = w_512^(i << (9 - a))
        const fr_t w = A::tw(fr_t::load(&tb.local[p.dir][i << (NTT_MAX_RADIX_LG - p.a)]));
#pragma unroll
        for (int l = 0; l < 9; l++) L.tw[l * L.NL + i] = w.v[l];
    }
    __syncthreads();

    // ---- stage groups: pairs of stages, a single stage first when a is odd
    int s = 0;
    bool first = true;
    while (s < p.a) {
        const int K = ((p.a - s) & 1) ? 1 : 2;
        const bool last_group = (s + K == p.a);
        const int lo_bits = p.a - s - K;
        const int ngroups = E >> K;
        for (int gi = tid; gi < ngroups; gi += nthr) {
            // non-last passes keep `col` fastest (coalesced rows of T elements); the last pass keeps the row fastest
            int col, q;
            if (!p.last || !first) {
                col = gi & (T - 1);
                q = gi >> p.lgT;
            } else {
                q = gi & ((R >> K) - 1);
                col = gi >> (p.a - K);
            }
            const int lo = q & ((1 << lo_bits) - 1);
            const int hi = q >> lo_bits;
            const int row0 = (hi << (p.a - s)) | lo;
            const int row_step = 1 << lo_bits;
            elem x[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (m >= (1 << K)) break;
                const int row = row0 + m * row_step;
                if (first) {
                    const size_t g = in_base + row * in_rho_stride + col * in_col_stride;
                    fr_t xin = load_fr_global(&p.in[g]);
                    if (p.coset_pre) xin = xin * tw_lookup(tb.g_lo[0], tb.g_hi[0], (uint32_t)g);
                    x[m] = A::from_canonical(xin);
                } else {
                    x[m] = L.get<elem>(row * T + col);
                }
            }
            if (K == 2)
                dif_group<2, A>(x, s, p.a, lo, L);
            else
                dif_group<1, A>(x, s, p.a, lo, L);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (m >= (1 << K)) break;
                const int row = row0 + m * row_step;
                if (!last_group) {
                    L.put<elem>(row * T + col, x[m]);
                    continue;
                }
                // ---- closing multiplication + store; row `row` holds output digit k = bitrev_a(row)
                const uint32_t k = bitrev32((uint32_t)row, p.a);
                size_t g;
                elem y;
                if (!p.last) {
                    g = in_base + ((size_t)k << p.s) + col;
                    if (p.tw_full) {
                        y = A::mul(x[m], fr_t::load(&p.tw_full[((size_t)k << p.s) + inner0 + col]));  // the table holds the arithmetic's own form
                    } else {
                        // composed on the fly from the two-level power tables (no table fits: the first pass of a 2^26 transform)
                        const uint32_t expo = (uint32_t)(((inner0 + col) * (size_t)k) << p.tw_shift);
                        fr_t w = fr_t::load(&tb.pow_lo[p.dir][expo & (NTT_TW_SIZE - 1)]);
                        const uint32_t h = expo >> NTT_TW_BITS;
                        if (h) w = w * fr_t::load(&tb.pow_hi[p.dir][h]);
                        y = A::mul(x[m], A::tw(w));
                    }
                } else {
                    g = (d1_0 + col) + (((size_t)mid + ((size_t)k << p.lg_mid)) << p.a1);
                    y = x[m];
                    if (!p.reduce_only) y = A::mul(y, p.scale_post == 0 ? A::one() : A::tw(fr_t::load(&tb.size_inv[p.lg_n])));
                    if (p.scale_post == 2) {
                        y = A::mul(y, A::tw(fr_t::load(&tb.g_lo[1][(uint32_t)g & (NTT_TW_SIZE - 1)])));
                        const uint32_t h = (uint32_t)g >> NTT_TW_BITS;
                        if (h) y = A::mul(y, A::tw(fr_t::load(&tb.g_hi[1][h])));
                    }
                    if (p.reduce_only) y = A::reduce_only(y);
                }
                const fr_t z = A::finish(y, p.a);
                store_fr_global(&p.out[g], z);
            }
        }
        __syncthreads();
        s += K;
        first = false;
    }
}

// out[i] = in[bitrev(i)]   (sppark `bit_rev`, polynomial.cuh:128,189; domain.rs:797-804 derange)
static __global__ void ntt_bitrev_kernel(const fr_mem_t* in, fr_mem_t* out, int lg_n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << lg_n)) return;
    size_t r = lg_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - lg_n)) : 0;
    const uint4* s = (const uint4*)&in[r];
    uint4* d = (uint4*)&out[i];
    d[0] = s[0];
    d[1] = s[1];
}
// polynomial_inner_multiply (polynomial.cuh:36-45): out[i] = a[i] * b[i] (memory Montgomery form, R = 2^256).
// mont261(x, y) = x y 2^-261 = (a b 2^256) 2^-5, so `fix` = 2^(5 + 261) mod r restores the form (see ff.hip.h);
// with fix_later the factor is left for the caller to fold into a later constant.
static __global__ void fr_pointwise_mul_kernel(fr_mem_t* out, const fr_mem_t* a, const fr_mem_t* b, size_t n, int fix_now) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        fr_t x = fr_t::load(&a[i]) * fr_t::load(&b[i]);
        if (fix_now) x = x.from_mem_mont();  // * 2^266 * 2^-261 = * 2^5
        x.store(&out[i]);
    }
}
// Fr::to_bigint (fp_256.rs:380-413) / from_bigint (fp_256.rs:362-377) over a vector: kzg10 convert_to_bigints
static __global__ void fr_to_bigint_kernel(fr_mem_t* out, const fr_mem_t* in, size_t n, int to_bigint) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        fr_t x = fr_t::load(&in[i]);
        // memory Montgomery a*2^256 -> a : internal(x) = mem * 2^5 ... use explicit forms:
        //   to_bigint:   a = mont261(mem, 2^261 * 2^-256 = 2^5)
        //   from_bigint: a -> internal a*2^261 (times R2) -> memory a*2^256 (times INT2MEM)
        if (to_bigint) {
            fr_t c = fr_t::zero();
            c.v[0] = 32;  // 2^5 as a plain integer
            x = x * c;
        } else {
            x = x.int_to_mont().to_mem_mont();  // a -> a*2^261 -> a*2^256
        }
        x.store(&out[i]);
    }
}

// ------------------------------------------------------------------------------------------
// Host-side driver
// ------------------------------------------------------------------------------------------
struct ntt_plan_t {
    int npass;
    int a[3];
};
static inline ntt_plan_t ntt_make_plan(int lg) {
    ntt_plan_t pl;
    constexpr int R8 = 8;  // preferred pass radix: a [2^8 x 8] tile is 72 KiB of LDS, two workgroups per CU
    if (lg <= R8) {
        pl.npass = 1;
        pl.a[0] = lg;
        pl.a[1] = pl.a[2] = 0;
    } else if (lg <= 2 * R8) {
        pl.npass = 2;
        pl.a[0] = lg / 2;
        pl.a[1] = lg - pl.a[0];
        pl.a[2] = 0;
    } else {  // up to 3 * 9 = 27 bits; 2^25 and 2^26 get one or two radix-2^9 passes
        pl.npass = 3;
        pl.a[0] = lg / 3;
        pl.a[1] = (lg - pl.a[0]) / 2;
        pl.a[2] = lg - pl.a[0] - pl.a[1];
    }
    return pl;
}
// tile width (log2) of a pass of radix 2^a whose tile dimension offers `avail` bits: [2^a x 2^lgT] elements of 36 bytes must
// fit the 96 KiB the pass kernel may use
// Small transforms (the 2^14 - 2^18 domains of a proof) would fill only 8 ... 128 of the chip's 256 CUs with [2^a x 8] tiles, and a
// thread's work does not depend on the tile width (one radix-4 group per stage round): such passes take narrower tiles until
// the launch has ntt_min_tiles workgroups (tuning.hip.h; the data is L2 resident at these sizes, so the shorter
// coalesced runs of a narrow tile cost nothing).
static inline int ntt_min_tiles() {
    return tuning().ntt_min_tiles;
}
static inline int ntt_tile_lg(int a, int avail, int lg_n, size_t nvec = 1) {
    int lgT = avail < 3 ? avail : 3;
    while (lgT > 0 && a + lgT > 11) lgT--;
    while (lgT > 0 && (((size_t)1 << (lg_n - a - lgT)) * nvec) < (size_t)ntt_min_tiles()) lgT--;  // a batched launch covers the chip with wide tiles sooner
    return lgT;
}

// ---- full closing-twiddle tables ----------------------------------------------------------------------------------
// The closing multiplication of a non-last pass needs W^((inner * k) << tw_shift); composing it from the two 4096-entry
// tables costs a second Fr product per element, and the NTT is ALU-bound: skipping that product is worth 9 %
// of a 2^24 transform.  So the composed twiddles of a pass shape (a, s, direction) are materialised in HBM, in exactly the
// order the pass stores its outputs (coalesced 32-byte reads next to the 32-byte stores): 2^(a+s) entries - 512 MiB for the
// first pass of a 2^24 transform, 2 MiB for its second pass.  The tables live in a per-device cache bounded by
// SNARKVM_HIP_NTT_TW_MB (default 1536 MiB): least recently used tables that no running call holds are evicted; when nothing
// can be evicted (or tuning ntt_full_tw=0) the pass composes its twiddles on the fly.
struct ntt_tw_entry {
    fr_mem_t* p = nullptr;
    size_t bytes = 0;
    uint64_t last_use = 0;
    int users = 0;
};
struct ntt_tw_cache_t {
    ntt_tw_entry ptr[2][NTT_MAX_RADIX_LG + 1][NTT_LG_MAX + 1];
    // the table of the pass BEFORE the last one, with 2^261 (forward) or 2^261 / n (inverse) folded in, per transform size:
    // the last pass then ends with Fp::mont_reduce_lazy() instead of a product by one / by n^-1
    ntt_tw_entry prelast[2][NTT_LG_MAX + 1];
    std::mutex mu;
    size_t bytes = 0;
    uint64_t tick = 0;
    template <class Fn>
    void for_each(Fn fn) {
        for (auto& d : ptr)
            for (auto& a : d)
                for (auto& e : a) fn(e);
        for (auto& d : prelast)
            for (auto& e : d) fn(e);
    }
};
// what a transform needs from its caller: the stream, the device's small tables, its twiddle cache, and the list of cache
// entries the call holds until its stream has been synchronised (released by ntt_tw_release)
struct ntt_ctx_t {
    hipStream_t st;
    const ntt_tables_t* tb;
    ntt_tw_cache_t* cache;
    std::vector<void*>* leases;
};
// fold: 0 plain, 1 times 2^261, 2 times 2^261 * size_inv (size_inv points at the Montgomery form of n^-1)
// arith_signed: the table serves frs.hip.h (R = 2^290): plain entries carry 2^290 instead of 2^261, folded ones 2^580 instead of 2^522
static __global__ void ntt_fill_full_tw_kernel(fr_mem_t* __restrict__ out, int a, int s, int tw_shift, const fr_mem_t* __restrict__ lo,
                                        const fr_mem_t* __restrict__ hi, int fold, const fr_mem_t* __restrict__ size_inv, int arith_signed) {
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= ((size_t)1 << (a + s))) return;
    const uint32_t k = (uint32_t)(idx >> s), inner = (uint32_t)(idx & (((size_t)1 << s) - 1));
    fr_t t = tw_lookup(lo, hi, (inner * k) << tw_shift);
    if (fold) {
        t = t * fr_t::from_table(arith_signed ? FrS::C580 : FrP::R2);  // Montgomery form of (2^261 mod r): the stored word becomes t * 2^522 (signed: t * 2^580)
        if (fold == 2) t = t * fr_t::load(size_inv);
    } else if (arith_signed) {
        t = frs_t::twiddle_form(t);
    }
    t.store(&out[idx]);
}
static inline void ntt_tw_release_entries(ntt_tw_cache_t& cache, std::vector<void*>& leases) {
    std::lock_guard<std::mutex> lk(cache.mu);
    for (void* e : leases) ((ntt_tw_entry*)e)->users--;
    leases.clear();
}
// prelast_lg != 0: this is the pass before the last one of a 2^prelast_lg transform -> the folded variant
static inline const fr_mem_t* ntt_get_full_tw(const ntt_ctx_t& cx, int a, int s, int tw_shift, int dir, int prelast_lg = 0, bool* folded = nullptr) {
    if (folded) *folded = false;
    const int enabled = tuning().ntt_full_tw;
    if (!enabled || !cx.cache || a + s > NTT_LG_MAX || a > NTT_MAX_RADIX_LG) return nullptr;
    const int fold_enabled = tuning().ntt_fold;
    static const size_t cap = (getenv("SNARKVM_HIP_NTT_TW_MB") ? (size_t)atoll(getenv("SNARKVM_HIP_NTT_TW_MB")) : 1536) << 20;
    if (prelast_lg && !fold_enabled) prelast_lg = 0;
    ntt_tw_cache_t& cache = *cx.cache;
    std::lock_guard<std::mutex> lk(cache.mu);
    ntt_tw_entry& slot = prelast_lg ? cache.prelast[dir][prelast_lg] : cache.ptr[dir][a][a + s];
    if (!slot.p) {
        const size_t n = (size_t)1 << (a + s);
        const size_t need = n * sizeof(fr_mem_t);
        if (need > cap) return nullptr;
        while (cache.bytes + need > cap) {  // evict the least recently used table nobody holds
            ntt_tw_entry* lru = nullptr;
            cache.for_each([&](ntt_tw_entry& e) {
                if (e.p && e.users == 0 && (!lru || e.last_use < lru->last_use)) lru = &e;
            });
            if (!lru) return nullptr;  // everything is in use: compose on the fly
            (void)hipFree(lru->p);
            cache.bytes -= lru->bytes;
            *lru = ntt_tw_entry();
        }
        if (hipMalloc((void**)&slot.p, need) != hipSuccess) {
            (void)hipGetLastError();
            slot.p = nullptr;
            return nullptr;  // out of memory: compose on the fly
        }
        slot.bytes = need;
        cache.bytes += need;
        const int fold = !prelast_lg ? 0 : (dir == NTT_INVERSE ? 2 : 1);
        hipLaunchKernelGGL(ntt_fill_full_tw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.st, slot.p, a, s, tw_shift, cx.tb->pow_lo[dir],
                           cx.tb->pow_hi[dir], fold, (const fr_mem_t*)(cx.tb->size_inv + (prelast_lg ? prelast_lg : 0)), tuning().ntt_signed ? 1 : 0);
        (void)hipStreamSynchronize(cx.st);  // other streams may use the table from now on
    }
    slot.last_use = ++cache.tick;
    slot.users++;
    if (cx.leases) cx.leases->push_back(&slot);
    if (folded) *folded = prelast_lg != 0;
    return slot.p;
}

static inline void ntt_launch_pass(hipStream_t st, const ntt_pass_t& p, const ntt_tables_t& tb, const ntt_batch_t* bt = nullptr, unsigned nvec = 1) {
    const size_t E = (size_t)1 << (p.a + p.lgT);
    const size_t ntiles = ((size_t)1 << p.lg_n) / E;
    {
        int threads = (int)(E / 4);  // one radix-4 group per thread
        if (threads < 64) threads = 64;
        if (threads > 512) threads = 512;
        const size_t shmem = (9 * E + 9 * ((size_t)1 << (p.a ? p.a - 1 : 0))) * sizeof(uint32_t);  // a [2^8 x 8] tile + its twiddles: 78 KB, two workgroups per CU
        const bool sg = tuning().ntt_signed != 0;
        if (bt && sg)
            hipLaunchKernelGGL((ntt_pass_kernel_v2<true, ntt_arith_s>), dim3((unsigned)ntiles, nvec), dim3(threads), shmem, st, p, tb, *bt);
        else if (bt)
            hipLaunchKernelGGL((ntt_pass_kernel_v2<true, ntt_arith_u>), dim3((unsigned)ntiles, nvec), dim3(threads), shmem, st, p, tb, *bt);
        else if (sg)
            hipLaunchKernelGGL((ntt_pass_kernel_v2<false, ntt_arith_s>), dim3((unsigned)ntiles), dim3(threads), shmem, st, p, tb, ntt_no_batch_t{});
        else
            hipLaunchKernelGGL((ntt_pass_kernel_v2<false, ntt_arith_u>), dim3((unsigned)ntiles), dim3(threads), shmem, st, p, tb, ntt_no_batch_t{});
    }
}

// NN-order transform of 2^lg elements held in `data`; `scratch` is a second buffer of the same size.
// The result is left in `data`.
// `vecs` != nullptr: the same transform of `nvec` (<= NTT_BATCH_MAX) distinct vectors in one launch per pass (needs >= 2 passes, i.e.
// lg > 8; `scratch` then holds nvec * 2^lg elements); `data` is ignored.
static inline void ntt_run_nn(const ntt_ctx_t& cx, fr_mem_t* data, fr_mem_t* scratch, int lg, int dir, int type, fr_mem_t* const* vecs = nullptr,
                              unsigned nvec = 1) {
    hipStream_t st = cx.st;
    const ntt_tables_t& tb = *cx.tb;
    if (lg == 0) {
        // size-1 transform: identity (coset shift g^0 = 1, n^-1 = 1)
        return;
    }
    const ntt_plan_t pl = ntt_make_plan(lg);
    ntt_batch_t bt;
    if (vecs) {
        for (unsigned i = 0; i < nvec; i++) bt.v[i] = vecs[i];
        bt.scratch = scratch;
    }
    const int scale_post = (dir == NTT_INVERSE) ? (type == NTT_COSET ? 2 : 1) : 0;
    const int coset_pre = (dir == NTT_FORWARD && type == NTT_COSET) ? 1 : 0;
    int consumed = 0;
    bool folded = false;  // the pass before the last one used a table with 2^261 [/ n] folded in
    for (int k = 0; k < pl.npass; k++) {
        ntt_pass_t p;
        p.lg_n = lg;
        p.a = pl.a[k];
        p.dir = dir;
        p.coset_pre = (k == 0) ? coset_pre : 0;
        p.last = (k == pl.npass - 1);
        p.scale_post = p.last ? scale_post : 0;
        p.a1 = p.lg_mid = p.s = p.tw_shift = 0;
        p.tw_full = nullptr;
        p.reduce_only = 0;
        if (!p.last) {
            p.s = lg - consumed - p.a;
            p.lgT = ntt_tile_lg(p.a, p.s, lg, nvec);
            p.tw_shift = NTT_LG_MAX - (p.a + p.s);
            const bool prelast = (k == pl.npass - 2);
            bool f = false;
            p.tw_full = ntt_get_full_tw(cx, p.a, p.s, p.tw_shift, dir, prelast ? lg : 0, &f);
            folded = f;
        } else {
            p.reduce_only = folded ? 1 : 0;
            p.a1 = (pl.npass >= 2) ? pl.a[0] : 0;
            p.lg_mid = (pl.npass == 3) ? pl.a[1] : 0;
            p.lgT = ntt_tile_lg(p.a, p.a1, lg, nvec);
        }
        if (pl.npass == 1) {
            p.in = data;
            p.out = scratch;
        } else if (k == 0) {
            p.in = data;
            p.out = scratch;
        } else if (!p.last) {
            p.in = scratch;
            p.out = scratch;
        } else {
            p.in = scratch;
            p.out = data;
        }
        if (vecs) {
            bt.in_scratch = (p.in == scratch);
            bt.out_scratch = (p.out == scratch);
            ntt_launch_pass(st, p, tb, &bt, nvec);
        } else {
            ntt_launch_pass(st, p, tb);
        }
        consumed += p.a;
    }
    if (pl.npass == 1)
        (void)hipMemcpyAsync(data, scratch, sizeof(fr_mem_t) << lg, hipMemcpyDeviceToDevice, st);
}

// Full FFI semantics (any order): bit-reversed inputs/outputs are handled with an explicit permutation pass.
static inline void ntt_run(const ntt_ctx_t& cx, fr_mem_t* data, fr_mem_t* scratch, int lg, int order, int dir, int type) {
    hipStream_t st = cx.st;
    const size_t n = (size_t)1 << lg;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (order == NTT_RN || order == NTT_RR) {
        hipLaunchKernelGGL(ntt_bitrev_kernel, dim3(blocks), dim3(256), 0, st, data, scratch, lg);
        (void)hipMemcpyAsync(data, scratch, sizeof(fr_mem_t) * n, hipMemcpyDeviceToDevice, st);
    }
    ntt_run_nn(cx, data, scratch, lg, dir, type);
    if (order == NTT_NR || order == NTT_RR) {
        hipLaunchKernelGGL(ntt_bitrev_kernel, dim3(blocks), dim3(256), 0, st, data, scratch, lg);
        (void)hipMemcpyAsync(data, scratch, sizeof(fr_mem_t) * n, hipMemcpyDeviceToDevice, st);
    }
}

}  // namespace sv
