// histbench.hip - round-4 experiment: what bounds the scalar-read kernel (32 B per scalar read once, 12 x 22-bit signed digits recoded
// in registers, level-1 bins (7 bits) counted in LDS, one row of 12 x 128 counters per 2 048-scalar tile)?
// Variants of the same kernel on 2^24 random scalars, timed with HIP events, outputs compared with variant 0:
//   0  round-3 kernel (one tile per workgroup, one LDS histogram, atomicAdd per digit)
//   1  read only (loads + a xor reduction: the bandwidth floor of this access pattern)
//   2  read + recode + digit extraction, no LDS atomics (xor reduction of the bins)
//   3  like 0 with R private copies of the histogram (lane & (R - 1)), summed at the end     [R = 2, 4, 8]
//   4  like 0 with 1 024 scalars per workgroup of 256 threads (more, smaller workgroups)
//   5  like 0 with 256 threads x 8 scalars
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/histbench.hip -o /tmp/histbench ; run: /tmp/histbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));               \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static constexpr int C = 22, ROWS = 12, HB = 7, LB = 14, B1 = 128, KEYS = ROWS * B1;
struct bias_t {
    uint32_t w[10];
};

__device__ __forceinline__ void recode(const uint4& lo, const uint4& hi, const bias_t& b, uint32_t* s) {
    s[0] = lo.x, s[1] = lo.y, s[2] = lo.z, s[3] = lo.w, s[4] = hi.x, s[5] = hi.y, s[6] = hi.z, s[7] = hi.w, s[8] = 0, s[9] = 0, s[10] = 0;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        carry += (uint64_t)s[k] + b.w[k];
        s[k] = (uint32_t)carry;
        carry >>= 32;
    }
}
template <int R>
__device__ __forceinline__ uint32_t digit(const uint32_t* s) {
    const int bit = C * R, wi = bit >> 5, sh = bit & 31;
    const uint32_t x = sh ? __builtin_amdgcn_alignbit(s[wi + 1], s[wi], (uint32_t)sh) : s[wi];
    return x & ((1u << C) - 1);
}
__device__ __forceinline__ bool bin_of(uint32_t u, uint32_t& bin) {
    const int dv = (int)u - (1 << (C - 1));
    if (dv == 0) return false;
    bin = (uint32_t)((dv < 0 ? -dv : dv) - 1) >> LB;
    return true;
}
template <class Fn, int... Rs>
__device__ __forceinline__ void for_rows(const uint32_t* s, Fn fn, std::integer_sequence<int, Rs...>) {
    (fn(Rs, digit<Rs>(s)), ...);
}

// THREADS threads, SPT scalars per thread, REP histogram copies, MODE 0 count / 1 read only / 2 no atomics
template <int THREADS, int SPT, int REP, int MODE>
__global__ void __launch_bounds__(THREADS) hist_kernel(const uint4* __restrict__ scalars, uint32_t* __restrict__ cnt, size_t n, bias_t bias, uint32_t* sink) {
    extern __shared__ uint32_t hist[];
    constexpr int TILE = THREADS * SPT;
    const uint32_t t = blockIdx.x;
    uint4 lo[SPT], hi[SPT];
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        const size_t i = (size_t)t * TILE + (size_t)q * THREADS + threadIdx.x;
        lo[q] = scalars[2 * i];
        hi[q] = scalars[2 * i + 1];
    }
    if (MODE == 1) {
        uint32_t x = 0;
#pragma unroll
        for (int q = 0; q < SPT; q++) x ^= lo[q].x ^ lo[q].y ^ lo[q].z ^ lo[q].w ^ hi[q].x ^ hi[q].y ^ hi[q].z ^ hi[q].w;
        if (x == sink[1]) sink[0] = x;
        return;
    }
    if (MODE == 0) {
        for (uint32_t i = threadIdx.x; i < KEYS * REP; i += THREADS) hist[i] = 0;
        __syncthreads();
    }
    uint32_t x = 0;
    const uint32_t copy = (REP > 1) ? (threadIdx.x & (REP - 1)) : 0u;
#pragma unroll
    for (int q = 0; q < SPT; q++) {
        uint32_t s[11];
        recode(lo[q], hi[q], bias, s);
        for_rows(s, [&](int r, uint32_t u) {
            uint32_t bin;
            if (bin_of(u, bin)) {
                if (MODE == 0)
                    atomicAdd(&hist[(r * B1 + bin) * REP + copy], 1u);  // copies interleaved: different copies never share a bank
                else
                    x += bin * 131u + (uint32_t)r;
            }
        }, std::make_integer_sequence<int, ROWS>{});
    }
    if (MODE == 2) {
        if (x == sink[1]) sink[0] = x;
        return;
    }
    __syncthreads();
    // 2 048-scalar rows of counters: a workgroup of TILE scalars adds its counts into the row of its 2 048-tile
    const size_t row = ((size_t)t * TILE) / 2048;
    for (uint32_t i = threadIdx.x; i < KEYS; i += THREADS) {
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < REP; c++) v += hist[i * REP + c];
        if (TILE == 2048)
            cnt[row * KEYS + i] = v;
        else
            atomicAdd(&cnt[row * KEYS + i], v);
    }
}

template <int THREADS, int SPT, int REP, int MODE>
static float run(const char* name, const uint4* d_s, uint32_t* d_cnt, size_t n, const bias_t& bias, uint32_t* d_sink, std::vector<uint32_t>* ref) {
    constexpr int TILE = THREADS * SPT;
    const unsigned blocks = (unsigned)(n / TILE);
    const size_t lds = MODE == 0 ? (size_t)KEYS * REP * 4 : 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        if (TILE != 2048) CK(hipMemsetAsync(d_cnt, 0, (n / 2048) * KEYS * 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((hist_kernel<THREADS, SPT, REP, MODE>), dim3(blocks), dim3(THREADS), lds, 0, d_s, d_cnt, n, bias, d_sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    const char* ok = "-";
    if (MODE == 0) {
        std::vector<uint32_t> got((n / 2048) * KEYS);
        CK(hipMemcpy(got.data(), d_cnt, got.size() * 4, hipMemcpyDeviceToHost));
        if (ref->empty()) {
            *ref = got;
            ok = "reference";
        } else {
            ok = got == *ref ? "identical" : "DIFFERENT";
        }
    }
    printf("%-44s %8.1f us  %6.0f GB/s  frac of 8 TB/s %.3f   counts: %s\n", name, best * 1e3, 32.0 * n / (best * 1e-3) / 1e9, 32.0 * n / (best * 1e-3) / 8e12, ok);
    return best;
}

int main() {
    const size_t n = (size_t)1 << 24;
    std::vector<uint32_t> h(n * 8);
    uint64_t st = 0x1234567;
    for (size_t i = 0; i < n * 8; i++) {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h[i] = (uint32_t)(z ^ (z >> 31));
        if ((i & 7) == 7) h[i] &= 0x0fffffffu;  // < 2^252
    }
    uint4* d_s;
    uint32_t *d_cnt, *d_sink;
    CK(hipMalloc(&d_s, n * 32));
    CK(hipMalloc(&d_cnt, (n / 2048) * KEYS * 4));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMemcpy(d_s, h.data(), n * 32, hipMemcpyHostToDevice));
    bias_t bias = {};
    for (int w = 0; w < ROWS; w++) {
        const int bit = C - 1 + C * w;
        bias.w[bit / 32] |= 1u << (bit % 32);
    }
    CK(hipFuncSetAttribute((const void*)hist_kernel<512, 4, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, KEYS * 8 * 4));
    CK(hipFuncSetAttribute((const void*)hist_kernel<1024, 2, 8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, KEYS * 8 * 4));
    CK(hipMemset(d_sink, 0x5a, 64));
    std::vector<uint32_t> ref;
    run<512, 4, 1, 0>("0  round 3: 512 x 4, one histogram", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 4, 1, 1>("1  read only (512 x 4)", d_s, d_cnt, n, bias, d_sink, &ref);
    run<256, 4, 1, 1>("1  read only (256 x 4)", d_s, d_cnt, n, bias, d_sink, &ref);
    run<256, 8, 1, 1>("1  read only (256 x 8)", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 4, 1, 2>("2  read + recode + digits, no atomics", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 4, 2, 0>("3  2 histogram copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 4, 4, 0>("3  4 histogram copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 4, 8, 0>("3  8 histogram copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<256, 4, 1, 0>("4  256 x 4 (1 024-scalar workgroups)", d_s, d_cnt, n, bias, d_sink, &ref);
    run<256, 4, 4, 0>("4  256 x 4, 4 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<256, 8, 1, 0>("5  256 x 8", d_s, d_cnt, n, bias, d_sink, &ref);
    run<256, 8, 4, 0>("5  256 x 8, 4 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 2, 1, 0>("6  1024 x 2", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 2, 2, 0>("6  1024 x 2, 2 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 2, 4, 0>("6  1024 x 2, 4 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 2, 8, 0>("6  1024 x 2, 8 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 2, 1, 2>("6  1024 x 2, no atomics", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 2, 1, 1>("6  1024 x 2, read only", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 1, 1, 0>("7  1024 x 1 (rows of 2 048 by global atomics)", d_s, d_cnt, n, bias, d_sink, &ref);
    run<1024, 1, 4, 0>("7  1024 x 1, 4 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 2, 4, 0>("8  512 x 2, 4 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    run<512, 1, 4, 0>("8  512 x 1, 4 copies", d_s, d_cnt, n, bias, d_sink, &ref);
    return 0;
}
