// bench_callers.cpp - the reference prover's call pattern against the C ABI, without an interpreter in the way: T host threads (the
// rayon workers of sonic_pc/mod.rs:186-245) each issue the 14 proof-sized G1 MSMs of a Varuna proof (credits.aleo transfer_private
// sizes: |R| = 2^16, |K| = 2^17) as 14 SINGLE-INSTANCE calls of snarkvm_hip_msm_registered_ex over one registered SRS, device-resident
// scalars, `rounds` proofs per thread.  Reports pairs/s over all threads and how the in-library coalescer grouped the calls; every result
// is compared (after affine normalisation) with the same call issued alone before the timed region.
//   build: g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_callers.cpp -o tools/bench_callers
//          -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib
//   run:   tools/bench_callers [threads ...]      (SNARKVM_HIP_TUNING=coalesce=0 for the A/B)
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "snarkvm_hip.h"

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));    \
            exit(2);                                                   \
        }                                                              \
    } while (0)
#define RK(x)                                                                               \
    do {                                                                                    \
        RustError r_ = (x);                                                                 \
        if (r_.code) {                                                                      \
            fprintf(stderr, "%s: error %d: %s\n", #x, r_.code, r_.message ? r_.message : ""); \
            exit(3);                                                                        \
        }                                                                                   \
    } while (0)

struct call_t {
    size_t n0, n1;     // plaintext pairs over powers[0, n0), hiding pairs over the bases at nmax
    size_t scalar_off; // first scalar (in elements of the device pool)
};

int main(int argc, char** argv) {
    const int lgR = 16, lgK = 17;
    const size_t nR = (size_t)1 << lgR, nK = (size_t)1 << lgK, nmax = (size_t)1 << (lgK + 1), nbases = nmax + 8;
    std::vector<int> thread_counts;
    for (int i = 1; i < argc; i++) thread_counts.push_back(atoi(argv[i]));
    if (thread_counts.empty()) thread_counts = {1, 2, 4, 8, 16};
    const int rounds = 4;  // proofs per thread
    CK(hipSetDevice(0));
    void* d_bases = nullptr;
    CK(hipMalloc(&d_bases, nbases * 104));
    RK(snarkvm_hip_g1_generate_bases_device(d_bases, 1, nbases));
    snarkvm_hip_bases_t* h = nullptr;
    RK(snarkvm_hip_register_bases_windowed(&h, d_bases, nbases, 104, 1, 17, 15));
    CK(hipFree(d_bases));
    // a pool of random Fr images (< 2^252) the calls slice their polynomials from
    const size_t pool_n = nmax + 65536;
    std::vector<uint64_t> pool(pool_n * 4);
    uint64_t st = 0xC0FFEE;
    for (auto& w : pool) {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        w = z ^ (z >> 31);
    }
    for (size_t i = 0; i < pool_n; i++) pool[4 * i + 3] &= 0x0fffffffffffffffull;
    void* d_pool = nullptr;
    CK(hipMalloc(&d_pool, pool_n * 32));
    CK(hipMemcpy(d_pool, pool.data(), pool_n * 32, hipMemcpyHostToDevice));
    // the 14 commitments / openings of one proof (snarkvm_amd/proofs.py::ProofShape.pairs)
    const std::vector<call_t> shape = {{nR - 2, 2, 1},  {nR, 0, 11},     {nR - 1, 2, 21}, {nR, 0, 31},     {nK - 1, 0, 41}, {nK - 1, 0, 51}, {nK - 1, 0, 61},
                                       {nK - 2, 0, 3},  {nK, 0, 5},      {nR, 0, 9},      {nK, 0, 11},     {nK - 1, 0, 13}, {nR - 1, 0, 17}, {nK - 1, 0, 19}};
    size_t pairs_per_proof = 0;
    for (const call_t& c : shape) pairs_per_proof += c.n0 + c.n1;
    auto issue = [&](const call_t& c, size_t salt, void* out) {
        RK(snarkvm_hip_msm_registered_ex(out, h, 0, c.n0, nmax, c.n1, (const uint8_t*)d_pool + 32 * (c.scalar_off + salt), 1, 1, 0));
    };
    const int max_threads = 32;
    // reference results: every (salt, call) issued alone
    std::vector<uint8_t> want((size_t)max_threads * rounds * shape.size() * 104);
    {
        std::vector<uint8_t> proj(144);
        for (int s = 0; s < max_threads * rounds; s++)
            for (size_t k = 0; k < shape.size(); k++) {
                issue(shape[k], (size_t)s, proj.data());
                RK(snarkvm_hip_g1_to_affine(&want[((size_t)s * shape.size() + k) * 104], proj.data(), 1));
            }
    }
    printf("| threads | proofs | wall ms | pairs/s | ms per proof-equivalent | coalescer: batches | instances per batch | largest | results |\n|---|---|---|---|---|---|---|---|---|\n");
    for (int T : thread_counts) {
        if (T > max_threads) T = max_threads;
        std::vector<std::vector<uint8_t>> got(T, std::vector<uint8_t>((size_t)rounds * shape.size() * 144));
        snarkvm_hip_coalescer_stats(nullptr, 1);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int r = 0; r < rounds; r++)
                    for (size_t k = 0; k < shape.size(); k++) issue(shape[k], (size_t)(t * rounds + r), &got[t][((size_t)r * shape.size() + k) * 144]);
            });
        for (auto& x : th) x.join();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        uint64_t cs[4] = {0, 0, 0, 0};
        snarkvm_hip_coalescer_stats(cs, 0);
        size_t bad = 0;
        std::vector<uint8_t> aff(104);
        for (int t = 0; t < T; t++)
            for (int r = 0; r < rounds; r++)
                for (size_t k = 0; k < shape.size(); k++) {
                    RK(snarkvm_hip_g1_to_affine(aff.data(), &got[t][((size_t)r * shape.size() + k) * 144], 1));
                    if (memcmp(aff.data(), &want[((size_t)(t * rounds + r) * shape.size() + k) * 104], 97) != 0) bad++;
                }
        const double pairs = (double)T * rounds * pairs_per_proof;
        printf("| %d | %d | %.1f | %.3e | %.2f | %llu | %.2f | %llu | %s |\n", T, T * rounds, ms, pairs / (ms * 1e-3), ms / (T * rounds), (unsigned long long)cs[0],
               cs[0] ? (double)cs[1] / (double)cs[0] : 0.0, (unsigned long long)cs[2], bad ? "MISMATCH" : "all identical to the calls issued alone");
        fflush(stdout);
        if (bad) return 1;
    }
    snarkvm_hip_free_bases(h);
    return 0;
}
