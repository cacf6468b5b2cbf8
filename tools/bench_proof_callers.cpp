// bench_proof_callers.cpp - the concurrent-caller mode of `bench.py --workload proofs64` WITHOUT an interpreter: T host threads each replay
// whole Varuna-proof-shaped call lists (snarkvm_amd/proofs.py::replay, same calls, sizes, operands and order: ~45 device-resident NTTs, the
// pointwise / division passes, 6 batched commitment rounds = 14 G1 MSMs of 2^16 - 2^17 pairs, optionally one 2^16 G2 MSM) over one registered
// SRS (17 tables x 15-bit windows), one proof per thread at a time.  Reports proofs/s and how the library's coalescer grouped the MSM calls;
// the 14 commitments of every proof are compared (affine) with a single-threaded replay of the same proof.
//   build: g++ -std=c++17 -O2 -pthread -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_proof_callers.cpp -o /tmp/bench_proof_callers
//          -L snarkvm_amd/lib -lsnarkvm_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/snarkvm_amd/lib -Wl,-rpath,/opt/rocm/lib
//   run:   /tmp/bench_proof_callers [g2 points file (200 B each, 2^16 of them) or -] [--scope] [threads ...]
//          --scope: every caller issues its proof inside ONE asynchronous scope (replay_scope below) instead of one synchronous call per step
//          --scope-await / --scope-await-in-stream: asynchronous scope, every round's commitments collected before the next round is issued (on further streams / on the scope's own)
//          --scope-sync: the same scope without SNARKVM_HIP_SCOPE_ASYNC_MSM: the commitment rounds are synchronous calls that meet in the coalescer
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "snarkvm_hip.h"

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(2);                                                \
        }                                                           \
    } while (0)
#define RK(x)                                                                                 \
    do {                                                                                      \
        RustError r_ = (x);                                                                   \
        if (r_.code) {                                                                        \
            fprintf(stderr, "%s: error %d: %s\n", #x, r_.code, r_.message ? r_.message : ""); \
            exit(3);                                                                          \
        }                                                                                     \
    } while (0)

static const int LG_R = 16, LG_K = 17, LG_G2 = 16;
static const size_t N_R = (size_t)1 << LG_R, N_K = (size_t)1 << LG_K, NMAX = (size_t)1 << (LG_K + 1);

struct keys_t {
    snarkvm_hip_bases_t* h = nullptr;
    snarkvm_hip_bases_g2_t* hg2 = nullptr;
    uint8_t* pool = nullptr;  // device: NMAX + 4096 Fr images (32 B each)
    uint64_t point[4];
};
struct workspace_t {
    uint8_t* v[4];  // a, b, c, d: NMAX elements each
    uint8_t* rows = nullptr;  // replay_scope: 28 vectors of NMAX elements
    uint64_t rem[3][4];
    hipStream_t st;
    std::vector<uint8_t> results;  // 14 x 144 B per proof
    uint8_t g2_out[288];
};

static void replay(const keys_t& K, workspace_t& w, size_t salt, uint8_t* out14) {
    uint8_t *a = w.v[0], *b = w.v[1], *c = w.v[2], *d = w.v[3];
    size_t nout = 0;
    auto load = [&](uint8_t* v, size_t n, size_t shift) {  // a fresh "polynomial" of n coefficients from the pool, the rest zero
        CK(hipMemcpyAsync(v, K.pool + 32 * (shift + salt), 32 * n, hipMemcpyDeviceToDevice, w.st));
        if (n < NMAX) CK(hipMemsetAsync(v + 32 * n, 0, 32 * (NMAX - n), w.st));
        CK(hipStreamSynchronize(w.st));
    };
    auto ntt = [&](uint8_t* v, int lg, int dir, int type = 0) { RK(snarkvm_hip_ntt_device(v, (uint32_t)lg, 0, dir, type)); };
    auto ntt_batch = [&](std::vector<uint8_t*> vs, int lg, std::vector<int> dirs) {
        RK(snarkvm_hip_ntt_device_batch((void* const*)vs.data(), vs.size(), (uint32_t)lg, 0, dirs.data(), nullptr));
    };
    auto product = [&](uint8_t* x, uint8_t* y, int lg) {
        ntt_batch({x, y}, lg, {0, 0});
        RK(snarkvm_hip_fr_mul_device(x, x, y, (size_t)1 << lg));
        ntt(x, lg, 1);
    };
    struct poly_t {
        const void* p;
        size_t n, hiding;
    };
    auto commit_round = [&](std::vector<poly_t> polys) {
        const size_t k = polys.size();
        std::vector<size_t> off0(k, 0), n0(k), off1(k, NMAX), n1(k);
        std::vector<const void*> ptrs(k);
        for (size_t i = 0; i < k; i++) ptrs[i] = polys[i].p, n0[i] = polys[i].n, n1[i] = polys[i].hiding;
        RK(snarkvm_hip_msm_registered_batch_ex(out14 + 144 * nout, K.h, k, off0.data(), n0.data(), off1.data(), n1.data(), ptrs.data(), 1, 1, 0));
        nout += k;
    };
    // round 1
    load(a, N_R, 1), ntt(a, LG_R, 1), load(b, N_R, 2), ntt(b, LG_R, 0), commit_round({{a, N_R - 2, 2}});
    // round 2
    load(a, N_R, 10), load(b, N_R, 11), load(c, N_R, 12);
    ntt_batch({a, b, c}, LG_R, {1, 1, 1});
    CK(hipMemcpyAsync(d, c, 32 * NMAX, hipMemcpyDeviceToDevice, w.st));
    CK(hipStreamSynchronize(w.st));
    product(a, b, LG_R + 1);
    RK(snarkvm_hip_fr_vec_op(1, a, a, d, nullptr, nullptr, 2 * N_R, 1));
    RK(snarkvm_hip_fr_divide_by_vanishing(b, c, a, 2 * N_R, N_R, 1));
    commit_round({{b, N_R, 0}});
    // round 3
    for (size_t m = 0; m < 3; m++) load(a, N_R, 20 + m), ntt(a, LG_R, 1), load(b, N_R, 30 + m), product(a, b, LG_R + 1);
    commit_round({{a, N_R - 1, 2}, {b, N_R, 0}});
    // round 4
    uint8_t* r4[3] = {a, b, c};
    for (size_t m = 0; m < 3; m++) {
        uint8_t* v = r4[m];
        load(v, N_K, 40 + m), ntt(v, LG_K, 1), load(d, N_K, 50 + m), ntt(d, LG_K, 1), load(d, N_K, 60 + m), ntt(d, LG_K, 1, 1);
        if (m == 0) load(d, N_K, 70), product(v, d, LG_K + 1);
    }
    commit_round({{a, N_K - 1, 0}, {b, N_K - 1, 0}, {c, N_K - 1, 0}});
    // round 5
    commit_round({{K.pool + 32 * (3 + salt), N_K - 2, 0}, {K.pool + 32 * (5 + salt), N_K, 0}, {K.pool + 32 * (9 + salt), N_R, 0}, {K.pool + 32 * (11 + salt), N_K, 0}});
    // openings
    const size_t os[3] = {13, 17, 19}, on[3] = {N_K, N_R, N_K};
    uint8_t* oq[3] = {b, c, d};
    uint64_t rem[4];
    for (int i = 0; i < 3; i++) {
        load(a, on[i], os[i]);
        RK(snarkvm_hip_fr_divide_by_linear(oq[i], rem, a, on[i], K.point, 1));
    }
    commit_round({{b, N_K - 1, 0}, {c, N_R - 1, 0}, {d, N_K - 1, 0}});
    if (K.hg2) RK(snarkvm_hip_msm_g2_registered(w.g2_out, K.hg2, 0, (size_t)1 << LG_G2, K.pool + 32 * (23 + salt), 1, 0));
    if (nout != 14) {
        fprintf(stderr, "replay: %zu results\n", nout);
        exit(4);
    }
}

// The same proof issued for overlap (snarkvm_amd/proofs.py::replay_single): ONE deferred-synchronisation scope per proof with
// SNARKVM_HIP_SCOPE_ASYNC_MSM | _STABLE_INPUTS - no call waits for the GPU, the operand copies go onto the scope's own stream, the
// independent transforms of a round are one batched call, the commitment rounds run on further streams and are finished by scope_end.
// w.rows: 28 vectors of NMAX elements (every committed vector keeps its row until the proof is done).
static uint32_t g_scope_flags = SNARKVM_HIP_SCOPE_ASYNC_MSM | SNARKVM_HIP_SCOPE_STABLE_INPUTS;  // --scope-sync: 0 (synchronous, coalesced commitment rounds)
static bool g_await_rounds = false;  // --scope-await: snarkvm_hip_scope_collect(out) after every commitment round (the Fiat-Shamir order of a real prover)
static bool g_in_stream = false;  // --scope-await-in-stream: those awaited rounds run on the scope's own stream (SNARKVM_HIP_SCOPE_MSM_IN_STREAM), the G2 MSM on a further one
static void replay_scope(const keys_t& K, workspace_t& w, size_t salt, uint8_t* out14) {
    size_t nout = 0;
    auto row = [&](int r) { return w.rows + (size_t)r * NMAX * 32; };
    RK(snarkvm_hip_scope_begin_ex(K.pool, g_scope_flags));
    hipStream_t st = (hipStream_t)snarkvm_hip_scope_stream();
    auto load = [&](int r, size_t n, size_t shift, int count = 1, size_t zero_to = 0) {
        for (int i = 0; i < count; i++) {
            CK(hipMemcpyAsync(row(r + i), K.pool + 32 * (shift + i + salt), 32 * n, hipMemcpyDeviceToDevice, st));
            if (zero_to > n) CK(hipMemsetAsync(row(r + i) + 32 * n, 0, 32 * (zero_to - n), st));
        }
    };
    auto ntt = [&](std::vector<int> rows, int lg, int dir, int type = 0) {
        std::vector<void*> ptrs;
        for (int r : rows) ptrs.push_back(row(r));
        std::vector<int> dirs(rows.size(), dir), types(rows.size(), type);
        RK(snarkvm_hip_ntt_device_batch(ptrs.data(), ptrs.size(), (uint32_t)lg, 0, dirs.data(), types.data()));
    };
    auto mul = [&](int x, int y, int lg, size_t count = 1) { RK(snarkvm_hip_fr_vec_op_strided(2, row(x), row(x), row(y), nullptr, nullptr, (size_t)1 << lg, count, NMAX)); };
    struct poly_t {
        const void* p;
        size_t n, hiding;
    };
    auto commit_round = [&](std::vector<poly_t> polys) {
        const size_t k = polys.size();
        std::vector<size_t> off0(k, 0), n0(k), off1(k, NMAX), n1(k);
        std::vector<const void*> ptrs(k);
        for (size_t i = 0; i < k; i++) ptrs[i] = polys[i].p, n0[i] = polys[i].n, n1[i] = polys[i].hiding;
        RK(snarkvm_hip_msm_registered_batch_ex(out14 + 144 * nout, K.h, k, off0.data(), n0.data(), off1.data(), n1.data(), ptrs.data(), 1, 1, 0));
        if (g_await_rounds && g_scope_flags) RK(snarkvm_hip_scope_collect(out14 + 144 * nout));
        nout += k;
    };
    if (K.hg2) RK(snarkvm_hip_msm_g2_registered(w.g2_out, K.hg2, 0, (size_t)1 << LG_G2, K.pool + 32 * (23 + salt), 1, 0));
    if (g_in_stream && g_await_rounds && g_scope_flags) RK(snarkvm_hip_scope_set_flags(g_scope_flags | SNARKVM_HIP_SCOPE_MSM_IN_STREAM));  // the awaited rounds on the scope's own stream
    load(26, N_R, 1, 2);  // round 1
    ntt({26}, LG_R, 1), ntt({27}, LG_R, 0);
    commit_round({{row(26), N_R - 2, 2}});
    load(0, N_R, 10, 3, 2 * N_R);  // round 2
    ntt({0, 1, 2}, LG_R, 1);
    ntt({0, 1}, LG_R + 1, 0), mul(0, 1, LG_R + 1), ntt({0}, LG_R + 1, 1);
    RK(snarkvm_hip_fr_vec_op(1, row(0), row(0), row(2), nullptr, nullptr, 2 * N_R, 1));
    RK(snarkvm_hip_fr_divide_by_vanishing(row(1), row(3), row(0), 2 * N_R, N_R, 1));
    commit_round({{row(1), N_R, 0}});
    load(4, N_R, 20, 3, 2 * N_R), load(7, N_R, 30, 3, 2 * N_R);  // round 3
    ntt({4, 5, 6}, LG_R, 1);
    ntt({4, 7, 5, 8, 6, 9}, LG_R + 1, 0), mul(4, 7, LG_R + 1, 3), ntt({4, 5, 6}, LG_R + 1, 1);
    commit_round({{row(6), N_R - 1, 2}, {row(9), N_R, 0}});
    load(10, N_K, 40, 3, 2 * N_K), load(13, N_K, 50, 3), load(16, N_K, 60, 3), load(19, N_K, 70, 1, 2 * N_K);  // round 4
    ntt({10, 11, 12, 13, 14, 15}, LG_K, 1);
    ntt({16, 17, 18}, LG_K, 1, 1);
    ntt({10, 19}, LG_K + 1, 0), mul(10, 19, LG_K + 1), ntt({10}, LG_K + 1, 1);
    commit_round({{row(10), N_K - 1, 0}, {row(11), N_K - 1, 0}, {row(12), N_K - 1, 0}});
    commit_round({{K.pool + 32 * (3 + salt), N_K - 2, 0}, {K.pool + 32 * (5 + salt), N_K, 0}, {K.pool + 32 * (9 + salt), N_R, 0}, {K.pool + 32 * (11 + salt), N_K, 0}});
    const size_t os[3] = {13, 17, 19}, on[3] = {N_K, N_R, N_K};
    for (int i = 0; i < 3; i++) {
        load(20 + i, on[i], os[i]);
        RK(snarkvm_hip_fr_divide_by_linear(row(23 + i), w.rem[i], row(20 + i), on[i], K.point, 1));
    }
    commit_round({{row(23), N_K - 1, 0}, {row(24), N_R - 1, 0}, {row(25), N_K - 1, 0}});
    RK(snarkvm_hip_scope_end());
    if (nout != 14) {
        fprintf(stderr, "replay_scope: %zu results\n", nout);
        exit(4);
    }
}

int main(int argc, char** argv) {
    const char* g2file = argc > 1 ? argv[1] : "-";
    std::vector<int> thread_counts;
    bool scope_mode = false;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--scope"))
            scope_mode = true;  // callers issue every proof inside one asynchronous scope (replay_scope)
        else if (!strcmp(argv[i], "--scope-await"))
            scope_mode = true, g_await_rounds = true;  // asynchronous scope, but every round's commitments are awaited before the next round is issued
        else if (!strcmp(argv[i], "--scope-await-in-stream"))
            scope_mode = true, g_await_rounds = true, g_in_stream = true;
        else if (!strcmp(argv[i], "--scope-sync"))
            scope_mode = true, g_scope_flags = 0;  // a scope per proof for the transforms and passes; the commitment rounds are synchronous calls (coalescer)
        else
            thread_counts.push_back(atoi(argv[i]));
    }
    if (thread_counts.empty()) thread_counts = {1, 4, 8, 16};
    const int nproofs = 64;
    CK(hipSetDevice(0));
    keys_t K;
    {
        const size_t nb = NMAX + 8;
        void* d_bases = nullptr;
        CK(hipMalloc(&d_bases, nb * 104));
        RK(snarkvm_hip_g1_generate_bases_device(d_bases, 1, nb));
        RK(snarkvm_hip_register_bases_windowed(&K.h, d_bases, nb, 104, 1, 17, 15));
        CK(hipFree(d_bases));
    }
    if (strcmp(g2file, "-") != 0) {
        FILE* f = fopen(g2file, "rb");
        std::vector<uint8_t> pts(((size_t)200) << LG_G2);
        if (!f || fread(pts.data(), 1, pts.size(), f) != pts.size()) {
            fprintf(stderr, "cannot read %zu bytes of G2 points from %s\n", pts.size(), g2file);
            return 5;
        }
        fclose(f);
        RK(snarkvm_hip_register_bases_g2(&K.hg2, pts.data(), (size_t)1 << LG_G2, 200, 17, 15));
    }
    const size_t pool_n = NMAX + 4096;
    {
        std::vector<uint64_t> pool(pool_n * 4);
        uint64_t st = 0xC0FFEE;
        for (auto& w : pool) {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            w = z ^ (z >> 31);
        }
        for (size_t i = 0; i < pool_n; i++) pool[4 * i + 3] &= 0x0fffffffffffffffull;  // < 2^252 < r: valid Montgomery images
        CK(hipMalloc((void**)&K.pool, pool_n * 32));
        CK(hipMemcpy(K.pool, pool.data(), pool_n * 32, hipMemcpyHostToDevice));
        memcpy(K.point, &pool[4 * 7], 32);
    }
    const int max_threads = 16;
    std::vector<workspace_t> ws(max_threads);
    for (auto& w : ws) {
        for (auto& v : w.v) CK(hipMalloc((void**)&v, NMAX * 32));
        CK(hipMalloc((void**)&w.rows, 28 * NMAX * 32));
        CK(hipMemset(w.rows, 0, 28 * NMAX * 32));
        CK(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
    }
    // reference: every proof replayed alone on workspace 0, commitments normalised
    std::vector<uint8_t> want((size_t)nproofs * 14 * 104), raw(14 * 144);
    const auto tr0 = std::chrono::steady_clock::now();
    for (int p = 0; p < nproofs; p++) {
        replay(K, ws[0], (size_t)p, raw.data());
        RK(snarkvm_hip_g1_to_affine(&want[(size_t)p * 14 * 104], raw.data(), 14));
    }
    const double serial_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count();
    printf("one caller, %d proofs one after the other (including the normalisation of the reference results): %.2f ms per proof\n\n", nproofs, serial_ms / nproofs);
    printf("callers: %s\n\n", !scope_mode ? "one synchronous call per step (replay)"
                                   : g_in_stream ? "one asynchronous scope per proof, every round's commitments awaited (snarkvm_hip_scope_collect) before the next round, the rounds on the scope's own stream (replay_scope, --scope-await-in-stream)"
                                   : g_await_rounds ? "one asynchronous scope per proof, every round's commitments awaited (snarkvm_hip_scope_collect) before the next round (replay_scope, --scope-await)"
                                   : g_scope_flags ? "one SNARKVM_HIP_SCOPE_ASYNC_MSM | _STABLE_INPUTS scope per proof (replay_scope)"
                                                   : "a scope per proof for the transforms and passes, synchronous commitment rounds through the coalescer (replay_scope, --scope-sync)");
    printf("| caller threads | proofs | wall ms | proofs/s | ms per proof | coalescer: batches | instances per batch | largest | results |\n|---|---|---|---|---|---|---|---|---|\n");
    for (int T : thread_counts) {
        if (T > max_threads) T = max_threads;
        std::vector<uint8_t> got((size_t)nproofs * 14 * 144);
        auto run = [&](workspace_t& w, size_t p, uint8_t* out) { scope_mode ? replay_scope(K, w, p, out) : replay(K, w, p, out); };
        for (int t = 0; t < T; t++) run(ws[t], 1000 + t, raw.data());  // warm-up: one proof per worker
        snarkvm_hip_coalescer_stats(nullptr, 1);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int p = t; p < nproofs; p += T) run(ws[t], (size_t)p, &got[(size_t)p * 14 * 144]);
            });
        for (auto& x : th) x.join();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        uint64_t cs[4] = {0, 0, 0, 0};
        snarkvm_hip_coalescer_stats(cs, 0);
        std::vector<uint8_t> aff((size_t)nproofs * 14 * 104);
        RK(snarkvm_hip_g1_to_affine(aff.data(), got.data(), (size_t)nproofs * 14));
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)nproofs * 14; i++)
            if (memcmp(&aff[i * 104], &want[i * 104], 97) != 0) bad++;
        printf("| %d | %d | %.1f | %.1f | %.2f | %llu | %.2f | %llu | %s |\n", T, nproofs, ms, nproofs / (ms * 1e-3), ms / nproofs, (unsigned long long)cs[0],
               cs[0] ? (double)cs[1] / (double)cs[0] : 0.0, (unsigned long long)cs[2], bad ? "MISMATCH" : "all 14 commitments of every proof identical to the serial replay");
        fflush(stdout);
        if (bad) return 1;
    }
    return 0;
}
