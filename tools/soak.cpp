// soak.cpp - randomized soak of the host runtime (csrc/runtime.hip.h: lanes, ticket coalescer, thread-local scopes with their further streams, pending
// finishes, deferred frees, the thread-exit clean-up) through the C ABI alone: T threads issue a seeded random mix of EVERY entry-point family for a
// given time, and every result is compared with the same call issued alone (a single-threaded pass over all (operation, variant) pairs before the
// threads start).  The reference's host logic is 314 lines behind a blocking token channel (algorithms/cuda/cuda/snarkvm.cu:73-312); this runtime is
// ~2 300 lines of hand-rolled concurrency, which is what this tool exists for.
//
//   soak <seconds> [threads=16] [seed=1] [logical devices=1]      (tools/soak.py builds it - plain, or with -fsanitize=thread - and keeps the report)
//
// Families:  stateless FFI (snarkvm_msm / _ntt / _polymul on host buffers) | registered sync (host scalars, device scalars, two base ranges, Montgomery
// scalars) | registered batch | concurrent proof-sized calls (meet in the coalescer by themselves) | scopes: synchronous, ASYNC_MSM (+ STABLE_INPUTS,
// + MSM_IN_STREAM), scope_collect of one call, host-operand calls inside a scope, a scope abandoned by a thread that exits | device memory calls |
// a deliberately bad request (must fail with a code and a message, and disturb nobody) every few calls.
// Exit code 0 = zero mismatches, zero unexpected errors, no thread stalled for 120 s.  One JSON line on stdout.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "snarkvm_hip.hpp"

struct Fr { uint64_t l[4]; };
struct G1Affine { uint64_t x[6], y[6]; uint8_t infinity; uint8_t pad[7]; };
struct G1Projective { uint64_t x[6], y[6], z[6]; };
static_assert(sizeof(G1Affine) == 104 && sizeof(G1Projective) == 144 && sizeof(Fr) == 32, "Rust layouts");

using Bytes = std::vector<uint8_t>;
using snarkvm_hip::check;
using snarkvm_hip::DeviceBuffer;

static uint64_t splitmix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- shared, read-only after set-up -------------------------------------------------------------------------------------------
static constexpr size_t NB = ((size_t)1 << 16) + 256;   // registered bases
static constexpr size_t NS = ((size_t)1 << 17) + 8192;  // pool of 250-bit values: canonical scalars AND valid Fr Montgomery images
static std::vector<G1Affine> g_bases;
static std::vector<Fr> g_pool;
static DeviceBuffer g_dpool;
static snarkvm_hip_bases_t* g_h = nullptr;
static const size_t MSM_N[4] = {1500, 5000, 20000, 65536};
static constexpr int V = 8;  // variants per operation

enum Op { FFI_MSM, FFI_NTT, FFI_POLYMUL, REG_HOST, REG_DEV_EX, REG_BATCH, SCOPE_SYNC, SCOPE_ASYNC, SCOPE_COLLECT, SCOPE_HOSTCALL, DEVMEM, NOPS };
static const char* OP_NAME[NOPS] = {"ffi_msm", "ffi_ntt", "ffi_polymul", "registered_host_scalars", "registered_device_scalars_ex", "registered_batch_ex",
                                    "scope_sync", "scope_async_msm", "scope_collect", "scope_host_operand_calls", "device_memory"};

static Bytes affine_of(const G1Projective* p, size_t n) {
    Bytes out(n * sizeof(G1Affine));
    check(snarkvm_hip_g1_to_affine(out.data(), p, n));
    return out;
}
static void append(Bytes& b, const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }

struct ThreadCtx {
    DeviceBuffer work;  // 3 rows of 2^17 Fr
    static constexpr size_t ROW = ((size_t)1 << 17) * 32;
    ThreadCtx() : work(3 * ROW) {}
    void* row(int r) const { return work.at((size_t)r * ROW); }
};

// One operation, variant v: returns a canonical image of everything it produced.
static Bytes run_op(int op, int v, ThreadCtx& tc) {
    Bytes out;
    const size_t n = MSM_N[v & 3], off = 7 * (size_t)v, soff = 11 * (size_t)v + 3;
    switch (op) {
    case FFI_MSM: {
        G1Projective r;
        check(snarkvm_msm(&r, g_bases.data() + off, n, g_pool.data() + soff, sizeof(G1Affine)));
        return affine_of(&r, 1);
    }
    case FFI_NTT: {
        const uint32_t lg = 10 + 2 * (uint32_t)(v & 3);
        std::vector<Fr> x(g_pool.begin() + soff, g_pool.begin() + soff + ((size_t)1 << lg));
        check(snarkvm_ntt(x.data(), lg, NN, (v & 4) ? Inverse : Forward, (v & 1) ? Coset : Standard));
        append(out, x.data(), x.size() * 32);
        return out;
    }
    case FFI_POLYMUL: {
        const uint32_t lg = 11 + 2 * (uint32_t)(v % 3);
        const size_t half = (size_t)1 << (lg - 1);
        std::vector<std::vector<Fr>> polys = {std::vector<Fr>(g_pool.begin() + soff, g_pool.begin() + soff + half - (size_t)v),
                                              std::vector<Fr>(g_pool.begin() + soff + 999, g_pool.begin() + soff + 999 + half)};
        Fr zero{};
        std::vector<Fr> prod = snarkvm_hip::polymul<Fr>((size_t)1 << lg, polys, {}, zero);
        append(out, prod.data(), prod.size() * 32);
        return out;
    }
    case REG_HOST: {
        G1Projective r;
        check(snarkvm_hip_msm_registered(&r, g_h, off, n, g_pool.data() + soff, 0, 0));
        return affine_of(&r, 1);
    }
    case REG_DEV_EX: {  // two base ranges, scalars on the device, Montgomery flag on odd variants
        G1Projective r;
        check(snarkvm_hip_msm_registered_ex(&r, g_h, off, n - 3, 100 + off, 3, g_dpool.at(32 * soff), 1, v & 1, 0));
        return affine_of(&r, 1);
    }
    case REG_BATCH: {
        G1Projective r[3];
        const size_t o0[3] = {off, 0, 5}, n0[3] = {n, 4096, MSM_N[(v + 1) & 3]}, o1[3] = {64, 0, 9}, n1[3] = {2, 0, 1};
        const void* sc[3] = {g_dpool.at(32 * soff), g_dpool.at(32 * (soff + 17)), g_dpool.at(32 * (soff + 4242))};
        check(snarkvm_hip_msm_registered_batch_ex(r, g_h, 3, o0, n0, o1, n1, sc, 1, 1, 0));
        return affine_of(r, 3);
    }
    case SCOPE_SYNC: {  // produce a polynomial, transform, square pointwise, transform back - one wait at the end
        const uint32_t lg = 12 + (uint32_t)(v & 3);
        const size_t m = (size_t)1 << lg;
        {
            snarkvm_hip::Scope scope(tc.row(0));
            tc.work.fill(0, 0, 2 * m * 32);
            tc.work.copy_from(0, g_dpool.at(32 * soff), m * 32);
            check(snarkvm_hip_ntt_device(tc.row(0), lg + 1, NN, Forward, Standard));
            check(snarkvm_hip_fr_vec_op(2, tc.row(0), tc.row(0), tc.row(0), nullptr, nullptr, 2 * m, 1));
            check(snarkvm_hip_ntt_device(tc.row(0), lg + 1, NN, Inverse, (v & 4) ? Coset : Standard));
            scope.end();
        }
        out.resize(2 * m * 32);
        tc.work.download(out.data(), out.size());
        return out;
    }
    case SCOPE_ASYNC: {  // commitments enqueued behind the scope's transforms; the input row is REUSED unless STABLE_INPUTS is set
        const uint32_t flags = SNARKVM_HIP_SCOPE_ASYNC_MSM | ((v & 1) ? SNARKVM_HIP_SCOPE_STABLE_INPUTS : 0) | ((v & 2) ? SNARKVM_HIP_SCOPE_MSM_IN_STREAM : 0);
        G1Projective r[2];
        {
            snarkvm_hip::Scope scope(tc.row(0), flags);
            tc.work.copy_from(0, g_dpool.at(32 * soff), 65536 * 32);
            check(snarkvm_hip_ntt_device(tc.row(0), 16, NN, Inverse, Standard));
            check(snarkvm_hip_msm_registered_ex(&r[0], g_h, off, n, 0, 0, tc.row(0), 1, 1, 0));
            const int second = (v & 1) ? 1 : 0;  // stable inputs: the next polynomial gets its own row
            tc.work.copy_from((size_t)second * ThreadCtx::ROW, g_dpool.at(32 * (soff + 555)), 65536 * 32);
            check(snarkvm_hip_ntt_device(tc.row(second), 16, NN, Forward, Standard));
            check(snarkvm_hip_msm_registered_ex(&r[1], g_h, 1, n, 0, 0, tc.row(second), 1, 1, 0));
            scope.end();
        }
        return affine_of(r, 2);
    }
    case SCOPE_COLLECT: {  // three calls enqueued, the second collected first, the rest by scope_end
        G1Projective r[3];
        memset(r, 0, sizeof r);
        snarkvm_hip::Scope scope(tc.row(0), SNARKVM_HIP_SCOPE_ASYNC_MSM | SNARKVM_HIP_SCOPE_STABLE_INPUTS);
        for (int k = 0; k < 3; k++) {
            tc.work.copy_from((size_t)k * ThreadCtx::ROW, g_dpool.at(32 * (soff + 1000 * (size_t)k)), MSM_N[(v + k) & 3] * 32);
            check(snarkvm_hip_msm_registered(&r[k], g_h, off + (size_t)k, MSM_N[(v + k) & 3], tc.row(k), 1, 0));
        }
        scope.collect(&r[1]);
        const Bytes mid = affine_of(&r[1], 1);  // (a host-operand call inside the scope: takes the scope's lane after its flush)
        scope.end();
        out = affine_of(r, 3);
        append(out, mid.data(), mid.size());
        return out;
    }
    case SCOPE_HOSTCALL: {  // host-operand calls in the middle of an asynchronous scope: they wait for what is pending and run on the scope's lane
        G1Projective r[3];
        std::vector<Fr> x(g_pool.begin() + soff, g_pool.begin() + soff + 4096);
        {
            snarkvm_hip::Scope scope(tc.row(0), SNARKVM_HIP_SCOPE_ASYNC_MSM);
            tc.work.copy_from(0, g_dpool.at(32 * soff), n * 32);
            check(snarkvm_hip_msm_registered(&r[0], g_h, off, n, tc.row(0), 1, 0));
            check(snarkvm_ntt(x.data(), 12, NN, Forward, Standard));
            check(snarkvm_hip_msm_registered(&r[1], g_h, off, 3000, g_pool.data() + soff, 0, 0));
            check(snarkvm_msm(&r[2], g_bases.data() + off, 2000, g_pool.data() + soff, sizeof(G1Affine)));
            scope.end();
        }
        out = affine_of(r, 3);
        append(out, x.data(), x.size() * 32);
        return out;
    }
    case DEVMEM: {
        const size_t bytes = 4096 + 32 * 1000 * (size_t)v;
        DeviceBuffer a(bytes), b(2 * bytes);
        a.upload(g_pool.data() + soff, bytes);
        b.fill(0, 0x5A, 2 * bytes);
        b.copy_from(bytes / 2, a.data(), bytes);
        out.resize(2 * bytes);
        b.download(out.data(), out.size());
        return out;
    }
    }
    return out;
}

// a request that must be refused - with a code and a message - and leave everybody else alone
static bool bad_request(unsigned k, ThreadCtx& tc) {
    G1Projective r;
    RustError e{0, nullptr};
    switch (k % 5) {
    case 0: e = snarkvm_hip_msm_registered(&r, g_h, NB - 10, 100, g_pool.data(), 0, 0); break;        // range past the end
    case 1: e = snarkvm_hip_ntt_device(tc.row(0), 27, NN, Forward, Standard); break;                     // lg > 26
    case 2: e = snarkvm_hip_memcpy_d2d(tc.row(0), (char*)tc.row(0) + 64, 4096); break;                   // overlapping ranges
    case 3: e = snarkvm_hip_msm_registered(&r, g_h, 0, 100, (const void*)0x1000, 1, 0); break;          // "device" scalars that are not device memory
    case 4: e = snarkvm_msm(&r, g_bases.data(), 2000, g_pool.data(), 96); break;                        // bad ffi_affine_sz
    }
    const bool refused = e.code != 0 && e.message != nullptr;
    std::free(e.message);
    return refused;
}

// a thread that opens an asynchronous scope, enqueues work and ENDS without scope_end: its lanes must return to the pool
static void abandon_scope(int v) {
    static std::mutex mu;
    static std::vector<std::unique_ptr<G1Projective>> keep;  // output buffers of abandoned calls stay valid for the life of the process
    G1Projective* r = new G1Projective();
    {
        std::lock_guard<std::mutex> lk(mu);
        keep.emplace_back(r);
    }
    std::thread t([=] {
        RustError e = snarkvm_hip_scope_begin_ex(g_dpool.data(), SNARKVM_HIP_SCOPE_ASYNC_MSM);
        if (e.code == 0) {
            e = snarkvm_hip_msm_registered(r, g_h, (size_t)v, 20000, g_dpool.at(32 * (size_t)v), 1, 0);
            std::free(e.message);
            e = snarkvm_hip_ntt_device(g_dpool.at(32 * 70000), 12, NN, Forward, Standard);  // (a private corner of the pool nobody reads: beyond the variants' ranges)
        }
        std::free(e.message);
    });
    t.join();
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int T = argc > 2 ? atoi(argv[2]) : 16;
    uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 0) : 1;
    const int ndev = argc > 4 ? atoi(argv[4]) : 1;
    try {
        if (ndev > 1) {
            std::vector<int32_t> ids((size_t)ndev, 0);
            const int vis = snarkvm_hip_device_count();
            for (int i = 0; i < ndev; i++) ids[(size_t)i] = i < vis ? i : 0;
            check(snarkvm_hip_set_devices(ids.data(), ids.size()));
        }
        // ---- operands
        uint64_t s = seed * 0x1234567ull + 99;
        g_pool.resize(NS);
        for (auto& f : g_pool) {
            for (auto& l : f.l) l = splitmix(s);
            f.l[3] &= ((uint64_t)1 << 58) - 1;  // < 2^250 < r: a canonical scalar and a valid Montgomery image alike
        }
        g_dpool = DeviceBuffer(NS * 32, 0);
        g_dpool.upload(g_pool.data(), NS * 32);
        g_bases.resize(NB);
        {
            DeviceBuffer d(NB * sizeof(G1Affine), 0);
            check(snarkvm_hip_g1_generate_bases_device(d.data(), 1, NB));
            d.download(g_bases.data(), NB * sizeof(G1Affine));
            check(snarkvm_hip_register_bases_windowed(&g_h, d.data(), NB, sizeof(G1Affine), 1, 17, 15));
        }
        // ---- every (operation, variant) once, alone
        std::vector<std::vector<Bytes>> expect(NOPS, std::vector<Bytes>(V));
        {
            ThreadCtx tc;
            for (int op = 0; op < NOPS; op++)
                for (int v = 0; v < V; v++) expect[(size_t)op][(size_t)v] = run_op(op, v, tc);
            for (int op = 0; op < NOPS; op++)  // and a second time: the alone pass must agree with itself
                for (int v = 0; v < V; v++)
                    if (run_op(op, v, tc) != expect[(size_t)op][(size_t)v]) {
                        printf("{\"ok\": false, \"error\": \"operation %s variant %d is not deterministic when issued alone\"}\n", OP_NAME[op], v);
                        return 2;
                    }
        }
        // ---- the soak
        std::atomic<uint64_t> calls[NOPS + 2], mismatches{0}, failures{0}, bad_not_refused{0};
        for (auto& c : calls) c = 0;
        std::vector<std::atomic<uint64_t>> progress((size_t)T);
        for (auto& p : progress) p = 0;
        std::atomic<bool> stop{false}, stalled{false};
        std::mutex msg_mu;
        std::string first_msg;
        auto note = [&](const std::string& m) {
            std::lock_guard<std::mutex> lk(msg_mu);
            if (first_msg.empty()) first_msg = m;
        };
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                uint64_t rs = seed * 1000003ull + (uint64_t)t;
                try {
                    ThreadCtx tc;
                    for (uint64_t it = 0; !stop.load(std::memory_order_relaxed); it++) {
                        const uint64_t r = splitmix(rs);
                        const int v = (int)((r >> 8) % V);
                        const int kind = (int)(r % 64);
                        if (kind == 0) {  // one in 64: a bad request
                            if (!bad_request((unsigned)(r >> 20), tc)) bad_not_refused++;
                            calls[NOPS]++;
                        } else if (kind == 1) {  // one in 64: a scope abandoned by a thread that exits
                            abandon_scope(v);
                            calls[NOPS + 1]++;
                        } else {
                            const int op = (int)((r >> 32) % NOPS);
                            try {
                                if (run_op(op, v, tc) != expect[(size_t)op][(size_t)v]) {
                                    mismatches++;
                                    note(std::string("mismatch: ") + OP_NAME[op] + " variant " + std::to_string(v) + " on thread " + std::to_string(t));
                                }
                            } catch (const snarkvm_hip::Error& e) {
                                failures++;
                                note(std::string("error in ") + OP_NAME[op] + " variant " + std::to_string(v) + ": " + e.what());
                                RustError x = snarkvm_hip_scope_end();  // a failed call must not leave this thread's scope open
                                std::free(x.message);
                            }
                            calls[op]++;
                        }
                        progress[(size_t)t]++;
                    }
                } catch (const std::exception& e) {
                    failures++;
                    note(std::string("thread ") + std::to_string(t) + " died: " + e.what());
                }
            });
        // watchdog: a thread that makes no progress for 120 s is a hang
        std::vector<uint64_t> last((size_t)T, 0);
        std::vector<double> since((size_t)T, 0.0);
        for (;;) {
            std::this_thread::sleep_for(std::chrono::milliseconds(500));
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (int t = 0; t < T; t++) {
                const uint64_t p = progress[(size_t)t].load();
                if (p != last[(size_t)t]) last[(size_t)t] = p, since[(size_t)t] = el;
                if (el - since[(size_t)t] > 120.0) stalled = true;
            }
            if (stalled) {
                printf("{\"ok\": false, \"error\": \"a thread made no progress for 120 s (hang)\", \"elapsed_s\": %.1f}\n", el);
                fflush(stdout);
                _exit(3);
            }
            if (el >= seconds) break;
        }
        stop = true;
        for (auto& t : th) t.join();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t co[4] = {0, 0, 0, 0};
        snarkvm_hip_coalescer_stats(co, 0);
        uint64_t total = 0;
        std::string per;
        for (int op = 0; op < NOPS; op++) {
            total += calls[op].load();
            per += std::string(op ? ", " : "") + "\"" + OP_NAME[op] + "\": " + std::to_string(calls[op].load());
        }
        const bool ok = mismatches == 0 && failures == 0 && bad_not_refused == 0;
        printf("{\"ok\": %s, \"seconds\": %.1f, \"threads\": %d, \"seed\": %llu, \"logical_devices\": %d, \"checked_calls\": %llu, \"calls\": {%s}, \"bad_requests_refused\": %llu, "
               "\"bad_requests_not_refused\": %llu, \"abandoned_scopes\": %llu, \"mismatches\": %llu, \"unexpected_errors\": %llu, "
               "\"coalescer\": {\"batches\": %llu, \"instances\": %llu, \"largest_batch\": %llu}, \"first_problem\": \"%s\"}\n",
               ok ? "true" : "false", el, T, (unsigned long long)seed, snarkvm_hip_num_devices(), (unsigned long long)total, per.c_str(),
               (unsigned long long)(calls[NOPS].load() - bad_not_refused.load()), (unsigned long long)bad_not_refused.load(), (unsigned long long)calls[NOPS + 1].load(),
               (unsigned long long)mismatches.load(), (unsigned long long)failures.load(), (unsigned long long)co[0], (unsigned long long)co[1], (unsigned long long)co[2],
               first_msg.c_str());
        snarkvm_hip_free_bases(g_h);
        g_dpool = DeviceBuffer();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        printf("{\"ok\": false, \"error\": \"set-up failed: %s\"}\n", e.what());
        return 2;
    }
}
