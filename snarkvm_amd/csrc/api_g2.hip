// api_g2.hip - the G2 (Fq2) entry points of the C ABI: the MSM engine, table precomputation and point encoding instantiated
// over fq2_t.  A separate translation unit only because these instantiations are half of the compile time: build.py compiles
// the units in parallel.
#include "runtime.hip.h"

struct snarkvm_hip_bases_g2 : bases_handle_t<fq2_t> {};

extern "C" {

RustError snarkvm_hip_msm_g2(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
    API_TRY
#ifdef SV_NO_G2  // development builds only (python -m snarkvm_amd.build --fast): skips the Fq2 kernel instantiations
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!out) throw hip_failure{hipErrorInvalidValue, "msm_g2: null output", __LINE__};
    g_rt.configure();
    if (npoints == 0) {
        write_infinity<fq2_t>(out);
    } else {
        if (!points || !scalars) throw hip_failure{hipErrorInvalidValue, "msm_g2: null argument", __LINE__};
        msm_host_chunked<fq2_t>(out, points, npoints, scalars, ffi_affine_sz);
    }
#endif
    API_CATCH
}

// ---- registered G2 bases (extension): same engine over fq2_t, precomputed tables remove most of the Horner chain of a
// one-shot G2 MSM
RustError snarkvm_hip_register_bases_g2(snarkvm_hip_bases_g2_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int tables,
                                        int window_bits) {
    API_TRY
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!handle || (npoints && !points)) throw hip_failure{hipErrorInvalidValue, "register_bases_g2: null argument", __LINE__};
    if (ffi_affine_sz < 200 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "register_bases_g2: bad stride", __LINE__};
    check_tables(tables, window_bits, "register_bases_g2");
    const int nd = g_rt.ndev();
    std::unique_ptr<snarkvm_hip_bases_g2> h(new snarkvm_hip_bases_g2());
    h->n = npoints;
    h->tables = tables;
    h->table_bits = window_bits ? window_bits : 256 / tables;
    h->d.assign(nd, nullptr);
    if (npoints) {
        try {
            std::vector<int> all;
            for (int d = 0; d < nd; d++) all.push_back(d);
            for_each_device(all, [&](int dev) {
                lane_guard lg(dev);
                lane_t& c = lg.c();
                HIP_TRY(hipMalloc((void**)&h->d[dev], (size_t)tables * npoints * sizeof(aff_mem_t<fq2_t>)));
                c.bases_tmp.ensure(npoints * ffi_affine_sz);
                HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, c.stream));
                convert_bases<fq2_t>(c, c.bases_tmp.as<uint8_t>(), ffi_affine_sz, npoints, h->d[dev]);
                precompute_tables_run<fq2_t>(c, h->d[dev], npoints, tables, h->table_bits);
                bases_to_lazy_form(c, h->d[dev], (size_t)tables * npoints);  // last: the tables are derived from one another in the exact form
                HIP_TRY(hipStreamSynchronize(c.stream));
            });
        } catch (...) {
            h->free_all();
            throw;
        }
    }
    *handle = h.release();
#endif
    API_CATCH
}
void snarkvm_hip_free_bases_g2(snarkvm_hip_bases_g2_t* h) {
    if (!h) return;
    h->free_all();
    delete h;
}
#ifndef SV_NO_G2
// a proof-sized G2 MSM of one caller: concurrent callers over the same handle are fused (runtime.hip.h::msm_coalesced)
static RustError msm_g2_single_coalesced(void* out, const snarkvm_hip_bases_g2_t* h, size_t offset, size_t npoints, const void* scalars, int scalars_on_device,
                                         int window_bits) {
    API_TRY
    if (scalars_on_device && g_rt.device_of(scalars) < 0)
        throw hip_failure{hipErrorInvalidValue, "device pointer does not belong to a device in use (snarkvm_hip_set_devices)", __LINE__};
    msm_ticket_t t;
    t.req.off0 = offset, t.req.n0 = npoints, t.req.scalars = scalars, t.req.out = out;
    t.on_device = scalars_on_device ? 1 : 0;
    t.window_bits = window_bits;
    msm_coalesced<fq2_t>(*h, &t, 1);
    API_CATCH
}
// the same call from a thread inside an SNARKVM_HIP_SCOPE_ASYNC_MSM scope: only enqueued (runtime.hip.h::msm_scope_enqueue)
static RustError msm_g2_scope_enqueue(void* out, const snarkvm_hip_bases_g2_t* h, size_t offset, size_t npoints, const void* scalars, int window_bits, bool* queued) {
    API_TRY
    msm_req_t one;
    one.off0 = offset, one.n0 = npoints, one.scalars = scalars, one.out = out;
    *queued = msm_scope_enqueue<fq2_t>(*h, &one, 1, 1, 0, window_bits);
    API_CATCH
}
#endif
RustError snarkvm_hip_msm_g2_registered(void* out, const snarkvm_hip_bases_g2_t* h, size_t offset, size_t npoints, const void* scalars,
                                        int scalars_on_device, int window_bits) {
#ifndef SV_NO_G2
    if (h && out && scalars && offset + npoints <= h->n && (!window_bits || (window_bits >= 2 && window_bits <= MSM_C_MAX)) && g_rt.configured) {
        if (tl_scope().lane && (tl_scope().flags & SNARKVM_HIP_SCOPE_ASYNC_MSM) && scalars_on_device && npoints) {
            bool queued = false;
            const RustError e = msm_g2_scope_enqueue(out, h, offset, npoints, scalars, window_bits, &queued);
            if (e.code || queued) return e;
        }
        if (msm_coalescible(*h, npoints, window_bits)) return msm_g2_single_coalesced(out, h, offset, npoints, scalars, scalars_on_device, window_bits);
    }
#endif
    API_BEGIN_DEV(device_for(scalars, (scalars_on_device && npoints) ? 1 : 0))
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!h || offset + npoints > h->n) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: range exceeds the registered bases", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: window_bits must be 0 or 2..23", __LINE__};
    if (!out || (npoints && !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: null argument", __LINE__};
    const uint4* d_sc = (const uint4*)scalars;
    if (!scalars_on_device && npoints) {
        c.scalars_tmp.ensure(npoints * 32);
        c.phase_begin("msm_h2d");
        HIP_TRY(hipMemcpyAsync(c.scalars_tmp.p, scalars, npoints * 32, hipMemcpyHostToDevice, c.stream));
        c.phase_end();
        d_sc = c.scalars_tmp.as<uint4>();
    }
    msm_run_sync<fq2_t>(c, h->d[c.dev->logical] + offset, d_sc, npoints, out, window_bits, nullptr, ~(size_t)0, 0, h->tables, h->n, h->table_bits);
#endif
    API_END
}
// A batch of independent G2 MSMs over one registered vector (BASELINE configs[4]: one per proof), fanned out like the G1 batch.
RustError snarkvm_hip_msm_g2_registered_batch(void* outs, const snarkvm_hip_bases_g2_t* h, size_t count, const size_t* offsets, const size_t* npoints,
                                              const void* const* scalars, int scalars_on_device, int window_bits) {
    API_TRY
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!h) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered_batch: null handle", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered_batch: window_bits must be 0 or 2..23", __LINE__};
    if (count && (!outs || !offsets || !npoints || !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered_batch: null argument", __LINE__};
    std::vector<msm_req_t> req = msm_requests<fq2_t>(outs, count, offsets, npoints, nullptr, nullptr, scalars);
    msm_batch_dispatch<fq2_t>(*h, req, scalars_on_device, 0, window_bits);
#endif
    API_CATCH
}

static RustError g2_deserialize_impl(void* out_affine, const void* bytes, size_t n, int compressed, int validate) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (n) {
        if (!out_affine || !bytes) throw hip_failure{hipErrorInvalidValue, "g2_deserialize: null argument", __LINE__};
        const size_t psz = compressed ? 96 : 192;
        c.bases_tmp.ensure(n * psz);
        c.poly[0].ensure(n * 200);
        c.serde_status.ensure(4);
        HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, bytes, n * psz, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(hipMemsetAsync(c.serde_status.p, 0, 4, c.stream));
        hipLaunchKernelGGL(g2_deserialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.bases_tmp.as<uint8_t>(), n, compressed, validate,
                           c.poly[0].as<uint8_t>(), c.serde_status.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        uint32_t st = 0;
        HIP_TRY(hipMemcpyAsync(&st, c.serde_status.p, 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(out_affine, c.poly[0].p, n * 200, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        serde_throw_on_status(st, "g2_deserialize");
    }
#endif
    API_END
}
static RustError g2_serialize_impl(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz, int compressed) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (n) {
        if (!out_bytes || !affine) throw hip_failure{hipErrorInvalidValue, "g2_serialize: null argument", __LINE__};
        if (ffi_affine_sz < 200 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "g2_serialize: bad stride", __LINE__};
        c.bases_tmp.ensure(n * ffi_affine_sz);
        const size_t psz = compressed ? 96 : 192;
        c.poly[0].ensure(n * psz);
        HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, affine, n * ffi_affine_sz, hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(g2_serialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.bases_tmp.as<uint8_t>(), ffi_affine_sz, n,
                           compressed, c.poly[0].as<uint8_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_bytes, c.poly[0].p, n * psz, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
#endif
    API_END
}

RustError snarkvm_hip_g2_deserialize(void* out_affine, const void* bytes, size_t n, int validate) { return g2_deserialize_impl(out_affine, bytes, n, 0, validate); }
RustError snarkvm_hip_g2_deserialize_compressed(void* out_affine, const void* bytes, size_t n, int validate) {
    return g2_deserialize_impl(out_affine, bytes, n, 1, validate);
}
RustError snarkvm_hip_g2_serialize(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz) { return g2_serialize_impl(out_bytes, affine, n, ffi_affine_sz, 0); }
RustError snarkvm_hip_g2_serialize_compressed(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz) {
    return g2_serialize_impl(out_bytes, affine, n, ffi_affine_sz, 1);
}

// ---- test hook: the lazy Fq2 arithmetic of the G2 accumulate kernel (ffl2.hip.h) against the exact arithmetic, on the host ----
// `points`: npoints (>= 2) Rust G2Affine records (200-byte stride) on the curve.  A chain of `iters` mixed additions of +- points[k]
// (the same point again -> doubling, its negative -> cancellation, restarts from infinity), every coordinate component of the
// accumulator compared with xyzz_t<fq2_t>::add_affine after every step; then products and squares of tight operands taken from the
// chain against fq2_t's.  0 = identical; > 0: first differing step; < 0: field case.
int snarkvm_hip_selftest_fq2_lazy(const void* points, size_t npoints, uint64_t seed, int iters) {
#ifdef SV_NO_G2
    (void)points, (void)npoints, (void)seed, (void)iters;
    return -1;
#else
    if (!points || npoints < 2) return -2;
    std::vector<aff_t<fq2_t>> pool;
    for (size_t i = 0; i < npoints; i++) {
        const uint32_t* src = (const uint32_t*)((const uint8_t*)points + 200 * i);
        pool.push_back({fq2_t::from_raw_words(src), fq2_t::from_raw_words(src + 24)});
    }
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto lazy_of = [](const fq2_t& v) { return fq2l_t{fql_canonical_from_exact(v.c0), fql_canonical_from_exact(v.c1)}; };
    auto same = [](const fq2l_t& l, const fq2_t& e) { return l.c0.to_exact() == e.c0 && l.c1.to_exact() == e.c1; };
    xyzz_t<fq2_t> exact = xyzz_t<fq2_t>::inf();
    xyzz_lazy2_t lazy = xyzz_lazy2_t::infinity();
    int prev = -1;
    bool prev_neg = false;
    for (int it = 0; it < iters; it++) {
        const uint64_t r = next();
        int k = (int)(r % npoints);
        bool neg = ((r >> 8) & 1) != 0;
        const int mode = (int)((r >> 16) % 16);
        if (mode == 0 && prev >= 0) k = prev, neg = prev_neg;
        if (mode == 1 && prev >= 0) k = prev, neg = !prev_neg;
        if (mode == 2) {
            exact = xyzz_t<fq2_t>::inf();
            lazy = xyzz_lazy2_t::infinity();
        }
        prev = k;
        prev_neg = neg;
        const aff_t<fq2_t> p = pool[k];
        exact.add_affine(p, neg);
        const fq2l_t px = lazy_of(p.x), py = lazy_of(p.y);
        if (!lazy.madd(px, py, neg)) {
            xyzz_t<fq2_t> e = lazy.to_exact();
            e.add_affine(p, neg);
            lazy = xyzz_lazy2_t::from_exact(e);
        }
        if (lazy.inf != exact.is_inf()) return it + 1;
        if (!lazy.inf && !(same(lazy.x, exact.x) && same(lazy.y, exact.y) && same(lazy.zz, exact.zz) && same(lazy.zzz, exact.zzz))) return it + 1;
        if (!lazy.inf && (it & 7) == 0) {  // products / squares of the accumulator's own (tight, mixed-sign) components
            if (!same(fq2l::mul(lazy.zz, lazy.zzz), exact.zz * exact.zzz)) return -(4 * it + 1);
            if (!same(fq2l::sqr(lazy.zzz), exact.zzz.sqr())) return -(4 * it + 2);
            if (!same(fq2l::mul(lazy.x, lazy.zz), exact.x * exact.zz)) return -(4 * it + 3);
            // the raw memory image and the dense conversion
            alignas(16) g2_lazy_partial_t raw;
            lazy.store_raw(&raw);
            const xyzz_t<fq2_t> back = xyzz_lazy2_t::exact_from_raw(&raw);
            if (!(back.x == exact.x && back.y == exact.y && back.zz == exact.zz && back.zzz == exact.zzz)) return -(4 * it + 4);
        }
    }
    return 0;
#endif
}

// ---- test hook: the lane-pair Fq2 arithmetic of the G2 accumulate kernel (ffl2p.hip.h) against the exact arithmetic, on the host ----
// The SAME source the kernel runs (fq2p::xyzz_pair_t) instantiated with the two-lane host exchange policy: a chain of `iters` mixed
// additions of +- points[k] - the same point again (doubling, resolved inside the pair arithmetic), its negative (cancellation),
// restarts from infinity - with every coordinate component compared with xyzz_t<fq2_t>::add_affine after every step, the raw partial-sum
// image and its conversion on the way.  0 = identical; > 0: first differing step; < 0: a conversion case.
int snarkvm_hip_selftest_fq2_pair(const void* points, size_t npoints, uint64_t seed, int iters) {
#ifdef SV_NO_G2
    (void)points, (void)npoints, (void)seed, (void)iters;
    return -1;
#else
    if (!points || npoints < 2) return -2;
    std::vector<aff_t<fq2_t>> pool;
    for (size_t i = 0; i < npoints; i++) {
        const uint32_t* src = (const uint32_t*)((const uint8_t*)points + 200 * i);
        pool.push_back({fq2_t::from_raw_words(src), fq2_t::from_raw_words(src + 24)});
    }
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto same = [](const fql_t (&l)[2], const fq2_t& e) { return l[0].to_exact() == e.c0 && l[1].to_exact() == e.c1; };
    xyzz_t<fq2_t> exact = xyzz_t<fq2_t>::inf();
    fq2p::xyzz_pair_t<fq2p::xp_host> pr;
    pr.inf = true;
    int prev = -1;
    bool prev_neg = false;
    for (int it = 0; it < iters; it++) {
        const uint64_t r = next();
        int k = (int)(r % npoints);
        bool neg = ((r >> 8) & 1) != 0;
        const int mode = (int)((r >> 16) % 16);
        if (mode == 0 && prev >= 0) k = prev, neg = prev_neg;
        if (mode == 1 && prev >= 0) k = prev, neg = !prev_neg;
        if (mode == 2) {
            exact = xyzz_t<fq2_t>::inf();
            pr.inf = true;
        }
        prev = k;
        prev_neg = neg;
        const aff_t<fq2_t> p = pool[k];
        exact.add_affine(p, neg);
        // the base slot as the kernel reads it: component c of x and y
        alignas(128) aff_mem_t<fq2_t> slot;
        const fq_t c406 = fq_t::from_table(FqLConv::C406);
        g2_lazy_slot_t::store(&slot, {p.x.c0 * c406, p.x.c1 * c406}, {p.y.c0 * c406, p.y.c1 * c406}, false);
        fql_t px[2], py[2];
        for (int c = 0; c < 2; c++) ((const g2_lazy_slot_t*)&slot)->component(c, px[c], py[c]);
        pr.madd(px, py, neg);
        if (pr.inf != exact.is_inf()) return it + 1;
        if (!pr.inf && !(same(pr.x, exact.x) && same(pr.y, exact.y) && same(pr.zz, exact.zz) && same(pr.zzz, exact.zzz))) return it + 1;
        if (!pr.inf && (it & 7) == 0) {  // the raw partial-sum image through the conversion the device kernel applies per component
            const fql_t* cs[4] = {pr.x, pr.y, pr.zz, pr.zzz};
            const fq2_t* es[4] = {&exact.x, &exact.y, &exact.zz, &exact.zzz};
            for (int c4 = 0; c4 < 4; c4++)
                for (int c = 0; c < 2; c++) {
                    fql_t back;
                    for (int j = 0; j < 13; j++) back.v[j] = cs[c4][c].v[j];
                    if (!(back.to_exact() == (c ? es[c4]->c1 : es[c4]->c0))) return -(it + 1);
                }
        }
    }
    return 0;
#endif
}

// The G2 fold and bit-plane kernels launched `iters` times over ONE fixed set of per-bucket partial-sum lists (2^(m + hb) buckets of one window; 0, 1 or 2
// entries per bucket - the sparse shape of a small MSM over window tables - built from +- points[k]): every launch must leave the same group elements.
// A difference is a race or a lost register, whatever the points are.  threads: per fold workgroup (64, 128, 256);
// plane_threads likewise; hex / quads: the kernels' switches.  report[0] = fold launches that differ from the first, report[1] = differing fold slots in
// total, report[2] = bit-plane launches that differ from the first, report[3] = differing planes in total, report[4 .. 7] = the first differing fold slots, report[8], [9] = fold slots / planes of the first launch the fast
// kernels flagged for the fix kernels (equal x coordinates met on the way).
RustError snarkvm_hip_devtest_g2_tail_repeat(const void* points, size_t npoints, int m, int hb, int threads, int plane_threads, int hex, int quads, int iters,
                                             uint32_t* report) {
    API_BEGIN
#ifdef SV_NO_G2
    (void)points, (void)npoints, (void)m, (void)hb, (void)threads, (void)plane_threads, (void)hex, (void)quads, (void)iters, (void)report;
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!points || npoints < 2 || !report || m < 1 || hb < 1 || hb > m || m + hb > 16 || iters < 2 || (threads != 64 && threads != 128 && threads != 256) ||
        (plane_threads != 64 && plane_threads != 128 && plane_threads != 256))
        throw hip_failure{hipErrorInvalidValue, "devtest_g2_tail_repeat: bad arguments", __LINE__};
    typedef xyzz_mem_t<fq2_t> mem_t;
    const uint32_t nb = 1u << (m + hb);
    std::vector<uint32_t> cnt(nb), start(nb);
    uint32_t E = 0;
    for (uint32_t k = 0; k < nb; k++) {
        const uint32_t h = (k * 2654435761u) >> 28;
        cnt[k] = h == 0 ? 2 : h == 1 ? 0 : 1;
        start[k] = E;
        E += cnt[k];
    }
    std::vector<mem_t> sums(E);
    for (uint32_t q = 0; q < E; q++) {
        const uint32_t* src = (const uint32_t*)((const uint8_t*)points + 200 * (size_t)((q * 7u + 3u) % npoints));
        xyzz_t<fq2_t> a = xyzz_t<fq2_t>::inf();
        a.add_affine({fq2_t::from_raw_words(src), fq2_t::from_raw_words(src + 24)}, ((q >> 3) & 1) != 0);
        store_xyzz<fq2_t>(&sums[q], a);
    }
    const size_t nslots = (size_t)1 << (m + 1), nbits = (size_t)m + 1, nplanes = 2 * nbits;
    c.part_a.ensure(E * sizeof(mem_t));
    c.cnt_a.ensure(nb * 4);
    c.start_a.ensure(nb * 4);
    c.fold_sums.ensure(2 * nslots * sizeof(mem_t));  // [0]: the first launch's output (the bit planes read it), [1]: the launch under test
    c.planes.ensure(nplanes * sizeof(mem_t));
    c.tail_flags.ensure((nslots + nplanes) * 4);
    std::vector<uint32_t> flag_copy(nslots + nplanes, 0);
    HIP_TRY(hipMemcpyAsync(c.part_a.p, sums.data(), E * sizeof(mem_t), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(c.cnt_a.p, cnt.data(), nb * 4, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(c.start_a.p, start.data(), nb * 4, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemsetAsync(c.fold_sums.p, 0, 2 * nslots * sizeof(mem_t), c.stream));
    std::vector<uint8_t> first(nslots * sizeof(mem_t)), got(nslots * sizeof(mem_t)), p_first(nplanes * sizeof(mem_t)), p_got(nplanes * sizeof(mem_t));
    for (int i = 0; i < 10; i++) report[i] = 0;
    int nfirst = 0;
    auto same_point = [](const xyzz_t<fq2_t>& a, const xyzz_t<fq2_t>& b) {  // as group elements (the coordinates of a point at infinity are arbitrary)
        if (a.is_inf() || b.is_inf()) return a.is_inf() == b.is_inf();
        return a.x * b.zz == b.x * a.zz && a.y * b.zzz == b.y * a.zzz;
    };
    for (int it = 0; it < iters; it++) {
        mem_t* out = c.fold_sums.as<mem_t>() + (it ? nslots : 0);
        uint32_t* fold_flags = c.tail_flags.as<uint32_t>();
        uint32_t* plane_flags = fold_flags + nslots;
        const dim3 fold_grid((1u << m) + (1u << hb), 1u), plane_grid((unsigned)nbits, 2u);
        hipLaunchKernelGGL((msm_fold_kernel<fq2_t, true>), fold_grid, dim3((unsigned)threads), 0, c.stream, (const mem_t*)c.part_a.as<mem_t>(),
                           (const uint32_t*)c.start_a.as<uint32_t>(), (const uint32_t*)c.cnt_a.as<uint32_t>(), out, m, hb, hex, quads & 2 ? 1 : 0, fold_flags);
        hipLaunchKernelGGL((msm_fold_fix_kernel<fq2_t>), fold_grid, dim3(64), 0, c.stream, (const mem_t*)c.part_a.as<mem_t>(), (const uint32_t*)c.start_a.as<uint32_t>(),
                           (const uint32_t*)c.cnt_a.as<uint32_t>(), out, m, hb, (const uint32_t*)fold_flags);
        hipLaunchKernelGGL((msm_bitplane_kernel<fq2_t, true>), plane_grid, dim3((unsigned)plane_threads), 0, c.stream, (const mem_t*)c.fold_sums.as<mem_t>(),
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, c.planes.as<mem_t>(), nb, m, hb, hex, quads & 1, plane_flags);
        hipLaunchKernelGGL((msm_bitplane_fix_kernel<fq2_t, true>), plane_grid, dim3(64), 0, c.stream, (const mem_t*)c.fold_sums.as<mem_t>(), (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, c.planes.as<mem_t>(), nb, m, hb, (const uint32_t*)plane_flags);
        if (!it) HIP_TRY(hipMemcpyAsync(flag_copy.data(), fold_flags, (nslots + nplanes) * 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(it ? got.data() : first.data(), out, nslots * sizeof(mem_t), hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(it ? p_got.data() : p_first.data(), c.planes.p, nplanes * sizeof(mem_t), hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        if (!it) continue;
        uint32_t bad = 0;
        for (size_t sl = 0; sl + 1 < nslots; sl++)  // (the last slot is never written)
            if (!same_point(load_xyzz<fq2_t>((const mem_t*)&first[sl * sizeof(mem_t)]), load_xyzz<fq2_t>((const mem_t*)&got[sl * sizeof(mem_t)]))) {
                bad++;
                if (nfirst < 4) report[4 + nfirst++] = (uint32_t)sl;
            }
        report[0] += bad ? 1 : 0;
        report[1] += bad;
        bad = 0;
        for (size_t pl = 0; pl < nplanes; pl++)
            if (!same_point(load_xyzz<fq2_t>((const mem_t*)&p_first[pl * sizeof(mem_t)]), load_xyzz<fq2_t>((const mem_t*)&p_got[pl * sizeof(mem_t)]))) bad++;
        report[2] += bad ? 1 : 0;
        report[3] += bad;
    }
    report[8] = report[9] = 0;  // outputs of the first launch the fast kernels flagged (fold slots, planes): the fix kernels' share of the work
    for (size_t sl = 0; sl + 1 < nslots; sl++) report[8] += flag_copy[sl] ? 1 : 0;  // (the unwritten slots' flags are whatever the buffer held)
    for (size_t pl = 0; pl < nplanes; pl++) report[9] += flag_copy[nslots + pl] ? 1 : 0;
#endif
    API_END
}

// The sixteen-lane cooperative Fq2 addition of the G2 tail trees (csrc/hex2.hip.h) on sixteen SIMULATED lanes - the same source as the kernel's, every lane
// with its own copy of the state, the pair exchange and the gathers indexing the other lanes' copies - against xyzz_t<fq2_t>::add: `iters` additions of partial
// sums built from +- points[k] (general ZZ / ZZZ), with the cases a tree meets: an operand at infinity (either, both), P + P (every lane must ask for the
// plain law), P + (-P).  0 = every lane of every addition ends with exactly the exact sum; > 0: first differing iteration; < 0: bad arguments.
int snarkvm_hip_selftest_g2_hex(const void* points, size_t npoints, uint64_t seed, int iters) {
#ifdef SV_NO_G2
    (void)points, (void)npoints, (void)seed, (void)iters;
    return -1;
#else
    if (!points || npoints < 2) return -2;
    std::vector<aff_t<fq2_t>> pool;
    for (size_t i = 0; i < npoints; i++) {
        const uint32_t* src = (const uint32_t*)((const uint8_t*)points + 200 * i);
        pool.push_back({fq2_t::from_raw_words(src), fq2_t::from_raw_words(src + 24)});
    }
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto partial = [&](int terms) {  // a partial sum with general ZZ, ZZZ
        xyzz_t<fq2_t> a = xyzz_t<fq2_t>::inf();
        for (int t = 0; t < terms; t++) {
            const uint64_t r = next();
            a.add_affine(pool[r % npoints], ((r >> 40) & 1) != 0);
        }
        return a;
    };
    for (int it = 0; it < iters; it++) {
        const int mode = (int)(next() % 12);
        xyzz_t<fq2_t> a = partial(1 + (int)(next() % 3)), b = partial(1 + (int)(next() % 3));
        if (mode == 0) b = a;                                  // doubling: the equal-x fallback on every lane
        if (mode == 1) b = a, b.y = b.y.neg();                 // cancellation: equal x as well
        if (mode == 2) a = xyzz_t<fq2_t>::inf();
        if (mode == 3) b = xyzz_t<fq2_t>::inf();
        if (mode == 4) a = xyzz_t<fq2_t>::inf(), b = xyzz_t<fq2_t>::inf();
        xyzz_t<fq2_t> want = a;
        want.add(b);
        if (hex_add_host_check(a, b, want)) return it + 1;
        // the flagged plain addition of the tail kernels (xyzz_t::add_flag): equal x coordinates of two finite operands raise the flag and leave the accumulator
        // alone - P + P and P - P alike -, everything else is the exact sum and leaves the flag down
        xyzz_t<fq2_t> f = a;
        bool flagged = false;
        f.add_flag(b, flagged);
        const bool same_x = !a.is_inf() && !b.is_inf() && a.x * b.zz == b.x * a.zz;
        if (flagged != same_x) return 100000 + it + 1;
        const xyzz_t<fq2_t>& expect = flagged ? a : want;
        if (!(f.x == expect.x && f.y == expect.y && f.zz == expect.zz && f.zzz == expect.zzz)) return 200000 + it + 1;
    }
    return 0;
#endif
}

}  // extern "C"
