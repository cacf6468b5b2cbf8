// api_fr.hip - the Fr-side entry points of the C ABI: NTT / polymul (the reference's snarkvm_ntt, snarkvm_polymul), the
// prover-round polynomial kernels and the setup-time group operations that are built on them.  A separate translation unit so
// that build.py can compile it next to api.hip (G1 MSM) and api_g2.hip.
#define SV_TU_NTT
#include "runtime.hip.h"

extern "C" {

// ---- NTT -------------------------------------------------------------------------------------
static void check_ntt_args(uint32_t lg, int order, int dir, int type) {
    if (lg > (uint32_t)NTT_LG_MAX) throw hip_failure{hipErrorMemoryAllocation, "ntt: lg_domain_size > 26 is not supported by this backend", __LINE__};
    if (order < 0 || order > 3 || dir < 0 || dir > 1 || type < 0 || type > 1) throw hip_failure{hipErrorInvalidValue, "ntt: bad enum value", __LINE__};
}
RustError snarkvm_ntt(void* inout, uint32_t lg, enum NTTInputOutputOrder order, enum NTTDirection dir, enum NTTType type) {
    API_BEGIN
    check_ntt_args(lg, (int)order, (int)dir, (int)type);
    if (!inout) throw hip_failure{hipErrorInvalidValue, "ntt: null buffer", __LINE__};
    const size_t bytes = sizeof(fr_mem_t) << lg;
    c.ntt_data.ensure(bytes);
    c.ntt_scratch.ensure(bytes);
    c.phase_begin("ntt_h2d");
    HIP_TRY(hipMemcpyAsync(c.ntt_data.p, inout, bytes, hipMemcpyHostToDevice, c.stream));
    c.phase_end();
    c.phase_begin("ntt_kernels");
    ntt_run(c.ntt_ctx(), c.ntt_data.as<fr_mem_t>(), c.ntt_scratch.as<fr_mem_t>(), (int)lg, (int)order, (int)dir, (int)type);
    c.phase_end();
    HIP_TRY(hipGetLastError());
    c.phase_begin("ntt_d2h");
    HIP_TRY(hipMemcpyAsync(inout, c.ntt_data.p, bytes, hipMemcpyDeviceToHost, c.stream));
    c.phase_end();
    HIP_TRY(hipStreamSynchronize(c.stream));
    API_END
}
RustError snarkvm_hip_ntt_device(void* d_inout, uint32_t lg, int order, int dir, int type) {
    API_BEGIN_DEV(device_for(d_inout, 1))
    check_ntt_args(lg, order, dir, type);
    c.ntt_scratch.ensure(sizeof(fr_mem_t) << lg);
    c.phase_begin("ntt_kernels");
    ntt_run(c.ntt_ctx(), (fr_mem_t*)d_inout, c.ntt_scratch.as<fr_mem_t>(), (int)lg, order, dir, type);
    c.phase_end();
    HIP_TRY(hipGetLastError());
    c.sync_or_defer();
    API_END
}
// `count` independent in-place transforms of 2^lg elements each over device vectors (the iNTTs of a prover round: second.rs:104-113
// transforms z_a, z_b, z_c at |R|; fourth.rs:174-231 nine vectors at |K|): every transform is enqueued on ONE lane's stream and the
// call synchronises once - a synchronous call per vector pays the ~0.1-0.2 ms of stream-synchronisation latency of this stack
// per transform (2.36 ms around the 2.16 ms of kernels at 2^24, 52 us around ~15 us at 2^16).  The same vector may appear more
// than once (stream order = list order).  directions / types: per vector, or NULL for all forward / all standard.
RustError snarkvm_hip_ntt_device_batch(void* const* d_inouts, size_t count, uint32_t lg, int order, const int* dirs, const int* types) {
    if (count == 0) return ok();
    if (!d_inouts || !d_inouts[0]) return fail((int)hipErrorInvalidValue, "snarkvm_hip: ntt_device_batch: null argument");
    API_BEGIN_DEV(device_for(d_inouts[0], 1))
    for (size_t k = 0; k < count; k++) {
        check_ntt_args(lg, order, dirs ? dirs[k] : 0, types ? types[k] : 0);
        if (!d_inouts[k]) throw hip_failure{hipErrorInvalidValue, "ntt_device_batch: null vector", __LINE__};
        // every vector must live on the lane's GPU: the kernels of this call run there
        const int dk = device_for(d_inouts[k], 1);
        if (g_rt.devs[dk]->physical != c.dev->physical) throw hip_failure{hipErrorInvalidValue, "ntt_device_batch: the vectors live on different devices", __LINE__};
    }
    c.phase_begin("ntt_kernels");
    // Runs of consecutive list entries with the same (direction, type) and no vector twice travel as ONE launch per pass
    // (blockIdx.y = vector; tuning ntt_batch=0: one launch sequence per vector).  List order is preserved between runs, so a
    // vector listed twice is still transformed twice, in order.
    const bool can_batch = tuning().ntt_batch && order == NTT_NN && lg > 8;
    std::vector<std::pair<size_t, size_t>> runs;
    size_t max_run = 1;
    for (size_t k = 0; k < count;) {
        const int dir = dirs ? dirs[k] : 0, type = types ? types[k] : 0;
        size_t e = k + 1;
        if (can_batch) {
            while (e < count && e - k < (size_t)NTT_BATCH_MAX && (dirs ? dirs[e] : 0) == dir && (types ? types[e] : 0) == type) {
                bool dup = false;
                for (size_t q = k; q < e && !dup; q++) dup = d_inouts[q] == d_inouts[e];
                if (dup) break;
                e++;
            }
        }
        runs.emplace_back(k, e);
        max_run = e - k > max_run ? e - k : max_run;
        k = e;
    }
    c.ntt_scratch.ensure((sizeof(fr_mem_t) << lg) * max_run);
    for (const auto& r : runs) {
        const size_t k = r.first, nv = r.second - r.first;
        const int dir = dirs ? dirs[k] : 0, type = types ? types[k] : 0;
        if (nv > 1)
            ntt_run_nn(c.ntt_ctx(), nullptr, c.ntt_scratch.as<fr_mem_t>(), (int)lg, dir, type, (fr_mem_t* const*)(d_inouts + k), (unsigned)nv);
        else
            ntt_run(c.ntt_ctx(), (fr_mem_t*)d_inouts[k], c.ntt_scratch.as<fr_mem_t>(), (int)lg, order, dir, type);
    }
    c.phase_end();
    HIP_TRY(hipGetLastError());
    c.sync_or_defer();
    API_END
}

// ---- polymul -----------------------------------------------------------------------------------
// PolyMultiplier::multiply on the device (multiplier.rs:70-134 through snarkvm.cu:188-247 / polynomial.cuh:104-266).  Like the
// reference's `Polynomial::Mul`, the upload of operand k + 1 (on the lane's second stream, into the other of two operand
// buffers) overlaps the transform and the pointwise product of operand k; events order the two streams:
//   ev[b]     "operand buffer b has been uploaded"      (alt -> main)
//   ev[2 + b] "operand buffer b has been consumed"      (main -> alt)
RustError snarkvm_polymul(void* out, size_t pcount, const void* polynomials, const void* plens, size_t ecount, const void* evaluations,
                          const void* elens, uint32_t lg) {
    // corner cases of snarkvm.cu:196-210 first (no device needed for the copy)
    const fr_mem_t* const* polys = (const fr_mem_t* const*)polynomials;
    const fr_mem_t* const* evals = (const fr_mem_t* const*)evaluations;
    const size_t* pl = (const size_t*)plens;
    const size_t* el = (const size_t*)elens;
    if (pcount + ecount == 0) return ok();
    if (pcount + ecount == 1 && pcount == 1) {
        memcpy(out, polys[0], sizeof(fr_mem_t) * pl[0]);
        return ok();
    }
    API_BEGIN
    check_ntt_args(lg, 0, 0, 0);
    const size_t n = (size_t)1 << lg;
    const size_t bytes = sizeof(fr_mem_t) * n;
    for (size_t k = 0; k < pcount; k++)
        if (pl[k] > n) throw hip_failure{hipErrorInvalidValue, "polymul: polynomial longer than the domain", __LINE__};
    for (size_t k = 0; k < ecount; k++)
        if (el[k] != n) throw hip_failure{hipErrorInvalidValue, "polymul: evaluation vector length != domain size", __LINE__};
    c.ntt_data.ensure(bytes);
    c.ntt_alt.ensure(bytes);
    c.ntt_scratch.ensure(bytes);
    c.ntt_acc.ensure(bytes);
    hipStream_t st = c.stream;
    fr_mem_t* buf[2] = {c.ntt_data.as<fr_mem_t>(), c.ntt_alt.as<fr_mem_t>()};
    fr_mem_t* acc = c.ntt_acc.as<fr_mem_t>();
    fr_mem_t* scratch = c.ntt_scratch.as<fr_mem_t>();
    const ntt_ctx_t cx = c.ntt_ctx();
    if (pcount + ecount == 1) {  // a single evaluation vector: zero-pad + inverse NTT (snarkvm.cu:203-208)
        HIP_TRY(hipMemsetAsync(acc, 0, bytes, st));
        HIP_TRY(hipMemcpyAsync(acc, evals[0], sizeof(fr_mem_t) * el[0], hipMemcpyHostToDevice, st));
        ntt_run(cx, acc, scratch, (int)lg, NTT_NN, NTT_INVERSE, NTT_STANDARD);
        HIP_TRY(hipMemcpyAsync(out, acc, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        const size_t total = pcount + ecount;
        // operand 0 goes straight into the accumulator on the main stream; operands k >= 1 alternate between the two buffers
        auto upload = [&](size_t k, fr_mem_t* dst, hipStream_t s) {
            if (k < pcount) {
                HIP_TRY(hipMemcpyAsync(dst, polys[k], sizeof(fr_mem_t) * pl[k], hipMemcpyHostToDevice, s));
                if (pl[k] < n) HIP_TRY(hipMemsetAsync(dst + pl[k], 0, sizeof(fr_mem_t) * (n - pl[k]), s));
            } else {
                HIP_TRY(hipMemcpyAsync(dst, evals[k - pcount], bytes, hipMemcpyHostToDevice, s));
            }
        };
        upload(0, acc, st);
        if (pcount > 0) ntt_run(cx, acc, scratch, (int)lg, NTT_NN, NTT_FORWARD, NTT_STANDARD);
        for (size_t k = 1; k < total; k++) {
            const int b = (int)(k & 1);
            // the kernels of operand k - 1 are already queued on the main stream: this (host-blocking, pageable) copy runs beside them
            if (k >= 3) HIP_TRY(hipStreamWaitEvent(c.alt, c.ev[2 + b], 0));  // buffer b was last consumed by operand k - 2
            upload(k, buf[b], c.alt);
            HIP_TRY(hipEventRecord(c.ev[b], c.alt));
            HIP_TRY(hipStreamWaitEvent(st, c.ev[b], 0));
            if (k < pcount) ntt_run(cx, buf[b], scratch, (int)lg, NTT_NN, NTT_FORWARD, NTT_STANDARD);
            hipLaunchKernelGGL(fr_pointwise_mul_kernel, dim3(blocks), dim3(256), 0, st, acc, acc, buf[b], n, 1);
            HIP_TRY(hipEventRecord(c.ev[2 + b], st));
        }
        ntt_run(cx, acc, scratch, (int)lg, NTT_NN, NTT_INVERSE, NTT_STANDARD);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, acc, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipStreamSynchronize(c.alt));
    }
    API_END
}

// ---- Fr vector helpers ---------------------------------------------------------------------------
RustError snarkvm_hip_fr_mul_device(void* d_out, const void* d_a, const void* d_b, size_t n) {
    API_BEGIN_DEV(device_for(d_out, n ? 1 : 0))
    if (n) {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(fr_pointwise_mul_kernel, dim3(blocks), dim3(256), 0, c.stream, (fr_mem_t*)d_out, (const fr_mem_t*)d_a, (const fr_mem_t*)d_b, n, 1);
        HIP_TRY(hipGetLastError());
        c.sync_or_defer();
    }
    API_END
}
RustError snarkvm_hip_fr_convert_device(void* d_out, const void* d_in, size_t n, int to_bigint) {
    API_BEGIN_DEV(device_for(d_out, n ? 1 : 0))
    if (n) {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(fr_to_bigint_kernel, dim3(blocks), dim3(256), 0, c.stream, (fr_mem_t*)d_out, (const fr_mem_t*)d_in, n, to_bigint);
        HIP_TRY(hipGetLastError());
        c.sync_or_defer();
    }
    API_END
}

// ---- prover-round polynomial kernels (poly.hip.h) ---------------------------------------------------
static fr_mem_t fr_mem_from_host(const void* p) {
    fr_mem_t m;
    memcpy(&m, p, sizeof m);
    return m;
}
static unsigned fr_grid(size_t n, unsigned block = 256) {
    const size_t b = (n + block - 1) / block;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
// operand `p` (n elements) as a device pointer: itself, or a staged copy in ctx.poly[slot]
static fr_mem_t* fr_stage_in(lane_t& c, int slot, const void* p, size_t n, int on_device) {
    if (on_device || !p) return (fr_mem_t*)p;
    c.poly[slot].ensure(sizeof(fr_mem_t) * (n ? n : 1));
    if (n) HIP_TRY(hipMemcpyAsync(c.poly[slot].p, p, sizeof(fr_mem_t) * n, hipMemcpyHostToDevice, c.stream));
    return c.poly[slot].as<fr_mem_t>();
}
static fr_mem_t* fr_stage_out(lane_t& c, int slot, void* p, size_t n, int on_device) {
    if (on_device || !p) return (fr_mem_t*)p;
    c.poly[slot].ensure(sizeof(fr_mem_t) * (n ? n : 1));
    return c.poly[slot].as<fr_mem_t>();
}
static void fr_finish_out(lane_t& c, fr_mem_t* d, void* p, size_t n, int on_device) {
    if (!on_device && p && n) HIP_TRY(hipMemcpyAsync(p, d, sizeof(fr_mem_t) * n, hipMemcpyDeviceToHost, c.stream));
}
// end of a vector call: device-resident operands may leave the wait to the calling thread's scope (runtime.hip.h), host operands never
static void fr_call_done(lane_t& c, int on_device) {
    if (on_device)
        c.sync_or_defer();
    else
        HIP_TRY(hipStreamSynchronize(c.stream));
}

static void check_strided(size_t n, size_t count, size_t stride, int on_device, const char* who) {
    (void)who;
    if (count == 0 || (count > 1 && (!on_device || stride < n)) || count > 65535)
        throw hip_failure{hipErrorInvalidValue, "strided batch of an Fr vector pass: needs device operands, 1 <= count <= 65535 and stride >= the vector length", __LINE__};
}
static RustError fr_vec_op_impl(int op, void* out, const void* a, const void* b, const void* c3, const void* scalar, size_t n, int on_device, size_t count,
                                size_t stride) {
    API_BEGIN_DEV(device_for(a, (on_device && n) ? 1 : 0))
    if (op < 0 || op > FR_OP_RSUB_SCALAR) throw hip_failure{hipErrorInvalidValue, "fr_vec_op: unknown op", __LINE__};
    check_strided(n, count, stride, on_device, "fr_vec_op");
    const bool need_b = op == FR_OP_ADD || op == FR_OP_SUB || op == FR_OP_MUL || op == FR_OP_MUL_SUB || op == FR_OP_AXPY;
    const bool need_c = op == FR_OP_MUL_SUB;
    const bool need_s = op == FR_OP_SCALE || op == FR_OP_SUB_SCALAR || op == FR_OP_AXPY || op == FR_OP_RSUB_SCALAR;
    if (n && (!out || !a || (need_b && !b) || (need_c && !c3) || (need_s && !scalar)))
        throw hip_failure{hipErrorInvalidValue, "fr_vec_op: missing operand", __LINE__};
    if (n) {
        fr_mem_t s{};
        if (need_s) s = fr_mem_from_host(scalar);
        const fr_mem_t* da = fr_stage_in(c, 0, a, n, on_device);
        const fr_mem_t* db = need_b ? fr_stage_in(c, 1, b, n, on_device) : nullptr;
        const fr_mem_t* dc = need_c ? fr_stage_in(c, 2, c3, n, on_device) : nullptr;
        fr_mem_t* dout = fr_stage_out(c, 3, out, n, on_device);
        hipLaunchKernelGGL(fr_vec_op_kernel, dim3(fr_grid(n), (unsigned)count), dim3(256), 0, c.stream, op, dout, da, db, dc, s, n, stride);
        HIP_TRY(hipGetLastError());
        fr_finish_out(c, dout, out, n, on_device);
        fr_call_done(c, on_device);
    }
    API_END
}
RustError snarkvm_hip_fr_vec_op(int op, void* out, const void* a, const void* b, const void* c3, const void* scalar, size_t n, int on_device) {
    return fr_vec_op_impl(op, out, a, b, c3, scalar, n, on_device, 1, 0);
}
RustError snarkvm_hip_fr_vec_op_strided(int op, void* out, const void* a, const void* b, const void* c3, const void* scalar, size_t n, size_t count, size_t stride) {
    return fr_vec_op_impl(op, out, a, b, c3, scalar, n, 1, count, stride);
}

// out[i - shift] = h_i = sum_{k >= i} in[k] m^(k - i) (and *first = h_0 when shift == 1); `out` may be null (only h_0 wanted).
// Scratch for the chunk values of every level lives in the lane's poly[4].
static void fr_suffix_horner(lane_t& c, const fr_mem_t* d_in, size_t n, const fr_mem_t& m, fr_mem_t* d_out, int shift, fr_mem_t* d_first, size_t count = 1,
                             size_t in_stride = 0, size_t out_stride = 0) {
    // `count` vectors in one launch sequence (blockIdx.y): inputs / outputs `in_stride` / `out_stride` elements apart, the chunk values
    // of vector y in its own slice of the scratch area, d_first[y] = h_0 of vector y
    hipStream_t st = c.stream;
    if (d_out && n >= 2048 && n <= ((size_t)1 << 20) && tuning().horner2) {
        // quotient wanted, a proof-sized polynomial: the three-launch form with a scan inside every workgroup (poly.hip.h); C coefficients per thread keep
        // <= 256 workgroups.  (Beyond 2^20 coefficients a thread would fold 32+ of them serially on a quarter of the chip's lanes - round 3 measured that
        // shape at 2^24: 1.69 vs 1.04 ms - and the chunk recursion below, 2^19 threads at its first level, is the better form.)
        uint32_t C = 8;
        while (((n + C - 1) / C + HORNER2_B - 1) / HORNER2_B > 256) C *= 2;
        const size_t nblocks = ((n + C - 1) / C + HORNER2_B - 1) / HORNER2_B;
        const size_t hs_stride = nblocks * HORNER2_B, bv_stride = nblocks;
        c.poly[4].ensure(sizeof(fr_mem_t) * (HORNER2_TAB + count * (hs_stride + bv_stride)));
        fr_mem_t* tab = c.poly[4].as<fr_mem_t>();
        fr_mem_t* hs = tab + HORNER2_TAB;
        fr_mem_t* bv = hs + count * hs_stride;
        hipLaunchKernelGGL(fr_horner2_tables_kernel, dim3(1), dim3(HORNER2_B), 0, st, m, tab, C);
        hipLaunchKernelGGL(fr_horner2_up_kernel, dim3((unsigned)nblocks, (unsigned)count), dim3(HORNER2_B), 0, st, d_in, n, m, (const fr_mem_t*)tab, hs, bv, C, in_stride,
                           hs_stride, bv_stride);
        hipLaunchKernelGGL(fr_horner2_down_kernel, dim3((unsigned)nblocks, (unsigned)count), dim3(HORNER2_B), 0, st, d_in, n, m, (const fr_mem_t*)tab, (const fr_mem_t*)hs,
                           (const fr_mem_t*)bv, C, d_out, shift, d_first, in_stride, hs_stride, bv_stride, out_stride);
        HIP_TRY(hipGetLastError());
        return;
    }
    int levels = 1;
    size_t total = 0;
    for (size_t t = n; t > 1;) {
        t = (t + POLY_CHUNK - 1) / POLY_CHUNK;
        total += t;
        levels++;
    }
    c.poly[4].ensure(sizeof(fr_mem_t) * (total * count + levels + 2));
    fr_mem_t* mult = c.poly[4].as<fr_mem_t>();
    fr_mem_t* cvbase = mult + levels + 1;
    const unsigned ny = (unsigned)count;
    hipLaunchKernelGGL(fr_horner_multipliers_kernel, dim3(1), dim3(1), 0, st, m, mult, levels);
    // up-sweep: level k holds the chunk values of level k - 1 (level 0 = the input)
    std::vector<const fr_mem_t*> in_at{d_in};
    std::vector<size_t> n_at{n}, stride_at{in_stride};
    fr_mem_t* next = cvbase;
    while (n_at.back() > 1) {
        const size_t cur = n_at.back();
        const size_t T = (cur + POLY_CHUNK - 1) / POLY_CHUNK;
        const int k = (int)n_at.size() - 1;
        hipLaunchKernelGGL(fr_horner_up_kernel, dim3((unsigned)((T + 255) / 256), ny), dim3(256), 0, st, in_at.back(), cur, mult + k, next, T, stride_at.back(), total);
        in_at.push_back(next);
        n_at.push_back(T);
        stride_at.push_back(total);
        next += T;
    }
    // the single value of the top level is h_0 of every level below; down-sweep turns each level's chunk values into
    // its suffix sums in place, the input level writes to `out`
    const int top = (int)n_at.size() - 1;
    for (int k = top; k >= 0; k--) {
        const size_t cur = n_at[k];
        const size_t T = (cur + POLY_CHUNK - 1) / POLY_CHUNK;
        const fr_mem_t* carry = (k < top) ? in_at[k + 1] : nullptr;
        if (k > 0) {
            if (k == top) continue;  // one element: it already is its own suffix sum
            hipLaunchKernelGGL(fr_horner_down_kernel, dim3((unsigned)((T + 255) / 256), ny), dim3(256), 0, st, in_at[k], cur, mult + k, carry, T,
                               (fr_mem_t*)in_at[k], 0, (fr_mem_t*)nullptr, total, total, total);
        } else if (d_out) {
            hipLaunchKernelGGL(fr_horner_down_kernel, dim3((unsigned)((T + 255) / 256), ny), dim3(256), 0, st, d_in, cur, mult, carry, T, d_out, shift,
                               d_first, in_stride, total, out_stride);
        } else if (d_first) {
            // only h_0: the value of the top level, or of the lone input element
            for (size_t y = 0; y < count; y++)
                HIP_TRY(hipMemcpyAsync(d_first + y, (top > 0 ? in_at[top] : d_in) + y * (top > 0 ? total : in_stride), sizeof(fr_mem_t), hipMemcpyDeviceToDevice, st));
        }
    }
    HIP_TRY(hipGetLastError());
}

static RustError fr_divide_by_linear_impl(void* quotient, void* remainder, const void* poly, size_t n, const void* point, int on_device, size_t count,
                                          size_t stride) {
    API_BEGIN_DEV(device_for(poly, (on_device && n) ? 1 : 0))
    if (!point || (n && !poly)) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_linear: missing operand", __LINE__};
    check_strided(n, count, stride, on_device, "fr_divide_by_linear");
    if (on_device && quotient && n > 1) {  // header: different workgroups own adjacent coefficient ranges - an in-place division would race
        const size_t span = sizeof(fr_mem_t) * ((count - 1) * stride + n);
        const uint8_t *q = (const uint8_t*)quotient, *p = (const uint8_t*)poly;
        if (q < p + span && p < q + span) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_linear: quotient overlaps poly (device operands must not alias)", __LINE__};
    }
    if (n == 0) {
        if (remainder) memset(remainder, 0, sizeof(fr_mem_t) * count);
    } else {
        const fr_mem_t z = fr_mem_from_host(point);
        const fr_mem_t* din = fr_stage_in(c, 0, poly, n, on_device);
        fr_mem_t* dq = (quotient && n > 1) ? fr_stage_out(c, 1, quotient, n - 1, on_device) : nullptr;
        c.poly[2].ensure(sizeof(fr_mem_t) * count);
        fr_mem_t* drem = c.poly[2].as<fr_mem_t>();
        fr_suffix_horner(c, din, n, z, dq, 1, drem, count, stride, stride);
        if (dq) fr_finish_out(c, dq, quotient, n - 1, on_device);
        // device operands inside a scope: delivered by snarkvm_hip_scope_end; host operands: the call waits below, the value is there on return
        if (remainder) c.host_result(remainder, drem, sizeof(fr_mem_t) * count, on_device != 0);
        fr_call_done(c, on_device);
    }
    API_END
}
RustError snarkvm_hip_fr_divide_by_linear(void* quotient, void* remainder, const void* poly, size_t n, const void* point, int on_device) {
    return fr_divide_by_linear_impl(quotient, remainder, poly, n, point, on_device, 1, 0);
}
RustError snarkvm_hip_fr_divide_by_linear_strided(void* quotients, void* remainders, const void* polys, size_t n, const void* point, size_t count, size_t stride) {
    return fr_divide_by_linear_impl(quotients, remainders, polys, n, point, 1, count, stride);
}

static void fr_batch_inverse_run(lane_t& c, fr_mem_t* d_v, size_t n, const fr_mem_t& coeff) {
    // >= 32 elements per thread amortise the per-thread Fermat inversion; cap the thread count for huge vectors
    size_t T = (n + 31) / 32;
    if (T > (size_t)1 << 17) T = (size_t)1 << 17;
    c.poly[4].ensure(sizeof(fr_mem_t) * n);
    hipLaunchKernelGGL(fr_batch_inverse_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, c.stream, d_v, n, coeff, c.poly[4].as<fr_mem_t>(), T);
    HIP_TRY(hipGetLastError());
}
RustError snarkvm_hip_fr_batch_inversion_and_mul(void* inout, size_t n, const void* coeff, int on_device) {
    API_BEGIN_DEV(device_for(inout, (on_device && n) ? 1 : 0))
    if (n) {
        if (!inout || !coeff) throw hip_failure{hipErrorInvalidValue, "fr_batch_inversion_and_mul: missing operand", __LINE__};
        fr_mem_t* dv = fr_stage_in(c, 0, inout, n, on_device);
        fr_batch_inverse_run(c, dv, n, fr_mem_from_host(coeff));
        fr_finish_out(c, dv, inout, n, on_device);
        fr_call_done(c, on_device);
    }
    API_END
}

static void fr_distribute_powers_run(lane_t& c, fr_mem_t* d_v, size_t n, const fr_mem_t& g, const fr_mem_t& cmul) {
    size_t T = (n + 31) / 32;
    if (T > (size_t)1 << 17) T = (size_t)1 << 17;
    hipLaunchKernelGGL(fr_distribute_powers_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, c.stream, d_v, n, g, cmul, T);
    HIP_TRY(hipGetLastError());
}
RustError snarkvm_hip_fr_distribute_powers(void* inout, size_t n, const void* g, const void* cmul, int on_device) {
    API_BEGIN_DEV(device_for(inout, (on_device && n) ? 1 : 0))
    if (n) {
        if (!inout || !g || !cmul) throw hip_failure{hipErrorInvalidValue, "fr_distribute_powers: missing operand", __LINE__};
        fr_mem_t* dv = fr_stage_in(c, 0, inout, n, on_device);
        fr_distribute_powers_run(c, dv, n, fr_mem_from_host(g), fr_mem_from_host(cmul));
        fr_finish_out(c, dv, inout, n, on_device);
        fr_call_done(c, on_device);
    }
    API_END
}

// TWO_ADIC_ROOT_OF_UNITY (fr.rs:115-120), memory form - the host copy of ntt.hip.h's device table
static const uint32_t FR_TWO_ADIC_ROOT_MEM_HOST[8] = {0xda3ad648u, 0xaf80da4du, 0xfc381dacu, 0x5e223adbu,
                                                      0xb2f92525u, 0x03ba0666u, 0x3befb0ceu, 0x0f906c5bu};
RustError snarkvm_hip_fr_lagrange_coefficients(void* out, uint32_t lg, const void* tau, int on_device) {
    API_BEGIN_DEV(device_for(out, (on_device && out) ? 1 : 0))
    if (lg > 30) throw hip_failure{hipErrorInvalidValue, "fr_lagrange_coefficients: lg_domain_size > 30", __LINE__};
    if (!out || !tau) throw hip_failure{hipErrorInvalidValue, "fr_lagrange_coefficients: missing operand", __LINE__};
    const size_t n = (size_t)1 << lg;
    // scalar set-up with the same arithmetic compiled for the host (domain.rs:118-147, 258-264)
    fr_t omega = fr_t::unpack(FR_TWO_ADIC_ROOT_MEM_HOST).from_mem_mont();
    for (uint32_t i = lg; i < 47; i++) omega = omega.sqr();
    const fr_mem_t tau_mem = fr_mem_from_host(tau);
    const fr_t tau_i = fr_t::load(&tau_mem).from_mem_mont();
    const fr_t t_size = tau_i.pow_u64((uint64_t)n);
    fr_mem_t one_mem, omega_mem;
    fr_t::one().to_mem_mont().store(&one_mem);
    omega.to_mem_mont().store(&omega_mem);
    fr_mem_t* du = fr_stage_out(c, 0, out, n, on_device);
    hipStream_t st = c.stream;
    hipLaunchKernelGGL(fr_fill_kernel, dim3(fr_grid(n)), dim3(256), 0, st, du, n, one_mem);
    fr_distribute_powers_run(c, du, n, omega_mem, one_mem);  // u_i = omega^i
    if (t_size == fr_t::one()) {
        hipLaunchKernelGGL(fr_onehot_kernel, dim3(fr_grid(n)), dim3(256), 0, st, du, n, tau_mem, one_mem);
    } else {
        fr_mem_t l_mem;
        ((t_size - fr_t::one()) * fr_t::from_u32((uint32_t)n).inverse()).to_mem_mont().store(&l_mem);
        hipLaunchKernelGGL(fr_vec_op_kernel, dim3(fr_grid(n)), dim3(256), 0, st, (int)FR_OP_RSUB_SCALAR, du, (const fr_mem_t*)du, (const fr_mem_t*)nullptr,
                           (const fr_mem_t*)nullptr, tau_mem, n, (size_t)0);  // tau - omega^i
        fr_batch_inverse_run(c, du, n, one_mem);
        fr_distribute_powers_run(c, du, n, omega_mem, l_mem);  // * l * omega^i
    }
    HIP_TRY(hipGetLastError());
    fr_finish_out(c, du, out, n, on_device);
    fr_call_done(c, on_device);
    API_END
}

static RustError fr_divide_by_vanishing_impl(void* quotient, void* remainder, const void* poly, size_t len, size_t domain_size, int on_device, size_t count,
                                             size_t stride) {
    API_BEGIN_DEV(device_for(poly, (on_device && len) ? 1 : 0))
    if (domain_size == 0) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_vanishing: empty domain", __LINE__};
    check_strided(len, count, stride, on_device, "fr_divide_by_vanishing");
    if (len) {
        if (!poly || !remainder || (len > domain_size && !quotient)) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_vanishing: missing operand", __LINE__};
        const size_t qlen = len > domain_size ? len - domain_size : 0;
        const size_t rlen = len < domain_size ? len : domain_size;
        const fr_mem_t* din = fr_stage_in(c, 0, poly, len, on_device);
        fr_mem_t* dq = qlen ? fr_stage_out(c, 1, quotient, qlen, on_device) : nullptr;
        fr_mem_t* dr = fr_stage_out(c, 2, remainder, rlen, on_device);
        const size_t threads = qlen > rlen ? qlen : rlen;
        hipLaunchKernelGGL(fr_fold_vanishing_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)count), dim3(256), 0, c.stream, din, len, domain_size, dq, dr,
                           stride);
        HIP_TRY(hipGetLastError());
        if (qlen) fr_finish_out(c, dq, quotient, qlen, on_device);
        fr_finish_out(c, dr, remainder, rlen, on_device);
        fr_call_done(c, on_device);
    }
    API_END
}
RustError snarkvm_hip_fr_divide_by_vanishing(void* quotient, void* remainder, const void* poly, size_t len, size_t domain_size, int on_device) {
    return fr_divide_by_vanishing_impl(quotient, remainder, poly, len, domain_size, on_device, 1, 0);
}
RustError snarkvm_hip_fr_divide_by_vanishing_strided(void* quotients, void* remainders, const void* polys, size_t len, size_t domain_size, size_t count, size_t stride) {
    return fr_divide_by_vanishing_impl(quotients, remainders, polys, len, domain_size, 1, count, stride);
}
RustError snarkvm_hip_fr_mul_by_vanishing(void* out, const void* poly, size_t len, size_t domain_size, int on_device) {
    API_BEGIN_DEV(device_for(out, (on_device && out) ? 1 : 0))
    const size_t olen = len + domain_size;
    if (olen) {
        if (!out || (len && !poly)) throw hip_failure{hipErrorInvalidValue, "fr_mul_by_vanishing: missing operand", __LINE__};
        const fr_mem_t* din = fr_stage_in(c, 0, poly, len, on_device);
        fr_mem_t* dout = fr_stage_out(c, 1, out, olen, on_device);
        hipLaunchKernelGGL(fr_mul_vanishing_kernel, dim3((unsigned)((olen + 255) / 256)), dim3(256), 0, c.stream, din, len, domain_size, dout);
        HIP_TRY(hipGetLastError());
        fr_finish_out(c, dout, out, olen, on_device);
        fr_call_done(c, on_device);
    }
    API_END
}

// ---- setup-time group operations (group.hip.h) -------------------------------------------------------
RustError snarkvm_hip_g1_fixed_base_msm(void* out_projective, const void* g_affine, const void* scalars, size_t n) {
    API_BEGIN
    if (n) {
        if (!out_projective || !g_affine || !scalars) throw hip_failure{hipErrorInvalidValue, "g1_fixed_base_msm: null argument", __LINE__};
        hipStream_t st = c.stream;
        // the base in the engine's native form, through the regular conversion kernel
        c.bases_tmp.ensure(256 + sizeof(g1_aff_mem_t));
        HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, g_affine, 104, hipMemcpyHostToDevice, st));
        g1_aff_mem_t* d_g = (g1_aff_mem_t*)(c.bases_tmp.as<uint8_t>() + 256);
        convert_bases<fq_t>(c, c.bases_tmp.as<uint8_t>(), 104, 1, d_g);
        g1_aff_mem_t g_native;
        HIP_TRY(hipMemcpyAsync(&g_native, d_g, sizeof g_native, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const size_t entries = (size_t)FIXED_OUTER << FIXED_WINDOW;
        c.poly[0].ensure(entries * sizeof(g1_aff_mem_t));
        c.poly[1].ensure(n * 32);
        c.poly[2].ensure(n * 144);
        hipLaunchKernelGGL(g1_fixed_table_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st, g_native, c.poly[0].as<g1_aff_mem_t>());
        HIP_TRY(hipMemcpyAsync(c.poly[1].p, scalars, n * 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(g1_fixed_msm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c.poly[0].as<g1_aff_mem_t>(),
                           c.poly[1].as<fr_mem_t>(), n, c.poly[2].as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_projective, c.poly[2].p, n * 144, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    API_END
}
RustError snarkvm_hip_g1_group_ntt(void* inout_projective, uint32_t lg, int inverse) {
    API_BEGIN
    if (lg > 24) throw hip_failure{hipErrorInvalidValue, "g1_group_ntt: lg_domain_size > 24", __LINE__};
    if (!inout_projective) throw hip_failure{hipErrorInvalidValue, "g1_group_ntt: null argument", __LINE__};
    const size_t n = (size_t)1 << lg;
    hipStream_t st = c.stream;
    c.poly[0].ensure(n * 144);
    c.poly[1].ensure(n * sizeof(g1_xyzz_mem_t));
    c.poly[2].ensure((n / 2 + 1) * sizeof(fr_mem_t));
    uint32_t* d_jac = c.poly[0].as<uint32_t>();
    g1_xyzz_mem_t* d_pts = c.poly[1].as<g1_xyzz_mem_t>();
    fr_mem_t* d_tw = c.poly[2].as<fr_mem_t>();
    HIP_TRY(hipMemcpyAsync(d_jac, inout_projective, n * 144, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(g1_jac_to_xyzz_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const uint32_t*)d_jac, d_pts, n);
    if (lg > 0) {
        // twiddles root^k (k < n/2) as canonical integers: ones -> distribute_powers -> to_bigint, all on the device
        fr_t omega = fr_t::unpack(FR_TWO_ADIC_ROOT_MEM_HOST).from_mem_mont();
        for (uint32_t i = lg; i < 47; i++) omega = omega.sqr();  // group_gen of the 2^lg domain (fft_field.rs:75-85)
        if (inverse) omega = omega.inverse();
        fr_mem_t one_mem, root_mem;
        fr_t::one().to_mem_mont().store(&one_mem);
        omega.to_mem_mont().store(&root_mem);
        const size_t h = n / 2;
        hipLaunchKernelGGL(fr_fill_kernel, dim3(fr_grid(h)), dim3(256), 0, st, d_tw, h, one_mem);
        fr_distribute_powers_run(c, d_tw, h, root_mem, one_mem);
        hipLaunchKernelGGL(fr_to_bigint_kernel, dim3(fr_grid(h)), dim3(256), 0, st, d_tw, (const fr_mem_t*)d_tw, h, 1);
        // four lanes per butterfly from 32 points on (group.hip.h: every lane of every wave must own a butterfly); tuning group_quad=0: one lane per butterfly
        const bool quad = n >= 32 && tuning().group_quad;
        for (size_t half = n / 2; half >= 1; half >>= 1) {
            if (quad)
                hipLaunchKernelGGL(g1_ntt_stage_quad_kernel, dim3((unsigned)(n * 2 / 64)), dim3(64), 0, st, d_pts, n, half, (const fr_mem_t*)d_tw, n / (2 * half));
            else
                hipLaunchKernelGGL(g1_ntt_stage_kernel, dim3((unsigned)((n / 2 + 63) / 64)), dim3(64), 0, st, d_pts, n, half, (const fr_mem_t*)d_tw, n / (2 * half));
        }
        hipLaunchKernelGGL(g1_bitrev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_pts, n, (int)lg);
        if (inverse) {  // * size_inv (domain.rs:190)
            fr_mem_t k_int;
            fr_t::from_u32((uint32_t)n).inverse().mont_to_int().store(&k_int);
            if (quad)
                hipLaunchKernelGGL(g1_scale_quad_kernel, dim3((unsigned)(n * 4 / 64)), dim3(64), 0, st, d_pts, n, k_int);
            else
                hipLaunchKernelGGL(g1_scale_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_pts, n, k_int);
        }
    }
    hipLaunchKernelGGL(g1_xyzz_to_jac_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const g1_xyzz_mem_t*)d_pts, d_jac, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(inout_projective, d_jac, n * 144, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    API_END
}

// ---- test hook: the signed-limb butterfly arithmetic of frs.hip.h against the exact arithmetic, on the host --------------------
// Passes of up to nine butterfly stages over a population of 16 values (u, v) -> (u + v, (u - v) w) with random canonical twiddles
// (w = 1 takes the product-free branch), no canonical form in between - exactly what ntt_pass_kernel_v2<., ntt_arith_s> does
// between a pass' loads and its closing product -, every value compared with the exact arithmetic after every stage; then the
// closing product, the bare reduction, and operands with extreme limbs.  0 = identical; > 0: first failing step; < 0: edge case.
int snarkvm_hip_selftest_fr_signed(uint64_t seed, int iters) {
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto rnd = [&]() {  // a canonical residue
        fr_t a;
        for (int i = 0; i < 9; i++) a.v[i] = (uint32_t)(next() & LIMB_MASK);
        a.v[8] &= 0x000fffffu;  // < 2^252 < r
        return a;
    };
    const int NV = 16;
    fr_t ex[NV];
    frs_t sg[NV];
    int step = 0;
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i < NV; i++) {
            ex[i] = rnd();
            if ((it & 7) == 7)
                for (int l = 0; l < 8; l++) ex[i].v[l] = (i & 1) ? LIMB_MASK : 0u;  // extreme limbs
            if (!(ex[i].v[8] < FrP::MOD[8])) ex[i].v[8] = FrP::MOD[8] - 1;
            sg[i] = frs_t::from_canonical(ex[i]);
        }
        const int stages = 1 + (int)(next() % 9);
        for (int s = 0; s < stages; s++) {
            const int dist = 1 << (s % 4);
            for (int i = 0; i < NV; i++) {
                if (i & dist) continue;
                const int j = i | dist;
                const bool unit = (next() & 3) == 0;
                const fr_t w = unit ? fr_t::one() : rnd();  // internal form (w * 2^261)
                const fr_t e_sum = ex[i] + ex[j], e_dif = (ex[i] - ex[j]) * w;
                const frs_t s_sum = frs_t::add_norm(sg[i], sg[j]);
                frs_t s_dif;
                if (unit) {
                    frs_t neg;
                    for (int l = 0; l < 9; l++) neg.v[l] = -sg[j].v[l];
                    s_dif = frs_t::add_norm(sg[i], neg);
                } else {
                    s_dif = frs_t::mul(frs_t::sub_raw(sg[i], sg[j]), frs_t::twiddle_form(w));
                }
                ex[i] = e_sum, ex[j] = e_dif, sg[i] = s_sum, sg[j] = s_dif;
                step++;
                // (a value that took the sum branch s times is up to 2^s r wide: it is compared after a product by one, the way a pass
                // ends - to_canonical alone covers (-3 r, 2 r))
                const fr_t one_s = fr_t::from_table(FrS::C290);
                if (frs_t::mul(sg[i], one_s).to_canonical() != ex[i] || frs_t::mul(sg[j], one_s).to_canonical() != ex[j]) return step;
            }
        }
        // closing product and bare reduction
        const fr_t w = rnd();
        for (int i = 0; i < NV; i++) {
            step++;
            if (frs_t::mul(sg[i], frs_t::twiddle_form(w)).to_canonical() != ex[i] * w) return step;
            fr_t one_plain = fr_t::zero();
            one_plain.v[0] = 1;
            if (frs_t::reduce_only(sg[i]).to_canonical() != frs_t::mul(sg[i], one_plain).to_canonical()) return -step;
            // the folded table form: (x * t 2^580 / 2^290) / 2^290 = x t
            const fr_t folded = w * fr_t::from_table(FrS::C580);
            if (frs_t::reduce_only(frs_t::mul(sg[i], folded)).to_canonical() != ex[i] * w) return -(1000000 + step);
        }
    }
    return 0;
}

}  // extern "C"
