/* snarkvm_hip.h - C ABI of the MI355X (gfx950) MSM / NTT backend for snarkVM.
 *
 * Part 1 is the drop-in boundary: the three symbols that `snarkvm-algorithms-cuda` binds
 * (reference: algorithms/cuda/src/lib.rs:42-69, defined by algorithms/cuda/cuda/snarkvm_api.cu:52-84).
 * A build of the reference with its `cuda` feature can link this library instead of the nvcc object
 * without touching any Rust caller (INTEGRATION.md shows the binding).
 *
 * Part 2 is an extension ABI (not in the reference): device-resident operands, SRS registration,
 * timing hooks.  Part 3 are test hooks.
 *
 * All functions are thread-safe and never unwind.  Concurrent callers (rayon workers: one commitment each) are handed
 * different (device, stream) lanes - the reference's resource channel, snarkvm.cu:84,146-150 - and block only when every
 * lane of every selected device is busy.  No torch / C++ types cross this boundary.
 */
#ifndef SNARKVM_HIP_H
#define SNARKVM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sppark `cuda::Error` as seen by Rust (algorithms/cuda/src/lib.rs:20 `sppark::cuda_error!()`):
 * code == 0 means success; `message` is NULL or a malloc()ed C string that the caller frees
 * (build.rs:79 -DTAKE_RESPONSIBILITY_FOR_ERROR_MESSAGE).  Any non-zero code makes the Rust caller
 * fall back to its CPU path (msm/variable_base/mod.rs:39-43, fft/domain.rs:385-391). */
typedef struct {
    int32_t code;
    char *message;
} RustError;

enum NTTInputOutputOrder { NN = 0, NR = 1, RN = 2, RR = 3 }; /* lib.rs:22-28 */
enum NTTDirection { Forward = 0, Inverse = 1 };              /* lib.rs:30-34 */
enum NTTType { Standard = 0, Coset = 1 };                    /* lib.rs:36-40 */

/* ---------------------------------------------------------------------------------------------
 * Part 1 - the reference's FFI
 * ------------------------------------------------------------------------------------------- */

/* lib.rs:43-49 / snarkvm_api.cu:53-62.  In-place NTT of 2^lg_domain_size Fr elements (32 B each,
 * Montgomery form, host memory).  NN/Forward/Standard == EvaluationDomain::fft_in_place;
 * Inverse includes the 1/n scaling; Forward+Coset multiplies x[j] by 22^j first; Inverse+Coset
 * multiplies the result by 22^-j (fft/domain.rs:201-206, 403-443).  Returns an error for
 * lg_domain_size > 26 (the caller then uses its CPU path). */
RustError snarkvm_ntt(void *inout, uint32_t lg_domain_size, enum NTTInputOutputOrder ntt_order,
                      enum NTTDirection ntt_direction, enum NTTType ntt_type);

/* lib.rs:51-60 / snarkvm_api.cu:64-75.  Product of `pcount` coefficient-form polynomials and
 * `ecount` evaluation-form vectors over the 2^lg_domain_size domain (PolyMultiplier::multiply,
 * fft/polynomial/multiplier.rs:70-134).  `polynomials` / `evaluations` are arrays of pointers to Fr
 * vectors, `plens` / `elens` their lengths (size_t); every elens[i] must equal the domain size.
 * `out` holds 2^lg_domain_size elements.  Corner cases follow snarkvm.cu:196-210. */
RustError snarkvm_polymul(void *out, size_t pcount, const void *polynomials, const void *plens, size_t ecount,
                          const void *evaluations, const void *elens, uint32_t lg_domain_size);

/* lib.rs:62-68 / snarkvm_api.cu:77-83.  out (G1Projective: Jacobian X, Y, Z, 3 x 48 B Montgomery)
 * = sum_i scalars[i] * points[i].  `points_with_infinity` is a Rust `[G1Affine]` (x, y, infinity flag;
 * stride `ffi_affine_sz` = 104 bytes), `scalars` are npoints canonical 256-bit integers < r. */
RustError snarkvm_msm(void *out, const void *points_with_infinity, size_t npoints, const void *scalars,
                      size_t ffi_affine_sz);
/* Multi-GPU: like the reference (snarkvm.cu:254-295) a call of >= 2^19 pairs is cut into point-range chunks that are dealt
 * to every selected device (and to two lanes per device, so that the upload of one chunk overlaps the computation of the
 * previous one); the per-chunk partial results are combined on the host.
 * Ownership: like the reference's symbol the call is STATELESS - `points_with_infinity` and `scalars` are only read during the
 * call and nothing about them is retained after return.
 * Opt-in extension (off unless SNARKVM_HIP_BASE_CACHE=1/2/4/8/16 is set in the environment): a host base range passed a SECOND
 * time with the same address and length is kept in HBM (converted, with precomputed tables - 17 x 15-bit below 2^18 points,
 * 16 x 16, 13 x 20 from 2^21, 12 x 22 from 2^24 - on every device) and later calls whose bases are a slice of it skip upload
 * and conversion - the reference's callers always pass slices of one long-lived `powers_of_beta_g` vector
 * (kzg10/mod.rs:117-119).  By setting the variable the caller promises that such vectors are immutable while the process
 * uses them (a hit is verified against raw copies of every 64th point of the slice, which cannot catch every mutation);
 * host memory is only read inside the slice the current call passed.  SNARKVM_HIP_BASE_CACHE_MB caps the HBM bytes per device
 * (default 65536).  Code that can be changed should call snarkvm_hip_register_bases* + snarkvm_hip_msm_registered*.
 * snarkvm_hip_set_base_cache(tables) is the API form of the variable (0, 1, 2, 4, 8 or 16; overrides the environment from then on; 0 also drops
 * every cached range) - for a host that cannot set the environment before the library is loaded. */
RustError snarkvm_hip_set_base_cache(int tables);

/* ---------------------------------------------------------------------------------------------
 * Part 2 - extension ABI (device-resident data, SRS registration, instrumentation)
 * ------------------------------------------------------------------------------------------- */

/* Number of visible HIP devices (0 if none).  Never fails. */
int snarkvm_hip_device_count(void);
/* Number of HIP streams (each with its own workspace) snarkvm_hip_msm_registered_batch cycles through for instances of
 * up to `npoints` pairs: 8 below 2^20, 3 above (SNARKVM_HIP_TUNING=lanes=N overrides).  A caller that wants allocation-free timed
 * regions warms up that many instances first (snarkvm_hip_alloc_stats tells whether anything grew). */
int snarkvm_hip_batch_lanes(size_t npoints);
/* Devices used by this process (the reference loops over `ngpus()`, snarkvm.cu:123-151).  Default: every visible device, or
 * the comma-separated list in SNARKVM_HIP_DEVICES.  snarkvm_hip_set_devices selects them explicitly; it must be called before
 * the first compute call (afterwards only the identical list is accepted).  An id may be listed more than once: each entry
 * is an independent logical device (own streams, workspaces, base replicas) - how the multi-device paths are tested on a
 * one-GPU box.  snarkvm_hip_set_device(d) == set_devices({d}): the one-process-per-GPU deployment (LOCAL_RANK mapping is the
 * caller's business).  snarkvm_hip_num_devices: logical devices in use (0 if none can be initialised). */
RustError snarkvm_hip_set_devices(const int32_t *ids, size_t n);
RustError snarkvm_hip_set_device(int device);
int snarkvm_hip_num_devices(void);

/* Device memory for a host that has no HIP binding of its own (a Rust prover that keeps its polynomials in HBM between the transforms
 * and the commitments; the reference's plugin owns every device byte itself, snarkvm.cu:51,123-151): what the device-resident entry
 * points below take as `d_*` pointers.  `device` of snarkvm_hip_malloc: index into the devices in use (snarkvm_hip_set_devices), or -1 =
 * the device of the calling thread's open scope, logical device 0 outside a scope.  Blocks are NOT counted by snarkvm_hip_alloc_stats (that
 * reports the library's own workspace growth).  snarkvm_hip_free waits for all queued work of the device first (hipFree).
 * Copies with a HOST side (h2d, d2h) return when the copy is complete: `src` may be reused, `dst` may be read.  Inside a scope of the
 * calling thread they are ordered behind the work the scope has queued on its stream (and wait for that, not for the scope's enqueued MSMs).
 * snarkvm_hip_memcpy_d2d and snarkvm_hip_memset are device-side operations like the snarkvm_hip_fr_* kernels: inside a scope they are only
 * enqueued on the scope's stream, in call order ("the prover produced a polynomial"); outside a scope they return when done.  d2d ranges
 * must not overlap; the two pointers may live on different devices in use. */
RustError snarkvm_hip_malloc(void **d_ptr, size_t bytes, int device);
RustError snarkvm_hip_free(void *d_ptr);
RustError snarkvm_hip_memcpy_h2d(void *d_dst, const void *src, size_t bytes);
RustError snarkvm_hip_memcpy_d2h(void *dst, const void *d_src, size_t bytes);
RustError snarkvm_hip_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes);
RustError snarkvm_hip_memset(void *d_dst, int value, size_t bytes);

/* Same as snarkvm_ntt but `d_inout` is device memory (the call runs on the device that owns it). */
RustError snarkvm_hip_ntt_device(void *d_inout, uint32_t lg_domain_size, int ntt_order, int ntt_direction,
                                 int ntt_type);
/* `count` independent in-place transforms of 2^lg_domain_size elements over device vectors on one device: one enqueue, ONE
 * synchronisation (the iNTTs of a prover round, e.g. z_a, z_b, z_c: algorithms/src/snark/varuna/ahp/prover/round_functions/
 * second.rs:104-113).  ntt_directions / ntt_types: one value per vector, or NULL for all forward / all standard.  A vector
 * listed twice is transformed twice, in list order.  Consecutive distinct vectors with the same direction and type share ONE
 * kernel launch per pass (up to 48 vectors). */
RustError snarkvm_hip_ntt_device_batch(void *const *d_inouts, size_t count, uint32_t lg_domain_size, int ntt_order,
                                       const int *ntt_directions, const int *ntt_types);

/* Deferred synchronisation for device-resident operands.  Between snarkvm_hip_scope_begin (d_any: any device pointer on the GPU
 * to use, or NULL for any GPU) and snarkvm_hip_scope_end, calls of THIS thread whose operands and results live in device memory
 * - snarkvm_hip_ntt_device, _ntt_device_batch, _fr_mul_device, _fr_convert_device, _memcpy_d2d, _memset and the snarkvm_hip_fr_* vector
 * kernels with on_device = 1 - are enqueued on one stream, in call order, and return without waiting; snarkvm_hip_scope_end waits once.  The
 * 32-byte host `remainder` of snarkvm_hip_fr_divide_by_linear with on_device = 1 is delivered by scope_end.  Every other call (MSMs,
 * host buffers, a pointer on another GPU) first waits for the scope's queued work, so results are the same as without a scope - and
 * then runs on the scope's own stream: a thread inside a scope never waits for a free stream.  Scopes do not nest; a scope must be
 * ended by the thread that began it (a thread that ENDS with its scope still open returns the scope's streams to the pool: the queued work is
 * waited for, results it still owed are dropped). */
RustError snarkvm_hip_scope_begin(const void *d_any);
RustError snarkvm_hip_scope_end(void);
/* The same with options.  SNARKVM_HIP_SCOPE_ASYNC_MSM: snarkvm_hip_msm_registered[_ex / _batch / _batch_ex] and
 * snarkvm_hip_msm_g2_registered[_batch] calls of this thread whose scalars live in the scope's GPU memory are ENQUEUED too (on further
 * streams of the scope, behind everything the scope has been given so far) and return at once: their `out` buffers are written by
 * snarkvm_hip_scope_end - or by any call that has to wait for the scope - and must stay valid until then; the scalar vectors may be
 * overwritten by later calls of the scope (those wait on the GPU until the MSM has read them).  How one prover thread keeps the GPU
 * busy across the rounds of a proof: the commitments of round k run beside the transforms of round k + 1 and their host finishes run
 * while the GPU works (the reference's rayon workers overlap the same way on the CPU, polycommit/sonic_pc/mod.rs:186-245).
 * SNARKVM_HIP_SCOPE_STABLE_INPUTS (with ASYNC_MSM): the caller promises not to touch the scalar vectors of its enqueued MSMs before the
 * scope ends; the scope's stream then never waits for an MSM (without the flag every later call of the scope waits, on the GPU, until
 * the MSM has read its scalars - which an MSM queued behind others does late).
 * A thread holds at most 1 + 7 streams per scope and a GPU lends at most 12 of its 16 streams to scopes: a scope_begin beyond that waits,
 * and a scope that finds no further stream free runs its MSMs on the streams it has.
 * snarkvm_hip_scope_stream: the hipStream_t the scope's device-resident calls are enqueued on (NULL outside a scope) - a caller that
 * produces operands with its own kernels or copies (hipMemcpyAsync, torch.cuda.ExternalStream) orders them with the scope's calls by
 * using this stream.
 * SNARKVM_HIP_SCOPE_MSM_IN_STREAM (with ASYNC_MSM): the MSMs are enqueued on the scope's OWN stream, in order with its transforms, instead of
 * a further stream - no hand-off between streams; for the MSMs a caller collects (snarkvm_hip_scope_collect) before it issues anything else,
 * beside which nothing could run anyway.  snarkvm_hip_scope_set_flags changes the flags of the open scope for the calls that follow: a prover
 * enqueues its independent MSM (G2) on a further stream first and then runs the transcript-ordered commitment rounds in-stream, each
 * collected before the next round's challenge is drawn, with the independent MSM filling the gaps. */
enum { SNARKVM_HIP_SCOPE_ASYNC_MSM = 1, SNARKVM_HIP_SCOPE_STABLE_INPUTS = 2, SNARKVM_HIP_SCOPE_MSM_IN_STREAM = 4 };
/* snarkvm_hip_scope_collect(out): waits until the MSM call that was given `out` as its (first) output buffer is done and writes its outputs
 * (out == NULL: every MSM this thread's scope has enqueued so far); the scope stays open, the work queued on its own stream is NOT waited for
 * and the other enqueued MSMs stay pending - except that outputs of other MSMs whose results have ALREADY arrived may be written too while this
 * call would only wait (their `out` buffers are owed by scope_end at the latest; this moves their host finish off the end of the scope).
 * What a prover needs between two rounds: the commitments of round k
 * go into the Fiat-Shamir transcript before the challenge of round k + 1 exists (snark/varuna/varuna.rs:336 ff.), while transforms that do
 * not depend on that challenge - and the tails of earlier MSMs, and an independent MSM - keep running. */
RustError snarkvm_hip_scope_begin_ex(const void *d_any, uint32_t flags);
RustError snarkvm_hip_scope_collect(const void *out);
RustError snarkvm_hip_scope_set_flags(uint32_t flags);
void *snarkvm_hip_scope_stream(void);

/* Register a base vector once (SRS powers are static per proving key; the reference re-uploads
 * 104 B/point on every call, snarkvm.cu:262-275).  `points` is a Rust `[G1Affine]` with the given
 * stride, in host (on_device = 0) or device (on_device = 1) memory.  Every selected device receives its own replica
 * (host source: parallel uploads; device source: peer copies of the finished tables).  Returns an opaque handle. */
typedef struct snarkvm_hip_bases snarkvm_hip_bases_t;
RustError snarkvm_hip_register_bases(snarkvm_hip_bases_t **handle, const void *points, size_t npoints,
                                     size_t ffi_affine_sz, int on_device);
/* Same, additionally precomputing `tables` (1, 2, 4, 8 or 16) multiples 2^(256/tables * j) * P_i of every base
 * (one-time cost; tables x 96 B per point of HBM).  An MSM over such a handle needs only 256/(tables*c) bucket
 * windows: the serial Horner tail and the bucket reduction shrink by `tables` while the result stays the same
 * group element. */
RustError snarkvm_hip_register_bases_tables(snarkvm_hip_bases_t **handle, const void *points, size_t npoints,
                                            size_t ffi_affine_sz, int on_device, int tables);
/* General form: table j = 2^(window_bits * j) * P_i for j < tables, with tables * window_bits >= 254 and window_bits <= 23.
 * With window_bits > 16 a large MSM runs ONE bucket window of 2^(window_bits - 1) buckets: only `tables` bucket additions
 * per scalar (12 at window_bits = 22 instead of 16) at the price of a deeper sort and a two-axis bucket fold - the
 * Pippenger optimum for n ~ 2^24.  Smaller MSMs over the same handle fall back to a divisor of window_bits. */
RustError snarkvm_hip_register_bases_windowed(snarkvm_hip_bases_t **handle, const void *points, size_t npoints,
                                              size_t ffi_affine_sz, int on_device, int tables, int window_bits);
void snarkvm_hip_free_bases(snarkvm_hip_bases_t *handle);

/* The reference's canonical encoding of G1 points (curves/src/templates/macros.rs:66-140, utilities/src/serialize/
 * flags.rs:72-99) decoded / encoded on the device: uncompressed = x, y as 48-byte little-endian canonical integers with
 * the infinity flag in bit 6 of the last byte (96 B; the body of a `.usrs` SRS file after its u64 count); compressed =
 * x with bit 7 = "y is the larger root", bit 6 = infinity (48 B; y recovered by a square root, affine.rs:140-150).
 * validate = 1 additionally checks curve and prime-order-subgroup membership (`Valid::check`, macros.rs:106-113).
 * Errors (both flag bits set, coordinate >= q, no square root, failed validation) return a non-zero code and name the
 * SerializationError in the message.
 * register_bases_serialized: bytes (host) -> a registered base vector, never materialising the Rust layout. */
RustError snarkvm_hip_register_bases_serialized(snarkvm_hip_bases_t **handle, const void *bytes, size_t npoints,
                                                int compressed, int validate, int tables);
/* bytes (host) -> Rust `[G1Affine]` (104 B stride, host) and back. */
RustError snarkvm_hip_g1_deserialize(void *out_affine, const void *bytes, size_t n, int compressed, int validate);
RustError snarkvm_hip_g1_serialize(void *out_bytes, const void *affine, size_t n, size_t ffi_affine_sz, int compressed);

/* G2 points, uncompressed (192 B: x.c0, x.c1, y.c0, y.c1 canonical little-endian, flags in the last byte; fp2.rs:425-455),
 * e.g. `beta-h.usrs`: bytes (host) <-> Rust `[G2Affine]` (200 B stride, host). */
RustError snarkvm_hip_g2_deserialize(void *out_affine, const void *bytes, size_t n, int validate);
RustError snarkvm_hip_g2_serialize(void *out_bytes, const void *affine, size_t n, size_t ffi_affine_sz);
/* Compressed G2 points (96 B: x.c0, x.c1 with the flags; y = `Fp2::sqrt` of x^3 + b' selected by the sign flag in the order
 * of fields/src/fp2.rs:240-250, all on the device; fields/src/fp2.rs:208-230, curves/src/templates/.../affine.rs:140-150). */
RustError snarkvm_hip_g2_deserialize_compressed(void *out_affine, const void *bytes, size_t n, int validate);
RustError snarkvm_hip_g2_serialize_compressed(void *out_bytes, const void *affine, size_t n, size_t ffi_affine_sz);

/* MSM over registered bases [offset, offset + npoints).  `scalars` in host (scalars_on_device = 0) or
 * device memory.  `out` is a 144-byte host buffer (Jacobian, as snarkvm_msm).  `window_bits` = 0 picks
 * the window size automatically. */
RustError snarkvm_hip_msm_registered(void *out, const snarkvm_hip_bases_t *handle, size_t offset, size_t npoints,
                                     const void *scalars, int scalars_on_device, int window_bits);

/* KZG10-shaped MSM over registered bases: sum over two base ranges [off0, off0+n0) and [off1, off1+n1) with
 * n0 + n1 consecutive scalars - the plaintext MSM and the hiding MSM of KZG10::commit
 * (polycommit/kzg10/mod.rs:119,149) in one launch.  scalars_montgomery = 1 fuses `convert_to_bigints`
 * (kzg10/mod.rs:469-474: Fr::to_bigint per coefficient) into the scalar-read phase. */
RustError snarkvm_hip_msm_registered_ex(void *out, const snarkvm_hip_bases_t *handle, size_t off0, size_t n0,
                                        size_t off1, size_t n1, const void *scalars, int scalars_on_device,
                                        int scalars_montgomery, int window_bits);

/* A batch of independent MSMs over one registered base vector (the commitments of a batch of proofs,
 * polycommit/sonic_pc/mod.rs:186-245 fans these out over a CPU pool).  Instance k covers bases
 * [offsets[k], offsets[k] + npoints[k]) with the scalar vector scalars[k]; results are written to
 * outs + 144 * k.  Instances are dealt to the selected devices (host scalars: round-robin; device scalars: the device that
 * owns them) and, on each device, pipelined over several HIP streams so that the latency-bound tail of one MSM overlaps
 * the accumulation of the next. */
RustError snarkvm_hip_msm_registered_batch(void *outs, const snarkvm_hip_bases_t *handle, size_t count,
                                           const size_t *offsets, const size_t *npoints, const void *const *scalars,
                                           int scalars_on_device, int scalars_montgomery, int window_bits);

/* The same with two base ranges per instance: bases [off0[k], off0[k] + n0[k]) followed by [off1[k], off1[k] + n1[k]) against
 * n0[k] + n1[k] consecutive scalars - a whole round of KZG10 commitments (plaintext + hiding MSM each, polycommit/kzg10/
 * mod.rs:119,149; degree-bounded ones start at a shifted-powers offset, sonic_pc/mod.rs:203-245) in one call. */
RustError snarkvm_hip_msm_registered_batch_ex(void *outs, const snarkvm_hip_bases_t *handle, size_t count, const size_t *off0,
                                              const size_t *n0, const size_t *off1, const size_t *n1,
                                              const void *const *scalars, int scalars_on_device, int scalars_montgomery,
                                              int window_bits);

/* out (144 B) = sum of n G1Projective points (host buffers).  Combines the per-device partial results of an MSM whose
 * point range was split over several GPUs (the host `dadd` loop of snarkvm.cu:290-295). */
RustError snarkvm_hip_g1_sum(void *out, const void *in_projective, size_t n);

/* `From<Projective> for Affine` (curves/src/templates/short_weierstrass_jacobian/affine.rs:331-353) for n
 * G1Projective (144 B) -> G1Affine (104 B), host buffers. */
RustError snarkvm_hip_g1_to_affine(void *out_affine, const void *in_projective, size_t n);

/* G2 variable-base MSM (the reference routes G2 through its CPU `standard::msm`,
 * msm/variable_base/{mod.rs:45-47,standard.rs:79-105}; north_star asks for it on the device).
 * `points_with_infinity` is a Rust `[G2Affine]` (x, y in Fq2 = 96 B each, infinity flag; stride 200 B);
 * `out` is a G2Projective (Jacobian, 288 B). */
RustError snarkvm_hip_msm_g2(void *out, const void *points_with_infinity, size_t npoints, const void *scalars,
                             size_t ffi_affine_sz);

/* Registered G2 bases (static G2 vectors, e.g. powers of beta H): the same precomputed-table scheme as
 * snarkvm_hip_register_bases_tables / _windowed (window_bits = 0: tables in {1, 2, 4, 8, 16} of 256 / tables bits).  Without
 * tables a G2 MSM ends in a serial chain of ~240 Fq2 doublings (~10 ms whatever its size); with them that chain is gone.
 * `points` is a host Rust `[G2Affine]` (stride >= 200), `out` a G2Projective (288 B). */
typedef struct snarkvm_hip_bases_g2 snarkvm_hip_bases_g2_t;
RustError snarkvm_hip_register_bases_g2(snarkvm_hip_bases_g2_t **handle, const void *points, size_t npoints,
                                        size_t ffi_affine_sz, int tables, int window_bits);
void snarkvm_hip_free_bases_g2(snarkvm_hip_bases_g2_t *handle);
RustError snarkvm_hip_msm_g2_registered(void *out, const snarkvm_hip_bases_g2_t *handle, size_t offset, size_t npoints,
                                        const void *scalars, int scalars_on_device, int window_bits);
/* A batch of independent G2 MSMs over one registered vector, fanned out over devices and lanes like
 * snarkvm_hip_msm_registered_batch; results are written to outs + 288 * k. */
RustError snarkvm_hip_msm_g2_registered_batch(void *outs, const snarkvm_hip_bases_g2_t *handle, size_t count,
                                              const size_t *offsets, const size_t *npoints, const void *const *scalars,
                                              int scalars_on_device, int window_bits);

/* Setup-time group operations.  Host buffers.
 * fixed_base_msm: out[i] (G1Projective, 144 B) = scalars[i] * g for one base `g` (Rust G1Affine, 104 B) and n Fr
 *   scalars in Montgomery form - FixedBase::msm (msm/fixed_base.rs:33-97; the window table is built on the device).
 * group_ntt: in-place radix-2 transform of 2^lg G1Projective points with Fr twiddles; inverse = 1 includes the 1/n
 *   scaling - `EvaluationDomain::ifft` over group elements as used by UniversalParams::lagrange_basis
 *   (polycommit/kzg10/data_structures.rs:68-72).  Results are group elements: compare after to_affine. */
RustError snarkvm_hip_g1_fixed_base_msm(void *out_projective, const void *g_affine, const void *scalars, size_t n);
RustError snarkvm_hip_g1_group_ntt(void *inout_projective, uint32_t lg_domain_size, int inverse);

/* Fr vector helpers on device memory: out[i] = a[i] * b[i] (polynomial_inner_multiply,
 * polynomial.cuh:36-45); Fr::to_bigint / from_bigint over a vector (kzg10/mod.rs:469-474). */
RustError snarkvm_hip_fr_mul_device(void *d_out, const void *d_a, const void *d_b, size_t n);
RustError snarkvm_hip_fr_convert_device(void *d_out, const void *d_in, size_t n, int to_bigint);

/* Prover-round Fr vector kernels: the O(n) passes the Varuna prover runs between its NTTs and commitments
 * (SURVEY.md 8f N2, and the `open` half of KZG10).  Vectors are Fr elements in the reference's memory form
 * (32 B, Montgomery); with on_device = 1 every vector pointer is device memory on the context's device, otherwise host
 * memory (staged through HBM by the call).  Scalar operands (`scalar`, `point`, `coeff`, `g`, `c`, `tau`) and the
 * `remainder` of divide_by_linear are always 32-byte HOST values.
 *
 * fr_vec_op: op 0 out = a + b | 1 a - b | 2 a * b | 3 a * b - c (round_functions/second.rs:110-111: rowcheck)
 *            | 4 a * scalar | 5 a - scalar (kzg10/mod.rs:295) | 6 a + b * scalar (second.rs:113) | 7 scalar - a. */
RustError snarkvm_hip_fr_vec_op(int op, void *out, const void *a, const void *b, const void *c, const void *scalar,
                                size_t n, int on_device);
/* `polynomial / (X - point)` and `polynomial.evaluate(point)` in one pass (KZG10::compute_witness_polynomial,
 * kzg10/mod.rs:213-236 via polynomial/mod.rs:222-256; DensePolynomial::evaluate, dense.rs:98-114): quotient gets
 * n - 1 coefficients (may be NULL when only the value is wanted), *remainder = p(point).  With on_device = 1 `quotient` must not
 * overlap `poly` (coefficient ranges are owned by different workgroups: an in-place division would race); such a call is refused. */
RustError snarkvm_hip_fr_divide_by_linear(void *quotient, void *remainder, const void *poly, size_t n,
                                          const void *point, int on_device);
/* batch_inversion_and_mul (fields/src/lib.rs:66-129): v_i <- coeff / v_i, zero elements stay zero. */
RustError snarkvm_hip_fr_batch_inversion_and_mul(void *inout, size_t n, const void *coeff, int on_device);
/* EvaluationDomain::distribute_powers_and_mul_by_const (fft/domain.rs:224-254): v_i <- v_i * c * g^i. */
RustError snarkvm_hip_fr_distribute_powers(void *inout, size_t n, const void *g, const void *c, int on_device);
/* EvaluationDomain::evaluate_all_lagrange_coefficients (fft/domain.rs:258-292) for the 2^lg domain. */
RustError snarkvm_hip_fr_lagrange_coefficients(void *out, uint32_t lg_domain_size, const void *tau, int on_device);
/* DensePolynomial::divide_by_vanishing_poly (dense.rs:161-169): division of `len` coefficients by X^domain_size - 1.
 * quotient: len - domain_size coefficients (untouched when len <= domain_size); remainder: min(len, domain_size). */
RustError snarkvm_hip_fr_divide_by_vanishing(void *quotient, void *remainder, const void *poly, size_t len,
                                             size_t domain_size, int on_device);
/* DensePolynomial::mul_by_vanishing_poly (dense.rs:153-159): out (len + domain_size coefficients) = p * (X^D - 1). */
RustError snarkvm_hip_fr_mul_by_vanishing(void *out, const void *poly, size_t len, size_t domain_size, int on_device);

/* Strided batches of the passes above on device memory - the same pass over one vector of every proof of a batch proved in lock
 * step (VarunaSNARK::prove_batch, snark/varuna/varuna.rs:336): member y of the batch uses every vector pointer advanced by
 * y * stride elements (stride >= the vector length); ONE kernel launch sequence for the whole batch.  fr_vec_op_strided: `scalar`
 * is shared.  fr_divide_by_linear_strided: one opening `point` for all members (a batch opening at one challenge point,
 * sonic_pc/mod.rs:316-337), `remainders` = count x 32 bytes of HOST memory.  Inside a snarkvm_hip_scope the calls are only
 * enqueued, like their single-vector forms. */
RustError snarkvm_hip_fr_vec_op_strided(int op, void *out, const void *a, const void *b, const void *c, const void *scalar,
                                        size_t n, size_t count, size_t stride);
RustError snarkvm_hip_fr_divide_by_linear_strided(void *quotients, void *remainders, const void *polys, size_t n,
                                                  const void *point, size_t count, size_t stride);
RustError snarkvm_hip_fr_divide_by_vanishing_strided(void *quotients, void *remainders, const void *polys, size_t len,
                                                     size_t domain_size, size_t count, size_t stride);

/* Synthetic base set for benchmarks: out[i] = (start + i) * G as Rust G1Affine (104 B stride) in
 * device memory. */
RustError snarkvm_hip_g1_generate_bases_device(void *d_out, uint64_t start, size_t npoints);

/* Per-phase timing of the most recent profiled MSM / NTT call (synchronous single-lane calls), measured with HIP events on
 * the stream the kernels were launched on.  Enable first; then read `count` (name, milliseconds) pairs. */
void snarkvm_hip_set_profiling(int enabled);
int snarkvm_hip_get_phase_count(void);
const char *snarkvm_hip_get_phase_name(int i);
double snarkvm_hip_get_phase_ms(int i);

/* How the in-library coalescer grouped concurrent callers of proof-sized G1 MSMs so far: out[4] = {batches dispatched, tickets
 * (MSM instances) in them, largest batch, batches of a single ticket}; reset != 0 clears the counters. */
void snarkvm_hip_coalescer_stats(uint64_t *out, int reset);

/* Workspace growth of the library since the last reset: out[5] = {device allocations, device bytes, pinned-host allocations,
 * pinned bytes, microseconds spent inside them}.  Every lane grows its buffers on demand (hipFree / hipMalloc: the one thing here that
 * synchronises the whole device behind the caller's back); a caller that wants an allocation-free timed region warms the same call
 * shape first and can check with this that nothing grew. */
void snarkvm_hip_alloc_stats(uint64_t *out, int reset);

/* Block until all queued work of every device in use has finished. */
RustError snarkvm_hip_synchronize(void);

/* ---------------------------------------------------------------------------------------------
 * Part 3 - test hooks
 * ------------------------------------------------------------------------------------------- */

/* The device arithmetic (ff.hip.h / ec.hip.h) compiled for the host and run on the CPU, so that the
 * limb arithmetic can be checked without a GPU.  field: 0 = Fr, 1 = Fq.  op: 0 add, 1 sub, 2 mul,
 * 3 sqr, 4 inverse, 5 neg, 6 from_bigint, 7 to_bigint.  Operands / results are in the reference's
 * memory form (Montgomery R = 2^256 / 2^384), n elements of 32 / 48 bytes. */
int snarkvm_hip_selftest_field(int field, int op, const void *a, const void *b, void *out, size_t n);
/* op: 0 = out(Jacobian 144 B) = sum_i (xyzz) points[i] * small_scalars[i] via mixed adds and doublings;
 * exercises every exceptional branch of ec.hip.h on the host. */
int snarkvm_hip_selftest_g1_msm_naive(const void *points_with_infinity, size_t npoints, size_t ffi_affine_sz,
                                      const void *scalars, void *out);
/* The MSM planner (window width, windows, digit rows, buckets, segment length) evaluated on the host:
 * out[10] = {c, W, J, digit rows, buckets per window, total buckets, S, S2, L, wide}.  Returns 0 if consistent. */
int snarkvm_hip_selftest_msm_plan(size_t n, int window_bits, int tables, int table_bits, uint32_t *out);
/* The host-side finish of an MSM on its own (no device): out (144 B) = sum_i 2^pos[i] * planes[i] over n G1Projective
 * memory images - the Horner chain that combines the bit-plane sums the device leaves (and the per-device partial results of
 * a split MSM, the reference's host `dadd`, snarkvm.cu:290-295).  Returns 0. */
int snarkvm_hip_selftest_g1_finish(const void *planes_projective, const int32_t *pos, size_t n, void *out);
/* The lazily reduced arithmetic of the accumulate kernel (csrc/ffl.hip.h) against the exact arithmetic, on the host: `iters`
 * chained mixed additions (doublings, cancellations and restarts from infinity included) compared coordinate by coordinate,
 * then the field routines at the edges of their operand ranges.  0 = identical; > 0: first differing step; < 0: field case. */
int snarkvm_hip_selftest_fq_lazy(uint64_t seed, int iters);
/* host only: the tail arithmetic of a G1 MSM (ffl.hip.h::fqz_t under the generic addition / doubling laws) against the exact one */
int snarkvm_hip_selftest_g1_lazy_tail(uint64_t seed, int iters);
/* The lazy Fq2 arithmetic of the G2 accumulate kernel (csrc/ffl2.hip.h) against the exact arithmetic, on the host: `iters` chained
 * mixed additions of +- points[k] (npoints >= 2 Rust G2Affine records on the curve, 200-byte stride), doublings, cancellations
 * and restarts from infinity included, every coordinate compared after every step; products, squares and the raw partial-sum image
 * on the way.  0 = identical; > 0: first differing step; < 0: a field / conversion case. */
int snarkvm_hip_selftest_fq2_lazy(const void *points, size_t npoints, uint64_t seed, int iters);
/* The lane-pair Fq2 arithmetic of the G2 accumulate kernel (csrc/ffl2p.hip.h: the c0 component of every value on the even lane, c1 on
 * the odd lane, operands exchanged inside the VALU) with both lanes of a pair run side by side on the host - the same source - against
 * the exact arithmetic: the chain of snarkvm_hip_selftest_fq2_lazy; doublings and cancellations are resolved inside the pair arithmetic.
 * 0 = identical; > 0: first differing step; < 0: a conversion case. */
int snarkvm_hip_selftest_fq2_pair(const void *points, size_t npoints, uint64_t seed, int iters);
/* The sixteen-lane cooperative Fq2 addition of the G2 tail trees (csrc/hex2.hip.h: lane 4 q + p of a DPP row computes Fq sub-product p of the quad
 * schedule's product q; pair exchange, combine, gather) run over sixteen simulated lanes on the host - the same source - against the exact addition:
 * `iters` additions of partial sums of +- points[k], with operands at infinity, P + P and P - P among them.  0 = every lane of every addition ends
 * with exactly the exact sum; > 0: first differing iteration. */
int snarkvm_hip_selftest_g2_hex(const void *points, size_t npoints, uint64_t seed, int iters);
/* Device check: the G2 fold / bit-plane kernels launched `iters` times over one fixed set of partial-sum lists (2^(m + hb) buckets) must leave
 * the same group elements.  report[10]: [0] fold launches differing from the first, [1] differing fold slots, [2] bit-plane launches differing,
 * [3] differing planes, [4..7] first differing fold slots, [8], [9] fold slots / planes the fast kernels handed to the fix kernels (equal x met). */
RustError snarkvm_hip_devtest_g2_tail_repeat(const void *points, size_t npoints, int m, int hb, int threads, int plane_threads,
                                             int hex, int quads, int iters, uint32_t *report);
/* The signed-limb butterfly arithmetic of the NTT passes (csrc/frs.hip.h) against the exact arithmetic, on the host: passes of up
 * to nine butterfly stages without a canonical form in between, the closing product, the bare reduction and the folded table
 * form.  0 = identical; > 0: first differing butterfly; < 0: a closing-step case. */
int snarkvm_hip_selftest_fr_signed(uint64_t seed, int iters);
/* Same field operations executed by a GPU kernel (one thread per element). */
RustError snarkvm_hip_devtest_field(int field, int op, const void *a, const void *b, void *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* SNARKVM_HIP_H */
