//! `snarkvm-algorithms-hip`: the MI355X backend behind the interface of `snarkvm-algorithms-cuda`.
//!
//! Part 1 re-creates the public surface of the reference crate (algorithms/cuda/src/lib.rs:22-168): the enums
//! `NTTInputOutputOrder` / `NTTDirection` / `NTTType`, and `NTT`, `polymul`, `msm` with the same generic parameters, panics and
//! `Result<_, Error>` behaviour, so `snarkvm-algorithms` compiles against it unchanged.  Part 2 (`resident`) binds the extension
//! ABI of include/snarkvm_hip.h: SRS vectors registered once in HBM, fused KZG10 commitments, batches.
//!
//! UNVERIFIED SOURCE: written without a Rust toolchain at hand.
#![allow(non_snake_case)]

use core::ffi::{c_char, c_void};
use std::fmt;

/// What the C side returns (`RustError` in include/snarkvm_hip.h; layout of sppark's `cuda::Error`, which the reference pulls in
/// through `sppark::cuda_error!()`, algorithms/cuda/src/lib.rs:20): `code == 0` is success, `message` is NULL or a C string
/// allocated with malloc() that the receiver frees.
#[repr(C)]
pub struct Error {
    pub code: i32,
    message: *mut c_char,
}

impl Error {
    pub fn is_ok(&self) -> bool {
        self.code == 0
    }

    pub fn message(&self) -> String {
        if self.message.is_null() {
            String::new()
        } else {
            unsafe { std::ffi::CStr::from_ptr(self.message) }.to_string_lossy().into_owned()
        }
    }

    fn into_result(self) -> Result<(), Error> {
        if self.is_ok() { Ok(()) } else { Err(self) }
    }
}

impl Drop for Error {
    fn drop(&mut self) {
        if !self.message.is_null() {
            unsafe { libc::free(self.message as *mut c_void) };
            self.message = core::ptr::null_mut();
        }
    }
}

impl fmt::Debug for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "snarkvm_hip error {}: {}", self.code, self.message())
    }
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        fmt::Debug::fmt(self, f)
    }
}

impl std::error::Error for Error {}

// the error only carries an owned C string
unsafe impl Send for Error {}
unsafe impl Sync for Error {}

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum NTTInputOutputOrder {
    NN = 0,
    NR = 1,
    RN = 2,
    RR = 3,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum NTTDirection {
    Forward = 0,
    Inverse = 1,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum NTTType {
    Standard = 0,
    Coset = 1,
}

/// The raw C ABI (include/snarkvm_hip.h).  Public for callers that keep their operands in device memory: the safe wrappers below take host slices.
pub mod sys {
    use super::{Error, NTTDirection, NTTInputOutputOrder, NTTType};
    use core::ffi::c_void;

    extern "C" {
        // ---- the reference's three symbols (include/snarkvm_hip.h part 1)
        pub fn snarkvm_ntt(inout: *mut c_void, lg_domain_size: u32, order: NTTInputOutputOrder, direction: NTTDirection, kind: NTTType) -> Error;
        pub fn snarkvm_polymul(
            out: *mut c_void,
            pcount: usize,
            polynomials: *const c_void,
            plens: *const c_void,
            ecount: usize,
            evaluations: *const c_void,
            elens: *const c_void,
            lg_domain_size: u32,
        ) -> Error;
        pub fn snarkvm_msm(out: *mut c_void, points_with_infinity: *const c_void, npoints: usize, scalars: *const c_void, ffi_affine_sz: usize) -> Error;

        // ---- extension ABI (part 2)
        pub fn snarkvm_hip_device_count() -> i32;
        pub fn snarkvm_hip_num_devices() -> i32;
        pub fn snarkvm_hip_set_devices(ids: *const i32, n: usize) -> Error;
        pub fn snarkvm_hip_register_bases_tables(handle: *mut *mut c_void, points: *const c_void, npoints: usize, ffi_affine_sz: usize, on_device: i32, tables: i32) -> Error;
        pub fn snarkvm_hip_register_bases_windowed(
            handle: *mut *mut c_void,
            points: *const c_void,
            npoints: usize,
            ffi_affine_sz: usize,
            on_device: i32,
            tables: i32,
            window_bits: i32,
        ) -> Error;
        pub fn snarkvm_hip_free_bases(handle: *mut c_void);
        pub fn snarkvm_hip_scope_begin(d_any: *const c_void) -> Error;
        pub fn snarkvm_hip_scope_begin_ex(d_any: *const c_void, flags: u32) -> Error;
        pub fn snarkvm_hip_scope_collect(out: *const c_void) -> Error;
        pub fn snarkvm_hip_scope_set_flags(flags: u32) -> Error;
        pub fn snarkvm_hip_scope_end() -> Error;
        pub fn snarkvm_hip_scope_stream() -> *mut c_void;
        pub fn snarkvm_hip_alloc_stats(out: *mut u64, reset: i32);
        pub fn snarkvm_hip_coalescer_stats(out: *mut u64, reset: i32);
        pub fn snarkvm_hip_msm_registered_ex(
            out: *mut c_void,
            handle: *const c_void,
            off0: usize,
            n0: usize,
            off1: usize,
            n1: usize,
            scalars: *const c_void,
            scalars_on_device: i32,
            scalars_montgomery: i32,
            window_bits: i32,
        ) -> Error;
        pub fn snarkvm_hip_msm_registered_batch_ex(
            outs: *mut c_void,
            handle: *const c_void,
            count: usize,
            off0: *const usize,
            n0: *const usize,
            off1: *const usize,
            n1: *const usize,
            scalars: *const *const c_void,
            scalars_on_device: i32,
            scalars_montgomery: i32,
            window_bits: i32,
        ) -> Error;
    }
}

fn log2_exact(domain_size: usize) -> u32 {
    if !domain_size.is_power_of_two() {
        panic!("domain_size is not power of 2"); // the reference panics with this message (lib.rs:84-86, 107-109)
    }
    domain_size.trailing_zeros()
}

/// In-place NTT of `domain_size` field elements (algorithms/cuda/src/lib.rs:77-97).
pub fn NTT<T>(domain_size: usize, inout: &mut [T], ntt_order: NTTInputOutputOrder, ntt_direction: NTTDirection, ntt_type: NTTType) -> Result<(), Error> {
    let lg = log2_exact(domain_size);
    assert!(inout.len() >= domain_size, "buffer shorter than the domain");
    unsafe { sys::snarkvm_ntt(inout.as_mut_ptr() as *mut c_void, lg, ntt_order, ntt_direction, ntt_type) }.into_result()
}

/// Product of coefficient-form polynomials and evaluation-form vectors over a `domain`-sized domain
/// (algorithms/cuda/src/lib.rs:100-145; PolyMultiplier::multiply, fft/polynomial/multiplier.rs:70-134).
pub fn polymul<T: Clone>(domain: usize, polynomials: &Vec<Vec<T>>, evaluations: &Vec<Vec<T>>, zero: &T) -> Result<Vec<T>, Error> {
    let lg = log2_exact(domain);
    let poly_ptrs: Vec<*const T> = polynomials.iter().map(|p| p.as_ptr()).collect();
    let poly_lens: Vec<usize> = polynomials.iter().map(|p| p.len()).collect();
    let eval_ptrs: Vec<*const T> = evaluations.iter().map(|e| e.as_ptr()).collect();
    let eval_lens: Vec<usize> = evaluations.iter().map(|e| e.len()).collect();
    let mut out = vec![zero.clone(); domain];
    unsafe {
        sys::snarkvm_polymul(
            out.as_mut_ptr() as *mut c_void,
            poly_ptrs.len(),
            poly_ptrs.as_ptr() as *const c_void,
            poly_lens.as_ptr() as *const c_void,
            eval_ptrs.len(),
            eval_ptrs.as_ptr() as *const c_void,
            eval_lens.as_ptr() as *const c_void,
            lg,
        )
    }
    .into_result()?;
    Ok(out)
}

/// sum_i scalars[i] * points[i] (algorithms/cuda/src/lib.rs:148-168).  `Affine` is the Rust in-memory affine point
/// (x, y, infinity; 104 bytes for BLS12-377 G1), `Scalar` a canonical `BigInteger256`.
pub fn msm<Affine, Projective, Scalar>(points: &[Affine], scalars: &[Scalar]) -> Result<Projective, Error> {
    let npoints = scalars.len();
    if npoints > points.len() {
        panic!("length mismatch {} points < {} scalars", points.len(), npoints);
    }
    let mut ret = core::mem::MaybeUninit::<Projective>::uninit();
    unsafe {
        sys::snarkvm_msm(ret.as_mut_ptr() as *mut c_void, points.as_ptr() as *const c_void, npoints, scalars.as_ptr() as *const c_void, core::mem::size_of::<Affine>())
            .into_result()?;
        Ok(ret.assume_init())
    }
}

/// Visible HIP devices / logical devices the backend uses (every visible one unless `set_devices` was called first).
pub fn device_count() -> usize {
    unsafe { sys::snarkvm_hip_device_count() }.max(0) as usize
}
pub fn num_devices() -> usize {
    unsafe { sys::snarkvm_hip_num_devices() }.max(0) as usize
}
/// Choose the devices (before the first compute call).
pub fn set_devices(ids: &[i32]) -> Result<(), Error> {
    unsafe { sys::snarkvm_hip_set_devices(ids.as_ptr(), ids.len()) }.into_result()
}

/// Extension: base vectors that stay in HBM (the reference re-uploads its SRS slice on every MSM, snarkvm.cu:262-275).
pub mod resident {
    use super::{sys, Error};
    use core::ffi::c_void;
    use core::marker::PhantomData;

    /// A registered vector of `Affine` points, replicated on every device in use, with `tables` precomputed multiples per point.
    pub struct Bases<Affine> {
        handle: *mut c_void,
        len: usize,
        _marker: PhantomData<Affine>,
    }
    unsafe impl<A> Send for Bases<A> {}
    unsafe impl<A> Sync for Bases<A> {}

    impl<Affine> Bases<Affine> {
        /// `tables` in {1, 2, 4, 8, 16}: table j holds 2^(256 / tables * j) * P.
        pub fn register(points: &[Affine], tables: i32) -> Result<Self, Error> {
            let mut handle = core::ptr::null_mut();
            unsafe { sys::snarkvm_hip_register_bases_tables(&mut handle, points.as_ptr() as *const c_void, points.len(), core::mem::size_of::<Affine>(), 0, tables) }
                .into_result()?;
            Ok(Self { handle, len: points.len(), _marker: PhantomData })
        }

        /// General geometry: table j = 2^(window_bits * j) * P with 254 <= tables * window_bits <= 288 (12 x 22 bits for n ~ 2^24).
        pub fn register_windowed(points: &[Affine], tables: i32, window_bits: i32) -> Result<Self, Error> {
            let mut handle = core::ptr::null_mut();
            unsafe {
                sys::snarkvm_hip_register_bases_windowed(&mut handle, points.as_ptr() as *const c_void, points.len(), core::mem::size_of::<Affine>(), 0, tables, window_bits)
            }
            .into_result()?;
            Ok(Self { handle, len: points.len(), _marker: PhantomData })
        }

        pub fn len(&self) -> usize {
            self.len
        }

        /// Concurrent callers (rayon workers, one commitment each: sonic_pc/mod.rs:186-245) of <= 2^18 pairs over windowed tables are fused
        /// inside the library (runtime.hip.h::msm_coalesced): no change on this side.
        /// One KZG10 commitment (polycommit/kzg10/mod.rs:98-156): `coeffs[..n0]` against bases `[off0, off0 + n0)` plus
        /// `coeffs[n0..]` against `[off1, off1 + n1)` (the hiding MSM).  `coeffs` are `Fr` elements in Montgomery form when
        /// `montgomery` is set - `convert_to_bigints` (kzg10/mod.rs:469-474) is then fused into the device's scalar read.
        pub fn commit<Fr, Projective>(&self, off0: usize, n0: usize, off1: usize, n1: usize, coeffs: &[Fr], montgomery: bool) -> Result<Projective, Error> {
            assert!(coeffs.len() >= n0 + n1 && off0 + n0 <= self.len && off1 + n1 <= self.len);
            let mut ret = core::mem::MaybeUninit::<Projective>::uninit();
            unsafe {
                sys::snarkvm_hip_msm_registered_ex(ret.as_mut_ptr() as *mut c_void, self.handle, off0, n0, off1, n1, coeffs.as_ptr() as *const c_void, 0, montgomery as i32, 0)
                    .into_result()?;
                Ok(ret.assume_init())
            }
        }

        /// All commitments of one `SonicKZG10::commit` call (sonic_pc/mod.rs:177-257) in one device launch: instance k =
        /// (off0[k], n0[k], off1[k], n1[k], coeffs[k]).
        pub fn commit_batch<Fr, Projective: Clone>(&self, ranges: &[(usize, usize, usize, usize)], coeffs: &[&[Fr]], montgomery: bool, zero: &Projective) -> Result<Vec<Projective>, Error> {
            assert_eq!(ranges.len(), coeffs.len());
            for (r, c) in ranges.iter().zip(coeffs.iter()) {
                // the same bounds `commit` asserts, per instance: a short slice would become an out-of-bounds host read
                assert!(c.len() >= r.1 + r.3 && r.0 + r.1 <= self.len && r.2 + r.3 <= self.len);
            }
            let off0: Vec<usize> = ranges.iter().map(|r| r.0).collect();
            let n0: Vec<usize> = ranges.iter().map(|r| r.1).collect();
            let off1: Vec<usize> = ranges.iter().map(|r| r.2).collect();
            let n1: Vec<usize> = ranges.iter().map(|r| r.3).collect();
            let ptrs: Vec<*const c_void> = coeffs.iter().map(|c| c.as_ptr() as *const c_void).collect();
            let mut outs = vec![zero.clone(); ranges.len()];
            unsafe {
                sys::snarkvm_hip_msm_registered_batch_ex(
                    outs.as_mut_ptr() as *mut c_void,
                    self.handle,
                    ranges.len(),
                    off0.as_ptr(),
                    n0.as_ptr(),
                    off1.as_ptr(),
                    n1.as_ptr(),
                    ptrs.as_ptr(),
                    0,
                    montgomery as i32,
                    0,
                )
            }
            .into_result()?;
            Ok(outs)
        }
    }

    /// Deferred synchronisation for device-resident operands (include/snarkvm_hip.h: snarkvm_hip_scope_begin / _end): while the guard
    /// lives, this thread's calls on device vectors are only enqueued; dropping it waits once.  One guard per thread at a time.
    /// The C scope belongs to the thread that began it (`snarkvm_hip_scope_end` on another thread is a no-op there and would leave the
    /// first thread's stream bound forever): the guard is neither `Send` nor `Sync`.
    pub struct Scope(core::marker::PhantomData<*mut ()>);
    /// `Scope::begin_with`: MSMs over registered bases with device-resident scalars are enqueued too; their outputs are written when the
    /// scope ends (include/snarkvm_hip.h: SNARKVM_HIP_SCOPE_ASYNC_MSM).  The output buffers must outlive the guard.
    pub const SCOPE_ASYNC_MSM: u32 = 1;
    /// With `SCOPE_ASYNC_MSM`: the caller leaves the scalar vectors of its enqueued MSMs untouched until the guard is dropped; the scope's
    /// stream then never waits for an MSM (SNARKVM_HIP_SCOPE_STABLE_INPUTS).
    pub const SCOPE_STABLE_INPUTS: u32 = 2;
    /// With `SCOPE_ASYNC_MSM`: MSMs are enqueued on the scope's own stream, in order with its transforms (no hand-off between streams) -
    /// for commitments that are collected before anything else is issued (SNARKVM_HIP_SCOPE_MSM_IN_STREAM).
    pub const SCOPE_MSM_IN_STREAM: u32 = 4;
    impl Scope {
        pub fn begin(device_ptr: *const c_void) -> Result<Self, Error> {
            unsafe { sys::snarkvm_hip_scope_begin(device_ptr) }.into_result()?;
            Ok(Scope(core::marker::PhantomData))
        }
        pub fn begin_with(device_ptr: *const c_void, flags: u32) -> Result<Self, Error> {
            unsafe { sys::snarkvm_hip_scope_begin_ex(device_ptr, flags) }.into_result()?;
            Ok(Scope(core::marker::PhantomData))
        }
        /// Waits for the MSM call that was given `out` as its first output (null: every MSM enqueued so far) and writes its outputs - a round's
        /// commitments for the transcript; the scope stays open and the other enqueued MSMs stay pending.
        pub fn collect(&self, out: *const c_void) -> Result<(), Error> {
            unsafe { sys::snarkvm_hip_scope_collect(out) }.into_result()
        }
        /// Changes the scope's flags for the calls that follow (what is enqueued stays where it is): a prover issues its independent MSM
        /// on a further stream and then switches the transcript-ordered commitment rounds to `SCOPE_MSM_IN_STREAM`.
        pub fn set_flags(&self, flags: u32) -> Result<(), Error> {
            unsafe { sys::snarkvm_hip_scope_set_flags(flags) }.into_result()
        }
        /// The `hipStream_t` the scope's calls are enqueued on (for the caller's own copies / kernels that feed them).
        pub fn stream(&self) -> *mut c_void {
            unsafe { sys::snarkvm_hip_scope_stream() }
        }
        /// Ends the scope and reports an error of the queued work (dropping the guard ignores it).
        pub fn end(self) -> Result<(), Error> {
            core::mem::forget(self);
            unsafe { sys::snarkvm_hip_scope_end() }.into_result()
        }
    }
    impl Drop for Scope {
        fn drop(&mut self) {
            let _ = unsafe { sys::snarkvm_hip_scope_end() };
        }
    }

    /// Workspace growth since the last reset: (device allocations, device bytes, pinned allocations, pinned bytes, microseconds inside them).
    pub fn alloc_stats(reset: bool) -> (u64, u64, u64, u64, u64) {
        let mut v = [0u64; 5];
        unsafe { sys::snarkvm_hip_alloc_stats(v.as_mut_ptr(), reset as i32) };
        (v[0], v[1], v[2], v[3], v[4])
    }

    /// How the in-library coalescer grouped concurrent callers of proof-sized MSMs: (batches, instances, largest batch, single-instance batches).
    pub fn coalescer_stats(reset: bool) -> (u64, u64, u64, u64) {
        let mut v = [0u64; 4];
        unsafe { sys::snarkvm_hip_coalescer_stats(v.as_mut_ptr(), reset as i32) };
        (v[0], v[1], v[2], v[3])
    }

    impl<Affine> Drop for Bases<Affine> {
        fn drop(&mut self) {
            unsafe { sys::snarkvm_hip_free_bases(self.handle) };
        }
    }
}
