// Links the prebuilt HIP backend instead of compiling device code here (the reference's build.rs drives nvcc,
// algorithms/cuda/build.rs:57-99).  libsnarkvm_hip.so is produced by `python -m snarkvm_amd.build`
// (hipcc --offload-arch=gfx950); point SNARKVM_HIP_LIB_DIR at the directory that holds it.
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=SNARKVM_HIP_LIB_DIR");
    let dir = match env::var_os("SNARKVM_HIP_LIB_DIR") {
        Some(d) => PathBuf::from(d),
        None => {
            // in-tree default: <repo>/snarkvm_amd/lib relative to <repo>/rust/snarkvm-algorithms-hip
            let manifest = PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR"));
            manifest.join("..").join("..").join("snarkvm_amd").join("lib")
        }
    };
    let lib = dir.join("libsnarkvm_hip.so");
    if !lib.exists() {
        panic!("{} not found: build it with `python -m snarkvm_amd.build` or set SNARKVM_HIP_LIB_DIR", lib.display());
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=snarkvm_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
