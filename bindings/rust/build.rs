//! Tells rustc where libconstriction_amd.so (built by `python -m constriction_amd.build`) and the HIP runtime live.
//!   CONSTRICTION_AMD_LIB_DIR   directory of libconstriction_amd.so (default: ../../constriction_amd/lib of this checkout)
//!   ROCM_PATH                  ROCm installation (default /opt/rocm) for libamdhip64.so
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let lib_dir = env::var("CONSTRICTION_AMD_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("../../constriction_amd/lib"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=constriction_amd");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    println!("cargo:rerun-if-env-changed=CONSTRICTION_AMD_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed=build.rs");
}
