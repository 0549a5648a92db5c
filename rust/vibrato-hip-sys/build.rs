// Links libvibrato_hip.so (built by vibrato_amd/build.py with hipcc --offload-arch=gfx950).
// VIBRATO_HIP_LIB_DIR names the directory that holds it (default: ../../vibrato_amd/lib).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("VIBRATO_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../vibrato_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=vibrato_hip");
    println!("cargo:rerun-if-env-changed=VIBRATO_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/vibrato_hip.h");
}
