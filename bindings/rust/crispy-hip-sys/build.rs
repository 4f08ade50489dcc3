// Tells rustc where libcrispy_hip.so lives.  CRISPY_HIP_LIB_DIR overrides the default, which is the in-tree
// build output of `make -C crispy_amd/csrc` (crispy_amd/libcrispy_hip.so) relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=CRISPY_HIP_LIB_DIR");
    let dir = env::var("CRISPY_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../crispy_amd")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=crispy_hip");
    // the library's own RUNPATH finds libamdhip64; the host binary needs one for libcrispy_hip.so itself
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
