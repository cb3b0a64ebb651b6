// Tells rustc where liboxmpl_hip.so lives.  OXMPL_HIP_LIB_DIR defaults to the in-tree build output
// (oxmpl_amd/lib, produced by `make -C oxmpl_amd/csrc` or `__graft_entry__.build()`).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("OXMPL_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../oxmpl_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=oxmpl_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=OXMPL_HIP_LIB_DIR");
}
