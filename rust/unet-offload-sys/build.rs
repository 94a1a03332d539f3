// Tells cargo where libunet_offload.so lives.  The library itself is built from HIP sources by
// `python __graft_entry__.py build` (hipcc --offload-arch=gfx950) at the repository root; this script only links.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=UNET_OFFLOAD_LIB_DIR");
    let dir = env::var("UNET_OFFLOAD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        // default: the in-tree build output, two levels above this crate
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../unet_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=unet_offload");
    // the HIP runtime the library itself depends on
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
