// UNTESTED (no Rust toolchain in this image).  Links libbrotli_decompressor.so, built by
// `make -C rust-brotli-decompressor_amd` at the repository root (hipcc --offload-arch=gfx950).
use std::env;
use std::path::PathBuf;

fn main() {
    // BROTLI_AMD_LIB_DIR overrides the in-tree location (two levels up from this crate)
    let dir = env::var("BROTLI_AMD_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../rust-brotli-decompressor_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=brotli_decompressor");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BROTLI_AMD_LIB_DIR");
}
