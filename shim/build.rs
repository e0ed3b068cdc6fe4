// libtgx.so is built by `make -C term_amd/csrc` (hipcc, gfx950); TGX_LIB_DIR names the directory it lies in
// (default: ../term_amd next to this crate).
fn main() {
    let dir = std::env::var("TGX_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../term_amd", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tgx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=TGX_LIB_DIR");
}
