// Link against libzjhip.so.  ZJHIP_LIB_DIR points at the directory holding it
// (in this repository: zune-jpeg_amd/, produced by `make -C zune-jpeg_amd/csrc`).
fn main() {
    if let Ok(dir) = std::env::var("ZJHIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zjhip");
    println!("cargo:rerun-if-env-changed=ZJHIP_LIB_DIR");
}
