// MINISTARK_LIB_DIR = the directory that holds libministark.so (this repository: mini-stark_amd/)
fn main() {
    let dir = std::env::var("MINISTARK_LIB_DIR").expect("set MINISTARK_LIB_DIR to the directory of libministark.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ministark");
    println!("cargo:rerun-if-env-changed=MINISTARK_LIB_DIR");
}
