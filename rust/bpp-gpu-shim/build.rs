// Links libbpp_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"` in the engine repository; it pulls in
// libamdhip64).  BPP_HIP_LIB_DIR = the directory that holds it (bulletproofs-plus_amd/).
fn main() {
    let dir = std::env::var("BPP_HIP_LIB_DIR").expect("set BPP_HIP_LIB_DIR to the directory that holds libbpp_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bpp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=BPP_HIP_LIB_DIR");
}
