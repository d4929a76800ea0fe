// Links libstark_mlwe_hip.so (built by `make -C stark_mlwe_amd/csrc`, hipcc --offload-arch=gfx950).
fn main() {
    let dir = std::env::var("STARK_MLWE_HIP_DIR").expect("set STARK_MLWE_HIP_DIR to the directory holding libstark_mlwe_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=stark_mlwe_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=STARK_MLWE_HIP_DIR");
}
