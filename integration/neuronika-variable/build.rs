// Links the HIP backend when the `hip` feature is on.  NEURONIKA_HIP_LIB_DIR = directory holding libneuronika_hip.so
// (built by `python -m neuronika_amd.build` from this repository: hipcc --offload-arch=gfx950).
fn main() {
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("NEURONIKA_HIP_LIB_DIR").expect("set NEURONIKA_HIP_LIB_DIR to the directory of libneuronika_hip.so");
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=neuronika_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
        println!("cargo:rerun-if-env-changed=NEURONIKA_HIP_LIB_DIR");
    }
}
