// Link libpc_hip.so.  PC_HIP_LIB_DIR points at the directory that holds it (in this repository: poly_commit_amd/, built by
// `python -m poly_commit_amd.build`); the HIP runtime comes from ROCM_PATH (default /opt/rocm).
fn main() {
    let dir = std::env::var("PC_HIP_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../poly_commit_amd", manifest)
    });
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=pc_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    println!("cargo:rerun-if-env-changed=PC_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
