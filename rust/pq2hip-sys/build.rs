// Links libpq2hip.so.  PQ2HIP_LIB_DIR names the directory that holds it (parquet2_amd/lib of the build tree).
fn main() {
    if let Ok(dir) = std::env::var("PQ2HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=pq2hip");
    println!("cargo:rerun-if-env-changed=PQ2HIP_LIB_DIR");
}
