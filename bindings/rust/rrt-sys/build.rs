// Links librrt.so.  RRT_LIB_DIR = directory holding it (default: ../../../rustraytracer_amd).
fn main() {
    let dir = std::env::var("RRT_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../../rustraytracer_amd", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rrt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RRT_LIB_DIR");
}
