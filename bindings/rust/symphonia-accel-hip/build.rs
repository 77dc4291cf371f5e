fn main() {
    if let Ok(dir) = std::env::var("SYMACCEL_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=symaccel");
    println!("cargo:rerun-if-env-changed=SYMACCEL_LIB_DIR");
}
