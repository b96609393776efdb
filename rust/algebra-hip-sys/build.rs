// build.rs -- tells rustc where libginger_hip.so lives.
// GINGER_HIP_LIB_DIR = the directory holding libginger_hip.so (ginger-lib_amd/ in the HIP repository, or wherever
// the library was installed).  The library itself is built by `python __graft_entry__.py` (hipcc, gfx950).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=GINGER_HIP_LIB_DIR");
    match env::var("GINGER_HIP_LIB_DIR") {
        Ok(dir) => {
            println!("cargo:rustc-link-search=native={}", dir);
            // run-time lookup without LD_LIBRARY_PATH
            println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
        }
        Err(_) => println!("cargo:warning=GINGER_HIP_LIB_DIR is not set: libginger_hip.so must be on the linker's default path"),
    }
    println!("cargo:rustc-link-lib=dylib=ginger_hip");
}
