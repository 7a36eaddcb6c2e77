// Links libzkgpu.so.  ZKGPU_LIB_DIR = the directory holding it (…/eigen-zkvm_amd after `make -C eigen-zkvm_amd/csrc`);
// the pattern follows recursion-gnark/ffi/build.rs:43-45 (a native library located from an environment variable), with an
// rpath so that the binary finds the .so at run time.
fn main() {
    let dir = std::env::var("ZKGPU_LIB_DIR").expect("set ZKGPU_LIB_DIR to the directory that holds libzkgpu.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZKGPU_LIB_DIR");
}
