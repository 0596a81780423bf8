// OPENINTEL_HIP_DIR = directory holding libopenintel_hip.so (this repository's openintel_amd/).
fn main() {
    println!("cargo:rerun-if-env-changed=OPENINTEL_HIP_DIR");
    let dir = std::env::var("OPENINTEL_HIP_DIR").expect("set OPENINTEL_HIP_DIR to the directory of libopenintel_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=openintel_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
