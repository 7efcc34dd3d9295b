// Tells rustc where libmi355fft.so lives.  MI355FFT_LIB_DIR = the directory that holds it (in this repository:
// rustfft_amd/lib, produced by `make -C rustfft_amd/csrc`); an rpath is added so test binaries find it without
// LD_LIBRARY_PATH.
fn main() {
    println!("cargo:rerun-if-env-changed=MI355FFT_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_LINK").is_none() {
        return;
    }
    let dir = std::env::var("MI355FFT_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../rustfft_amd/lib", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=mi355fft");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
