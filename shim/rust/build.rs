// friedrich/build.rs with the `friedrich_mi355x` feature: plain link lines, no build dependency.
//   FRIEDRICH_AMD_LIB_DIR = directory of libfriedrich_amd.so (python -m friedrich_amd.build puts it in friedrich_amd/lib/)
fn main()
{
    if std::env::var("CARGO_FEATURE_FRIEDRICH_MI355X").is_ok()
    {
        let dir = std::env::var("FRIEDRICH_AMD_LIB_DIR").expect("set FRIEDRICH_AMD_LIB_DIR to the directory of libfriedrich_amd.so");
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=friedrich_amd");
        // the library carries no DT_NEEDED on the HIP runtime: the host process provides the one runtime of the process
        println!("cargo:rustc-link-search=native=/opt/rocm/lib");
        println!("cargo:rustc-link-lib=dylib=amdhip64");
        println!("cargo:rerun-if-env-changed=FRIEDRICH_AMD_LIB_DIR");
    }
}
