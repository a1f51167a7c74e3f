// `tari_bulletproofs_plus = { path = "reference" }`: make ./reference a symlink to $BPP_REFERENCE before cargo resolves it.
// (cargo evaluates path dependencies before build scripts run, so create the link by hand the first time:
//   ln -s "$BPP_REFERENCE" rust/ref-dump/reference )
fn main() {
    println!("cargo:rerun-if-env-changed=BPP_REFERENCE");
    if !std::path::Path::new("reference/Cargo.toml").exists() {
        panic!("rust/ref-dump/reference must be (a symlink to) a checkout of tari-project/bulletproofs-plus v0.4.1");
    }
}
