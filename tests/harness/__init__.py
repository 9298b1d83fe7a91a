"""Test harness only: Python drivers of protocols that lie OUTSIDE the hot-path scope of SURVEY.md section 8 (HyraxPC commit / open /
check, LinearCodePCS open / check) over the C ABI.  They exist so that the GPU suite can put the library's primitives (pc_hip_msm_many,
pc_hip_fr_lincomb, pc_hip_ligero_commit, ...) through whole protocol runs against the Python restatement; the host layer a prover
would bind is the C++ one (poly_commit_amd/host/*.hpp) or the Rust shim (rust/poly-commit-hip)."""
