//! `poly-commit-hip`: the commit/open hot path of `ark-poly-commit` on an MI355X (gfx950) through `libpc_hip.so`.
//!
//! `ark-poly-commit` forbids `unsafe` (`poly-commit/src/lib.rs:11`) and its schemes call
//! `<E::G1 as VariableBaseMSM>::msm_bigint` by name (`kzg10/mod.rs:175,255`; `ipa_pc/mod.rs:64`), so the hook cannot sit
//! below the trait.  This crate therefore implements the reference's own trait
//! [`ark_poly_commit::PolynomialCommitment`] (`poly-commit/src/lib.rs:164-577`) for new scheme TYPES whose associated
//! types are the reference's (`marlin_pc::{UniversalParams, CommitterKey, VerifierKey, Commitment, Randomness}`,
//! `kzg10::Proof`, `ipa_pc::*`): keys, commitments, states and proofs are interchangeable with `MarlinKZG10` /
//! `InnerProductArgPC`, `setup` / `trim` / `check` delegate to the reference, and `commit` / `open` restate its glue
//! line by line with the data-parallel calls swapped:
//!
//! | reference call | here |
//! |---|---|
//! | `msm_bigint(&powers_of_g[lz..], &coeffs)` (`kzg10/mod.rs:175-178`, `:255-258`; `ipa_pc/mod.rs:64`) | [`kzg10_hip::msm`] -> `pc_hip_msm` / `pc_hip_msm_batch` on the resident key |
//! | `skip_leading_zeros_and_convert_to_bigints` (`kzg10/mod.rs:452-470`) | fused: coefficients cross as they lie in memory (`PC_SCALARS_MONTGOMERY`), `base_offset = lz` |
//! | `p / (x - z)` (`kzg10/mod.rs:217-240`) | `pc_hip_witness_poly`, the quotient never leaves HBM |
//! | `p += (challenge_j, polynomial)` (`marlin_pc/mod.rs:281-287`) | `pc_hip_fr_lincomb` over the device copies of the polynomials |
//! | `ck.powers` / `ck.comm_key` | resident in HBM, uploaded the first time a key is seen ([`device::resident`]) |
//! | IPA halving loop (`ipa_pc/mod.rs:664-711`) | `pc_hip_msm_async` x2, `pc_hip_ipa_fold_dots`, `pc_hip_ipa_round2_msms` + `pc_hip_ec_fold2_from` (rounds 1-2), `pc_hip_ec_fold` / `pc_hip_ipa_key_scalars` |
//! | `reed_solomon` (`linear_codes/utils.rs:112-127`) | [`ligero::HipUnivariateLigero`] -> `pc_hip_ntt_batch` |
//!
//! Below [`device::min_pairs`] pairs the shim keeps `ark-ec`'s CPU `msm_bigint` (a launch sequence costs ~1 ms; see
//! `workloads.latency` of the repository's bench line for the measured crossover).
//!
//! Everything validated by the reference before an MSM (`check_degree_is_too_large`, `check_degrees_and_bounds`,
//! `MissingRng`, hiding bounds) is validated here in the same order before any FFI call; RNG draws happen exactly where
//! the reference draws (`kzg10/mod.rs:189`), so commitments and proofs are bit-identical for the same seed.  A non-zero
//! `pc_status` becomes `Error::InvalidParameters(String)` (`error.rs:117`).
//!
//! There is no Rust toolchain in the image this crate was written in: it has not been compiled there.  The C ABI it binds
//! is exercised by the repository's GPU tests through `ctypes` and C++; `tools/check_ffi_decls.py` keeps [`ffi`] in step
//! with `include/pc_hip.h`; `tests/conventions.rs` holds the assertions about arkworks' in-memory layouts and byte
//! conventions that the library restates from memory (they must pass before the backend is trusted).
#![allow(clippy::too_many_arguments, clippy::type_complexity)]

pub mod curve;
pub mod device;
pub mod ffi;
pub mod group;
pub mod ipa_pc;
pub mod kzg10_hip;
pub mod ligero;
pub mod marlin_kzg10;
pub mod sonic_kzg10;

pub use curve::{HipCurve, HipField};
pub use group::HipGroupKey;
pub use ipa_pc::HipIpaPC;
pub use ligero::HipUnivariateLigero;
pub use marlin_kzg10::HipMarlinKZG10;
pub use sonic_kzg10::HipSonicKZG10;
