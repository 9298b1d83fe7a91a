//! The byte-level and layout conventions of arkworks 0.5 that `libpc_hip.so` restates from the crates' published behaviour
//! (they live in ark-ff / ark-ec / ark-serialize / ark-poly / ark-crypto-primitives, none of which is under the reference
//! checkout the library was written against).  Every assertion here compares the REAL crates with what the library or this
//! crate assumes; a failure is a convention difference (never an arithmetic one: affine points and canonical residues are
//! unique) and must be fixed before the backend is trusted.  Needs a GPU for the device-side halves.
//!
//! `ROOT_*`: generated from poly_commit_amd/csrc/field_constants.h (the NTT's omega); tools/check_ffi_decls.py re-checks them.
use ark_ec::{AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{FftField, Field, PrimeField, UniformRand};
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial, EvaluationDomain, GeneralEvaluationDomain};
use ark_serialize::CanonicalSerialize;
use ark_std::test_rng;
use core::ffi::c_void;
use poly_commit_hip::{curve::{HipCurve, HipField}, device, ffi};

// TWO_ADIC_ROOT_OF_UNITY in Montgomery limbs, as the library's twiddle generator holds it
const ROOT_BLS12_381_FR: [u64; 4] = [0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec, 0x0af53ae352a31e64, 0x5bf3adda19e9b27b];
const ROOT_BN254_FR: [u64; 4] = [0x636e735580d13d9c, 0xa22bf3742445ffd6, 0x56452ac01eb203d8, 0x1860ef942963f9e7];
const ROOT_PALLAS_FR: [u64; 4] = [0x218077428c9942de, 0xcc49578921b60494, 0xac2e5d27b2efbee2, 0x0b79fa897f2db056];

fn root_matches<F: HipField + FftField>(limbs: [u64; 4]) {
    assert_eq!(F::TWO_ADIC_ROOT_OF_UNITY.to_mont_limbs(), limbs, "TWO_ADIC_ROOT_OF_UNITY differs from the library's constant");
}

#[test]
fn two_adic_roots_of_unity() {
    root_matches::<ark_bls12_381::Fr>(ROOT_BLS12_381_FR);
    root_matches::<ark_bn254::Fr>(ROOT_BN254_FR);
    root_matches::<ark_pallas::Fr>(ROOT_PALLAS_FR);
}

#[test]
fn in_memory_layouts() {
    // the zero-copy path is only taken when these hold; they are expected to hold on x86-64 / aarch64 with rustc's current
    // layout of `Affine {x, y, infinity}` -- if they do not, the crate repacks (slower, still correct)
    assert!(<ark_bls12_381::Fr as HipField>::layout_is_abi());
    assert!(<ark_bn254::Fr as HipField>::layout_is_abi());
    assert!(<ark_pallas::Fr as HipField>::layout_is_abi());
    assert_eq!(core::mem::size_of::<ark_bls12_381::G1Affine>(), 104);
    assert_eq!(core::mem::size_of::<ark_bn254::G1Affine>(), 72);
    assert_eq!(core::mem::size_of::<ark_pallas::Affine>(), 72);
    assert!(<ark_bls12_381::G1Affine as HipCurve>::layout_is_abi());
    assert!(<ark_bn254::G1Affine as HipCurve>::layout_is_abi());
    assert!(<ark_pallas::Affine as HipCurve>::layout_is_abi());
}

fn msm_matches<G>(n: usize)
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
{
    let rng = &mut test_rng();
    let bases: Vec<G> = (0..n).map(|_| G::Group::rand(rng).into_affine()).collect();
    let scalars: Vec<G::ScalarField> = (0..n).map(|_| G::ScalarField::rand(rng)).collect();
    let want = <G::Group as VariableBaseMSM>::msm_bigint(&bases, &scalars.iter().map(|s| s.into_bigint()).collect::<Vec<_>>());
    std::env::set_var("PC_HIP_MIN_PAIRS", "1");
    let got = poly_commit_hip::kzg10_hip::msm::<G>(&bases, poly_commit_hip::kzg10_hip::Scalars::Host(&scalars)).unwrap();
    assert_eq!(got.into_affine(), want.into_affine());
}

#[test]
fn msm_equals_ark_ec() {
    msm_matches::<ark_bls12_381::G1Affine>(5000);
    msm_matches::<ark_bn254::G1Affine>(5000);
    msm_matches::<ark_pallas::Affine>(5000);
}

fn ntt_matches<F: HipField + FftField>(m: usize, rho_inv: usize) {
    let rng = &mut test_rng();
    let msg: Vec<F> = (0..m).map(|_| F::rand(rng)).collect();
    let want = GeneralEvaluationDomain::<F>::new(m * rho_inv).unwrap().fft(&msg);          // reed_solomon, linear_codes/utils.rs:112-127
    let got = poly_commit_hip::ligero::encode_matrix(&msg, 1, m, rho_inv).unwrap();
    assert_eq!(got, want);
}

#[test]
fn ntt_equals_ark_poly_fft() {
    ntt_matches::<ark_bls12_381::Fr>(1 << 10, 4);
    ntt_matches::<ark_bls12_381::Fr>(300, 4);          // zero padding to the next power of two
    ntt_matches::<ark_bn254::Fr>(1 << 12, 2);
    ntt_matches::<ark_pallas::Fr>(1 << 9, 4);
}

/// `serialize_uncompressed` / `serialize_compressed` of G1 points (both roots, infinity) through the library's decoder:
/// `pc_hip_srs_load_serialized` must give back exactly the points ark-serialize wrote (the SWFlags convention -- 0x80 =
/// YIsNegative = y > -y -- and ark-bls12-381's zcash encoding are what is being pinned).
fn serialized_points_round_trip<G>(compressed: bool)
where
    G: HipCurve + CanonicalSerialize,
{
    let rng = &mut test_rng();
    let mut pts: Vec<G> = (0..64).map(|_| G::Group::rand(rng).into_affine()).collect();
    let negs: Vec<G> = pts.iter().map(|p| (-p.into_group()).into_affine()).collect();
    pts.extend(negs);
    pts.push(G::zero());
    let mut bytes = Vec::new();
    if compressed { pts.serialize_compressed(&mut bytes).unwrap() } else { pts.serialize_uncompressed(&mut bytes).unwrap() }
    let c = device::ctx().unwrap();
    let (mut srs, mut n, mut used) = (core::ptr::null_mut(), 0usize, 0usize);
    let rc = unsafe { ffi::pc_hip_srs_load_serialized(c.raw, G::CURVE, bytes.as_ptr() as *const c_void, bytes.len(), compressed as i32, 0, &mut srs, &mut n, &mut used) };
    assert_eq!(rc, ffi::PC_OK);
    assert_eq!((n, used), (pts.len(), bytes.len()));
    let w = 2 * G::FQ_LIMBS;
    let mut back = vec![0u64; n * w];
    assert_eq!(unsafe { ffi::pc_hip_srs_read(c.raw, srs, 0, n, back.as_mut_ptr() as *mut c_void) }, ffi::PC_OK);
    for (i, p) in pts.iter().enumerate() {
        assert_eq!(G::read_xy(&back[i * w..(i + 1) * w]), *p, "point {} ({})", i, if compressed { "compressed" } else { "uncompressed" });
    }
    unsafe { ffi::pc_hip_srs_free(srs) };
}

#[test]
fn ark_serialize_point_encodings() {
    for compressed in [false, true] {
        serialized_points_round_trip::<ark_bls12_381::G1Affine>(compressed);
        serialized_points_round_trip::<ark_bn254::G1Affine>(compressed);
        serialized_points_round_trip::<ark_pallas::Affine>(compressed);
    }
    // the advisor's vector: BN254's generator (1, 2) has y <= -y: YIsPositive, no flag bit
    let mut b = Vec::new();
    ark_bn254::G1Affine::generator().serialize_compressed(&mut b).unwrap();
    assert_eq!(b[0], 1);
    assert!(b[1..].iter().all(|x| *x == 0));
}

/// Merkle conventions (`ByteDigestConverter`, two-to-one order, heap order of the inner nodes): one `MerkleTree::new` root
/// against `pc_hip_merkle_tree` for the Config the reference's Ligero tests instantiate
/// (linear_codes/univariate_ligero/tests.rs:21-37).
#[test]
fn merkle_root_matches_ark_crypto_primitives() {
    use ark_crypto_primitives::{crh::{sha256::Sha256, CRHScheme, TwoToOneCRHScheme}, merkle_tree::{ByteDigestConverter, Config, MerkleTree}};
    use ark_pcs_bench_templates::LeafIdentityHasher;
    struct P;
    impl Config for P {
        type Leaf = Vec<u8>;
        type LeafDigest = <LeafIdentityHasher as CRHScheme>::Output;
        type LeafInnerDigestConverter = ByteDigestConverter<Self::LeafDigest>;
        type InnerDigest = <Sha256 as TwoToOneCRHScheme>::Output;
        type LeafHash = LeafIdentityHasher;
        type TwoToOneHash = Sha256;
    }
    let leaves: Vec<Vec<u8>> = (0..64u8).map(|i| (0..32).map(|j| i.wrapping_mul(31).wrapping_add(j)).collect()).collect();
    let tree = MerkleTree::<P>::new(&(), &(), leaves.iter().map(|l| l.as_slice())).unwrap();
    let flat: Vec<u8> = leaves.iter().flatten().copied().collect();
    let mut nodes = vec![0u8; 63 * 32];
    let c = device::ctx().unwrap();
    assert_eq!(unsafe { ffi::pc_hip_merkle_tree(c.raw, ffi::PC_HASH_SHA256, flat.as_ptr() as *const c_void, ffi::PC_MEM_HOST, 64, 1, nodes.as_mut_ptr() as *mut c_void, ffi::PC_MEM_HOST) }, ffi::PC_OK);
    assert_eq!(&nodes[..32], &tree.root()[..]);
}

/// The whole drop-in: `HipMarlinKZG10` against `MarlinKZG10` on the same key, polynomials, point and sponge -- identical
/// commitments and proof, and the reference's `check` accepts the device's proof.
#[test]
fn marlin_kzg10_commit_open_equal_the_reference() {
    use ark_bls12_381::{Bls12_381, Fr};
    use ark_pcs_bench_templates::test_sponge;
    use ark_poly_commit::{marlin_pc::MarlinKZG10, LabeledPolynomial, PolynomialCommitment};
    use poly_commit_hip::HipMarlinKZG10;
    type Poly = DensePolynomial<Fr>;
    type Cpu = MarlinKZG10<Bls12_381, Poly>;
    type Hip = HipMarlinKZG10<Bls12_381, Poly>;
    let rng = &mut test_rng();
    let d = (1 << 12) - 1;
    let pp = Cpu::setup(d, None, rng).unwrap();
    let (ck, vk) = Cpu::trim(&pp, d, 1, Some(&[d - 7])).unwrap();
    let polys = vec![
        LabeledPolynomial::new("a".into(), Poly::rand(d, rng), None, None),
        LabeledPolynomial::new("b".into(), Poly::rand(d, rng), None, Some(1)),
        LabeledPolynomial::new("c".into(), Poly::rand(d - 7, rng), Some(d - 7), None),
    ];
    let point = Fr::rand(rng);
    let seed = || <rand_chacha::ChaCha20Rng as rand_chacha::rand_core::SeedableRng>::seed_from_u64(7);
    let (c_cpu, s_cpu) = Cpu::commit(&ck, &polys, Some(&mut seed())).unwrap();
    let (c_hip, s_hip) = Hip::commit(&ck, &polys, Some(&mut seed())).unwrap();
    assert_eq!(c_cpu.iter().map(|c| c.commitment().clone()).collect::<Vec<_>>(), c_hip.iter().map(|c| c.commitment().clone()).collect::<Vec<_>>());
    assert_eq!(s_cpu, s_hip);
    let p_cpu = Cpu::open(&ck, &polys, &c_cpu, &point, &mut test_sponge::<Fr>(), &s_cpu, None).unwrap();
    let p_hip = Hip::open(&ck, &polys, &c_hip, &point, &mut test_sponge::<Fr>(), &s_hip, None).unwrap();
    assert_eq!(p_cpu, p_hip);
    let values: Vec<Fr> = polys.iter().map(|p| ark_poly::Polynomial::evaluate(p.polynomial(), &point)).collect();
    assert!(Cpu::check(&vk, &c_hip, &point, values, &p_hip, &mut test_sponge::<Fr>(), None).unwrap());
}

/// `open_combinations` / `check_combinations` (the shapes of the reference's `single_equation_test`, `two_equation_test` and
/// `two_equation_degree_bound_test`, `poly-commit/src/lib.rs:1302-1384`, whose templates are private to the reference's own test
/// module), generic over a (reference type, drop-in type) pair with the same associated types: the proofs the drop-in's
/// `open_combinations` makes are the reference's, bit for bit (`evals: None`, one proof per query point over the COMBINED
/// polynomials), and the reference's verifier accepts them -- for combinations of several polynomials with coefficients other than
/// one, with a constant term, and for a degree-bounded polynomial alone in its equation; a wrong evaluation is rejected.
/// (Round-3 advisor finding for Sonic, round-4 review for Marlin and the IPA: the trait's DEFAULT prover paired with the reference
/// types' overriding verifiers rejects every honest proof of a real combination.)
fn combinations_equal_the_reference_and_verify<F, Cpu, Hip>(d: usize, bound: usize, seed_v: u64)
where
    F: PrimeField,
    Cpu: ark_poly_commit::PolynomialCommitment<F, DensePolynomial<F>>,
    Hip: ark_poly_commit::PolynomialCommitment<F, DensePolynomial<F>, UniversalParams = Cpu::UniversalParams, CommitterKey = Cpu::CommitterKey,
                                               VerifierKey = Cpu::VerifierKey, Commitment = Cpu::Commitment, CommitmentState = Cpu::CommitmentState,
                                               Proof = Cpu::Proof, BatchProof = Cpu::BatchProof, Error = Cpu::Error>,
    Cpu::Commitment: PartialEq + core::fmt::Debug,
    Cpu::CommitmentState: PartialEq + core::fmt::Debug,
    Cpu::BatchProof: PartialEq + core::fmt::Debug,
    Cpu::Error: core::fmt::Debug,
{
    use ark_pcs_bench_templates::test_sponge;
    use ark_poly_commit::{Evaluations, LCTerm, LabeledPolynomial, LinearCombination, QuerySet};
    type Poly<F> = DensePolynomial<F>;
    let rng = &mut test_rng();
    let pp = Cpu::setup(d, None, rng).unwrap();
    let (ck, vk) = Cpu::trim(&pp, d, 1, Some(&[bound])).unwrap();
    let polys = vec![
        LabeledPolynomial::new("a".into(), Poly::<F>::rand(d, rng), None, None),
        LabeledPolynomial::new("b".into(), Poly::<F>::rand(d, rng), None, Some(1)),
        LabeledPolynomial::new("c".into(), Poly::<F>::rand(bound, rng), Some(bound), None),
    ];
    let seed = || <rand_chacha::ChaCha20Rng as rand_chacha::rand_core::SeedableRng>::seed_from_u64(seed_v);
    let (comms, states) = Hip::commit(&ck, &polys, Some(&mut seed())).unwrap();
    let (comms_cpu, states_cpu) = Cpu::commit(&ck, &polys, Some(&mut seed())).unwrap();
    assert_eq!(comms.iter().map(|c| c.commitment().clone()).collect::<Vec<_>>(), comms_cpu.iter().map(|c| c.commitment().clone()).collect::<Vec<_>>());
    assert_eq!(states, states_cpu);
    // eq0 = 2a + 3b - 5 (two polynomials, coefficients != 1, a constant), eq1 = a - b, eq2 = c alone (degree-bounded: coefficient one)
    let mut eq0 = LinearCombination::empty("eq0");
    eq0.push((F::from(2u64), "a".to_string().into())); eq0.push((F::from(3u64), "b".to_string().into())); eq0.push((-F::from(5u64), LCTerm::One));
    let mut eq1 = LinearCombination::empty("eq1");
    eq1.push((F::from(1u64), "a".to_string().into())); eq1.push((-F::from(1u64), "b".to_string().into()));
    let mut eq2 = LinearCombination::empty("eq2");
    eq2.push((F::from(1u64), "c".to_string().into()));
    let lcs = vec![eq0, eq1, eq2];
    let (z0, z1) = (F::rand(rng), F::rand(rng));
    let mut query_set = QuerySet::new();
    let mut evals = Evaluations::new();
    let ev = |l: &str, z: F| ark_poly::Polynomial::evaluate(polys.iter().find(|p| p.label() == l).unwrap().polynomial(), &z);
    for (label, pname, z) in [("eq0", "z0", z0), ("eq1", "z0", z0), ("eq1", "z1", z1), ("eq2", "z1", z1)] {
        query_set.insert((label.to_string(), (pname.to_string(), z)));
        let v = match label { "eq0" => F::from(2u64) * ev("a", z) + F::from(3u64) * ev("b", z) - F::from(5u64), "eq1" => ev("a", z) - ev("b", z), _ => ev("c", z) };
        evals.insert((label.to_string(), z), v);
    }
    let p_hip = Hip::open_combinations(&ck, &lcs, &polys, &comms, &query_set, &mut test_sponge::<F>(), &states, Some(&mut seed())).unwrap();
    let p_cpu = Cpu::open_combinations(&ck, &lcs, &polys, &comms, &query_set, &mut test_sponge::<F>(), &states, Some(&mut seed())).unwrap();
    assert_eq!(p_hip.proof, p_cpu.proof);
    assert!(p_hip.evals.is_none() && p_cpu.evals.is_none());
    assert!(Cpu::check_combinations(&vk, &lcs, &comms, &query_set, &evals, &p_hip, &mut test_sponge::<F>(), rng).unwrap());
    assert!(Hip::check_combinations(&vk, &lcs, &comms, &query_set, &evals, &p_hip, &mut test_sponge::<F>(), rng).unwrap());
    // the batch methods the combinations run through, directly: the drop-in's batch_open equals the reference's, and both batch_checks accept it
    let mut qs = QuerySet::new();
    let mut vals = Evaluations::new();
    for (label, pname, z) in [("a", "z0", z0), ("b", "z0", z0), ("b", "z1", z1), ("c", "z1", z1)] {
        qs.insert((label.to_string(), (pname.to_string(), z)));
        vals.insert((label.to_string(), z), ev(label, z));
    }
    let b_hip = Hip::batch_open(&ck, &polys, &comms, &qs, &mut test_sponge::<F>(), &states, Some(&mut seed())).unwrap();
    let b_cpu = Cpu::batch_open(&ck, &polys, &comms, &qs, &mut test_sponge::<F>(), &states, Some(&mut seed())).unwrap();
    assert_eq!(b_hip, b_cpu);
    assert!(Cpu::batch_check(&vk, &comms, &qs, &vals, &b_hip, &mut test_sponge::<F>(), rng).unwrap());
    assert!(Hip::batch_check(&vk, &comms, &qs, &vals, &b_hip, &mut test_sponge::<F>(), rng).unwrap());
    // a wrong evaluation is rejected
    let mut bad = evals.clone();
    *bad.get_mut(&("eq0".to_string(), z0)).unwrap() += F::from(1u64);
    assert!(!Cpu::check_combinations(&vk, &lcs, &comms, &query_set, &bad, &p_hip, &mut test_sponge::<F>(), rng).unwrap());
}

#[test]
fn sonic_open_combinations_equal_the_reference_and_verify() {
    use ark_bls12_381::{Bls12_381, Fr};
    type Poly = DensePolynomial<Fr>;
    let d = (1 << 12) - 1;
    combinations_equal_the_reference_and_verify::<Fr, ark_poly_commit::sonic_pc::SonicKZG10<Bls12_381, Poly>, poly_commit_hip::HipSonicKZG10<Bls12_381, Poly>>(d, d - 9, 11);
}

#[test]
fn marlin_open_combinations_equal_the_reference_and_verify() {
    use ark_bls12_381::{Bls12_381, Fr};
    type Poly = DensePolynomial<Fr>;
    let d = (1 << 12) - 1;
    combinations_equal_the_reference_and_verify::<Fr, ark_poly_commit::marlin_pc::MarlinKZG10<Bls12_381, Poly>, poly_commit_hip::HipMarlinKZG10<Bls12_381, Poly>>(d, d - 9, 13);
}

#[test]
fn ipa_open_combinations_equal_the_reference_and_verify() {
    // (the reference's own IPA tests run over JubJub, ipa_pc/mod.rs:1056-1064; Pallas is BASELINE configs[3]'s curve)
    use ark_pallas::{Affine, Fr};
    use blake2::Blake2s256;
    type Poly = DensePolynomial<Fr>;
    let d = (1 << 10) - 1;           // d + 1 a power of two (ipa_pc/mod.rs:350,375)
    combinations_equal_the_reference_and_verify::<Fr, ark_poly_commit::ipa_pc::InnerProductArgPC<Affine, Blake2s256, Poly>,
                                                  poly_commit_hip::HipIpaPC<Affine, Blake2s256, Poly>>(d, d - 9, 17);
}

/// Residency is bounded and observable: a key that `commit` made resident shows up in `pc_hip_ctx_bytes_resident`, `release`
/// gives its device memory back, and results do not depend on either.
#[test]
fn resident_keys_are_accounted_and_released() {
    use ark_bls12_381::{Bls12_381, Fr};
    use ark_poly_commit::{marlin_pc::MarlinKZG10, LabeledPolynomial, PolynomialCommitment};
    use poly_commit_hip::HipMarlinKZG10;
    type Poly = DensePolynomial<Fr>;
    type Cpu = MarlinKZG10<Bls12_381, Poly>;
    type Hip = HipMarlinKZG10<Bls12_381, Poly>;
    let rng = &mut test_rng();
    let d = (1 << 13) - 1;
    let pp = Cpu::setup(d, None, rng).unwrap();
    let (ck, _vk) = Cpu::trim(&pp, d, 0, None).unwrap();
    let polys = vec![LabeledPolynomial::new("a".into(), Poly::rand(d, rng), None, None)];
    let before = device::bytes_resident().unwrap();
    let (c1, _) = Hip::commit(&ck, &polys, None).unwrap();
    let held = device::bytes_resident().unwrap();
    assert!(held[5] == before[5] + 1 && held[1] >= before[1] + (d + 1) * 96 && held[2] > before[2]);     // the key and its window table
    assert!(Hip::release(&ck));
    device::trim().unwrap();
    let after = device::bytes_resident().unwrap();
    assert_eq!(after[5], before[5]);
    assert!(after[1] == before[1] && after[2] == before[2]);
    let (c2, _) = Hip::commit(&ck, &polys, None).unwrap();                                                // uploaded again, same commitment
    assert_eq!(c1[0].commitment(), c2[0].commitment());
    let (c_cpu, _) = Cpu::commit(&ck, &polys, None).unwrap();
    assert_eq!(c_cpu[0].commitment(), c2[0].commitment());
}
