//! ref-golden: golden vectors from the REAL reference for the commit/open hot path.
//!
//!     cd rust/ref-golden && cargo run --release            # writes ../../tests/golden/ref_arkworks.json
//!     cargo run --release -- /some/where/ref_arkworks.json   # or another path
//!
//! Every input is regenerated from a seed exactly as `oracle/pyref.py` does it (`gen_bases`: P_i = (i+1) G, `gen_scalars`:
//! SplitMix64, four little-endian u64 limbs, top limb masked to the modulus bit length, rejection sampled), so the file only
//! holds seeds, sizes and the reference's OUTPUTS.  `tests/test_ref_golden_cpu.py` checks pyref and the C++ oracle against the
//! file, `tests/test_ref_golden_gpu.py` checks the HIP library; both skip (loudly) while the file does not exist.
//!
//! What is called, and what it pins:
//!   constants     FftField / PrimeField constants of the three scalar fields, the curves' generators
//!   serialize     CanonicalSerialize of G1 points (both modes, both signs, infinity) and of Fr elements
//!   msm           <G::Group as VariableBaseMSM>::msm_bigint                (the call at kzg10/mod.rs:175,255; ipa_pc/mod.rs:64)
//!   kzg           KZG10::commit / KZG10::open                               (kzg10/mod.rs:157-210, 287-310)
//!   marlin_open   MarlinKZG10::commit / open over three polynomials         (marlin/marlin_pc/mod.rs:172-336)
//!   ipa           InnerProductArgPC::commit / open over Pallas, n = 2^10    (ipa_pc/mod.rs:403-723)
//!   reed_solomon  GeneralEvaluationDomain::new(m rho_inv).fft(msg)          (= reed_solomon, linear_codes/utils.rs:112-127, which is pub(crate))
//!   ligero        LinearCodePCS<UnivariateLigero<..>>::commit               (linear_codes/mod.rs:228-298) with bench-templates' hashers
//!
//! Sponge-derived challenges are recorded in the file (a clone of the sponge is squeezed in the order the scheme squeezes it),
//! so the consumers need not restate Poseidon: they take the challenges as inputs, as the C ABI does.
//!
//! NEVER COMPILED in the image this was written in (no rustc).  Expect to fix a bound or an import on first contact; the
//! schema of the output is fixed by `tools/ref_golden_rehearsal.py`, which writes the same file from pyref.
use ark_crypto_primitives::{
    crh::{sha256::Sha256, CRHScheme, TwoToOneCRHScheme},
    merkle_tree::{ByteDigestConverter, Config},
    sponge::{
        poseidon::{PoseidonConfig, PoseidonSponge},
        CryptographicSponge, FieldElementSize,
    },
};
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{BigInteger, FftField, One, PrimeField, UniformRand, Zero};
use ark_pcs_bench_templates::{FieldToBytesColHasher, LeafIdentityHasher};
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial, EvaluationDomain, GeneralEvaluationDomain, Polynomial};
use ark_poly_commit::{
    ipa_pc, kzg10,
    linear_codes::{LigeroPCParams, LinearCodePCS, UnivariateLigero},
    marlin_pc, LabeledPolynomial, PolynomialCommitment,
};
use ark_serialize::CanonicalSerialize;
use ark_std::{borrow::Cow, test_rng};
use blake2::Blake2s256;

// ------------------------------------------------------------------------------------------------
// seeded inputs: the same streams as oracle/pyref.py
// ------------------------------------------------------------------------------------------------
struct SplitMix64(u64);

impl SplitMix64 {
    fn next(&mut self) -> u64 {
        self.0 = self.0.wrapping_add(0x9E37_79B9_7F4A_7C15);
        let mut z = self.0;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
        z ^ (z >> 31)
    }
}

/// pyref.gen_scalars: n elements uniform in [0, p).
fn gen_scalars<F: PrimeField>(seed: u64, n: usize) -> Vec<F> {
    let bits = F::MODULUS_BIT_SIZE as usize;
    let nl = (bits + 63) / 64;
    let top_bits = bits - 64 * (nl - 1);
    let top_mask = if top_bits == 64 { u64::MAX } else { (1u64 << top_bits) - 1 };
    let mut rng = SplitMix64(seed);
    let mut out = Vec::with_capacity(n);
    while out.len() < n {
        let mut limbs: Vec<u64> = (0..nl).map(|_| rng.next()).collect();
        limbs[nl - 1] &= top_mask;
        let mut le_bits = Vec::with_capacity(64 * nl);
        for l in &limbs {
            for b in 0..64 {
                le_bits.push((l >> b) & 1 == 1);
            }
        }
        let big = <F::BigInt as BigInteger>::from_bits_le(&le_bits);
        if let Some(f) = F::from_bigint(big) {
            // None when the integer is >= p: rejected, as pyref rejects it
            out.push(f);
        }
    }
    out
}

/// pyref.gen_bases: P_i = (i + 1) G for the curve's standard generator.
fn gen_bases<G: AffineRepr>(n: usize) -> Vec<G> {
    let g = G::generator().into_group();
    let mut cur = g;
    let mut proj = Vec::with_capacity(n);
    for _ in 0..n {
        proj.push(cur);
        cur += &g;
    }
    G::Group::normalize_batch(&proj)
}

// ------------------------------------------------------------------------------------------------
// JSON by hand (no serde: one dependency less to resolve)
// ------------------------------------------------------------------------------------------------
fn hex(bytes: &[u8]) -> String {
    let mut s = String::with_capacity(2 * bytes.len());
    for b in bytes {
        s.push_str(&format!("{:02x}", b));
    }
    s
}

/// a field element as "0x..." of its canonical integer (big-endian digits)
fn fe<F: PrimeField>(f: &F) -> String {
    format!("\"0x{}\"", hex(&f.into_bigint().to_bytes_be()))
}

fn fes<F: PrimeField>(v: &[F]) -> String {
    format!("[{}]", v.iter().map(|f| fe(f)).collect::<Vec<_>>().join(", "))
}

/// an affine point as [x, y] (canonical integers) or null for the point at infinity
fn pt<G: AffineRepr>(p: &G) -> String
where
    G::BaseField: PrimeField,
{
    match p.xy() {
        None => "null".to_string(),
        Some((x, y)) => format!("[{}, {}]", fe(&x), fe(&y)),
    }
}

fn pts<G: AffineRepr>(v: &[G]) -> String
where
    G::BaseField: PrimeField,
{
    format!("[{}]", v.iter().map(|p| pt(p)).collect::<Vec<_>>().join(", "))
}

fn bytes_json(b: &[u8]) -> String {
    format!("\"{}\"", hex(b))
}

fn ser_unc<T: CanonicalSerialize>(t: &T) -> Vec<u8> {
    let mut v = Vec::new();
    t.serialize_uncompressed(&mut v).unwrap();
    v
}

fn ser_cmp<T: CanonicalSerialize>(t: &T) -> Vec<u8> {
    let mut v = Vec::new();
    t.serialize_compressed(&mut v).unwrap();
    v
}

fn obj(fields: Vec<(&str, String)>) -> String {
    format!("{{{}}}", fields.iter().map(|(k, v)| format!("\"{}\": {}", k, v)).collect::<Vec<_>>().join(", "))
}

fn arr(items: Vec<String>) -> String {
    format!("[{}]", items.join(",\n  "))
}

fn s(x: &str) -> String {
    format!("\"{}\"", x)
}

// ------------------------------------------------------------------------------------------------
// the sponge of the reference's tests and benches (bench-templates/src/lib.rs:221-247, where it is private)
// ------------------------------------------------------------------------------------------------
fn test_sponge<F: PrimeField>() -> PoseidonSponge<F> {
    let (full_rounds, partial_rounds, alpha) = (8usize, 31usize, 17u64);
    let mds = vec![
        vec![F::one(), F::zero(), F::one()],
        vec![F::one(), F::one(), F::zero()],
        vec![F::zero(), F::one(), F::one()],
    ];
    let mut ark = Vec::new();
    let mut rng = test_rng();
    for _ in 0..(full_rounds + partial_rounds) {
        ark.push((0..3).map(|_| F::rand(&mut rng)).collect::<Vec<F>>());
    }
    PoseidonSponge::new(&PoseidonConfig::new(full_rounds, partial_rounds, alpha, mds, ark, 2, 1))
}

/// the next `count` challenges a scheme would squeeze with `CHALLENGE_SIZE` = Truncated(128) (lib.rs:580)
fn replay_challenges<F: PrimeField>(sponge: &PoseidonSponge<F>, count: usize) -> Vec<F> {
    let mut clone = sponge.clone();
    (0..count).map(|_| clone.squeeze_field_elements_with_sizes::<F>(&[FieldElementSize::Truncated(128)])[0]).collect()
}

// ------------------------------------------------------------------------------------------------
// sections
// ------------------------------------------------------------------------------------------------
fn constants<G: AffineRepr>(curve: &str) -> String
where
    G::BaseField: PrimeField,
    G::ScalarField: FftField,
{
    let g = G::generator();
    let two_g = (g.into_group() + g.into_group()).into_affine();
    obj(vec![
        ("curve", s(curve)),
        ("generator", pt(&g)),
        ("two_generator", pt(&two_g)),
        ("fq_modulus", format!("\"0x{}\"", hex(&<G::BaseField as PrimeField>::MODULUS.to_bytes_be()))),
        ("fr_modulus", format!("\"0x{}\"", hex(&<G::ScalarField as PrimeField>::MODULUS.to_bytes_be()))),
        ("fr_multiplicative_generator", fe(&<G::ScalarField as FftField>::GENERATOR)),
        ("fr_two_adicity", format!("{}", <G::ScalarField as FftField>::TWO_ADICITY)),
        ("fr_two_adic_root_of_unity", fe(&<G::ScalarField as FftField>::TWO_ADIC_ROOT_OF_UNITY)),
        ("fr_root_of_unity_2p11", fe(&<G::ScalarField as FftField>::get_root_of_unity(1u64 << 11).unwrap())),
        ("gen_scalars_seed_0x5eed0001_first4", fes(&gen_scalars::<G::ScalarField>(0x5EED_0001, 4))),
        ("gen_bases_first3", pts(&gen_bases::<G>(3))),
    ])
}

fn serialize_section<G: AffineRepr + CanonicalSerialize>(curve: &str) -> String
where
    G::BaseField: PrimeField,
{
    // points whose y is the smaller / the larger of {y, -y}, and infinity: the flag conventions of ark-ec's SW encoding and of
    // ark-bls12-381's own (zcash) encoding are what the library's decoder and the IPA transcript restate
    let base = gen_bases::<G>(6);
    let mut list: Vec<G> = base.clone();
    list.extend(base.iter().map(|p| (-p.into_group()).into_affine()));
    list.push(G::zero());
    let items: Vec<String> = list
        .iter()
        .map(|p| obj(vec![("point", pt(p)), ("uncompressed", bytes_json(&ser_unc(p))), ("compressed", bytes_json(&ser_cmp(p)))]))
        .collect();
    let frs = gen_scalars::<G::ScalarField>(0x5E71_A11E, 3);
    let fr_items: Vec<String> = frs.iter().map(|f| obj(vec![("value", fe(f)), ("bytes", bytes_json(&ser_unc(f)))])).collect();
    obj(vec![
        ("curve", s(curve)),
        ("points", arr(items)),
        ("fr", arr(fr_items)),
        ("vec_of_3_points_compressed", bytes_json(&ser_cmp(&base[..3].to_vec()))),
        ("vec_of_3_points_uncompressed", bytes_json(&ser_unc(&base[..3].to_vec()))),
        ("vec_of_3_fr_compressed", bytes_json(&ser_cmp(&frs))),
    ])
}

/// msm_bigint with the adversarial scalar mix of SURVEY.md section 8d: zeros, ones, r - 1, a repeated base
fn msm_case<G: AffineRepr>(curve: &str, n: usize, seed: u64) -> String
where
    G::BaseField: PrimeField,
    G::Group: VariableBaseMSM<MulBase = G>,
{
    let mut bases = gen_bases::<G>(n);
    let mut scalars = gen_scalars::<G::ScalarField>(seed, n);
    // fixed positions: consumers apply the same edits
    scalars[0] = G::ScalarField::zero();
    scalars[1] = G::ScalarField::one();
    scalars[2] = -G::ScalarField::one();
    scalars[n - 1] = G::ScalarField::zero();
    bases[3] = bases[4];
    let bigints: Vec<_> = scalars.iter().map(|x| x.into_bigint()).collect();
    let r = <G::Group as VariableBaseMSM>::msm_bigint(&bases, &bigints).into_affine();
    obj(vec![
        ("curve", s(curve)),
        ("n", format!("{}", n)),
        ("seed", format!("{}", seed)),
        ("edits", s("scalars[0]=0, scalars[1]=1, scalars[2]=r-1, scalars[n-1]=0, bases[3]=bases[4]")),
        ("result", pt(&r)),
    ])
}

type UniPoly<F> = DensePolynomial<F>;

/// KZG10::commit + KZG10::open, hiding off; `zero_low` low-index coefficients are zero (kzg10/mod.rs:452-461 skips them)
fn kzg_case<E: Pairing>(curve: &str, degree: usize, seed: u64, z_seed: u64, zero_low: usize) -> String
where
    <E::G1Affine as AffineRepr>::BaseField: PrimeField,
{
    let n = degree + 1;
    let bases = gen_bases::<E::G1Affine>(n);
    let mut coeffs = gen_scalars::<E::ScalarField>(seed, n);
    for c in coeffs.iter_mut().take(zero_low) {
        *c = E::ScalarField::zero();
    }
    let z = gen_scalars::<E::ScalarField>(z_seed, 1)[0];
    let poly = UniPoly::<E::ScalarField>::from_coefficients_vec(coeffs);
    let powers = kzg10::Powers::<E> { powers_of_g: Cow::Owned(bases), powers_of_gamma_g: Cow::Owned(Vec::new()) };
    let (comm, rand) = kzg10::KZG10::<E, UniPoly<E::ScalarField>>::commit(&powers, &poly, None, None).unwrap();
    let proof = kzg10::KZG10::<E, UniPoly<E::ScalarField>>::open(&powers, &poly, z, &rand).unwrap();
    let value = poly.evaluate(&z);
    obj(vec![
        ("curve", s(curve)),
        ("degree", format!("{}", degree)),
        ("seed", format!("{}", seed)),
        ("z_seed", format!("{}", z_seed)),
        ("zero_low", format!("{}", zero_low)),
        ("commitment", pt(&comm.0)),
        ("proof_w", pt(&proof.w)),
        ("value", fe(&value)),
        ("commitment_compressed", bytes_json(&ser_cmp(&comm))),
        ("proof_compressed", bytes_json(&ser_cmp(&proof))),
    ])
}

/// MarlinKZG10::commit + open of three polynomials of different degrees at one point, no degree bounds, hiding off
fn marlin_case<E: Pairing>(curve: &str, n: usize, seed0: u64, z_seed: u64) -> String
where
    <E::G1Affine as AffineRepr>::BaseField: PrimeField,
{
    let degrees = [n - 1, n - 3, n / 2];
    let ck = marlin_pc::CommitterKey::<E> {
        powers: gen_bases::<E::G1Affine>(n),
        shifted_powers: None,
        powers_of_gamma_g: Vec::new(),
        enforced_degree_bounds: None,
        max_degree: n - 1,
    };
    let polys: Vec<LabeledPolynomial<E::ScalarField, UniPoly<E::ScalarField>>> = degrees
        .iter()
        .enumerate()
        .map(|(j, d)| {
            let co = gen_scalars::<E::ScalarField>(seed0 + j as u64, d + 1);
            LabeledPolynomial::new(format!("p{}", j), UniPoly::<E::ScalarField>::from_coefficients_vec(co), None, None)
        })
        .collect();
    let z = gen_scalars::<E::ScalarField>(z_seed, 1)[0];
    let (comms, states) = marlin_pc::MarlinKZG10::<E, UniPoly<E::ScalarField>>::commit(&ck, &polys, None).unwrap();
    let mut sponge = test_sponge::<E::ScalarField>();
    let challenges = replay_challenges(&sponge, polys.len()); // one squeeze per polynomial without a degree bound (marlin_pc/mod.rs:283)
    let proof = marlin_pc::MarlinKZG10::<E, UniPoly<E::ScalarField>>::open(&ck, &polys, &comms, &z, &mut sponge, &states, None).unwrap();
    let values: Vec<E::ScalarField> = polys.iter().map(|p| p.evaluate(&z)).collect();
    let comm_pts: Vec<E::G1Affine> = comms.iter().map(|c| c.commitment().comm.0).collect();
    obj(vec![
        ("curve", s(curve)),
        ("n", format!("{}", n)),
        ("degrees", format!("[{}, {}, {}]", degrees[0], degrees[1], degrees[2])),
        ("seed0", format!("{}", seed0)),
        ("z_seed", format!("{}", z_seed)),
        ("opening_challenges", fes(&challenges)),
        ("commitments", pts(&comm_pts)),
        ("values", fes(&values)),
        ("proof_w", pt(&proof.w)),
    ])
}

/// InnerProductArgPC over Pallas: commit + open of two polynomials on a key of n = 2^log_n synthetic generators
fn ipa_case(log_n: usize, seed0: u64, z_seed: u64) -> String {
    type G = ark_pallas::Affine;
    type F = ark_pallas::Fr;
    type PC = ipa_pc::InnerProductArgPC<G, Blake2s256, UniPoly<F>>;
    let n = 1usize << log_n;
    let all = gen_bases::<G>(n + 2);
    let ck = ipa_pc::CommitterKey::<G> { comm_key: all[..n].to_vec(), h: all[n], s: all[n + 1], max_degree: n - 1 };
    let degrees = [n - 1, n - 5];
    let polys: Vec<LabeledPolynomial<F, UniPoly<F>>> = degrees
        .iter()
        .enumerate()
        .map(|(j, d)| LabeledPolynomial::new(format!("p{}", j), UniPoly::<F>::from_coefficients_vec(gen_scalars::<F>(seed0 + j as u64, d + 1)), None, None))
        .collect();
    let z = gen_scalars::<F>(z_seed, 1)[0];
    let (comms, states) = PC::commit(&ck, &polys, None).unwrap();
    let mut sponge = test_sponge::<F>();
    // open squeezes once before the loop and twice per polynomial inside it (ipa_pc/mod.rs:502, 525, 556); without degree bounds
    // the challenge that multiplies polynomial j is squeeze number 2 j
    let squeezed = replay_challenges(&sponge, 2 * polys.len() + 1);
    let used: Vec<F> = (0..polys.len()).map(|j| squeezed[2 * j]).collect();
    let proof = PC::open(&ck, &polys, &comms, &z, &mut sponge, &states, None).unwrap();
    let comm_pts: Vec<G> = comms.iter().map(|c| c.commitment().comm).collect();
    obj(vec![
        ("curve", s("pallas")),
        ("log_n", format!("{}", log_n)),
        ("degrees", format!("[{}, {}]", degrees[0], degrees[1])),
        ("seed0", format!("{}", seed0)),
        ("z_seed", format!("{}", z_seed)),
        ("key", s("comm_key = gen_bases[0..n], h = gen_bases[n], s = gen_bases[n+1]")),
        ("opening_challenges", fes(&used)),
        ("commitments", pts(&comm_pts)),
        ("l_vec", pts(&proof.l_vec)),
        ("r_vec", pts(&proof.r_vec)),
        ("final_comm_key", pt(&proof.final_comm_key)),
        ("c", fe(&proof.c)),
        ("hiding_comm_is_none", format!("{}", proof.hiding_comm.is_none())),
    ])
}

/// one Reed-Solomon row: what linear_codes/utils.rs:112-127 computes for a message of m coefficients
fn reed_solomon_case<F: PrimeField + FftField>(field: &str, m: usize, rho_inv: usize, seed: u64) -> String {
    let msg = gen_scalars::<F>(seed, m);
    let out = GeneralEvaluationDomain::<F>::new(m * rho_inv).unwrap().fft(&msg);
    obj(vec![("field", s(field)), ("m", format!("{}", m)), ("rho_inv", format!("{}", rho_inv)), ("seed", format!("{}", seed)), ("output", fes(&out))])
}

struct MerkleTreeParams;

impl Config for MerkleTreeParams {
    type Leaf = Vec<u8>;
    type LeafDigest = <LeafIdentityHasher as CRHScheme>::Output;
    type LeafInnerDigestConverter = ByteDigestConverter<Self::LeafDigest>;
    type InnerDigest = <Sha256 as TwoToOneCRHScheme>::Output;
    type LeafHash = LeafIdentityHasher;
    type TwoToOneHash = Sha256;
}

/// LinearCodePCS::commit of one polynomial with the types of the reference's own Ligero tests
/// (linear_codes/univariate_ligero/tests.rs:21-45): Blake2s column hashes, identity leaf hash, SHA-256 tree
fn ligero_case<F: PrimeField + FftField>(field: &str, poly_len: usize, seed: u64) -> String {
    type ColH<F> = FieldToBytesColHasher<F, Blake2s256>;
    type PCS<F> = LinearCodePCS<UnivariateLigero<F, MerkleTreeParams, UniPoly<F>, ColH<F>>, F, UniPoly<F>, MerkleTreeParams, ColH<F>>;
    let mut rng = test_rng();
    let leaf_hash_param = <LeafIdentityHasher as CRHScheme>::setup(&mut rng).unwrap();
    let two_to_one_hash_param = <Sha256 as TwoToOneCRHScheme>::setup(&mut rng).unwrap();
    let col_hash_params = <ColH<F> as CRHScheme>::setup(&mut rng).unwrap();
    let pp: LigeroPCParams<F, MerkleTreeParams, ColH<F>> = LigeroPCParams::new(128, 4, true, leaf_hash_param, two_to_one_hash_param, col_hash_params);
    let (ck, _vk) = PCS::<F>::trim(&pp, 0, 0, None).unwrap();
    let poly = LabeledPolynomial::new("p".to_string(), UniPoly::<F>::from_coefficients_vec(gen_scalars::<F>(seed, poly_len)), None, None);
    let (comms, _states) = PCS::<F>::commit(&ck, &[poly], None).unwrap();
    // LinCodePCCommitment's fields are pub(crate); its CanonicalSerialize is public: n_rows, n_cols, n_ext_cols (u64 LE each),
    // then the root (Vec<u8>: u64 LE length + bytes)
    let bytes = ser_unc(comms[0].commitment());
    obj(vec![
        ("field", s(field)),
        ("poly_len", format!("{}", poly_len)),
        ("seed", format!("{}", seed)),
        ("rho_inv", "4".to_string()),
        ("sec_param", "128".to_string()),
        ("col_hash", s("blake2s")),
        ("tree_hash", s("sha256")),
        ("commitment_uncompressed", bytes_json(&bytes)),
    ])
}

fn main() {
    let default = format!("{}/../../tests/golden/ref_arkworks.json", env!("CARGO_MANIFEST_DIR"));
    let path = std::env::args().nth(1).unwrap_or(default);
    use ark_bls12_381::Bls12_381;
    use ark_bn254::Bn254;
    let doc = obj(vec![
        ("generator", s("rust/ref-golden: ark-poly-commit (reference checkout) + ark-ec/ark-ff/ark-poly/ark-serialize 0.5")),
        ("schema", "1".to_string()),
        (
            "constants",
            arr(vec![
                constants::<ark_bls12_381::G1Affine>("bls12_381"),
                constants::<ark_bn254::G1Affine>("bn254"),
                constants::<ark_pallas::Affine>("pallas"),
            ]),
        ),
        (
            "serialize",
            arr(vec![
                serialize_section::<ark_bls12_381::G1Affine>("bls12_381"),
                serialize_section::<ark_bn254::G1Affine>("bn254"),
                serialize_section::<ark_pallas::Affine>("pallas"),
            ]),
        ),
        (
            "msm",
            arr(vec![
                msm_case::<ark_bls12_381::G1Affine>("bls12_381", 300, 0x5EED_0001),
                msm_case::<ark_bn254::G1Affine>("bn254", 300, 0x5EED_0100),
                msm_case::<ark_pallas::Affine>("pallas", 300, 0x5EED_0400),
                msm_case::<ark_bls12_381::G1Affine>("bls12_381", 1 << 14, 0x5EED_0002),
                msm_case::<ark_bn254::G1Affine>("bn254", 1 << 14, 0x5EED_0101),
                msm_case::<ark_pallas::Affine>("pallas", 1 << 14, 0x5EED_0401),
            ]),
        ),
        (
            "kzg",
            arr(vec![
                kzg_case::<Bls12_381>("bls12_381", 31, 0x5EED_0001, 7, 0),
                kzg_case::<Bls12_381>("bls12_381", 31, 0x5EED_0003, 8, 2),
                kzg_case::<Bls12_381>("bls12_381", 1 << 12, 0x5EED_0001, 7, 0),
                kzg_case::<Bn254>("bn254", 31, 0x5EED_0100, 7, 0),
                kzg_case::<Bn254>("bn254", 1 << 12, 0x5EED_0100, 7, 3),
            ]),
        ),
        (
            "marlin_open",
            arr(vec![marlin_case::<Bls12_381>("bls12_381", 256, 0x5EED_0200, 9), marlin_case::<Bn254>("bn254", 256, 0x5EED_0210, 9)]),
        ),
        ("ipa", arr(vec![ipa_case(4, 0x5EED_0410, 11), ipa_case(10, 0x5EED_0420, 11)])),
        (
            "reed_solomon",
            arr(vec![
                reed_solomon_case::<ark_bls12_381::Fr>("bls12_381", 512, 4, 0x5EED_0500),
                reed_solomon_case::<ark_bn254::Fr>("bn254", 512, 4, 0x5EED_0501),
                reed_solomon_case::<ark_bls12_381::Fr>("bls12_381", 300, 4, 0x5EED_0502),
            ]),
        ),
        ("ligero", arr(vec![ligero_case::<ark_bls12_381::Fr>("bls12_381", 1 << 12, 0x5EED_0510), ligero_case::<ark_bn254::Fr>("bn254", 1000, 0x5EED_0511)])),
    ]);
    std::fs::write(&path, doc + "\n").expect("writing the golden file");
    eprintln!("wrote {}", path);
}
