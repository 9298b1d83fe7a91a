//! `HipIpaPC<G, D, P>`: `PolynomialCommitment` with the associated types of the reference's `InnerProductArgPC`
//! (`poly-commit/src/ipa_pc/mod.rs:338-345`): `setup` / `trim` / `check` / `batch_check` / `check_combinations` delegate,
//! `open_combinations` (`:858-965`) is restated over the device `open`, `commit` (`:403-473`) and `open`
//! (`:475-723`) are restated with `cm_commit`'s MSM (`:54-72`) and the body of the halving loop (`:664-711`) on the device.
//!
//! Per round of `open` (n -> n/2) the reference does two MSMs of n/2 pairs, two inner products, the folds of the
//! coefficient and z vectors, `k_l += k_r * u` with one full-width scalar multiplication PER ELEMENT plus
//! `normalize_batch`, and one Blake2s challenge.  Here the three vectors live in HBM for all rounds:
//!   l, r          `pc_hip_msm_async` x2 on the resident key (halves addressed by `base_offset`)   `:671,674`
//!   <c_r, z_l>..  both inner products of a round come out of the pass that folds the vectors (`pc_hip_ipa_fold_dots`) `:672,675`
//!   h' * <..>     `pc_hip_point_mul` (one point, host -- as in the reference)
//!   transcript    `serialize_uncompressed` + `compute_random_oracle_challenge`: the reference's own Rust, two points a round
//!   folds         `pc_hip_ipa_fold_dots`: c_l += u^-1 c_r, z_l += u z_r + the next round's inner products, 64 bytes back `:691-697`
//!   key fold      rounds 1 and 2 in one step from the committer key's two-level fold table (`pc_hip_ec_fold2_from`, round 2's commitments by
//!                 `pc_hip_ipa_round2_msms`), then `pc_hip_ec_fold` (GLV ladder per element + batched normalisation) while n > 2^16; from there on the key
//!                 stays FIXED and the folds act on per-base factors s_j (`pc_hip_ipa_key_scalars`): the round's MSMs run
//!                 over the fixed key with scalars c * s, `final_comm_key = sum_j s_j K_j` is one last MSM -- the same
//!                 points, bit for bit, without a latency-bound ladder pass per round                  `:699-707`
//! The loop itself is ONE library call (`pc_hip_ipa_open_rounds`, the transcript handed in as a callback: `next_round_challenge` below):
//! the library then also keeps the fixed key as a key object with its own window table, refilled per opening, which the entry points
//! above cannot express (57-58 ms against 62-63 ms at 2^22 over Pallas).  The same loop spelled out over the single entry points is
//! `open_rounds_by_entry_points` (`PC_HIP_IPA_EXPLICIT_LOOP=1` selects it: same proof, bit for bit).
//! `compute_random_oracle_challenge`, `check_degrees_and_bounds`, `shift_polynomial` are private in the reference
//! (`:74-87`, `:205-239`) and restated verbatim.
use ark_crypto_primitives::sponge::CryptographicSponge;
use ark_ec::{AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{Field, One, PrimeField, UniformRand, Zero};
use ark_poly::{DenseUVPolynomial, Polynomial};
use ark_poly_commit::{
    ipa_pc::{Commitment, CommitterKey, InnerProductArgPC, Proof, Randomness, UniversalParams, VerifierKey},
    BatchLCProof, Error, Evaluations, LabeledCommitment, LabeledPolynomial, LinearCombination, PCCommitmentState, PCCommitterKey,
    PolynomialCommitment, QuerySet, CHALLENGE_SIZE,
};
use ark_serialize::CanonicalSerialize;
use ark_std::{collections::BTreeMap, convert::TryInto, marker::PhantomData, ops::Mul, rand::RngCore, string::{String, ToString}, vec::Vec};
use core::ffi::c_void;
use digest::Digest;

use crate::curve::{HipCurve, HipField};
use crate::device::{self, check, ctx, DevicePoly};
use crate::ffi;
use crate::kzg10_hip::{msm, Scalars};

/// Rounds with n at most this keep the key fixed in the loop over the single entry points (see the module doc);
/// `PC_HIP_IPA_FIXED_KEY_BELOW` overrides.  (`pc_hip_ipa_open_rounds` has its own default, 2^17: there the fixed key has a window table.)
const FIXED_KEY_BELOW: usize = 1 << 16;

/// A key handle of an opening and whether this opening owns it (a working key or a copy: freed on drop, which hands a working key back
/// to its committer key's cache; the resident committer key itself is not ours to free).
struct KeyGuard(*mut ffi::pc_srs, bool);
impl Drop for KeyGuard {
    fn drop(&mut self) {
        if self.1 {
            unsafe { ffi::pc_hip_srs_free(self.0) }
        }
    }
}

pub struct HipIpaPC<G: AffineRepr, D: Digest, P: DenseUVPolynomial<G::ScalarField>> {
    _projective: PhantomData<G>,
    _digest: PhantomData<D>,
    _poly: PhantomData<P>,
}

impl<G, D, P> HipIpaPC<G, D, P>
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
    D: Digest,
    P: DenseUVPolynomial<G::ScalarField>,
{
    /// `cm_commit` (`ipa_pc/mod.rs:54-72`): the MSM on the device, the optional `hiding_generator * randomizer` on the host.
    fn cm_commit(comm_key: &[G], scalars: Scalars<G::ScalarField>, hiding_generator: Option<G>, randomizer: Option<G::ScalarField>)
        -> Result<G::Group, Error> {
        let mut comm = msm::<G>(comm_key, scalars)?;
        if randomizer.is_some() {
            assert!(hiding_generator.is_some());
            comm += &hiding_generator.unwrap().mul(randomizer.unwrap());
        }
        Ok(comm)
    }

    // ipa_pc/mod.rs:74-87
    fn compute_random_oracle_challenge(bytes: &[u8]) -> G::ScalarField {
        let mut i = 0u64;
        let mut challenge = None;
        while challenge.is_none() {
            let mut hash_input = bytes.to_vec();
            hash_input.extend(i.to_le_bytes());
            let hash = D::digest(hash_input.as_slice());
            challenge = <G::ScalarField as Field>::from_random_bytes(&hash);
            i += 1;
        }
        challenge.unwrap()
    }

    // ipa_pc/mod.rs:205-228
    fn check_degrees_and_bounds(supported_degree: usize, p: &LabeledPolynomial<G::ScalarField, P>) -> Result<(), Error> {
        if p.degree() > supported_degree {
            return Err(Error::TooManyCoefficients { num_coefficients: p.degree() + 1, num_powers: supported_degree + 1 });
        }
        if let Some(bound) = p.degree_bound() {
            if bound < p.degree() || bound > supported_degree {
                return Err(Error::IncorrectDegreeBound { poly_degree: p.degree(), degree_bound: bound, supported_degree, label: p.label().to_string() });
            }
        }
        Ok(())
    }

    // ipa_pc/mod.rs:230-239
    fn shift_polynomial(ck: &CommitterKey<G>, p: &P, degree_bound: usize) -> P {
        if p.is_zero() {
            P::zero()
        } else {
            let mut shifted_polynomial_coeffs = vec![G::ScalarField::zero(); ck.supported_degree() - degree_bound];
            shifted_polynomial_coeffs.extend_from_slice(p.coeffs());
            P::from_coefficients_vec(shifted_polynomial_coeffs)
        }
    }

    /// The committer key of an opening as a resident handle.  It stays resident and untouched across openings: round 1's MSMs run on
    /// it (with its window table), its folds go OUT OF PLACE into working keys (from the key's fold table, built once per key).  A key
    /// that is a sub-slice of a larger resident allocation is copied instead.
    fn opening_key(ck: &CommitterKey<G>) -> Result<KeyGuard, Error> {
        let c = ctx()?;
        let d1 = ck.comm_key.len();
        let (resident, off) = device::resident(&ck.comm_key[..])?;
        if off == 0 && resident.n == d1 {
            if d1 >= 2 && !resident.fold_table_built.swap(true, std::sync::atomic::Ordering::SeqCst) {
                // the library's choice of form: two levels (rounds 1 and 2 in one step) with the widest digits that fit half of the
                // free device memory; one level on small keys
                let _ = unsafe { ffi::pc_hip_srs_precompute_fold(c.raw, resident.srs) };      // refused / OOM: the ladder fold stays in use
            }
            Ok(KeyGuard(resident.srs, false))
        } else {
            let mut copy = core::ptr::null_mut();
            let key_src = (unsafe { ffi::pc_hip_srs_device_ptr(resident.srs) } as usize + off * 16 * G::FQ_LIMBS) as *const c_void;
            check(c, unsafe { ffi::pc_hip_srs_upload(c.raw, G::CURVE, key_src, d1, 0, ffi::PC_MEM_DEVICE, &mut copy) })?;
            Ok(KeyGuard(copy, true))
        }
    }

    /// `pc_ipa_challenge_fn` of `pc_hip_ipa_open_rounds`: the reference's transcript step (`ipa_pc/mod.rs:681-689`) on the round's
    /// `(l, r)`.  `user`: the running `round_challenge` (read, replaced); the new one goes back as Montgomery limbs.
    unsafe extern "C" fn next_round_challenge(user: *mut c_void, l_xy: *const c_void, r_xy: *const c_void, out_u_mont: *mut c_void) {
        let w = 2 * G::FQ_LIMBS;
        let round_challenge = &mut *(user as *mut G::ScalarField);
        let l = G::read_xy(core::slice::from_raw_parts(l_xy as *const u64, w));
        let r = G::read_xy(core::slice::from_raw_parts(r_xy as *const u64, w));
        let mut byte_vec = Vec::new();
        round_challenge.serialize_uncompressed(&mut byte_vec).unwrap();
        l.serialize_uncompressed(&mut byte_vec).unwrap();
        r.serialize_uncompressed(&mut byte_vec).unwrap();
        *round_challenge = Self::compute_random_oracle_challenge(byte_vec.as_slice());
        let u = round_challenge.to_mont_limbs();
        core::ptr::copy_nonoverlapping(u.as_ptr(), out_u_mont as *mut u64, 4);
    }

    /// The halving loop (`ipa_pc/mod.rs:641-711`) on device-resident vectors, one library call.  `coeffs`: the d + 1 padded
    /// coefficients of the combined polynomial (consumed); returns `(l_vec, r_vec, final_comm_key, c)`.
    fn open_rounds(ck: &CommitterKey<G>, coeffs: DevicePoly, point: G::ScalarField, h_prime: G, round_challenge: G::ScalarField)
        -> Result<(Vec<G>, Vec<G>, G, G::ScalarField), Error> {
        if std::env::var_os("PC_HIP_IPA_EXPLICIT_LOOP").is_some() {
            return Self::open_rounds_by_entry_points(ck, coeffs, point, h_prime, round_challenge);
        }
        let c = ctx()?;
        let d1 = ck.comm_key.len();
        let log_d = ark_std::log2(d1) as usize;
        let w = 2 * G::FQ_LIMBS;
        let guard = Self::opening_key(ck)?;
        let mut h_xy = vec![0u64; w];
        h_prime.write_xy(&mut h_xy);
        let fixed_below: usize = std::env::var("PC_HIP_IPA_FIXED_KEY_BELOW").ok().and_then(|v| v.parse().ok()).unwrap_or(0);      // 0: the library's default
        let mut running = round_challenge;                       // the callback's state (:615-625 produced the first value)
        let mut l_xy = vec![0u64; log_d.max(1) * w];
        let mut r_xy = vec![0u64; log_d.max(1) * w];
        let mut fk = vec![0u64; w];
        let mut c0 = [0u64; 4];
        check(c, unsafe { ffi::pc_hip_ipa_open_rounds(c.raw, guard.0, coeffs.dev, d1, point.to_mont_limbs().as_ptr() as *const c_void,
                                                      h_xy.as_ptr() as *const c_void, Some(Self::next_round_challenge),
                                                      &mut running as *mut G::ScalarField as *mut c_void, fixed_below,
                                                      l_xy.as_mut_ptr() as *mut c_void, r_xy.as_mut_ptr() as *mut c_void,
                                                      fk.as_mut_ptr() as *mut c_void, c0.as_mut_ptr() as *mut c_void,
                                                      core::ptr::null_mut(), core::ptr::null_mut()) })?;
        let l_vec = (0..log_d).map(|k| G::read_xy(&l_xy[k * w..(k + 1) * w])).collect();
        let r_vec = (0..log_d).map(|k| G::read_xy(&r_xy[k * w..(k + 1) * w])).collect();
        Ok((l_vec, r_vec, G::read_xy(&fk), <G::ScalarField as HipField>::from_mont_limbs(c0)))
    }

    /// The same loop over the single entry points (what `pc_hip_ipa_open_rounds` does inside, minus the fixed key's own table).
    fn open_rounds_by_entry_points(ck: &CommitterKey<G>, coeffs: DevicePoly, point: G::ScalarField, h_prime: G, mut round_challenge: G::ScalarField)
        -> Result<(Vec<G>, Vec<G>, G, G::ScalarField), Error> {
        let c = ctx()?;
        let fid = <G::ScalarField as HipField>::FIELD_OF;
        let d1 = ck.comm_key.len();
        let log_d = ark_std::log2(d1) as usize;
        let limbs = |x: &G::ScalarField| x.to_mont_limbs();
        let mut guard = Self::opening_key(ck)?;
        let mut key = guard.0;

        // powers of z (:641-649)
        let z = DevicePoly::alloc(d1)?;
        check(c, unsafe { ffi::pc_hip_fr_powers(c.raw, fid, limbs(&point).as_ptr() as *const c_void, d1, z.dev) })?;

        let mut h_xy = vec![0u64; 2 * G::FQ_LIMBS];
        h_prime.write_xy(&mut h_xy);
        let fixed_below = std::env::var("PC_HIP_IPA_FIXED_KEY_BELOW").ok().and_then(|v| v.parse().ok()).unwrap_or(FIXED_KEY_BELOW);
        let mut fixed: Option<(usize, DevicePoly, DevicePoly)> = None;      // (n0, s, scalars of l | scalars of r)
        let w = 2 * G::FQ_LIMBS;
        // A resident committer key with a TWO-level fold table: round 1 leaves the key alone, round 2's commitments run on the committer key
        // by linearity (pc_hip_ipa_round2_msms) and the key after both folds then comes out of the table in one step (pc_hip_ec_fold2_from)
        let (mut fold_levels, mut fold_width): (core::ffi::c_uint, core::ffi::c_uint) = (0, 0);
        let two_level = !guard.1 && d1 >= 8 && d1 / 2 > fixed_below
            && unsafe { ffi::pc_hip_srs_fold_table_info(key, &mut fold_levels, &mut fold_width) } == ffi::PC_OK && fold_levels == 2;
        let mut u_first: Option<G::ScalarField> = None;

        // the inner products of the first round; every later round gets its pair from the pass that folds the vectors
        let mut dots = [[0u64; 4]; 2];
        check(c, unsafe { ffi::pc_hip_ipa_fold_dots(c.raw, fid, coeffs.dev, z.dev, d1, core::ptr::null(), core::ptr::null(), dots.as_mut_ptr() as *mut c_void) })?;
        let mut u_prev: Option<G::ScalarField> = None;

        let mut l_vec = Vec::with_capacity(log_d);
        let mut r_vec = Vec::with_capacity(log_d);
        let mut n = d1;
        while n > 1 {
            let h = n / 2;
            if fixed.is_none() && n <= fixed_below {
                let s = DevicePoly::alloc(n)?;
                check(c, unsafe { ffi::pc_hip_fr_powers(c.raw, fid, limbs(&G::ScalarField::one()).as_ptr() as *const c_void, n, s.dev) })?;   // s = (1, 1, ...)
                fixed = Some((n, s, DevicePoly::alloc(2 * n)?));
                u_prev = None;                      // the key itself carries every fold so far
            }
            // l = cm_commit(key_l, coeffs_r) + h' * <coeffs_r, z_l>;  r = cm_commit(key_r, coeffs_l) + h' * <coeffs_l, z_r>   (:671-675)
            let mut lr_xy = vec![0u64; 2 * w];
            let mut lr_inf = [0i32; 2];
            let (mut jl, mut jr) = (core::ptr::null_mut(), core::ptr::null_mut());
            let (lp, rp) = lr_xy.split_at_mut(w);
            let rc = if let Some((n0, s, alr)) = fixed.as_ref() {
                // fold of the factors by the previous challenge (size 2n) + this round's scalar vectors in one call, then the two
                // commitments on two pipelines of the fixed key
                let (fu, fm) = match u_prev.as_ref() { Some(u) => (limbs(u), 2 * n), None => ([0u64; 4], 0) };
                check(c, unsafe { ffi::pc_hip_ipa_key_scalars(c.raw, fid, coeffs.dev, n, s.dev, *n0, if fm != 0 { fu.as_ptr() as *const c_void } else { core::ptr::null() }, fm,
                                                              alr.dev, alr.at(*n0)) })?;
                check(c, unsafe { ffi::pc_hip_msm_async(c.raw, key, 0, alr.dev, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE, *n0,
                                                        lp.as_mut_ptr() as *mut c_void, &mut lr_inf[0], &mut jl) })?;
                unsafe { ffi::pc_hip_msm_async(c.raw, key, 0, alr.at(*n0), ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE, *n0,
                                               rp.as_mut_ptr() as *mut c_void, &mut lr_inf[1], &mut jr) }
            } else if let Some(u1) = u_first.as_ref() {
                // round 2 on the committer key: MSM(K'[a .. a + q), s) = MSM(K[a .. a + q), s) + u1 MSM(K[a + 2q .. a + 3q), s), blocking
                unsafe { ffi::pc_hip_ipa_round2_msms(c.raw, key, coeffs.dev, h, limbs(u1).as_ptr() as *const c_void, lp.as_mut_ptr() as *mut c_void, &mut lr_inf[0],
                                                     rp.as_mut_ptr() as *mut c_void, &mut lr_inf[1]) }
            } else {
                check(c, unsafe { ffi::pc_hip_msm_async(c.raw, key, 0, coeffs.at(h), ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE, h,
                                                        lp.as_mut_ptr() as *mut c_void, &mut lr_inf[0], &mut jl) })?;
                unsafe { ffi::pc_hip_msm_async(c.raw, key, h, coeffs.dev, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE, h,
                                               rp.as_mut_ptr() as *mut c_void, &mut lr_inf[1], &mut jr) }
            };
            let w1 = if jl.is_null() { ffi::PC_OK } else { unsafe { ffi::pc_hip_job_wait(c.raw, jl) } };      // always reap the queued job
            check(c, rc)?;
            check(c, w1)?;
            if !jr.is_null() {
                check(c, unsafe { ffi::pc_hip_job_wait(c.raw, jr) })?;
            }
            let (mut hl, mut hr) = (vec![0u64; w], vec![0u64; w]);
            check(c, unsafe { ffi::pc_hip_point_mul(G::CURVE, h_xy.as_ptr() as *const c_void, dots[0].as_ptr() as *const c_void, hl.as_mut_ptr() as *mut c_void) })?;
            check(c, unsafe { ffi::pc_hip_point_mul(G::CURVE, h_xy.as_ptr() as *const c_void, dots[1].as_ptr() as *const c_void, hr.as_mut_ptr() as *mut c_void) })?;
            let l = (G::read_xy(&lr_xy[..w]).into_group() + G::read_xy(&hl).into_group()).into_affine();     // normalize_batch(&[l, r]) (:677)
            let r = (G::read_xy(&lr_xy[w..]).into_group() + G::read_xy(&hr).into_group()).into_affine();
            l_vec.push(l);
            r_vec.push(r);

            // :681-689, the reference's own transcript code
            let mut byte_vec = Vec::new();
            round_challenge.serialize_uncompressed(&mut byte_vec).unwrap();
            l.serialize_uncompressed(&mut byte_vec).unwrap();
            r.serialize_uncompressed(&mut byte_vec).unwrap();
            round_challenge = Self::compute_random_oracle_challenge(byte_vec.as_slice());
            let round_challenge_inv = round_challenge.inverse().unwrap();

            // coeffs_l += u^-1 coeffs_r, z_l += u z_r (:691-697) and the next round's two inner products, one pass, 64 bytes back
            check(c, unsafe { ffi::pc_hip_ipa_fold_dots(c.raw, fid, coeffs.dev, z.dev, h, limbs(&round_challenge).as_ptr() as *const c_void,
                                                        limbs(&round_challenge_inv).as_ptr() as *const c_void, dots.as_mut_ptr() as *mut c_void) })?;
            if fixed.is_some() {                                                                                                                           // :699-707
                u_prev = Some(round_challenge);             // applied to the factors at the top of the next round
            } else if two_level && u_first.is_none() && n == d1 {
                u_first = Some(round_challenge);            // round 1: the key stays; round 2 runs on it as well
            } else if let Some(u1) = u_first.take() {
                // round 2: the key after both folds, K'' = K_0 + u2 K_1 + u1 K_2 + u1 u2 K_3 over the quarters of the committer key
                let mut work = core::ptr::null_mut();
                check(c, unsafe { ffi::pc_hip_ec_fold2_from(c.raw, key, h, limbs(&u1).as_ptr() as *const c_void, limbs(&round_challenge).as_ptr() as *const c_void, &mut work) })?;
                guard = KeyGuard(work, true);
                key = work;
            } else if guard.1 {
                check(c, unsafe { ffi::pc_hip_ec_fold(c.raw, key, h, limbs(&round_challenge).as_ptr() as *const c_void) })?;
            } else {
                // the first fold: out of place, the committer key stays as it is
                let mut work = core::ptr::null_mut();
                check(c, unsafe { ffi::pc_hip_ec_fold_from(c.raw, key, h, limbs(&round_challenge).as_ptr() as *const c_void, &mut work) })?;
                guard = KeyGuard(work, true);
                key = work;
            }
            n = h;
        }
        let mut fk = vec![0u64; w];
        if let Some((n0, s, _)) = fixed.as_ref() {
            if let Some(u) = u_prev.as_ref() {      // the last fold (size 2)
                check(c, unsafe { ffi::pc_hip_ipa_key_scalars(c.raw, fid, core::ptr::null(), 0, s.dev, *n0, limbs(u).as_ptr() as *const c_void, 2,
                                                              core::ptr::null_mut(), core::ptr::null_mut()) })?;
            }
            let mut inf = 0i32;
            check(c, unsafe { ffi::pc_hip_msm(c.raw, key, 0, s.dev, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE, *n0, fk.as_mut_ptr() as *mut c_void, &mut inf) })?;
        } else {
            check(c, unsafe { ffi::pc_hip_srs_read(c.raw, key, 0, 1, fk.as_mut_ptr() as *mut c_void) })?;
        }
        let c0: Vec<G::ScalarField> = coeffs.download(1)?;
        Ok((l_vec, r_vec, G::read_xy(&fk), c0[0]))
    }
}

impl<G, D, P> PolynomialCommitment<G::ScalarField, P> for HipIpaPC<G, D, P>
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
    D: Digest,
    P: DenseUVPolynomial<G::ScalarField, Point = G::ScalarField>,
{
    type UniversalParams = UniversalParams<G>;
    type CommitterKey = CommitterKey<G>;
    type VerifierKey = VerifierKey<G>;
    type Commitment = Commitment<G>;
    type CommitmentState = Randomness<G>;
    type Proof = Proof<G>;
    type BatchProof = Vec<Self::Proof>;
    type Error = Error;

    fn setup<R: RngCore>(max_degree: usize, num_vars: Option<usize>, rng: &mut R) -> Result<Self::UniversalParams, Self::Error> {
        InnerProductArgPC::<G, D, P>::setup(max_degree, num_vars, rng)                         // ipa_pc/mod.rs:347-373
    }

    fn trim(pp: &Self::UniversalParams, supported_degree: usize, supported_hiding_bound: usize, enforced_degree_bounds: Option<&[usize]>)
        -> Result<(Self::CommitterKey, Self::VerifierKey), Self::Error> {
        InnerProductArgPC::<G, D, P>::trim(pp, supported_degree, supported_hiding_bound, enforced_degree_bounds)   // :375-401; the key is uploaded at first use
    }

    fn commit<'a>(ck: &Self::CommitterKey, polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<G::ScalarField, P>>,
                  rng: Option<&mut dyn RngCore>) -> Result<(Vec<LabeledCommitment<Self::Commitment>>, Vec<Self::CommitmentState>), Self::Error>
    where
        P: 'a,
    {
        let rng = &mut ark_poly_commit::optional_rng::OptionalRng(rng);
        let mut comms = Vec::new();
        let mut states = Vec::new();
        for labeled_polynomial in polynomials {
            Self::check_degrees_and_bounds(ck.supported_degree(), labeled_polynomial)?;
            let polynomial: &P = labeled_polynomial.polynomial();
            let label = labeled_polynomial.label();
            let hiding_bound = labeled_polynomial.hiding_bound();
            let degree_bound = labeled_polynomial.degree_bound();

            let state = if let Some(h) = hiding_bound { Randomness::rand(h, degree_bound.is_some(), None, rng) } else { Randomness::empty() };

            // the polynomial's device copy serves both MSMs here and the combination in `open`
            let coeffs = polynomial.coeffs();
            let dev = if coeffs.len() >= device::min_pairs() { Some(device::device_poly(coeffs)?) } else { None };
            let sc = || match dev.as_ref() {
                Some(d) => Scalars::Device { buf: d, first: 0, n: coeffs.len() },
                None => Scalars::Host(coeffs),
            };
            let comm = Self::cm_commit(&ck.comm_key[..(polynomial.degree() + 1)], sc(), Some(ck.s), Some(state.rand))?.into();      // :443-449
            let shifted_comm = match degree_bound {                                                                                   // :451-459
                Some(d) => Some(Self::cm_commit(&ck.comm_key[(ck.supported_degree() - d)..], sc(), Some(ck.s), state.shifted_rand)?.into()),
                None => None,
            };
            let commitment = Commitment { comm, shifted_comm };
            comms.push(LabeledCommitment::new(label.to_string(), commitment, degree_bound));
            states.push(state);
        }
        Ok((comms, states))
    }

    fn open<'a>(ck: &Self::CommitterKey, labeled_polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<G::ScalarField, P>>,
                commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>, point: &'a P::Point,
                sponge: &mut impl CryptographicSponge, states: impl IntoIterator<Item = &'a Self::CommitmentState>,
                rng: Option<&mut dyn RngCore>) -> Result<Self::Proof, Self::Error>
    where
        Self::Commitment: 'a,
        Self::CommitmentState: 'a,
        P: 'a,
    {
        // :489-560, the reference's combination loop; `combined_polynomial += (cur_challenge, polynomial)` is kept on the host
        // for the shifted (zero-padded) polynomials and gathered as (challenge, polynomial) terms for the plain ones
        let mut combined_polynomial = P::zero();
        let mut combined_rand = G::ScalarField::zero();
        let mut combined_commitment_proj = G::Group::zero();
        let mut has_hiding = false;
        let mut cur_challenge: G::ScalarField = sponge.squeeze_field_elements_with_sizes(&[CHALLENGE_SIZE])[0];

        for (labeled_polynomial, (labeled_commitment, state)) in labeled_polynomials.into_iter().zip(commitments.into_iter().zip(states)) {
            let label = labeled_polynomial.label();
            assert_eq!(labeled_polynomial.label(), labeled_commitment.label());
            Self::check_degrees_and_bounds(ck.supported_degree(), labeled_polynomial)?;
            let polynomial = labeled_polynomial.polynomial();
            let degree_bound = labeled_polynomial.degree_bound();
            let hiding_bound = labeled_polynomial.hiding_bound();
            let commitment = labeled_commitment.commitment();

            combined_polynomial += (cur_challenge, polynomial);
            combined_commitment_proj += &commitment.comm.mul(cur_challenge);
            if hiding_bound.is_some() {
                has_hiding = true;
                combined_rand += &(cur_challenge * &state.rand);
            }
            cur_challenge = sponge.squeeze_field_elements_with_sizes(&[CHALLENGE_SIZE])[0];

            let has_degree_bound = degree_bound.is_some();
            assert_eq!(has_degree_bound, commitment.shifted_comm.is_some(), "shifted_comm mismatch for {}", label);
            assert_eq!(degree_bound, labeled_commitment.degree_bound(), "labeled_comm degree bound mismatch for {}", label);
            if let Some(degree_bound) = degree_bound {
                let shifted_polynomial = Self::shift_polynomial(ck, polynomial, degree_bound);
                combined_polynomial += (cur_challenge, &shifted_polynomial);
                combined_commitment_proj += &commitment.shifted_comm.unwrap().mul(cur_challenge);
                if hiding_bound.is_some() {
                    let shifted_rand = state.shifted_rand;
                    assert!(shifted_rand.is_some(), "shifted_rand.is_none() for {}", label);
                    combined_rand += &(cur_challenge * &shifted_rand.unwrap());
                }
            }
            cur_challenge = sponge.squeeze_field_elements_with_sizes(&[CHALLENGE_SIZE])[0];
        }

        let combined_v = combined_polynomial.evaluate(point);
        let d = ck.supported_degree();
        let mut combined_commitment;
        let mut hiding_commitment = None;

        if has_hiding {                                                                                            // :573-609
            let mut rng = rng.expect("hiding commitments require randomness");
            let mut hiding_polynomial = P::rand(d, &mut rng);
            hiding_polynomial -= &P::from_coefficients_slice(&[hiding_polynomial.evaluate(point)]);
            let hiding_rand = G::ScalarField::rand(&mut rng);
            let hiding_commitment_proj = Self::cm_commit(ck.comm_key.as_slice(), Scalars::Host(hiding_polynomial.coeffs()), Some(ck.s), Some(hiding_rand))?;
            let mut batch = G::Group::normalize_batch(&[combined_commitment_proj, hiding_commitment_proj]);
            hiding_commitment = Some(batch.pop().unwrap());
            combined_commitment = batch.pop().unwrap();

            let mut byte_vec = Vec::new();
            combined_commitment.serialize_uncompressed(&mut byte_vec).unwrap();
            point.serialize_uncompressed(&mut byte_vec).unwrap();
            combined_v.serialize_uncompressed(&mut byte_vec).unwrap();
            hiding_commitment.unwrap().serialize_uncompressed(&mut byte_vec).unwrap();
            let hiding_challenge = Self::compute_random_oracle_challenge(byte_vec.as_slice());
            combined_polynomial += (hiding_challenge, &hiding_polynomial);
            combined_rand += &(hiding_challenge * &hiding_rand);
            combined_commitment_proj += &(hiding_commitment.unwrap().mul(hiding_challenge) - &ck.s.mul(combined_rand));
        }
        let combined_rand = if has_hiding { Some(combined_rand) } else { None };

        combined_commitment = combined_commitment_proj.into_affine();
        let mut byte_vec = Vec::new();                                                                             // :621-629
        combined_commitment.serialize_uncompressed(&mut byte_vec).unwrap();
        point.serialize_uncompressed(&mut byte_vec).unwrap();
        combined_v.serialize_uncompressed(&mut byte_vec).unwrap();
        let round_challenge = Self::compute_random_oracle_challenge(byte_vec.as_slice());
        let h_prime = ck.h.mul(round_challenge).into_affine();

        // Pads the coefficients with zeroes to get the number of coeff to be d+1 (:632-638)
        let mut coeffs = combined_polynomial.coeffs().to_vec();
        if coeffs.len() < d + 1 {
            coeffs.resize(d + 1, G::ScalarField::zero());
        }

        let (l_vec, r_vec, final_comm_key, c) = Self::open_rounds(ck, DevicePoly::upload(&coeffs)?, *point, h_prime, round_challenge)?;
        Ok(Proof { l_vec, r_vec, final_comm_key, c, hiding_comm: hiding_commitment, rand: combined_rand })
    }

    fn check<'a>(vk: &Self::VerifierKey, commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>, point: &'a P::Point,
                 values: impl IntoIterator<Item = G::ScalarField>, proof: &Self::Proof, sponge: &mut impl CryptographicSponge,
                 rng: Option<&mut dyn RngCore>) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        // The verifier's one large computation, cm_commit(vk.comm_key, check_poly.compute_coeffs()) (:759-765), has a device
        // form as well (pc_hip_ipa_key_scalars + pc_hip_msm; C++ rendering: poly_commit_amd/host/ipa_pc.hpp::check).  The
        // succinct part (:91-203) is private to the reference, so the drop-in verifier stays the reference's own.
        InnerProductArgPC::<G, D, P>::check(vk, commitments, point, values, proof, sponge, rng)
    }

    // ---- the three methods InnerProductArgPC OVERRIDES beyond the required ones (ipa_pc/mod.rs:775-1048), so that no trait method
    // of this type resolves to another implementation than the reference type's.  The trait's DEFAULT `open_combinations`
    // (lib.rs:445-487) opens the individual polynomials and returns `evals: Some(..)`; `InnerProductArgPC::check_combinations`
    // rebuilds ONE commitment per equation and batch-checks the proofs against those, ignoring `evals` -- it rejects such a proof
    // for every non-trivial combination (round-4 review).

    fn batch_check<'a, R: RngCore>(vk: &Self::VerifierKey, commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
                                   query_set: &QuerySet<P::Point>, values: &Evaluations<P::Point, G::ScalarField>, proof: &Self::BatchProof,
                                   sponge: &mut impl CryptographicSponge, rng: &mut R) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        InnerProductArgPC::<G, D, P>::batch_check(vk, commitments, query_set, values, proof, sponge, rng)        // :775-856
    }

    fn check_combinations<'a, R: RngCore>(vk: &Self::VerifierKey, linear_combinations: impl IntoIterator<Item = &'a LinearCombination<G::ScalarField>>,
                                          commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
                                          eqn_query_set: &QuerySet<P::Point>, eqn_evaluations: &Evaluations<P::Point, G::ScalarField>,
                                          proof: &BatchLCProof<G::ScalarField, Self::BatchProof>, sponge: &mut impl CryptographicSponge,
                                          rng: &mut R) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        InnerProductArgPC::<G, D, P>::check_combinations(vk, linear_combinations, commitments, eqn_query_set, eqn_evaluations, proof, sponge, rng)   // :969-1048
    }

    // `InnerProductArgPC::open_combinations` (ipa_pc/mod.rs:858-965), restated: polynomial, randomness and commitment (with their
    // shifted twins, `combine_shifted_rand` / `combine_shifted_comm` :241-265) are combined per equation under the reference's
    // degree-bound rules, the combined commitments go through `construct_labeled_commitments` (:267-300: `comm`, then `shifted_comm`
    // when the equation carries a degree bound), and `Self::batch_open` -- the trait's provided method, which the reference type
    // does not override either -- opens the COMBINED polynomials with the `open` above, i.e. on the device.
    fn open_combinations<'a>(ck: &Self::CommitterKey, linear_combinations: impl IntoIterator<Item = &'a LinearCombination<G::ScalarField>>,
                             polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<G::ScalarField, P>>,
                             commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>, query_set: &QuerySet<P::Point>,
                             sponge: &mut impl CryptographicSponge, states: impl IntoIterator<Item = &'a Self::CommitmentState>,
                             rng: Option<&mut dyn RngCore>) -> Result<BatchLCProof<G::ScalarField, Self::BatchProof>, Self::Error>
    where
        Self::CommitmentState: 'a,
        Self::Commitment: 'a,
        P: 'a,
    {
        let label_poly_map = polynomials.into_iter().zip(states).zip(commitments).map(|((p, s), c)| (p.label(), (p, s, c))).collect::<BTreeMap<_, _>>();
        let mut lc_polynomials = Vec::new();
        let mut lc_states = Vec::new();
        let mut lc_commitments: Vec<G::Group> = Vec::new();
        let mut lc_info = Vec::new();
        for lc in linear_combinations {
            let lc_label = lc.label().clone();
            let mut poly = P::zero();
            let mut degree_bound = None;
            let mut hiding_bound = None;
            let mut combined_comm = G::Group::zero();
            let mut combined_shifted_comm: Option<G::Group> = None;
            let mut combined_rand = G::ScalarField::zero();
            let mut combined_shifted_rand: Option<G::ScalarField> = None;
            let num_polys = lc.len();
            for (coeff, label) in lc.iter().filter(|(_, l)| !l.is_one()) {
                let label: &String = label.try_into().expect("cannot be one!");
                let &(cur_poly, cur_rand, cur_comm) = label_poly_map.get(label).ok_or(Error::MissingPolynomial { label: label.to_string() })?;
                if num_polys == 1 && cur_poly.degree_bound().is_some() {                       // :899-909
                    assert!(coeff.is_one(), "Coefficient must be one for degree-bounded equations");
                    degree_bound = cur_poly.degree_bound();
                } else if cur_poly.degree_bound().is_some() {
                    return Err(Error::EquationHasDegreeBounds(lc_label));
                }
                hiding_bound = core::cmp::max(hiding_bound, cur_poly.hiding_bound());          // Some(_) > None, always
                poly += (*coeff, cur_poly.polynomial());
                combined_rand += &(cur_rand.rand * coeff);
                if let Some(new_rand) = cur_rand.shifted_rand {                                // combine_shifted_rand, :241-252
                    let coeff_new_rand = new_rand * coeff;
                    combined_shifted_rand = Some(combined_shifted_rand.map_or(coeff_new_rand, |r| r + &coeff_new_rand));
                }
                let commitment = cur_comm.commitment();
                combined_comm += &commitment.comm.mul(*coeff);
                if let Some(new_comm) = commitment.shifted_comm {                              // combine_shifted_comm, :254-265
                    let coeff_new_comm = new_comm.mul(*coeff);
                    combined_shifted_comm = Some(combined_shifted_comm.map_or(coeff_new_comm, |c| c + &coeff_new_comm));
                }
            }
            lc_polynomials.push(LabeledPolynomial::new(lc_label.clone(), poly, degree_bound, hiding_bound));
            lc_states.push(Randomness { rand: combined_rand, shifted_rand: combined_shifted_rand });
            lc_commitments.push(combined_comm);
            if let Some(combined_shifted_comm) = combined_shifted_comm {
                lc_commitments.push(combined_shifted_comm);
            }
            lc_info.push((lc_label, degree_bound));
        }
        // construct_labeled_commitments (:267-300)
        let comms = G::Group::normalize_batch(&lc_commitments);
        let mut labeled = Vec::new();
        let mut i = 0;
        for (label, degree_bound) in lc_info.into_iter() {
            let commitment = if degree_bound.is_some() {
                i += 2;
                Commitment { comm: comms[i - 2].clone(), shifted_comm: Some(comms[i - 1].clone()) }
            } else {
                i += 1;
                Commitment { comm: comms[i - 1].clone(), shifted_comm: None }
            };
            labeled.push(LabeledCommitment::new(label, commitment, degree_bound));
        }
        let proof = Self::batch_open(ck, lc_polynomials.iter(), labeled.iter(), query_set, sponge, lc_states.iter(), rng)?;
        Ok(BatchLCProof { proof, evals: None })
    }
    // batch_open: the trait's provided method (lib.rs:269-371) -- InnerProductArgPC does not override it -- calling the `open` above.
}
