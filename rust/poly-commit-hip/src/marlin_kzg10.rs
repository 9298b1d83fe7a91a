//! `HipMarlinKZG10<E, P>`: `PolynomialCommitment` with EXACTLY the associated types of the reference's `MarlinKZG10`
//! (`poly-commit/src/marlin/marlin_pc/mod.rs:61-68`), so universal parameters, keys, commitments, commitment states and
//! proofs are interchangeable and `setup` / `trim` / `check` (and the provided `batch_*` / `*_combinations` methods of
//! the trait) simply delegate to the reference.  `commit` (`:172-242`) and `open` (`:245-336`) are restated line by line
//! with `kzg10::KZG10::commit/open` replaced by [`crate::kzg10_hip`]:
//!
//! * `commit`: all polynomials are validated first, in order (an error surfaces before any device work, like the
//!   reference's early return); their plain MSMs against `ck.powers()` then run as ONE `pc_hip_msm_batch` over the device
//!   copies when they have the same length (config 3: 64 polynomials of degree 2^20), else one `pc_hip_msm` each; the
//!   hiding draws and the shifted commitments follow in the reference's order, so the RNG stream is consumed identically.
//! * `open`: the challenge loop is the reference's (`:266-308`); the combination `p += (challenge_j, polynomial)` (`:286`)
//!   is accumulated on the device (`pc_hip_fr_lincomb` over the cached copies) when the polynomials are large, the
//!   witness division and its MSM run there too (`kzg10_hip::open_device`); degree-bounded polynomials keep the
//!   reference's host path for their shifted witnesses (`:289-307`) and only swap the final MSM.
use ark_crypto_primitives::sponge::CryptographicSponge;
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::Zero;
use ark_poly::DenseUVPolynomial;
use ark_poly_commit::{
    kzg10,
    marlin_pc::{Commitment, CommitterKey, MarlinKZG10, Randomness, UniversalParams, VerifierKey},
    Error, LabeledCommitment, LabeledPolynomial, PCCommitmentState, PCCommitterKey, PolynomialCommitment, CHALLENGE_SIZE,
};
use ark_std::{marker::PhantomData, ops::Div, rand::RngCore};
use core::ffi::c_void;

use crate::curve::{HipCurve, HipField};
use crate::device::{self, check, ctx, DevicePoly};
use crate::{ffi, kzg10_hip};

pub struct HipMarlinKZG10<E: Pairing, P: DenseUVPolynomial<E::ScalarField>> {
    _engine: PhantomData<E>,
    _poly: PhantomData<P>,
}

// kzg10/mod.rs:424-449 (pub(crate) in the reference): restated verbatim
fn check_degrees_and_bounds<F: ark_ff::PrimeField, P: DenseUVPolynomial<F>>(supported_degree: usize, max_degree: usize,
                                                                             enforced_degree_bounds: Option<&[usize]>,
                                                                             p: &LabeledPolynomial<F, P>) -> Result<(), Error> {
    if let Some(bound) = p.degree_bound() {
        let enforced_degree_bounds = enforced_degree_bounds.ok_or(Error::UnsupportedDegreeBound(bound))?;
        if enforced_degree_bounds.binary_search(&bound).is_err() {
            Err(Error::UnsupportedDegreeBound(bound))
        } else if bound < p.degree() || bound > max_degree {
            Err(Error::IncorrectDegreeBound { poly_degree: p.degree(), degree_bound: p.degree_bound().unwrap(), supported_degree, label: p.label().to_string() })
        } else {
            Ok(())
        }
    } else {
        Ok(())
    }
}

// marlin_pc/mod.rs:34-53 (pub(crate) in the reference): restated verbatim
fn shift_polynomial<E: Pairing, P: DenseUVPolynomial<E::ScalarField>>(ck: &CommitterKey<E>, p: &P, degree_bound: usize) -> P {
    if p.is_zero() {
        P::zero()
    } else {
        let enforced_degree_bounds = ck.enforced_degree_bounds.as_ref().expect("Polynomial requires degree bounds, but `ck` does not support any");
        let largest_enforced_degree_bound = enforced_degree_bounds.last().unwrap();
        let mut shifted_polynomial_coeffs = vec![E::ScalarField::zero(); largest_enforced_degree_bound - degree_bound];
        shifted_polynomial_coeffs.extend_from_slice(p.coeffs());
        P::from_coefficients_vec(shifted_polynomial_coeffs)
    }
}

impl<E, P> PolynomialCommitment<E::ScalarField, P> for HipMarlinKZG10<E, P>
where
    E: Pairing,
    E::G1Affine: HipCurve,
    E::ScalarField: HipField,
    E::G1: VariableBaseMSM<MulBase = E::G1Affine>,
    P: DenseUVPolynomial<E::ScalarField, Point = E::ScalarField>,
    for<'a, 'b> &'a P: Div<&'b P, Output = P>,
{
    type UniversalParams = UniversalParams<E>;
    type CommitterKey = CommitterKey<E>;
    type VerifierKey = VerifierKey<E>;
    type Commitment = Commitment<E>;
    type CommitmentState = Randomness<E::ScalarField, P>;
    type Proof = kzg10::Proof<E>;
    type BatchProof = Vec<Self::Proof>;
    type Error = Error;

    fn setup<R: RngCore>(max_degree: usize, num_vars: Option<usize>, rng: &mut R) -> Result<Self::UniversalParams, Self::Error> {
        MarlinKZG10::<E, P>::setup(max_degree, num_vars, rng)                              // marlin_pc/mod.rs:71-78
    }

    fn trim(pp: &Self::UniversalParams, supported_degree: usize, supported_hiding_bound: usize, enforced_degree_bounds: Option<&[usize]>)
        -> Result<(Self::CommitterKey, Self::VerifierKey), Self::Error> {
        let (ck, vk) = MarlinKZG10::<E, P>::trim(pp, supported_degree, supported_hiding_bound, enforced_degree_bounds)?;   // :80-169
        // The SRS -> HBM upload belongs here (once per key).  `ck` is returned BY VALUE and moved by the caller, so the
        // residency registry (device::resident) keys on the address the key has when it is first USED; uploading here
        // would register an address that dies with this stack frame.  A caller that wants the upload inside `trim`'s time
        // calls `HipMarlinKZG10::warm(&ck)` once the key is where it will live.
        Ok((ck, vk))
    }

    fn commit<'a>(ck: &Self::CommitterKey, polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<E::ScalarField, P>>,
                  rng: Option<&mut dyn RngCore>) -> Result<(Vec<LabeledCommitment<Self::Commitment>>, Vec<Self::CommitmentState>), Self::Error>
    where
        P: 'a,
    {
        let rng = &mut ark_poly_commit::optional_rng::OptionalRng(rng);
        let polys: Vec<&LabeledPolynomial<E::ScalarField, P>> = polynomials.into_iter().collect();
        let enforced_degree_bounds: Option<&[usize]> = ck.enforced_degree_bounds.as_ref().map(|bounds| bounds.as_slice());
        let powers = ck.powers();

        // The plain MSMs of all polynomials against ck.powers(): one batched pass when they share a length.  The reference
        // validates polynomial i before committing to it (:197-207, kzg10/mod.rs:163): validate the prefix that the batch
        // would cover first, so that an invalid polynomial yields the same error with nothing computed for it.
        let mut plain: Vec<Option<E::G1>> = vec![None; polys.len()];
        let all_valid = polys.iter().all(|p| {
            check_degrees_and_bounds(ck.supported_degree(), ck.max_degree, enforced_degree_bounds, p).is_ok()
                && kzg10_hip::check_degree_is_too_large(p.polynomial().degree(), powers.size()).is_ok()
        });
        let len0 = polys.first().map(|p| p.polynomial().coeffs().len()).unwrap_or(0);
        let batchable = all_valid && polys.len() >= 2 && len0 >= device::min_pairs()
            && polys.iter().all(|p| p.polynomial().coeffs().len() == len0 && !p.polynomial().coeffs()[0].is_zero());
        if batchable {
            let devs = polys.iter().map(|p| device::device_poly(p.polynomial().coeffs())).collect::<Result<Vec<_>, _>>()?;
            for (slot, c) in plain.iter_mut().zip(kzg10_hip::msm_batch::<E::G1Affine>(&powers.powers_of_g, &devs, 0, len0)?) {
                *slot = Some(c);
            }
        }

        let mut commitments = Vec::new();
        let mut states = Vec::new();
        for (i, p) in polys.iter().enumerate() {
            let label = p.label();
            let degree_bound = p.degree_bound();
            let hiding_bound = p.hiding_bound();
            let polynomial: &P = p.polynomial();
            check_degrees_and_bounds(ck.supported_degree(), ck.max_degree, enforced_degree_bounds, p)?;            // :197-207

            let (comm, rand) = match plain[i] {                                                                       // was :217
                Some(c) => kzg10_hip::finish_commit::<E, P>(&powers, c, hiding_bound, Some(rng))?,
                None => kzg10_hip::commit::<E, P>(&powers, polynomial, hiding_bound, Some(rng))?,
            };
            let (shifted_comm, shifted_rand) = if let Some(degree_bound) = degree_bound {
                let shifted_powers = ck.shifted_powers(degree_bound).ok_or(Error::UnsupportedDegreeBound(degree_bound))?;
                // shifted_powers(bound) is a tail of ck.shifted_powers: the same resident key, found with its offset
                let (shifted_comm, shifted_rand) = kzg10_hip::commit::<E, P>(&shifted_powers, polynomial, hiding_bound, Some(rng))?;   // was :223
                (Some(shifted_comm), Some(shifted_rand))
            } else {
                (None, None)
            };
            let comm = Commitment { comm, shifted_comm };
            let state = Randomness { rand, shifted_rand };
            commitments.push(LabeledCommitment::new(label.to_string(), comm, degree_bound));
            states.push(state);
        }
        Ok((commitments, states))
    }

    fn open<'a>(ck: &Self::CommitterKey, labeled_polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<E::ScalarField, P>>,
                _commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>, point: &'a P::Point,
                sponge: &mut impl CryptographicSponge, states: impl IntoIterator<Item = &'a Self::CommitmentState>,
                _rng: Option<&mut dyn RngCore>) -> Result<Self::Proof, Self::Error>
    where
        P: 'a,
        Self::CommitmentState: 'a,
        Self::Commitment: 'a,
    {
        let mut r = kzg10::Randomness::empty();
        let mut shifted_w = P::zero();
        let mut shifted_r = kzg10::Randomness::empty();
        let mut shifted_r_witness = P::zero();
        let mut enforce_degree_bound = false;

        // p = sum_j challenge_j * polynomial_j (:286) is formed after the loop: on the device when large, with P's own
        // `+=` otherwise (the same field operations either way)
        let mut terms: Vec<(E::ScalarField, &P)> = Vec::new();
        for (polynomial, rand) in labeled_polynomials.into_iter().zip(states) {
            let degree_bound = polynomial.degree_bound();
            assert_eq!(degree_bound.is_some(), rand.shifted_rand.is_some());
            let enforced_degree_bounds: Option<&[usize]> = ck.enforced_degree_bounds.as_ref().map(|bounds| bounds.as_slice());
            check_degrees_and_bounds(ck.supported_degree(), ck.max_degree, enforced_degree_bounds, polynomial)?;

            // compute next challenges challenge^j and challenge^{j+1}.
            let challenge_j = sponge.squeeze_field_elements_with_sizes(&[CHALLENGE_SIZE])[0];                        // :282
            terms.push((challenge_j, polynomial.polynomial()));
            r += (challenge_j, &rand.rand);

            if let Some(degree_bound) = degree_bound {                                                               // :289-307, the reference's host path
                enforce_degree_bound = true;
                let shifted_rand = rand.shifted_rand.as_ref().unwrap();
                let (witness, shifted_rand_witness) = kzg10::KZG10::<E, P>::compute_witness_polynomial(polynomial.polynomial(), *point, shifted_rand)?;
                let challenge_j_1 = sponge.squeeze_field_elements_with_sizes(&[CHALLENGE_SIZE])[0];
                let shifted_witness = shift_polynomial(ck, &witness, degree_bound);
                shifted_w += (challenge_j_1, &shifted_witness);
                shifted_r += (challenge_j_1, shifted_rand);
                if let Some(shifted_rand_witness) = shifted_rand_witness {
                    shifted_r_witness += (challenge_j_1, &shifted_rand_witness);
                }
            }
        }

        let powers = ck.powers();
        let n_out = terms.iter().map(|(_, q)| q.coeffs().len()).max().unwrap_or(0);
        let proof = if n_out >= device::min_pairs() {                                                                // was :310
            let c = ctx()?;
            let devs = terms.iter().map(|(_, q)| device::device_poly(q.coeffs())).collect::<Result<Vec<_>, _>>()?;
            let ptrs: Vec<*const c_void> = devs.iter().map(|d| d.dev as *const c_void).collect();
            let lens: Vec<usize> = terms.iter().map(|(_, q)| q.coeffs().len()).collect();
            let xi: Vec<[u64; 4]> = terms.iter().map(|(ch, _)| ch.to_mont_limbs()).collect();
            let comb = DevicePoly::alloc(n_out)?;
            check(c, unsafe {
                ffi::pc_hip_fr_lincomb(c.raw, <E::ScalarField as HipField>::FIELD_OF, ptrs.as_ptr(), ffi::PC_MEM_DEVICE, lens.as_ptr(), terms.len(),
                                       xi.as_ptr() as *const c_void, comb.dev, ffi::PC_MEM_DEVICE, n_out)
            })?;
            // KZG10::open checks p.degree() against the powers (kzg10/mod.rs:293); the combination's degree is at most n_out - 1
            // (a vanishing top coefficient only makes the reference's bound looser)
            kzg10_hip::open_device::<E, P>(&powers, &comb, n_out, *point, &r)?
        } else {
            let mut p = P::zero();
            for (challenge_j, q) in &terms {
                p += (*challenge_j, *q);
            }
            kzg10_hip::open::<E, P>(&powers, &p, *point, &r)?
        };
        let mut w = proof.w.into_group();
        let mut random_v = proof.random_v;

        if enforce_degree_bound {
            let shifted_proof = kzg10_hip::open_with_witness_polynomial::<E, P>(&ck.shifted_powers(None).unwrap(), *point, &shifted_r, &shifted_w,
                                                                                Some(&shifted_r_witness))?;       // was :315
            w += &shifted_proof.w.into_group();
            if let Some(shifted_random_v) = shifted_proof.random_v {
                random_v = random_v.map(|v| v + &shifted_random_v);
            }
        }
        Ok(kzg10::Proof { w: w.into_affine(), random_v })
    }

    fn check<'a>(vk: &Self::VerifierKey, commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>, point: &'a P::Point,
                 values: impl IntoIterator<Item = E::ScalarField>, proof: &Self::Proof, sponge: &mut impl CryptographicSponge,
                 rng: Option<&mut dyn RngCore>) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        // verifier side (pairings, O(#commitments) group operations): the reference's, unchanged (:339-373)
        MarlinKZG10::<E, P>::check(vk, commitments, point, values, proof, sponge, rng)
    }
    // batch_open / batch_check / open_combinations / check_combinations: the trait's provided methods (lib.rs:269-577), which
    // call the `open` / `check` above.  (MarlinKZG10 overrides batch_check / *_combinations with the `Marlin` helper struct,
    // marlin_pc/mod.rs:376-530, which is private to the reference; the provided methods are semantically equivalent.)
}

impl<E, P> HipMarlinKZG10<E, P>
where
    E: Pairing,
    E::G1Affine: HipCurve,
    P: DenseUVPolynomial<E::ScalarField>,
{
    /// Upload `ck.powers` (and `ck.shifted_powers`) and build their window tables now instead of at the first `commit`:
    /// call it once after `trim`, when `ck` sits where it will live (the registry keys on its address).
    pub fn warm(ck: &CommitterKey<E>) -> Result<(), Error> {
        device::resident(&ck.powers[..])?;
        if let Some(sp) = ck.shifted_powers.as_ref() {
            device::resident(&sp[..])?;
        }
        Ok(())
    }

    /// Drop the device copies (bases, window tables, pipelines) of `ck`'s powers now, instead of waiting for the key cache's
    /// byte budget (`PC_HIP_KEY_BUDGET_MB`) to push them out: `trim` returns the key by value (`marlin_pc/mod.rs:80-169`), so
    /// no `Drop` of the reference's type can do this.  A later `commit` / `open` with the same key uploads it again.
    pub fn release(ck: &CommitterKey<E>) -> bool {
        let mut any = device::release(&ck.powers[..]);
        if let Some(sp) = ck.shifted_powers.as_ref() {
            any |= device::release(&sp[..]);
        }
        any
    }
}
