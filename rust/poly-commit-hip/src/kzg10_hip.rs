//! `KZG10::commit` / `open` / `open_with_witness_polynomial` (`poly-commit/src/kzg10/mod.rs:157-310`) restated with the
//! MSM on the resident key and the witness division on the device.  Validation, error values, the RNG draw
//! (`Randomness::rand`, `:189`) and the hiding MSM over `powers_of_gamma_g` (a handful of pairs: host, `ark-ec`) are the
//! reference's, in the reference's order, so results are bit-identical for the same inputs and seed.
//!
//! `check_degree_is_too_large` / `check_hiding_bound` are `pub(crate)` in the reference (`:393-421`) and are restated here
//! verbatim (same `Error` variants and fields).
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{PrimeField, Zero};
use ark_poly::DenseUVPolynomial;
use ark_poly_commit::{
    kzg10::{Commitment, Powers, Proof, Randomness},
    Error, PCCommitmentState,
};
use ark_std::rand::RngCore;
use core::ffi::c_void;
use std::sync::Arc;

use crate::curve::{pack_scalars, HipCurve, HipField};
use crate::device::{self, check, ctx, DevicePoly};
use crate::ffi;

/// Where the scalars of an MSM live.
pub enum Scalars<'a, F: HipField> {
    /// `&[F]` as it lies in host memory (Montgomery form: `convert_to_bigints`, `kzg10/mod.rs:463-470`, is fused into the digit kernel).
    Host(&'a [F]),
    /// `n` elements of a device buffer starting at element `first`.
    Device { buf: &'a DevicePoly, first: usize, n: usize },
}

/// `<G::Group as VariableBaseMSM>::msm_bigint(&bases[..], &scalars)` with `bases` = a resident key slice
/// (`kzg10/mod.rs:175-178`, `:255-258`; `ipa_pc/mod.rs:64`).  `min(len)` pairs, like `msm_bigint`.
pub fn msm<G>(bases: &[G], scalars: Scalars<G::ScalarField>) -> Result<G::Group, Error>
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
{
    let n_scalars = match &scalars {
        Scalars::Host(s) => s.len(),
        Scalars::Device { n, .. } => *n,
    };
    let n = n_scalars.min(bases.len());
    if n == 0 {
        return Ok(G::Group::zero());
    }
    if n < device::min_pairs() {
        // below the measured crossover the reference's own CPU path is kept
        let host: Vec<G::ScalarField> = match &scalars {
            Scalars::Host(s) => s[..n].to_vec(),
            Scalars::Device { buf, first, .. } => {
                let all: Vec<G::ScalarField> = buf.download(first + n)?;
                all[*first..].to_vec()
            }
        };
        let ints: Vec<_> = host.iter().map(|s| s.into_bigint()).collect();
        return Ok(<G::Group as VariableBaseMSM>::msm_bigint(&bases[..n], &ints));
    }
    let c = ctx()?;
    let (key, base_offset) = device::resident(bases)?;
    let mut xy = [0u64; 12];
    let mut inf = 0i32;
    let rc = match scalars {
        Scalars::Host(s) if <G::ScalarField as HipField>::layout_is_abi() => unsafe {
            ffi::pc_hip_msm(c.raw, key.srs, base_offset, s.as_ptr() as *const c_void, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_HOST, n,
                            xy.as_mut_ptr() as *mut c_void, &mut inf)
        },
        Scalars::Host(s) => {
            let packed = pack_scalars(&s[..n]);
            unsafe {
                ffi::pc_hip_msm(c.raw, key.srs, base_offset, packed.as_ptr() as *const c_void, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_HOST, n,
                                xy.as_mut_ptr() as *mut c_void, &mut inf)
            }
        }
        Scalars::Device { buf, first, .. } => unsafe {
            ffi::pc_hip_msm(c.raw, key.srs, base_offset, buf.at(first), ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE, n,
                            xy.as_mut_ptr() as *mut c_void, &mut inf)
        },
    };
    check(c, rc)?;
    Ok(if inf != 0 { G::Group::zero() } else { G::read_xy(&xy).into_group() })
}

/// Several equal-length, device-resident scalar vectors against the same key slice in ONE pass: the per-polynomial loop of
/// `MarlinKZG10::commit` (`marlin_pc/mod.rs:192-237`) as `pc_hip_msm_batch`.
pub fn msm_batch<G>(bases: &[G], polys: &[Arc<DevicePoly>], first: usize, n: usize) -> Result<Vec<G::Group>, Error>
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
{
    let c = ctx()?;
    let (key, base_offset) = device::resident(bases)?;
    let k = polys.len();
    let w = 2 * G::FQ_LIMBS;
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.at(first) as *const c_void).collect();
    let lens = vec![n; k];
    let offs = vec![base_offset; k];
    let mut out = vec![0u64; k * w];
    let mut inf = vec![0i32; k];
    check(c, unsafe {
        ffi::pc_hip_msm_batch(c.raw, key.srs, offs.as_ptr(), ptrs.as_ptr(), lens.as_ptr(), k, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_DEVICE,
                              out.as_mut_ptr() as *mut c_void, inf.as_mut_ptr())
    })?;
    Ok((0..k).map(|j| if inf[j] != 0 { G::Group::zero() } else { G::read_xy(&out[j * w..(j + 1) * w]).into_group() }).collect())
}

/// The same batch straight from the polynomials' HOST slices (`pc_hip_msm_batch` with `PC_MEM_HOST`): the library copies the polynomials
/// of a pass to the pipeline that will run it while the other pipeline runs the pass before, so 64 polynomials of degree 2^20 cost one
/// exposed copy of 8 (100 ms against 92 ms for device-resident vectors) instead of 2 GiB of PCIe in front of the batch.
pub fn msm_batch_host<G>(bases: &[G], polys: &[&[G::ScalarField]], n: usize) -> Result<Vec<G::Group>, Error>
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
{
    // a safe fn must not let the C side read past a slice: every polynomial holds n coefficients, the key n bases
    if polys.iter().any(|p| p.len() < n) || bases.len() < n {
        return Err(Error::IncorrectInputLength(format!("msm_batch_host: {} coefficients per polynomial asked of shorter inputs", n)));
    }
    if !<G::ScalarField as HipField>::layout_is_abi() {
        // (an Fp layout other than 4 little-endian u64 limbs: repack through device copies, the path above)
        let devs = polys.iter().map(|p| device::device_poly(&p[..n])).collect::<Result<Vec<_>, _>>()?;
        return msm_batch::<G>(bases, &devs, 0, n);
    }
    let c = ctx()?;
    let (key, base_offset) = device::resident(bases)?;
    let k = polys.len();
    let w = 2 * G::FQ_LIMBS;
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.as_ptr() as *const c_void).collect();
    let lens = vec![n; k];
    let offs = vec![base_offset; k];
    let mut out = vec![0u64; k * w];
    let mut inf = vec![0i32; k];
    check(c, unsafe {
        ffi::pc_hip_msm_batch(c.raw, key.srs, offs.as_ptr(), ptrs.as_ptr(), lens.as_ptr(), k, ffi::PC_SCALARS_MONTGOMERY, ffi::PC_MEM_HOST,
                              out.as_mut_ptr() as *mut c_void, inf.as_mut_ptr())
    })?;
    Ok((0..k).map(|j| if inf[j] != 0 { G::Group::zero() } else { G::read_xy(&out[j * w..(j + 1) * w]).into_group() }).collect())
}

// kzg10/mod.rs:393-402
pub(crate) fn check_degree_is_too_large(degree: usize, num_powers: usize) -> Result<(), Error> {
    let num_coefficients = degree + 1;
    if num_coefficients > num_powers {
        Err(Error::TooManyCoefficients { num_coefficients, num_powers })
    } else {
        Ok(())
    }
}

// kzg10/mod.rs:405-421
pub(crate) fn check_hiding_bound(hiding_poly_degree: usize, num_powers: usize) -> Result<(), Error> {
    if hiding_poly_degree == 0 {
        Err(Error::HidingBoundIsZero)
    } else if hiding_poly_degree >= num_powers {
        Err(Error::HidingBoundToolarge { hiding_poly_degree, num_powers })
    } else {
        Ok(())
    }
}

fn leading_zeros<F: Zero>(coeffs: &[F]) -> usize {
    // skip_leading_zeros_and_convert_to_bigints, kzg10/mod.rs:452-461: the LOW-index zero coefficients
    coeffs.iter().take_while(|c| c.is_zero()).count()
}

/// The hiding part of `KZG10::commit` (`kzg10/mod.rs:180-206`), untouched: sample the blinding polynomial where the
/// reference samples it, commit to it on the host (hiding_bound + 2 pairs), add.
fn add_hiding<E, P>(powers: &Powers<E>, commitment: &mut E::G1, hiding_bound: Option<usize>, rng: Option<&mut dyn RngCore>)
    -> Result<Randomness<E::ScalarField, P>, Error>
where
    E: Pairing,
    P: DenseUVPolynomial<E::ScalarField>,
{
    let mut randomness = Randomness::<E::ScalarField, P>::empty();
    if let Some(hiding_degree) = hiding_bound {
        let mut rng = rng.ok_or(Error::MissingRng)?;
        randomness = Randomness::rand(hiding_degree, false, None, &mut rng);
        check_hiding_bound(randomness.blinding_polynomial.degree(), powers.powers_of_gamma_g.len())?;
    }
    let random_ints: Vec<_> = randomness.blinding_polynomial.coeffs().iter().map(|s| s.into_bigint()).collect();
    let random_commitment = <E::G1 as VariableBaseMSM>::msm_bigint(&powers.powers_of_gamma_g, random_ints.as_slice()).into_affine();
    *commitment += &random_commitment;
    Ok(randomness)
}

/// `KZG10::commit` (`kzg10/mod.rs:157-210`).  With the opt-in polynomial cache (`device::device_poly`) the device copy stays for
/// `open`; without it the host slice goes to `pc_hip_msm`, which overlaps the copy with the MSM itself.
pub fn commit<E, P>(powers: &Powers<E>, polynomial: &P, hiding_bound: Option<usize>, rng: Option<&mut dyn RngCore>)
    -> Result<(Commitment<E>, Randomness<E::ScalarField, P>), Error>
where
    E: Pairing,
    E::G1Affine: HipCurve,
    E::ScalarField: HipField,
    P: DenseUVPolynomial<E::ScalarField>,
{
    check_degree_is_too_large(polynomial.degree(), powers.size())?;
    let coeffs = polynomial.coeffs();
    let lz = leading_zeros(coeffs);
    let mut commitment: E::G1 = if coeffs.len() - lz >= device::min_pairs() && device::poly_cache_bytes() > 0 {
        // opt-in polynomial cache: the device copy made here serves `open`
        let dev = device::device_poly(coeffs)?;
        msm::<E::G1Affine>(&powers.powers_of_g[lz..], Scalars::Device { buf: &dev, first: lz, n: coeffs.len() - lz })?
    } else {
        msm::<E::G1Affine>(&powers.powers_of_g[lz..], Scalars::Host(&coeffs[lz..]))?
    };
    let randomness = add_hiding::<E, P>(powers, &mut commitment, hiding_bound, rng)?;
    Ok((Commitment(commitment.into()), randomness))
}

/// The second half of `commit` for a commitment whose plain MSM was already computed in a batch (`msm_batch`).
pub(crate) fn finish_commit<E, P>(powers: &Powers<E>, mut commitment: E::G1, hiding_bound: Option<usize>, rng: Option<&mut dyn RngCore>)
    -> Result<(Commitment<E>, Randomness<E::ScalarField, P>), Error>
where
    E: Pairing,
    P: DenseUVPolynomial<E::ScalarField>,
{
    let randomness = add_hiding::<E, P>(powers, &mut commitment, hiding_bound, rng)?;
    Ok((Commitment(commitment.into()), randomness))
}

/// `KZG10::open_with_witness_polynomial` (`kzg10/mod.rs:243-284`) for a witness polynomial on the HOST (the shifted
/// witnesses of degree-bounded polynomials are assembled there, `marlin_pc/mod.rs:289-307`).
pub fn open_with_witness_polynomial<E, P>(powers: &Powers<E>, point: P::Point, randomness: &Randomness<E::ScalarField, P>,
                                          witness_polynomial: &P, hiding_witness_polynomial: Option<&P>) -> Result<Proof<E>, Error>
where
    E: Pairing,
    E::G1Affine: HipCurve,
    E::ScalarField: HipField,
    P: DenseUVPolynomial<E::ScalarField, Point = E::ScalarField>,
{
    check_degree_is_too_large(witness_polynomial.degree(), powers.size())?;
    let coeffs = witness_polynomial.coeffs();
    let lz = leading_zeros(coeffs);
    let w = msm::<E::G1Affine>(&powers.powers_of_g[lz..], Scalars::Host(&coeffs[lz..]))?;
    finish_open::<E, P>(powers, point, randomness, w, hiding_witness_polynomial)
}

// kzg10/mod.rs:262-283: the hiding part of the proof, unchanged (tiny, host)
fn finish_open<E, P>(powers: &Powers<E>, point: P::Point, randomness: &Randomness<E::ScalarField, P>, mut w: E::G1,
                     hiding_witness_polynomial: Option<&P>) -> Result<Proof<E>, Error>
where
    E: Pairing,
    P: DenseUVPolynomial<E::ScalarField, Point = E::ScalarField>,
{
    let random_v = if let Some(hiding_witness_polynomial) = hiding_witness_polynomial {
        let blinding_evaluation = randomness.blinding_polynomial.evaluate(&point);
        let random_witness_coeffs: Vec<_> = hiding_witness_polynomial.coeffs().iter().map(|s| s.into_bigint()).collect();
        w += &<E::G1 as VariableBaseMSM>::msm_bigint(&powers.powers_of_gamma_g, &random_witness_coeffs);
        Some(blinding_evaluation)
    } else {
        None
    };
    Ok(Proof { w: w.into_affine(), random_v })
}

/// `KZG10::open` (`kzg10/mod.rs:287-310`) for a polynomial that is already ON THE DEVICE (`n` coefficients of `p_dev`):
/// the witness polynomial `p / (x - z)` is computed there (`pc_hip_witness_poly`, `:217-240`) and committed without
/// leaving HBM; the hiding witness (degree <= hiding bound + 1) is divided on the host as in the reference.
pub fn open_device<E, P>(powers: &Powers<E>, p_dev: &DevicePoly, n: usize, point: E::ScalarField, rand: &Randomness<E::ScalarField, P>)
    -> Result<Proof<E>, Error>
where
    E: Pairing,
    E::G1Affine: HipCurve,
    E::ScalarField: HipField,
    P: DenseUVPolynomial<E::ScalarField, Point = E::ScalarField>,
    for<'a, 'b> &'a P: core::ops::Div<&'b P, Output = P>,
{
    // KZG10::open's check on p.degree() (:293): the caller trimmed trailing zeros when it formed `n`
    check_degree_is_too_large(n.saturating_sub(1), powers.size())?;
    let hiding_witness = if rand.is_hiding() {
        let divisor = P::from_coefficients_vec(vec![-point, E::ScalarField::from(1u64)]);
        Some(&rand.blinding_polynomial / &divisor)                                   // :228-236
    } else {
        None
    };
    let w = if n <= 1 {
        E::G1::zero()                                                                // constant polynomial: zero witness
    } else {
        let c = ctx()?;
        let q = DevicePoly::alloc(n - 1)?;
        let z = point.to_mont_limbs();
        check(c, unsafe {
            ffi::pc_hip_witness_poly(c.raw, <E::ScalarField as HipField>::FIELD_OF, p_dev.dev as *const c_void, ffi::PC_MEM_DEVICE, n,
                                     z.as_ptr() as *const c_void, q.dev, ffi::PC_MEM_DEVICE)
        })?;
        // skip_leading_zeros (:250-251) would shift the base slice past zero low coefficients; an MSM skips zero scalars
        // anyway (a zero digit adds nothing), so the full quotient goes against powers_of_g[0..]
        msm::<E::G1Affine>(&powers.powers_of_g[..], Scalars::Device { buf: &q, first: 0, n: n - 1 })?
    };
    finish_open::<E, P>(powers, point, rand, w, hiding_witness.as_ref())
}

/// `KZG10::open` for a polynomial in host memory (`kzg10/mod.rs:287-310`).
pub fn open<E, P>(powers: &Powers<E>, p: &P, point: E::ScalarField, rand: &Randomness<E::ScalarField, P>) -> Result<Proof<E>, Error>
where
    E: Pairing,
    E::G1Affine: HipCurve,
    E::ScalarField: HipField,
    P: DenseUVPolynomial<E::ScalarField, Point = E::ScalarField>,
    for<'a, 'b> &'a P: core::ops::Div<&'b P, Output = P>,
{
    check_degree_is_too_large(p.degree(), powers.size())?;
    let n = p.coeffs().len();
    if n < device::min_pairs() {
        // small polynomials: ark-poly's division + the CPU MSM inside `msm`
        let divisor = P::from_coefficients_vec(vec![-point, E::ScalarField::from(1u64)]);
        let witness = p / &divisor;
        let hiding = if rand.is_hiding() { Some(&rand.blinding_polynomial / &divisor) } else { None };
        return open_with_witness_polynomial::<E, P>(powers, point, rand, &witness, hiding.as_ref());
    }
    if device::poly_cache_bytes() > 0 {
        let dev = device::device_poly(p.coeffs())?;
        return open_device::<E, P>(powers, &dev, n, point, rand);
    }
    // host coefficients: copy + witness division + MSM as ONE call (pc_hip_kzg_open; large polynomials in parts, top part first: the
    // copy + division of a part under the accumulation of the part above)
    let hiding_witness = if rand.is_hiding() {
        let divisor = P::from_coefficients_vec(vec![-point, E::ScalarField::from(1u64)]);
        Some(&rand.blinding_polynomial / &divisor)                                   // :228-236
    } else {
        None
    };
    let c = ctx()?;
    let (key, base_offset) = device::resident(&powers.powers_of_g[..])?;
    let z = point.to_mont_limbs();
    let mut xy = [0u64; 12];
    let mut inf = 0i32;
    let rc = if <E::ScalarField as HipField>::layout_is_abi() {
        unsafe {
            ffi::pc_hip_kzg_open(c.raw, key.srs, base_offset, p.coeffs().as_ptr() as *const c_void, ffi::PC_MEM_HOST, n, z.as_ptr() as *const c_void,
                                 xy.as_mut_ptr() as *mut c_void, &mut inf)
        }
    } else {
        let packed = pack_scalars(p.coeffs());
        unsafe {
            ffi::pc_hip_kzg_open(c.raw, key.srs, base_offset, packed.as_ptr() as *const c_void, ffi::PC_MEM_HOST, n, z.as_ptr() as *const c_void,
                                 xy.as_mut_ptr() as *mut c_void, &mut inf)
        }
    };
    check(c, rc)?;
    let w = if inf != 0 { E::G1::zero() } else { <E::G1Affine as HipCurve>::read_xy(&xy).into_group() };
    finish_open::<E, P>(powers, point, rand, w, hiding_witness.as_ref())
}
