//! What the C ABI needs to know about an arkworks curve: its `pc_curve` id and how field elements / affine points map to
//! the ABI's buffers (little-endian 64-bit limbs, MONTGOMERY form -- the in-memory form of `Fp<MontBackend<_, N>, N>`,
//! `include/pc_hip.h` "Conventions").
//!
//! arkworks types are `repr(Rust)`, so the zero-copy path (handing `&[G1Affine]` / `&[Fr]` to the library as they lie in
//! memory, `stride_bytes = size_of::<G1Affine>()`) is only taken after [`HipCurve::layout_is_abi`] /
//! [`HipField::layout_is_abi`] verified the layout on real values at run time; otherwise points and scalars are repacked
//! through the public accessors (`Fp.0.0` = the Montgomery limbs, `Fp::new_unchecked`).
use ark_ec::{short_weierstrass::Affine, AffineRepr};
use ark_ff::{BigInt, Fp, PrimeField};
use core::ffi::c_int;

use crate::ffi;

/// A scalar field the library computes in: 4 x 64-bit Montgomery limbs.
pub trait HipField: PrimeField {
    /// `pc_curve` whose SCALAR field this is (the `field_of` argument of the `pc_hip_fr_*` / NTT entry points).
    const FIELD_OF: c_int;
    fn to_mont_limbs(&self) -> [u64; 4];
    fn from_mont_limbs(l: [u64; 4]) -> Self;
    /// `&[Self]` may be passed as `n x 32` bytes with `PC_SCALARS_MONTGOMERY`.
    fn layout_is_abi() -> bool {
        if core::mem::size_of::<Self>() != 32 || core::mem::align_of::<Self>() > 8 {
            return false;
        }
        let probe = [Self::one(), Self::from(0x0123_4567_89ab_cdefu64), -Self::one()];
        probe.iter().all(|v| {
            let raw: [u64; 4] = unsafe { core::ptr::read_unaligned(v as *const Self as *const [u64; 4]) };
            raw == v.to_mont_limbs()
        })
    }
}

/// A G1 / Pedersen group in short-Weierstrass affine form that the library has kernels for.
pub trait HipCurve: AffineRepr {
    const CURVE: c_int;
    /// 64-bit limbs of one base-field element (6 for BLS12-381, 4 for BN254 / Pallas).
    const FQ_LIMBS: usize;
    /// x || y in Montgomery limbs; the point at infinity is (0, 0) in the packed form.
    fn write_xy(&self, out: &mut [u64]);
    /// Inverse of `write_xy` (all-zero limbs = infinity).  The library returns points that are on the curve by construction.
    fn read_xy(limbs: &[u64]) -> Self;
    /// `&[Self]` may be passed with `stride_bytes = size_of::<Self>()`: x at offset 0, y right behind it, the `infinity`
    /// flag in the byte at `2 * size_of::<Fq>()` (what `pc_hip_srs_upload` reads for strides above the packed size).
    fn layout_is_abi() -> bool;
}

macro_rules! impl_hip_field {
    ($fr:ty, $id:expr) => {
        impl HipField for $fr {
            const FIELD_OF: c_int = $id;
            #[inline]
            fn to_mont_limbs(&self) -> [u64; 4] {
                (self.0).0
            }
            #[inline]
            fn from_mont_limbs(l: [u64; 4]) -> Self {
                Fp::new_unchecked(BigInt::new(l))
            }
        }
    };
}
impl_hip_field!(ark_bls12_381::Fr, ffi::PC_CURVE_BLS12_381);
impl_hip_field!(ark_bn254::Fr, ffi::PC_CURVE_BN254);
impl_hip_field!(ark_pallas::Fr, ffi::PC_CURVE_PALLAS);

macro_rules! impl_hip_curve {
    ($cfg:ty, $fq:ty, $n:expr, $id:expr) => {
        impl HipCurve for Affine<$cfg> {
            const CURVE: c_int = $id;
            const FQ_LIMBS: usize = $n;
            fn write_xy(&self, out: &mut [u64]) {
                debug_assert!(out.len() >= 2 * $n);
                if self.infinity {
                    out[..2 * $n].fill(0);
                } else {
                    out[..$n].copy_from_slice(&(self.x.0).0);
                    out[$n..2 * $n].copy_from_slice(&(self.y.0).0);
                }
            }
            fn read_xy(limbs: &[u64]) -> Self {
                if limbs[..2 * $n].iter().all(|w| *w == 0) {
                    return Self::identity();
                }
                let mut x = [0u64; $n];
                let mut y = [0u64; $n];
                x.copy_from_slice(&limbs[..$n]);
                y.copy_from_slice(&limbs[$n..2 * $n]);
                let (x, y): ($fq, $fq) = (Fp::new_unchecked(BigInt::new(x)), Fp::new_unchecked(BigInt::new(y)));
                Self::new_unchecked(x, y)
            }
            fn layout_is_abi() -> bool {
                let fb = core::mem::size_of::<$fq>();
                if fb != 8 * $n || core::mem::size_of::<Self>() <= 2 * fb {
                    return false;
                }
                let g = <Self as AffineRepr>::generator();
                let id = Self::identity();
                let bytes = |p: &Self| -> Vec<u8> {
                    unsafe { core::slice::from_raw_parts(p as *const Self as *const u8, core::mem::size_of::<Self>()) }.to_vec()
                };
                let limb_bytes = |l: &[u64]| -> Vec<u8> { l.iter().flat_map(|w| w.to_le_bytes()).collect() };
                let gb = bytes(&g);
                gb[..fb] == limb_bytes(&(g.x.0).0)[..] && gb[fb..2 * fb] == limb_bytes(&(g.y.0).0)[..] && gb[2 * fb] == 0 && bytes(&id)[2 * fb] == 1
            }
        }
    };
}
impl_hip_curve!(ark_bls12_381::g1::Config, ark_bls12_381::Fq, 6, ffi::PC_CURVE_BLS12_381);
impl_hip_curve!(ark_bn254::g1::Config, ark_bn254::Fq, 4, ffi::PC_CURVE_BN254);
impl_hip_curve!(ark_pallas::PallasConfig, ark_pallas::Fq, 4, ffi::PC_CURVE_PALLAS);

/// Packed `n x (x || y)` limbs of a slice of points (the ABI's 96 / 64-byte form).
pub fn pack_points<G: HipCurve>(pts: &[G]) -> Vec<u64> {
    let w = 2 * G::FQ_LIMBS;
    let mut out = vec![0u64; pts.len() * w];
    for (p, o) in pts.iter().zip(out.chunks_exact_mut(w)) {
        p.write_xy(o);
    }
    out
}

/// Packed `n x 4` Montgomery limbs of a slice of scalars.
pub fn pack_scalars<F: HipField>(s: &[F]) -> Vec<u64> {
    let mut out = Vec::with_capacity(4 * s.len());
    for v in s {
        out.extend_from_slice(&v.to_mont_limbs());
    }
    out
}
