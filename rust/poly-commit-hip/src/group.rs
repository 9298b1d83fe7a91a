//! One committer key over the GPUs of a node, driven from ONE process (`pc_hip_group_*`).  The reference has
//! no multi-device path; this is the form a prover that holds one `CommitterKey` binds.  The key is cut into N contiguous
//! chunks, one per device; every call runs the complete single-device path on each chunk in parallel and adds the N partial
//! points on the host (N x 96 bytes: raw bucket arrays never move).  Results are bit-identical to the single-device calls.
//! (One process per GPU over RCCL is the other form of the same protocol: `poly_commit_amd/sharded.py`, `bench.py --gpus N`.)
use ark_ec::{CurveGroup, VariableBaseMSM};
use ark_poly_commit::Error;
use core::ffi::{c_int, c_void};
use core::marker::PhantomData;

use crate::curve::{pack_points, pack_scalars, HipCurve, HipField};
use crate::device::strerror;
use crate::ffi;

pub struct HipGroupKey<G: HipCurve> {
    g: *mut ffi::pc_group,
    srs: *mut ffi::pc_group_srs,
    n: usize,
    _g: PhantomData<G>,
}
unsafe impl<G: HipCurve> Send for HipGroupKey<G> {}
unsafe impl<G: HipCurve> Sync for HipGroupKey<G> {}

impl<G: HipCurve> Drop for HipGroupKey<G> {
    fn drop(&mut self) {
        unsafe {
            ffi::pc_hip_group_srs_free(self.srs);
            ffi::pc_hip_group_destroy(self.g);
        }
    }
}

fn ok(rc: c_int) -> Result<(), Error> {
    if rc == ffi::PC_OK { Ok(()) } else { Err(Error::InvalidParameters(format!("pc_hip_group: {}", strerror(rc)))) }
}

impl<G> HipGroupKey<G>
where
    G: HipCurve,
    G::ScalarField: HipField,
    G::Group: VariableBaseMSM<MulBase = G>,
{
    /// `trim` for the sharded key (`marlin_pc/mod.rs:80-169`): chunk d of `bases` goes to device `devices[d]` (plus the one
    /// base below it, so that commit and open address the same resident chunk); `precompute` also builds each chunk's
    /// window table.
    pub fn upload(devices: &[i32], bases: &[G], precompute: bool) -> Result<Self, Error> {
        let (mut g, mut srs) = (core::ptr::null_mut(), core::ptr::null_mut());
        ok(unsafe { ffi::pc_hip_group_create(devices.as_ptr(), devices.len() as c_int, &mut g) })?;
        let rc = if G::layout_is_abi() {
            unsafe { ffi::pc_hip_group_srs_upload(g, G::CURVE, bases.as_ptr() as *const c_void, bases.len(), core::mem::size_of::<G>(), precompute as c_int, &mut srs) }
        } else {
            let packed = pack_points(bases);
            unsafe { ffi::pc_hip_group_srs_upload(g, G::CURVE, packed.as_ptr() as *const c_void, bases.len(), 0, precompute as c_int, &mut srs) }
        };
        if rc != ffi::PC_OK {
            unsafe { ffi::pc_hip_group_destroy(g) };
            return ok(rc).map(|_| unreachable!());
        }
        Ok(Self { g, srs, n: bases.len(), _g: PhantomData })
    }

    pub fn len(&self) -> usize {
        self.n
    }
    pub fn is_empty(&self) -> bool {
        self.n == 0
    }

    fn scalars_ptr<'a>(s: &'a [G::ScalarField], keep: &'a mut Vec<u64>) -> *const c_void {
        if <G::ScalarField as HipField>::layout_is_abi() {
            s.as_ptr() as *const c_void
        } else {
            *keep = pack_scalars(s);
            keep.as_ptr() as *const c_void
        }
    }

    /// `msm_bigint(&powers_of_g[base_offset..], &coeffs)` over the sharded key (`kzg10/mod.rs:175-178`): every device reduces
    /// its chunk to one point, the N points are added.
    pub fn msm(&self, base_offset: usize, coeffs: &[G::ScalarField]) -> Result<G::Group, Error> {
        let mut keep = Vec::new();
        let mut xy = [0u64; 12];
        let mut inf = 0i32;
        ok(unsafe {
            ffi::pc_hip_group_msm(self.g, self.srs, base_offset, Self::scalars_ptr(coeffs, &mut keep), ffi::PC_SCALARS_MONTGOMERY, coeffs.len(),
                                  xy.as_mut_ptr() as *mut c_void, &mut inf)
        })?;
        Ok(if inf != 0 { <G::Group as ark_ff::Zero>::zero() } else { G::read_xy(&xy).into_group() })
    }

    /// `MarlinKZG10::commit`'s loop over k polynomials (`marlin_pc/mod.rs:192-237`; BASELINE configs[2]).
    pub fn msm_batch(&self, polys: &[&[G::ScalarField]]) -> Result<Vec<G::Group>, Error> {
        let k = polys.len();
        let w = 2 * G::FQ_LIMBS;
        let mut keeps: Vec<Vec<u64>> = (0..k).map(|_| Vec::new()).collect();
        let ptrs: Vec<*const c_void> = polys.iter().zip(keeps.iter_mut()).map(|(p, keep)| Self::scalars_ptr(p, keep)).collect();
        let lens: Vec<usize> = polys.iter().map(|p| p.len()).collect();
        let mut out = vec![0u64; k * w];
        let mut inf = vec![0i32; k];
        ok(unsafe {
            ffi::pc_hip_group_msm_batch(self.g, self.srs, ptrs.as_ptr(), lens.as_ptr(), k, ffi::PC_SCALARS_MONTGOMERY, out.as_mut_ptr() as *mut c_void, inf.as_mut_ptr())
        })?;
        Ok((0..k).map(|j| if inf[j] != 0 { <G::Group as ark_ff::Zero>::zero() } else { G::read_xy(&out[j * w..(j + 1) * w]).into_group() }).collect())
    }

    /// `KZG10::open`, hiding off (`kzg10/mod.rs:287-310`): witness polynomial + its MSM over the sharded key; also returns
    /// `p(z)`.  Per device one evaluation of its shard, the division carries composed on the host (N field elements), one
    /// division scan, one MSM.
    pub fn open(&self, coeffs: &[G::ScalarField], z: &G::ScalarField) -> Result<(G::Group, G::ScalarField), Error> {
        let mut keep = Vec::new();
        let mut xy = [0u64; 12];
        let mut inf = 0i32;
        let mut val = [0u64; 4];
        let zl = z.to_mont_limbs();
        ok(unsafe {
            ffi::pc_hip_group_kzg_open(self.g, self.srs, Self::scalars_ptr(coeffs, &mut keep), coeffs.len(), zl.as_ptr() as *const c_void,
                                       xy.as_mut_ptr() as *mut c_void, &mut inf, val.as_mut_ptr() as *mut c_void)
        })?;
        let w = if inf != 0 { <G::Group as ark_ff::Zero>::zero() } else { G::read_xy(&xy).into_group() };
        Ok((w, <G::ScalarField as HipField>::from_mont_limbs(val)))
    }
}
