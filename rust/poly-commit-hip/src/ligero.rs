//! Ligero's Reed-Solomon encoder on the device: `HipUnivariateLigero<F, C, P, H>` is the reference's
//! `UnivariateLigero` (`poly-commit/src/linear_codes/univariate_ligero/mod.rs`) with `LinearEncode::encode` -- the
//! `reed_solomon(msg, rho_inv)` = `GeneralEvaluationDomain::new(m * rho_inv).fft(msg)` of `linear_codes/utils.rs:112-127`
//! -- replaced by `pc_hip_ntt_batch`.  `LinearCodePCS<HipUnivariateLigero<..>, F, P, C, H>` is then a drop-in
//! `PolynomialCommitment` with the reference's own parameter, commitment, state and proof types.
//!
//! What the trait surface allows: `LinearEncode::compute_matrices` (`linear_codes/mod.rs:118-138`) calls `encode` once per
//! matrix row from a rayon `par_iter`; its return type `Matrix<F>` has crate-private constructors
//! (`poly-commit/src/utils.rs:49-61`), so a foreign crate can override `encode` but not `compute_matrices`.  Each row is
//! therefore one blocking `pc_hip_ntt_batch(rows = 1)` call (the context serialises concurrent callers); the whole-matrix
//! forms -- ONE `pc_hip_ntt_batch` for all rows, or `pc_hip_ligero_commit` for encode + column digests + Merkle tree with
//! nothing but the root leaving HBM (or, with the encoded matrix coming back in slabs beside the kernels, [`commit_matrices`]) --
//! are exposed as [`encode_matrix`] / [`commit_root`] for callers that can take flat
//! buffers, and become the trait path with a one-line upstream change (`pub fn new_from_flat`).
use ark_crypto_primitives::{
    crh::{CRHScheme, TwoToOneCRHScheme},
    merkle_tree::Config,
};
use ark_ff::PrimeField;
use ark_poly::DenseUVPolynomial;
use ark_poly_commit::{
    linear_codes::{LigeroPCParams, LinCodeParametersInfo, LinearEncode},
    Error,
};
use ark_std::{marker::PhantomData, rand::RngCore};
use core::ffi::{c_int, c_uint, c_void};

use crate::curve::{pack_scalars, HipField};
use crate::device::{check, ctx};
use crate::ffi;

pub struct HipUnivariateLigero<F: PrimeField, C: Config, P: DenseUVPolynomial<F>, H: CRHScheme> {
    _phantom: PhantomData<(F, C, P, H)>,
}

fn next_log2(n: usize) -> u32 {
    ark_std::log2(n)      // ceil(log2 n): GeneralEvaluationDomain::new(n) takes the next power of two
}

/// `rows` messages of `in_cols` coefficients each (row-major) -> `rows x 2^log_n` evaluations, natural order, arkworks' omega.
/// (Host memory on both sides: the library runs the call in slabs of rows, the transformed slabs travelling back beside the kernels
/// of the next ones -- 512 x 2^15 -> 512 x 2^17 over BLS12-381 Fr in 41 ms instead of 55.)
pub fn encode_matrix<F: HipField>(msgs: &[F], rows: usize, in_cols: usize, rho_inv: usize) -> Result<Vec<F>, Error> {
    assert_eq!(msgs.len(), rows * in_cols);
    let c = ctx()?;
    let log_n = next_log2(in_cols * rho_inv);
    if log_n > F::TWO_ADICITY {
        return Err(Error::EncodingError);        // the reference panics here ("cannot accomodate FFT", utils.rs:120-124)
    }
    let n = 1usize << log_n;
    let mut out = vec![[0u64; 4]; rows * n];
    let rc = if F::layout_is_abi() {
        unsafe { ffi::pc_hip_ntt_batch(c.raw, F::FIELD_OF, msgs.as_ptr() as *const c_void, ffi::PC_MEM_HOST, rows, in_cols, log_n as c_uint,
                                       out.as_mut_ptr() as *mut c_void, ffi::PC_MEM_HOST) }
    } else {
        let packed = pack_scalars(msgs);
        unsafe { ffi::pc_hip_ntt_batch(c.raw, F::FIELD_OF, packed.as_ptr() as *const c_void, ffi::PC_MEM_HOST, rows, in_cols, log_n as c_uint,
                                       out.as_mut_ptr() as *mut c_void, ffi::PC_MEM_HOST) }
    };
    check(c, rc)?;
    Ok(out.into_iter().map(F::from_mont_limbs).collect())
}

/// Steps 1-3 of `LinearCodePCS::commit` (`linear_codes/mod.rs:248-277`) for one coefficient matrix in one call: encode every
/// row, digest every column (`FieldToBytesColHasher<F, D>`, D = SHA-256 or BLAKE2s), build the Merkle tree (byte-digest
/// two-to-one hash, `ByteDigestConverter`); returns `(root, leaves)`.  `hash ids`: `ffi::PC_HASH_*`.
pub fn commit_root<F: HipField>(mat: &[F], rows: usize, in_cols: usize, rho_inv: usize, col_hash: c_int, tree_hash: c_int)
    -> Result<([u8; 32], Vec<[u8; 32]>), Error> {
    assert_eq!(mat.len(), rows * in_cols);
    let c = ctx()?;
    let log_n = next_log2(in_cols * rho_inv);
    let n = 1usize << log_n;
    let mut leaves = vec![[0u8; 32]; n];
    let mut nodes = vec![[0u8; 32]; (1usize << log_n.max(1)) - 1];
    let packed;
    let src = if F::layout_is_abi() { mat.as_ptr() as *const c_void } else { packed = pack_scalars(mat); packed.as_ptr() as *const c_void };
    check(c, unsafe {
        ffi::pc_hip_ligero_commit(c.raw, F::FIELD_OF, src, ffi::PC_MEM_HOST, rows, in_cols, log_n as c_uint, col_hash, tree_hash, 1,
                                  core::ptr::null_mut(), ffi::PC_MEM_HOST, leaves.as_mut_ptr() as *mut c_void, nodes.as_mut_ptr() as *mut c_void)
    })?;
    Ok((nodes[0], leaves))
}

/// `commit_root` that also brings the encoded matrix back (`rows x 2^log_n`, row-major): with the coefficient matrix the caller
/// already holds, everything `LinCodePCCommitmentState` keeps for `open` (`linear_codes/mod.rs:264-268`: mat, ext_mat, leaves).
/// Matrix and encoded matrix both live in host memory here, which is the shape `pc_hip_ligero_commit` runs in slabs of rows: a slab
/// is copied in, encoded and absorbed into the column digests while the slabs before it travel back (2^24 coefficients over
/// BLS12-381 Fr: 2 GiB of encoded matrix, 39-43 ms for the call against 58-60 ms as one matrix).
pub fn commit_matrices<F: HipField>(mat: &[F], rows: usize, in_cols: usize, rho_inv: usize, col_hash: c_int, tree_hash: c_int)
    -> Result<([u8; 32], Vec<[u8; 32]>, Vec<F>), Error> {
    assert_eq!(mat.len(), rows * in_cols);
    let c = ctx()?;
    let log_n = next_log2(in_cols * rho_inv);
    if log_n > F::TWO_ADICITY {
        return Err(Error::EncodingError);
    }
    let n = 1usize << log_n;
    let mut leaves = vec![[0u8; 32]; n];
    let mut nodes = vec![[0u8; 32]; (1usize << log_n.max(1)) - 1];
    let mut ext = vec![[0u64; 4]; rows * n];
    let packed;
    let src = if F::layout_is_abi() { mat.as_ptr() as *const c_void } else { packed = pack_scalars(mat); packed.as_ptr() as *const c_void };
    check(c, unsafe {
        ffi::pc_hip_ligero_commit(c.raw, F::FIELD_OF, src, ffi::PC_MEM_HOST, rows, in_cols, log_n as c_uint, col_hash, tree_hash, 1,
                                  ext.as_mut_ptr() as *mut c_void, ffi::PC_MEM_HOST, leaves.as_mut_ptr() as *mut c_void,
                                  nodes.as_mut_ptr() as *mut c_void)
    })?;
    Ok((nodes[0], leaves, ext.into_iter().map(F::from_mont_limbs).collect()))
}

/// `commit_root` with the rows of the coefficient matrix spread over several devices (`pc_hip_group_ligero_commit`): every device
/// encodes its rows, the column digests are chained through the devices (48 bytes per column travel from device to device instead
/// of a transpose of the encoded matrix), the last device builds the tree.  Same `(root, leaves)`, bit for bit.
pub fn commit_root_group<F: HipField>(devices: &[i32], mat: &[F], rows: usize, in_cols: usize, rho_inv: usize, col_hash: c_int, tree_hash: c_int)
    -> Result<([u8; 32], Vec<[u8; 32]>), Error> {
    assert_eq!(mat.len(), rows * in_cols);
    let log_n = next_log2(in_cols * rho_inv);
    let n = 1usize << log_n;
    let mut leaves = vec![[0u8; 32]; n];
    let mut nodes = vec![[0u8; 32]; (1usize << log_n.max(1)) - 1];
    let packed;
    let src = if F::layout_is_abi() { mat.as_ptr() as *const c_void } else { packed = pack_scalars(mat); packed.as_ptr() as *const c_void };
    let mut g = core::ptr::null_mut();
    let fail = |rc: c_int| Error::InvalidParameters(format!("pc_hip_group: {}", crate::device::strerror(rc)));
    let rc = unsafe { ffi::pc_hip_group_create(devices.as_ptr(), devices.len() as c_int, &mut g) };
    if rc != ffi::PC_OK { return Err(fail(rc)); }
    let rc = unsafe {
        ffi::pc_hip_group_ligero_commit(g, F::FIELD_OF, src, rows, in_cols, log_n as c_uint, col_hash, tree_hash, 1, core::ptr::null_mut(),
                                        leaves.as_mut_ptr() as *mut c_void, nodes.as_mut_ptr() as *mut c_void)
    };
    unsafe { ffi::pc_hip_group_destroy(g) };
    if rc != ffi::PC_OK { return Err(fail(rc)); }
    Ok((nodes[0], leaves))
}

impl<F, C, P, H> LinearEncode<F, C, P, H> for HipUnivariateLigero<F, C, P, H>
where
    F: HipField,
    C: Config,
    P: DenseUVPolynomial<F>,
    P::Point: Into<F>,
    H: CRHScheme,
{
    type LinCodePCParams = LigeroPCParams<F, C, H>;

    // univariate_ligero/mod.rs:37-54
    fn setup<R: RngCore>(_max_degree: usize, _num_vars: Option<usize>, _rng: &mut R, leaf_hash_param: <<C as Config>::LeafHash as CRHScheme>::Parameters,
                         two_to_one_hash_param: <<C as Config>::TwoToOneHash as TwoToOneCRHScheme>::Parameters, col_hash_params: H::Parameters)
        -> Self::LinCodePCParams {
        Self::LinCodePCParams::new(128, 4, true, leaf_hash_param, two_to_one_hash_param, col_hash_params)
    }

    /// was `Ok(reed_solomon(msg, param.rho_inv))` (univariate_ligero/mod.rs:56-58)
    fn encode(msg: &[F], param: &Self::LinCodePCParams) -> Result<Vec<F>, Error> {
        let rho_inv = param.distance().1;          // distance() = (rho_inv - 1, rho_inv); the field itself is crate-private
        encode_matrix(msg, 1, msg.len(), rho_inv)
    }

    fn poly_to_vec(polynomial: &P) -> Vec<F> {
        polynomial.coeffs().to_vec()
    }

    fn point_to_vec(point: P::Point) -> Vec<F> {
        vec![point.into()]
    }

    // univariate_ligero/mod.rs:69-87
    fn tensor(z: &F, left: usize, right: usize) -> (Vec<F>, Vec<F>) {
        let mut left_out = Vec::with_capacity(left);
        let mut pow_a = F::one();
        for _ in 0..left {
            left_out.push(pow_a);
            pow_a *= z;
        }
        let mut right_out = Vec::with_capacity(right);
        let mut pow_b = F::one();
        for _ in 0..right {
            right_out.push(pow_b);
            pow_b *= pow_a;
        }
        (left_out, right_out)
    }
}
