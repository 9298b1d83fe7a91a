"""HyraxPC commit / open / check above the C ABI (poly-commit/src/hyrax/mod.rs), one GPU.

Hyrax commits a multilinear polynomial of n variables (n even) as dim = 2^(n/2) Pedersen commitments, one per row of
its dim x dim evaluation matrix (hyrax/mod.rs:233-242, inside a par_iter): the batched small-MSM shape.  Here

  commit   all rows in ONE pc_hip_msm_many pass over the key extended by the hiding generator,
           row_com_i = MSM(com_key, row_i) + h * r_i = MSM(com_key || h, row_i || r_i)          (:239)
  open     t.row_mul(l) as one device linear combination of the resident rows (:341), <lt, r> as a device dot
           product (:350), com_d as one MSM over the extended key (:368), z = d + c * lt on the device (:387)
  check    t' = MSM(row_coms, l) and com(z, z_d) on the device (:495-498), the two equations on the host

The sponge is the caller's (the reference absorbs the key, the row commitments and the point into a generic
CryptographicSponge, :335-341 / :377-385): its challenge c is an argument, and so are the prover's random field
elements, in the order the reference draws them -- a test can then replay the oracle's proof bit for bit.
Field elements are Montgomery limbs ((4,) uint64), points x||y; the matrix stays in HBM between commit and open.
"""
import numpy as np

from poly_commit_amd import _ffi
from poly_commit_amd.sharded import FR_MODULUS, _R, _int_to_limbs, _limbs_to_int


class InvalidNumberOfVariables(ValueError):
    """hyrax Error::InvalidNumberOfVariables (error.rs): odd number of variables, or more than the key supports."""


class IncorrectCommitmentSize(ValueError):
    """hyrax Error::IncorrectCommitmentSize: a commitment whose row count is not 2^(n/2)."""


def _to_int(curve, a):
    p = FR_MODULUS[curve]
    return _limbs_to_int(a) * pow(_R, -1, p) % p


def _to_mont(curve, v):
    p = FR_MODULUS[curve]
    return _int_to_limbs(v % p * _R % p)


def _vec_to_int(curve, arr):
    """(k, 4) Montgomery limbs -> list of canonical ints."""
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    raw = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, 4).tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") * rinv % p for i in range(len(raw) // 32)]


def _vec_to_mont(curve, vals):
    """list of canonical ints -> (k, 4) Montgomery limbs."""
    p = FR_MODULUS[curve]
    return np.frombuffer(b"".join((v % p * _R % p).to_bytes(32, "little") for v in vals), dtype="<u8").astype(np.uint64).reshape(-1, 4)


def tensor_prime(curve, values):
    """hyrax/utils.rs:27-39 on canonical ints."""
    p = FR_MODULUS[curve]
    out = [1]
    for val in reversed(values):
        out = [v * (1 - val) % p for v in out] + [v * val % p for v in out]
    return out


def _tensors(curve, point_mont):
    n = len(point_mont)
    point_rev = [_to_int(curve, x) for x in reversed(list(point_mont))]                  # :297
    return tensor_prime(curve, point_rev[n // 2:]), tensor_prime(curve, point_rev[:n // 2])   # l, r


class HyraxKey:
    """HyraxCommitterKey / HyraxVerifierKey {com_key, h} (hyrax/data_structures.rs) with the extended key
    com_key[..dim] || h resident on the device, one resident SRS per matrix dimension in use."""

    def __init__(self, ctx, curve, com_key, h):
        self.ctx, self.curve = ctx, curve
        self.com_key = np.ascontiguousarray(com_key, dtype=np.uint64)
        self.h = np.ascontiguousarray(h, dtype=np.uint64)
        self._ext = {}

    def ext(self, dim):
        if dim not in self._ext:
            self._ext[dim] = self.ctx.upload_srs(self.curve, np.concatenate([self.com_key[:dim], self.h[None]]))
        return self._ext[dim]

    def close(self):
        for s in self._ext.values():
            s.free()
        self._ext = {}


def commit(key, evals_dev, rands_mont):
    """HyraxPC::commit for one polynomial (hyrax/mod.rs:214-252).  evals_dev: torch cuda int64 (2^n, 4), the
    polynomial's evaluations (DenseMultilinearExtension::to_evaluations) as Montgomery limbs; rands_mont: (dim, 4)
    the row randomisers (the reference draws them inside the row loop).
    Returns (row_coms (dim, 2*Fq) uint64, state) with state = (rands, matrix rows on the device)."""
    import torch
    total = evals_dev.shape[0]
    n = total.bit_length() - 1
    if total != 1 << n or n % 2 == 1 or n > key.com_key.shape[0]:          # :220-228
        raise InvalidNumberOfVariables(f"{n} variables")
    dim = 1 << (n // 2)
    if dim > key.com_key.shape[0]:
        raise InvalidNumberOfVariables(f"{n} variables need {dim} generators, the key has {key.com_key.shape[0]}")
    rands_mont = np.ascontiguousarray(rands_mont, dtype=np.uint64).reshape(dim, 4)
    # flat_to_matrix_column_major (hyrax/utils.rs:13-21): row r = flat[r], flat[dim + r], ...
    mat = evals_dev.view(dim, dim, 4).transpose(0, 1).contiguous()
    rd = torch.from_numpy(rands_mont.view(np.int64)).to(evals_dev.device)
    ext = torch.cat([mat, rd.view(dim, 1, 4)], dim=1).contiguous()          # row_i || r_i
    row_coms, _ = key.ext(dim).msm_many(ext.data_ptr(), m=dim + 1, n_msms=dim, montgomery=True)
    return row_coms, (rands_mont, mat)


def open(key, state, point_mont, r_eval, d_mont, r_d, r_b, c):   # noqa: A001 (the reference's name)
    """HyraxPC::open for one polynomial (hyrax/mod.rs:287-402).  r_eval, d_mont (dim, 4), r_d, r_b: the random
    elements in the order the reference draws them (:352, :361-362, :367, :371); c: the sponge's challenge (:385).
    Returns the HyraxProof fields (com_eval, com_d, com_b, z (dim, 4), z_d, z_b) and the evaluation."""
    import torch
    ctx, curve = key.ctx, key.curve
    p = FR_MODULUS[curve]
    rands, mat = state
    dim = mat.shape[0]
    n = len(point_mont)
    if n % 2 == 1 or 1 << (n // 2) != dim:
        raise InvalidNumberOfVariables(f"point of {n} variables, matrix of dimension {dim}")
    l, r = _tensors(curve, point_mont)
    l_m = _vec_to_mont(curve, l)
    dev = mat.device
    lt = torch.empty((dim, 4), dtype=torch.int64, device=dev)
    ctx.fr_lincomb(curve, [mat.data_ptr() + 32 * dim * i for i in range(dim)], l_m, n_out=dim, out=lt.data_ptr(), lens=[dim] * dim)   # :341
    r_lt = sum(a * b for a, b in zip(l, _vec_to_int(curve, rands))) % p                                                 # :345-348
    r_dev = torch.from_numpy(_vec_to_mont(curve, r).view(np.int64)).to(dev)
    ev = ctx.fr_dot(curve, lt.data_ptr(), r_dev.data_ptr(), dim)                                                    # :350
    g0 = key.com_key[0]
    com_eval = _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, g0, ev), _ffi.point_mul(curve, key.h, r_eval)]))    # :353
    d_mont = np.ascontiguousarray(d_mont, dtype=np.uint64).reshape(dim, 4)
    b = sum(a * x for a, x in zip(r, _vec_to_int(curve, d_mont))) % p                                                   # :364
    d_ext = torch.from_numpy(np.concatenate([d_mont, np.asarray(r_d, dtype=np.uint64).reshape(1, 4)]).view(np.int64)).to(dev)
    com_d = key.ext(dim).msm(d_ext.data_ptr(), n=dim + 1, montgomery=True)[0]                                       # :368
    com_b = _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, g0, _to_mont(curve, b)), _ffi.point_mul(curve, key.h, r_b)]))   # :372
    z = torch.empty((dim, 4), dtype=torch.int64, device=dev)
    ctx.fr_lincomb(curve, [d_ext.data_ptr(), lt.data_ptr()], np.stack([_to_mont(curve, 1), np.asarray(c, dtype=np.uint64)]),
                   n_out=dim, out=z.data_ptr(), lens=[dim, dim])                                                    # :387
    ci = _to_int(curve, c)
    z_d = _to_mont(curve, ci * r_lt + _to_int(curve, r_d))
    z_b = _to_mont(curve, ci * _to_int(curve, r_eval) + _to_int(curve, r_b))
    return (com_eval, com_d, com_b, z.cpu().numpy().view(np.uint64), z_d, z_b), ev


def check(key, row_coms, point_mont, proof, c):
    """HyraxPC::check for one commitment (hyrax/mod.rs:418-511): equation (14) on the host, then t' = MSM(row_coms, l)
    and the commitment of (z, z_d) on the device for equation (13).  Returns bool."""
    import torch
    ctx, curve = key.ctx, key.curve
    p = FR_MODULUS[curve]
    n = len(point_mont)
    if n % 2 == 1:
        raise InvalidNumberOfVariables(f"{n} variables")
    dim = 1 << (n // 2)
    row_coms = np.ascontiguousarray(row_coms, dtype=np.uint64)
    if row_coms.shape[0] != dim:
        raise IncorrectCommitmentSize(f"encountered {row_coms.shape[0]}, expected {dim}")
    com_eval, com_d, com_b, z, z_d, z_b = proof
    l, r = _tensors(curve, point_mont)
    z = np.ascontiguousarray(z, dtype=np.uint64).reshape(dim, 4)
    ip = sum(a * x for a, x in zip(r, _vec_to_int(curve, z))) % p
    com_dp = _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, key.com_key[0], _to_mont(curve, ip)), _ffi.point_mul(curve, key.h, z_b)]))   # :486
    if not (com_dp == _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, com_eval, c), com_b]))).all():
        return False
    rows = ctx.upload_srs(curve, row_coms)
    try:
        l_dev = torch.from_numpy(_vec_to_mont(curve, l).view(np.int64)).cuda()
        t_prime = rows.msm(l_dev.data_ptr(), n=dim, montgomery=True)[0]                                             # :495
    finally:
        rows.free()
    z_ext = torch.from_numpy(np.concatenate([z, np.asarray(z_d, dtype=np.uint64).reshape(1, 4)]).view(np.int64)).cuda()
    com_z_zd = key.ext(dim).msm(z_ext.data_ptr(), n=dim + 1, montgomery=True)[0]                                    # :498
    return bool((com_z_zd == _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, t_prime, c), com_d]))).all())
