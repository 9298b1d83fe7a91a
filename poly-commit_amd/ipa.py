"""InnerProductArgPC::open halving loop (poly-commit/src/ipa_pc/mod.rs:664-711) on one GPU.

All vectors (comm_key, coefficients, powers of z) stay in HBM for the whole proof; per round
only the two 64-byte points L, R come down and one challenge goes up.  The Fiat-Shamir hash
that produces the challenge (`compute_random_oracle_challenge`, :74-87, Blake2s over
ark-serialize bytes) is host work on two points and stays with the caller: it is passed in as
`next_challenge(l_xy, r_xy) -> u (Montgomery Fr limbs)`.
"""
import numpy as np

from . import _ffi
from .sharded import FR_MODULUS, _R, _int_to_limbs, _limbs_to_int


def ipa_open_rounds(ctx, curve, comm_key, coeffs_dev, n, point_mont, h_prime_xy, next_challenge, timings=None):
    """comm_key: n x (x||y) host array; coeffs_dev: torch cuda int64 tensor (n,4), Montgomery,
    CONSUMED (folded in place).  Returns (l_vec, r_vec, final_comm_key, c) as numpy arrays."""
    import time
    import torch
    assert n & (n - 1) == 0

    class _T:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            self.t = time.perf_counter()

        def __exit__(self, *a):
            if timings is not None:
                timings[self.name] = timings.get(self.name, 0.0) + (time.perf_counter() - self.t) * 1e3
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    with _T("upload_key"):
        srs = ctx.upload_srs(curve, np.ascontiguousarray(comm_key))
    h_prime_xy = np.ascontiguousarray(h_prime_xy)
    z = torch.empty((n, 4), dtype=torch.int64, device=coeffs_dev.device)
    ctx.fr_powers(curve, point_mont, n, z.data_ptr())
    cptr, zptr = coeffs_dev.data_ptr(), z.data_ptr()
    l_vec, r_vec = [], []
    while n > 1:
        h = n // 2
        # l = cm_commit(key_l, coeffs_r) + h' * <coeffs_r, z_l>;  r = cm_commit(key_r, coeffs_l) + h' * <coeffs_l, z_r>
        with _T("msm_enqueue"):
            jl = srs.msm_async(cptr + 32 * h, n=h, base_offset=0, montgomery=True)
            jr = srs.msm_async(cptr, n=h, base_offset=h, montgomery=True)
        with _T("fr_dot"):
            ip_l = ctx.fr_dot(curve, cptr + 32 * h, zptr, h)
            ip_r = ctx.fr_dot(curve, cptr, zptr + 32 * h, h)
        with _T("host_point_mul"):
            hl = _ffi.point_mul(curve, h_prime_xy, ip_l)               # h'.mul(inner_product): one point, host
            hr = _ffi.point_mul(curve, h_prime_xy, ip_r)
        with _T("msm_wait"):
            l = _ffi.points_sum(curve, np.stack([jl.wait()[0], hl]))
            r = _ffi.points_sum(curve, np.stack([jr.wait()[0], hr]))
        l_vec.append(l)
        r_vec.append(r)
        u = np.ascontiguousarray(next_challenge(l, r), dtype=np.uint64)
        ui = pow(_limbs_to_int(u) * rinv % p, -1, p) * _R % p          # u^-1, Montgomery
        with _T("fr_fold"):
            ctx.fr_fold(curve, cptr, cptr + 32 * h, h, _int_to_limbs(ui))   # coeffs_l += u^-1 coeffs_r
            ctx.fr_fold(curve, zptr, zptr + 32 * h, h, u)                   # z_l += u z_r
        with _T("ec_fold"):
            srs.ec_fold(h, u)                                               # key_l += u key_r, normalised
        n = h
    final_key = srs.read(0, 1)[0]
    c = coeffs_dev[0].cpu().numpy().view(np.uint64).copy()
    srs.free()
    return np.stack(l_vec), np.stack(r_vec), final_key, c
