"""GPU parity for the KZG10 commit/open data path (hiding off):
commit = MSM(powers, coeffs) (kzg10/mod.rs:157-210), open = witness polynomial (:217-240)
+ MSM (:243-284), against the CPU oracle's restatement, and the reference's own property
checks: commitment homomorphism (kzg10/mod.rs:520-544) and p(z) = q(z)(z - z0) + p(z0)."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254", "pallas"]


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 66, 4097, 70000])
def test_witness_poly(ctx, curve, n):
    co = O.f_to_mont(curve, 1, O.gen_scalars(curve, n, n))
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 4242, 1))[0]
    got = ctx.witness_poly(curve, co, z)
    want = O.witness_poly(curve, co, z)
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.parametrize("curve", CURVES)
def test_commit_open_deg_2_12(ctx, curve):
    """BASELINE config 1 size: degree 2^12 -> 4097 coefficients."""
    d = 1 << 12
    powers = O.gen_bases(curve, d + 1)
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0001, d + 1))
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 31337, 1))[0]
    srs = ctx.upload_srs(curve, powers)
    comm, _ = srs.msm(coeffs, montgomery=True)
    rc, want = O.kzg_commit(curve, powers, coeffs)
    assert rc == 0 and (comm == want).all()
    q = ctx.witness_poly(curve, coeffs, z)
    w, _ = srs.msm(q, montgomery=True)
    rc, want_w = O.kzg_open(curve, powers, coeffs, z)
    assert rc == 0 and (w == want_w).all()
    srs.free()


def test_commit_leading_zeros_equals_offset_msm(ctx):
    """skip_leading_zeros_and_convert_to_bigints (kzg10/mod.rs:452-461): committing with
    zero low-order coefficients equals the MSM over powers[lz..] on the remaining ones."""
    curve = "bls12_381"
    n, lz = 3000, 17
    powers = O.gen_bases(curve, n)
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 3, n))
    coeffs[:lz] = 0
    srs = ctx.upload_srs(curve, powers)
    full, _ = srs.msm(coeffs, montgomery=True)
    sliced, _ = srs.msm(np.ascontiguousarray(coeffs[lz:]), base_offset=lz, montgomery=True)
    rc, want = O.kzg_commit(curve, powers, coeffs)
    assert (full == want).all() and (sliced == want).all()
    srs.free()


def test_witness_identity(ctx):
    """p(x) = q(x) (x - z) + p(z) checked at a random point with Python big ints."""
    curve = "bn254"
    fr = R.CURVES[curve]["fr"]
    p = R.FIELDS[fr]["p"]
    n = 5000
    co = O.gen_scalars(curve, 8, n)
    zc = O.gen_scalars(curve, 9, 2)
    z, x = O.limbs_to_ints(zc)
    q = ctx.witness_poly(curve, O.f_to_mont(curve, 1, co), O.f_to_mont(curve, 1, zc[:1])[0])
    qi = O.fr_from_mont_array(curve, q)
    ci = O.limbs_to_ints(co)
    assert R.poly_eval(fr, ci, x) == (R.poly_eval(fr, qi, x) * (x - z) + R.poly_eval(fr, ci, z)) % p
