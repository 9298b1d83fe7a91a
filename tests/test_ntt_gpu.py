"""GPU parity: pc_hip_ntt_batch == CPU oracle NTT, bit for bit.

Restates test_reed_solomon (linear_codes/utils.rs:303-331): encoded[j] must equal
pol.evaluate(domain.element(j)) -- natural order, arkworks' omega -- plus the Ligero matrix
shapes of SURVEY.md section 8d."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254", "pallas"]


@pytest.mark.parametrize("curve", CURVES)
def test_reed_solomon_small(ctx, curve):
    """sizes 2^1..2^9, rho_inv = 3 -> domain = next_pow2(3 m), as in the reference test."""
    fr = R.CURVES[curve]["fr"]
    for i in range(1, 10):
        m = 1 << i
        size = 1
        while size < 3 * m:
            size <<= 1
        lg = size.bit_length() - 1
        co = O.gen_scalars(curve, 100 + i, m)
        mont = O.f_to_mont(curve, 1, co).reshape(1, m, 4)
        got = ctx.ntt_batch(curve, mont, lg)
        want = O.ntt_batch(curve, mont, lg)
        assert (got == want).all(), (curve, i)
        if i <= 4:   # independent check: Horner evaluation at omega^j (Python big ints)
            vals = O.fr_from_mont_array(curve, got[0])
            ci = O.limbs_to_ints(co)
            w = R.root_of_unity(fr, lg)
            p = R.FIELDS[fr]["p"]
            for j in range(size):
                assert vals[j] == R.poly_eval(fr, ci, pow(w, j, p))


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("log_n,rows,in_cols", [(0, 3, 1), (1, 2, 2), (2, 5, 3), (5, 4, 32), (10, 3, 256), (11, 8, 512),
                                                (13, 2, 2048), (14, 2, 16384)])
def test_ntt_shapes(ctx, curve, log_n, rows, in_cols):
    co = O.gen_scalars(curve, 7 * log_n + rows, rows * in_cols)
    mont = O.f_to_mont(curve, 1, co).reshape(rows, in_cols, 4)
    got = ctx.ntt_batch(curve, mont, log_n)
    want = O.ntt_batch(curve, mont, log_n)
    assert (got == want).all()


def test_ntt_ligero_2_20_shape(ctx):
    """2^20 coefficients -> 128 x 8192 matrix -> 128 NTTs of size 2^15."""
    curve = "bls12_381"
    n_rows, n_cols, _ = O.ligero_dims(255, 1 << 20)
    assert (n_rows, n_cols) == (128, 8192)
    rows = 8   # a slice of the matrix is enough for parity at this size
    co = O.gen_scalars(curve, 0x5EED0500, rows * n_cols)
    mont = O.f_to_mont(curve, 1, co).reshape(rows, n_cols, 4)
    got = ctx.ntt_batch(curve, mont, 15)
    want = O.ntt_batch(curve, mont, 15)
    assert (got == want).all()


def test_ntt_2_17_linearity_and_dc(ctx):
    """Full config-5 row size (2^17, input 2^15): size-independent properties.
    out[0] = sum of inputs; NTT(a + b) = NTT(a) + NTT(b); one row checked against the oracle."""
    curve = "bls12_381"
    fr = R.FIELDS["bls12_381_fr"]["p"]
    m, lg = 1 << 15, 17
    a = O.gen_scalars(curve, 1, m)
    b = O.gen_scalars(curve, 2, m)
    ai, bi = O.limbs_to_ints(a), O.limbs_to_ints(b)
    s = O.ints_to_limbs([(x + y) % fr for x, y in zip(ai, bi)], 4)
    mat = np.stack([O.f_to_mont(curve, 1, v) for v in (a, b, s)])
    got = ctx.ntt_batch(curve, mat, lg)
    va, vb, vs = (O.fr_from_mont_array(curve, got[k][:64]) for k in range(3))
    assert vs == [(x + y) % fr for x, y in zip(va, vb)]
    assert va[0] == sum(ai) % fr
    want = O.ntt_batch(curve, mat[:1], lg)
    assert (got[0] == want[0]).all()
