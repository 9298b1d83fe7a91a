"""The GLV form of the window table (pc_hip_srs_precompute_ex, PC_HIP_TABLE_GLV): half the table, every scalar split on the device
into k1 + k2 lambda, the digits of k2 in a second bucket set over the SAME table points, phi applied once to that set's reduced sum.
Every result must be the point the full table and the table-free path give (kzg10/mod.rs:175-178, :255-258 through either)."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381", "bn254", "pallas"]


@pytest.mark.parametrize("curve", CURVES)
def test_glv_table_msm_sizes_and_edge_scalars(ctx, curve):
    nmax = 4200
    r = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
    bases = O.gen_bases(curve, nmax)
    bases[7] = 0
    bases[100] = bases[99]
    srs = ctx.upload_srs(curve, bases)
    full = srs.precompute(min_pairs=1, glv=False).bytes_resident()["window_tables"]
    srs.precompute(min_pairs=1, glv=True)
    half = srs.bytes_resident()["window_tables"]
    assert 0 < half <= 0.6 * full, (half, full)
    for n in (1, 2, 31, 32, 33, 257, 4097, 4200):
        sc = O.gen_scalars(curve, 0x61F0 + n, n)
        got, inf = srs.msm(sc)
        assert (got == O.msm_pippenger(curve, bases, sc, 8, 2)).all() and not inf, (curve, n)
        assert ctx.last_msm_shape()["window_table"]
    n = 1500
    rnd = O.gen_scalars(curve, 99, n)
    cases = {
        "zeros": np.zeros((n, 4), dtype=np.uint64), "ones": O.ints_to_limbs([1] * n, 4), "r-1": O.ints_to_limbs([r - 1] * n, 4),
        "same": np.repeat(rnd[:1], n, axis=0), "half-zero": np.where((np.arange(n) % 2 == 0)[:, None], rnd, 0).astype(np.uint64),
        "low-hamming": O.ints_to_limbs([1 << (i % 253) for i in range(n)], 4),
        "small": O.ints_to_limbs([i for i in range(n)], 4), "128-bit": O.ints_to_limbs([(1 << 128) - 1 - i for i in range(n)], 4),
    }
    for name, sc in cases.items():
        sc = np.ascontiguousarray(sc)
        got, inf = srs.msm(sc)
        want = O.msm_pippenger(curve, bases, sc, 8, 2)
        assert (got == want).all() and inf == (not want.any()), (curve, name)
    # Montgomery-form scalars, a base offset, min(len) truncation
    sc = O.gen_scalars(curve, 5, 3000)
    got, _ = srs.msm(O.f_to_mont(curve, 1, sc), base_offset=1500, montgomery=True)
    assert (got == O.msm_pippenger(curve, np.ascontiguousarray(bases[1500:]), np.ascontiguousarray(sc[:nmax - 1500]), 8, 2)).all()
    srs.free()


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_glv_table_kzg_commit_open_and_batch(ctx, curve):
    import torch
    n = (1 << 15) + 1
    powers = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, powers)
    srs.precompute(glv=True)
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0015, n))
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 3, 1))[0]
    assert (srs.msm(coeffs, montgomery=True)[0] == O.kzg_commit(curve, powers, coeffs)[1]).all()
    assert (srs.kzg_open(coeffs, z)[0] == O.kzg_open(curve, powers, coeffs, z)[1]).all()
    # the batch of MarlinKZG10::commit (many-MSM passes over the key's GLV table: two bucket sets per polynomial)
    k = 11
    host = [O.gen_scalars(curve, 0x5EED0100 + j, n) for j in range(k)]
    polys = [torch.from_numpy(O.f_to_mont(curve, 1, h).view(np.int64)).cuda() for h in host]
    comms = srs.msm_batch([t.data_ptr() for t in polys], [n] * k)
    for j in range(k):
        assert (comms[j] == O.msm_pippenger(curve, powers, host[j], 8, 2)).all(), j
    srs.free()


def test_glv_table_randomised_against_the_full_table(ctx):
    """Random sizes and scalar distributions: GLV table == full table == table-free, the three through the same key."""
    import random
    rnd = random.Random(2024)
    for curve in CURVES:
        nmax = 1 << 13
        bases = O.gen_bases(curve, nmax)
        keys = [ctx.upload_srs(curve, bases) for _ in range(3)]
        keys[1].precompute(min_pairs=1, glv=False)
        keys[2].precompute(min_pairs=1, glv=True)
        r = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
        for _ in range(12):
            n = rnd.randrange(1, nmax + 1)
            kind = rnd.randrange(4)
            if kind == 0:
                sc = O.gen_scalars(curve, rnd.randrange(1 << 30), n)
            elif kind == 1:
                sc = O.ints_to_limbs([rnd.randrange(1 << rnd.randrange(1, 255)) % r for _ in range(n)], 4)
            elif kind == 2:
                vals = [rnd.randrange(r) for _ in range(5)]
                sc = O.ints_to_limbs([vals[rnd.randrange(5)] for _ in range(n)], 4)
            else:
                sc = O.ints_to_limbs([(r - 1 - rnd.randrange(1 << 20)) for _ in range(n)], 4)
            outs = [k.msm(sc)[0] for k in keys]
            assert (outs[0] == outs[1]).all() and (outs[0] == outs[2]).all(), (curve, n, kind)
        for k in keys:
            k.free()
