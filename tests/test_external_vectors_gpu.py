"""The GPU leg of the three-way check pyref = C++ oracle = HIP on known answers from OUTSIDE this repository
(tests/golden/external.json; CPU legs: tests/test_external_vectors_cpu.py), through the C ABI."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R
from test_external_vectors_cpu import EXT, h, multiples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_published_multiples_through_pc_hip_msm(ctx, curve):
    g, ms = multiples(curve)
    # one-point key and a key of 64 copies of g (k = sum of 64 scalars): the table-free path and the window table
    for copies in (1, 64):
        base = O.points_to_array(curve, [g] * copies)
        srs = ctx.upload_srs(curve, base)
        for table in (False, True):
            if table:
                srs.precompute(min_pairs=1)
            for k, want in ms:
                parts = [k // copies + (1 if i < k % copies else 0) for i in range(copies)]
                got, inf = srs.msm(O.ints_to_limbs(parts, 4))
                assert not inf and O.array_to_points(curve, got)[0] == want, (curve, k, copies, table)
        srs.free()


def test_zcash_compressed_bytes_from_the_device_writer(ctx):
    curve = "bls12_381"
    g, _ = multiples(curve)
    vec = EXT["bls12_381_g1"]["zcash_compressed"]
    pts = [R.ec_mul(curve, int(e["k"]), g) if int(e["k"]) else None for e in vec]
    srs = ctx.upload_srs(curve, O.points_to_array(curve, pts))
    data = bytes(srs.serialize(compressed=True))
    assert int.from_bytes(data[:8], "little") == len(pts)
    for i, e in enumerate(vec):
        assert data[8 + 48 * i: 8 + 48 * (i + 1)].hex() == e["hex"], e["k"]
    # ... and back: the loader decompresses the published encodings to the published points
    srs2, used = ctx.load_serialized_srs(curve, data, compressed=True)
    assert used == len(data)
    assert O.array_to_points(curve, srs2.read(0, len(pts))) == pts
    srs.free()
    srs2.free()


def test_ntt_uses_the_published_roots_of_unity(ctx):
    # a length-2^k NTT of the monomial x evaluates to omega^j: the library's root against the published 2^28 / 2^32-th roots
    for curve, lg_full, key, sub in (("bn254", 28, "bn254_g1", "fr_two_adic_root_of_unity_2p28"),
                                     ("bls12_381", 32, "bls12_381_g1", "fr_two_adic_root_of_unity_2p32"),
                                     ("pallas", 32, "pallas", "scalar_field_root_of_unity_2p32")):
        p = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
        root = h(EXT[key][sub])
        for lg in (4, 10):
            w = pow(root, 1 << (lg_full - lg), p)
            mat = O.fr_mont_array(curve, [0, 1]).reshape(1, 2, 4)
            out = ctx.ntt_batch(curve, mat, lg)
            got = O.fr_from_mont_array(curve, out.reshape(-1, 4))
            assert got == [pow(w, j, p) for j in range(1 << lg)], (curve, lg)
