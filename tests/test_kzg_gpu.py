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


def _int_fr(curve):
    return R.FIELDS[R.CURVES[curve]["fr"]]["p"]


@pytest.mark.parametrize("curve,d", [("bls12_381", 1 << 16), ("bn254", 1 << 14)])
def test_true_srs_end_to_end_trapdoor_check(ctx, curve, d):
    """end_to_end_test (kzg10/mod.rs:546-575) at scale with a TRUE SRS built on the GPU
    (KZG10::setup's g.batch_mul(powers_of_beta), :68-76): commit and open on the GPU, then the
    verifier's pairing equation e(C - v g, h) = e(W, beta h - z h) (:314-333) is checked in G1
    with the known trapdoor:  C - v*g == (beta - z) * W."""
    import torch
    import poly_commit_amd as pc
    p = _int_fr(curve)
    n = d + 1
    beta = O.gen_scalars(curve, 0xBE7A, 1)
    beta_m = O.f_to_mont(curve, 1, beta)[0]
    g = O.gen_bases(curve, 1)[0]
    pw = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.fr_powers(curve, beta_m, n, pw.data_ptr())                       # powers_of_beta
    aw = 2 * O.fq_limbs(curve)
    srs_pts = torch.empty((n, aw), dtype=torch.int64, device="cuda")
    ctx.fixed_base_batch_mul(curve, g, pw.data_ptr(), n, srs_pts.data_ptr())
    srs = ctx.upload_srs(curve, srs_pts.data_ptr(), n=n)
    # spot-check the SRS against host scalar multiplications
    host = srs.read(0, 3)
    b = O.limbs_to_ints(beta)[0]
    for i in range(3):
        want = pc.point_mul(curve, g, O.fr_mont_array(curve, [pow(b, i, p)])[0])
        assert (host[i] == want).all()
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0001, n))
    z = O.gen_scalars(curve, 0x2EE7, 1)
    z_m = O.f_to_mont(curve, 1, z)[0]
    comm, _ = srs.msm(coeffs, montgomery=True)
    q = ctx.witness_poly(curve, coeffs, z_m)
    w, _ = srs.msm(q, montgomery=True)
    v_m = O.poly_eval(curve, coeffs, z_m)
    zi = O.limbs_to_ints(z)[0]
    neg_v = O.fr_mont_array(curve, [(-O.fr_from_mont_array(curve, v_m.reshape(1, 4))[0]) % p])[0]
    lhs = pc.points_sum(curve, np.stack([comm, pc.point_mul(curve, g, neg_v)]))
    rhs = pc.point_mul(curve, w, O.fr_mont_array(curve, [(b - zi) % p])[0])
    assert (lhs == rhs).all() and lhs.any()
    srs.free()


def test_sharded_engine_world1_matches_oracle(ctx):
    """poly_commit_amd/sharded.py with the HIP engine (device-resident buffers, async pipelines)
    on one rank: same commitment / proof as the oracle."""
    import torch
    from poly_commit_amd import sharded
    curve, n = "bls12_381", 5000
    bases = O.gen_bases(curve, n + 1)
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 11, n))
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 12, 1))[0]
    eng = sharded.HipEngine(ctx, curve)
    job = sharded.ShardedKzg(eng, curve)
    job.load_srs_chunk(bases)
    job.set_point(z)
    cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
    f1, f2 = job.commit_async(cdev, n), job.open_async(cdev, n)
    comm, proof = f1.result(), f2.result()
    powers = np.ascontiguousarray(bases[1:])
    rc, want_c = O.kzg_commit(curve, powers, coeffs)
    rc2, want_w = O.kzg_open(curve, powers, coeffs, z)
    assert (comm == want_c).all() and (proof == want_w).all()
    eng.srs.free()


@pytest.mark.parametrize("curve", ["bls12_381", "bn254", "pallas"])
def test_fr_lincomb_matches_bigint(ctx, curve):
    """MarlinKZG10::open's p = sum_j xi_j p_j (marlin_pc/mod.rs:281-287), ragged lengths, host and
    device-resident inputs; followed by the witness division it feeds (:307-312)."""
    import torch
    lens = [1000, 1, 517, 1000, 64, 0]
    polys_i = [R.gen_scalars(curve + "_fr", 0x5EED0700 + j, n) for j, n in enumerate(lens)]
    xi_i = R.gen_scalars(curve + "_fr", 0x5EED0777, len(lens))
    want = R.fr_lincomb(curve + "_fr", polys_i, xi_i)
    polys = [O.fr_mont_array(curve, p) if len(p) else np.zeros((0, 4), dtype=np.uint64) for p in polys_i]
    xi = O.fr_mont_array(curve, xi_i)
    got = ctx.fr_lincomb(curve, polys, xi)
    assert O.fr_from_mont_array(curve, got) == want
    # device-resident polynomials and output; truncated output length
    dev = [torch.from_numpy(p.view(np.int64)).cuda() if len(p) else torch.zeros((1, 4), dtype=torch.int64, device="cuda") for p in polys]
    out = torch.empty((600, 4), dtype=torch.int64, device="cuda")
    ctx.fr_lincomb(curve, [d.data_ptr() for d in dev], xi, n_out=600, out=out.data_ptr(), lens=lens)
    assert O.fr_from_mont_array(curve, out.cpu().numpy().view(np.uint64)) == want[:600]
    # k = 0 gives the zero polynomial
    z = ctx.fr_lincomb(curve, [], np.zeros((0, 4), dtype=np.uint64), n_out=5)
    assert not z.any()


# The one literal input/output vector the reference holds on this path: test_row_mul,
# poly-commit/src/utils.rs:274-286 (Matrix::row_mul, used by Ligero's open, linear_codes/mod.rs:539).
ROW_MUL_ROWS = [[10, 100, 4], [23, 1, 0], [55, 58, 9]]
ROW_MUL_V = [12, 41, 55]
ROW_MUL_WANT = [4088, 4431, 543]


@pytest.mark.parametrize("curve", ["bls12_381", "bn254", "pallas"])
def test_row_mul_reference_vector(ctx, curve):
    """`mat.row_mul(&v) == [4088, 4431, 543]` (utils.rs:274-286) through pc_hip_fr_lincomb, the entry point
    that replaces row_mul; given in the integers, so it holds in every scalar field."""
    rows = [O.fr_mont_array(curve, r) for r in ROW_MUL_ROWS]
    got = ctx.fr_lincomb(curve, rows, O.fr_mont_array(curve, ROW_MUL_V))
    assert O.fr_from_mont_array(curve, got) == ROW_MUL_WANT


@pytest.mark.parametrize("curve", ["bls12_381", "bn254", "pallas"])
def test_poly_eval_matches_oracle(ctx, curve):
    """pc_hip_poly_eval (the up-sweep of the division scan alone) == Horner evaluation of the oracle,
    host and device-resident coefficients, lengths around the chunk sizes of the scan levels."""
    import torch
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xE7A1, 1))[0]
    for n in (1, 2, 7, 8, 9, 127, 128, 129, 1000, 4097, 70000):
        co = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xE7A2 + n, n))
        want = O.poly_eval(curve, co, z)
        assert (ctx.poly_eval(curve, co, z) == want).all(), n
        dev = torch.from_numpy(co.view(np.int64)).cuda()
        assert (ctx.poly_eval(curve, dev.data_ptr(), z, n=n) == want).all(), n
    assert not ctx.poly_eval(curve, np.zeros((0, 4), dtype=np.uint64), z).any()
    # and it is element 0 of the division scan (what the sharded open used to read back)
    co = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xE7A3, 5000))
    scan = ctx.div_scan(curve, co, z)
    assert (scan[0] == ctx.poly_eval(curve, co, z)).all()


def _gold():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def test_golden_ligero_commit_lincomb_msm_many(ctx):
    """The committed fixtures (tests/golden/golden.json, tools/gen_golden.py) for the newer entry points,
    through the C ABI: fused Ligero commit (leaves, nodes, root), open's linear combination, many short MSMs."""
    G = _gold()
    hexs = lambda xs: [int(x, 16) for x in xs]
    for case in G["ligero_commit"]:
        curve = case["curve"]
        mat = np.stack([O.fr_mont_array(curve, hexs(row)) for row in case["matrix"]])
        for v in case["variants"]:
            nodes, leaves = ctx.ligero_commit(curve, np.ascontiguousarray(mat), case["log_n"], col_hash=v["col_hash"],
                                              tree_hash=v["tree_hash"], len_prefix=v["len_prefix"])
            assert [leaves[j].tobytes().hex() for j in range(32)] == v["leaves"]
            assert [nodes[j].tobytes().hex() for j in range(31)] == v["nodes"] and nodes[0].tobytes().hex() == v["root"]
    for case in G["fr_lincomb"]:
        curve = case["curve"]
        polys = [O.fr_mont_array(curve, hexs(q)) for q in case["polys"]]
        got = ctx.fr_lincomb(curve, polys, O.fr_mont_array(curve, hexs(case["xi"])))
        assert O.fr_from_mont_array(curve, got) == hexs(case["result"])
    for case in G["msm_many"]:
        curve = case["curve"]
        bases = O.points_to_array(curve, [None if p is None else (int(p[0], 16), int(p[1], 16)) for p in case["bases"]])
        srs = ctx.upload_srs(curve, bases)
        rows = np.stack([O.ints_to_limbs(hexs(row), 4) for row in case["rows"]])
        got, inf = srs.msm_many(np.ascontiguousarray(rows))
        pts = O.array_to_points(curve, got)
        for k, want in enumerate(case["results"]):
            assert pts[k] == (None if want is None else (int(want[0], 16), int(want[1], 16)))
        assert list(inf) == [w is None for w in case["results"]]
        srs.free()


@pytest.mark.parametrize("curve,compressed", [("bls12_381", True), ("bls12_381", False), ("bn254", True), ("bn254", False), ("pallas", False),
                                              ("pallas", True)])
def test_srs_from_ark_serialize_bytes(ctx, curve, compressed):
    """pc_hip_srs_load_serialized: the head of a serialized kzg10::UniversalParams (Vec<G1Affine>: u64 length + points,
    kzg10/data_structures.rs:57-112) decoded on the device into a resident SRS; a commitment over it equals the oracle's."""
    import poly_commit_amd as pc
    n = 300
    pts = R.gen_bases(curve, n)
    pts[7] = R.ec_neg(curve, pts[7])
    data = R.ser_g1_vec(curve, pts, compressed) + b"\x01\x02\x03trailing fields of the structure"
    srs, used = ctx.load_serialized_srs(curve, data, compressed)
    assert srs.n == n and used == len(R.ser_g1_vec(curve, pts, compressed))
    arr = O.points_to_array(curve, pts)
    assert (srs.read(0, n) == arr).all()
    s = O.gen_scalars(curve, 0x5E71A, n)
    assert (srs.msm(s)[0] == O.msm_pippenger(curve, arr, s, 8, 1)).all()
    # the writer: the resident points back to exactly the bytes CanonicalSerialize wrote (pc_hip_srs_serialize), whole and in part
    assert srs.serialize(compressed=compressed) == R.ser_g1_vec(curve, pts, compressed)
    assert srs.serialize(5, 20, compressed=not compressed) == R.ser_g1_vec(curve, pts[5:25], not compressed)
    srs.free()
    # only the first 100 points (a committer key shorter than the ceremony)
    srs, used2 = ctx.load_serialized_srs(curve, data, compressed, max_points=100)
    assert srs.n == 100 and used2 == used and (srs.read(0, 100) == arr[:100]).all()
    srs.free()
    with pytest.raises(pc.PcHipError):                       # truncated
        ctx.load_serialized_srs(curve, data[:used // 2], compressed)
    bad = bytearray(data)
    bad[8 + (47 if curve == "bls12_381" else 0)] ^= 1         # first x off by one: not on the curve / wrong point
    if not compressed:
        with pytest.raises(pc.PcHipError):
            ctx.load_serialized_srs(curve, bytes(bad), compressed)


def test_srs_with_infinity_round_trips_through_the_writer(ctx):
    """Infinity in a key (flag 0x40, x = y = 0) survives load -> serialize in every encoding, Pallas compressed included."""
    for curve in ("bls12_381", "bn254", "pallas"):
        pts = R.gen_bases(curve, 9)
        pts[3] = None
        pts[6] = R.ec_neg(curve, pts[6])
        for compressed in (False, True):
            data = R.ser_g1_vec(curve, pts, compressed)
            srs, used = ctx.load_serialized_srs(curve, data, compressed)
            assert used == len(data) and srs.serialize(compressed=compressed) == data
            srs.free()
