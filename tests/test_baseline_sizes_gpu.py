"""GPU parity at BASELINE.json's OWN sizes (round-1 verdict: every headline number except 2^20
BLS12-381 was a timing of unchecked output).

  configs[1] / north star   KZG commit+open, BLS12-381, degree 2^24 (window table c = 22 and table-free)
  configs[2]                64 x MarlinKZG10<Bn254> commits of degree 2^20 (pc_hip_msm_batch)
  configs[3]                InnerProductArgPC over Pallas, n = 2^22: all 22 halving rounds
  configs[4]                Ligero over BLS12-381 Fr, 2^24 coefficients: 512 NTTs of 2^15 -> 2^17

Full-size checks use what the domain offers that does not need a second full-size computation:
a TRUE structured reference string beta^i * g built on the GPU makes commitment and proof
closed-form (C = p(beta) g, W = q(beta) g -- the verifier's pairing equation, kzg10/mod.rs:314-333,
with the trapdoor known); all-equal scalars give a geometric series; linearity ties a batch
together.  Where the oracle finishes in tens of seconds on the box's host cores it is run as well
(one 2^24 MSM, the 2^22 IPA rounds, 4 of the 64 batch MSMs, every NTT row)."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu
CORES = os.cpu_count() or 8


def _p(curve):
    return R.FIELDS[R.CURVES[curve]["fr"]]["p"]


def _oracle_mul(curve, g, k):
    """k * g by the CPU oracle's double-and-add (NOT the product's host routine): the closed forms below are computed outside the library under test."""
    fr = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
    return O.msm_naive(curve, np.ascontiguousarray(g).reshape(1, -1), O.ints_to_limbs([k % fr], 4))


def _mont1(curve, v):
    return O.fr_mont_array(curve, [v % _p(curve)])[0]


def _fr_int(curve, limbs_mont):
    return O.fr_from_mont_array(curve, np.asarray(limbs_mont, dtype=np.uint64).reshape(1, 4))[0]


def test_kzg_commit_open_deg_2p24_bls12_381_true_srs(ctx):
    """MarlinKZG10<Bls12_381> commit + open at degree 2^24 (hiding off): commit = MSM of 2^24 + 1 pairs
    (kzg10/mod.rs:175-178), open = witness polynomial on the device (:217-240) + MSM of 2^24 pairs (:255-258),
    on the default window-table path (c = 22, 12 digits per scalar) and table-free."""
    import torch
    import poly_commit_amd as pc
    curve, d = "bls12_381", 1 << 24
    n, p = d + 1, _p("bls12_381")
    beta = O.limbs_to_ints(O.gen_scalars(curve, 0xBE7A24, 1))[0]
    zi = O.limbs_to_ints(O.gen_scalars(curve, 0x2EE724, 1))[0]
    beta_m, z_m = _mont1(curve, beta), _mont1(curve, zi)
    g = O.gen_bases(curve, 1)[0]
    # KZG10::setup (kzg10/mod.rs:68-76): powers of beta, then g.batch_mul(powers) -- both on the device
    pw = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.fr_powers(curve, beta_m, n, pw.data_ptr())
    pts = torch.empty((n, 2 * O.fq_limbs(curve)), dtype=torch.int64, device="cuda")
    ctx.fixed_base_batch_mul(curve, g, pw.data_ptr(), n, pts.data_ptr())
    srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)
    del pw
    for i in (0, 1, 2, 12345, d - 1, d):          # spot-check the SRS against host scalar multiplications
        assert (srs.read(i, 1)[0] == _oracle_mul(curve, g, (pow(beta, i, p)))).all(), i

    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0024, n))
    cdev = torch.from_numpy(coeffs.view(np.int64)).cuda()
    p_beta = _fr_int(curve, O.poly_eval(curve, coeffs, beta_m))
    p_z = _fr_int(curve, O.poly_eval(curve, coeffs, z_m))
    want_c = _oracle_mul(curve, g, (p_beta))
    want_w = _oracle_mul(curve, g, ((p_beta - p_z) * pow(beta - zi, -1, p)))

    qdev = torch.empty((n - 1, 4), dtype=torch.int64, device="cuda")
    ctx.witness_poly(curve, cdev.data_ptr(), z_m, out=qdev.data_ptr(), n=n)
    # device evaluation agrees with the oracle's Horner
    assert _fr_int(curve, ctx.poly_eval(curve, cdev.data_ptr(), z_m, n=n)) == p_z

    def commit_open(tag):
        comm, inf_c = srs.msm(cdev, n=n, montgomery=True)
        proof, inf_w = srs.msm(qdev, n=n - 1, montgomery=True)
        assert not inf_c and not inf_w
        assert (comm == want_c).all(), f"commitment differs from p(beta) g ({tag})"
        assert (proof == want_w).all(), f"opening proof differs from q(beta) g ({tag})"
        return comm

    commit_open("table-free")
    srs.precompute()                               # the bench's default: c = 22 at this size
    comm = commit_open("window table")

    # all-equal scalars: one bucket per digit holds every base (the distribution the chunked accumulate exists for)
    s = O.limbs_to_ints(O.gen_scalars(curve, 0x5A3E, 1))[0]
    same = torch.from_numpy(np.ascontiguousarray(np.repeat(O.fr_mont_array(curve, [s]), n, axis=0)).view(np.int64)).cuda()
    geo = s * (pow(beta, n, p) - 1) * pow(beta - 1, -1, p) % p
    got, _ = srs.msm(same, n=n, montgomery=True)
    assert (got == _oracle_mul(curve, g, (geo))).all(), "all-equal scalars"
    del same

    # and one full-size comparison with the oracle's Pippenger on the host cores
    bases_host = srs.read(0, n)
    want = O.msm_pippenger(curve, bases_host, O.f_from_mont(curve, 1, coeffs), CORES, 2)
    assert (comm == want).all(), "commitment differs from the oracle MSM"
    srs.free()


@pytest.mark.parametrize("table", [False, True])
def test_marlin_batch_64_polys_deg_2p20_bn254(ctx, table):
    """configs[2] on one GPU: MarlinKZG10<Bn254>::commit's loop over 64 polynomials of degree 2^20
    (marlin_pc/mod.rs:192-237) as one pc_hip_msm_batch against the same resident powers."""
    import torch
    import poly_commit_amd as pc
    curve, n, k = "bn254", (1 << 20) + 1, 64
    p = _p(curve)
    bases = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, bases)
    if table:
        srs.precompute()
    host = [O.gen_scalars(curve, 0x5EED0100 + j, n) for j in range(k)]
    polys = [torch.from_numpy(O.f_to_mont(curve, 1, h).view(np.int64)).cuda() for h in host]
    comms = srs.msm_batch([t.data_ptr() for t in polys], [n] * k)
    for j in (0, 21, 42, 63):
        assert (comms[j] == O.msm_pippenger(curve, bases, host[j], CORES, 2)).all(), j
    # all 64 tied together by linearity: sum_j xi_j C_j == commit(sum_j xi_j p_j)  (what MarlinKZG10::open relies on)
    xi = R.gen_scalars(curve + "_fr", 0x5EED0777, k)
    xi_m = O.fr_mont_array(curve, xi)
    lhs = O.msm_naive(curve, np.ascontiguousarray(comms), O.ints_to_limbs(xi, 4))     # by the oracle, not the library's host routines
    comb = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.fr_lincomb(curve, [t.data_ptr() for t in polys], xi_m, n_out=n, out=comb.data_ptr(), lens=[n] * k)
    rhs, _ = srs.msm(comb, n=n, montgomery=True)
    assert (lhs == rhs).all() and rhs.any()
    srs.free()


def test_ipa_open_rounds_pallas_2p22(ctx):
    """configs[3]: InnerProductArgPC over Pallas, d + 1 = 2^22: cm_commit (ipa_pc/mod.rs:54-72) and a whole open
    (:475-723, hiding off: combination, Fiat-Shamir transcript, all 22 halving rounds) against the oracle's
    restatement -- Proof{l_vec, r_vec, final_comm_key, c} bit for bit."""
    import torch
    from poly_commit_amd import ipa
    curve, n = "pallas", 1 << 22
    lg = 22
    key = O.gen_bases(curve, n + 1)
    comm_key, h_prime = np.ascontiguousarray(key[:n]), np.ascontiguousarray(key[n])
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE22, n))
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B22, 1))[0]
    # the Pedersen commitment itself (config 4's commit)
    srs = ctx.upload_srs(curve, comm_key)
    cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
    comm, _ = srs.msm(cdev, n=n, montgomery=True)
    assert (comm == O.msm_pippenger(curve, comm_key, O.f_from_mont(curve, 1, coeffs), CORES, 1)).all()
    # the committer key stays resident with its once-per-key tables (window table: round 1's MSMs; fold table: the first key fold),
    # as in bench.py's workloads.ipa; the commitment over the window table is the same point
    srs.precompute()
    srs.precompute_fold()
    assert (srs.msm(cdev, n=n, montgomery=True)[0] == comm).all()
    # open(): one polynomial, opening challenge from the (caller's) sponge, random-oracle challenges from the
    # transcript (ipa_pc/mod.rs:615-625, 681-688) -- the whole Proof{l_vec, r_vec, final_comm_key, c}
    xi = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A122, 1))
    (l, r, fk, c), rc0 = ipa.ipa_open(ctx, curve, srs, h_prime, [cdev.data_ptr()], [n], [comm], point, xi)
    srs.free()
    assert l.shape[0] == lg
    # oracle: same combination (xi * p, xi * C), same first challenge, rounds with its own transcript
    xi_i = O.fr_from_mont_array(curve, xi)[0]
    comb = O.fr_mont_array(curve, [v * xi_i % _p(curve) for v in O.fr_from_mont_array(curve, coeffs)])
    ccomm = O.points_to_array(curve, [R.ec_mul(curve, xi_i, O.array_to_points(curve, comm)[0])])[0]
    want_rc0 = O.ipa_first_challenge(curve, ccomm, point, O.poly_eval(curve, comb, point))
    assert (rc0 == want_rc0).all()
    hp = O.points_to_array(curve, [R.ec_mul(curve, _fr_int(curve, want_rc0), O.array_to_points(curve, h_prime)[0])])[0]
    want_l, want_r, want_key, want_c, _ = O.ipa_rounds_fs(curve, comm_key, comb, point, hp, want_rc0, threads=CORES)
    assert (l == want_l).all() and (r == want_r).all()
    assert (fk == want_key).all() and (c == want_c).all()


def test_ligero_encode_2p24_coeffs_all_rows(ctx):
    """configs[4]: 2^24 BLS12-381 Fr coefficients -> 512 x 32768 matrix -> 512 forward NTTs of size 2^17
    (linear_codes/mod.rs:118-138, utils.rs:112-127): every row against the oracle's NTT, every row's
    out[0] = sum of its coefficients, and Horner evaluations at omega^j at random positions
    (test_reed_solomon's statement, linear_codes/utils.rs:324-329)."""
    curve = "bls12_381"
    fr = R.CURVES[curve]["fr"]
    p = R.FIELDS[fr]["p"]
    poly_len = 1 << 24
    n_rows, n_cols, _ = O.ligero_dims(255, poly_len)
    assert (n_rows, n_cols) == (512, 32768)
    log_n = 17
    co = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0500, poly_len)).reshape(n_rows, n_cols, 4)
    got = ctx.ntt_batch(curve, co, log_n)
    want = O.ntt_batch(curve, co, log_n, threads=CORES)
    assert (got == want).all()
    rnd = random.Random(17)
    w = O.fr_from_mont_array(curve, O.root_of_unity(curve, log_n).reshape(1, 4))[0]
    assert pow(w, 1 << log_n, p) == 1 and pow(w, 1 << (log_n - 1), p) != 1
    for _ in range(24):
        r, j = rnd.randrange(n_rows), rnd.randrange(1 << log_n)
        row = O.fr_from_mont_array(curve, co[r])
        if _ % 8 == 0:
            j = 0                                   # out[r][0] = sum of the row
        assert O.fr_from_mont_array(curve, got[r, j].reshape(1, 4))[0] == R.poly_eval(fr, row, pow(w, j, p)), (r, j)


@pytest.mark.parametrize("curve,lg,table", [("bn254", 12, False), ("bn254", 16, True), ("bls12_381", 16, False),
                                            ("pallas", 10, True)])
def test_msm_call_just_below_the_srs_length(ctx, curve, lg, table):
    """The KZG open shape (kzg10/mod.rs:255-258): n - 1 pairs over a 2^k SRS runs with the smaller window of
    its bracket, i.e. more digits per scalar than the SRS length itself (round-1 advisor finding: the
    plan's entry buffer was sized for powers of two only)."""
    n_srs = 1 << lg
    b = O.gen_bases(curve, n_srs)
    srs = ctx.upload_srs(curve, b)
    if table:
        srs.precompute()
    for n, off in ((n_srs - 1, 1), (n_srs - 1, 0), (n_srs // 2 - 1, 3), (31, n_srs - 31)):
        s = O.gen_scalars(curve, 77 + n, n)
        got, _ = srs.msm(s, base_offset=off)
        assert (got == O.msm_pippenger(curve, np.ascontiguousarray(b[off:off + n]), s, 8, 1)).all(), (n, off)
    srs.free()
