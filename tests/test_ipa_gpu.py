"""GPU parity for InnerProductArgPC (ipa_pc/mod.rs) against the oracle's restatement: the halving rounds with
supplied challenges, and whole openings -- combination, Fiat-Shamir transcript (Blake2s over ark-serialize
bytes), rounds -- through both host layers above the C ABI (the C++ mirror and the Python harness)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


# fkb: size from which the rounds keep the key fixed and fold per-base factors (pc_hip_ipa_key_scalars);
# None = the default, 0 = never (every round folds the key with pc_hip_ec_fold), 64 = switch in the middle
@pytest.mark.parametrize("curve,n,fkb", [("pallas", 1 << 10, None), ("pallas", 1 << 10, 0), ("pallas", 1 << 13, 64),
                                         ("pallas", 1 << 14, 0), ("pallas", 4, None), ("bls12_381", 1 << 8, 16),
                                         ("bls12_381", 1 << 13, 0), ("bn254", 1 << 9, None), ("bn254", 1 << 9, 0)])
def test_ipa_open_rounds(ctx, curve, n, fkb):
    import torch
    from poly_commit_amd import ipa
    lg = n.bit_length() - 1
    key = O.gen_bases(curve, n + 1)
    comm_key, h_prime = key[:n], key[n]
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE, n))
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B, 1))[0]
    ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1, lg))
    want_l, want_r, want_key, want_c = O.ipa_rounds(curve, np.ascontiguousarray(comm_key), coeffs, point,
                                                    np.ascontiguousarray(h_prime), ch)
    for python_loop in (False, True):      # the library's own loop (pc_hip_ipa_open_rounds) and the same sequence driven round by round from Python
        it = iter(range(lg))
        cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
        l, r, fk, c = ipa.ipa_open_rounds(ctx, curve, comm_key, cdev, n, point, h_prime, lambda L, R_: ch[next(it)], fixed_key_below=fkb,
                                          python_loop=python_loop)
        assert (l == want_l).all() and (r == want_r).all(), python_loop
        assert (fk == want_key).all() and (c == want_c).all(), python_loop


@pytest.mark.parametrize("curve,n,fkb,tables", [("pallas", 1 << 13, 64, True), ("pallas", 1 << 13, 64, False), ("bn254", 1 << 12, 0, True),
                                                ("bls12_381", 1 << 11, 16, True), ("pallas", 1 << 9, None, True)])
def test_ipa_open_rounds_on_a_resident_key(ctx, curve, n, fkb, tables):
    """The committer key stays resident and untouched across openings: round 1's MSMs run on it (window table), its first fold goes
    out of place into a new key (pc_hip_ec_fold_from) -- from the key's fold table when pc_hip_srs_precompute_fold built one, by the
    ladder otherwise.  Two openings in a row on the same resident key, each bit for bit the oracle's."""
    import torch
    from poly_commit_amd import ipa
    lg = n.bit_length() - 1
    key = O.gen_bases(curve, n + 1)
    key[5] = 0                                                            # an infinity among the generators (both halves)
    key[n // 2 + 7] = 0
    comm_key, h_prime = np.ascontiguousarray(key[:n]), key[n]
    srs = ctx.upload_srs(curve, comm_key)
    if tables:
        srs.precompute(min_pairs=1)
        srs.precompute_fold()
    before = srs.read(0, n).copy()
    for rep in range(2):
        coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE + rep, n))
        point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B + rep, 1))[0]
        ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1 + rep, lg))
        want_l, want_r, want_key, want_c = O.ipa_rounds(curve, comm_key, coeffs, point, np.ascontiguousarray(h_prime), ch)
        it = iter(range(lg))
        cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
        l, r, fk, c = ipa.ipa_open_rounds(ctx, curve, srs, cdev, n, point, h_prime, lambda L, R_: ch[next(it)], fixed_key_below=fkb)
        assert (l == want_l).all() and (r == want_r).all() and (fk == want_key).all() and (c == want_c).all()
    assert (srs.read(0, n) == before).all()                               # the committer key itself is as it was
    srs.free()


@pytest.mark.parametrize("curve,n,fkbs", [("pallas", 1 << 14, (1 << 12, 1 << 12, 1 << 13)), ("pallas", 1 << 13, (1 << 13, 1 << 13)),
                                          ("bn254", 1 << 13, (1 << 12, 1 << 12))])
def test_ipa_late_rounds_on_the_cached_fixed_key(ctx, curve, n, fkbs):
    """From 2^12 points up the fixed key of the late rounds is a key object of its own that belongs to the committer key
    (pc_hip_ipa_open_rounds): the working key's points are copied into it and its window table is refilled by every opening.  First
    opening (object + table built), a second with the same switch (refilled in place), a third with another switch (object replaced;
    n0 = n: no fold before the switch) -- each bit for bit the oracle's; the object is the committer key's and goes with it."""
    import torch
    from poly_commit_amd import ipa
    lg = n.bit_length() - 1
    key = O.gen_bases(curve, n + 1)
    key[3] = 0
    comm_key, h_prime = np.ascontiguousarray(key[:n]), key[n]
    keys_before = ctx.bytes_resident()["n_keys"]
    srs = ctx.upload_srs(curve, comm_key)
    srs.precompute(min_pairs=1)
    srs.precompute_fold()
    for rep, fkb in enumerate(fkbs):
        if rep and rep == len(fkbs) - 1:
            ctx.trim()                                                    # drops the cached working and fixed keys: the last opening makes them again
        coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xF1CED + rep, n))
        point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B0 + rep, 1))[0]
        ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A2 + rep, lg))
        want = O.ipa_rounds(curve, comm_key, coeffs, point, np.ascontiguousarray(h_prime), ch)
        it = iter(range(lg))
        cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
        got = ipa.ipa_open_rounds(ctx, curve, srs, cdev, n, point, h_prime, lambda L, R_: ch[next(it)], fixed_key_below=fkb, python_loop=False)
        assert all((a == b).all() for a, b in zip(got, want)), (rep, fkb)
    # the committer key, its working key and its fixed key -- or the committer key alone when it is itself the fixed key (no fold before
    # the switch and a window table of its own: nothing is copied)
    assert ctx.bytes_resident()["n_keys"] >= keys_before + (2 if fkbs[-1] < n else 1)
    srs.free()
    assert ctx.bytes_resident()["n_keys"] == keys_before


@pytest.mark.parametrize("curve,n,fkb,levels,w", [("pallas", 1 << 13, 64, 2, 2), ("pallas", 1 << 13, 64, 2, 3), ("pallas", 1 << 13, 64, 2, 4),
                                                  ("pallas", 1 << 12, 1 << 10, 2, 4), ("pallas", 1 << 12, 1 << 11, 2, 4), ("bn254", 1 << 11, 16, 2, 3),
                                                  ("bls12_381", 1 << 10, 16, 2, 4), ("bls12_381", 1 << 10, 16, 1, 4), ("pallas", 1 << 12, 0, 1, 3),
                                                  ("pallas", 8, 0, 2, 4), ("pallas", 1 << 12, 64, 2, 5), ("bn254", 1 << 10, 16, 1, 5), ("pallas", 1 << 16, None, 0, 0)])
def test_ipa_open_rounds_with_the_general_fold_table(ctx, curve, n, fkb, levels, w):
    """pc_hip_srs_precompute_fold_ex: the fold table in its general form.  Two levels: round 1 leaves the key alone, round 2's four MSMs run
    on the committer key (linearity), pc_hip_ec_fold2_from then gives the key after both folds out of the table (width-w NAF digits,
    three terms).  One level with wider digits: the first fold as before with fewer additions.  levels = w = 0: the library's choice.
    Infinities sit in all four quarters of the key; two openings in a row on the same resident key; every proof bit for bit the oracle's
    (which folds the key round by round, ipa_pc/mod.rs:699-707); the committer key itself is untouched."""
    import torch
    from poly_commit_amd import ipa
    lg = n.bit_length() - 1
    key = O.gen_bases(curve, n + 1)
    for j in (1, n // 4 + 2, n // 2 + 3, 3 * n // 4 + 1):
        key[j] = 0
    comm_key, h_prime = np.ascontiguousarray(key[:n]), key[n]
    srs = ctx.upload_srs(curve, comm_key)
    srs.precompute(min_pairs=1)
    srs.precompute_fold(levels, w)
    got_levels, got_w = srs.fold_table_info()
    assert (got_levels, got_w) == (levels, w) if levels else (got_levels == 2 and 2 <= got_w <= 4)
    rows = 131 << (got_w - 2)
    assert srs.bytes_resident()["fold_table"] == rows * (n - (n >> got_levels)) * comm_key.shape[1] * 8
    before = srs.read(0, n).copy()
    for rep in range(2):
        coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE + rep, n))
        point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B + rep, 1))[0]
        ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1 + rep, lg))
        want_l, want_r, want_key, want_c = O.ipa_rounds(curve, comm_key, coeffs, point, np.ascontiguousarray(h_prime), ch)
        it = iter(range(lg))
        cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
        tm = {}
        l, r, fk, c = ipa.ipa_open_rounds(ctx, curve, srs, cdev, n, point, h_prime, lambda L, R_: ch[next(it)], fixed_key_below=fkb, timings=tm,
                                          python_loop=(rep == 1))      # the library's loop, then the round-by-round one
        assert (l == want_l).all() and (r == want_r).all() and (fk == want_key).all() and (c == want_c).all()
        kinds = tm.get("ec_fold_kind", [])
        limit = ipa.FIXED_KEY_BELOW if fkb is None else fkb
        if got_levels == 2 and n // 2 > limit:
            assert kinds[:2] == ["deferred", "table2"], kinds
        elif got_levels == 1 and n > limit:
            assert kinds[0] == "table1", kinds
    assert (srs.read(0, n) == before).all()
    srs.free()


def test_ipa_rounds_randomised_differential():
    """tools/ipa_fuzz.py for a few seconds: random curve, size, fold-table form, fixed-key switch, resident / host key, infinities among
    the generators, zero coefficients -- whole openings against the oracle (1343 openings with 209 two-level and 280 one-level table
    folds in 240 s without a mismatch on the round-6 library)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ipa_fuzz.py"), "15", "20260930"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_ipa_open_rounds_one_call_edge_cases(ctx):
    """pc_hip_ipa_open_rounds: n = 1 (no round: final key = key[0], c = the coefficient), a prefix of a longer resident key, an exception
    in the caller's challenge function (reported after the loop, nothing left in flight), invalid arguments."""
    import torch
    import poly_commit_amd as pc
    curve = "pallas"
    key = O.gen_bases(curve, 65)
    srs = ctx.upload_srs(curve, np.ascontiguousarray(key[:64]))
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 3, 1))[0]
    co = O.f_to_mont(curve, 1, O.gen_scalars(curve, 4, 64))
    cdev = torch.from_numpy(co[:1].view(np.int64).copy()).cuda()
    l, r, fk, c = srs.ipa_open_rounds(cdev.data_ptr(), 1, point, key[64], lambda L, R_: None)
    assert len(l) == 0 and (fk == key[0]).all() and (c == co[0]).all()
    for n in (16, 64):                                   # a prefix of the resident key, and all of it
        ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 5, 6))
        want = O.ipa_rounds(curve, np.ascontiguousarray(key[:n]), np.ascontiguousarray(co[:n]), point, np.ascontiguousarray(key[64]), ch[:n.bit_length() - 1])
        it = iter(range(6))
        cdev = torch.from_numpy(co[:n].view(np.int64).copy()).cuda()
        got = srs.ipa_open_rounds(cdev.data_ptr(), n, point, key[64], lambda L, R_: ch[next(it)], fixed_key_below=4)
        assert all((a == b).all() for a, b in zip(got, want)), n
    cdev = torch.from_numpy(co.view(np.int64).copy()).cuda()
    with pytest.raises(ZeroDivisionError):
        srs.ipa_open_rounds(cdev.data_ptr(), 64, point, key[64], lambda L, R_: 1 // 0)
    it = iter(range(6))
    cdev = torch.from_numpy(co.view(np.int64).copy()).cuda()
    got = srs.ipa_open_rounds(cdev.data_ptr(), 64, point, key[64], lambda L, R_: ch[next(it)])      # the context is as usable as before
    want = O.ipa_rounds(curve, np.ascontiguousarray(key[:64]), co, point, np.ascontiguousarray(key[64]), ch)
    assert all((a == b).all() for a, b in zip(got, want))
    with pytest.raises(pc.PcHipError):
        srs.ipa_open_rounds(cdev.data_ptr(), 48, point, key[64], lambda L, R_: ch[0])             # not a power of two
    with pytest.raises(pc.PcHipError):
        srs.ipa_open_rounds(cdev.data_ptr(), 128, point, key[64], lambda L, R_: ch[0])            # longer than the key
    srs.free()


def test_fold2_from_without_a_two_level_table_is_the_two_folds(ctx):
    """pc_hip_ec_fold2_from on a key with no table / a one-level table: the two folds one after the other, the same key as two calls."""
    curve, n = "pallas", 1 << 10
    key = np.ascontiguousarray(O.gen_bases(curve, n))
    u = O.f_to_mont(curve, 1, O.gen_scalars(curve, 77, 2))
    srs = ctx.upload_srs(curve, key)
    want = srs.fold_from(n // 2, u[0])
    want.ec_fold(n // 4, u[1])
    ref = want.read(0, n // 4).copy()
    want.free()
    for form in (None, (1, 3), (2, 2)):
        if form:
            srs.precompute_fold(*form)
        k2 = srs.fold2_from(n // 4, u[0], u[1])
        assert (k2.read(0, n // 4) == ref).all(), form
        k2.free()
    srs.free()


def _open_inputs(curve, n, k=2):
    key = O.gen_bases(curve, n + 1)
    comm_key, h = np.ascontiguousarray(key[:n]), np.ascontiguousarray(key[n])
    lens = [n - 3 * j for j in range(k)]
    polys = [O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x0FE0 + j, m)) for j, m in enumerate(lens)]
    comms = [O.msm_pippenger(curve, comm_key, O.f_from_mont(curve, 1, q), 8, 1) for q in polys]
    xi = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x0FE9, k))
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x0FEA, 1))[0]
    return comm_key, h, polys, comms, xi, point


def _oracle_open(curve, comm_key, h, polys, comms, xi, point):
    """The same opening through the oracle: combination in Python big ints, transcript + rounds in oracle.cpp."""
    fr = R.CURVES[curve]["fr"]
    n = len(comm_key)
    comb_i = R.fr_lincomb(fr, [O.fr_from_mont_array(curve, q) for q in polys], O.fr_from_mont_array(curve, xi))
    comb_i += [0] * (n - len(comb_i))
    ccomm = None
    for cm, x in zip(comms, O.fr_from_mont_array(curve, xi)):
        ccomm = R.ec_add(curve, ccomm, R.ec_mul(curve, x, O.array_to_points(curve, cm)[0]))
    comb = O.fr_mont_array(curve, comb_i)
    v = O.poly_eval(curve, comb, point)
    rc0 = O.ipa_first_challenge(curve, O.points_to_array(curve, [ccomm])[0], point, v)
    h_prime = O.points_to_array(curve, [R.ec_mul(curve, O.fr_from_mont_array(curve, rc0.reshape(1, 4))[0], O.array_to_points(curve, h)[0])])[0]
    return O.ipa_rounds_fs(curve, comm_key, comb, point, h_prime, rc0, threads=os.cpu_count() or 8)


@pytest.mark.parametrize("curve,n", [("pallas", 1 << 10), ("bn254", 1 << 6), ("bls12_381", 1 << 7)])
def test_ipa_open_whole_proof_python_host(ctx, curve, n):
    """Proof{l_vec, r_vec, final_comm_key, c} of InnerProductArgPC::open (ipa_pc/mod.rs:715-722), two polynomials,
    random-oracle challenges from the transcript: the device path vs the oracle, bit for bit."""
    import torch
    from poly_commit_amd import ipa
    comm_key, h, polys, comms, xi, point = _open_inputs(curve, n)
    dev = [torch.from_numpy(q.view(np.int64).copy()).cuda() for q in polys]
    (l, r, fk, c), _ = ipa.ipa_open(ctx, curve, comm_key, h, [d.data_ptr() for d in dev], [len(q) for q in polys], comms, point, xi)
    wl, wr, wfk, wc, _ = _oracle_open(curve, comm_key, h, polys, comms, xi, point)
    assert (l == wl).all() and (r == wr).all() and (fk == wfk).all() and (c == wc).all()
    # the same with the committer key resident (cloned on the device for the destructive folds, itself untouched)
    key_srs = ctx.upload_srs(curve, comm_key)
    dev = [torch.from_numpy(q.view(np.int64).copy()).cuda() for q in polys]
    (l2, r2, fk2, c2), _ = ipa.ipa_open(ctx, curve, key_srs, h, [d.data_ptr() for d in dev], [len(q) for q in polys], comms, point, xi)
    assert (l2 == wl).all() and (r2 == wr).all() and (fk2 == wfk).all() and (c2 == wc).all()
    assert (key_srs.read(0, min(n, 8)) == comm_key[:min(n, 8)]).all()
    key_srs.free()


@pytest.mark.parametrize("curve,n", [("pallas", 1 << 9), ("bn254", 1 << 4), ("bls12_381", 1 << 5)])
def test_ipa_device_proof_passes_reference_check(ctx, curve, n):
    """A proof made on the device satisfies the reference's verifier equations: InnerProductArgPC::check restated in
    Python big ints (oracle/pyref.py, independent of the prover's code paths) and the device-side check of
    poly_commit_amd/ipa.py (succinct check on the host, the verifier's final-key MSM -- ipa_pc/mod.rs:759-765 -- on the
    GPU) both accept it, and both reject it once c, the claimed value or final_comm_key is altered."""
    import torch
    from poly_commit_amd import ipa
    comm_key, h, polys, comms, xi, point = _open_inputs(curve, n)
    dev = [torch.from_numpy(q.view(np.int64).copy()).cuda() for q in polys]
    (l, r, fk, c), _ = ipa.ipa_open(ctx, curve, comm_key, h, [d.data_ptr() for d in dev], [len(q) for q in polys], comms, point, xi)
    values = [O.poly_eval(curve, q, point) for q in polys]
    assert ipa.ipa_check(ctx, curve, comm_key, h, comms, point, values, (l, r, fk, c), xi) is True
    one = np.array([1, 0, 0, 0], dtype=np.uint64)
    bad_c = c.copy(); bad_c[0] ^= np.uint64(1)
    assert ipa.ipa_check(ctx, curve, comm_key, h, comms, point, values, (l, r, fk, bad_c), xi) is False
    bad_v = [values[0].copy(), values[1]]; bad_v[0][1] ^= np.uint64(4)
    assert ipa.ipa_check(ctx, curve, comm_key, h, comms, point, bad_v, (l, r, fk, c), xi) is False
    assert ipa.ipa_check(ctx, curve, comm_key, h, comms, point, values, (l, r, comm_key[1], c), xi) is False
    with pytest.raises(ValueError):
        ipa.ipa_check(ctx, curve, comm_key, h, comms, point, values, (l[:-1], r[:-1], fk, c), xi)
    # the independent verifier (Python big ints)
    fr = R.CURVES[curve]["fr"]
    to_i = lambda a: O.fr_from_mont_array(curve, np.ascontiguousarray(a).reshape(-1, 4))      # noqa: E731
    pts = lambda a: O.array_to_points(curve, np.ascontiguousarray(a).reshape(-1, comm_key.shape[1]))   # noqa: E731
    proof_i = (pts(l), pts(r), pts(fk)[0], to_i(c)[0])
    args = (curve, pts(comm_key), pts(h)[0], [pts(cm)[0] for cm in comms], to_i(point)[0], [to_i(v)[0] for v in values])
    assert R.ipa_check(*args, proof_i, to_i(xi)) is True
    assert R.ipa_check(*args, (proof_i[0], proof_i[1], proof_i[2], (proof_i[3] + 1) % R.FIELDS[fr]["p"]), to_i(xi)) is False


@pytest.mark.parametrize("curve,n", [("pallas", 64), ("bn254", 16), ("bls12_381", 8)])
def test_ipa_hiding_and_degree_bounds_device(ctx, curve, n):
    """InnerProductArgPC with hiding and degree bounds through poly_commit_amd/ipa.py (commitments with the hiding term and
    the shifted key, the combination with shifted polynomials, the hiding polynomial of open, the verifier's combination):
    bit for bit against the restatement in oracle/pyref.py, then through both verifiers."""
    import torch
    from poly_commit_amd import ipa
    from test_oracle_cpu import _ipa_general_case
    fr = R.CURVES[curve]["fr"]
    p = R.FIELDS[fr]["p"]
    key, h, s, polys_i, ch_i, point_i, hp_i, hr_i = _ipa_general_case(curve, n)
    want = R.ipa_open_general(curve, key, h, s, polys_i, point_i, ch_i, hp_i, hr_i)
    m = lambda v: O.fr_mont_array(curve, v)                          # noqa: E731
    pa = lambda P: O.points_to_array(curve, [P])[0]                  # noqa: E731
    comm_key, h_xy, s_xy = O.points_to_array(curve, key), pa(h), pa(s)
    polys = []
    for q in polys_i:
        dev = torch.from_numpy(m(q["coeffs"]).view(np.int64)).cuda()
        rand = m([q["rand"]])[0] if q["hiding"] else None
        srand = m([q["shifted_rand"]])[0] if (q["hiding"] and q["degree_bound"] is not None) else None
        comm, sh = ipa.ipa_commit_general(ctx, curve, comm_key, s_xy, dev.data_ptr(), dev.shape[0], q["degree_bound"], rand, srand)
        assert O.array_to_points(curve, comm.reshape(1, -1))[0] == q["comm"]
        assert (sh is None) == (q["shifted_comm"] is None) and (sh is None or O.array_to_points(curve, sh.reshape(1, -1))[0] == q["shifted_comm"])
        polys.append(dict(dev=dev, comm=comm, shifted_comm=sh, degree_bound=q["degree_bound"], hiding=q["hiding"], rand=rand, shifted_rand=srand))
    hp = torch.from_numpy(m(hp_i).view(np.int64)).cuda()
    (l, r, fk, c, hc, rand), _ = ipa.ipa_open_general(ctx, curve, comm_key, h_xy, s_xy, polys, m([point_i])[0], m(ch_i), hp, m([hr_i])[0])
    pts = lambda a: O.array_to_points(curve, np.ascontiguousarray(a).reshape(-1, comm_key.shape[1]))   # noqa: E731
    assert pts(l) == want[0] and pts(r) == want[1] and pts(fk)[0] == want[2] and pts(hc)[0] == want[4]
    assert O.fr_from_mont_array(curve, np.stack([c, rand])) == [want[3], want[5]]
    vals = [R.poly_eval(fr, q["coeffs"], point_i) for q in polys_i]
    proof = (l, r, fk, c, hc, rand)
    assert ipa.ipa_check_general(ctx, curve, comm_key, h_xy, s_xy, polys, m([point_i])[0], m(vals), proof, m(ch_i)) is True
    bad_rand = m([(want[5] + 1) % p])[0]
    assert ipa.ipa_check_general(ctx, curve, comm_key, h_xy, s_xy, polys, m([point_i])[0], m(vals), (l, r, fk, c, hc, bad_rand), m(ch_i)) is False
    bad_vals = [vals[0], (vals[1] + 1) % p, vals[2]]
    assert ipa.ipa_check_general(ctx, curve, comm_key, h_xy, s_xy, polys, m([point_i])[0], m(bad_vals), proof, m(ch_i)) is False
    assert R.ipa_check_general(curve, key, h, s, polys_i, point_i, vals, (want[0], want[1], want[2], want[3], want[4], want[5]), ch_i) is True


@pytest.mark.parametrize("curve,n", [("pallas", 1 << 8), ("bn254", 1 << 5), ("bls12_381", 1 << 5)])
def test_ipa_open_whole_proof_cpp_host_mirror(curve, n, tmp_path):
    """The same through the C++ host mirror (host/ipa_pc.hpp: InnerProductArgPC::open, host/transcript.hpp):
    what the Rust shim would do, in the language that builds here."""
    comm_key, h, polys, comms, xi, point = _open_inputs(curve, n)
    libdir = os.path.join(ROOT, "poly_commit_amd")
    exe = os.path.join(ROOT, "tests", "cpp", "ipa_open_driver")
    src = exe + ".cpp"
    deps = [src, os.path.join(libdir, "libpc_hip.so")] + [os.path.join(libdir, "host", f) for f in os.listdir(os.path.join(libdir, "host"))]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src, "-L" + libdir, "-lpc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<III", O.CURVES[curve], n, len(polys)))
        f.write(comm_key.tobytes()); f.write(h.tobytes())
        for q in polys:
            f.write(struct.pack("<I", len(q))); f.write(q.tobytes())
        for cm in comms:
            f.write(cm.tobytes())
        f.write(point.tobytes()); f.write(xi.tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ipa two-level OK" in r.stdout and "ipa one-call OK" in r.stdout, r.stdout      # the explicit loop, its two-level form and pc_hip_ipa_open_rounds agree
    lg, nq = n.bit_length() - 1, 2 * O.fq_limbs(curve)
    got = np.fromfile(fout, dtype=np.uint64)
    assert got.size == (2 * lg + 1) * nq + 4
    wl, wr, wfk, wc, _ = _oracle_open(curve, comm_key, h, polys, comms, xi, point)
    assert (got[:lg * nq].reshape(lg, nq) == wl).all() and (got[lg * nq:2 * lg * nq].reshape(lg, nq) == wr).all()
    assert (got[2 * lg * nq:(2 * lg + 1) * nq] == wfk).all() and (got[(2 * lg + 1) * nq:] == wc).all()


@pytest.mark.parametrize("curve,n", [("pallas", 32), ("bn254", 16), ("bls12_381", 8)])
def test_ipa_hiding_and_degree_bounds_cpp_host_mirror(curve, n, tmp_path):
    """The same through the C++ host mirror (host/ipa_pc.hpp: commit_general / open_general / check_general); the driver
    also runs the mirror's verifier on honest and altered inputs."""
    from test_oracle_cpu import _ipa_general_case
    key, h, s, polys_i, ch_i, point_i, hp_i, hr_i = _ipa_general_case(curve, n)
    want = R.ipa_open_general(curve, key, h, s, polys_i, point_i, ch_i, hp_i, hr_i)
    libdir = os.path.join(ROOT, "poly_commit_amd")
    exe = os.path.join(ROOT, "tests", "cpp", "ipa_general_driver")
    src = exe + ".cpp"
    deps = [src, os.path.join(libdir, "libpc_hip.so")] + [os.path.join(libdir, "host", f) for f in os.listdir(os.path.join(libdir, "host"))]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src, "-L" + libdir, "-lpc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    m = lambda v: O.fr_mont_array(curve, v)                          # noqa: E731
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<III", O.CURVES[curve], n, len(polys_i)))
        f.write(O.points_to_array(curve, key + [h, s]).tobytes())
        for q in polys_i:
            f.write(struct.pack("<I", len(q["coeffs"]))); f.write(m(q["coeffs"]).tobytes())
            f.write(struct.pack("<iI", -1 if q["degree_bound"] is None else q["degree_bound"], 1 if q["hiding"] else 0))
            f.write(m([q["rand"], q["shifted_rand"]]).tobytes())
        f.write(m([point_i]).tobytes()); f.write(m(ch_i).tobytes()); f.write(m(hp_i).tobytes()); f.write(m([hr_i]).tobytes())
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    aw = 2 * O.fq_limbs(curve)
    raw = np.fromfile(fout, dtype=np.uint64)
    k, rounds = len(polys_i), (n - 1).bit_length()
    off = 0
    coms = O.array_to_points(curve, raw[off:off + 2 * k * aw].reshape(2 * k, aw)); off += 2 * k * aw
    for j, q in enumerate(polys_i):
        assert coms[2 * j] == q["comm"] and coms[2 * j + 1] == q["shifted_comm"]
    lr = O.array_to_points(curve, raw[off:off + (2 * rounds + 1) * aw].reshape(2 * rounds + 1, aw)); off += (2 * rounds + 1) * aw
    assert lr[:rounds] == want[0] and lr[rounds:2 * rounds] == want[1] and lr[2 * rounds] == want[2]
    assert O.fr_from_mont_array(curve, raw[off:off + 4].reshape(1, 4))[0] == want[3]; off += 4
    assert O.array_to_points(curve, raw[off:off + aw].reshape(1, aw))[0] == want[4]; off += aw
    assert O.fr_from_mont_array(curve, raw[off:off + 4].reshape(1, 4))[0] == want[5]
