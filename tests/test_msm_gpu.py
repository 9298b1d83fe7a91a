"""GPU parity: pc_hip_msm (HIP Pippenger) == CPU oracle, bit for bit, through the C ABI.

Mirrors the reference's cross-implementation pattern (streaming_kzg/tests.rs:40-84: two MSM
implementations must return equal commitments) and the MSM-linearity check of
kzg10/mod.rs:520-544 (add_commitments_test)."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu

CURVES = ["bls12_381", "bn254", "pallas"]


def _check(srs, curve, bases, scalars, **kw):
    got, inf = srs.msm(scalars, **kw)
    off = kw.get("base_offset", 0)
    want = O.msm_pippenger(curve, bases[off:], scalars, 8, 1)
    assert (got == want).all(), (curve, len(scalars), kw)
    assert inf == (not want.any())


@pytest.mark.parametrize("curve", CURVES)
def test_msm_small_sizes(ctx, curve):
    nmax = 4200
    bases = O.gen_bases(curve, nmax)
    srs = ctx.upload_srs(curve, bases)
    for n in (0, 1, 2, 31, 32, 33, 257, 4097):
        scalars = O.gen_scalars(curve, 0x5EED0001 + n, max(n, 1))[:n]
        _check(srs, curve, bases, scalars)
    srs.free()


@pytest.mark.parametrize("curve", CURVES)
def test_msm_edge_scalars(ctx, curve):
    n = 1500
    r = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
    bases = O.gen_bases(curve, n)
    bases[7] = 0            # a point at infinity among the bases
    bases[100] = bases[99]  # repeated base
    srs = ctx.upload_srs(curve, bases)
    rnd = O.gen_scalars(curve, 99, n)
    cases = {
        "zeros": np.zeros((n, 4), dtype=np.uint64),
        "ones": O.ints_to_limbs([1] * n, 4),
        "r-1": O.ints_to_limbs([r - 1] * n, 4),
        "same": np.repeat(rnd[:1], n, axis=0),
        "half-zero": np.where((np.arange(n) % 2 == 0)[:, None], rnd, 0).astype(np.uint64),
        "low-hamming": O.ints_to_limbs([1 << (i % 254) for i in range(n)], 4),
        "window-carry": O.ints_to_limbs([(1 << 254) - 1 - (i << 60) for i in range(n)], 4) if curve != "bn254" else
                        O.ints_to_limbs([(1 << 253) - 1 - (i << 60) for i in range(n)], 4),
    }
    for name, sc in cases.items():
        sc = np.ascontiguousarray(sc)
        got, inf = srs.msm(sc)
        want = O.msm_pippenger(curve, bases, sc, 8, 1)
        assert (got == want).all(), (curve, name)
    # P + (-P): scalars (1, r-1) on the repeated base pair cancel
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[99] = O.ints_to_limbs([1], 4)[0]
    sc[100] = O.ints_to_limbs([r - 1], 4)[0]
    got, inf = srs.msm(sc)
    assert inf and not got.any()
    srs.free()


@pytest.mark.parametrize("curve", CURVES)
def test_msm_offset_truncation_montgomery(ctx, curve):
    n = 3000
    bases = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, bases)
    scalars = O.gen_scalars(curve, 5, n)
    # KZG10::commit passes powers_of_g[num_leading_zeros..] (kzg10/mod.rs:175-178)
    _check(srs, curve, bases, scalars[:1000], base_offset=1234)
    # msm_bigint uses min(len) pairs: more scalars than bases left
    got, _ = srs.msm(scalars, base_offset=2000)
    want = O.msm_pippenger(curve, bases[2000:], scalars[:1000], 8, 1)
    assert (got == want).all()
    # Montgomery-form scalars (the polynomial's coefficient slice as it lies in memory)
    mont = O.f_to_mont(curve, 1, scalars)
    got, _ = srs.msm(mont, montgomery=True)
    want = O.msm_pippenger(curve, bases, scalars, 8, 1)
    assert (got == want).all()
    srs.free()


@pytest.mark.parametrize("c,T", [(4, 1), (7, 3), (10, 16), (13, 64), (16, 0)])
def test_msm_tunings(ctx, c, T):
    curve = "bls12_381"
    n = 5000
    bases = O.gen_bases(curve, n)
    scalars = O.gen_scalars(curve, 1234, n)
    ctx.set_msm_tuning(c, T)
    try:
        srs = ctx.upload_srs(curve, bases)
        _check(srs, curve, bases, scalars)
        srs.free()
    finally:
        ctx.set_msm_tuning(0, 0)


@pytest.mark.parametrize("T", [1, 2, 5, 8])
@pytest.mark.parametrize("n,table", [(700, False), (5000, True), (40000, False), (70001, True)])
def test_msm_chains_of_cut_buckets(ctx, T, n, table):
    """The joins of buckets that chunk edges cut (k_accumulate's in-workgroup scan, k_accumulate_edges, the segmented reduction
    behind them): small chunks against distributions whose buckets span a few lanes, a whole workgroup, several workgroups --
    uniform, a handful of values, all equal, mostly zero -- with and without the window table; bit-identical to the oracle."""
    curve = "bls12_381"
    bases = O.gen_bases(curve, n)
    rnd = O.gen_scalars(curve, 0xC4A1 + n, n)
    idx = np.arange(n)
    dists = {
        "uniform": rnd,
        "five_values": np.ascontiguousarray(rnd[idx % 5]),
        "all_equal": np.ascontiguousarray(np.repeat(rnd[:1], n, axis=0)),
        "runs": np.ascontiguousarray(rnd[idx // 300]),
        "sparse": np.ascontiguousarray(np.where((idx % 13 == 0)[:, None], rnd, 0).astype(np.uint64)),
    }
    ctx.set_msm_tuning(0, T)
    try:
        srs = ctx.upload_srs(curve, bases)
        if table:
            srs.precompute()
        for name, sc in dists.items():
            got, _ = srs.msm(sc)
            assert (got == O.msm_pippenger(curve, bases, sc, 8, 1)).all(), (name, T, n, table)
        srs.free()
    finally:
        ctx.set_msm_tuning(0, 0)


def test_msm_linearity(ctx):
    """commit(f*p) == f*commit(p): add_commitments_test, kzg10/mod.rs:520-544."""
    curve = "bls12_381"
    n = 2048
    fr = R.FIELDS["bls12_381_fr"]["p"]
    bases = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, bases)
    s = O.limbs_to_ints(O.gen_scalars(curve, 77, n))
    f = 0x1234567
    c1, _ = srs.msm(O.ints_to_limbs(s, 4))
    c2, _ = srs.msm(O.ints_to_limbs([f * x % fr for x in s], 4))
    p1 = O.array_to_points(curve, c1)[0]
    assert R.ec_mul(curve, f, p1) == O.array_to_points(curve, c2)[0]
    srs.free()


def test_msm_2_16_bls(ctx):
    curve = "bls12_381"
    n = (1 << 16) + 1
    bases = O.gen_bases(curve, n)
    scalars = O.gen_scalars(curve, 0x5EED0001, n)
    srs = ctx.upload_srs(curve, bases)
    _check(srs, curve, bases, scalars)
    srs.free()


def test_msm_async_pipeline_and_batch(ctx):
    """pc_hip_msm_async / pc_hip_job_wait / pc_hip_msm_batch: several MSMs in flight on the
    SRS's independent pipelines return the same points as the blocking call."""
    import ctypes as C
    curve = "bn254"
    n = 6000
    bases = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, bases)
    scal = [O.gen_scalars(curve, 1000 + k, n - 37 * k) for k in range(7)]
    want = [O.msm_pippenger(curve, bases, s, 8, 1) for s in scal]
    jobs = [srs.msm_async(s) for s in scal]            # more jobs than lanes: lanes are recycled
    for j, w in zip(reversed(jobs), reversed(want)):     # wait out of order
        got, _ = j.wait()
        assert (got == w).all()
    # batch entry point (MarlinKZG10::commit's loop over polynomials, marlin_pc/mod.rs:192-237)
    lib = ctx.lib
    k = len(scal)
    ptrs = (C.c_void_p * k)(*[s.ctypes.data for s in scal])
    lens = (C.c_size_t * k)(*[len(s) for s in scal])
    out = np.zeros((k, 8), dtype=np.uint64)
    infs = (C.c_int * k)()
    ctx.check(lib.pc_hip_msm_batch(ctx.h, srs.h, None, ptrs, lens, k, 0, 0, C.c_void_p(out.ctypes.data), infs))
    for i in range(k):
        assert (out[i] == want[i]).all() and infs[i] == 0
    srs.free()


def test_msm_full_size_2_20_parity_and_linearity(ctx):
    """BASELINE configs[1] size (2^20 + 1 pairs): parity with the oracle and, independently of
    it, linearity msm(s) + msm(t) == msm(s + t)."""
    import poly_commit_amd as pc
    curve = "bls12_381"
    n = (1 << 20) + 1
    fr = R.FIELDS["bls12_381_fr"]["p"]
    bases = O.gen_bases(curve, n)
    s = O.gen_scalars(curve, 0x5EED0001, n)
    t = O.gen_scalars(curve, 0x5EED0002, n)
    srs = ctx.upload_srs(curve, bases)
    a, _ = srs.msm(s)
    b, _ = srs.msm(t)
    # s + t mod r with numpy object ints would be slow: use the Montgomery path, (s + t) via host big ints on a sample is not
    # enough, so add limb-wise in Python ints
    st = O.ints_to_limbs([(x + y) % fr for x, y in zip(O.limbs_to_ints(s), O.limbs_to_ints(t))], 4)
    c, _ = srs.msm(st)
    assert (pc.points_sum(curve, np.stack([a, b])) == c).all()
    assert (a == O.msm_pippenger(curve, bases, s, 16, 1)).all()
    srs.free()


def test_msm_large_adversarial_and_known_answer(ctx):
    """2^18 pairs with scalar distributions that defeat a naive bucket method (every scalar
    equal -> one bucket per window holds everything; two values; mostly zero), and the closed
    form  sum_i 1 * (i+1)G = n(n+1)/2 * G  that needs no oracle MSM at all."""
    import poly_commit_amd as pc
    curve = "bls12_381"
    n = 1 << 18
    fr = R.FIELDS["bls12_381_fr"]["p"]
    bases = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, bases)
    g = bases[0]
    ones = O.ints_to_limbs([1], 4).repeat(n, axis=0)
    got, _ = srs.msm(np.ascontiguousarray(ones))
    want = pc.point_mul(curve, g, O.fr_mont_array(curve, [n * (n + 1) // 2 % fr])[0])
    assert (got == want).all()
    rnd = O.gen_scalars(curve, 5, n)
    same = np.ascontiguousarray(np.repeat(rnd[:1], n, axis=0))
    got, _ = srs.msm(same)
    want = pc.point_mul(curve, want, O.f_to_mont(curve, 1, rnd[:1])[0])          # k * sum P_i
    assert (got == want).all()
    two = np.ascontiguousarray(np.where((np.arange(n) % 2 == 0)[:, None], rnd[:1], rnd[1:2]))
    assert (srs.msm(two)[0] == O.msm_pippenger(curve, bases, two, 16, 1)).all()
    sparse = np.ascontiguousarray(np.where((np.arange(n) % 97 == 0)[:, None], rnd, 0).astype(np.uint64))
    assert (srs.msm(sparse)[0] == O.msm_pippenger(curve, bases, sparse, 16, 1)).all()
    srs.free()


def test_msm_2_22_known_answer_and_parity(ctx):
    """Beyond the bench size: 2^22 pairs (auto window c = 18)."""
    import poly_commit_amd as pc
    curve = "bn254"
    n = 1 << 22
    fr = R.FIELDS["bn254_fr"]["p"]
    bases = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, bases)
    ones = np.ascontiguousarray(O.ints_to_limbs([1], 4).repeat(n, axis=0))
    got, _ = srs.msm(ones)
    assert (got == pc.point_mul(curve, bases[0], O.fr_mont_array(curve, [n * (n + 1) // 2 % fr])[0])).all()
    s = O.gen_scalars(curve, 77, n)
    assert (srs.msm(s)[0] == O.msm_pippenger(curve, bases, s, 64, 1)).all()
    srs.free()


def test_rust_affine_layout_and_threads(ctx):
    """(i) SRS handed over in arkworks' in-memory Affine layout {x, y, infinity: bool} (104-byte stride for
    BLS12-381), infinity flag honoured; (ii) one ctx shared by several host threads (the reference calls
    the MSM from rayon workers, hyrax/mod.rs:233-242): calls are serialised, results unaffected."""
    import ctypes as C
    import threading
    curve = "bls12_381"
    n = 3000
    packed = O.gen_bases(curve, n)
    rust = np.zeros((n, 13), dtype=np.uint64)          # 104 bytes: x(48) y(48) flag(1) + padding
    rust[:, :12] = packed
    rust[5, :12] = 0xDEADBEEF                          # garbage coordinates, but flagged as the identity
    rust[5, 12] = 1
    want_bases = packed.copy()
    want_bases[5] = 0
    srs = ctx.upload_srs(curve, rust, n=n, stride_bytes=104)
    scal = [O.gen_scalars(curve, 900 + k, n) for k in range(6)]
    want = [O.msm_pippenger(curve, want_bases, s, 8, 1) for s in scal]
    got = [None] * len(scal)

    def work(k):
        got[k] = srs.msm(scal[k])[0]
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(scal))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(len(scal)):
        assert (got[k] == want[k]).all()
    srs.free()


@pytest.mark.parametrize("curve", CURVES)
def test_msm_window_table_small(ctx, curve):
    """pc_hip_srs_precompute: MSMs against the window table (one shared bucket set) return the
    same points as the table-free path and the oracle -- several window widths, base offsets,
    truncation, Montgomery scalars, infinity / repeated bases, short calls below min_pairs."""
    n = 3000
    bases = O.gen_bases(curve, n)
    bases[7] = 0
    bases[21] = bases[20]
    s = O.gen_scalars(curve, 0x7AB1E, n)
    want = O.msm_pippenger(curve, bases, s, 8, 1)
    srs = ctx.upload_srs(curve, bases)
    assert (srs.msm(s)[0] == want).all()
    r = R.FIELDS[curve + "_fr"]["p"]
    top = O.ints_to_limbs([r - 1] * n, 4)
    for c in (0, 5, 9, 12, 16):
        srs.precompute(window_bits=c, min_pairs=1)
        assert (srs.msm(s)[0] == want).all(), c
        assert (srs.msm(O.f_to_mont(curve, 1, s), montgomery=True)[0] == want).all(), c
        assert (srs.msm(np.ascontiguousarray(s[:1000]), base_offset=500)[0] == O.msm_pippenger(curve, bases[500:], s[:1000], 8, 1)).all(), c
        assert (srs.msm(top)[0] == O.msm_pippenger(curve, bases, top, 8, 1)).all(), c
        got, inf = srs.msm(np.zeros((n, 4), dtype=np.uint64))
        assert inf and not got.any()
    # default threshold: calls shorter than a quarter of the SRS take the table-free path
    srs.precompute()
    assert (srs.msm(np.ascontiguousarray(s[:100]))[0] == O.msm_pippenger(curve, bases, s[:100], 8, 1)).all()
    assert (srs.msm(s)[0] == want).all()
    jobs = [srs.msm_async(s) for _ in range(4)]
    assert all((j.wait()[0] == want).all() for j in jobs)
    srs.free()


def test_msm_window_table_2_20_and_fold_invalidation(ctx):
    """Table mode at the bench size (auto c = 20, 13 digits) against the table-free result and the
    closed form; pc_hip_ec_fold drops the table (the key changed) and later MSMs use the new key."""
    import poly_commit_amd as pc
    curve = "bls12_381"
    n = (1 << 20) + 1
    fr = R.FIELDS["bls12_381_fr"]["p"]
    bases = O.gen_bases(curve, n)
    s = O.gen_scalars(curve, 0x5EED0001, n)
    srs = ctx.upload_srs(curve, bases)
    plain, _ = srs.msm(s)
    srs.precompute()
    assert (srs.msm(s)[0] == plain).all()
    ones = np.ascontiguousarray(O.ints_to_limbs([1], 4).repeat(n, axis=0))
    assert (srs.msm(ones)[0] == pc.point_mul(curve, bases[0], O.fr_mont_array(curve, [n * (n + 1) // 2 % fr])[0])).all()
    q = np.ascontiguousarray(s[: n - 1])
    assert (srs.msm(q)[0] == O.msm_pippenger(curve, bases, q, 16, 1)).all()          # open-side length, same table
    srs.free()
    # fold: key' = lo + u * hi (ipa_pc/mod.rs:699-707) on a small key with a table
    curve = "pallas"
    m = 2048
    kb = O.gen_bases(curve, m)
    srs = ctx.upload_srs(curve, kb)
    srs.precompute(min_pairs=1)
    t = O.gen_scalars(curve, 9, m)
    assert (srs.msm(t)[0] == O.msm_pippenger(curve, kb, t, 8, 1)).all()
    u = O.f_to_mont(curve, 1, O.gen_scalars(curve, 10, 1))[0]
    srs.ec_fold(m // 2, u)
    folded = srs.read(0, m // 2)
    t2 = np.ascontiguousarray(t[: m // 2])
    assert (srs.msm(t2)[0] == O.msm_pippenger(curve, folded, t2, 8, 1)).all()
    srs.free()


@pytest.mark.parametrize("curve", CURVES)
def test_msm_many_small(ctx, curve):
    """pc_hip_msm_many: B MSMs over the same m bases == B separate oracle MSMs; zero rows give infinity;
    base offset; Montgomery scalars; a second call reuses the cached table; a different shape rebuilds."""
    m, B = 200, 37
    bases = O.gen_bases(curve, m + 50)
    bases[60] = bases[59]
    srs = ctx.upload_srs(curve, bases)
    s = O.gen_scalars(curve, 0x4A11, B * m).reshape(B, m, 4)
    s[5] = 0
    s[6, :, :] = s[6, :1, :]
    r = R.FIELDS[curve + "_fr"]["p"]
    s[7] = O.ints_to_limbs([r - 1] * m, 4)
    s = np.ascontiguousarray(s)
    for off in (0, 50):
        want = np.stack([O.msm_pippenger(curve, bases[off:off + m], np.ascontiguousarray(s[k]), 4, 1) for k in range(B)])
        got, inf = srs.msm_many(s, base_offset=off)
        assert (got == want).all(), off
        assert inf[5] and inf.sum() == 1
        got2, _ = srs.msm_many(O.f_to_mont(curve, 1, s.reshape(-1, 4)).reshape(B, m, 4), base_offset=off, montgomery=True)
        assert (got2 == want).all()
    # the ordinary MSM entry points are unaffected, and a different shape is served as well
    assert (srs.msm(np.ascontiguousarray(s[0]))[0] == O.msm_pippenger(curve, bases, np.ascontiguousarray(s[0]), 4, 1)).all()
    got3, _ = srs.msm_many(np.ascontiguousarray(s[:3, :64]))
    assert (got3 == np.stack([O.msm_pippenger(curve, bases[:64], np.ascontiguousarray(s[k, :64]), 4, 1) for k in range(3)])).all()
    got4, _ = srs.msm_many(np.ascontiguousarray(s[:1, :1]))
    assert (got4[0] == O.msm_pippenger(curve, bases[:1], np.ascontiguousarray(s[0, :1]), 1, 1)).all()
    srs.free()


def test_msm_many_hyrax_shape_1024x1024(ctx):
    """Hyrax's shape for 2^20 evaluations: 1024 row commitments of 1024 pairs each (hyrax/mod.rs:233-242),
    one call.  Spot rows against the oracle; the sum of all rows against ONE big MSM with column-summed
    scalars (linearity) -- a check that involves every row without 1024 oracle MSMs."""
    import poly_commit_amd as pc
    curve = "bn254"
    m = B = 1024
    fr = R.FIELDS["bn254_fr"]["p"]
    bases = O.gen_bases(curve, m)
    srs = ctx.upload_srs(curve, bases)
    s = O.gen_scalars(curve, 0x4A12, B * m).reshape(B, m, 4)
    got, inf = srs.msm_many(s)
    assert not inf.any()
    for k in (0, 511, 1023):
        assert (got[k] == O.msm_pippenger(curve, bases, np.ascontiguousarray(s[k]), 8, 1)).all()
    cols = [0] * m
    ints = O.limbs_to_ints(s.reshape(-1, 4))
    for k in range(B):
        row = ints[k * m:(k + 1) * m]
        cols = [(a + b) % fr for a, b in zip(cols, row)]
    want_sum = O.msm_pippenger(curve, bases, O.ints_to_limbs(cols, 4), 8, 1)
    assert (pc.points_sum(curve, got) == want_sum).all()
    srs.free()


def test_msm_many_and_precompute_argument_errors(ctx):
    """Error convention of the newer entry points: negative status, nothing computed."""
    import ctypes as C
    curve = "bn254"
    bases = O.gen_bases(curve, 64)
    srs = ctx.upload_srs(curve, bases)
    s = O.gen_scalars(curve, 3, 64)
    out = np.zeros((2, 8), dtype=np.uint64)
    lib = ctx.lib
    vp = C.c_void_p
    # bases[base_offset + m) must lie inside the SRS
    assert lib.pc_hip_msm_many(ctx.h, srs.h, 60, vp(s.ctypes.data), 0, 0, 8, 2, vp(out.ctypes.data), None) == -1
    assert lib.pc_hip_msm_many(ctx.h, srs.h, 0, None, 0, 0, 8, 2, vp(out.ctypes.data), None) == -1
    assert lib.pc_hip_msm_many(ctx.h, srs.h, 0, vp(s.ctypes.data), 0, 0, 8, 2, None, None) == -1
    # too many entries for the 31-bit entry index: PC_ERR_TOO_LARGE, before any allocation
    assert lib.pc_hip_msm_many(ctx.h, srs.h, 0, vp(s.ctypes.data), 0, 0, 64, 1 << 26, vp(out.ctypes.data), None) == -5
    # zero MSMs / zero pairs are fine
    assert lib.pc_hip_msm_many(ctx.h, srs.h, 0, vp(s.ctypes.data), 0, 0, 8, 0, vp(out.ctypes.data), None) == 0
    inf = (C.c_int * 2)()
    assert lib.pc_hip_msm_many(ctx.h, srs.h, 0, vp(s.ctypes.data), 0, 0, 0, 2, vp(out.ctypes.data), inf) == 0
    assert list(inf) == [1, 1] and not out.any()
    assert lib.pc_hip_srs_precompute(ctx.h, srs.h, 1, 0) == -1
    assert lib.pc_hip_srs_precompute(ctx.h, srs.h, 24, 0) == -1
    assert lib.pc_hip_srs_precompute(None, srs.h, 0, 0) == -1
    srs.free()


@pytest.mark.parametrize("curve,k", [("bn254", 11), ("bls12_381", 3)])
def test_msm_batch_fast_path_over_window_table(ctx, curve, k):
    """pc_hip_msm_batch with equal-length device-resident polynomials and a window table: groups of 8 polynomials per
    many-MSM pass (here 8 + 3, resp. one group of 3), a base offset, against the oracle -- and the same batch on the
    per-polynomial path (no table) gives the same points."""
    import torch
    n_srs, off, m = 40000, 5, 30000
    b = O.gen_bases(curve, n_srs)
    host = [O.gen_scalars(curve, 0x8B0 + j, m) for j in range(k)]
    dev = [torch.from_numpy(O.f_to_mont(curve, 1, h).view(np.int64)).cuda() for h in host]
    srs = ctx.upload_srs(curve, b)
    slow = srs.msm_batch([t.data_ptr() for t in dev], [m] * k, base_offsets=[off] * k)
    srs.precompute(min_pairs=1)
    fast = srs.msm_batch([t.data_ptr() for t in dev], [m] * k, base_offsets=[off] * k)
    assert (fast == slow).all()
    for j in (0, k // 2, k - 1):
        assert (fast[j] == O.msm_pippenger(curve, np.ascontiguousarray(b[off:off + m]), host[j], 8, 1)).all(), j
    # HOST polynomials (what MarlinKZG10::commit hands over): the same passes, every pass's polynomials staged on its pipeline
    mont = [np.ascontiguousarray(O.f_to_mont(curve, 1, h)) for h in host]
    from_host = srs.msm_batch(mont, [m] * k, base_offsets=[off] * k, host=True)
    assert (from_host == fast).all()
    assert (srs.msm_batch(host, [m] * k, base_offsets=[off] * k, montgomery=False, host=True) == fast).all()      # canonical scalars
    # unequal lengths fall back to the per-polynomial pipelines
    lens = [m - j for j in range(k)]
    mixed = srs.msm_batch([t.data_ptr() for t in dev], lens, base_offsets=[off] * k)
    assert (mixed[k - 1] == O.msm_pippenger(curve, np.ascontiguousarray(b[off:off + lens[-1]]), host[-1][:lens[-1]], 8, 1)).all()
    srs.free()


def test_msm_repeated_call_through_captured_graph(monkeypatch):
    """PC_HIP_GRAPHS=1: the second identical call (same resident bases, same device scalar buffer) is captured into a
    hipGraph, later ones replay it; a different buffer in between takes another slot.  Every result equals the oracle's."""
    import torch
    import poly_commit_amd as pc
    monkeypatch.setenv("PC_HIP_GRAPHS", "1")
    ctx = pc.Context(0)
    curve, n = "bls12_381", 5000
    bases = O.gen_bases(curve, n)
    sc = [O.gen_scalars(curve, 0x6A0 + k, n) for k in range(2)]
    want = [O.msm_pippenger(curve, bases, s, 8, 1) for s in sc]
    dev = [torch.from_numpy(s.view(np.int64).copy()).cuda() for s in sc]
    srs = ctx.upload_srs(curve, bases)
    for rnd in range(4):                      # 3 pipelines: every (pipeline, buffer) pair is seen, captured and replayed
        for k in (0, 1, 0):
            got, _ = srs.msm(dev[k].data_ptr(), n=n)
            assert (got == want[k]).all(), (rnd, k)
    srs.precompute(min_pairs=1)               # new pipelines (window table): capture again
    for rnd in range(4):
        for k in (1, 0):
            got, _ = srs.msm(dev[k].data_ptr(), n=n)
            assert (got == want[k]).all(), (rnd, k)
    srs.free()
    ctx.close()


def test_msm_graph_replay_survives_a_growing_scratch_buffer():
    """Calls of at most 2^18 pairs on device scalars replay a captured hipGraph from their third occurrence on (the library's default).
    A captured graph holds the addresses of the pipeline's grow-only scratch as they were; a LARGER call on the same pipeline then
    reallocates that scratch.  Round 6: the replay after it read freed memory (a GPU memory access fault in the second IPA opening of
    a process).  Now every device free on the pipeline invalidates its graphs: small, small, small on each of the three pipelines,
    then a large call on each, then the small ones again -- all results the oracle's."""
    import torch
    import poly_commit_amd as pc
    ctx = pc.Context(0)
    curve, n_small, n_big = "pallas", 3000, 1 << 16
    bases = O.gen_bases(curve, n_big)
    sc_small, sc_big = O.gen_scalars(curve, 0x6B0, n_small), O.gen_scalars(curve, 0x6B1, n_big)
    want_small = O.msm_pippenger(curve, np.ascontiguousarray(bases[:n_small]), sc_small, 8, 1)
    want_big = O.msm_pippenger(curve, bases, sc_big, 8, 1)
    d_small = torch.from_numpy(sc_small.view(np.int64).copy()).cuda()
    d_big = torch.from_numpy(sc_big.view(np.int64).copy()).cuda()
    srs = ctx.upload_srs(curve, bases)
    for rep in range(3):
        for _ in range(9):                    # three pipelines in rotation: plain, captured, replayed on each
            assert (srs.msm(d_small.data_ptr(), n=n_small)[0] == want_small).all()
        for _ in range(3):                    # every pipeline's sort scratch grows
            assert (srs.msm(d_big.data_ptr(), n=n_big)[0] == want_big).all()
        for _ in range(7):                    # (7: the rotation shifts, so that across the repetitions every pipeline meets both sizes in both orders)
            assert (srs.msm(d_small.data_ptr(), n=n_small)[0] == want_small).all()
        ctx.trim()                            # idle pipelines give their scratch back: the graphs must go with it
    srs.free()
    ctx.close()


def test_msm_graph_path_randomised_differential():
    """tools/graph_fuzz.py for a few seconds: resident buffers and call shapes repeated in random order on one key (plain, captured,
    replayed on every pipeline), with growing scratch, pc_hip_ctx_trim, table builds and in-place key folds in between (21 442 cases with
    11 176 repeats in 150 s without a mismatch on the round-6 library)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_fuzz.py"), "12", "20260930"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_msm_randomised_differential():
    """tools/msm_fuzz.py for a few seconds: random sizes / chunk lengths / window widths / tables / scalar distributions on all three
    curves against the oracle (6000 cases in 150 s without a mismatch on the round-3 library; this keeps ~400 of them in the suite)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "msm_fuzz.py"), "10", "20260926"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_msm_equal_partial_sums_meet_in_the_joins(ctx, curve):
    """ONE base under ONE scalar: every chunk of a bucket's run sums to the same point, so the in-workgroup scan of k_accumulate, the
    edge kernel, the segmented and the (two-lane) bucket reduction all add EQUAL partial sums -- XyzzD::add / the two-lane half_add take
    their doubling branch on coordinates that left the accumulation lazily reduced (BLS12-381) or canonical (BN254).  Round-3 advisor
    finding; tests/test_emu_cpu.py steps the same shape on the CPU."""
    g = O.gen_bases(curve, 3)
    for n, chunk in ((64, 0), (4096, 0), (4096, 2), (5000, 5), (1 << 15, 0)):
        b = np.ascontiguousarray(np.repeat(g[2:3], n, axis=0))
        for k in (O.gen_scalars(curve, 5, 1), O.ints_to_limbs([1], 4), O.ints_to_limbs([3], 4)):
            sc = np.ascontiguousarray(np.repeat(k, n, axis=0))
            want = O.msm_pippenger(curve, b, sc, 8, 2)
            if chunk:
                ctx.set_msm_tuning(0, chunk)
            srs = ctx.upload_srs(curve, b)
            ctx.set_msm_tuning(0, 0)
            for table in (False, True):
                if table:
                    srs.precompute(min_pairs=1)
                got, _ = srs.msm(sc)
                assert (got == want).all(), (curve, n, chunk, table)
            srs.free()
