"""Multi-GPU entry points of the C ABI (pc_hip_group_*): one committer key sharded over N contexts driven from one
process.  On a one-GPU box the group lists device 0 several times -- the chunking, the halo base, the division
carries and the fold of the partial points are exactly what N devices would run.  Results must be bit-identical to
the oracle (and hence to the single-device path)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve,n,ndev,table", [("bls12_381", 5000, 3, False), ("bn254", 4097, 2, True), ("pallas", 1 << 12, 4, False),
                                                ("bn254", 5, 3, False), ("bls12_381", (1 << 13) + 1, 8, True), ("bn254", 8 * 1000 + 3, 8, False)])
def test_group_commit_open_matches_oracle(curve, n, ndev, table):
    import poly_commit_amd as pc
    g = pc.Group([0] * ndev)
    powers = O.gen_bases(curve, n)
    srs = g.upload_srs(curve, powers, precompute=table)
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x6A0 + n, n))
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x6B0, 1))[0]
    comm, _ = srs.msm(coeffs, montgomery=True)
    rc, want_c = O.kzg_commit(curve, powers, coeffs)
    assert rc == 0 and (comm == want_c).all()
    proof, value = srs.kzg_open(coeffs, z)
    rc, want_w = O.kzg_open(curve, powers, coeffs, z)
    assert rc == 0 and (proof == want_w).all()
    assert (value == O.poly_eval(curve, coeffs, z)).all()
    # a slice that starts inside chunk 1 and ends inside the last chunk; a polynomial shorter than the key
    off, m = n // ndev + 1, n - n // ndev - 2
    if m > 0:
        s = O.gen_scalars(curve, 0x6C0, m)
        got, _ = srs.msm(s, base_offset=off)
        assert (got == O.msm_pippenger(curve, np.ascontiguousarray(powers[off:off + m]), s, 8, 1)).all()
    short = np.ascontiguousarray(coeffs[: max(2, n // 2 - 1)])
    proof, value = srs.kzg_open(short, z)
    rc, want_w = O.kzg_open(curve, powers, short, z)
    assert rc == 0 and (proof == want_w).all() and (value == O.poly_eval(curve, short, z)).all()
    srs.free()
    g.close()


def test_group_batch_and_ntt_rows():
    """configs[2] shape (k polynomials against one sharded SRS) and configs[4] shape (rows split over devices)."""
    import poly_commit_amd as pc
    curve, n, k = "bn254", 3000, 5
    g = pc.Group([0, 0, 0])
    powers = O.gen_bases(curve, n)
    srs = g.upload_srs(curve, powers)
    polys = [O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x700 + j, n - 7 * j)) for j in range(k)]
    got = srs.msm_batch(polys)
    for j in range(k):
        rc, want = O.kzg_commit(curve, powers, polys[j])
        assert rc == 0 and (got[j] == want).all(), j
    srs.free()
    mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x710, 7 * 64)).reshape(7, 64, 4)
    assert (g.ntt_batch(curve, mat, 8) == O.ntt_batch(curve, mat, 8)).all()
    g.close()


@pytest.mark.parametrize("curve,n,ndev,table", [("bls12_381", 5000, 1, True), ("bls12_381", 6001, 2, True), ("bn254", 4097, 3, False), ("pallas", 1 << 12, 4, False)])
def test_group_commit_open_async_jobs(curve, n, ndev, table):
    """pc_hip_group_commit_open_async: commit + open of one polynomial as one asynchronous job (persistent worker thread per
    device, persistent shard / quotient buffers, two jobs in flight) -- bit-identical to the oracle's commit and open with 1, 2,
    3, 4 contexts on device 0, from host coefficients and from device-resident shards, and to the blocking group calls."""
    import torch
    import poly_commit_amd as pc
    g = pc.Group([0] * ndev)
    powers = O.gen_bases(curve, n)
    srs = g.upload_srs(curve, powers, precompute=table)
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x6B1, 1))[0]
    polys = [O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x6D0 + k, n - 3 * k)) for k in range(5)]
    jobs = [srs.commit_open_async(p, z) for p in polys]            # more jobs than the ring: the call itself throttles
    for p, job in zip(polys, jobs):
        comm, proof, val = job.wait()
        rc1, want_c = O.kzg_commit(curve, powers, p)
        rc2, want_w = O.kzg_open(curve, powers, p, z)
        assert rc1 == 0 and rc2 == 0
        assert (comm == want_c).all() and (proof == want_w).all() and (val == O.poly_eval(curve, p, z)).all()
    # the blocking calls of the same group agree
    c2, _ = srs.msm(polys[0], montgomery=True)
    w2, v2 = srs.kzg_open(polys[0], z)
    assert (c2 == O.kzg_commit(curve, powers, polys[0])[1]).all() and (w2 == O.kzg_open(curve, powers, polys[0], z)[1]).all()
    # device-resident shards: shard d on device d (all device 0 here), cut the way the key is cut
    p = polys[1]
    per = (n + ndev - 1) // ndev
    shards = [torch.from_numpy(np.ascontiguousarray(p[min(len(p), d * per):min(len(p), (d + 1) * per)]).view(np.int64)).cuda() for d in range(ndev)]
    torch.cuda.synchronize()
    comm, proof, val = srs.commit_open_async([t.data_ptr() if t.numel() else 0 for t in shards], z, n=len(p)).wait()
    assert (comm == O.kzg_commit(curve, powers, p)[1]).all() and (proof == O.kzg_open(curve, powers, p, z)[1]).all()
    srs.free()
    g.close()


@pytest.mark.parametrize("ndev,rows", [(1, 7), (2, 8), (3, 7), (4, 16), (4, 3), (8, 32), (8, 19)])
def test_group_ligero_commit_chained_digests(ndev, rows):
    """pc_hip_group_ligero_commit: rows encoded on different device contexts, column digests chained through them
    (pc_hip_column_hash_part), tree on the last -- node array and leaves identical to the single-context pc_hip_ligero_commit."""
    import poly_commit_amd as pc
    curve, in_cols, log_n = "bls12_381", 40, 8
    mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x11CE + rows, rows * in_cols)).reshape(rows, in_cols, 4)
    ctx = pc.Context(0)
    want_nodes, want_leaves = ctx.ligero_commit(curve, mat, log_n)
    g = pc.Group([0] * ndev)
    for col_hash, tree_hash in (("blake2s", "sha256"), ("sha256", "blake2s")):
        if col_hash != "blake2s":
            want_nodes, want_leaves = ctx.ligero_commit(curve, mat, log_n, col_hash=col_hash, tree_hash=tree_hash)
        nodes, leaves = g.ligero_commit(curve, mat, log_n, col_hash, tree_hash)
        assert (leaves == want_leaves).all() and (nodes == want_nodes).all(), (ndev, rows, col_hash)
    g.close()
    ctx.close()
