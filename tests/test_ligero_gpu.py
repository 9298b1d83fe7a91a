"""GPU parity for LinearCodePCS (univariate Ligero) commit / open / check through tests/harness/ligero.py against the
Python restatement in oracle/pyref.py: commitment (root, shape), opening proof (v, queried columns, Merkle paths,
well-formedness vector) bit for bit; then both verifiers on honest and altered proofs."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve,poly_len,wf", [("bn254", 300, True), ("bls12_381", 1 << 12, False), ("bls12_381", 5000, True)])
def test_ligero_commit_open_check_vs_oracle(ctx, curve, poly_len, wf):
    import torch
    from harness import ligero
    fr = R.CURVES[curve]["fr"]
    p = R.FIELDS[fr]["p"]
    co = R.gen_scalars(fr, 0x810, poly_len)
    want = R.ligero_commit(fr, co)
    m = lambda v: O.fr_mont_array(curve, v)                          # noqa: E731
    dev = torch.from_numpy(m(co).view(np.int64)).cuda()
    com, state = ligero.commit(ctx, curve, dev)
    assert (com["n_rows"], com["n_cols"], com["n_ext_cols"], com["root"]) == (want["n_rows"], want["n_cols"], want["n_ext_cols"], want["root"])
    z = R.gen_scalars(fr, 0x811, 1)[0]
    t = R.ligero_num_queries(fr, want["n_ext_cols"])
    assert t == ligero.calculate_t(p.bit_length(), 128, (3, 4), want["n_ext_cols"])
    idx = [(i * 7919 + 13) % want["n_ext_cols"] for i in range(t)]                 # what the caller's sponge would produce
    r = R.gen_scalars(fr, 0x812, want["n_rows"]) if wf else None
    want_pr = R.ligero_open(fr, want, z, idx, r)
    pr = ligero.open(ctx, curve, state, m([z])[0], idx, m(r) if wf else None)
    assert O.fr_from_mont_array(curve, pr["v"]) == want_pr["v"]
    assert [O.fr_from_mont_array(curve, c) for c in pr["columns"]] == want_pr["columns"]
    assert pr["paths"] == want_pr["paths"]
    assert (pr["well_formedness"] is None) == (not wf) and (not wf or O.fr_from_mont_array(curve, pr["well_formedness"]) == want_pr["well_formedness"])
    value = R.poly_eval(fr, co, z)
    args = (ctx, curve, com, m([z])[0])
    assert ligero.check(*args, m([value])[0], pr, idx, m(r) if wf else None) is True
    assert R.ligero_check(fr, want, z, value, want_pr, idx, r) is True
    assert ligero.check(*args, m([(value + 1) % p])[0], pr, idx, m(r) if wf else None) is False
    bad = dict(pr); bad["v"] = pr["v"].copy(); bad["v"][0, 0] ^= np.uint64(1)
    with pytest.raises(ligero.InvalidCommitment):
        ligero.check(*args, m([value])[0], bad, idx, m(r) if wf else None)
    bad = dict(pr); bad["columns"] = pr["columns"].copy(); bad["columns"][1, 0, 0] ^= np.uint64(1)
    with pytest.raises(ligero.InvalidCommitment):
        ligero.check(*args, m([value])[0], bad, idx, m(r) if wf else None)
    shifted = [idx[1]] + idx[1:]
    if shifted != idx:
        with pytest.raises(ligero.InvalidCommitment):
            ligero.check(*args, m([value])[0], pr, shifted, m(r) if wf else None)



def test_multilinear_ligero_device(ctx):
    """MultilinearLigero through the same device entry points (rho_inv = 2, tensor_vec tensors): opening and both verifiers
    against the restatement and the multilinear extension's value."""
    import torch
    from harness import ligero
    curve, n_vars = "bls12_381", 10
    fr = R.CURVES[curve]["fr"]
    evals = R.gen_scalars(fr, 0x830, 1 << n_vars)
    point = R.gen_scalars(fr, 0x831, n_vars)
    want = R.ligero_commit(fr, evals, rho_inv=2)
    m = lambda v: O.fr_mont_array(curve, v)                          # noqa: E731
    com, state = ligero.commit(ctx, curve, torch.from_numpy(m(evals).view(np.int64)).cuda(), rho_inv=2)
    assert (com["n_rows"], com["n_cols"], com["n_ext_cols"], com["root"]) == (want["n_rows"], want["n_cols"], want["n_ext_cols"], want["root"])
    ab = ligero.multilinear_tensor(curve, m(point), com["n_cols"])
    assert ab == R.ligero_multilinear_tensor(fr, point, want["n_cols"])
    idx = [(i * 911 + 3) % com["n_ext_cols"] for i in range(R.ligero_num_queries(fr, com["n_ext_cols"], rho_inv=2))]
    r = R.gen_scalars(fr, 0x832, com["n_rows"])
    pr = ligero.open(ctx, curve, state, None, idx, m(r), tensors=ab)
    want_pr = R.ligero_open(fr, want, None, idx, r, tensors=ab)
    assert O.fr_from_mont_array(curve, pr["v"]) == want_pr["v"] and pr["paths"] == want_pr["paths"]
    value = R.mle_evaluate(fr, evals, point)
    assert ligero.check(ctx, curve, com, None, m([value])[0], pr, idx, m(r), rho_inv=2, tensors=ab) is True
    assert ligero.check(ctx, curve, com, None, m([value + 1])[0], pr, idx, m(r), rho_inv=2, tensors=ab) is False


@pytest.mark.parametrize("curve,hash_name", [("bls12_381", "blake2s"), ("bn254", "sha256"), ("pallas", "blake2s")])
def test_column_digests_chained_over_row_slabs(ctx, curve, hash_name):
    """pc_hip_column_hash_part: the digests of the columns of a matrix whose rows arrive slab by slab (the rows of the encoded
    matrix on different devices, ShardedRows.commit) -- the chaining state of every column handed from slab to slab -- are the
    digests of pc_hip_column_hash over the whole matrix and of the oracle (FieldToBytesColHasher, bench-templates/src/lib.rs:
    327-337), for even slabs, an odd last slab, column ranges, one slab that is first and last; an odd slab that is not the last
    is refused."""
    import torch
    import poly_commit_amd as pc
    rows, n_cols = 23, 96
    mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x51AB, rows * n_cols)).reshape(rows, n_cols, 4)
    fr = R.CURVES[curve]["fr"]
    can = O.fr_from_mont_array(curve, np.ascontiguousarray(mat).reshape(-1, 4))
    want = np.stack([np.frombuffer(R.column_digest(fr, [can[r * n_cols + j] for r in range(rows)], hash_name), dtype=np.uint8) for j in range(n_cols)])
    assert (ctx.column_hash(curve, mat, hash_name) == want).all()
    dev = torch.from_numpy(np.ascontiguousarray(mat).view(np.int64)).cuda()
    for cuts in ([0, 23], [0, 8, 23], [0, 2, 4, 22, 23], [0, 10, 20, 23]):
        state = torch.zeros((n_cols, 12), dtype=torch.int32, device="cuda")
        out = torch.zeros((n_cols, 8), dtype=torch.int32, device="cuda")
        for k in range(len(cuts) - 1):
            lo, hi = cuts[k], cuts[k + 1]
            first, last = k == 0, k == len(cuts) - 2
            # two column ranges per slab, as the pipelined exchange of ShardedRows.commit issues them
            for c0, c1 in ((0, 40), (40, n_cols)):
                ctx.column_hash_part(curve, dev.data_ptr() + lo * n_cols * 32, hi - lo, n_cols, rows, state.data_ptr(), first, last,
                                     out.data_ptr(), hash_name, c0, c1 - c0)
        torch.cuda.synchronize()
        assert (out.cpu().numpy().view(np.uint8).reshape(n_cols, 32) == want).all(), cuts
    state = torch.zeros((n_cols, 12), dtype=torch.int32, device="cuda")
    with pytest.raises(pc.PcHipError):
        ctx.column_hash_part(curve, dev.data_ptr(), 3, n_cols, rows, state.data_ptr(), True, False, 0, hash_name)


@pytest.mark.parametrize("curve,rows,in_cols,log_n", [("bls12_381", 37, 200, 9), ("bn254", 64, 128, 8), ("pallas", 11, 33, 7)])
def test_host_to_host_commit_in_row_slabs(ctx, curve, rows, in_cols, log_n, monkeypatch):
    """pc_hip_ligero_commit with the matrix and the encoded matrix on the host runs in slabs of rows (one slab copied in and encoded
    while the ones before are copied out by helper threads; PC_HIP_LIGERO_SLAB_MB / _HELPERS / _PIN, read per call): encoded matrix,
    column digests and tree equal to the whole-matrix path of the same call (slab size 0) and to the device-resident path, which the
    tests above pin to the oracle -- for slabs of 2, 4 and 10 rows, an odd last slab, 1 to 4 helpers, and a slab size that leaves one
    slab (whole-matrix path)."""
    import torch
    n = 1 << log_n
    mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x57AB + rows, rows * in_cols)).reshape(rows, in_cols, 4)
    monkeypatch.setenv("PC_HIP_LIGERO_SLAB_MB", "0")
    ext0 = np.zeros((rows, n, 4), dtype=np.uint64)
    nodes0, leaves0 = ctx.ligero_commit(curve, mat, log_n, ext_out=ext0)
    dev = torch.from_numpy(mat.view(np.int64)).cuda()
    ext_dev = torch.zeros((rows, n, 4), dtype=torch.int64, device="cuda")
    nodes_d, leaves_d = ctx.ligero_commit(curve, dev, log_n, rows=rows, in_cols=in_cols, ext_out=ext_dev)
    assert (nodes_d == nodes0).all() and (leaves_d == leaves0).all() and (ext_dev.cpu().numpy().view(np.uint64) == ext0).all()
    row_mb = n * 32 / 1048576.0
    for slab_rows, helpers in ((2, "3"), (4, "1"), (10, "4"), (2, "2"), (rows, "3")):
        monkeypatch.setenv("PC_HIP_LIGERO_SLAB_MB", repr(slab_rows * row_mb * 1.01))
        monkeypatch.setenv("PC_HIP_LIGERO_HELPERS", helpers)      # threads copying slabs out (slab buffers: helpers + 1)
        ext = np.full((rows, n, 4), 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
        nodes, leaves = ctx.ligero_commit(curve, mat, log_n, ext_out=ext)
        assert (ext == ext0).all(), (curve, slab_rows)
        assert (leaves == leaves0).all() and (nodes == nodes0).all(), (curve, slab_rows)
        assert (ctx.ntt_batch(curve, mat, log_n) == ext0).all(), (curve, slab_rows, "pc_hip_ntt_batch host -> host takes the same slabs, without digests")
        nodes, _ = ctx.ligero_commit(curve, mat, log_n, ext_out=ext, want_leaves=False, col_hash="sha256", tree_hash="blake2s", len_prefix=False)
        monkeypatch.setenv("PC_HIP_LIGERO_SLAB_MB", "0")
        nodes_w, _ = ctx.ligero_commit(curve, mat, log_n, ext_out=ext, want_leaves=False, col_hash="sha256", tree_hash="blake2s", len_prefix=False)
        assert (nodes == nodes_w).all(), (curve, slab_rows, "sha256/blake2s")
