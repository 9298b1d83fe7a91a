"""GPU parity for HyraxPC (poly-commit/src/hyrax/mod.rs) through tests/harness/hyrax.py: the row commitments (one
pc_hip_msm_many pass), the opening proof and the verifier against the Python big-int restatement in oracle/pyref.py,
bit for bit; at the 2^20-evaluation size (1024 rows of 1024 pairs) through the verifier's equations and sampled rows."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as R

pytestmark = pytest.mark.gpu


def _inputs(curve, n_vars, seed=0x4A0):
    fr = R.CURVES[curve]["fr"]
    dim = 1 << (n_vars // 2)
    pts = O.gen_bases(curve, dim + 1)
    evals = R.gen_scalars(fr, seed, 1 << n_vars)
    rands = R.gen_scalars(fr, seed + 1, dim)
    point = R.gen_scalars(fr, seed + 2, n_vars)
    rnd = R.gen_scalars(fr, seed + 3, dim + 3)            # r_eval, d[dim], r_d, r_b in the reference's draw order
    c = R.gen_scalars(fr, seed + 4, 1)[0]
    return dim, pts, evals, rands, point, rnd, c


@pytest.mark.parametrize("curve,n_vars", [("bn254", 8), ("pallas", 10), ("bls12_381", 6), ("bn254", 2)])
def test_hyrax_commit_open_check_vs_oracle(ctx, curve, n_vars):
    import torch
    from harness import hyrax
    dim, pts, evals, rands, point, rnd, c = _inputs(curve, n_vars)
    key_i, h_i = O.array_to_points(curve, pts[:dim]), O.array_to_points(curve, pts[dim:dim + 1])[0]
    want_rows, mat_i = R.hyrax_commit(curve, key_i, h_i, evals, rands)
    want_proof, want_eval = R.hyrax_open(curve, key_i, h_i, mat_i, rands, point, rnd[0], rnd[1:1 + dim], rnd[1 + dim], rnd[2 + dim], c)
    assert R.hyrax_check(curve, key_i, h_i, want_rows, point, want_proof, c) is True
    assert want_eval == R.mle_evaluate(R.CURVES[curve]["fr"], evals, point)

    m = lambda v: O.fr_mont_array(curve, v)                          # noqa: E731
    key = hyrax.HyraxKey(ctx, curve, pts[:dim], pts[dim])
    ev_dev = torch.from_numpy(m(evals).view(np.int64)).cuda()
    row_coms, state = hyrax.commit(key, ev_dev, m(rands))
    assert O.array_to_points(curve, row_coms) == want_rows
    proof, ev = hyrax.open(key, state, m(point), m([rnd[0]])[0], m(rnd[1:1 + dim]), m([rnd[1 + dim]])[0], m([rnd[2 + dim]])[0], m([c])[0])
    assert O.fr_from_mont_array(curve, ev.reshape(1, 4))[0] == want_eval
    pt = lambda a: O.array_to_points(curve, np.ascontiguousarray(a).reshape(1, -1))[0]      # noqa: E731
    assert (pt(proof[0]), pt(proof[1]), pt(proof[2])) == want_proof[:3]
    assert O.fr_from_mont_array(curve, proof[3]) == want_proof[3]
    assert O.fr_from_mont_array(curve, np.stack([proof[4], proof[5]])) == list(want_proof[4:])
    assert hyrax.check(key, row_coms, m(point), proof, m([c])[0]) is True
    # altered proofs / commitments are rejected, malformed inputs raise the reference's errors
    bad = list(proof); bad[5] = m([(want_proof[5] + 1) % R.FIELDS[R.CURVES[curve]["fr"]]["p"]])[0]
    assert hyrax.check(key, row_coms, m(point), tuple(bad), m([c])[0]) is False
    bad = list(proof); bad[3] = proof[3].copy(); bad[3][0, 0] ^= np.uint64(1)
    assert hyrax.check(key, row_coms, m(point), tuple(bad), m([c])[0]) is False
    if dim > 1:
        swapped = row_coms.copy(); swapped[[0, 1]] = swapped[[1, 0]]
        assert hyrax.check(key, swapped, m(point), proof, m([c])[0]) is False
        with pytest.raises(hyrax.IncorrectCommitmentSize):
            hyrax.check(key, row_coms[:-1], m(point), proof, m([c])[0])
    with pytest.raises(hyrax.InvalidNumberOfVariables):
        hyrax.check(key, row_coms, m(point)[:-1], proof, m([c])[0])
    with pytest.raises(hyrax.InvalidNumberOfVariables):
        hyrax.commit(key, ev_dev[: 1 << (n_vars - 1)], m(rands))
    key.close()


def test_hyrax_2p20_evaluations_bn254(ctx):
    """BASELINE-scale Hyrax: 2^20 evaluations = 1024 row commitments of 1024 pairs (+ the hiding term) in one
    pc_hip_msm_many pass; sampled rows against the oracle's Pippenger, the whole opening through the verifier."""
    import torch
    from harness import hyrax
    curve, n_vars = "bn254", 20
    dim = 1 << (n_vars // 2)
    pts = O.gen_bases(curve, dim + 1)
    evals = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4B0, 1 << n_vars))
    rands = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4B1, dim))
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4B2, n_vars))
    rnd = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4B3, dim + 3))
    c = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4B4, 1))[0]
    key = hyrax.HyraxKey(ctx, curve, pts[:dim], pts[dim])
    ev_dev = torch.from_numpy(evals.view(np.int64)).cuda()
    row_coms, state = hyrax.commit(key, ev_dev, rands)
    ext_key = np.ascontiguousarray(pts[:dim + 1])
    for row in (0, 1, 517, dim - 1):
        sc = np.concatenate([evals[row::dim], rands[row:row + 1]])          # row r = flat[r], flat[dim + r], ... || r_row
        want = O.msm_pippenger(curve, ext_key, O.f_from_mont(curve, 1, np.ascontiguousarray(sc)), 8, 1)
        assert (row_coms[row] == want).all()
    proof, ev = hyrax.open(key, state, point, rnd[0], rnd[1:1 + dim], rnd[1 + dim], rnd[2 + dim], c)
    assert hyrax.check(key, row_coms, point, proof, c) is True
    bad = list(proof); bad[4] = bad[4].copy(); bad[4][0] ^= np.uint64(2)
    assert hyrax.check(key, row_coms, point, tuple(bad), c) is False
    key.close()


@pytest.mark.parametrize("curve,n_vars", [("bn254", 8), ("pallas", 6), ("bls12_381", 4)])
def test_hyrax_cpp_host_mirror(curve, n_vars, tmp_path):
    """The same through the C++ host mirror (poly_commit_amd/host/hyrax.hpp): what the Rust shim would do, in the
    language that builds here.  The driver also runs the mirror's `check` (honest / altered / malformed inputs)."""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dim, pts, evals, rands, point, rnd, c = _inputs(curve, n_vars, seed=0x4C0)
    key_i, h_i = O.array_to_points(curve, pts[:dim]), O.array_to_points(curve, pts[dim:dim + 1])[0]
    want_rows, mat_i = R.hyrax_commit(curve, key_i, h_i, evals, rands)
    want_proof, want_eval = R.hyrax_open(curve, key_i, h_i, mat_i, rands, point, rnd[0], rnd[1:1 + dim], rnd[1 + dim], rnd[2 + dim], c)
    libdir = os.path.join(root, "poly_commit_amd")
    exe = os.path.join(root, "tests", "cpp", "hyrax_driver")
    src = exe + ".cpp"
    deps = [src, os.path.join(libdir, "libpc_hip.so")] + [os.path.join(libdir, "host", f) for f in os.listdir(os.path.join(libdir, "host"))]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src, "-L" + libdir, "-lpc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    m = lambda v: O.fr_mont_array(curve, v)                          # noqa: E731
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<II", O.CURVES[curve], n_vars))
        f.write(np.ascontiguousarray(pts[:dim + 1]).tobytes())
        for v in (evals, rands, point, [rnd[0]], rnd[1:1 + dim], [rnd[1 + dim]], [rnd[2 + dim]], [c]):
            f.write(m(v).tobytes())
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    aw = pts.shape[1]
    raw = np.fromfile(fout, dtype=np.uint64)
    rows = raw[:dim * aw].reshape(dim, aw)
    coms = raw[dim * aw:(dim + 3) * aw].reshape(3, aw)
    rest = raw[(dim + 3) * aw:].reshape(-1, 4)
    assert O.array_to_points(curve, rows) == want_rows
    assert tuple(O.array_to_points(curve, coms)) == want_proof[:3]
    got = O.fr_from_mont_array(curve, rest)
    assert got[:dim] == want_proof[3] and got[dim:dim + 2] == list(want_proof[4:]) and got[dim + 2] == want_eval
