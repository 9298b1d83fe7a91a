#!/usr/bin/env python3
"""Generate tests/golden/protocols.json: digests of whole protocol transcripts produced by oracle/pyref.py
(InnerProductArgPC with hiding and degree bounds, HyraxPC, univariate and multilinear Ligero) on fixed seeds.

The device paths are compared with pyref directly in the -m gpu tests; these digests pin pyref ITSELF, so that an
accidental change of a convention in the restatement (byte order, challenge order, tensor order) shows up as a diff of
a committed file and not only as an agreement of two things that changed together."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref as R  # noqa: E402


def digest(obj):
    """SHA-256 over a canonical JSON rendering (ints as hex, bytes as hex, points as [x, y] or null)."""
    def canon(o):
        if o is None or isinstance(o, (bool, str)):
            return o
        if isinstance(o, int):
            return hex(o)
        if isinstance(o, (bytes, bytearray)):
            return "0x" + bytes(o).hex()
        if isinstance(o, dict):
            return {k: canon(o[k]) for k in sorted(o)}
        return [canon(x) for x in o]
    return hashlib.sha256(json.dumps(canon(obj), sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def ipa_general_case(curve, n=8):
    fr = R.CURVES[curve]["fr"]
    pts = R.gen_bases(curve, n + 2)
    key, h, s = pts[:n], pts[n], pts[n + 1]
    co = [R.gen_scalars(fr, 0x700 + j, n - 1 - j) for j in range(3)]
    spec = [(None, False), (n - 2, True), (n - 2, False)]
    rs = R.gen_scalars(fr, 0x710, 6)
    polys = []
    for j, (db, hid) in enumerate(spec):
        rand = rs[2 * j] if hid else 0
        srand = rs[2 * j + 1] if (hid and db is not None) else 0
        comm, sh = R.ipa_commit_general(curve, key, s, co[j], db, rand, srand)
        polys.append(dict(coeffs=co[j], comm=comm, shifted_comm=sh, degree_bound=db, hiding=hid, rand=rand, shifted_rand=srand))
    ch = R.gen_scalars(fr, 0x720, 2 * len(polys) + 1)
    point = R.gen_scalars(fr, 0x730, 1)[0]
    hp = R.gen_scalars(fr, 0x740, n)
    hr = R.gen_scalars(fr, 0x741, 1)[0]
    proof = R.ipa_open_general(curve, key, h, s, polys, point, ch, hp, hr)
    return {"comms": [[q["comm"], q["shifted_comm"]] for q in polys], "proof": list(proof)}


def hyrax_case(curve, n_vars=4):
    fr = R.CURVES[curve]["fr"]
    dim = 1 << (n_vars // 2)
    pts = R.gen_bases(curve, dim + 1)
    evals = R.gen_scalars(fr, 0x3A0, 1 << n_vars)
    rands = R.gen_scalars(fr, 0x3A1, dim)
    point = R.gen_scalars(fr, 0x3A2, n_vars)
    rnd = R.gen_scalars(fr, 0x3A3, dim + 3)
    c = R.gen_scalars(fr, 0x3A4, 1)[0]
    rows, mat = R.hyrax_commit(curve, pts[:dim], pts[dim], evals, rands)
    proof, value = R.hyrax_open(curve, pts[:dim], pts[dim], mat, rands, point, rnd[0], rnd[1:1 + dim], rnd[1 + dim], rnd[2 + dim], c)
    return {"row_coms": rows, "proof": list(proof), "value": value}


def ligero_case(field, multilinear):
    if multilinear:
        evals = R.gen_scalars(field, 0x820, 1 << 8)
        st = R.ligero_commit(field, evals, rho_inv=2)
        point = R.gen_scalars(field, 0x821, 8)
        ab = R.ligero_multilinear_tensor(field, point, st["n_cols"])
        idx = [(i * 31 + 7) % st["n_ext_cols"] for i in range(10)]
        r = R.gen_scalars(field, 0x822, st["n_rows"])
        pr = R.ligero_open(field, st, None, idx, r, tensors=ab)
    else:
        co = R.gen_scalars(field, 0x800, 300)
        st = R.ligero_commit(field, co)
        z = R.gen_scalars(field, 0x801, 1)[0]
        idx = [(i * 37 + 5) % st["n_ext_cols"] for i in range(12)]
        r = R.gen_scalars(field, 0x802, st["n_rows"])
        pr = R.ligero_open(field, st, z, idx, r)
    return {"shape": [st["n_rows"], st["n_cols"], st["n_ext_cols"]], "root": st["root"], "v": pr["v"], "well_formedness": pr["well_formedness"],
            "columns": pr["columns"], "paths": [[p[0], p[1], p[2]] for p in pr["paths"]]}


def build():
    out = {"ipa_general": {}, "hyrax": {}, "ligero": {}}
    for curve in R.CURVES:
        out["ipa_general"][curve] = digest(ipa_general_case(curve))
        out["hyrax"][curve] = digest(hyrax_case(curve))
    out["ligero"]["bn254_fr:univariate"] = digest(ligero_case("bn254_fr", False))
    out["ligero"]["bls12_381_fr:multilinear"] = digest(ligero_case("bls12_381_fr", True))
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "protocols.json")
    json.dump(build(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
