#!/usr/bin/env python3
"""Generate tests/golden/*.json from oracle/pyref.py (pure-Python big ints).

The reference publishes no golden vectors for this path and cannot be run here (Rust,
un-vendored arithmetic crates), so these fixtures pin the C++ oracle and the HIP library to
an independent arbitrary-precision implementation of the published curve/field
definitions.  Values are canonical integers in hex (not Montgomery)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as R

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def hx(v):
    return None if v is None else hex(v)


def pt(P):
    return None if P is None else [hex(P[0]), hex(P[1])]


gold = {"msm": [], "ntt": [], "kzg": [], "ipa": [], "ligero_dims": [], "roots": {}, "gen_scalars": {},
        "ligero_commit": [], "fr_lincomb": [], "msm_many": []}
for curve in R.CURVES:
    fr = R.CURVES[curve]["fr"]
    r = R.FIELDS[fr]["p"]
    bases = R.gen_bases(curve, 12)
    # --- MSM known answers (incl. edge scalars) -------------------------------------------
    sc = R.gen_scalars(fr, 0x5EED0001, 12)
    cases = {
        "random": sc,
        "edge": [0, 1, r - 1, 2, r - 2, 1 << 254 if r > (1 << 254) else 1 << 253, 0, 7, sc[0], sc[0], 1, r - 1],
    }
    for name, s in cases.items():
        gold["msm"].append({"curve": curve, "name": name, "bases": [pt(b) for b in bases],
                            "scalars": [hex(x) for x in s], "result": pt(R.msm(curve, bases, s))})
    gold["gen_scalars"][curve] = [hex(x) for x in R.gen_scalars(fr, 0x5EED0001, 4)]
    # --- NTT: size-16 domain, 5 inputs (zero padded) --------------------------------------
    co = R.gen_scalars(fr, 77, 5)
    gold["ntt"].append({"curve": curve, "log_n": 4, "input": [hex(x) for x in co],
                        "output": [hex(x) for x in R.ntt_naive(fr, co, 4)]})
    gold["roots"][curve] = {"two_adic_root": hex(R.two_adic_root(fr)), "omega_2^4": hex(R.root_of_unity(fr, 4)),
                            "two_adicity": R.two_adicity(r)}
    # --- KZG commit / open (hiding off), with leading zeros ------------------------------
    coeffs = [0, 0] + R.gen_scalars(fr, 5, 9)
    z = R.gen_scalars(fr, 6, 1)[0]
    gold["kzg"].append({"curve": curve, "powers": [pt(b) for b in bases], "coeffs": [hex(x) for x in coeffs],
                        "z": hex(z), "commitment": pt(R.kzg_commit(curve, bases, coeffs)),
                        "witness": [hex(x) for x in R.witness_polynomial(fr, coeffs, z)],
                        "proof_w": pt(R.kzg_open(curve, bases, coeffs, z)),
                        "value": hex(R.poly_eval(fr, coeffs, z))})
    # --- IPA halving rounds, n = 4, fixed challenges -------------------------------------
    key = bases[:4]
    cs = R.gen_scalars(fr, 9, 4)
    zp = R.gen_scalars(fr, 10, 1)[0]
    ch = R.gen_scalars(fr, 11, 2)
    hp = bases[7]
    l, rr, fk, c = R.ipa_rounds(curve, key, cs, zp, hp, ch)
    gold["ipa"].append({"curve": curve, "comm_key": [pt(b) for b in key], "coeffs": [hex(x) for x in cs],
                        "point": hex(zp), "h_prime": pt(hp), "challenges": [hex(x) for x in ch],
                        "l_vec": [pt(x) for x in l], "r_vec": [pt(x) for x in rr], "final_comm_key": pt(fk),
                        "c": hex(c)})
    # --- Ligero commit steps 1-3 on a 4 x 8 matrix, rho^-1 = 4: encode, column digests, Merkle tree -----
    mat = [R.gen_scalars(fr, 0x5EED0500 + i, 8) for i in range(4)]
    ext = [R.ntt(fr, row, 5) for row in mat]                       # 8 * 4 = 32 = 2^5 evaluations per row
    cols = [[ext[i][j] for i in range(4)] for j in range(32)]
    entry = {"curve": curve, "matrix": [[hex(x) for x in row] for row in mat], "log_n": 5, "variants": []}
    for col_hash, tree_hash, lp in (("blake2s", "sha256", True), ("sha256", "blake2s", False)):
        leaves = [R.column_digest(fr, c, col_hash) for c in cols]
        nodes = R.merkle_tree(leaves, tree_hash, lp)
        sib, path = R.merkle_path(nodes, leaves, 5)
        entry["variants"].append({"col_hash": col_hash, "tree_hash": tree_hash, "len_prefix": lp,
                                  "leaves": [l.hex() for l in leaves], "root": nodes[0].hex(),
                                  "nodes": [x.hex() for x in nodes],
                                  "path_of_leaf_5": [sib.hex()] + [x.hex() for x in path]})
    gold["ligero_commit"].append(entry)
    # --- open's linear combination, ragged lengths ---------------------------------------------------
    polys = [R.gen_scalars(fr, 0x5EED0700 + j, n) for j, n in enumerate((9, 1, 5))]
    xi = R.gen_scalars(fr, 0x5EED0777, 3)
    gold["fr_lincomb"].append({"curve": curve, "polys": [[hex(x) for x in q] for q in polys], "xi": [hex(x) for x in xi],
                               "result": [hex(x) for x in R.fr_lincomb(fr, polys, xi)]})
    # --- many short MSMs over the same bases (Hyrax rows): 3 rows of 6 scalars over bases[:6] -------
    rows = [R.gen_scalars(fr, 0x4A11 + k, 6) for k in range(3)]
    rows[1] = [0] * 6
    gold["msm_many"].append({"curve": curve, "bases": [pt(b) for b in bases[:6]], "rows": [[hex(x) for x in row] for row in rows],
                             "results": [pt(R.msm(curve, bases[:6], row)) for row in rows]})
for field, bits in (("bls12_381_fr", 255), ("bn254_fr", 254)):
    for lg in (12, 16, 20, 22, 24):
        for rho in (4, 2):
            n, m, t = R.ligero_dimensions(field, 1 << lg, rho)
            gold["ligero_dims"].append({"field_bits": bits, "poly_len": 1 << lg, "rho_inv": rho, "n_rows": n,
                                        "n_cols": m, "t": t})
with open(os.path.join(OUT, "golden.json"), "w") as f:
    json.dump(gold, f, indent=0)
print("wrote", os.path.join(OUT, "golden.json"), os.path.getsize(os.path.join(OUT, "golden.json")), "bytes")
