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


gold = {"msm": [], "ntt": [], "kzg": [], "ipa": [], "ligero_dims": [], "roots": {}, "gen_scalars": {}}
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
for field, bits in (("bls12_381_fr", 255), ("bn254_fr", 254)):
    for lg in (12, 16, 20, 22, 24):
        for rho in (4, 2):
            n, m, t = R.ligero_dimensions(field, 1 << lg, rho)
            gold["ligero_dims"].append({"field_bits": bits, "poly_len": 1 << lg, "rho_inv": rho, "n_rows": n,
                                        "n_cols": m, "t": t})
with open(os.path.join(OUT, "golden.json"), "w") as f:
    json.dump(gold, f, indent=0)
print("wrote", os.path.join(OUT, "golden.json"), os.path.getsize(os.path.join(OUT, "golden.json")), "bytes")
