#!/usr/bin/env python3
"""Writes a STAND-IN for tests/golden/ref_arkworks.json from oracle/pyref.py + the C++ oracle: the same schema, the same cases and
seeds as rust/ref-golden/src/main.rs, outputs computed by the restatements instead of the arkworks crates.

A rehearsal pins NOTHING (the restatement is compared with itself).  It exists so that the consumers of the real file
(tests/ref_golden_lib.py, tests/test_ref_golden_{cpu,gpu}.py) run against the schema in every test run, and so that the first
person with cargo can diff their ref_arkworks.json against a file of known shape.  The `generator` field says what it is and
`ref_golden_lib.is_reference_file` refuses it; this script refuses to write to tests/golden/ref_arkworks.json.

usage: python tools/ref_golden_rehearsal.py OUT.json [--small]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as O  # noqa: E402
import pyref as R  # noqa: E402
import ref_golden_lib as G  # noqa: E402

CURVES = ("bls12_381", "bn254", "pallas")


def cases(small=False):
    """The case list of rust/ref-golden/src/main.rs::main (keep the two in step)."""
    big = 1 << (10 if small else 14)
    kd = 1 << (8 if small else 12)
    return {
        "constants": [{"curve": c} for c in CURVES],
        "serialize": [{"curve": c} for c in CURVES],
        "msm": [{"curve": c, "n": 300, "seed": s} for c, s in zip(CURVES, (0x5EED0001, 0x5EED0100, 0x5EED0400))] +
               [{"curve": c, "n": big, "seed": s} for c, s in zip(CURVES, (0x5EED0002, 0x5EED0101, 0x5EED0401))],
        "kzg": [{"curve": "bls12_381", "degree": 31, "seed": 0x5EED0001, "z_seed": 7, "zero_low": 0},
                {"curve": "bls12_381", "degree": 31, "seed": 0x5EED0003, "z_seed": 8, "zero_low": 2},
                {"curve": "bls12_381", "degree": kd, "seed": 0x5EED0001, "z_seed": 7, "zero_low": 0},
                {"curve": "bn254", "degree": 31, "seed": 0x5EED0100, "z_seed": 7, "zero_low": 0},
                {"curve": "bn254", "degree": kd, "seed": 0x5EED0100, "z_seed": 7, "zero_low": 3}],
        "marlin_open": [{"curve": "bls12_381", "n": 256, "degrees": [255, 253, 128], "seed0": 0x5EED0200, "z_seed": 9},
                        {"curve": "bn254", "n": 256, "degrees": [255, 253, 128], "seed0": 0x5EED0210, "z_seed": 9}],
        "ipa": [{"curve": "pallas", "log_n": 4, "degrees": [15, 11], "seed0": 0x5EED0410, "z_seed": 11},
                {"curve": "pallas", "log_n": 6 if small else 10, "degrees": [(64 if small else 1024) - 1, (64 if small else 1024) - 5], "seed0": 0x5EED0420, "z_seed": 11}],
        "reed_solomon": [{"field": "bls12_381", "m": 512, "rho_inv": 4, "seed": 0x5EED0500},
                         {"field": "bn254", "m": 512, "rho_inv": 4, "seed": 0x5EED0501},
                         {"field": "bls12_381", "m": 300, "rho_inv": 4, "seed": 0x5EED0502}],
        "ligero": [{"field": "bls12_381", "poly_len": 1 << 12, "seed": 0x5EED0510, "rho_inv": 4, "sec_param": 128, "col_hash": "blake2s", "tree_hash": "sha256"},
                   {"field": "bn254", "poly_len": 1000, "seed": 0x5EED0511, "rho_inv": 4, "sec_param": 128, "col_hash": "blake2s", "tree_hash": "sha256"}],
    }


def stand_in_challenges(curve, seed, k):
    """The real file records what the Poseidon sponge squeezed (128-bit truncated challenges); the stand-in takes 128-bit values of
    a seeded stream -- any value is a valid input to the consumers."""
    fr = R.CURVES[curve]["fr"]
    return [hex(v & ((1 << 128) - 1)) for v in R.gen_scalars(fr, seed, k)]


def build(small=False):
    eng = G.OracleEngine()
    doc = {"generator": "REHEARSAL tools/ref_golden_rehearsal.py (oracle/pyref.py + oracle/oracle.cpp) -- NOT the reference, pins nothing",
           "schema": G.SCHEMA}
    for section, lst in cases(small).items():
        out = []
        for case in lst:
            case = dict(case)
            if section in G.STATIC_SECTIONS:
                case.update(G.STATIC_SECTIONS[section](case))
            else:
                if section == "marlin_open":
                    case["opening_challenges"] = stand_in_challenges(case["curve"], 0xC4A11, 3)
                if section == "ipa":
                    case["opening_challenges"] = stand_in_challenges(case["curve"], 0xC4A12, 2)
                case.update(G.SECTIONS[section](eng, case))
            out.append(case)
        doc[section] = out
    return doc


if __name__ == "__main__":
    out = sys.argv[1]
    if os.path.abspath(out) == os.path.abspath(G.REF_FILE):
        raise SystemExit("refusing to write a rehearsal file to tests/golden/ref_arkworks.json: that name is for the output of rust/ref-golden")
    json.dump(build("--small" in sys.argv), open(out, "w"), indent=1)
    print("wrote", out)
