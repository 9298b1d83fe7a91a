"""The oracle against REFERENCE outputs: tests/golden/ref_arkworks.json, written by rust/ref-golden (the real arkworks crates on the
seeded inputs of oracle/pyref.py).  No Rust toolchain exists in the image this repository is built in, so the file is absent until
someone with cargo runs the recipe (rust/README.md): the reference-pinned tests SKIP with that reason and the rehearsal tests run
the same consumers against a stand-in of the same schema (tools/ref_golden_rehearsal.py), which pins nothing but keeps the plumbing
honest."""
import os
import sys

import pytest

import ref_golden_lib as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

DOC = G.load()
needs_reference_file = pytest.mark.skipif(not G.is_reference_file(DOC), reason=G.SKIP_REASON)


def _run_all(doc, engine, sections=None):
    bad = []
    for section in (sections or G.SECTIONS):
        for i, case in enumerate(doc.get(section, [])):
            if G.case_pairs(section, case) > engine.max_pairs:
                continue
            d = G.check_section(engine, section, case)
            if d:
                bad.append((engine.name, section, i, d))
    return bad


# ---- reference-pinned (skipped until the file exists) ---------------------------------------------------------------------------
@needs_reference_file
def test_constants_and_byte_conventions_match_arkworks():
    """Generators, moduli, FftField constants (GENERATOR, TWO_ADIC_ROOT_OF_UNITY), the seeded input streams, and CanonicalSerialize of
    points (both modes, both signs, infinity: the SWFlags bits and ark-bls12-381's own encoding) and of Fr."""
    bad = [(s, c["curve"], G.check_static(s, c)) for s in G.STATIC_SECTIONS for c in DOC[s]]
    assert not [b for b in bad if b[2]], bad


@needs_reference_file
def test_cpp_oracle_equals_arkworks():
    """msm_bigint, KZG10::commit / open, MarlinKZG10::open, InnerProductArgPC::open (whole proof), the Reed-Solomon rows and the
    Ligero commitment of the real crates == oracle/oracle.cpp."""
    assert _run_all(DOC, G.OracleEngine()) == []


@needs_reference_file
def test_pyref_equals_arkworks_small_cases():
    assert _run_all(DOC, G.PyrefEngine()) == []


def test_reference_file_state_is_reported():
    """Not a skip: states in every run whether parity is pinned by the reference."""
    if G.is_reference_file(DOC):
        assert set(G.SECTIONS) | set(G.STATIC_SECTIONS) <= set(DOC)
    else:
        assert DOC is None or not G.is_reference_file(DOC)
        print("\n[ref-golden] " + G.SKIP_REASON)


# ---- rehearsal: the same consumers on a stand-in of the same schema ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def rehearsal():
    import ref_golden_rehearsal as RG
    return RG.build(small=True)


def test_rehearsal_file_is_never_taken_for_the_reference(rehearsal):
    assert not G.is_reference_file(rehearsal) and rehearsal["schema"] == G.SCHEMA
    assert set(rehearsal) >= set(G.SECTIONS) | set(G.STATIC_SECTIONS)


def test_rehearsal_case_list_matches_the_rust_recipe():
    """The (section, size, seed) list of tools/ref_golden_rehearsal.py is the one rust/ref-golden/src/main.rs writes: every seed and
    size literal of the Rust main() appears in the Python list (the two files are kept in step by hand)."""
    import re
    import ref_golden_rehearsal as RG
    src = open(os.path.join(ROOT, "rust", "ref-golden", "src", "main.rs")).read()
    main = src[src.index("fn main()"):]
    seeds = {int(m.replace("_", ""), 16) for m in re.findall(r"0x5EED_[0-9A-F]{4}", main)}
    want = set()
    for lst in RG.cases().values():
        for c in lst:
            want |= {v for k, v in c.items() if k in ("seed", "seed0")}
    assert seeds == want, (sorted(hex(s) for s in seeds ^ want))
    for section in G.SECTIONS:
        assert f'"{section}"' in main


def test_rehearsal_pyref_agrees_with_the_oracle_written_stand_in(rehearsal):
    """pyref recomputes the small cases of a stand-in the C++ oracle wrote: two restatements, one schema."""
    assert _run_all(rehearsal, G.PyrefEngine(), ("msm", "kzg", "ipa", "reed_solomon")) == []


def test_rehearsal_detects_a_wrong_value(rehearsal):
    import copy
    doc = copy.deepcopy(rehearsal)
    doc["kzg"][0]["proof_w"][0] = hex(int(doc["kzg"][0]["proof_w"][0], 16) ^ 1)
    doc["ligero"][1]["commitment_uncompressed"] = doc["ligero"][1]["commitment_uncompressed"][:-2] + "00"
    bad = _run_all(doc, G.OracleEngine(), ("kzg", "ligero"))
    assert [(b[1], b[2]) for b in bad] == [("kzg", 0), ("ligero", 1)]
