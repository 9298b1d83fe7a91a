"""The HIP library against REFERENCE outputs (tests/golden/ref_arkworks.json, written by rust/ref-golden from the real arkworks crates)
through the C ABI.  Skipped, loudly, while the file does not exist (no Rust toolchain in this image: rust/README.md has the one
command); the rehearsal test below runs the very same consumer against a stand-in of the same schema written by the CPU oracle, so
that the day the file appears nothing but the data is new."""
import os
import sys

import pytest

import ref_golden_lib as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu

DOC = G.load()


def _run_all(doc, engine):
    bad = []
    for section in G.SECTIONS:
        for i, case in enumerate(doc.get(section, [])):
            d = G.check_section(engine, section, case)
            if d:
                bad.append((engine.name, section, i, d))
    return bad


@pytest.mark.skipif(not G.is_reference_file(DOC), reason=G.SKIP_REASON)
def test_hip_library_equals_arkworks(ctx):
    """pc_hip_msm (table-free and window table), KZG commit / pc_hip_kzg_open, pc_hip_fr_lincomb, the whole IPA proof through
    poly_commit_amd/ipa.py, pc_hip_ntt_batch and pc_hip_ligero_commit == the real crates' outputs."""
    assert _run_all(DOC, G.HipEngine(ctx)) == []


def test_rehearsal_hip_library_equals_the_oracle_written_stand_in(ctx):
    """Same consumer, stand-in data (tools/ref_golden_rehearsal.py: the C++ oracle's outputs in the reference file's schema)."""
    import ref_golden_rehearsal as RG
    doc = RG.build(small=False)
    assert not G.is_reference_file(doc)
    assert _run_all(doc, G.HipEngine(ctx)) == []
