"""The Rust sources have never met rustc (no toolchain in this image).  tools/rust_lint.py is the check that can run here: balanced
delimiters and generics, declared lifetimes, every `ffi::` symbol declared, every FFI call's arity and pointer / length order against
src/ffi.rs (which tools/check_ffi_decls.py ties to include/pc_hip.h), crate-internal paths that resolve, no duplicate fns.  The tree
must be clean, and the lint must FIND each kind of mistake when one is planted in a copy (a lint that finds nothing proves nothing)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rust_lint as L  # noqa: E402

SHIM = os.path.join(ROOT, "rust", "poly-commit-hip")
GOLDEN = os.path.join(ROOT, "rust", "ref-golden")


def test_both_crates_are_clean_and_the_lint_saw_the_code():
    for k in L.STATS:
        L.STATS[k] = 0
    for crate in (SHIM, GOLDEN):
        findings, n_files = L.lint_crate(crate)
        assert findings == [], "\n".join(findings)
        assert n_files >= 1
    # it looked at something: the shim makes some seventy FFI calls
    assert L.STATS["ffi_calls_checked"] >= 60 and L.STATS["ffi_names_checked"] >= 120
    assert L.STATS["paths_checked"] >= 30 and L.STATS["headers_checked"] >= 200


def _mutated(tmp_path, rel, old, new, count=1):
    dst = tmp_path / "crate"
    shutil.copytree(SHIM, dst)
    f = dst / rel
    src = f.read_text()
    assert old in src, (rel, old)
    f.write_text(src.replace(old, new, count))
    return L.lint_crate(str(dst))[0]


MUTATIONS = [
    # (file, old, new, what the lint must say)
    ("src/kzg10_hip.rs", "ffi::pc_hip_msm_batch(c.raw, key.srs, offs.as_ptr(), ptrs.as_ptr(), lens.as_ptr(), k,",
     "ffi::pc_hip_msm_batch(c.raw, key.srs, offs.as_ptr(), ptrs.as_ptr(), k,", "arguments"),                       # a dropped argument
    ("src/kzg10_hip.rs", "ffi::pc_hip_msm_batch(c.raw, key.srs, offs.as_ptr(), ptrs.as_ptr(), lens.as_ptr(), k,",
     "ffi::pc_hip_msm_batch(c.raw, key.srs, offs.as_ptr(), ptrs.as_ptr(), ptrs.len(), lens.as_ptr(),", "is an integer"),   # pointer / length swapped
    ("src/kzg10_hip.rs", "ffi::pc_hip_msm_batch(", "ffi::pc_hip_msm_batched(", "not declared"),                     # a symbol the library does not have
    ("src/marlin_kzg10.rs", "impl<E, P> PolynomialCommitment<E::ScalarField, P> for HipMarlinKZG10<E, P>",
     "impl<E, P> PolynomialCommitment<E::ScalarField, P> for HipMarlinKZG10<E, P", "balance"),                      # an unclosed generic list
    ("src/ipa_pc.rs", "fn check<'a>(vk: &Self::VerifierKey", "fn check(vk: &Self::VerifierKey", "lifetime"),        # a lifetime nobody declares
    ("src/kzg10_hip.rs", "pub fn msm_batch_host<G>", "pub fn msm_batch<G>", "defined twice"),                       # two fns of one name
    ("src/lib.rs", "pub mod device;", "pub mod devices;", "no file"),                                               # a module without a file
]


@pytest.mark.parametrize("rel,old,new,expect", MUTATIONS)
def test_lint_finds_a_planted_mistake(tmp_path, rel, old, new, expect):
    findings = _mutated(tmp_path, rel, old, new)
    assert any(expect in f for f in findings), (expect, findings[:5])


def test_lint_finds_an_unbalanced_brace_and_a_bad_path(tmp_path):
    findings = _mutated(tmp_path, "src/group.rs", "{", "{{", 1)
    assert any("never closed" in f or "unbalanced" in f for f in findings)


def test_lint_finds_a_path_to_nothing(tmp_path):
    findings = _mutated(tmp_path, "src/marlin_kzg10.rs", "use crate::curve::{", "use crate::curvee::{", 1)
    assert any("no module or item" in f for f in findings), findings[:5]
    findings = _mutated(tmp_path / "b", "src/marlin_kzg10.rs", "use crate::device::{self, check,", "use crate::device::{self, not_there, check,", 1)
    assert any("is not defined in" in f for f in findings), findings[:5]
