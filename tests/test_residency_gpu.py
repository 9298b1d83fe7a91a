"""Residency accounting of the C ABI (pc_hip_ctx_bytes_resident / pc_hip_srs_bytes_resident / pc_hip_ctx_trim): what the Rust shim's
key cache budgets against, since MarlinKZG10::trim (marlin_pc/mod.rs:80-169) returns the key by value and cannot own device memory."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_bytes_resident_follow_uploads_tables_and_frees():
    import poly_commit_amd as pc
    ctx = pc.Context(0)
    base = ctx.bytes_resident()
    curve, n = "bls12_381", 1 << 14
    powers = O.gen_bases(curve, n)
    srs = ctx.upload_srs(curve, powers)
    a = ctx.bytes_resident()
    assert a["n_keys"] == base["n_keys"] + 1 and a["keys"] - base["keys"] == n * 96
    assert a["device_total"] - base["device_total"] >= n * 96
    k = srs.bytes_resident()
    assert k["bases"] == n * 96 and k["window_tables"] == 0 and k["fold_table"] == 0 and k["pipelines"] > 0
    srs.precompute()
    b = ctx.bytes_resident()
    k2 = srs.bytes_resident()
    assert k2["window_tables"] > 0 and k2["window_tables"] % (n * 128) == 0          # 96-byte points padded to 128-byte lines
    assert b["window_tables"] - a["window_tables"] == k2["window_tables"]
    # (precompute re-creates the key's pipelines for the table geometry: compare the totals without them)
    assert (b["device_total"] - k2["pipelines"]) - (a["device_total"] - k["pipelines"]) >= k2["window_tables"]
    # results unchanged by all of this
    sc = O.gen_scalars(curve, 5, n)
    want = O.msm_pippenger(curve, powers, sc, 8, 2)
    assert (srs.msm(sc)[0] == want).all()
    coeffs = O.f_to_mont(curve, 1, sc)
    z = coeffs[7]
    w0 = srs.kzg_open(coeffs, z)[0]
    assert ctx.bytes_resident()["scratch"] > 0
    ctx.trim()
    t = ctx.bytes_resident()
    assert t["scratch"] == 0 and t["keys"] == b["keys"] and t["window_tables"] == b["window_tables"]
    assert (srs.kzg_open(coeffs, z)[0] == w0).all() and (srs.msm(sc)[0] == want).all()      # everything comes back on demand
    srs.free()
    e = ctx.bytes_resident()
    assert e["n_keys"] == base["n_keys"] and e["keys"] == base["keys"] and e["window_tables"] == base["window_tables"]
    ctx.trim()
    assert ctx.bytes_resident()["device_total"] <= base["device_total"] + (1 << 20)
    ctx.close()


def test_fold_table_refused_above_the_free_memory_share():
    """pc_hip_srs_precompute_fold: 131 x n/2 points; refused (PC_ERR_UNSUPPORTED) instead of exhausting a shared device."""
    import os
    import subprocess
    import sys
    child = r'''
import sys, os
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import oracle_lib as O
import poly_commit_amd as pc
ctx = pc.Context(0)
srs = ctx.upload_srs("pallas", O.gen_bases("pallas", 1 << 12))
try:
    srs.precompute_fold()
    print("built")
except Exception as e:
    print("refused", e)
before = ctx.bytes_resident()["fold_tables"]
print("fold bytes", before)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, PC_HIP_FOLD_TABLE_MAX_FRAC="0.00000001"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "refused" in r.stdout and "fold bytes 0" in r.stdout, r.stdout + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-c", child], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    # (2^12 points: one level; the library's choice of digit width is the widest, 4: 4 x 131 rows of the upper half)
    assert r.returncode == 0 and "built" in r.stdout and "fold bytes %d" % (4 * 131 * (1 << 11) * 64) in r.stdout, r.stdout + r.stderr[-2000:]


def _child(code, env=None, timeout=600):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pre = "import sys, os\nROOT = %r\nsys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]\n" % root
    return subprocess.run([sys.executable, "-c", pre + code], env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)


def test_key_freed_after_its_context_was_shut_down():
    """pc_hip_shutdown releases what the keys still alive hold and leaves them as tombstones: a key dropped AFTER its context (Rust Drop
    order of an Arc<ResidentKey>) must not touch the deleted context (advisor, round 5)."""
    r = _child(r'''
import ctypes as C
import numpy as np
import oracle_lib as O
import poly_commit_amd as pc
from poly_commit_amd import _ffi
lib = _ffi.load_library()
ctx = pc.Context(0)
srs = ctx.upload_srs("bn254", O.gen_bases("bn254", 1 << 12))
srs.precompute(min_pairs=1)
sc = O.gen_scalars("bn254", 3, 1 << 12)
assert (srs.msm(sc)[0] == O.msm_pippenger("bn254", O.gen_bases("bn254", 1 << 12), sc, 8, 1)).all()
h, k = ctx.h, srs.h
lib.pc_hip_shutdown(h)          # the context goes first ...
lib.pc_hip_srs_free(k)          # ... the key after it
ctx.h = None; srs.h = None
ctx2 = pc.Context(0)            # and the device is as usable as before
srs2 = ctx2.upload_srs("bn254", O.gen_bases("bn254", 64))
assert (srs2.msm(sc[:64])[0] == O.msm_naive("bn254", O.gen_bases("bn254", 64), sc[:64])).all()
print("ok")
''')
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


def test_trim_gives_back_the_second_buffers_of_host_calls_in_parts():
    """A blocking MSM on HOST scalars above the split size allocates a second sort output and bucket array on its pipeline
    (MsmPlan::begin_parts); pc_hip_ctx_trim releases them (advisor, round 5) and the next host call brings them back."""
    r = _child(r'''
import numpy as np
import oracle_lib as O
import poly_commit_amd as pc
ctx = pc.Context(0)
n = 1 << 13
powers = O.gen_bases("bls12_381", n)
srs = ctx.upload_srs("bls12_381", powers)
sc = O.gen_scalars("bls12_381", 9, n)
want = O.msm_pippenger("bls12_381", powers, sc, 8, 1)
assert (srs.msm(sc)[0] == want).all()
a = srs.bytes_resident()["pipelines"]
ctx.trim()
b = srs.bytes_resident()["pipelines"]
assert b < a, (a, b)
assert (srs.msm(sc)[0] == want).all()
assert srs.bytes_resident()["pipelines"] == a
print("ok", a, b)
''', env={"PC_HIP_HOST_SPLIT_LOG2": "10"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]


@pytest.mark.parametrize("budget_mb", ["1", "0"])
def test_host_batch_staging_budget(budget_mb):
    """pc_hip_msm_batch on HOST polynomials stages G polynomials per pipeline on the device: with a budget too small for 8 (or for
    2: the per-polynomial pipeline) the call gives the same commitments and keeps at most the budget resident (advisor, round 5)."""
    r = _child(r'''
import numpy as np
import oracle_lib as O
import poly_commit_amd as pc
ctx = pc.Context(0)
curve, m, k = "bn254", 1 << 14, 10
powers = O.gen_bases(curve, m)
srs = ctx.upload_srs(curve, powers)
srs.precompute(min_pairs=1)
polys = [O.f_to_mont(curve, 1, O.gen_scalars(curve, 40 + j, m)) for j in range(k)]
base = ctx.bytes_resident()["device_total"]
out = srs.msm_batch(polys, [m] * k, montgomery=True, host=True)
for j in (0, 3, 9):
    assert (out[j] == O.msm_pippenger(curve, powers, O.f_from_mont(curve, 1, polys[j]), 8, 1)).all(), j
budget = int(os.environ["PC_HIP_BATCH_STAGE_MAX_MB"]) << 20
G = 8
while G >= 2 and G * m * 32 > budget: G //= 2
grown = ctx.bytes_resident()["device_total"] - base
print("ok G", G, grown)
''', env={"PC_HIP_BATCH_STAGE_MAX_MB": budget_mb})
    assert r.returncode == 0 and "ok G" in r.stdout, r.stdout + r.stderr[-3000:]
    assert ("ok G 2" if budget_mb == "1" else "ok G 1") in r.stdout
