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
    assert r.returncode == 0 and "built" in r.stdout and "fold bytes %d" % (131 * (1 << 11) * 64) in r.stdout, r.stdout + r.stderr[-2000:]
