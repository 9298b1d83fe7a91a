"""Blocking commit / open of 2^k HOST coefficients (the trait-shaped calls) with the phase brackets of the MSM in parts:
    PC_HIP_HOST_PARTS=<count | weights> python tools/host_parts_probe.py [log_n=24]
PC_PROBE_COLD=1: the timed calls run after another key with its tables was built and freed.
One JSON line: wall ms of pc_hip_msm(PC_MEM_HOST) and pc_hip_kzg_open(PC_MEM_HOST), the resident MSM beside them, phases of the last call."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
curve = "bls12_381"
ctx = pc.Context(0)
g = O.gen_bases(curve, 1)[0]
beta = B.seed_fr(curve, 0xBE7A25)
pts = B.true_srs_points(ctx, curve, g, beta, 0, n)
srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)
del pts
srs.precompute()
co = B.rand_fr_device(0x5EED0100, n)
host = B.host_u64(co)
z = B.mont_limbs(curve, B.seed_fr(curve, 0x2A))
ctx.set_timing(True)
want_c, _ = srs.msm(co.data_ptr(), n=n, montgomery=True)
want_w, _ = srs.kzg_open(co.data_ptr(), z, n=n)
for _ in range(2):
    c, _ = srs.msm(host, montgomery=True); w, _ = srs.kzg_open(host, z)
out = {"parts": os.environ.get("PC_HIP_HOST_PARTS", "default"), "log_n": lg, "parity": bool((c == want_c).all() and (w == want_w).all())}
if os.environ.get("PC_PROBE_COLD"):      # a key with its tables, freed: from here on the runtime pins the host buffers anew in every copy (see tools/ligero_stream_probe.py)
    c2, m = "pallas", 1 << 22
    p2 = B.true_srs_points(ctx, c2, O.gen_bases(c2, 1)[0], B.seed_fr(c2, 0xA11CE5), 0, m + 1)
    s2 = ctx.upload_srs(c2, p2.data_ptr(), n=m)
    del p2
    s2.precompute(); s2.precompute_fold(); s2.free()
    out["state"] = "after a key was freed"
for name, fn in (("resident_msm", lambda: srs.msm(co.data_ptr(), n=n, montgomery=True)), ("host_commit", lambda: srs.msm(host, montgomery=True)),
                 ("resident_open", lambda: srs.kzg_open(co.data_ptr(), z, n=n)), ("host_open", lambda: srs.kzg_open(host, z))):
    t = time.perf_counter()
    for _ in range(4):
        fn()
    out[name + "_ms"] = round((time.perf_counter() - t) / 4 * 1e3, 2)
    out[name + "_phases"] = [round(float(x), 2) for x in ctx.last_msm_phases_ms()[:7]]
B.emit(out)
