"""Blocking MSM latency vs size (device-resident scalars, key trimmed to the size, with and without its window table):
wall clock per call, the hipEvent phase brackets of the last call (digits+histogram, scan, scatter+fine sort, accumulate,
segmented reduction, bucket reduction) and the plan's shape {window bits, digits per scalar, buckets, table used}."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc
ctx = pc.Context(0)
ctx.set_timing(True)
if os.environ.get("PC_SWEEP_CHUNK"): ctx.set_msm_tuning(0, int(os.environ["PC_SWEEP_CHUNK"]))
out = {}
curves = os.environ.get("PC_SWEEP_CURVES", "bls12_381").split(",")
sizes = [int(x) for x in os.environ.get("PC_SWEEP_LOGS", "8,10,12,14,16,18,20").split(",")]
for curve in curves:
    nmax = 1 << max(sizes)
    bases = O.gen_bases(curve, nmax)
    sc = torch.from_numpy(O.f_to_mont(curve, 1, O.gen_scalars(curve, 3, nmax)).view(np.int64)).cuda()
    torch.cuda.synchronize()
    row = {}
    for lg in sizes:
        n = 1 << lg
        for table in (False, True):
            srs = ctx.upload_srs(curve, bases, n=n)
            if table: srs.precompute()
            for _ in range(3): srs.msm(sc.data_ptr(), n=n, montgomery=True)
            reps = 20
            t = time.perf_counter()
            for _ in range(reps): srs.msm(sc.data_ptr(), n=n, montgomery=True)
            ms = (time.perf_counter() - t) / reps * 1e3
            ph = ctx.last_msm_phases_ms()
            row[f"2^{lg}" + ("+table" if table else "")] = {"ms": round(ms, 3), "phases": [round(x, 3) for x in ph[:6]],
                                                           "sum_phases": round(sum(ph[:6]), 3), "shape": ctx.last_msm_shape()}
            srs.free()
    out[curve] = row
print(json.dumps(out))
