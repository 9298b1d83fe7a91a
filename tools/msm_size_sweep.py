"""Blocking MSM latency vs size (device-resident scalars), BLS12-381 and BN254."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc
ctx = pc.Context(0)
ctx.set_timing(True)
out = {}
for curve in ("bls12_381", "bn254"):
    nmax = 1 << 20
    bases = O.gen_bases(curve, nmax)
    srs = ctx.upload_srs(curve, bases)
    sc = torch.from_numpy(O.f_to_mont(curve, 1, O.gen_scalars(curve, 3, nmax)).view(np.int64)).cuda()
    torch.cuda.synchronize()
    row = {}
    for lg in (6, 8, 10, 12, 14, 16, 18, 20):
        n = 1 << lg
        for _ in range(3): srs.msm(sc.data_ptr(), n=n, montgomery=True)
        t = time.perf_counter()
        reps = 10
        for _ in range(reps): srs.msm(sc.data_ptr(), n=n, montgomery=True)
        ms = (time.perf_counter() - t) / reps * 1e3
        ph = ctx.last_msm_phases_ms()
        row[f"2^{lg}"] = {"ms": round(ms, 3), "phases": [round(x, 2) for x in ph[:6]]}
    out[curve] = row
    srs.free()
print(json.dumps(out))
