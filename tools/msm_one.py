"""A few blocking MSMs of one curve and size over a true SRS with its window table (device-resident scalars): the workload the
rocprofv3 kernel-stats / PMC passes of the 8-limb curves run (BASELINE configs[3]: the Pallas cm_commit MSM at 2^22).
    python tools/msm_one.py <curve> <log_n> [reps=5]      -> one JSON line (phase brackets of the last call, shape)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc

curve, lg = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1 << lg
ctx = pc.Context(0)
ctx.set_timing(True)
g = O.gen_bases(curve, 1)[0]
beta = B.seed_fr(curve, 0xBE7A25)
pts = B.true_srs_points(ctx, curve, g, beta, 0, n)
srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)
del pts
srs.precompute()
co = B.rand_fr_device(0x5EED0100, n)
for _ in range(2):
    c, _ = srs.msm(co.data_ptr(), n=n, montgomery=True)
t = time.perf_counter()
for _ in range(reps):
    c, _ = srs.msm(co.data_ptr(), n=n, montgomery=True)
ms = (time.perf_counter() - t) / reps * 1e3
pb = B.from_mont_limbs(curve, O.poly_eval(curve, B.host_u64(co), B.mont_limbs(curve, beta)))
B.emit({"curve": curve, "n": n, "blocking_ms": ms, "phases_ms": [round(x, 3) for x in ctx.last_msm_phases_ms()[:6]], "shape": ctx.last_msm_shape(),
        "parity_ok": bool((c == B.oracle_scalar_mul(curve, g, pb)).all())})
