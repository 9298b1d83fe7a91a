"""The 8-limb curves by themselves (BASELINE configs[2] and [3]): one blocking MSM with its hipEvent phase brackets and the plan's
shape, the 64-polynomial BN254 batch, the Pallas commit -- for one build of the library (PC_HIP_LIB selects a tuning variant built
by poly_commit_amd/build.py with PC_HIP_VARIANT / PC_HIP_CXXFLAGS).  One JSON line on stdout.
    python tools/n8_probe.py [log_bn254=20] [log_pallas=22] [polys=64]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B          # (redirects fd 1 to stderr; results go through B.emit)
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc
from poly_commit_amd import sharded

lg_bn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lg_pa = int(sys.argv[2]) if len(sys.argv) > 2 else 22
polys = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ctx = pc.Context(0)
ctx.set_timing(True)
out = {"lib": os.environ.get("PC_HIP_LIB", "default")}


def single(curve, n, reps=10):
    g = O.gen_bases(curve, 1)[0]
    beta = B.seed_fr(curve, 0xBE7A25)
    pts = B.true_srs_points(ctx, curve, g, beta, 0, n)
    srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)
    del pts
    srs.precompute()
    co = B.rand_fr_device(0x5EED0100, n)
    for _ in range(3):
        c, _ = srs.msm(co.data_ptr(), n=n, montgomery=True)
    t = time.perf_counter()
    for _ in range(reps):
        srs.msm(co.data_ptr(), n=n, montgomery=True)
    ms = (time.perf_counter() - t) / reps * 1e3
    ph = ctx.last_msm_phases_ms()
    shape = ctx.last_msm_shape()
    pb = B.from_mont_limbs(curve, O.poly_eval(curve, B.host_u64(co), B.mont_limbs(curve, beta)))
    ok = bool((c == B.oracle_scalar_mul(curve, g, pb)).all())
    adds = n * shape["digits_per_scalar"]
    r = {"n": n, "blocking_ms": ms, "phases_ms": [round(x, 3) for x in ph[:6]], "shape": shape, "parity_ok": ok,
         "accumulate_madd_per_s": adds / (ph[3] * 1e-3) if ph[3] > 0 else None}
    return r, srs, co, g, beta


r, srs, co, g, beta = single("bn254", (1 << lg_bn) + 1)
out["bn254_single"] = r
# the batch: `polys` commitments over the same key (pc_hip_msm_batch: many-MSM passes of 8)
n = (1 << lg_bn) + 1
vec = [B.rand_fr_device(0x5EED0100 + j * 131, n) for j in range(polys)]
ptrs = [v.data_ptr() for v in vec]
eng = sharded.HipEngine(ctx, "bn254")
eng.srs = srs
job = sharded.ShardedBatch(eng, "bn254", 0, 1, None)
for _ in range(2):
    res = job.commit_batch(vec, [n] * polys)
torch.cuda.synchronize()
t = time.perf_counter()
steps = 4
for _ in range(steps):
    res = job.commit_batch(vec, [n] * polys)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / steps * 1e3
bm = B.mont_limbs("bn254", beta)
ok = True
for j in (0, polys // 2, polys - 1):
    pb = B.from_mont_limbs("bn254", ctx.poly_eval("bn254", vec[j].data_ptr(), bm, n=n))
    ok = ok and bool((res[j] == B.oracle_scalar_mul("bn254", g, pb)).all())
out["bn254_batch"] = {"polys": polys, "ms_per_step": ms, "ms_per_commitment": ms / polys, "parity_ok": ok, "shape": ctx.last_msm_shape()}
srs.free()
del vec, co
torch.cuda.empty_cache()

r, srs, co, g, beta = single("pallas", 1 << lg_pa)
out["pallas_single"] = r
srs.free()
B.emit(out)
