"""MSM timing on adversarial scalar distributions (2^20 pairs, BLS12-381)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import oracle_lib as O
import poly_commit_amd as pc
curve, n = "bls12_381", 1 << 20
ctx = pc.Context(0)
bases = O.gen_bases(curve, n)
srs = ctx.upload_srs(curve, bases)
rnd = O.gen_scalars(curve, 5, n)
cases = {
    "uniform": rnd,
    "all_equal": np.ascontiguousarray(np.repeat(rnd[:1], n, axis=0)),
    "two_values": np.ascontiguousarray(np.where((np.arange(n) % 2 == 0)[:, None], rnd[:1], rnd[1:2])),
    "all_ones": np.ascontiguousarray(O.ints_to_limbs([1], 4).repeat(n, axis=0)),
    "90pct_zero": np.ascontiguousarray(np.where((np.arange(n) % 10 == 0)[:, None], rnd, 0).astype(np.uint64)),
    "small_64bit": np.ascontiguousarray(np.concatenate([rnd[:, :1], np.zeros((n, 3), dtype=np.uint64)], axis=1)),
}
out = {}
for name, sc in cases.items():
    srs.msm(sc)
    t = time.perf_counter()
    for _ in range(3):
        srs.msm(sc)
    out[name] = round((time.perf_counter() - t) / 3 * 1e3, 2)
print(json.dumps(out))
