"""Randomised differential test of pc_hip_msm against the CPU oracle: random sizes, chunk lengths, window widths (or the window
table), scalar distributions whose buckets span from a fraction of a chunk to many workgroups.  `python tools/msm_fuzz.py [seconds] [seed]`.
Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import oracle_lib as O
import poly_commit_amd as pc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pc.Context(0)
NMAX = 1 << 17
bases = {c: O.gen_bases(c, NMAX) for c in ("bls12_381", "bn254", "pallas")}
pool = {c: O.gen_scalars(c, 0xF022, NMAX) for c in bases}
t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    curve = rng.choice(list(bases))
    n = rng.choice([rng.randint(1, 300), rng.randint(300, 5000), rng.randint(5000, NMAX)])
    T = rng.choice([0, 0, 1, 2, 3, 5, 8, 13, 16, 40])
    table = rng.random() < 0.5
    c = 0 if table else rng.choice([0, 0, 4, 7, 9, 11, 13])
    kind = rng.choice(["uniform", "few", "equal", "runs", "sparse", "small", "mixed"])
    idx = np.arange(n)
    src = pool[curve]
    if kind == "uniform": sc = src[:n]
    elif kind == "few": sc = src[idx % rng.randint(2, 9)]
    elif kind == "equal": sc = np.repeat(src[rng.randint(0, 99):][:1], n, axis=0)
    elif kind == "runs": sc = src[idx // rng.randint(2, 4000)]
    elif kind == "sparse": sc = np.where((idx % rng.randint(2, 50) == 0)[:, None], src[:n], 0).astype(np.uint64)
    elif kind == "small": sc = src[:n] & np.array([0xFFFFFFFFFFFFFFFF, rng.choice([0, 0xFFFF]), 0, 0], dtype=np.uint64)
    else: sc = np.where((idx % 3 == 0)[:, None], src[idx % 7], src[:n])
    sc = np.ascontiguousarray(sc)
    ctx.set_msm_tuning(c, T)
    srs = ctx.upload_srs(curve, bases[curve], n=n)
    if table: srs.precompute()
    got, _ = srs.msm(sc)
    srs.free()
    want = O.msm_pippenger(curve, bases[curve][:n], sc, 16, 1)
    cases += 1
    if not (got == want).all():
        bad += 1
        print("MISMATCH", curve, n, "T", T, "c", c, "table", table, kind, flush=True)
ctx.set_msm_tuning(0, 0)
print(f"msm_fuzz: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
