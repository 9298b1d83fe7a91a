"""Randomised differential test of the hipGraph replay path of pc_hip_msm (device scalars, at most 2^18 pairs: the library's default):
a few resident buffers and (length, offset) shapes called again and again in random order on one key per curve -- so that calls are run
plain, captured and replayed on all three pipelines in every interleaving -- with larger calls that make a pipeline's scratch grow,
pc_hip_ctx_trim, window-table builds and in-place key folds thrown in.  Every result against the CPU oracle.
`python tools/graph_fuzz.py [seconds] [seed]`; exit code 1 on any mismatch."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
import oracle_lib as O
import poly_commit_amd as pc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pc.Context(0)
NMAX = 1 << 16
t0, cases, bad, replays_possible = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    curve = rng.choice(["bls12_381", "bn254", "pallas"])
    n_key = rng.choice([1 << 10, 1 << 13, NMAX])
    bases = O.gen_bases(curve, n_key)
    host = [O.gen_scalars(curve, 0xF100 + k, n_key) for k in range(3)]
    if rng.random() < 0.3:
        host[1][::3] = 0
    dev = [torch.from_numpy(h.view(np.int64).copy()).cuda() for h in host]
    srs = ctx.upload_srs(curve, bases)
    cur = bases.copy()
    shapes = [(rng.randint(32, n_key), 0)] + [(rng.randint(32, n_key // 2), rng.randint(0, n_key // 2)) for _ in range(2)] + [(n_key, 0)]
    memo = {}
    seen = set()
    for step in range(rng.randint(20, 60)):
        ev = rng.random()
        if ev < 0.04:
            ctx.trim()
        elif ev < 0.08:
            srs.precompute(min_pairs=1); memo.clear()
        elif ev < 0.11 and n_key >= 64:
            u = O.f_to_mont(curve, 1, O.gen_scalars(curve, rng.randint(1, 1 << 30), 1))[0]
            srs.ec_fold(n_key // 2, u)                     # the key changes in place: same addresses, other points
            cur = srs.read(0, n_key).copy(); memo.clear()
        k = rng.randrange(3)
        n, off = rng.choice(shapes)
        mont = rng.random() < 0.5
        sc = host[k][:n]
        key = (k, n, off)
        if key not in memo:
            memo[key] = O.msm_pippenger(curve, np.ascontiguousarray(cur[off:off + n]), np.ascontiguousarray(sc[:min(n, n_key - off)]), 16, 1)
        ptr = dev[k].data_ptr()
        if mont:
            md = torch.from_numpy(O.f_to_mont(curve, 1, np.ascontiguousarray(host[k])).view(np.int64).copy()).cuda()
            got, _ = srs.msm(md.data_ptr(), n=n, base_offset=off, montgomery=True)
        else:
            got, _ = srs.msm(ptr, n=n, base_offset=off)
        replays_possible += (key, mont) in seen
        seen.add((key, mont))
        cases += 1
        if not (got == memo[key]).all():
            bad += 1
            print("MISMATCH", curve, n_key, key, "mont", mont, "step", step, flush=True)
    srs.free()
print(f"graph_fuzz: {cases} cases ({replays_possible} repeats of an earlier call), {bad} mismatches")
sys.exit(1 if bad else 0)
