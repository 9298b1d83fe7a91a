"""Randomised differential test of the IPA halving rounds (poly_commit_amd/ipa.py over the C ABI) against the CPU oracle: random curve,
size, fold-table form (none / one level / two levels, digit widths 2..5 or the library's choice), switch to the fixed key, resident or
host key, points at infinity among the generators, zero coefficients.  `python tools/ipa_fuzz.py [seconds] [seed]`; exit 1 on a mismatch."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
import oracle_lib as O
import poly_commit_amd as pc
from poly_commit_amd import ipa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pc.Context(0)
t0, cases, bad, kinds_seen = time.time(), 0, 0, {}
while time.time() - t0 < budget:
    curve = rng.choice(["pallas", "pallas", "bn254", "bls12_381"])
    lg = rng.randint(1, 11 if curve == "pallas" else 9)
    big = curve != "bls12_381" and rng.random() < 0.06                 # now and then a size where the late rounds get the cached fixed key (n0 >= 2^12)
    if big:
        lg = rng.randint(12, 13)
    n = 1 << lg
    key = O.gen_bases(curve, n + 1)
    for _ in range(rng.choice([0, 0, 1, 3])):
        key[rng.randrange(n)] = 0
    comm_key, h_prime = np.ascontiguousarray(key[:n]), key[n]
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, rng.getrandbits(30), n))
    if rng.random() < 0.2:
        coeffs[rng.randrange(n):] = 0
    point = O.f_to_mont(curve, 1, O.gen_scalars(curve, rng.getrandbits(30), 1))[0]
    ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, rng.getrandbits(30), lg))
    fkb = rng.choice([1 << 12, 1 << 13]) if big else rng.choice([0, 0, 2, 8, 64, 1 << 9, None])
    resident = rng.random() < 0.8
    form = None
    want = O.ipa_rounds(curve, comm_key, coeffs, point, np.ascontiguousarray(h_prime), ch)
    srs = ctx.upload_srs(curve, comm_key) if resident else comm_key
    if resident:
        if rng.random() < 0.7:
            srs.precompute(min_pairs=1)
        if n >= 2 and rng.random() < 0.85:
            form = rng.choice([(0, 0), (1, 2), (1, 4), (1, 5), (2, 2), (2, 3), (2, 4), (2, 5), (0, 3)])
            if form[0] == 2 and n < 4:
                form = (1, form[1])
            srs.precompute_fold(*form)
    for rep in range(rng.choice([1, 1, 2])):
        it = iter(range(lg))
        cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
        tm = {}
        got = ipa.ipa_open_rounds(ctx, curve, srs, cdev, n, point, h_prime, lambda L, R_: ch[next(it)], fixed_key_below=fkb, timings=tm,
                                  python_loop=rng.random() < 0.3)
        cases += 1
        for k in tm.get("ec_fold_kind", []):
            kinds_seen[k] = kinds_seen.get(k, 0) + 1
        if not all((a == b).all() for a, b in zip(got, want)):
            bad += 1
            print("MISMATCH", curve, n, "fkb", fkb, "resident", resident, "form", form, flush=True)
    if resident:
        srs.free()
print(f"ipa_fuzz: {cases} openings, folds by kind {kinds_seen}, {bad} mismatches")
sys.exit(1 if bad else 0)
