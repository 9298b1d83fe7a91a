"""Randomised differential run of the scalar-field kernels against the CPU oracle, like tools/msm_fuzz.py for the MSM:
  div    pc_hip_poly_div_scan (kzg10/mod.rs:217-240 as a scan; random lengths, with and without a carry, host and device input,
         split at a random position and chained through the carry like the sharded open does)
  ntt    pc_hip_ntt_batch (linear_codes/utils.rs:112-127; random log_n <= 14, rows, ragged in_cols incl. non powers of two;
         whole output against the oracle's NTT and test_reed_solomon's property out[j] == row(omega^j) at random j, utils.rs:303-331)
  fold   pc_hip_ipa_fold_dots (ipa_pc/mod.rs:672,675,691-697; random power-of-two m, with and without the fold; folded vectors
         and both inner products against Python big ints)
`python tools/field_fuzz.py [seconds] [seed] [kinds]`.  One line per failure and a summary; exit code 1 on any mismatch."""
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
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["div", "ntt", "fold"]
CURVES = ("bls12_381", "bn254", "pallas")
ctx = pc.Context(0)
MOD = {c: None for c in CURVES}


def modulus(curve):
    if MOD[curve] is None:
        import pyref as R
        MOD[curve] = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
    return MOD[curve]


def rand_fr(curve, n, seed, kind):
    """n canonical scalars: uniform, sparse (mostly zero), tiny, all p - 1, or all equal"""
    s = O.gen_scalars(curve, seed, max(n, 1))[:n]
    if kind == "sparse":
        s = np.where((np.arange(n) % 7 == 0)[:, None], s, 0).astype(np.uint64)
    elif kind == "tiny":
        s = s & np.array([0xFF, 0, 0, 0], dtype=np.uint64)
    elif kind == "pm1":
        s = np.repeat(np.frombuffer((modulus(curve) - 1).to_bytes(32, "little"), dtype="<u8").astype(np.uint64)[None], n, axis=0)
    elif kind == "equal":
        s = np.repeat(s[:1], n, axis=0)
    return np.ascontiguousarray(s)


def to_int(curve, mont_limbs):
    return O.fr_from_mont_array(curve, np.ascontiguousarray(mont_limbs).reshape(-1, 4))


def case_div(curve):
    n = rng.choice([rng.randint(1, 40), rng.randint(40, 3000), rng.randint(3000, 300000)])
    kind = rng.choice(["uniform", "uniform", "sparse", "tiny", "pm1", "equal"])
    co = O.f_to_mont(curve, 1, rand_fr(curve, n, rng.randint(1, 1 << 30), kind))
    z = O.f_to_mont(curve, 1, rand_fr(curve, 1, rng.randint(1, 1 << 30), rng.choice(["uniform", "tiny", "pm1"])))[0]
    carry = O.f_to_mont(curve, 1, rand_fr(curve, 1, rng.randint(1, 1 << 30), "uniform"))[0] if rng.random() < 0.5 else None
    # oracle: the scan with a carry c is the quotient of (coeffs || c) by (x - z) with the remainder on top: witness_poly of the
    # extended polynomial gives out[1..n], and out[0] = p(z) of it
    ext = np.concatenate([co, carry[None] if carry is not None else np.zeros((1, 4), dtype=np.uint64)])
    want = np.concatenate([O.poly_eval(curve, ext, z)[None], O.witness_poly(curve, ext, z)])[:n]
    if rng.random() < 0.5:
        got = ctx.div_scan(curve, co, z, carry_in=carry)
    else:
        dev = torch.from_numpy(co.view(np.int64)).cuda()
        out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.div_scan(curve, dev.data_ptr(), z, carry_in=carry, out=out.data_ptr(), n=n)
        got = out.cpu().numpy().view(np.uint64)
    ok = bool((got == want).all())
    if ok and n >= 2:       # the same scan as two chained pieces (what a shard boundary does)
        cut = rng.randint(1, n - 1)
        hi = ctx.div_scan(curve, np.ascontiguousarray(co[cut:]), z, carry_in=carry)
        lo = ctx.div_scan(curve, np.ascontiguousarray(co[:cut]), z, carry_in=hi[0])
        ok = bool((np.concatenate([lo, hi]) == want).all())
    return ok, f"n={n} {kind} carry={carry is not None}"


def case_ntt(curve):
    log_n = rng.randint(0, 14)
    N = 1 << log_n
    rows = rng.choice([1, 2, 3, 8, 9, rng.randint(1, 40)]) if log_n <= 11 else rng.randint(1, 4)
    in_cols = rng.choice([N, max(1, N // 2), max(1, N // 4), rng.randint(1, N), rng.randint(1, max(1, N // 4))])
    kind = rng.choice(["uniform", "uniform", "sparse", "pm1"])
    mat = O.f_to_mont(curve, 1, rand_fr(curve, rows * in_cols, rng.randint(1, 1 << 30), kind)).reshape(rows, in_cols, 4)
    got = ctx.ntt_batch(curve, mat, log_n)
    ok = bool((got == O.ntt_batch(curve, mat, log_n)).all())
    if ok:      # test_reed_solomon's property at a few random (row, j)
        p = modulus(curve)
        w = to_int(curve, O.root_of_unity(curve, log_n))[0] if log_n else 1
        for _ in range(3):
            r, j = rng.randrange(rows), rng.randrange(N)
            zj = np.frombuffer(pow(w, j, p).to_bytes(32, "little"), dtype="<u8").astype(np.uint64)
            want = O.poly_eval(curve, np.ascontiguousarray(mat[r]), O.f_to_mont(curve, 1, zj[None])[0])
            ok = ok and bool((got[r, j] == want).all())
    return ok, f"log_n={log_n} rows={rows} in_cols={in_cols} {kind}"


def case_fold(curve):
    p = modulus(curve)
    m = 1 << rng.randint(0, 13)
    fold = rng.random() < 0.7
    size = 2 * m if fold else m
    kind = rng.choice(["uniform", "uniform", "sparse", "pm1"])
    c = O.f_to_mont(curve, 1, rand_fr(curve, size, rng.randint(1, 1 << 30), kind))
    z = O.f_to_mont(curve, 1, rand_fr(curve, size, rng.randint(1, 1 << 30), "uniform"))
    ci, zi = to_int(curve, c), to_int(curve, z)
    cd, zd = torch.from_numpy(c.view(np.int64)).cuda(), torch.from_numpy(z.view(np.int64)).cuda()
    u = ui = None
    if fold:
        uv = rng.randrange(1, p)
        u = O.fr_mont_array(curve, [uv])[0]
        ui = O.fr_mont_array(curve, [pow(uv, -1, p)])[0]
        ci = [(ci[i] + pow(uv, -1, p) * ci[m + i]) % p for i in range(m)]
        zi = [(zi[i] + uv * zi[m + i]) % p for i in range(m)]
    dots = ctx.ipa_fold_dots(curve, cd.data_ptr(), zd.data_ptr(), m, u, ui)
    h = m // 2
    want_l = sum(ci[h + i] * zi[i] for i in range(h)) % p
    want_r = sum(ci[i] * zi[h + i] for i in range(h)) % p
    got = to_int(curve, dots)
    ok = got == [want_l, want_r]
    if fold:
        ok = ok and to_int(curve, cd.cpu().numpy().view(np.uint64)[:m]) == ci and to_int(curve, zd.cpu().numpy().view(np.uint64)[:m]) == zi
    return ok, f"m={m} fold={fold} {kind}"


CASES = {"div": case_div, "ntt": case_ntt, "fold": case_fold}
t0, n_cases, bad = time.time(), {k: 0 for k in kinds}, 0
while time.time() - t0 < budget:
    k = rng.choice(kinds)
    curve = rng.choice(CURVES)
    ok, what = CASES[k](curve)
    n_cases[k] += 1
    if not ok:
        bad += 1
        print("MISMATCH", k, curve, what, flush=True)
print(f"field_fuzz: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
