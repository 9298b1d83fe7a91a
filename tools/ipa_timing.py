"""Time the IPA halving rounds (BASELINE configs[3] shape) on one GPU: Pallas, n = 2^log_n."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc
from poly_commit_amd import ipa

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fkb = (1 << int(sys.argv[2])) if len(sys.argv) > 2 else None        # log2 of the size from which the key stays fixed (default: ipa.FIXED_KEY_BELOW)
curve = "pallas"
n = 1 << log_n
ctx = pc.Context(0)
if os.environ.get("PC_IPA_C"):          # window width of the table-free MSMs (rounds 3+), experiment switch
    ctx.set_msm_tuning(int(os.environ["PC_IPA_C"]), 0)
key = O.gen_bases(curve, n + 1)
coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE, n))
point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B, 1))[0]
ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1, log_n))
srs = ctx.upload_srs(curve, np.ascontiguousarray(key[:n]))
t_tables = time.perf_counter()
if os.environ.get("PC_IPA_TABLES", "1") != "0":        # once per committer key: window table + fold table
    srs.precompute(); srs.precompute_fold()
t_tables = time.perf_counter() - t_tables
cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
torch.cuda.synchronize()
for _ in range(3):      # every pipeline of the SRS exists (streams + workspace are created on first use)
    srs.msm(cdev.data_ptr(), n=n, montgomery=True)
t = time.perf_counter(); srs.msm(cdev.data_ptr(), n=n, montgomery=True); t_commit = time.perf_counter() - t
# several openings on the same resident key: the first pays once-per-process allocations (the half-size working key and its
# pipelines, kept by the committer key afterwards); the best of the rest is what a prover sees per opening
runs = []
for rep in range(int(os.environ.get("PC_IPA_REPS", "3"))):
    it = iter(range(log_n))
    work = cdev.clone()
    torch.cuda.synchronize()
    tm = {}
    t = time.perf_counter()
    ipa.ipa_open_rounds(ctx, curve, srs, work, n, point, key[n], lambda L, R_: ch[next(it)], timings=tm, fixed_key_below=fkb)
    t_open = time.perf_counter() - t
    per_round = tm.pop("per_round_ms", [])
    tm.pop("ec_fold_per_round_ms", None)
    kinds = tm.pop("ec_fold_kind", None)
    runs.append((t_open, tm, per_round))
first = runs[0][0]
t_open, tm, per_round = min(runs, key=lambda r: r[0])
print(json.dumps({"workload": f"InnerProductArgPC over Pallas, n = 2^{log_n}: commit MSM + {log_n} halving rounds (challenges supplied)",
                  "commit_ms": t_commit * 1e3, "open_rounds_ms": t_open * 1e3, "first_open_rounds_ms": first * 1e3, "openings": len(runs),
                  "commit_pairs_per_s": n / t_commit, "open_msm_pairs_per_s": 2 * n / t_open, "fixed_key_below": fkb or (ipa.FIXED_KEY_BELOW if os.environ.get("PC_IPA_PY_LOOP") == "1" else ipa.library_fixed_key_below()),
                  "key_tables": os.environ.get("PC_IPA_TABLES", "1") != "0", "fold_table": list(srs.fold_table_info()) + [srs.bytes_resident()["fold_table"]],
                  "key_tables_build_ms": t_tables * 1e3,
                  "open_breakdown_ms": {k: round(v, 1) for k, v in tm.items()}, "per_round_ms": per_round}))
