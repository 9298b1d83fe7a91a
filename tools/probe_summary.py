"""One line per result of a tools/gpu_probe.sh call (gpurun_out/<tag>_*): step / blocking / phase times of the bench lines, the NTT
passes, the batch, the IPA opening; the head of every rocprofv3 kernel-stats file; the SQ-counter shares of the heaviest kernels."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "probe"
G = "gpurun_out"


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except Exception:
                pass
    return None


for f in sorted(glob.glob(f"{G}/{tag}_*.out")):
    name = os.path.basename(f)[len(tag) + 1:-4]
    if name.startswith(("micro", "tests")):
        print("==", name); print("".join(open(f).readlines()[-(40 if name.startswith("micro") else 6):])); continue
    d = last_json(f)
    if not d:
        print("==", name, "no JSON"); continue
    r = lambda v: round(v, 2) if isinstance(v, (int, float)) else v      # noqa: E731
    if "ntt_phase_ms" in d:
        print("==", name, "ms/step", r(d["ms_per_step"]), {k: r(v) for k, v in d["ntt_phase_ms"].items()}, "arith", r((d["roofline"].get("arithmetic") or {}).get("frac")), d["parity"]["horner_spot_checks_ok"], d["parity"]["one_row_vs_oracle_ntt_ok"])
    elif "ms_per_commitment" in d:
        rf = d.get("roofline") or {}
        print("==", name, "ms/step", r(d["ms_per_step"]), "acc/pass", r(rf.get("kernel_ms")), "frac", r((rf.get("arithmetic") or {}).get("frac")), {k: r(v) for k, v in (rf.get("pass_phase_ms_sum") or {}).items()}, d["parity"]["all_commitments_closed_form_ok"])
    elif "open_rounds_ms" in d:
        print("==", name, "commit", r(d["commit_ms"]), "open", r(d["open_rounds_ms"]), d["open_breakdown_ms"], d["per_round_ms"][:6])
    elif "blocking_msm_ms" in d:
        print("==", name, "step", r(d["ms_per_step"]), "blocking", r(d["blocking_msm_ms"]), {k: r(v) for k, v in d["msm_phase_ms"].items()}, d["parity"]["commit_ok"], d["parity"]["open_ok"])
    else:
        print("==", name, json.dumps(d)[:900])
for f in sorted(glob.glob(f"{G}/{tag}_prof_*/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i > 12:
            break
        print("  ", row[0][:90], row[1:4])
for f in sorted(glob.glob(f"{G}/{tag}_sq_*/**/*counter_collection.csv", recursive=True)):
    agg = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:70]
        agg.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    print("==", f)
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        print("  ", k, "valu_share", round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "issue_stall_share", round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3), "wait_share", round(c.get("SQ_WAIT_ANY", 0) / wc, 3))
for f in sorted(glob.glob(f"{G}/{tag}_lds_*/**/*counter_collection.csv", recursive=True)):
    agg = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:70]
        agg.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    print("==", f)
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:4]:
        act = c.get("SQ_LDS_IDX_ACTIVE", 0) or 1
        print("  ", k, "lds_active", c.get("SQ_LDS_IDX_ACTIVE"), "bank_conflict", c.get("SQ_LDS_BANK_CONFLICT"), "conflict_share", round(c.get("SQ_LDS_BANK_CONFLICT", 0) / act, 3), "insts_lds", c.get("SQ_INSTS_LDS"))
