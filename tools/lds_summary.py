"""LDS bank-conflict summary of one rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES pass:
    python tools/lds_summary.py <counter_collection.csv> <out.json>
per kernel: LDS-array cycles, conflict cycles, their ratio, LDS instructions (sums over the launches of the run)."""
import csv, json, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*$", "", r["Kernel_Name"]).strip()
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -- python bench.py --workload ntt (tools/gpu_full_run.sh)", "kernels": {}}
for k, c in acc.items():
    if not c.get("SQ_LDS_IDX_ACTIVE"):
        continue
    out["kernels"][k] = {"launches": len(launches[k]), "SQ_LDS_IDX_ACTIVE": c["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": c.get("SQ_LDS_BANK_CONFLICT"),
                         "conflict_share_of_lds_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS")}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1200])
