"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) -> profiles/rNN_pmc_traffic.json.

usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [old.json]

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-byte request,
so reads are doubled (the guide's correction); WRITE_SIZE is taken as is.  Sections of `old.json`
that these two passes do not produce (SQ counters of an earlier pass) are carried over."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: defaultdict(float))      # kernel -> dispatch id -> value (summed over XCDs/SEs)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).strip()
            acc[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {"launches": len(v), "avg_per_launch_KiB": sum(v.values()) / len(v)} for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    old = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
    fs, ws = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    acc = next(k for k in fs if "k_accumulate" in k)
    doc = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on: python bench.py "
                  "--steps 2 --warmup 1 --no-cpu-baseline --inflight 0 (KZG commit+open, BLS12-381, 2^20, default = SRS window table)",
        "note": "FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per "
                "128-B request, so reads are doubled; WRITE_SIZE taken as is. The bucket-accumulation gather is 6 x 16 B per "
                "lane at random addresses (not the wide coalesced stream the 2x was calibrated on): treat as an estimate.",
        "accumulate_fetch_KiB_per_launch_raw": fs[acc]["avg_per_launch_KiB"],
        "accumulate_write_KiB_per_launch": ws[acc]["avg_per_launch_KiB"],
        "accumulate_hbm_bytes_per_launch": (2 * fs[acc]["avg_per_launch_KiB"] + ws[acc]["avg_per_launch_KiB"]) * 1024,
        "per_kernel_hbm": {"FETCH_SIZE": fs, "WRITE_SIZE": ws},
    }
    for k in ("accumulate_valu", "accumulate_valu_note", "per_kernel_sq"):
        if k in old:
            doc[k + "_table_free_run"] = old[k]
    json.dump(doc, open(out, "w"), indent=1)
    print(acc, doc["accumulate_hbm_bytes_per_launch"] / 1e9, "GB/launch")


if __name__ == "__main__":
    main()
