"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) merged into profiles/rNN_pmc_traffic.json under one key per workload.

usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <key>
       key e.g. "bls12_381:2^24:table" (what bench.py looks up for roofline.traffic) or "ntt:bls12_381:2^24"

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-byte request,
so reads are doubled (the guide's correction); WRITE_SIZE is taken as is."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    """kernel -> {launches, avg_per_launch_KiB} over the launches of the kernel's LARGEST launch geometry only.

    A profiled process also launches the same kernel on other sizes (the row slabs of the host-to-host Ligero calls, the parity
    checks' single rows, half-size parts of a host MSM): averaging those in made round 5's NTT figure 13x too small.  Launches are
    classed by (Grid_Size, Workgroup_Size); the class with the largest grid is the workload's own launch."""
    acc = defaultdict(lambda: defaultdict(float))      # (kernel, geometry) -> dispatch id -> value (summed over XCDs/SEs)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).strip()
            geom = (int(float(r.get("Grid_Size") or 0)), int(float(r.get("Workgroup_Size") or 0)))
            acc[(name, geom)][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    for (name, geom), v in acc.items():
        if name in out and out[name]["grid_size"] >= geom[0]:
            continue
        out[name] = {"launches": len(v), "avg_per_launch_KiB": sum(v.values()) / len(v), "grid_size": geom[0], "workgroup_size": geom[1]}
    for name in out:
        out[name]["launches_all_geometries"] = sum(len(v) for (n, _), v in acc.items() if n == name)
    return out


def main():
    fetch, write, out, key = sys.argv[1:5]
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc.setdefault("note", "FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per "
                           "128-B request, so reads are doubled; WRITE_SIZE taken as is.  The bucket-accumulation gather is 6 x 16 B per "
                           "lane at random addresses (not the wide coalesced stream the 2x was calibrated on): treat as an upper estimate.  "
                           "Source: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/gpu_full_run.sh).")
    fs, ws = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    hbm = {k: (2 * fs[k]["avg_per_launch_KiB"] + ws.get(k, {"avg_per_launch_KiB": 0})["avg_per_launch_KiB"]) * 1024 for k in fs}
    doc.setdefault("per_kernel_hbm_bytes_per_launch", {})[key] = {k: v for k, v in hbm.items() if v > 1e6}
    doc.setdefault("per_kernel_raw_KiB", {})[key] = {"FETCH_SIZE": fs, "WRITE_SIZE": ws}
    acc = [k for k in fs if "k_accumulate" in k]
    if acc:
        doc.setdefault("accumulate_hbm_bytes_per_launch", {})[key] = hbm[acc[0]]
        doc.setdefault("accumulate_fetch_raw_plus_write_bytes_per_launch", {})[key] = (
            fs[acc[0]]["avg_per_launch_KiB"] + ws.get(acc[0], {"avg_per_launch_KiB": 0})["avg_per_launch_KiB"]) * 1024
        print(key, acc[0], hbm[acc[0]] / 1e9, "GB/launch")
    ntt = [k for k in fs if "k_ntt_pass" in k]
    if ntt:
        doc.setdefault("ntt_hbm_bytes_per_batch", {})[key] = sum(hbm[k] for k in ntt)
        print(key, "ntt passes", sum(hbm[k] for k in ntt) / 1e9, "GB/batch")
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
