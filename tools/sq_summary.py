"""VALU-busy summary from a rocprofv3 SQ counter pass -> profiles/rNN_valu.json.

usage: python tools/sq_summary.py [label=]<counter_collection.csv> [[label=]<counter_collection.csv> ...] <out.json>
       (one "kernels" table per input, keyed by its label -- default: the csv's directory name)

Pass: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
      SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES   (8 SQ slots: one pass, no trace domains beside --kernel-trace).
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md).  Reported:
  valu_share_of_wave_cycles   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (how much of a resident wave's life is VALU issue)
  issue_stall_share           SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      (dependency / pipe stalls)
  wait_share                  SQ_WAIT_ANY / SQ_WAVE_CYCLES           (s_waitcnt on memory, barriers)
  valu_busy                   SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel duration * f): the gfx94x VALUBusy formula with
                              the duration from the kernel's own timestamps and f = 2.4 GHz (the nominal clock; under
                              sustained VALU load the part clocks lower, so this UNDERSTATES the busy fraction)
"""
import csv
import json
import re
import sys
from collections import defaultdict

SIMDS, CLOCK = 256 * 4, 2.4e9


def main():
    *paths, out = sys.argv[1:]
    doc = {"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY "
                     "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -- python bench.py ... --inflight 0 (see tools/gpu_full_run.sh)",
           "units": "SQ_* cycle counters are quad-cycles summed over waves; per-launch averages", "workloads": {}}
    import os
    for spec in paths:
        label, _, path = spec.rpartition("=")
        label = label or os.path.basename(os.path.dirname(os.path.abspath(path)))
        doc["workloads"][label] = {}
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
        dur = defaultdict(dict)
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).strip()
            acc[name][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            dur[name][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, v in acc.items():
            avg = {c: sum(d.values()) / len(d) for c, d in v.items()}
            ns = sum(dur[k].values()) / len(dur[k])
            if ns < 50e3 or not avg.get("SQ_WAVE_CYCLES"):
                continue
            wc = avg["SQ_WAVE_CYCLES"]
            doc["workloads"][label][k] = {
                "launches": len(dur[k]), "avg_duration_ms_under_pmc": ns / 1e6, "waves": avg.get("SQ_WAVES"),
                "SQ_INSTS_VALU": avg.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": avg.get("SQ_ACTIVE_INST_VALU"),
                "SQ_WAVE_CYCLES": wc, "SQ_WAIT_INST_ANY": avg.get("SQ_WAIT_INST_ANY"), "SQ_WAIT_ANY": avg.get("SQ_WAIT_ANY"),
                "valu_share_of_wave_cycles": avg.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                "issue_stall_share": avg.get("SQ_WAIT_INST_ANY", 0) / wc, "wait_share": avg.get("SQ_WAIT_ANY", 0) / wc,
                "valu_busy": avg.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (SIMDS * ns * 1e-9 * CLOCK),
                "valu_insts_per_wave": avg.get("SQ_INSTS_VALU", 0) / max(avg.get("SQ_WAVES", 1), 1)}
    json.dump(doc, open(out, "w"), indent=1)
    for label, ks in doc["workloads"].items():
      for k, v in ks.items():
        print(f"{label:12s} {k[:60]:60s} {v['avg_duration_ms_under_pmc']:8.3f} ms  VALUBusy {v['valu_busy']:.2f}  valu/stall/wait of wave cycles "
              f"{v['valu_share_of_wave_cycles']:.2f}/{v['issue_stall_share']:.2f}/{v['wait_share']:.2f}")


if __name__ == "__main__":
    main()
