#!/usr/bin/env python3
"""Timeline of the kernels inside a time window of a rocprofv3 --kernel-trace CSV: what ran when, on which queue, with the idle gaps --
for latency-bound phases (the fixed-key rounds of an IPA opening) where a per-kernel average says nothing.

usage: python tools/trace_window.py <kernel_trace.csv> [--tail-ms 3.0] [--before KERNEL_SUBSTRING] [--n 80]
  default window: the last --tail-ms milliseconds before the LAST launch whose name contains --before (default: the end of the trace)"""
import csv
import sys


def main():
    path = sys.argv[1]
    tail_ms, before, limit = 3.0, None, 80
    a = sys.argv[2:]
    while a:
        if a[0] == "--tail-ms":
            tail_ms = float(a[1]); a = a[2:]
        elif a[0] == "--before":
            before = a[1]; a = a[2:]
        elif a[0] == "--n":
            limit = int(a[1]); a = a[2:]
        else:
            a = a[1:]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?"),
                         r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
    rows.sort()
    end = rows[-1][1]
    if before:
        hits = [r for r in rows if before in r[2]]
        if hits:
            end = hits[-1][1]
    t0 = end - int(tail_ms * 1e6)
    win = [r for r in rows if r[1] >= t0 and r[0] <= end]
    busy, last_end = 0, t0
    print(f"{len(win)} launches in the last {tail_ms} ms before {before or 'the end'}")
    for s, e, name, q, grid, wg in win[:limit]:
        gap = (s - last_end) / 1e3
        print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  q{q:>3}  grid {grid:>9}/{wg:<4} {name}")
        last_end = max(last_end, e)
    # union of busy time
    iv = sorted((max(s, t0), min(e, end)) for s, e, *_ in win)
    cur_s, cur_e, tot = None, None, 0
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    print(f"device busy (union of kernel intervals): {tot / 1e3:.1f} us of {tail_ms * 1e3:.0f} us = {tot / (tail_ms * 1e6):.2f}")


if __name__ == "__main__":
    main()
