#!/usr/bin/env python3
"""Where the window-table build of a 2^24 BLS12-381 key spends its time (the round-5 review: 0.58 s on some boxes, 1.59 s on others,
outside the timed region but inside every real `trim`).  Times, on one box: a bare device allocation of the table's size (first and
second time), the build itself (first, and again after dropping the table), and the free.

usage: python tools/table_build_probe.py [log2 n]      -> one JSON line"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402
import poly_commit_amd as pc  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = (1 << lg) + 1
    ctx = pc.Context(0)
    curve = "bls12_381"
    g = O.gen_bases(curve, 1)[0]
    t0 = time.perf_counter()
    pts = bench.true_srs_points(ctx, curve, g, bench.seed_fr(curve, 0xBE7A24), -1, n)
    torch.cuda.synchronize()
    out = {"log_n": lg, "srs_gen_ms": (time.perf_counter() - t0) * 1e3}
    srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)
    del pts
    torch.cuda.empty_cache()
    table_bytes = 12 * n * 128
    out["table_bytes"] = table_bytes
    for tag in () if os.environ.get("PC_PROBE_SKIP_ALLOCS") else ("alloc_1", "alloc_2"):
        t0 = time.perf_counter()
        p = ctx.malloc(table_bytes)
        t1 = time.perf_counter()
        ctx.free_dev(p)
        t2 = time.perf_counter()
        out[tag + "_ms"] = (t1 - t0) * 1e3
        out[tag.replace("alloc", "free") + "_ms"] = (t2 - t1) * 1e3
    if os.environ.get("PC_PROBE_FIRST_TOUCH"):
        # touch as much fresh device memory as the table will take, through torch's allocator, and give it back to the driver: if the
        # first build's extra seconds are the driver preparing never-used VRAM, they show up here instead
        t0 = time.perf_counter()
        x = torch.empty(table_bytes, dtype=torch.uint8, device="cuda")
        x.zero_()
        torch.cuda.synchronize()
        out["first_touch_zero_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        x.zero_()
        torch.cuda.synchronize()
        out["second_zero_ms"] = (time.perf_counter() - t0) * 1e3
        del x
        torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info()
    out["free_gb_before_build"] = free_b / 1e9
    for tag in ("build_1", "build_2", "build_3"):
        t0 = time.perf_counter()
        srs.precompute()
        out[tag + "_ms"] = (time.perf_counter() - t0) * 1e3       # (precompute drops the previous table first)
    out["resident"] = srs.bytes_resident()
    t0 = time.perf_counter()
    srs.free()
    out["free_key_ms"] = (time.perf_counter() - t0) * 1e3
    os.write(bench._RESULT_FD, (json.dumps(out) + "\n").encode())          # (importing bench points fd 1 at stderr)


if __name__ == "__main__":
    main()
