"""pc_hip_ligero_commit host -> host (BASELINE configs[4]: 512 x 2^15 -> 512 x 2^17 over BLS12-381 Fr) against the encoded bytes per
slab and the number of helper threads (PC_HIP_LIGERO_SLAB_MB, PC_HIP_LIGERO_HELPERS, both read per call; slab 0 = the whole-matrix
path): ms per blocking call, results compared with the whole-matrix path's -- in a quiet process, where the runtime finds the pages of
the same host buffers pinned from the call before, and (--cold) after a key with its tables has been built and freed, from when on
it pins them anew in every call (the state a long-running prover is in).  PC_HIP_LIGERO_TRACE=1 adds where the threads spent each call.
Run on the GPU box:  python tools/ligero_stream_probe.py [--cold] [helpers:slab_mb ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
import poly_commit_amd as pc  # noqa: E402

curve, rows, in_cols, log_n = "bls12_381", 512, 1 << 15, 17
flags = [a for a in sys.argv[1:] if a.startswith("--")]
cases = [a for a in sys.argv[1:] if not a.startswith("--")] or ["3:0", "1:32", "2:32", "3:16", "3:32", "3:64", "3:128", "4:32"]
ctx = pc.Context(0)
mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x11, rows * in_cols)).reshape(rows, in_cols, 4)
n = 1 << log_n
ext0 = np.empty((rows, n, 4), dtype=np.uint64)
os.environ["PC_HIP_LIGERO_SLAB_MB"] = "0"
nodes0, leaves0 = ctx.ligero_commit(curve, mat, log_n, ext_out=ext0)
ext = np.empty((rows, n, 4), dtype=np.uint64)


def sweep(state):
    for case in cases:
        os.environ["PC_HIP_LIGERO_HELPERS"], os.environ["PC_HIP_LIGERO_SLAB_MB"] = case.split(":")
        ext[:] = 0
        nodes, leaves = ctx.ligero_commit(curve, mat, log_n, ext_out=ext)
        same = bool((nodes == nodes0).all() and (leaves == leaves0).all() and (ext == ext0).all())
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            ctx.ligero_commit(curve, mat, log_n, ext_out=ext)
            ts.append(round((time.perf_counter() - t0) * 1e3, 1))
        print(f"{state}: helpers {case.split(':')[0]}, slab {case.split(':')[1]:>3} MB: median {sorted(ts)[3]:5.1f} ms  calls {ts}  equal to the whole-matrix path: {same}", flush=True)


sweep("quiet")
if "--cold" in flags:
    import bench as B
    c2, m = "pallas", 1 << 22
    pts = B.true_srs_points(ctx, c2, O.gen_bases(c2, 1)[0], B.seed_fr(c2, 0xA11CE5), 0, m + 1)
    srs = ctx.upload_srs(c2, pts.data_ptr(), n=m)
    del pts
    srs.precompute()
    srs.precompute_fold()
    srs.free()
    sweep("after a key was freed")
ctx.close()
