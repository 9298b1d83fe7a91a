"""pc_hip_ligero_commit host -> host (BASELINE configs[4]: 512 x 2^15 -> 512 x 2^17 over BLS12-381 Fr) against the encoded bytes per
slab (PC_HIP_LIGERO_SLAB_MB, read per call; 0 = the whole-matrix path): ms per blocking call, and the results compared with the
whole-matrix path's.  Run on the GPU box:  python tools/ligero_stream_probe.py [slab_mb ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
import poly_commit_amd as pc  # noqa: E402

curve, rows, in_cols, log_n = "bls12_381", 512, 1 << 15, 17
slabs = [float(a) for a in sys.argv[1:]] or [0, 16, 32, 64, 128, 256]
ctx = pc.Context(0)
mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x11, rows * in_cols)).reshape(rows, in_cols, 4)
n = 1 << log_n
ext0 = np.empty((rows, n, 4), dtype=np.uint64)
os.environ["PC_HIP_LIGERO_SLAB_MB"] = "0"
nodes0, leaves0 = ctx.ligero_commit(curve, mat, log_n, ext_out=ext0)
ext = np.empty((rows, n, 4), dtype=np.uint64)
for mb in slabs:
    os.environ["PC_HIP_LIGERO_SLAB_MB"] = repr(mb)
    ext[:] = 0
    nodes, leaves = ctx.ligero_commit(curve, mat, log_n, ext_out=ext)
    same = bool((nodes == nodes0).all() and (leaves == leaves0).all() and (ext == ext0).all())
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.ligero_commit(curve, mat, log_n, ext_out=ext)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"slab {mb:6.0f} MB: best {min(ts):6.2f} ms  median {sorted(ts)[2]:6.2f} ms  equal to the whole-matrix path: {same}", flush=True)
ctx.close()
