"""Time pc_hip_msm_many on Hyrax's shape (sqrt(n) row commitments of sqrt(n) pairs, hyrax/mod.rs:233-242)
against the same rows issued one by one through pc_hip_msm_batch.  Prints one JSON line per shape."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import poly_commit_amd as pc  # noqa: E402
import oracle_lib as O  # noqa: E402  (input generation only)


def main():
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    ctx = pc.Context(0)
    for lg in (16, 20, 22):
        m = B = 1 << (lg // 2)
        bases = O.gen_bases(curve, m)
        srs = ctx.upload_srs(curve, bases)
        base = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4A13, 1 << 16))
        sc = torch.from_numpy(np.resize(base, (B * m, 4)).view(np.int64)).cuda()
        srs.msm_many(sc.data_ptr(), m=m, n_msms=B, montgomery=True)          # builds the small table
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            srs.msm_many(sc.data_ptr(), m=m, n_msms=B, montgomery=True)
        many_ms = (time.perf_counter() - t0) / reps * 1e3
        rows = min(B, 256)
        ptrs = [sc.data_ptr() + 32 * m * k for k in range(rows)]
        srs.msm_batch(ptrs, [m] * rows)
        t0 = time.perf_counter()
        srs.msm_batch(ptrs, [m] * rows)
        one_by_one_ms = (time.perf_counter() - t0) * 1e3 * (B / rows)
        print(json.dumps({"workload": f"{B} MSMs of {m} pairs ({curve}), Hyrax commit of 2^{lg} evaluations",
                          "msm_many_ms": many_ms, "pairs_per_s": B * m / many_ms * 1e3,
                          "one_by_one_ms_extrapolated": one_by_one_ms, "speedup": one_by_one_ms / many_ms}))
        srs.free()


if __name__ == "__main__":
    main()
