"""Time HyraxPC commit / open / check through tests/harness/hyrax.py on one GPU (bn254, 2^n evaluations).
Prints one JSON line per size.  Inputs are generated with the oracle's generators; nothing is checked here
(tests/test_hyrax_gpu.py does that)."""
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
from harness import hyrax  # noqa: E402
import oracle_lib as O  # noqa: E402  (input generation only)


def main():
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    ctx = pc.Context(0)
    for n_vars in (16, 20, 22):
        dim = 1 << (n_vars // 2)
        pts = O.gen_bases(curve, dim + 1)
        base = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4D0, 1 << 16))
        evals = torch.from_numpy(np.resize(base, (1 << n_vars, 4)).view(np.int64)).cuda()
        rands = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4D1, dim))
        point = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4D2, n_vars))
        rnd = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4D3, dim + 3))
        c = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x4D4, 1))[0]
        key = hyrax.HyraxKey(ctx, curve, pts[:dim], pts[dim])
        row_coms, state = hyrax.commit(key, evals, rands)        # first call builds the small window table of the key
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            row_coms, state = hyrax.commit(key, evals, rands)
        commit_ms = (time.perf_counter() - t0) / reps * 1e3
        hyrax.open(key, state, point, rnd[0], rnd[1:1 + dim], rnd[1 + dim], rnd[2 + dim], c)      # first call allocates the MSM pipelines of the key
        t0 = time.perf_counter()
        for _ in range(reps):
            proof, ev = hyrax.open(key, state, point, rnd[0], rnd[1:1 + dim], rnd[1 + dim], rnd[2 + dim], c)
        open_ms = (time.perf_counter() - t0) / reps * 1e3
        hyrax.check(key, row_coms, point, proof, c)
        t0 = time.perf_counter()
        for _ in range(reps):
            ok = hyrax.check(key, row_coms, point, proof, c)
        check_ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"workload": f"HyraxPC over {curve}: {n_vars} variables = {dim} row commitments of {dim}+1 pairs",
                          "commit_ms": commit_ms, "commit_pairs_per_s": dim * (dim + 1) / (commit_ms * 1e-3),
                          "open_ms": open_ms, "check_ms": check_ms, "check_ok": bool(ok)}))
        key.close()


if __name__ == "__main__":
    main()
