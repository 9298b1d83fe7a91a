"""Time MarlinKZG10::open's linear combination (marlin_pc/mod.rs:281-287) on the device at
BASELINE configs[2] size: 64 BN254 polynomials of 2^20+1 coefficients, device-resident, followed
by the witness division and the opening MSM it feeds.  Prints one JSON line."""
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
    curve, k, n = "bn254", 64, (1 << 20) + 1
    ctx = pc.Context(0)
    base = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0100, n))
    polys = [torch.from_numpy(np.roll(base, j, axis=0).view(np.int64)).cuda() for j in range(k)]
    xi = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED01FF, k))
    out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ptrs = [p.data_ptr() for p in polys]
    lens = [n] * k
    for _ in range(2):
        ctx.fr_lincomb(curve, ptrs, xi, n_out=n, out=out.data_ptr(), lens=lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.fr_lincomb(curve, ptrs, xi, n_out=n, out=out.data_ptr(), lens=lens)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    byts = k * n * 32 + n * 32
    print(json.dumps({"workload": f"fr_lincomb {k} x (2^20+1) BN254 Fr, device-resident", "ms": ms,
                      "GBps": byts / ms / 1e6, "hbm_frac": byts / ms / 1e6 / 8000.0}))


if __name__ == "__main__":
    main()
