"""PCIe-inclusive MSM rate: scalars handed over as HOST buffers (what a Rust shim passes)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import oracle_lib as O
import poly_commit_amd as pc
curve, n = "bls12_381", (1 << 20) + 1
ctx = pc.Context(0)
bases = O.gen_bases(curve, n)
srs = ctx.upload_srs(curve, bases)
coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 1, n))
import torch
cdev = torch.from_numpy(coeffs.view(np.int64).copy()).cuda()
pinned = torch.from_numpy(coeffs.view(np.int64).copy()).pin_memory()
torch.cuda.synchronize()
def t(fn, k=10):
    fn(); s = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - s) / k * 1e3
r = {"device_resident_ms": t(lambda: srs.msm(cdev.data_ptr(), n=n, montgomery=True)),
     "host_pageable_ms": t(lambda: srs.msm(coeffs, montgomery=True)),
     "host_pinned_ms": t(lambda: srs.msm(pinned, n=n, montgomery=True))}
r["pairs_per_s_device"] = n / r["device_resident_ms"] * 1e3
r["pairs_per_s_host_pageable"] = n / r["host_pageable_ms"] * 1e3
r["pairs_per_s_host_pinned"] = n / r["host_pinned_ms"] * 1e3
print(json.dumps(r))
