"""pc_hip_msm with HOST scalars and pc_hip_kzg_open with HOST coefficients run large inputs in parts (the PCIe copy and the sort of a part
under the accumulation of the one before; abi.hip host_split_min, default 2^21 pairs (PC_HIP_HOST_SPLIT_LOG2); host_parts, default 4).  The split is forced down to 2^10 here
(PC_HIP_HOST_SPLIT_LOG2, read once per process: hence the subprocess) and compared with the oracle's kzg_commit / kzg_open
(kzg10/mod.rs:157-210, :287-310) bit for bit; the unsplit paths of the same calls run in the same child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import oracle_lib as O
import poly_commit_amd as pc
ctx = pc.Context(0)
for curve in ("bls12_381", "bn254", "pallas"):
    nmax = 5000
    powers = O.gen_bases(curve, nmax)
    srs = ctx.upload_srs(curve, powers)
    for table in (False, True, "glv"):
        if table:
            srs.precompute(min_pairs=1, glv=(table == "glv"))      # the full window table, then its GLV form (pre-split scalars, two bucket sets)
        for n in (1, 2, 3, 1023, 1024, 1025, 2049, 4097, 5000):
            coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0001 + n, n))
            if n >= 1025:
                coeffs[3] = 0; coeffs[n // 2] = 0          # zero coefficients on both sides of the cut
            z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 7 + n, 1))[0]
            rc, want_c = O.kzg_commit(curve, powers, coeffs)
            rc2, want_w = O.kzg_open(curve, powers, coeffs, z)
            assert rc == 0 and rc2 == 0
            got_c, _ = srs.msm(coeffs, montgomery=True)                          # host scalars: split from 2^10 pairs
            assert (got_c == want_c).all(), ("commit", curve, n, table)
            can = O.f_from_mont(curve, 1, coeffs)
            got_c2, _ = srs.msm(can)                                             # canonical form through the same split
            assert (got_c2 == want_c).all(), ("commit canonical", curve, n, table)
            got_w, inf = srs.kzg_open(coeffs, z)                                 # host coefficients: split from 2^10
            assert (got_w == want_w).all() and inf == (not want_w.any()), ("open host", curve, n, table)
            dev = torch.from_numpy(coeffs.view(np.int64)).cuda()
            got_wd, _ = srs.kzg_open(dev.data_ptr(), z, n=n)                      # device coefficients: never split
            assert (got_wd == want_w).all(), ("open device", curve, n, table)
        # base offsets and min(len) truncation through the split
        n = 3000
        sc = O.gen_scalars(curve, 99, n)
        for off in (1, 1234, nmax - 1500):
            got, _ = srs.msm(sc, base_offset=off)
            k = min(n, nmax - off)
            assert (got == O.msm_pippenger(curve, np.ascontiguousarray(powers[off:off + k]), np.ascontiguousarray(sc[:k]), 8, 2)).all(), (curve, off)
            coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 5, 1500))
            z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 6, 1))[0]
            got, _ = srs.kzg_open(coeffs, z, base_offset=off)
            rc, want = O.kzg_open(curve, np.ascontiguousarray(powers[off:]), coeffs, z)
            assert (got == want).all(), ("open offset", curve, off)
    srs.free()
ctx.close()
print("host-split ok")
'''


@pytest.mark.parametrize("parts", ["4", "0", "2", "7"])
def test_host_inputs_in_parts_against_the_oracle(parts):
    """PC_HIP_HOST_PARTS: 4 (the default) / 2 / 7 = ONE MSM in that many parts on one pipeline (MsmPlan::begin_parts: copy and sort of part
    k + 1 beside the accumulation of part k, second bucket array merged into the first, one reduction); 0 = round 4's two half-size MSMs
    on two pipelines."""
    env = dict(os.environ, PC_HIP_HOST_SPLIT_LOG2="10", PC_HIP_HOST_PARTS=parts)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "host-split ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_kzg_open_rejects_a_quotient_longer_than_the_key(ctx):
    import numpy as np
    import oracle_lib as O
    curve = "bn254"
    srs = ctx.upload_srs(curve, O.gen_bases(curve, 16))
    coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 3, 18))
    z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 4, 1))[0]
    with pytest.raises(Exception):
        srs.kzg_open(coeffs, z)                 # 17 quotient coefficients, 16 powers
    got, _ = srs.kzg_open(np.ascontiguousarray(coeffs[:17]), z)
    rc, want = O.kzg_open(curve, O.gen_bases(curve, 16), np.ascontiguousarray(coeffs[:17]), z)
    assert (got == want).all()
    srs.free()
