"""Loops the small-chunk MSM tests of tests/test_msm_gpu.py in ONE process without pytest's capture (a native abort then shows its own
message on stderr).  python tools/crash_hunt.py [seconds]"""
import os, sys, time, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
faulthandler.enable()
import poly_commit_amd as pc
import test_msm_gpu as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = pc.Context(0)
t0, it = time.time(), 0
while time.time() - t0 < budget:
    for c, t in [(4, 1), (7, 3), (10, 16), (13, 64), (16, 0)]:
        T.test_msm_tunings.__wrapped__(ctx, c, t) if hasattr(T.test_msm_tunings, "__wrapped__") else T.test_msm_tunings(ctx, c, t)
    for n, table in [(700, False), (5000, True), (40000, False), (70001, True)]:
        for t in (1, 2, 5, 8):
            sys.stderr.write("it %d n %d table %s T %d\n" % (it, n, table, t)); sys.stderr.flush()
            T.test_msm_chains_of_cut_buckets(ctx, t, n, table)
    it += 1
print("crash_hunt: %d iterations, no abort" % it)
