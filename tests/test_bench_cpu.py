"""bench.py's op-count helpers against brute-force counts of the kernels' own loop structure (no GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _count_like_kernel(log_n, in_cols):
    """Walks the index space of csrc/ntt.hpp (k_ntt_pass_a / k_ntt_pass_b / lds_ntt_stages) and counts the products it executes."""
    N = 1 << log_n
    lg1 = (log_n + 1) // 2
    lg2 = log_n - lg1
    N1, N2 = 1 << lg1, 1 << lg2
    zskip = 0
    while zskip < lg1 and in_cols <= (N >> (zskip + 1)):
        zskip += 1

    def stages(lines, lg, first):
        prod = bfly = 0
        length, halfs, quarters = 1 << lg, (1 << lg) >> 1, (1 << lg) >> 2
        s = first
        if ((lg - first + 1) & 1) and s <= lg:
            h = 1 << (s - 1)
            for b in range(lines * halfs):
                j = (b & (halfs - 1)) & (h - 1)
                prod += 1 if j else 0
                bfly += 1
            s += 1
        while s + 1 <= lg:
            h = 1 << (s - 1)
            for b in range(lines * quarters):
                j = (b & (quarters - 1)) & (h - 1)
                prod += (3 if j else 0) + 1
                bfly += 4
            s += 2
        return prod, bfly
    pa, ba = stages(N2, lg1, zskip + 1)
    pb, bb = stages(N1, lg2, 1)
    between = sum(1 for i2 in range(N2) for j1 in range(N1) if i2 * j1)
    return pa + pb + between, ba + bb


def test_ntt_products_match_the_kernel_loops():
    import bench
    for log_n in range(0, 13):
        for in_cols in {1 << log_n, max(1, (1 << log_n) // 4), max(1, (1 << log_n) // 2 + 1), 1}:
            p, b, nominal = bench.ntt_products(log_n, in_cols)
            assert (p, b) == _count_like_kernel(log_n, in_cols), (log_n, in_cols)
            assert nominal == (1 << log_n) // 2 * log_n
            assert p <= nominal + (1 << log_n)


def test_naf_weight():
    import bench
    for v, (w, top) in {0: (0, 0), 1: (1, 0), 3: (2, 2), 7: (2, 3), 0b10101: (3, 4), (1 << 130) - 1: (2, 130)}.items():
        assert bench._naf_weight_and_top(v) == (w, top), v


def test_glv_split_reads_the_library_header():
    import random
    import bench
    lam = {"pallas": 0x06819a58283e528e511db4d81cf70f5a0fed467d47c033af2aa9d2e050aa0e4f}
    for curve in ("bls12_381", "bn254", "pallas"):
        gc = bench.glv_header_constants(curve)
        r = bench.fr_modulus(curve)
        rng = random.Random(5)
        for _ in range(200):
            k = rng.randrange(r)
            k1, k2 = bench.glv_split(gc, k)
            assert abs(k1) < 1 << 130 and abs(k2) < 1 << 130
            if curve in lam:
                assert (k1 + k2 * lam[curve] - k) % r == 0
