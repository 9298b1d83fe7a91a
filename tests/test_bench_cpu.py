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


# ---- the line the driver parses (round 5's 25 KB line came back with parsed = null) ----------------------------------------------
def _recorded_lines():
    import glob
    import json
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[456]_*.json"))):
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        if isinstance(d, dict) and "metric" in d and "ms_per_step" in d:
            yield os.path.basename(f), d


def test_stdout_line_is_at_most_6144_bytes_and_keeps_the_contract():
    """compact_line over every full record under profiles/ (the 25 KB round-5 line among them): <= 6 KB, the contract's keys, a
    `roofline` with bound / achieved / peak / unit / frac / traffic and a `cpu_baseline` with value / unit / cores / kind / sample."""
    import json
    import bench
    seen = 0
    for name, d in _recorded_lines():
        c = bench.compact_line(d)
        line = json.dumps(c)
        assert len(line) <= bench.LINE_MAX_BYTES == 6144, (name, len(line))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            assert c[k] == d[k] or abs(c[k] - d[k]) <= 1e-5 * abs(d[k]), (name, k)
        assert "workload" in c["config"]
        if d.get("roofline"):
            for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
                assert k in c["roofline"], (name, k)
            assert abs(c["roofline"]["frac"] - d["roofline"]["frac"]) <= 1e-5 * d["roofline"]["frac"]
            if "algorithmic_bytes_per_launch" in d["roofline"]:
                assert c["roofline"]["algorithmic_bytes_per_launch"] == int(d["roofline"]["algorithmic_bytes_per_launch"])
        if d.get("cpu_baseline"):
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in c["cpu_baseline"], (name, k)
        if "workloads" in d:
            assert set(c["workloads"]) == set(d["workloads"])
            for n, w in c["workloads"].items():
                assert "parity_ok" in w and "ms" in w, (name, n)
        seen += 1
    assert seen >= 8          # the recorded lines of rounds 4 and 5 are in the tree


def test_stdout_line_sheds_blocks_rather_than_grow():
    import json
    import bench
    fat = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "w" * 500},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.0 / 8000, "traffic": None, "kernel": "k" * 900},
           "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "port", "sample": "s" * 900},
           "parity": {"commit_ok": True, "method": "x" * 4000},
           "workloads": {f"w{i}": {"ms_per_step": 1.0, "parity": {"a_ok": True}, "roofline": {"frac": 0.1}, "junk": ["y" * 90] * 40} for i in range(64)},
           "per_rank_ms_per_step": [1.0] * 8, "msm_phase_ms": {f"p{i}": 0.123456789 for i in range(200)}}
    c = bench.compact_line(fat)
    assert len(json.dumps(c)) <= 6144
    assert c["roofline"]["frac"] == 1.0 / 8000 and c["cpu_baseline"]["kind"] == "port" and c["parity"] == {"commit_ok": True}
    assert c["per_rank_ms_per_step"] == [1.0] * 8


def test_emit_writes_the_detail_file_and_one_short_line(tmp_path, monkeypatch):
    import json
    import bench
    rd, wr = os.pipe()
    monkeypatch.setattr(bench, "_RESULT_FD", wr)
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "detail.json"))
    name, d = next(x for x in _recorded_lines() if x[0] == "r05_bench_n1.json")
    bench.emit(d)
    os.close(wr)
    out = os.read(rd, 1 << 16).decode()
    assert out.count("\n") == 1 and len(out) <= 6145
    assert json.loads(out)["detail"] == str(tmp_path / "detail.json")
    assert json.load(open(tmp_path / "detail.json")) == d


# ---- the counter summaries behind roofline.traffic ----------------------------------------------------------------------------------
def test_pmc_summary_takes_the_launches_of_the_largest_geometry(tmp_path):
    """tools/pmc_summary.py: a kernel's traffic per launch is averaged over the launches of its LARGEST grid only (round 5 averaged the
    320 slab launches of the host-to-host Ligero leg into the whole-batch NTT launch: 0.21 GB instead of 7.2 GB)."""
    import csv
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as P
    f = tmp_path / "c.csv"
    rows = [("1", "16777216", "512", "void pc::k_ntt_pass_a<F>(a, b)", "FETCH_SIZE", "1000"), ("1", "16777216", "512", "void pc::k_ntt_pass_a<F>(a, b)", "FETCH_SIZE", "500"),
            ("2", "16777216", "512", "void pc::k_ntt_pass_a<F>(a, b)", "FETCH_SIZE", "1700")] + \
           [(str(10 + i), "262144", "512", "void pc::k_ntt_pass_a<F>(a, b)", "FETCH_SIZE", "20") for i in range(50)] + \
           [("3", "16777216", "512", "void pc::k_ntt_pass_a<F>(a, b)", "WRITE_SIZE", "9")]
    with open(f, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Dispatch_Id", "Grid_Size", "Workgroup_Size", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerows(rows)
    got = P.per_kernel(str(f), "FETCH_SIZE")["void pc::k_ntt_pass_a<F>"]
    assert got["launches"] == 2 and got["avg_per_launch_KiB"] == 1600.0 and got["grid_size"] == 16777216 and got["launches_all_geometries"] == 52


def test_bench_refuses_a_counter_traffic_below_the_algorithmic_bytes(capsys):
    import bench
    assert bench.checked_traffic(7.2e9, 2.684e9) == 7.2e9
    assert bench.checked_traffic(2.08e8, 2.684e9) is None          # round 5's figure
    assert bench.checked_traffic(None, 2.684e9) is None and bench.checked_traffic(5.0, None) == 5.0
    # and the committed summary holds the NTT batch's traffic above its algorithmic bytes
    assert bench.pmc_traffic("ntt:bls12_381:2^24", "ntt_hbm_bytes_per_batch") > 2684354560
