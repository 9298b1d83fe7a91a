"""bench.py end to end on a GPU box: the line the driver parses, the blocks it carries, and the N > 1 path launched the way
the driver launches N = 1 (`python bench.py --gpus N` with no WORLD_SIZE: the script starts its own ranks).  A one-GPU box
runs the two ranks on device 0 (PC_BENCH_DEVICES=0,0; RCCL refuses two ranks on one device, so that run uses gloo for the
collective -- the per-rank chunks, the exchange protocol and the in-line parity check are the same code)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_MAX = 6144          # bench.LINE_MAX_BYTES: the driver keeps a bounded tail of stdout (round 5's 25 KB line came back unparsed)
_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "parity", "detail")


def _check_line(line):
    """The stdout line itself: short, and carrying what the contract asks of it."""
    assert len(line) <= LINE_MAX, f"stdout line is {len(line)} bytes"
    c = json.loads(line)
    for k in _LINE_KEYS:
        assert k in c, k
    assert "workload" in c["config"]
    if c["roofline"] is not None:
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in c["roofline"], k
    return c


def _detail(tmp, line):
    """The full record the line points at (what the assertions below read), with the line under `_line`."""
    d = json.load(open(tmp))
    d["_line"] = line
    return d


def run_bench(argv, env_extra=None, timeout=900):
    import tempfile
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.pop("PC_BENCH_FULL_LINE", None)          # (tools/gpu_full_run.sh exports it for its own records: the tests check the driver's line)
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory() as td:
        env["PC_BENCH_DETAIL"] = os.path.join(td, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout)
        assert r.returncode == 0, r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, f"expected ONE line on stdout, got {len(lines)}"
        return _detail(env["PC_BENCH_DETAIL"], _check_line(lines[0]))


def test_default_line_small():
    """Every block of the default (N = 1) line, at scaled-down sizes."""
    d = run_bench(["--small", "--steps", "3"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "pairs/s" and d["value"] > 0
    assert d["parity"]["commit_ok"] and d["parity"]["open_ok"] and d["parity"]["device_poly_eval_ok"]
    assert d["parity"]["commitments_checked"] == 3 and d["parity"]["proofs_checked"] == 3
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["launches"] == 6
    assert 2 * rf["kernel_ms"] <= d["ms_per_step"] * 1.0001          # by construction (union of the launches' intervals)
    assert d["trait_shaped"]["parity_ok"] and d["trait_shaped"]["ms_per_commit_open"] > 0
    assert d["secondary"]["parity"]["commit_ok"] and d["secondary"]["parity"]["open_ok"]
    w = d["workloads"]
    assert all(r["parity_ok"] for r in w["latency"]["rows"].values())
    assert "cpu_port_commit_open_ms" in w["latency"]["rows"]["2^12"]          # BASELINE configs[0]
    assert w["batch"]["parity"]["all_commitments_closed_form_ok"] and w["batch"]["parity"]["oracle_horner_ok"]
    assert w["ipa"]["parity"]["commit_ok"] and w["ipa"]["parity"]["final_comm_key_ok"]
    assert w["ligero"]["parity"]["horner_spot_checks_ok"] and w["ligero"]["parity"]["one_row_vs_oracle_ntt_ok"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    # the line the driver parses carries the same figures in short form
    c = d["_line"]
    assert c["value"] == pytest.approx(d["value"], rel=1e-4) and c["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-4)
    assert c["roofline"]["frac"] == pytest.approx(rf["frac"], rel=1e-4) and c["roofline"]["arithmetic"]["frac"] > 0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["value_trait_shaped"]["ms_per_step"] == pytest.approx(d["trait_shaped"]["ms_per_commit_open"], rel=1e-4)
    assert set(c["workloads"]) == {"latency", "batch", "ipa", "ligero"}
    for name in ("batch", "ipa", "ligero"):
        wl = c["workloads"][name]
        assert wl["parity_ok"] is True and wl["ms"] > 0 and wl["roofline_frac"] > 0 and wl["arithmetic_frac"] > 0 and wl["cpu_ratio"] > 0, (name, wl)
    assert c["workloads"]["latency"]["parity_ok"] is True and c["secondary"]["parity_ok"] is True


def test_two_ranks_self_launched_kzg():
    """`python bench.py --gpus 2 --log-degree 16`: ranks started by the script, each with its REAL chunk of the one true
    SRS, every commitment / proof of the timed region checked against the closed form of the whole polynomial."""
    d = run_bench(["--gpus", "2", "--log-degree", "16", "--steps", "4", "--no-cpu-baseline"], {"PC_BENCH_DEVICES": "0,0"})
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert d["config"]["pairs_per_step"] == 2 * 2 * (1 << 16) - 1
    assert d["parity"]["commit_ok"] and d["parity"]["open_ok"]
    assert d["parity"]["commitments_checked"] == 4 and d["parity"]["proofs_checked"] == 4
    assert d["roofline"]["frac"] > 0 and d["exchange_host_ms"]["calls"] >= 4


def test_two_ranks_blocking_kzg():
    d = run_bench(["--gpus", "2", "--log-degree", "14", "--steps", "3", "--inflight", "0", "--no-cpu-baseline"], {"PC_BENCH_DEVICES": "0,0"})
    assert d["n_gpus"] == 2 and d["parity"]["commit_ok"] and d["parity"]["open_ok"]


def test_two_ranks_batch_and_rows():
    d = run_bench(["--gpus", "2", "--workload", "batch", "--small", "--polys", "8", "--steps", "2"], {"PC_BENCH_DEVICES": "0,0"})
    assert d["n_gpus"] == 2 and d["parity"]["all_commitments_closed_form_ok"] and d["parity"]["oracle_horner_ok"]
    d = run_bench(["--gpus", "2", "--workload", "ntt", "--small", "--steps", "2"], {"PC_BENCH_DEVICES": "0,0"})
    assert d["n_gpus"] == 2 and d["parity"]["horner_spot_checks_ok"] and d["parity"]["one_row_vs_oracle_ntt_ok"]
    # the commitment's root over rows that live on two ranks (chained column digests), re-hashed on the host
    assert d["sharded_commit"]["root_equal_on_all_ranks"] and d["sharded_commit"]["root_vs_host_rehash_ok"] is True


def test_group_mode_one_process():
    """`python bench.py --mode group`: one process, persistent worker thread per device context, commit+open jobs in flight;
    with 2 contexts on device 0 the results stay bit-identical to the closed forms (host and device-resident coefficients)."""
    d = run_bench(["--mode", "group", "--gpus", "2", "--log-degree", "14", "--steps", "4"], {"PC_BENCH_DEVICES": "0,0"})
    assert d["n_gpus"] == 2 and d["parity"]["all_steps_ok"] and d["parity"]["checked"] == 4
    d = run_bench(["--mode", "group", "--gpus", "1", "--log-degree", "14", "--steps", "3", "--group-coeffs", "device"])
    assert d["n_gpus"] == 1 and d["parity"]["all_steps_ok"]


def test_two_ranks_started_by_torchrun_as_the_driver_does(tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`:
    the command line of the driver's scaling runs (the ranks come from the environment, nothing is self-launched): one JSON line
    from rank 0, weak scaling, both ranks' chunks in the parity check."""
    env = dict(os.environ, PC_BENCH_DEVICES="0,0", PC_BENCH_DETAIL=str(tmp_path / "detail.json"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PC_BENCH_FULL_LINE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29791", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-degree", "14", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = _detail(env["PC_BENCH_DETAIL"], _check_line(lines[0]))
    assert d["_line"]["n_gpus"] == 2 and len(d["_line"]["per_rank_ms_per_step"]) == 2
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dist"]["world_size"] == 2
    assert d["parity"]["commit_ok"] and d["parity"]["open_ok"] and d["parity"]["commitments_checked"] == 3


# ---- eight ranks: the driver's scaling run rehearsed on one device (no 8-GPU node has ever been available to a round) -------------
EIGHT = {"PC_BENCH_DEVICES": ",".join(["0"] * 8)}


def test_eight_ranks_started_by_torchrun_as_the_driver_does(tmp_path):
    """The driver's N = 8 command line, all eight ranks on device 0 (gloo carries the collective: RCCL refuses ranks that share a
    device): ONE polynomial of 8 x 2^13 coefficients over eight real chunks of one true SRS, every commitment / proof of the timed
    region against the closed form of the WHOLE polynomial; the line carries what the N = 1 line carries."""
    env = dict(os.environ, PC_BENCH_DETAIL=str(tmp_path / "detail.json"), **EIGHT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PC_BENCH_FULL_LINE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29793", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--log-degree", "13", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = _detail(env["PC_BENCH_DETAIL"], _check_line(lines[0]))
    c = d["_line"]          # the N > 1 line obeys the same cap and carries roofline + per-rank ms_per_step
    assert c["n_gpus"] == 8 and len(c["per_rank_ms_per_step"]) == 8 and c["roofline"]["frac"] > 0 and c["roofline"]["arithmetic"]["frac"] > 0
    assert c["parity"]["commit_ok"] and c["parity"]["open_ok"] and c["exchange_host_ms"]["calls"] >= 3
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["scaling"] == "weak" and d["dist"]["world_size"] == 8
    assert len(d["per_rank_ms_per_step"]) == 8 and all(x > 0 for x in d["per_rank_ms_per_step"])
    assert d["config"]["pairs_per_step"] == 8 * 2 * (1 << 13) - 1
    assert d["parity"]["commit_ok"] and d["parity"]["open_ok"] and d["parity"]["commitments_checked"] == 3 and d["parity"]["proofs_checked"] == 3
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["frac"] > 0 and rf["kernel_ms"] > 0 and rf["arithmetic"]["frac"] > 0
    assert d["exchange_host_ms"]["calls"] >= 3
    assert d.get("cpu_baseline") is None          # reported on rank 0 at N = 1 only (the bench contract)


def test_eight_ranks_batch_and_rows():
    d = run_bench(["--gpus", "8", "--workload", "batch", "--small", "--polys", "8", "--steps", "2"], EIGHT, timeout=1200)
    assert d["n_gpus"] == 8 and d["dist"]["world_size"] == 8 and len(d["per_rank_ms_per_step"]) == 8
    assert d["parity"]["all_commitments_closed_form_ok"] and d["parity"]["oracle_horner_ok"]
    d = run_bench(["--gpus", "8", "--workload", "ntt", "--small", "--steps", "2"], EIGHT, timeout=1200)
    assert d["n_gpus"] == 8 and d["parity"]["horner_spot_checks_ok"] and d["parity"]["one_row_vs_oracle_ntt_ok"]
    assert d["sharded_commit"]["root_equal_on_all_ranks"] and d["sharded_commit"]["root_vs_host_rehash_ok"] is True
    assert d["roofline"]["frac"] > 0


def test_group_mode_eight_contexts():
    d = run_bench(["--mode", "group", "--gpus", "8", "--log-degree", "12", "--steps", "3"], EIGHT, timeout=1200)
    assert d["n_gpus"] == 8 and d["parity"]["all_steps_ok"] and d["parity"]["checked"] == 3
