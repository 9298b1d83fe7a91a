#!/usr/bin/env python3
"""bench.py -- the commit/open hot path of arkworks-rs/poly-commit on MI355X (BASELINE.json).

`value` (the driver's line): one step = one MarlinKZG10<Bls12_381> commit + one single-point open of one dense
polynomial of degree 2^24 per GPU, hiding off -- the shape bench-templates times (bench-templates/src/lib.rs:69-84,
106-138):
    commit : MSM of d+1 pairs over the resident SRS           (kzg10/mod.rs:175-178)
    open   : witness polynomial p/(x-z) on the device          (kzg10/mod.rs:217-240)
             MSM of d pairs                                    (kzg10/mod.rs:255-258)
SRS, coefficients and the quotient stay in HBM; only the two 96-byte affine results come back.  value = G1 (base,
scalar) pairs per second, whole job.  The same JSON line carries, timed in the same run:
    secondary            the 2^20 case (BASELINE configs[1])
    trait_shaped         the same commit+open through BLOCKING calls with HOST coefficients (what the trait's
                         commit(&poly)/open(&poly) hands over) at 2^24 and 2^20
    workloads.latency    blocking commit+open latency 2^10 .. 2^24 beside the CPU port (configs[0] = 2^12; crossover)
    workloads.batch      configs[2]: 64 x MarlinKZG10<Bn254> commits of degree 2^20
    workloads.ipa        configs[3]: InnerProductArgPC over Pallas, n = 2^22: commit + open
    workloads.ligero     configs[4]: Ligero over BLS12-381 Fr, 2^24 coefficients: 512 NTTs of 2^17 + digests + tree
    roofline, cpu_baseline, parity

Every workload runs on a TRUE structured reference string beta^i g generated on the device, so that its results are
checked in-line against closed forms (C = p(beta) g, W = q(beta) g -- the verifier's pairing equation with the
trapdoor known) evaluated by the CPU oracle: `parity` in the line says what was compared.

N > 1 (`--gpus N`: this script launches its own ranks with torch.distributed.run when WORLD_SIZE is not set; the
driver's torchrun invocation works as well): ONE polynomial of degree N * 2^24 whose SRS and coefficients are sharded
in contiguous chunks (weak scaling: fixed pairs per GPU); each rank runs the full Pippenger on its chunk of the SAME
true SRS and the partial commitments / proofs are combined with one all_gather + EC adds per step (RCCL has no EC
reduce op); the division carry crosses ranks as one Fr element.  See poly_commit_amd/sharded.py.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C
# stdio when its first communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the
# result line goes to the saved descriptor.
_RESULT_FD = os.dup(1)
os.dup2(2, 1)

# The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The MSM
# pipelines keep seven streams busy; once RCCL adds its own, independent streams share a queue and serialise
# (measured with one rank: 76.2 ms/step at 4 queues, 72.5 at 8 = the figure without RCCL).  Must be set before HIP
# initialises, i.e. before torch touches the GPU.
# (--mode group: the group's contexts and their pipelines are more streams than 4 queues as well -- a kernel trace showed the
# sort of one MSM queued behind the reductions of another on the same hardware queue: 77.0 ms/step at 4 queues)
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PC_BENCH_FORCE_DIST") or "group" in sys.argv:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


# ------------------------------------------------------------------------------------------------------------
# the result line.  The driver parses ONE JSON line from stdout and keeps only a bounded tail of it: the line is a
# compact summary of at most LINE_MAX_BYTES (round 5's 25 KB line came back unparsed); everything else -- per-fold
# tables, notes, definitions, the other configs' full blocks -- goes to the detail file next to it.
# ------------------------------------------------------------------------------------------------------------
LINE_MAX_BYTES = 6144
DETAIL_PATH = os.environ.get("PC_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
_PROSE_KEYS = ("note", "method", "definition", "kernels", "folds", "ms_calls", "per_round_ms", "loop_rates_per_s", "pass_phase_ms_sum")


def _g(d, path, default=None):
    for k in path.split("."):
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


def _num(v, sig=6):
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    if v == int(v) and abs(v) < 2 ** 53:          # byte and launch counts that travelled as floats stay exact
        return int(v) if abs(v) >= 1e6 else v
    return float(f"{v:.{sig}g}")


def _slim(o, str_max=96):
    """No prose, short strings, five significant digits: what is left of a block in the line."""
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in _PROSE_KEYS or k.endswith(("_note", "_definition", "_source", "_is", "_method")):
                continue
            out[k] = _slim(v, str_max)
        return out
    if isinstance(o, (list, tuple)):
        return [_slim(v, str_max) for v in o]
    if isinstance(o, str):
        return o if len(o) <= str_max else o[:str_max - 3] + "..."
    return _num(o)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _roofline_line(rf):
    if not isinstance(rf, dict):
        return None
    out = _pick(rf, ("bound", "kernel", "kernel_ms", "launches", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch"))
    ar = rf.get("arithmetic")
    if isinstance(ar, dict):
        out["arithmetic"] = _pick(ar, ("bound", "achieved", "peak", "frac", "unit"))
    return _slim(out, 100)


def _cpu_line(cb):
    if not isinstance(cb, dict):
        return None
    out = _pick(cb, ("value", "unit", "cores", "kind", "sample", "logical_cpus", "agrees_with_gpu"))
    if _g(cb, "per_core.value") is not None:
        out["per_core_value"] = cb["per_core"]["value"]
    return _slim(out, 120)


def _ratio(a, b):
    return None if not a or not b else a / b


def _workload_line(name, w):
    """{ms, roofline_frac, arithmetic_frac, cpu_ratio, parity_ok} (+ the two or three figures a reader of that config looks for)."""
    if not isinstance(w, dict) or "error" in w:
        return _slim(w)
    if name == "latency":
        rows = w.get("rows", {})
        return {"ms": {k: _num(r.get("gpu_commit_open_ms"), 4) for k, r in rows.items()},
                "cpu_port_ms": {k: _num(r["cpu_port_commit_open_ms"], 4) for k, r in rows.items() if "cpu_port_commit_open_ms" in r},
                "parity_ok": all(r.get("parity_ok") for r in rows.values()) if rows else None,
                "cpu_faster_up_to_log_degree": w.get("cpu_faster_up_to_log_degree")}
    par = w.get("parity") or {}
    flags = [v for k, v in par.items() if k.endswith("_ok")]
    out = {"ms": w.get("open_ms") if name == "ipa" else w.get("ms_per_step"),
           "roofline_frac": _g(w, "roofline.frac"), "arithmetic_frac": _g(w, "roofline.arithmetic.frac"),
           "kernel_ms": _g(w, "roofline.kernel_ms"), "traffic": _g(w, "roofline.traffic"),
           "cpu_ratio": _ratio(w.get("open_coeffs_per_s") if name == "ipa" else w.get("value"), _g(w, "cpu_baseline.value")),
           "cpu_cores": _g(w, "cpu_baseline.cores"),
           "parity_ok": all(bool(f) for f in flags) if flags else None}
    if name == "ipa":
        out.update(commit_ms=w.get("commit_ms"), open_arithmetic_frac=_g(w, "roofline_open.frac"),
                   fixed_key_rounds_ms=_g(w, "roofline_open.fixed_key_rounds.ms_total"), msm_wait_ms=_g(w, "roofline_open.msm_wait_ms"))
    if name == "ligero":
        out.update(ntt_phase_ms=w.get("ntt_phase_ms"), column_hash_ms=w.get("column_hash_blake2s_ms"), tree_ms=w.get("merkle_tree_sha256_ms"),
                   trait_shaped_ms=_g(w, "trait_shaped.ms_per_commit"))
    if name == "batch":
        out.update(trait_shaped_ms=_g(w, "trait_shaped.ms_per_step"), ms_per_commitment=w.get("ms_per_commitment"))
    return _slim(out)


def compact_line(obj):
    """The driver's line: the contract's keys, compact `roofline` / `cpu_baseline` / `parity`, one short entry per extra workload."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: _num(obj[k]) if not isinstance(obj[k], str) else obj[k] for k in head if k in obj}
    out["config"] = _slim(_pick(obj.get("config", {}), ("workload", "curve", "log_degree", "pairs_per_step", "inflight", "srs_window_table",
                                                         "srs_window_table_build_ms", "parallelism", "polys_per_s", "mode")), 200)
    for k in ("dist", "per_rank_ms_per_step", "commit_open_per_s", "value_trait_shaped", "value_h2d_inclusive", "blocking_msm_ms",
              "exchange_host_ms", "msm_phase_ms", "ntt_phase_ms", "column_hash_blake2s_ms", "merkle_tree_sha256_ms", "ms_per_commitment",
              "sharded_commit", "srs_window_table_build_ms"):
        if obj.get(k) is not None:
            out[k] = _slim(obj[k])
    out["roofline"] = _roofline_line(obj.get("roofline"))
    if obj.get("cpu_baseline") is not None:
        out["cpu_baseline"] = _cpu_line(obj["cpu_baseline"])
        out["gpu_over_cpu"] = _num(_ratio(obj.get("value"), _g(obj, "cpu_baseline.value")), 4)
    out["parity"] = _slim({k: v for k, v in (obj.get("parity") or {}).items() if not isinstance(v, str) or len(v) < 40})
    sec = obj.get("secondary")
    if isinstance(sec, dict):
        out["secondary"] = _slim({"log_degree": _g(sec, "config.log_degree"), "value": sec.get("value"), "ms_per_step": sec.get("ms_per_step"),
                                  "steps": sec.get("steps"), "blocking_msm_ms": sec.get("blocking_msm_ms"),
                                  "trait_shaped_ms": _g(sec, "trait_shaped.ms_per_commit_open"),
                                  "roofline_frac": _g(sec, "roofline.frac"), "arithmetic_frac": _g(sec, "roofline.arithmetic.frac"),
                                  "kernel_ms": _g(sec, "roofline.kernel_ms"),
                                  "parity_ok": bool(_g(sec, "parity.commit_ok") and _g(sec, "parity.open_ok"))})
    if isinstance(obj.get("workloads"), dict):
        out["workloads"] = {n: _workload_line(n, w) for n, w in obj["workloads"].items()}
    for k in ("bench_wall_s", "small_sizes"):
        if k in obj:
            out[k] = _num(obj[k], 4)
    out["detail"] = os.path.relpath(DETAIL_PATH, ROOT) if DETAIL_PATH.startswith(ROOT) else DETAIL_PATH
    # never above the cap: shed the optional blocks, least important first
    for victim in ("msm_phase_ms", "exchange_host_ms", "value_h2d_inclusive", "secondary", "workloads", "dist", "sharded_commit", "config"):
        if len(json.dumps(out)) <= LINE_MAX_BYTES:
            break
        out.pop(victim, None)
    return out


def emit(obj):
    try:
        with open(DETAIL_PATH, "w") as f:
            json.dump(obj, f)
    except OSError as e:
        log(f"detail file not written ({e})")
    if os.environ.get("PC_BENCH_FULL_LINE"):          # tools/gpu_full_run.sh, tools/gpu_probe.sh: the full record on stdout
        os.write(_RESULT_FD, (json.dumps(obj) + "\n").encode())
        return
    line = json.dumps(compact_line(obj))
    assert len(line) <= LINE_MAX_BYTES, len(line)
    os.write(_RESULT_FD, (line + "\n").encode())


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


PAIR_BYTES = {"bls12_381": 128, "bn254": 96, "pallas": 96}   # affine base + 32-byte scalar
HBM_PEAK_GBPS = 8000.0                                        # MI355X_MICROARCH.md: 8 TB/s spec
FR_BITS = {"bls12_381": 255, "bn254": 254, "pallas": 255}


# ------------------------------------------------------------------------------------------------------------
# launching: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own ranks
# ------------------------------------------------------------------------------------------------------------
def self_launch(n_gpus, argv):
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("launching", " ".join(cmd))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    if line is not None:
        os.write(_RESULT_FD, (line + "\n").encode())
    sys.exit(proc.returncode if proc.returncode else (0 if line is not None else 1))


class Dist:
    """World of this run: torch.distributed on RCCL ('nccl') -- or gloo when several ranks share one GPU
    (PC_BENCH_DEVICES=0,0: RCCL refuses two ranks on one device; used by the GPU test on a one-GPU box)."""

    def __init__(self, args):
        import torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        devs = os.environ.get("PC_BENCH_DEVICES")
        self.device = int(devs.split(",")[self.local_rank]) if devs else self.local_rank
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if self.device >= torch.cuda.device_count():
            raise SystemExit(f"rank {self.rank}: device {self.device} requested, {torch.cuda.device_count()} visible "
                             "(PC_BENCH_DEVICES=0,0,... maps ranks onto fewer GPUs)")
        torch.cuda.set_device(self.device)
        self.dist = None
        self.backend = None
        if self.world > 1 or os.environ.get("PC_BENCH_FORCE_DIST"):
            import torch.distributed as dist
            shared = devs is not None and len(set(devs.split(","))) < len(devs.split(","))
            self.backend = args.backend or ("gloo" if shared else "nccl")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            else:
                dist.init_process_group("gloo")
            self.dist = dist
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def gather_floats(self, vals):
        """list of floats -> (world, len) array on every rank."""
        import torch
        v = np.asarray(vals, dtype=np.float64).reshape(-1)
        if self.dist is None:
            return v.reshape(1, -1)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.from_numpy(v.copy()).to(dev)
        out = torch.empty(self.world * t.numel(), dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape(self.world, -1)

    def gather_u64(self, arr):
        from poly_commit_amd import sharded
        return sharded.all_gather_u64(self.dist, self.world, arr)

    def info(self):
        if self.dist is None:
            return None
        return {"backend": self.backend + (" (RCCL)" if self.backend == "nccl" else ""), "world_size": self.dist.get_world_size(),
                "devices": os.environ.get("PC_BENCH_DEVICES") or "one GPU per rank"}

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------
# synthetic inputs, generated on the device
# ------------------------------------------------------------------------------------------------------------
def fr_modulus(curve):
    from poly_commit_amd import sharded
    return sharded.FR_MODULUS[curve]


def mont_limbs(curve, v):
    """Python int -> Montgomery-form Fr, (4,) uint64."""
    p = fr_modulus(curve)
    return np.frombuffer(((v % p) * (1 << 256) % p).to_bytes(32, "little"), dtype="<u8").astype(np.uint64)


def from_mont_limbs(curve, limbs):
    p = fr_modulus(curve)
    v = int.from_bytes(np.ascontiguousarray(limbs, dtype="<u8").tobytes(), "little")
    return v * pow(1 << 256, -1, p) % p


def seed_fr(curve, seed):
    """One reproducible field element as a Python int (the oracle's SplitMix64 stream)."""
    import oracle_lib as O
    return int.from_bytes(O.gen_scalars(curve, seed, 1).tobytes(), "little") % fr_modulus(curve)


def rand_fr_device(seed, n):
    """n field elements in their in-memory (Montgomery) form, uniformly below 2^252 < r for all three scalar fields:
    what DensePolynomial::rand's coefficients look like in memory (marlin_pc/mod.rs:550-556), without 2^24 CPU draws."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    t[:, 3] &= (1 << 60) - 1
    return t


def host_u64(t):
    """device / host int64 tensor -> contiguous uint64 numpy array."""
    return np.ascontiguousarray(t.cpu().numpy().view(np.uint64))


def true_srs_points(ctx, curve, g_xy, beta, first_power, count):
    """beta^(first_power + j) * g for j < count as a device tensor: KZG10::setup's `g.batch_mul(&powers_of_beta)`
    (kzg10/mod.rs:68-83) through pc_hip_fr_powers / pc_hip_fr_lincomb / pc_hip_fixed_base_batch_mul."""
    import torch
    p = fr_modulus(curve)
    pw = torch.empty((count, 4), dtype=torch.int64, device="cuda")
    ctx.fr_powers(curve, mont_limbs(curve, beta), count, pw.data_ptr())
    if first_power != 0:
        lead = pow(beta, first_power, p) if first_power > 0 else pow(pow(beta, -1, p), -first_power, p)
        sc = torch.empty_like(pw)
        ctx.fr_lincomb(curve, [pw.data_ptr()], mont_limbs(curve, lead).reshape(1, 4), n_out=count, out=sc.data_ptr(), lens=[count])
        pw = sc
    fq_limbs = 6 if curve == "bls12_381" else 4
    pts = torch.empty((count, 2 * fq_limbs), dtype=torch.int64, device="cuda")
    ctx.fixed_base_batch_mul(curve, g_xy, pw.data_ptr(), count, pts.data_ptr())
    torch.cuda.synchronize()
    return pts


def oracle_scalar_mul(curve, g_xy, k):
    """k * g on the CPU oracle (one double-and-add): the closed forms the device results are compared with."""
    import oracle_lib as O
    p = fr_modulus(curve)
    sc = np.frombuffer((k % p).to_bytes(32, "little"), dtype="<u8").astype(np.uint64).reshape(1, 4)
    return O.msm_naive(curve, np.ascontiguousarray(g_xy).reshape(1, -1), np.ascontiguousarray(sc))


_PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json")


def checked_traffic(traffic, algorithmic_bytes):
    """Counter traffic of a launch can exceed its algorithmic bytes (re-reads), never undercut them: a figure below means the PMC summary
    averaged launches of another size into this one (round 5's NTT block did) -- refuse it rather than print it."""
    if traffic is None or algorithmic_bytes is None:
        return traffic
    if traffic < 0.98 * algorithmic_bytes:
        log(f"PMC traffic {traffic:.3e} B is below the algorithmic {algorithmic_bytes:.3e} B of the launch: refused (traffic = null)")
        return None
    return traffic


def pmc_traffic(key, field="accumulate_hbm_bytes_per_launch"):
    """HBM bytes per launch from the newest committed PMC summary that has the key (profiles/r05_pmc_traffic.json, else r04, r03: FETCH_SIZE
    doubled per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE, separate rocprofv3 --pmc passes of the same workload -- NOT a
    measurement of this run)."""
    for name in _PMC_FILES:
        try:
            v = json.load(open(os.path.join(ROOT, "profiles", name))).get(field, {}).get(key)
            if v is not None:
                _PMC_HIT[0] = "profiles/" + name
                return v
        except Exception:
            pass
    return None


_PMC_HIT = [None]


def pmc_source():
    if _PMC_HIT[0]:
        return _PMC_HIT[0]
    for name in _PMC_FILES:
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            return "profiles/" + name
    return None


def union_ms(intervals):
    """Total length of the union of [a, b] intervals."""
    tot, end = 0.0, -1e30
    for a, b in sorted(intervals):
        if b <= end:
            continue
        tot += b - max(a, end)
        end = b
    return tot


_MICRO = None
_MADD_COMMITTED = {"bls12_381": 6.67e9, "bn254": 14.04e9, "pallas": 15.9e9}      # profiles/r05_microbench.txt
# committed figures of the lines tools/microbench prints since round 5 (profiles/r05_microbench.txt), used when the binary is missing
_MICRO_COMMITTED = {"fmul": {"bls12_381_fq": 57.4e9, "bn254_fq": 129.3e9, "pallas_fq": 157.0e9, "bls12_381_fr": 129.9e9, "bn254_fr": 130.6e9, "pallas_fr": 157.0e9},
                    "butterfly": {"bls12_381_fr": 111.1e9, "bn254_fr": 112.5e9, "pallas_fr": 128.0e9},
                    "butterfly4": {"bls12_381_fr": 110.3e9, "bn254_fr": 113.3e9, "pallas_fr": 129.7e9},
                    "jac_dbl": {"bls12_381": 8.80e9, "bn254": 17.55e9, "pallas": 20.5e9}, "jac_madd": {"bls12_381": 5.76e9, "bn254": 11.89e9, "pallas": 13.79e9}}


def microbench():
    """tools/microbench run ONCE on this GPU (rank 0) and parsed: the memory-free loops the kernels are priced against --
    'madd' (XYZZ += affine per curve, best of 2 / 3 / 4 waves per SIMD), 'fmul' (Montgomery products per field), 'butterfly' /
    'butterfly4' (the NTT's radix-2 butterfly and the radix-4 register group per scalar field), 'jac_dbl' / 'jac_madd' (the
    IPA key fold's ladder steps per curve).  Values per second."""
    global _MICRO
    if _MICRO is not None:
        return _MICRO
    _MICRO = {"madd": {}, "fmul": {}, "butterfly": {}, "butterfly4": {}, "jac_dbl": {}, "jac_madd": {}, "source": None}
    exe = os.path.join(ROOT, "tools", "microbench")
    if os.path.exists(exe):
        try:
            import subprocess
            txt = subprocess.run([exe], capture_output=True, text=True, timeout=240).stdout
            fl = lambda t: [float(x) for x in t if x.replace(".", "", 1).isdigit() and "." in x]      # noqa: E731
            for line in txt.splitlines():
                t = line.split()
                ls = line.strip()
                if ls.startswith("kernel form") and "M madd/s" in line and fl(t):
                    _MICRO["madd"][t[2]] = max(fl(t)) * 1e6
                elif ls.startswith("fmul ") and "G mulmod/s" in line:
                    _MICRO["fmul"][t[1]] = fl(t)[-1] * 1e9
                elif ls.startswith("ntt butterfly ") and "G butterflies/s" in line:
                    _MICRO["butterfly"][t[2]] = fl(t[:6])[-1] * 1e9
                elif ls.startswith("ntt radix-4 group ") and "G butterflies/s" in line:
                    _MICRO["butterfly4"][t[3]] = fl(t[:7])[-1] * 1e9
                elif ls.startswith("jacobian ") and "M ops/s" in line and len(fl(t)) >= 2:
                    _MICRO["jac_dbl"][t[1]], _MICRO["jac_madd"][t[1]] = fl(t)[0] * 1e6, fl(t)[1] * 1e6
            _MICRO["source"] = "tools/microbench, this run"
        except Exception:
            pass
    return _MICRO


def micro_rate(kind, name):
    """(rate per second, source) of one microbench line; the committed figure when the binary did not run; (None, None) if neither."""
    m = microbench()
    if name in m.get(kind, {}):
        return m[kind][name], m["source"]
    if name in _MICRO_COMMITTED.get(kind, {}):
        return _MICRO_COMMITTED[kind][name], "profiles/r05_microbench.txt"
    return None, None


def madd_peak(curve):
    """Mixed additions per second of a pure-arithmetic loop (no memory traffic) of the same XYZZ += affine addition the accumulate
    kernel runs for this curve (lazily reduced where the field allows it): tools/microbench measured live on this GPU (rank 0,
    once: its 'kernel form <curve> at 2 / 3 / 4 waves per SIMD' line, best of the three), else the committed figure."""
    m = microbench()
    if curve in m["madd"]:
        return {"madd_per_s": m["madd"][curve], "source": m["source"]}
    return {"madd_per_s": _MADD_COMMITTED[curve], "source": "profiles/r05_microbench.txt"}


def ntt_products(log_n, in_cols):
    """Twiddle products and butterflies one row of pc_hip_ntt_batch executes (csrc/ntt.hpp, kept in step with it by
    tests/test_bench_cpu.py): the four-step split N = 2^lg1 x 2^lg2, the stages pass A skips over the zero padding, the products
    the kernels skip because the twiddle is 1, and the omega_N^(i2 j1) products between the passes.  Returns (products,
    butterflies executed, nominal (N/2) log2 N butterflies of SURVEY.md 8d)."""
    N = 1 << log_n
    lg1 = (log_n + 1) // 2
    lg2 = log_n - lg1
    zskip = 0
    while zskip < lg1 and in_cols <= (N >> (zskip + 1)):
        zskip += 1

    def stages(lines, lg, first):
        prod = bfly = 0
        length = 1 << lg
        s = first
        if (lg - first + 1) & 1 and s <= lg:
            h = 1 << (s - 1)
            prod += lines * (length // 2) * (h - 1) // h
            bfly += lines * (length // 2)
            s += 1
        while s + 1 <= lg:
            h = 1 << (s - 1)
            groups = lines * (length // 4)
            prod += groups * 3 * (h - 1) // h + groups        # x1 w1, x3 w1, a2 w2 unless j == 0; a3 w2' always
            bfly += groups * 4
            s += 2
        return prod, bfly
    pa, ba = stages(1 << lg2, lg1, zskip + 1)
    pb, bb = stages(1 << lg1, lg2, 1)
    between = ((1 << lg1) - 1) * ((1 << lg2) - 1)
    return pa + pb + between, ba + bb, (N // 2) * log_n


def msm_roofline(curve, pairs_per_launch, digits, kernel_ms, launches, kernel, traffic_key=None, extra=None):
    """The prescribed HBM roofline of one accumulate launch (128 / 96 algorithmic bytes per pair) and, beside it, the
    bound that actually holds (VALU: mixed additions per second against a memory-free loop of the same addition)."""
    bytes_per_launch = pairs_per_launch * PAIR_BYTES[curve]
    ach = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms and kernel_ms > 0 else None
    pk = madd_peak(curve)
    adds = pairs_per_launch * digits
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS if ach else None,
         "traffic": checked_traffic(pmc_traffic(traffic_key), bytes_per_launch) if traffic_key else None,
         "traffic_source": f"{pmc_source()} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, FETCH_SIZE doubled per "
                           "the guide's gfx950 note; not measured in this run)",
         "kernel": kernel, "kernel_ms": kernel_ms, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
         "arithmetic": {"bound": "valu", "unit": "mixed additions/s (XYZZ += affine, 8M + 2S in Fq)",
                        "achieved": adds / (kernel_ms * 1e-3) if kernel_ms else None, "peak": pk["madd_per_s"],
                        "frac": adds / (kernel_ms * 1e-3) / pk["madd_per_s"] if kernel_ms else None, "peak_source": pk["source"],
                        "additions_per_launch": adds, "digits_per_scalar": digits}}
    if extra:
        r.update(extra)
    return r


def effective_cores():
    """CPUs this process can actually use: the logical CPU count, cut by the scheduler affinity and by the container's cgroup CPU
    quota (the GPU boxes of this pool show 256 logical CPUs and a quota of 16: 256 threads there are 16 cores' worth)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(math.ceil(int(q) / int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except Exception:
            pass
    return n


def cpu_baseline(curve, srs, log_d, budget_s=30.0):
    """The CPU port (oracle/fast_msm.hpp: ark-ec's signed-digit bucket method with XYZZ buckets, an unrolled 64-bit CIOS multiplier
    and (window, chunk) tasks over all cores) timed on this box's cores on the same workload: ONE MSM over the leading 2^k points of
    the resident true SRS -- k = log_d (the full size the metric is quoted on) when the box does it within the budget -- plus the
    single-thread rate at 2^20 with ark-ec's own window rule (the per-core figure: ark-ec parallelises over windows only)."""
    import oracle_lib as O
    cores = effective_cores()
    n0 = 1 << min(18, log_d)
    b = srs.read(1, n0)
    s = O.gen_scalars(curve, 1, n0)
    t = time.perf_counter()
    O.msm_pippenger(curve, b, s, cores, 2)
    rate = n0 / max(time.perf_counter() - t, 1e-6)
    t = time.perf_counter()
    O.msm_pippenger(curve, b, s, 1, 2)
    rate1 = n0 / max(time.perf_counter() - t, 1e-6)
    # single thread: 2^20 if that fits a third of the budget (the rate per pair falls slowly with the size: wider windows)
    lg1 = min(18, log_d)
    while lg1 < min(20, log_d) and (1 << (lg1 + 1)) / rate1 < budget_s / 3:
        lg1 += 1
    lg = min(18, log_d)
    while lg < log_d and (1 << (lg + 1)) / rate * 1.2 < budget_s / 2:
        lg += 1
    n = 1 << lg
    b = srs.read(1, n)
    s = O.gen_scalars(curve, 2, n)
    t = time.perf_counter()
    got = O.msm_pippenger(curve, b, s, cores, 2)
    dt = time.perf_counter() - t
    n1 = 1 << lg1
    t = time.perf_counter()
    got1 = O.msm_pippenger(curve, b[:n1], s[:n1], 1, 2)
    dt1 = time.perf_counter() - t
    # the port is itself checked here: the GPU's MSM over the same pairs (pc_hip_msm on the resident key) must give the same point
    dev, _ = srs.msm(s, n=n, base_offset=1)
    dev1, _ = srs.msm(np.ascontiguousarray(s[:n1]), n=n1, base_offset=1)
    return {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "logical_cpus": os.cpu_count(),
            "sample": f"1 MSM of 2^{lg} {curve} G1 pairs (the leading points of the same true SRS) on {cores} threads (= the CPUs the container's "
                      f"cgroup quota / affinity gives this process, of {os.cpu_count()} logical), {dt:.2f} s; "
                      f"restated ark-ec signed-digit bucket method (oracle/fast_msm.hpp), NOT ark-ec itself",
            "per_core": {"value": n1 / dt1, "unit": "pairs/s", "threads": 1,
                         "sample": f"1 MSM of 2^{lg1} pairs on ONE thread with ark-ec's window rule, {dt1:.2f} s"},
            "agrees_with_gpu": bool((got == dev).all() and (got1 == dev1).all()),
            "note": "a port of the algorithm, not the reference crate (no Rust toolchain here): ark-ec 0.5 itself parallelises over its "
                    "~16 windows only; its published per-core rates are of the order of 1e5 pairs/s"}


def cpu_baseline_batch(curve, srs, vec, n, polys, budget_s=12.0):
    """configs[2] beside the GPU: the CPU port's MSM of ONE polynomial of the batch (the same resident SRS chunk, the same
    coefficients) on this box's cores, repeated while the budget lasts -- MarlinKZG10::commit loops its polynomials sequentially
    (marlin_pc/mod.rs:192), so the batch costs `polys` times that."""
    import oracle_lib as O
    cores = effective_cores()
    b = srs.read(0, n)
    done, dt, ok = 0, 0.0, True
    while done < min(polys, 4) and (done == 0 or dt * (done + 1) / done < budget_s):
        sc = O.f_from_mont(curve, 1, host_u64(vec[done]))
        t = time.perf_counter()
        got = O.msm_pippenger(curve, b, sc, cores, 2)
        dt += time.perf_counter() - t
        dev, _ = srs.msm(vec[done].data_ptr(), n=n, montgomery=True)
        ok = ok and bool((got == dev).all())
        done += 1
    return {"value": done * n / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "logical_cpus": os.cpu_count(),
            "sample": f"{done} of the {polys} polynomials: one MSM of {n} {curve} G1 pairs each over the same resident SRS on {cores} threads, {dt:.2f} s "
                      f"(oracle/fast_msm.hpp: restated ark-ec signed-digit bucket method, NOT ark-ec itself)",
            "projected_ms_per_step": polys * n / (done * n / dt) * 1e3, "agrees_with_gpu": ok}


def cpu_baseline_ipa(curve, log_n_full, budget_s=12.0):
    """configs[3] beside the GPU: the oracle's InnerProductArgPC::open halving loop (oracle.cpp orc_ipa_rounds: two MSMs, two inner
    products, the coefficient / z folds and the key fold k_l += u k_r with batch normalisation per round, ipa_pc/mod.rs:664-711) on
    this box's cores at the largest n = 2^k that fits the budget; the cost per element of the key fold is size-independent, so the
    rate (coefficients of the opened polynomial per second) carries to 2^22 up to the MSMs' window widths."""
    import oracle_lib as O
    cores = effective_cores()
    lg, dt, n = 10, None, None
    while True:
        n = 1 << lg
        key = O.gen_bases(curve, n + 1)
        co = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xA11CE, n))
        z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xB0B, 1))[0]
        ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1, lg))
        t = time.perf_counter()
        O.ipa_rounds(curve, np.ascontiguousarray(key[:n]), co, z, key[n], ch, cores)
        dt = time.perf_counter() - t
        if lg >= log_n_full or dt * 2.2 > budget_s / 2 or lg >= 18:
            break
        lg += 1
    return {"value": n / dt, "unit": "coefficients/s (one opening of n coefficients)", "cores": cores, "kind": "port", "logical_cpus": os.cpu_count(),
            "sample": f"one open halving loop at n = 2^{lg} ({lg} rounds, challenges supplied) on {cores} threads, {dt:.2f} s; the oracle's restatement of "
                      f"ipa_pc/mod.rs:664-711 (full-width double-and-add key fold, Pippenger round MSMs), NOT ark-ec itself",
            "projected_open_ms_at_full_size": (1 << log_n_full) / (n / dt) * 1e3}


def cpu_baseline_ntt(curve, rows, n_cols, log_n, budget_s=10.0):
    """configs[4] beside the GPU: the oracle's batched radix-2 NTT (oracle.cpp orc_ntt_batch: iterative DIT, precomputed root powers,
    one thread per row) of a few rows of the same shape on this box's cores."""
    import oracle_lib as O
    cores = effective_cores()
    r = min(rows, max(cores, 1))
    mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0500, r * n_cols)).reshape(r, n_cols, 4)
    O.ntt_batch(curve, mat[:1], log_n, 1)                     # (root tables, first-touch)
    reps, dt = 0, 0.0
    while reps == 0 or (dt * (reps + 1) / reps < budget_s and reps < 4):
        t = time.perf_counter()
        O.ntt_batch(curve, mat, log_n, cores)
        dt += time.perf_counter() - t
        reps += 1
    return {"value": reps * r * n_cols / dt, "unit": "coeffs/s", "cores": cores, "kind": "port", "logical_cpus": os.cpu_count(),
            "sample": f"{reps} x {r} rows of {n_cols} coefficients -> 2^{log_n} evaluations (the same zero-padded forward NTT, linear_codes/utils.rs:112-127) on "
                      f"{cores} threads, {dt:.2f} s; the oracle's radix-2 NTT, NOT ark-poly itself",
            "projected_ms_per_step": rows * n_cols / (reps * r * n_cols / dt) * 1e3}


# ------------------------------------------------------------------------------------------------------------
# KZG commit + open (configs[1], north star)
# ------------------------------------------------------------------------------------------------------------
class KzgSetup:
    """One rank's resident state for a sharded KZG job on the TRUE SRS beta^i g: chunk r holds powers
    [r n - 1, (r+1) n) (one below its coefficients, so that commit and open address the same resident chunk)."""

    def __init__(self, ctx, D, args, curve, n, seed):
        import torch
        import oracle_lib as O
        from poly_commit_amd import sharded
        self.ctx, self.D, self.curve, self.n = ctx, D, curve, n
        self.p = fr_modulus(curve)
        self.g = O.gen_bases(curve, 1)[0]
        self.beta = seed_fr(curve, 0xBE7A24)
        self.z = seed_fr(curve, 0x2EE7)
        self.eng = sharded.HipEngine(ctx, curve)
        self.eng.glv_table = None if args.glv_table < 0 else bool(args.glv_table)
        self.job = sharded.ShardedKzg(self.eng, curve, D.rank, D.world, D.dist)
        t0 = time.perf_counter()
        pts = true_srs_points(ctx, curve, self.g, self.beta, D.rank * n - 1, n + 1)
        self.srs_gen_ms = (time.perf_counter() - t0) * 1e3
        self.job.load_srs_chunk(pts.data_ptr(), precompute=bool(args.precompute), n=n + 1)
        del pts
        self.coeffs = rand_fr_device(seed + D.rank, n)
        self.job.set_point(mont_limbs(curve, self.z))
        torch.cuda.synchronize()

    def closed_forms(self):
        """(commitment, proof) of the WHOLE polynomial from oracle evaluations of every rank's shard:
        C = p(beta) g, W = ((p(beta) - p(z)) / (beta - z)) g."""
        import oracle_lib as O
        host = host_u64(self.coeffs)
        ev = np.concatenate([O.poly_eval(self.curve, host, mont_limbs(self.curve, self.beta)),
                             O.poly_eval(self.curve, host, mont_limbs(self.curve, self.z))])
        # the device's own evaluation must agree with the oracle's Horner
        dev_z = self.ctx.poly_eval(self.curve, self.coeffs.data_ptr(), mont_limbs(self.curve, self.z), n=self.n)
        ok_eval = bool((dev_z == ev[4:]).all())
        allv = self.D.gather_u64(ev)
        p, n = self.p, self.n
        pb = pz = 0
        for r in range(self.D.world):
            pb = (pb + pow(self.beta, r * n, p) * from_mont_limbs(self.curve, allv[r, :4])) % p
            pz = (pz + pow(self.z, r * n, p) * from_mont_limbs(self.curve, allv[r, 4:])) % p
        want_c = oracle_scalar_mul(self.curve, self.g, pb)
        want_w = oracle_scalar_mul(self.curve, self.g, (pb - pz) * pow(self.beta - self.z, -1, p) % p)
        return want_c, want_w, ok_eval

    def free(self):
        import torch
        self.eng.srs.free()
        del self.coeffs
        torch.cuda.empty_cache()


def kzg_case(ctx, D, args, curve, log_degree, steps, warmup, with_h2d, seed=0x5EED0001):
    """One KZG commit+open workload on this rank.  Returns a dict (timings are the max over ranks)."""
    import collections
    import torch
    world, rank, dist = D.world, D.rank, D.dist
    d = 1 << log_degree
    n = d + 1 if world == 1 else d          # coefficients held by this rank
    S = KzgSetup(ctx, D, args, curve, n, seed)
    eng, job, coeffs = S.eng, S.job, S.coeffs

    depth = max(0, args.inflight)
    eng_blocking = depth == 0 and dist is None      # --inflight 0: the blocking ABI calls themselves (pc_hip_msm)
    eng.blocking = eng_blocking
    pending = collections.deque()
    results = []                            # (kind, point) of everything that left the pipeline, in order

    def drain():
        if dist is not None and pending:
            kinds = [k for k, _ in pending]
            _, outs = job.exchange(None, 0, [f for _, f in pending])
            pending.clear()
            results.extend(zip(kinds, outs))
        while pending:
            k, f = pending.popleft()
            results.append((k, f.result()))

    def step_resident(_k):
        # commit and open of one polynomial; up to `depth` results stay in flight so that the latency-bound tail of one
        # MSM overlaps the bucket accumulation of the next
        if dist is not None and depth > 0:
            # N > 1, pipelined: ONE collective per step -- this step's shard evaluations (the division carries) travel with
            # the partial points of the step that left the pipeline (ShardedKzg.exchange), issued while the previous
            # step's MSMs are still in flight
            done = [pending.popleft() for _ in range(max(0, len(pending) - 2 * (depth - 1)))]
            carry, outs = job.exchange(coeffs, n, [f for _, f in done])
            results.extend(zip([k for k, _ in done], outs))
            pending.append(("commit", job.commit_async(coeffs, n)))
            pending.append(("open", job.open_async(coeffs, n, prepared=True, carry=carry)))
            return
        carry = job.open_prepare(coeffs, n)
        pending.append(("commit", job.commit_async(coeffs, n)))
        if depth == 0:                       # strictly blocking calls: the commitment is back before the open starts
            k, f = pending.popleft()
            results.append((k, f.result()))
        pending.append(("open", job.open_async(coeffs, n, prepared=True, carry=carry)))
        while len(pending) > depth:
            k, f = pending.popleft()
            results.append((k, f.result()))

    def timed(step_fn, steps, warmup):
        for k in range(warmup):
            step_fn(k)
        drain()
        eng.phases, eng.marks = [], []
        results.clear()
        D.barrier()
        torch.cuda.synchronize()
        ctx.set_timing(True)                 # re-bases the absolute marks at the start of the timed region
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step_fn(k)
        drain()                              # every commitment and proof is on the host here
        torch.cuda.synchronize()
        D.barrier()
        dt = time.perf_counter() - t0
        per_rank = D.gather_floats([dt])[:, 0]
        return float(per_rank.max()), [float(x) for x in per_rank], list(eng.phases), list(eng.marks)

    if steps is None and dist is not None:
        steps = 20          # every rank must run the same number of steps
    if steps is None:       # no --steps: long enough for >= 1.2 s of timed region (the driver's gpu_busy sampler needs it)
        t0 = time.perf_counter()
        step_resident(0); drain()
        torch.cuda.synchronize()
        est = time.perf_counter() - t0
        steps = max(10, int(math.ceil(1.2 / max(est, 1e-4))))
    dt, per_rank_dt, phases, marks = timed(step_resident, steps, warmup)
    timed_results = list(results)

    # accumulate launches of the timed region: [marks[3], marks[4]] of every MSM of this rank.  Launches of different
    # pipelines overlap, so the per-launch duration the region sustains is the UNION of the intervals / launches
    # (<= step time / 2 by construction); the plain average of the brackets is reported beside it.
    acc_iv = [(m[3], m[4]) for m in marks if m is not None and m[3] >= 0 and m[4] >= m[3]]
    acc_union_ms = union_ms(acc_iv) / max(1, len(acc_iv))
    acc_bracket_ms = float(np.mean([b - a for a, b in acc_iv])) if acc_iv else 0.0

    # ---- in-line parity: every commitment / proof of the timed region against the closed forms ---------------
    want_c, want_w, ok_eval = S.closed_forms()
    n_c = sum(1 for k, _ in timed_results if k == "commit")
    n_w = sum(1 for k, _ in timed_results if k == "open")
    ok_c = n_c == steps and all((pt == want_c).all() for k, pt in timed_results if k == "commit")
    ok_w = n_w == steps and all((pt == want_w).all() for k, pt in timed_results if k == "open")
    parity = {"commitments_checked": n_c, "proofs_checked": n_w, "commit_ok": bool(ok_c), "open_ok": bool(ok_w),
              "device_poly_eval_ok": ok_eval,
              "method": "true SRS beta^i g built on the device; every commitment / proof of the timed region == p(beta) g / "
                        "((p(beta) - p(z)) / (beta - z)) g, with p(beta), p(z) from the CPU oracle's Horner over every rank's "
                        "shard and the scalar multiplication of g by the oracle (the verifier's pairing equation, "
                        "kzg10/mod.rs:314-333, with the trapdoor known)"}

    # ---- the same steps with the coefficients handed over as HOST memory (what the Rust shim holds):
    # one pinned H2D copy per polynomial (commit and open share it), triple-buffered so that the copy of
    # step k+1 overlaps the MSMs of step k.  Reported beside `value`, never as `value`.
    h2d = None
    host = None
    if with_h2d and world == 1:
        host = coeffs.cpu().pin_memory()
        bufs = [torch.empty_like(coeffs) for _ in range(3)]
        cs = torch.cuda.Stream()
        evs = [torch.cuda.Event() for _ in range(3)]

        def issue_copy(k):
            with torch.cuda.stream(cs):
                bufs[k % 3].copy_(host, non_blocking=True)
                evs[k % 3].record(cs)

        started = set()

        def step_h2d(k):
            if k not in started:
                issue_copy(k); started.add(k)
            issue_copy(k + 1); started.add(k + 1)     # its buffer was last read by step k-2, drained by now
            evs[k % 3].synchronize()
            pending.append(("commit", job.commit_async(bufs[k % 3], n)))
            if depth == 0:
                kk, f = pending.popleft(); results.append((kk, f.result()))
            pending.append(("open", job.open_async(bufs[k % 3], n)))
            while len(pending) > min(depth, 2):
                kk, f = pending.popleft(); results.append((kk, f.result()))

        dt_h, _, _, _ = timed(step_h2d, steps, warmup)
        torch.cuda.synchronize()
        h2d = {"ms_per_step": dt_h / steps * 1e3, "value": (2 * n - 1) * steps / dt_h, "unit": "pairs/s",
               "note": "coefficients start in pinned HOST memory every step: one H2D copy of the polynomial per "
                       "commit+open (32 B/coefficient), triple-buffered on its own stream so it overlaps the previous "
                       "step's MSMs (SURVEY.md 8d: scalars H2D included)"}
        del bufs

    # ---- trait-shaped: what a PolynomialCommitment::commit(&poly) / open(&poly) caller gets -- blocking calls, the
    # coefficients in (pageable) host memory at every call, nothing kept on the device in between:
    #   commit = pc_hip_msm(PC_MEM_HOST, MONTGOMERY)                             (kzg10/mod.rs:157-210)
    #   open   = pc_hip_witness_poly(host -> device) + pc_hip_msm(PC_MEM_DEVICE) (kzg10/mod.rs:287-310)
    trait = None
    if world == 1 and not args.no_trait:
        hostc = host_u64(host if host is not None else coeffs)
        zm = mont_limbs(curve, S.z)
        reps = 3 if log_degree >= 22 else 10

        def trait_step():
            c, _ = eng.srs.msm(hostc, n=n, base_offset=1, montgomery=True)
            w, _ = eng.srs.kzg_open(hostc, zm, n=n, base_offset=1)
            return c, w
        c, w = trait_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            trait_step()
        dt_t = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):                 # the commit alone
            eng.srs.msm(hostc, n=n, base_offset=1, montgomery=True)
        dt_c = (time.perf_counter() - t0) / reps
        # ... and with the device copy of the polynomial that the shim keeps between commit and open
        # (rust/poly-commit-hip/src/device.rs, device_poly): commit = upload + MSM on the copy, open = witness + MSM on it
        pbuf = ctx.malloc(n * 32)

        def cached_step():
            ctx.memcpy_h2d(pbuf, hostc[:n])
            c, _ = eng.srs.msm(pbuf, n=n, base_offset=1, montgomery=True)
            w, _ = eng.srs.kzg_open(pbuf, zm, n=n, base_offset=1)
            return c, w
        c2, w2 = cached_step()
        t0 = time.perf_counter()
        for _ in range(reps):
            cached_step()
        dt_s = (time.perf_counter() - t0) / reps
        ctx.free_dev(pbuf)
        trait = {"ms_per_commit_open": dt_t * 1e3, "commit_ms": dt_c * 1e3, "open_ms": (dt_t - dt_c) * 1e3,
                 "with_shim_polynomial_cache_ms": dt_s * 1e3, "with_shim_polynomial_cache_parity_ok": bool((c2 == want_c).all() and (w2 == want_w).all()),
                 "commit_open_per_s": 1.0 / dt_t, "value": (2 * n - 1) / dt_t, "unit": "pairs/s",
                 "parity_ok": bool((c == want_c).all() and (w == want_w).all()),
                 "note": "blocking pc_hip_msm with PC_MEM_HOST coefficients (pageable numpy memory, the H2D inside the call), then "
                         "pc_hip_kzg_open with the same host coefficients (copy + witness division + MSM in one call): the call "
                         "sequence of the trait's commit(&poly) / open(&poly) with nothing cached between them; from 2^21 "
                         "coefficients on both calls run as ONE MSM in parts (PC_HIP_HOST_PARTS, default weights 1,2,5,8): the PCIe copy "
                         "and the sort of a part under the accumulation of the one before, one bucket reduction and one host tail"}

    # The kernels without a second pipeline competing for the CUs: strictly serial MSMs after the timed region.
    eng.blocking = False
    eng.phases, eng.marks = [], []
    for _ in range(3):
        job.commit_async(coeffs, n).result()
    sp = np.mean(np.array(eng.phases), axis=0) if eng.phases else np.zeros(8)
    t0 = time.perf_counter()
    for _ in range(3):
        job.commit_async(coeffs, n).result()
    blocking_msm_ms = (time.perf_counter() - t0) / 3 * 1e3
    in_region = bool(acc_iv)
    if not in_region:           # --inflight 0 (blocking calls leave no per-launch marks): the serial figure stands in
        acc_union_ms = float(sp[3])

    pairs_per_step = (2 * n - 1) if world == 1 else world * (2 * n) - 1
    acc_serial_ms = float(sp[3])
    launch_pairs = ((2 * n - 1) / 2.0 if world == 1 else n - 0.5 / world) if in_region else float(n)
    bytes_per_launch = launch_pairs * PAIR_BYTES[curve]
    achieved = bytes_per_launch / (acc_union_ms * 1e-3) / 1e9 if acc_union_ms > 0 else None
    ach_serial = n * PAIR_BYTES[curve] / (acc_serial_ms * 1e-3) / 1e9 if acc_serial_ms > 0 else None
    shape = ctx.last_msm_shape()
    arith = None
    pk = madd_peak(curve) if rank == 0 else None
    if pk and acc_union_ms > 0:
        adds = launch_pairs * shape["digits_per_scalar"]
        arith = {"bound": "valu", "unit": "mixed additions/s (XYZZ += affine, 8M + 2S in Fq)",
                 "achieved": adds / (acc_union_ms * 1e-3), "peak": pk["madd_per_s"],
                 "frac": adds / (acc_union_ms * 1e-3) / pk["madd_per_s"], "peak_source": pk["source"],
                 "additions_per_launch": adds, "window_bits": shape["window_bits"],
                 "digits_per_scalar": shape["digits_per_scalar"], "buckets": shape["buckets"],
                 "note": "reported beside the prescribed HBM roofline: the kernel is modular arithmetic on the VALU; peak = a "
                         "memory-free loop of the same addition on this GPU (the builder's own micro-benchmark: a "
                         "self-referential bound; against the bare v_mad_u64_u32 issue rate the kernel is at ~53 %)"}
    res = {
        "log_degree": log_degree, "steps": steps, "warmup": warmup, "dt": dt, "pairs_per_step": pairs_per_step,
        "value": pairs_per_step * steps / dt, "ms_per_step": dt / steps * 1e3,
        "per_rank_ms_per_step": [x / steps * 1e3 for x in per_rank_dt],
        "commit_open_per_s": steps / dt,
        "value_h2d_inclusive": h2d, "trait_shaped": trait, "parity": parity,
        "srs_gen_ms": S.srs_gen_ms, "srs_window_table_build_ms": eng.precompute_ms,
        "exchange_host_ms": ({k: (v / max(1, job.exchange_ms["calls"]) if k != "calls" else v) for k, v in job.exchange_ms.items()}
                             if dist is not None else None),
        "blocking_msm_ms": blocking_msm_ms,
        "msm_phase_ms": {k: float(v) for k, v in zip(
            ["digits_hist", "scan", "scatter_fine_sort", "accumulate", "seg_reduce", "bucket_reduce"], sp[:6])},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                     "traffic": pmc_traffic(f"{curve}:2^{log_degree}:{'table' if args.precompute else 'table-free'}") if world == 1 else None,
                     "traffic_source": f"{pmc_source()}: PMC FETCH_SIZE (doubled per the gfx950 note of the guide) + WRITE_SIZE per launch, "
                                       "separate rocprofv3 --pmc passes of this workload on an earlier box (tools/gpu_full_run.sh) -- the one figure of this block that is NOT "
                                       "measured in this run (null when that size / table mode was not profiled); undoubled FETCH_SIZE is about half: "
                                       "for this kernel's 16-byte gathers the raw figure is the plausible one",
                     "kernel": "pc::k_accumulate (bucket accumulation), launched twice per step (commit MSM, open MSM)",
                     "kernel_ms": acc_union_ms,
                     "kernel_ms_definition": ("hipEvent marks on the MSM pipelines' own streams INSIDE the timed region "
                                              "(pc_hip_last_msm_marks_ms): union of the [start, end] intervals of all accumulate launches / "
                                              "launches -- consecutive launches of different pipelines overlap, the union is what the region "
                                              "spent per launch (2 x kernel_ms <= ms_per_step by construction)") if in_region else
                                             "the `serial` figure (this run's steps are blocking calls, which leave no per-launch marks)",
                     "launches": len(acc_iv),
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "arithmetic": arith,
                     "bracket_ms_mean": acc_bracket_ms,
                     "serial": {"kernel_ms": acc_serial_ms, "achieved": ach_serial,
                                "frac": ach_serial / HBM_PEAK_GBPS if ach_serial else None,
                                "algorithmic_bytes_per_launch": n * PAIR_BYTES[curve],
                                "note": "3 blocking commit MSMs after the timed region (no second pipeline sharing the SIMDs): the "
                                        "duration rocprofv3's per-kernel average of an --inflight 0 run reproduces"}},
    }
    S.free()
    return res


# ------------------------------------------------------------------------------------------------------------
# blocking latency by size, CPU port beside it (configs[0] = 2^12; the crossover SURVEY 8(b) asks for)
# ------------------------------------------------------------------------------------------------------------
def latency_sweep(ctx, curve, logs, cpu_max_log):
    import torch
    import oracle_lib as O
    g = O.gen_bases(curve, 1)[0]
    p = fr_modulus(curve)
    beta, z = seed_fr(curve, 0xBE7A24), seed_fr(curve, 0x2EE7)
    zm = mont_limbs(curve, z)
    nmax = (1 << max(logs)) + 1
    pts = true_srs_points(ctx, curve, g, beta, 0, nmax)
    cores = effective_cores()
    rows = {}
    crossover = None
    ctx.set_timing(False)      # (blocking calls as a caller makes them: no per-phase event marks, which also keep small calls off their launch graphs)
    for lg in logs:
        n = (1 << lg) + 1
        srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)         # the key trimmed to this degree (MarlinKZG10::trim)
        t0 = time.perf_counter()
        srs.precompute()
        tbl_ms = (time.perf_counter() - t0) * 1e3
        co = rand_fr_device(0x5EED1000 + lg, n)
        hostc = host_u64(co)

        def gpu_step():
            c, _ = srs.msm(hostc, n=n, montgomery=True)
            w, _ = srs.kzg_open(hostc, zm, n=n)
            return c, w
        for _ in range(3):
            c, w = gpu_step()
        reps = 3 if lg >= 22 else 10 if lg >= 18 else 30
        t0 = time.perf_counter()
        for _ in range(reps):
            gpu_step()
        gpu_ms = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            srs.msm(hostc, n=n, montgomery=True)
        gpu_commit_ms = (time.perf_counter() - t0) / reps * 1e3
        row = {"gpu_commit_open_ms": gpu_ms, "gpu_commit_ms": gpu_commit_ms, "srs_window_table_build_ms": tbl_ms}
        # closed-form parity of this size's commitment and proof
        pb = from_mont_limbs(curve, O.poly_eval(curve, hostc, mont_limbs(curve, beta)))
        pz = from_mont_limbs(curve, O.poly_eval(curve, hostc, zm))
        row["parity_ok"] = bool((c == oracle_scalar_mul(curve, g, pb)).all() and
                                (w == oracle_scalar_mul(curve, g, (pb - pz) * pow(beta - z, -1, p) % p)).all())
        if lg <= cpu_max_log:
            bh = srs.read(0, n)
            creps = 5 if lg <= 14 else 2
            best = None
            for threads in ((1, cores) if lg <= 14 else (cores,)):        # small MSMs: one thread can beat the fork/join of all cores
                t0 = time.perf_counter()
                for _ in range(creps):
                    O.kzg_commit(curve, bh, hostc, threads)
                    O.kzg_open(curve, bh, hostc, zm, threads)
                dtc = (time.perf_counter() - t0) / creps * 1e3
                best = dtc if best is None else min(best, dtc)
            row["cpu_port_commit_open_ms"] = best
            row["gpu_over_cpu"] = best / gpu_ms
            if best < gpu_ms:
                crossover = lg
        rows[f"2^{lg}"] = row
        srs.free()
        del co
    del pts
    torch.cuda.empty_cache()
    ctx.set_timing(True)
    return {"curve": curve, "rows": rows,
            "cpu_faster_up_to_log_degree": crossover,
            "note": "blocking trait-shaped commit+open (host coefficients, key trimmed to the degree, window table built at trim) "
                    f"against the CPU port (oracle kzg_commit + kzg_open, best of 1 and {cores} threads up to 2^14, {cores} above); "
                    "cpu_faster_up_to_log_degree = largest measured size at which the CPU port wins (null: the GPU wins at every "
                    "measured size) -- the threshold below which the shim keeps ark-ec's msm_bigint"}


# ------------------------------------------------------------------------------------------------------------
# configs[2]: 64 x MarlinKZG10<Bn254> commits of degree 2^20, SRS sharded in contiguous chunks
# ------------------------------------------------------------------------------------------------------------
def batch_case(ctx, D, args, log_degree, polys, steps, warmup):
    import torch
    import oracle_lib as O
    from poly_commit_amd import sharded
    curve = "bn254"
    world, rank = D.world, D.rank
    p = fr_modulus(curve)
    total = (1 << log_degree) + 1
    lo, hi = sharded.ShardedBatch.chunk_range(total, rank, world)
    n = hi - lo
    g = O.gen_bases(curve, 1)[0]
    beta = seed_fr(curve, 0xBE7A25)
    eng = sharded.HipEngine(ctx, curve)
    eng.glv_table = None if args.glv_table < 0 else bool(args.glv_table)
    job = sharded.ShardedBatch(eng, curve, rank, world, D.dist)
    pts = true_srs_points(ctx, curve, g, beta, lo, n)            # this rank's REAL chunk of the one SRS
    job.load_srs_chunk(pts.data_ptr(), precompute=bool(args.precompute), n=n)
    del pts
    vec = [rand_fr_device(0x5EED0100 + j * 131 + rank * 7919, n) for j in range(polys)]
    lens = [n] * polys
    torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = job.commit_batch(vec, lens)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = job.commit_batch(vec, lens)
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    per_rank = D.gather_floats([dt])[:, 0]
    dt = float(per_rank.max())
    # parity: C_j = p_j(beta) g with p_j(beta) = sum_r beta^(lo_r) p_{j,r}(beta); every polynomial from the device's
    # evaluation kernel, a few of them also from the oracle's Horner
    bm = mont_limbs(curve, beta)
    dev_ev = np.stack([ctx.poly_eval(curve, v.data_ptr(), bm, n=n) for v in vec])
    n_oracle = min(4, polys)
    pick = sorted(set(int(round(i * (polys - 1) / max(1, n_oracle - 1))) for i in range(n_oracle)))
    ok_eval = all((O.poly_eval(curve, host_u64(vec[j]), bm) == dev_ev[j]).all() for j in pick)
    allv = D.gather_u64(dev_ev.reshape(-1)).reshape(world, polys, 4)
    ok = True
    for j in range(polys):
        pb = 0
        for r in range(world):
            lo_r, _ = sharded.ShardedBatch.chunk_range(total, r, world)
            pb = (pb + pow(beta, lo_r, p) * from_mont_limbs(curve, allv[r, j])) % p
        ok = ok and bool((out[j] == oracle_scalar_mul(curve, g, pb)).all())
    # the accumulate launches of ONE more step, by hipEvent brackets on the two pass pipelines (pc_hip_last_msm_phases_ms after a batch:
    # sums over the passes, the union of their accumulate intervals, the pass count)
    ctx.set_timing(True)
    job.commit_batch(vec, lens)
    ph = ctx.last_msm_phases_ms()
    shape = ctx.last_msm_shape()
    passes = int(round(ph[7])) if ph[7] > 0 else 0
    roof = None
    if passes and rank == 0:
        roof = msm_roofline(curve, polys * n / passes, shape["digits_per_scalar"], ph[6] / passes, passes,
                            "pc::k_accumulate<bn254> (bucket accumulation of one many-MSM pass: 8 polynomials, one bucket set each)",
                            traffic_key=f"bn254:batch{polys}x2^{log_degree}:table" if world == 1 else None,
                            extra={"kernel_ms_definition": "union of the [start, end] hipEvent intervals of the passes' accumulate launches / passes (the passes of the two "
                                                           "pipelines overlap with each other's sort and reductions)",
                                   "bracket_ms_mean": ph[3] / passes, "window_bits": shape["window_bits"], "buckets_per_polynomial": shape["buckets"] // max(1, min(8, polys)),
                                   "pass_phase_ms_sum": {k: float(v) for k, v in zip(["digits_hist", "scan", "scatter_fine_sort", "accumulate", "seg_reduce", "bucket_reduce"], ph[:6])}})
    # trait-shaped: MarlinKZG10::commit hands the polynomials over in HOST memory (marlin_pc/mod.rs:172-242): one blocking
    # pc_hip_msm_batch(PC_MEM_HOST) over pageable arrays, every pass's polynomials staged beside the other pipeline's pass
    trait = None
    if world == 1 and not args.no_trait:
        hostv = [host_u64(v) for v in vec]
        got = eng.srs.msm_batch(hostv, lens, host=True)
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            got = eng.srs.msm_batch(hostv, lens, host=True)
        dt_t = (time.perf_counter() - t0) / reps
        trait = {"ms_per_step": dt_t * 1e3, "value": polys * n / dt_t, "unit": "pairs/s",
                 "parity_ok": bool(all((got[j] == out[j]).all() for j in range(polys))),
                 "note": f"one blocking pc_hip_msm_batch over {polys} polynomials in pageable HOST memory ({polys * n * 32 / 2**20:.0f} MiB of PCIe "
                         "inside the call): the polynomials of a pass are copied to the pipeline that will run it while the other "
                         "pipeline runs the pass before"}
        del hostv
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline_batch(curve, eng.srs, vec, n, polys)
        except Exception as e:          # the baseline must never cost the measured line
            cpu = {"error": repr(e)}
    eng.srs.free()
    del vec
    torch.cuda.empty_cache()
    pairs = polys * total
    return {"workload": f"{polys} x MarlinKZG10<Bn254> commit, deg 2^{log_degree}, one SRS in {world} contiguous chunk(s) (BASELINE configs[2])",
            "value": pairs * steps / dt, "unit": "pairs/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "ms_per_commitment": dt / steps * 1e3 / polys, "per_rank_ms_per_step": [float(x) / steps * 1e3 for x in per_rank],
            "srs_window_table_build_ms": eng.precompute_ms, "roofline": roof, "trait_shaped": trait, "cpu_baseline": cpu,
            "parity": {"all_commitments_closed_form_ok": bool(ok), "oracle_horner_checked": len(pick), "oracle_horner_ok": bool(ok_eval),
                       "method": "every C_j == p_j(beta) g on the true SRS (p_j(beta) from the device's evaluation kernel, "
                                 f"{len(pick)} of them re-evaluated by the oracle's Horner; scalar multiplication by the oracle)"}}


# ------------------------------------------------------------------------------------------------------------
# configs[3]: InnerProductArgPC over Pallas, n = 2^22
# ------------------------------------------------------------------------------------------------------------
def _naf_weight_and_top(v):
    """(non-zero digits, index of the top digit) of the non-adjacent form of v >= 0 (ec.hpp NafMasks::from_scalar)."""
    w, top, i = 0, 0, 0
    while v:
        if v & 1:
            d = 2 - (v & 3)
            v -= d
            w += 1
            top = i
        v >>= 1
        i += 1
    return w, top


def glv_header_constants(curve):
    """The lattice constants of the device's GLV split, read from the generated header the library is compiled with
    (poly_commit_amd/csrc/glv_constants.h: floor reciprocals G1, G2 and the short basis (A1, B1), (A2, B2) with their signs)."""
    import re
    txt = open(os.path.join(ROOT, "poly_commit_amd", "csrc", "glv_constants.h")).read()
    body = txt[txt.index(f"struct pc_glv_{curve} {{"):]
    body = body[:body.index("\n};")]

    def arr(name):
        m = re.search(name + r"\[\d+\] = \{([^}]*)\}", body)
        return sum(int(x.strip().rstrip("ul"), 16) << (64 * i) for i, x in enumerate(m.group(1).split(",")))

    def flag(name):
        return int(re.search(name + r" = (\d)", body).group(1))
    return {k: arr(k) for k in ("G1", "G2", "A1", "B1", "A2", "B2")} | {k: flag(k) for k in ("N1_NEG", "N2_NEG", "A1_NEG", "B1_NEG", "A2_NEG", "B2_NEG")}


def glv_split(gc, k):
    """k = k1 + k2 lambda (mod r) exactly as csrc/glv.hpp glv_decompose computes it (truncated quotients)."""
    sg = lambda v, neg: -v if neg else v      # noqa: E731
    c1, c2 = sg((gc["G1"] * k) >> 384, gc["N1_NEG"]), sg((gc["G2"] * k) >> 384, gc["N2_NEG"])
    a1, b1, a2, b2 = sg(gc["A1"], gc["A1_NEG"]), sg(gc["B1"], gc["B1_NEG"]), sg(gc["A2"], gc["A2_NEG"]), sg(gc["B2"], gc["B2_NEG"])
    return k - c1 * a1 - c2 * a2, -c1 * b1 - c2 * b2


def _wnaf_weight_and_top(v, w):
    """Non-zero digits and the position of the highest one in the width-w NAF of v (csrc/glv.hpp wnaf_digits)."""
    weight, top, bit = 0, 0, 0
    while v:
        if v & 1:
            d = v & ((1 << w) - 1)
            if d >= 1 << (w - 1):
                d -= 1 << w
            v -= d
            weight, top = weight + 1, bit
        v >>= 1
        bit += 1
    return weight, top


def ipa_open_roofline(curve, log_n, challenges_mont, fold_rounds, per_round_ms, breakdown_ms, open_ms, fold_kinds=None, naf_w=2):
    """Places InnerProductArgPC::open (ipa_pc/mod.rs:664-711) against ceilings.  Its largest part is the key fold k_l += u k_r: the first
    fold from the committer key's fold table (EcFoldTableBody: one XYZZ mixed addition per non-zero NAF digit of the GLV halves of u,
    plus one base-field product for the phi half), the next ones by the shared GLV ladder (EcFoldGlvBody: Jacobian doublings and
    Jacobian += affine).  Each fold's ceiling = elements x (operations / the memory-free loop rate of that operation, tools/microbench);
    the blocking wall time of the fold call is set against it.  The rounds on the fixed key (n <= 2^16) are latency-bound: their
    floor is stated, not priced."""
    try:
        gc = glv_header_constants(curve)
    except Exception as e:
        return {"error": "GLV constants unavailable: " + repr(e)}
    madd = madd_peak(curve)["madd_per_s"]
    jd, src = micro_rate("jac_dbl", curve)
    ja, _ = micro_rate("jac_madd", curve)
    fm, _ = micro_rate("fmul", f"{curve}_fq")
    folds, tot_ms, tot_ceiling, madd_eq = [], 0.0, 0.0, 0.0
    r_mod = fr_modulus(curve)
    kinds = fold_kinds or (["table1"] + ["ladder"] * (len(fold_rounds) - 1))
    for i, (h, ms) in enumerate(fold_rounds):
        u = from_mont_limbs(curve, challenges_mont[i])
        kind = kinds[i] if i < len(kinds) else "ladder"
        if kind == "deferred":          # two-level table: round 1 leaves the key alone, round 2 folds twice in one step
            continue
        if kind in ("table1", "table2"):
            # one XYZZ mixed addition per non-zero width-w NAF digit of the GLV halves of every term's scalar, one base-field product per
            # digit of the phi half; table2: three terms (u2, u1, u1 u2) over the quarters of the committer key
            scal = [u] if kind == "table1" else [u, from_mont_limbs(curve, challenges_mont[i - 1]), u * from_mont_limbs(curve, challenges_mont[i - 1]) % r_mod]
            w1 = w2 = 0
            for sc in scal:
                k1, k2 = glv_split(gc, sc)
                w1 += _wnaf_weight_and_top(abs(k1), naf_w)[0]
                w2 += _wnaf_weight_and_top(abs(k2), naf_w)[0]
            ceil_s = h * ((w1 + w2) / madd + (w2 / fm if fm else 0.0))
            ops = {"xyzz_mixed_additions": w1 + w2, "phi_products": w2, "terms": len(scal), "naf_width": naf_w}
            kern = f"pc::EcFoldTableBody<{curve}> + XyzzBatchAffineBody"
        else:
            k1, k2 = glv_split(gc, u)
            (w1, t1), (w2, t2) = _naf_weight_and_top(abs(k1)), _naf_weight_and_top(abs(k2))
            nd, na = max(t1, t2), w1 + w2 + 1
            ceil_s = h * (nd / jd + na / ja) if jd and ja else None
            ops = {"jacobian_doublings": nd, "jacobian_mixed_additions": na}
            kern = f"pc::EcFoldGlvBody<{curve}> + JacBatchAffineBody"
        folds.append({"round": i + 1, "elements": h, "kind": kind, "kernel": kern, "ops_per_element": ops, "wall_ms": ms,
                      "ceiling_ms": ceil_s * 1e3 if ceil_s else None, "frac": ceil_s * 1e3 / ms if ceil_s and ms else None})
        if ceil_s:
            tot_ms += ms
            tot_ceiling += ceil_s * 1e3
            madd_eq += ceil_s * madd
    n_fold = len(fold_rounds)
    tail = per_round_ms[n_fold + 1:] if len(per_round_ms) > n_fold + 1 else []
    return {"bound": "valu", "unit": "ms of memory-free arithmetic per ms measured",
            "kernels": "the key folds of the first rounds (" + ", ".join(sorted({f['kernel'] for f in folds})) + ")",
            "achieved": tot_ms, "peak": tot_ceiling, "frac": tot_ceiling / tot_ms if tot_ms else None,
            "definition": "sum over the folds of [elements x (operations per element / memory-free loop rate of that operation)] / sum of the folds' "
                          "blocking wall times (kernels + batched normalisation + launch); operations from the NAF of the GLV halves of each round challenge",
            "madd_equivalents_per_s": madd_eq / (tot_ms * 1e-3) if tot_ms else None, "madd_loop_per_s": madd,
            "loop_rates_per_s": {"xyzz_madd": madd, "jacobian_dbl": jd, "jacobian_madd": ja, "fq_mul": fm}, "loop_source": src,
            "folds": folds, "fold_share_of_open": tot_ms / open_ms if open_ms else None,
            "msm_wait_ms": breakdown_ms.get("msm_wait"), "host_point_mul_ms": breakdown_ms.get("host_point_mul"),
            "fixed_key_rounds": {"count": len(tail), "ms_each_mean": float(np.mean(tail)) if tail else None, "ms_total": float(np.sum(tail)) if tail else None,
                                 "note": "this breakdown is of the opening driven round by round: rounds with n <= 2^16 keep the key and run two 2^16-pair MSMs over "
                                         "per-base factors: each is one scalar kernel, two pipelined MSM launches (sort, accumulate, three reduction levels, "
                                         "64-byte download), two host point multiplications and the Horner tails -- a launch / latency floor of 0.8-0.9 ms per "
                                         "round that no kernel rate changes.  Inside pc_hip_ipa_open_rounds (open_ms) the fixed key has 2^17 points and its "
                                         "own window table, refilled per opening: 0.57 ms per round"}}


def ipa_case(ctx, log_n, reps, with_cpu=True):
    import torch
    import oracle_lib as O
    from poly_commit_amd import ipa
    curve = "pallas"
    n = 1 << log_n
    p = fr_modulus(curve)
    g = O.gen_bases(curve, 1)[0]
    t0 = time.perf_counter()
    # hash-free synthetic generators a^i g (the reference derives them by try-and-increment,
    # ipa_pc/mod.rs:302-325): n for the committer key + h' as one more
    a = seed_fr(curve, 0xA11CE5)
    pts = true_srs_points(ctx, curve, g, a, 0, n + 1)
    key_gen_ms = (time.perf_counter() - t0) * 1e3
    srs = ctx.upload_srs(curve, pts.data_ptr(), n=n)
    h_prime = host_u64(pts[n])
    del pts
    # once per committer key, like a KZG key's window table: the window table (the commit MSM and the first round's MSMs) and the
    # fold table of the upper half (the first key fold of every opening)
    t0 = time.perf_counter()
    srs.precompute()
    srs.precompute_fold()
    key_tables_ms = (time.perf_counter() - t0) * 1e3
    fold_table_form, fold_table_bytes = srs.fold_table_info(), srs.bytes_resident()["fold_table"]
    cdev = rand_fr_device(0xA11CE, n)
    point = mont_limbs(curve, seed_fr(curve, 0xB0B))
    ch = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC4A1, log_n))
    torch.cuda.synchronize()
    for _ in range(3):      # every pipeline of the SRS exists (streams + workspace are created on first use)
        comm, _ = srs.msm(cdev.data_ptr(), n=n, montgomery=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        comm, _ = srs.msm(cdev.data_ptr(), n=n, montgomery=True)
    t_commit = (time.perf_counter() - t0) / reps
    ph_c, shape_c = ctx.last_msm_phases_ms(), ctx.last_msm_shape()
    best, tm_best, proof = None, None, None
    # (the openings run without the per-phase hipEvent marks the commit above reported: with marks on, the library keeps the small MSMs of
    # the late rounds on plain launches instead of replaying their captured graphs -- a caller does not run with marks on)
    ctx.set_timing(False)
    for _ in range(reps):
        it = iter(range(log_n))
        tm = {}
        work = cdev.clone()                 # the folds act in place on the coefficient vector
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proof = ipa.ipa_open_rounds(ctx, curve, srs, work, n, point, h_prime, lambda L, R_: ch[next(it)], timings=tm)      # pc_hip_ipa_open_rounds
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, tm_best = dt, tm
    # the same opening driven round by round through the single entry points (what the line's open_breakdown_ms splits; not the timed figure)
    it = iter(range(log_n))
    tm_py = {}
    work = cdev.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    proof_py = ipa.ipa_open_rounds(ctx, curve, srs, work, n, point, h_prime, lambda L, R_: ch[next(it)], timings=tm_py, python_loop=True)
    py_ms = (time.perf_counter() - t0) * 1e3
    loops_agree = all(bool((np.asarray(a) == np.asarray(b)).all()) for a, b in zip(proof, proof_py))
    for k in ("per_round_ms", "ec_fold_per_round_ms", "ec_fold_kind"):
        tm_py.pop(k, None)
    tm_best = dict(tm_best, **{k: v for k, v in tm_py.items() if k != "ec_fold"})
    ctx.set_timing(True)
    per_round = tm_best.pop("per_round_ms", [])
    fold_rounds = tm_best.pop("ec_fold_per_round_ms", [])
    fold_kinds = tm_best.pop("ec_fold_kind", None)
    roof_open = ipa_open_roofline(curve, log_n, ch, fold_rounds, per_round, tm_best, best * 1e3, fold_kinds, fold_table_form[1] or 2)
    roof_open["fold_table"] = {"levels": fold_table_form[0], "naf_width": fold_table_form[1], "bytes": fold_table_bytes}
    cpu = None
    if with_cpu:
        try:
            cpu = cpu_baseline_ipa(curve, log_n)
        except Exception as e:
            cpu = {"error": repr(e)}
    host = host_u64(cdev)
    pa = from_mont_limbs(curve, O.poly_eval(curve, host, mont_limbs(curve, a)))
    ok_commit = bool((comm == oracle_scalar_mul(curve, g, pa)).all())
    # final_comm_key = sum_j (prod_i u_i^{bit_{log_n-1-i}(j)}) a^j g = prod_i (1 + u_i a^(2^(log_n-1-i))) g
    us = [from_mont_limbs(curve, ch[i]) for i in range(log_n)]
    fk = 1
    for i, u in enumerate(us):
        fk = fk * (1 + u * pow(a, 1 << (log_n - 1 - i), p)) % p
    final_key = np.asarray(proof[2]).reshape(-1)
    ok_key = bool((final_key == oracle_scalar_mul(curve, g, fk)).all())
    srs.free()
    del cdev
    torch.cuda.empty_cache()
    return {"workload": f"InnerProductArgPC over Pallas, n = 2^{log_n}: cm_commit MSM + open's {log_n} halving rounds (BASELINE configs[3])",
            "commit_ms": t_commit * 1e3, "open_ms": best * 1e3, "open_ms_is": "one pc_hip_ipa_open_rounds call (the loop inside the library), best of the openings",
            "open_round_by_round_ms": py_ms, "open_round_by_round_is": "the same opening driven through the single entry points from Python (one run; the breakdown's source)",
            "commit_pairs_per_s": n / t_commit, "open_msm_pairs_per_s": 2 * n / best,
            "open_coeffs_per_s": n / best, "log_n": log_n,
            "open_breakdown_ms": {k: round(v, 2) for k, v in tm_best.items()}, "per_round_ms": [round(x, 2) for x in per_round],
            "roofline": msm_roofline(curve, n, shape_c["digits_per_scalar"], float(ph_c[3]), 1,
                                     "pc::k_accumulate<pallas> (bucket accumulation of the cm_commit MSM; the opening's 2 x 22 round MSMs run the same kernel "
                                     "on n/2 .. 1 pairs)", traffic_key=f"pallas:2^{log_n}:table",
                                     extra={"kernel_ms_definition": "hipEvent bracket of the accumulate launch of the last blocking commit MSM",
                                            "commit_msm_phase_ms": {k: float(v) for k, v in zip(["digits_hist", "scan", "scatter_fine_sort", "accumulate", "seg_reduce", "bucket_reduce"], ph_c[:6])},
                                            "window_bits": shape_c["window_bits"], "buckets": shape_c["buckets"]}),
            "roofline_open": roof_open, "cpu_baseline": cpu,
            "key_gen_ms": key_gen_ms, "key_tables_build_ms": key_tables_ms,
            "key_tables_note": "window table + fold table of the committer key (two levels: 131 x 2^(w-2) rows of the upper three quarters), built once per key outside the timing",
            "parity": {"commit_ok": ok_commit, "final_comm_key_ok": ok_key, "library_loop_equals_round_by_round_ok": loops_agree,
                       "method": "generators a^i g built on the device: commitment == p(a) g (oracle Horner + scalar multiplication); "
                                 "final_comm_key == prod_i (1 + u_i a^(2^(log n - 1 - i))) g for the supplied round challenges u_i "
                                 "(ipa_pc/mod.rs:699-701 folded in closed form); the whole Proof is compared with the oracle at this "
                                 "size in tests/test_baseline_sizes_gpu.py"}}


# ------------------------------------------------------------------------------------------------------------
# configs[4]: Ligero over BLS12-381 Fr, 2^24 coefficients
# ------------------------------------------------------------------------------------------------------------
def ligero_case(ctx, D, curve, log_len, steps, warmup, with_cpu=True, with_trait=True):
    import torch
    import oracle_lib as O
    from poly_commit_amd import sharded
    world, rank = D.world, D.rank
    poly_len = 1 << log_len
    n_rows, n_cols, _ = O.ligero_dims(FR_BITS[curve], poly_len, 4)
    log_n = (n_cols * 4 - 1).bit_length()
    N = 1 << log_n
    shard = sharded.ShardedRows(sharded.HipEngine(ctx, curve), rank, world)   # rows are independent: shard by rows, no collective
    r_lo, r_hi = shard.row_range(n_rows)
    rows = r_hi - r_lo
    x = rand_fr_device(0x5EED0500 + rank, rows * n_cols)
    y = torch.empty((rows * N, 4), dtype=torch.int64, device="cuda")
    ph = np.zeros(2)
    ctx.set_timing(True)
    for _ in range(warmup):
        shard.encode(x, rows, n_cols, log_n, y)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        shard.encode(x, rows, n_cols, log_n, y)
        ph += np.array(ctx.last_ntt_phases_ms())
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    per_rank = D.gather_floats([dt])[:, 0]
    dt = float(per_rank.max())
    ph /= steps
    # the next steps of LinearCodePCS::commit on the resident matrix (linear_codes/mod.rs:256-277)
    hash_ms = merkle_ms = None
    if world == 1:
        leaves = torch.empty((N, 32), dtype=torch.uint8, device="cuda")
        nodes = torch.empty((N - 1, 32), dtype=torch.uint8, device="cuda")
        ctx.column_hash(curve, y.data_ptr(), "blake2s", out=leaves.data_ptr(), rows=rows, n_cols=N)
        ctx.merkle_tree(leaves.data_ptr(), "sha256", True, out=nodes.data_ptr(), n_leaves=N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.column_hash(curve, y.data_ptr(), "blake2s", out=leaves.data_ptr(), rows=rows, n_cols=N)
        torch.cuda.synchronize()
        hash_ms = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.merkle_tree(leaves.data_ptr(), "sha256", True, out=nodes.data_ptr(), n_leaves=N)
        torch.cuda.synchronize()
        merkle_ms = (time.perf_counter() - t0) / 3 * 1e3
    # trait-shaped (world == 1): LinearCodePCS::commit keeps mat, ext_mat and the leaves in its commitment state on the HOST
    # (linear_codes/mod.rs:248-297): one pc_hip_ligero_commit with the coefficient matrix coming from, and the encoded matrix going to,
    # pageable host memory -- 0.5 GiB up, 2 GiB down
    trait = None
    if world == 1 and with_trait and rows * N * 32 <= (4 << 30):
        hx = host_u64(x).reshape(rows, n_cols, 4)
        hext = np.empty((rows, N, 4), dtype=np.uint64)
        for _ in range(2):      # the first call faults the pages of hext in, the second still pays for the pins the first one left behind
            nodes_t, _ = ctx.ligero_commit(curve, hx, log_n, ext_out=hext)
        calls = []
        for _ in range(6):
            t0 = time.perf_counter()
            nodes_t, _ = ctx.ligero_commit(curve, hx, log_n, ext_out=hext)
            calls.append(time.perf_counter() - t0)
        dt_t = sorted(calls)[len(calls) // 2]
        ok_t = bool((hext[rows - 1] == host_u64(y.view(rows, N, 4)[rows - 1])).all() and (hext[0, :64] == host_u64(y.view(rows, N, 4)[0, :64])).all())
        trait = {"ms_per_commit": dt_t * 1e3, "ms_per_commit_is": "median of six calls after two untimed ones", "ms_calls": [round(c * 1e3, 2) for c in calls],
                 "ms_per_commit_mean": sum(calls) / len(calls) * 1e3, "value": rows * n_cols / dt_t, "unit": "coeffs/s", "parity_ok": ok_t,
                 "pcie_bytes": int(rows * (n_cols + N) * 32),
                 "note": "one blocking pc_hip_ligero_commit: coefficient matrix from pageable host memory, encoded matrix + leaves + tree nodes back to it "
                         "(what LinCodePCCommitmentState holds), in slabs of rows: the copy in, the NTT, the chained column digests of one slab run "
                         "beside the previous slab's way out -- the call is the encoded matrix's PCIe time plus a few ms"}
        del hx, hext
    # N > 1: a column's digest needs the rows of every rank -- the digests' chaining states travel from rank to rank
    # (ShardedRows.commit / pc_hip_column_hash_part: 48 bytes per column and hop instead of a transpose of the matrix), the last
    # rank builds the tree and broadcasts the root
    chain = None
    if world > 1:
        root, _ = shard.commit(y, n_rows, N, D.dist, "blake2s", "sha256", blocks=8)
        torch.cuda.synchronize()
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            root2, _ = shard.commit(y, n_rows, N, D.dist, "blake2s", "sha256", blocks=8)
        torch.cuda.synchronize()
        D.barrier()
        chain_ms = (time.perf_counter() - t0) / 3 * 1e3
        roots = D.gather_u64(np.frombuffer(bytes(root) , dtype=np.uint64))
        same = bool((roots == roots[0]).all()) and bytes(root2) == bytes(root)
        ok_root = None
        if n_rows % (2 * world) == 0 and n_rows * N * 32 <= (64 << 20):      # small enough to re-hash the whole matrix on the host
            import hashlib
            import pyref as R
            full = D.gather_u64(host_u64(y).reshape(-1))
            if rank == 0:
                can = np.ascontiguousarray(O.f_from_mont(curve, 1, full.reshape(-1, 4))).view(np.uint8).reshape(n_rows, N, 32)
                pre = int(n_rows).to_bytes(8, "little")
                leaves = [hashlib.blake2s(pre + can[:, j, :].tobytes()).digest() for j in range(N)]
                ok_root = R.merkle_tree(leaves, "sha256", True)[0] == bytes(root)
        chain = {"column_digests_and_tree_ms": chain_ms, "root_equal_on_all_ranks": same, "root_vs_host_rehash_ok": ok_root,
                 "bytes_per_hop": N * 48,
                 "note": "chained column digests: rank r absorbs its rows into the states rank r - 1 left, 8 column ranges in flight; "
                         "root re-hashed on the host (hashlib) when the whole matrix is at most 64 MB"}
    # parity (test_reed_solomon's property, linear_codes/utils.rs:303-331): out[r][j] == row_r(omega^j), a few rows and
    # columns by the oracle's Horner; one whole row against the oracle's NTT
    w = from_mont_limbs(curve, O.root_of_unity(curve, log_n))
    p = fr_modulus(curve)
    ok = True
    ym = y.view(rows, N, 4)
    xm = x.view(rows, n_cols, 4)
    for r, j in ((0, 0), (0, 1), (rows // 2, N // 3), (rows - 1, N - 1), (rows - 1, N // 2 + 5)):
        want = O.poly_eval(curve, host_u64(xm[r]), mont_limbs(curve, pow(w, j, p)))
        ok = ok and bool((host_u64(ym[r, j]) == want).all())
    row = host_u64(xm[rows - 1]).reshape(1, n_cols, 4)
    ok_row = bool((O.ntt_batch(curve, row, log_n, os.cpu_count() or 8)[0] == host_u64(ym[rows - 1])).all())
    del x, y
    torch.cuda.empty_cache()
    alg_bytes = rows * (n_cols + N) * 32
    kern_ms = float(ph.sum())
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
    # the bound that holds (the kernels are VALU-bound on the 255-bit Montgomery product): twiddle products per second against the
    # memory-free butterfly loops of tools/microbench for this scalar field
    prod_row, bfly_row, nominal_row = ntt_products(log_n, n_cols)
    fr = f"{curve}_fr"
    b2, src2 = micro_rate("butterfly", fr) if rank == 0 else (None, None)
    b4, _ = micro_rate("butterfly4", fr) if rank == 0 else (None, None)
    fm, _ = micro_rate("fmul", fr) if rank == 0 else (None, None)
    ach_p = rows * prod_row / (kern_ms * 1e-3) if kern_ms > 0 else None
    peak = max([v for v in (b2, b4) if v] or [0]) or None
    arith = {"bound": "valu", "unit": "twiddle products/s (one 255-bit Montgomery product + one modular add + one modular sub = one butterfly)",
             "achieved": ach_p, "peak": peak, "frac": ach_p / peak if ach_p and peak else None,
             "peak_definition": "the faster of tools/microbench's two memory-free butterfly loops for this field: the radix-2 butterfly (x + w y, x - w y) and the "
                                "radix-4 register group of two stages on four elements (what lds_ntt_stages runs per LDS round trip)",
             "peak_source": src2, "butterfly_loop_per_s": b2, "radix4_group_loop_per_s": b4, "fmul_per_s": fm,
             "products_per_launch": rows * prod_row, "butterflies_executed_per_launch": rows * bfly_row,
             "nominal_butterflies_per_launch": rows * nominal_row,
             "count_definition": "products the two passes execute per batch: (N/2) log2 N butterflies minus the stages pass A skips over the zero padding "
                                 "(rho^-1 = 4: two of seventeen) and the products with twiddle 1, plus the omega_N^(i2 j1) products between the passes "
                                 "(bench.ntt_products mirrors csrc/ntt.hpp)"}
    cpu = None
    if rank == 0 and with_cpu:
        try:
            cpu = cpu_baseline_ntt(curve, n_rows, n_cols, log_n)
        except Exception as e:
            cpu = {"error": repr(e)}
    return {"workload": f"LigeroPCS over {curve} Fr, 2^{log_len} coefficients: {n_rows} x {n_cols} matrix, {n_rows} forward NTTs of size 2^{log_n} "
                        f"(BASELINE configs[4]); rows sharded over {world} GPU(s), no collective",
            "value": world * rows * n_cols * steps / dt, "unit": "coeffs/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "per_rank_ms_per_step": [float(v) / steps * 1e3 for v in per_rank],
            "ntt_phase_ms": {"pass_a": float(ph[0]), "pass_b": float(ph[1])},
            "column_hash_blake2s_ms": hash_ms, "merkle_tree_sha256_ms": merkle_ms, "sharded_commit": chain, "trait_shaped": trait,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS if achieved else None,
                         "traffic": checked_traffic(pmc_traffic(f"ntt:{curve}:2^{log_len}", "ntt_hbm_bytes_per_batch"), alg_bytes) if world == 1 else None,
                         "traffic_source": f"{pmc_source()} (separate rocprofv3 --pmc passes of this workload; not measured in this run)",
                         "kernel": "pc::k_ntt_pass_a + pc::k_ntt_pass_b (one batched NTT = both), hipEvent brackets on the context's stream inside the timed region",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes, "arithmetic": arith},
            "cpu_baseline": cpu,
            "parity": {"horner_spot_checks_ok": bool(ok), "one_row_vs_oracle_ntt_ok": ok_row,
                       "method": "out[r][j] == row_r(omega^j) by the oracle's Horner at 5 (row, column) spots (test_reed_solomon's property, "
                                 "linear_codes/utils.rs:303-331) and one whole row against the oracle's NTT"}}


# ------------------------------------------------------------------------------------------------------------
# --mode group: ONE process drives N devices through pc_hip_group_* (the form a Rust prover holding one CommitterKey binds)
# ------------------------------------------------------------------------------------------------------------
def group_case(args, curve, log_degree, steps, warmup):
    import collections
    import torch
    import oracle_lib as O
    import poly_commit_amd as pc
    devs = [int(x) for x in os.environ["PC_BENCH_DEVICES"].split(",")] if os.environ.get("PC_BENCH_DEVICES") else list(range(args.gpus))
    N = len(devs)
    n = N * (1 << log_degree) + 1
    p = fr_modulus(curve)
    g_xy = O.gen_bases(curve, 1)[0]
    beta, z = seed_fr(curve, 0xBE7A24), seed_fr(curve, 0x2EE7)
    zm = mont_limbs(curve, z)
    torch.cuda.set_device(devs[0])
    ctx0 = pc.Context(devs[0])
    t0 = time.perf_counter()
    pts = true_srs_points(ctx0, curve, g_xy, beta, 0, n)
    powers = host_u64(pts)                       # the CommitterKey's Vec<G1Affine>, packed: what the prover hands to `trim`
    del pts
    torch.cuda.empty_cache()
    srs_gen_ms = (time.perf_counter() - t0) * 1e3
    grp = pc.Group(devs)
    t0 = time.perf_counter()
    srs = grp.upload_srs(curve, powers, precompute=bool(args.precompute))
    upload_ms = (time.perf_counter() - t0) * 1e3
    del powers
    coeffs_dev0 = rand_fr_device(0x5EED0001, n)
    host = host_u64(coeffs_dev0)
    per = (n + N - 1) // N
    shards = []
    if args.group_coeffs == "pinned":
        host_pin = torch.from_numpy(host.view(np.int64)).pin_memory()
        host = host_pin.numpy().view(np.uint64)
    if args.group_coeffs == "device":
        for d, dev in enumerate(devs):
            with torch.cuda.device(dev):
                shards.append(torch.from_numpy(np.ascontiguousarray(host[min(n, d * per):min(n, (d + 1) * per)]).view(np.int64)).cuda(dev))
        for dev in set(devs):
            torch.cuda.synchronize(dev)
    arg = [t.data_ptr() for t in shards] if shards else host
    depth = max(1, min(2, args.inflight if args.inflight > 0 else 1))
    pending, results = collections.deque(), []

    def run(k):
        for _ in range(k):
            pending.append(srs.commit_open_async(arg, zm, n=n, want_value=args.group_value))
            while len(pending) > depth - (1 if args.inflight == 0 else 0):
                results.append(pending.popleft().wait())
        while pending:
            results.append(pending.popleft().wait())
    run(warmup)
    results.clear()
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    pb = from_mont_limbs(curve, O.poly_eval(curve, host, mont_limbs(curve, beta)))
    pz = from_mont_limbs(curve, O.poly_eval(curve, host, zm))
    want_c = oracle_scalar_mul(curve, g_xy, pb)
    want_w = oracle_scalar_mul(curve, g_xy, (pb - pz) * pow(beta - z, -1, p) % p)
    ok = len(results) == steps and all((c == want_c).all() and (w == want_w).all() and (not args.group_value or from_mont_limbs(curve, v) == pz)
                                       for c, w, v in results)
    srs.free()
    grp.close()
    ctx0.close()
    return {"metric": "MSM G1-scalar-pairs/sec inside KZG commit+open, ONE process driving N devices through pc_hip_group_commit_open_async",
            "value": (2 * n - 1) * steps / dt, "unit": "pairs/s", "n_gpus": N, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (381-bit Fq / 255-bit Fr modular integer)",
            "data": "synthetic", "commit_open_per_s": steps / dt,
            "config": {"workload": f"MarlinKZG10<{curve}> commit+open of ONE polynomial of {N} x 2^{log_degree} + 1 coefficients, key sharded over {N} device context(s) "
                                   f"{devs} in one process (pc_hip_group_*), jobs in flight: {depth}", "mode": "group", "coefficients": args.group_coeffs,
                       "srs_gen_ms": srs_gen_ms, "group_srs_upload_and_table_ms": upload_ms},
            "parity": {"all_steps_ok": bool(ok), "checked": len(results),
                       "method": "every commitment, proof and p(z) of the timed region against the closed forms on the true SRS (oracle Horner + scalar multiplication)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: as many as give >= 1.2 s of timed region, at least 10)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-degree", type=int, default=24,
                    help="degree of the primary workload (default 2^24: the north-star size; BASELINE.json's metric is "
                         "quoted at 2^20 and 2^24 -- the other one is reported in the `secondary` block)")
    ap.add_argument("--secondary-log-degree", type=int, default=20, help="0 = no secondary block")
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-coefficients (H2D-inclusive) leg")
    ap.add_argument("--no-trait", action="store_true", help="skip the trait-shaped leg (profile runs: its half-size launches would enter the per-kernel averages)")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=2,
                    help="commit/open results allowed in flight (0 = strictly sequential, blocking calls)")
    ap.add_argument("--workload", default="kzg", choices=["kzg", "ntt", "batch"],
                    help="what `value` measures: kzg (default, BASELINE configs[1] / north star), ntt (configs[4]) or batch (configs[2])")
    ap.add_argument("--workloads", default=None,
                    help="comma list of the extra blocks timed in the same run at N = 1 (latency,batch,ipa,ligero); "
                         "default: all for the default kzg run; 'none' disables")
    ap.add_argument("--small", action="store_true", help="scaled-down sizes for every block (tests on a shared box); says so in the line")
    ap.add_argument("--polys", type=int, default=64)
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="default: nccl (RCCL); gloo when ranks share a GPU")
    ap.add_argument("--mode", default="ranks", choices=["ranks", "group"],
                    help="ranks (default): one process per GPU over torch.distributed; group: ONE process drives the --gpus devices through "
                         "pc_hip_group_commit_open_async (persistent worker thread per device)")
    ap.add_argument("--group-value", type=int, default=0, help="--mode group: also return p(z) with every proof (one more evaluation pass per shard)")
    ap.add_argument("--group-coeffs", default="host", choices=["host", "pinned", "device"], help="--mode group: coefficients handed over as one host array (pageable, or page-locked), or as resident per-device shards")
    ap.add_argument("--glv-table", type=int, default=-1,
                    help="form of the SRS window table: 1 = GLV (half the memory: the windows of the scalars' 128-bit halves, "
                         "pc_hip_srs_precompute_ex PC_HIP_TABLE_GLV), 0 = full, -1 (default) = the library's policy (full unless HBM is tight)")
    ap.add_argument("--precompute", type=int, default=1,
                    help="1 (default): build the SRS window table in HBM once after the upload "
                         "(pc_hip_srs_precompute; part of SRS residency, outside the timed region); 0: table-free MSM")
    args = ap.parse_args()
    if args.mode == "group":
        r = group_case(args, args.curve, min(args.log_degree, 16) if args.small else args.log_degree, args.steps or 10, args.warmup)
        emit(r)
        if not r["parity"]["all_steps_ok"]:
            raise SystemExit("parity check FAILED")
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, sys.argv[1:])

    import torch  # noqa: F401
    import poly_commit_amd as pc

    D = Dist(args)
    world, rank = D.world, D.rank
    curve = args.curve
    ctx = pc.Context(D.device)
    if args.window_bits or args.chunk:
        ctx.set_msm_tuning(args.window_bits, args.chunk)
    ctx.set_timing(True)
    small = args.small
    t_start = time.perf_counter()

    if args.workload == "ntt":
        r = ligero_case(ctx, D, curve, 16 if small else 24, args.steps or (5 if small else 100), args.warmup, with_cpu=not args.no_cpu_baseline,
                        with_trait=not args.no_trait)
        if rank == 0:
            emit({"metric": "Ligero Reed-Solomon NTT input coefficients/sec (LigeroPCS over BLS12-381 Fr, 2^24 coeffs)",
                  "value": r["value"], "unit": "coeffs/s", "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
                  "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                  "dtype": "u32 limbs (255-bit Fr modular integer)", "data": "synthetic", "dist": D.info(),
                  "config": {"workload": r["workload"], "parallelism": "1 GPU" if world == 1 else f"rows sharded over {world} GPUs: no collective for the NTTs, 48 bytes per column and hop for the column digests"},
                  **{k: r[k] for k in ("per_rank_ms_per_step", "ntt_phase_ms", "column_hash_blake2s_ms", "merkle_tree_sha256_ms", "sharded_commit", "roofline", "parity")}})
        ok = r["parity"]["horner_spot_checks_ok"] and r["parity"]["one_row_vs_oracle_ntt_ok"]
        if r["sharded_commit"] is not None:
            ok = ok and r["sharded_commit"]["root_equal_on_all_ranks"] and r["sharded_commit"]["root_vs_host_rehash_ok"] is not False
        D.close()
        if not ok:
            raise SystemExit("parity check FAILED")
        return
    if args.workload == "batch":
        r = batch_case(ctx, D, args, 14 if small else (20 if args.log_degree == 24 else args.log_degree), args.polys, args.steps or 10, args.warmup)
        if rank == 0:
            emit({"metric": "MSM G1-scalar-pairs/sec, batched MarlinKZG10<Bn254> commit (64 polys, deg 2^20, SRS sharded)",
                  "value": r["value"], "unit": "pairs/s", "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
                  "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                  "dtype": "u32 limbs (254-bit modular integer)", "data": "synthetic", "dist": D.info(),
                  "config": {"workload": r["workload"], "polys_per_s": args.polys / (r["ms_per_step"] * 1e-3),
                             "parallelism": "1 GPU" if world == 1 else f"SRS sharded over {world} GPUs, all_gather of {args.polys} partial points"},
                  **{k: r[k] for k in ("per_rank_ms_per_step", "ms_per_commitment", "srs_window_table_build_ms", "roofline", "parity")}})
        ok = r["parity"]["all_commitments_closed_form_ok"] and r["parity"]["oracle_horner_ok"]
        D.close()
        if not ok:
            raise SystemExit("parity check FAILED")
        return

    log_degree = min(args.log_degree, 16) if small else args.log_degree
    sec_log = args.secondary_log_degree if not small else (12 if args.secondary_log_degree else 0)
    prim = kzg_case(ctx, D, args, curve, log_degree, args.steps, args.warmup, with_h2d=not args.no_h2d)
    log(f"primary 2^{log_degree}: {prim['ms_per_step']:.2f} ms/step, parity {prim['parity']['commit_ok']}/{prim['parity']['open_ok']} "
        f"({time.perf_counter() - t_start:.1f} s)")
    sec = None
    if world == 1 and sec_log and sec_log != log_degree:
        sec = kzg_case(ctx, D, args, curve, sec_log, None, args.warmup, with_h2d=not args.no_h2d)
        log(f"secondary 2^{sec_log}: {sec['ms_per_step']:.2f} ms/step ({time.perf_counter() - t_start:.1f} s)")

    # ---- the other BASELINE configs, timed in the same run (N = 1) ------------------------------------------
    want = args.workloads
    if want is None:
        want = "latency,batch,ipa,ligero" if world == 1 else "none"
    want = [] if want == "none" else [w.strip() for w in want.split(",") if w.strip()]
    workloads = {}
    if world == 1:
        for name in want:
            t0 = time.perf_counter()
            try:
                if name == "latency":
                    logs = (10, 12, 14) if small else (10, 12, 14, 16, 18, 20, 22, 24)
                    workloads[name] = latency_sweep(ctx, curve, logs, cpu_max_log=14 if small else 20)
                elif name == "batch":
                    workloads[name] = batch_case(ctx, D, args, 14 if small else 20, 8 if small else args.polys, 3, 1)
                elif name == "ipa":
                    workloads[name] = ipa_case(ctx, 14 if small else 22, 2 if small else 5, with_cpu=not args.no_cpu_baseline)
                elif name == "ligero":
                    workloads[name] = ligero_case(ctx, D, curve, 16 if small else 24, 5 if small else 20, 2, with_cpu=not args.no_cpu_baseline, with_trait=not args.no_trait)
                else:
                    workloads[name] = {"error": "unknown workload"}
            except Exception as e:      # one block failing must not lose the driver's line
                import traceback
                traceback.print_exc()
                workloads[name] = {"error": f"{type(e).__name__}: {e}"}
            workloads[name]["block_wall_s"] = time.perf_counter() - t0
            log(f"workload {name}: {time.perf_counter() - t0:.1f} s")

    cpu = None
    if not args.no_cpu_baseline and world == 1 and rank == 0:      # reported baseline: rank 0 at N = 1 only
        t0 = time.perf_counter()
        import oracle_lib as O
        g = O.gen_bases(curve, 1)[0]
        lgc = min(log_degree, 14) if small else log_degree
        pts = true_srs_points(ctx, curve, g, seed_fr(curve, 0xBE7A24), -1, (1 << lgc) + 1)
        srs = ctx.upload_srs(curve, pts.data_ptr(), n=(1 << lgc) + 1)
        del pts
        cpu = cpu_baseline(curve, srs, lgc, budget_s=3.0 if small else 30.0)
        srs.free()
        log(f"cpu baseline: {time.perf_counter() - t0:.1f} s")

    all_ok = prim["parity"]["commit_ok"] and prim["parity"]["open_ok"] and (sec is None or (sec["parity"]["commit_ok"] and sec["parity"]["open_ok"]))
    if rank == 0:
        def cfg(lg):
            return (f"MarlinKZG10<{curve}> commit+open, dense poly deg 2^{lg} per GPU, true SRS resident, hiding off "
                    f"(BASELINE metric sizes: 2^20 = configs[1], 2^24 = north-star target)" + (" [--small test sizes]" if small else ""))
        out = {
            "metric": "MSM G1-scalar-pairs/sec inside KZG commit+open (MarlinKZG10<Bls12_381> shape, hiding off), inputs HBM-resident "
                      "(host-input blocking calls: value_trait_shaped)",
            "value": prim["value"], "unit": "pairs/s", "n_gpus": world, "steps": prim["steps"], "warmup": prim["warmup"],
            "ms_per_step": prim["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Fq / 255-bit Fr modular integer)", "data": "synthetic",
            "dist": D.info(), "per_rank_ms_per_step": prim["per_rank_ms_per_step"],
            "parity": dict(prim["parity"], reference_pinned="no -- oracle-only: the reference holds no golden vector on this path and its "
                                                            "arithmetic crates are not buildable here (DESIGN.md section 2)"),
            "config": {"workload": cfg(log_degree), "curve": curve, "log_degree": log_degree,
                       "pairs_per_step": prim["pairs_per_step"], "inflight": max(0, args.inflight),
                       "srs_window_table": bool(args.precompute), "srs_window_table_form": {-1: "library policy (full unless HBM is tight)", 0: "full", 1: "GLV (half size)"}[args.glv_table],
                       "srs_window_table_build_ms": prim["srs_window_table_build_ms"],    # once per key, outside the timed region
                       "srs_gen_ms": prim["srs_gen_ms"],
                       "value_is": "inputs resident in HBM when the timed region starts (the bench contract); the PCIe-inclusive rates are "
                                   "value_h2d_inclusive (pinned host coefficients, copy overlapped) and trait_shaped (pageable, blocking calls)",
                       "coefficients": "device-resident when the timed region starts (value); pinned host memory "
                                       "(value_h2d_inclusive); pageable host memory, blocking calls (trait_shaped)",
                       "parallelism": "1 GPU" if world == 1 else f"SRS/coefficients of ONE polynomial of {world} x 2^{log_degree} coefficients sharded in "
                                                                 f"{world} contiguous chunks, one all_gather of partial points + division carries per step"},
            "commit_open_per_s": prim["commit_open_per_s"],
            # what a caller of the trait gets: commit(&poly) + open(&poly) as two blocking calls on pageable host coefficients
            "value_trait_shaped": None if not prim["trait_shaped"] else {
                "value": prim["trait_shaped"]["value"], "unit": "pairs/s", "ms_per_step": prim["trait_shaped"]["ms_per_commit_open"],
                "commit_ms": prim["trait_shaped"]["commit_ms"], "open_ms": prim["trait_shaped"]["open_ms"],
                "parity_ok": prim["trait_shaped"]["parity_ok"], "coefficients": "pageable host memory, PCIe inside both calls"},
            "value_h2d_inclusive": prim["value_h2d_inclusive"],
            "trait_shaped": prim["trait_shaped"],
            "blocking_msm_ms": prim["blocking_msm_ms"],
            "exchange_host_ms": prim["exchange_host_ms"],
            "msm_phase_ms": prim["msm_phase_ms"],
            "roofline": prim["roofline"],
        }
        if sec is not None:
            out["secondary"] = {
                "config": {"workload": cfg(sec["log_degree"]), "log_degree": sec["log_degree"],
                           "pairs_per_step": sec["pairs_per_step"],
                           "srs_window_table_build_ms": sec["srs_window_table_build_ms"]},
                "value": sec["value"], "unit": "pairs/s", "steps": sec["steps"], "warmup": sec["warmup"],
                "ms_per_step": sec["ms_per_step"], "commit_open_per_s": sec["commit_open_per_s"],
                "value_h2d_inclusive": sec["value_h2d_inclusive"], "trait_shaped": sec["trait_shaped"],
                "blocking_msm_ms": sec["blocking_msm_ms"], "parity": sec["parity"],
                "msm_phase_ms": sec["msm_phase_ms"], "roofline": sec["roofline"]}
        if workloads:
            out["workloads"] = workloads
        if cpu is not None:
            out["cpu_baseline"] = cpu
        out["bench_wall_s"] = time.perf_counter() - t_start
        emit(out)
    D.close()
    if not all_ok:
        raise SystemExit("parity check FAILED (see the `parity` block of the line)")


if __name__ == "__main__":
    main()
