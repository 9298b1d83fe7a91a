#!/usr/bin/env python3
"""bench.py -- KZG commit+open hot path on MI355X (BASELINE.json configs[1]).

One step = one MarlinKZG10<Bls12_381> commit + one single-point open of one dense
polynomial of degree 2^24 per GPU (primary; the 2^20 case BASELINE.json also names is reported in the
`secondary` block of the same JSON line), hiding off (the shape bench-templates times,
bench-templates/src/lib.rs:69-84,106-138):
    commit : MSM of d+1 pairs over the resident SRS           (kzg10/mod.rs:175-178)
    open   : witness polynomial p/(x-z) on the device          (kzg10/mod.rs:217-240)
             MSM of d pairs                                    (kzg10/mod.rs:255-258)
SRS, coefficients and the evaluation point's quotient stay in HBM; only the two 96-byte
affine results come back to the host.  value = G1 (base, scalar) pairs per second, whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): ONE polynomial of degree
N*2^24 whose SRS and coefficients are sharded in contiguous chunks (weak scaling: fixed pairs
per GPU); each rank runs the full Pippenger on its chunk and the partial commitments /
opening proofs are combined with an all_gather + EC adds (RCCL has no EC reduce op); the
division carry crosses ranks as one Fr element.  See poly_commit_amd/sharded.py.
"""
import argparse
import json
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
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PC_BENCH_FORCE_DIST"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def emit(obj):
    os.write(_RESULT_FD, (json.dumps(obj) + "\n").encode())


PAIR_BYTES = {"bls12_381": 128, "bn254": 96, "pallas": 96}   # affine base + 32-byte scalar
HBM_PEAK_GBPS = 8000.0                                        # MI355X_MICROARCH.md: 8 TB/s spec


_MADD_PEAK = {}


def madd_peak(curve):
    """Mixed additions per second of a pure-arithmetic loop (no memory traffic) of the same XYZZ += affine addition
    the accumulate kernel runs: tools/microbench measured live on this GPU (rank 0, once), else the committed figure."""
    if curve != "bls12_381":
        return None
    if curve in _MADD_PEAK:
        return _MADD_PEAK[curve]
    res = None
    exe = os.path.join(ROOT, "tools", "microbench")
    if os.path.exists(exe):
        try:
            import subprocess
            txt = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
            for line in txt.splitlines():
                if line.startswith("XYZZ madd bls12_381"):
                    res = {"madd_per_s": float(line.split()[-3]) * 1e6, "source": "tools/microbench, this run"}
        except Exception:
            res = None
    if res is None:
        res = {"madd_per_s": 5.726e9, "source": "profiles/r02_microbench.txt"}
    _MADD_PEAK[curve] = res
    return res


def cpu_baseline(curve, log_d, budget_s=15.0):
    """CPU restatement of ark-ec's Pippenger (oracle/, 'port'), timed on this box's cores on a
    bounded sample: the largest power-of-two MSM that fits the time budget."""
    import oracle_lib as O
    cores = os.cpu_count() or 1
    n0 = 1 << 14
    b = O.gen_bases(curve, n0)
    s = O.gen_scalars(curve, 1, n0)
    t = time.time()
    O.msm_pippenger(curve, b, s, cores, 1)
    rate = n0 / max(time.time() - t, 1e-6)
    lg = 14
    while lg < log_d and (1 << (lg + 1)) / rate < budget_s:
        lg += 1
    n = 1 << lg
    b = O.gen_bases(curve, n)
    s = O.gen_scalars(curve, 2, n)
    best = None
    for mode in (1, 0):   # chunk-parallel and window-parallel schedules; keep the faster
        t = time.time()
        O.msm_pippenger(curve, b, s, cores, mode)
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
        if dt > budget_s:
            break
    return {"value": n / best, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"1 MSM of 2^{lg} {curve} G1 pairs, restated ark-ec signed-digit Pippenger "
                      f"(oracle/oracle.cpp), best of chunk-/window-parallel, {cores} threads"}


def bench_ntt(args):
    """BASELINE configs[4]: LigeroPCS over BLS12-381 Fr, 2^24 coefficients, rho_inv = 4 ->
    512 x 32768 matrix -> 512 forward NTTs of size 2^17 (linear_codes/mod.rs:118-138)."""
    import torch
    import oracle_lib as O
    import poly_commit_amd as pc
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    curve = args.curve
    poly_len = 1 << 24
    n_rows, n_cols, _ = O.ligero_dims(255 if curve != "bn254" else 254, poly_len, 4)
    log_n = (n_cols * 4 - 1).bit_length()
    from poly_commit_amd import sharded
    ctx = pc.Context(local_rank)
    ctx.set_timing(True)
    shard = sharded.ShardedRows(sharded.HipEngine(ctx, curve), rank, world)   # rows are independent: shard by rows, no collective
    r_lo, r_hi = shard.row_range(n_rows)
    rows = r_hi - r_lo
    co = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0500 + rank, rows * n_cols))
    x = torch.from_numpy(co.view(np.int64)).cuda()
    y = torch.empty((rows << log_n, 4), dtype=torch.int64, device="cuda")
    # the step after the encoding in LinearCodePCS::commit (linear_codes/mod.rs:256-263): column digests
    leaves = torch.empty((1 << log_n, 32), dtype=torch.uint8, device="cuda")
    hash_ms = merkle_ms = None
    if world == 1:
        ctx.column_hash(curve, y.data_ptr(), "blake2s", out=leaves.data_ptr(), rows=rows, n_cols=1 << log_n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.column_hash(curve, y.data_ptr(), "blake2s", out=leaves.data_ptr(), rows=rows, n_cols=1 << log_n)
        torch.cuda.synchronize()
        hash_ms = (time.perf_counter() - t0) / 3 * 1e3
        # ... and the Merkle tree over them (create_merkle_tree, linear_codes/mod.rs:506-521)
        nodes = torch.empty(((1 << log_n) - 1, 32), dtype=torch.uint8, device="cuda")
        ctx.merkle_tree(leaves.data_ptr(), "sha256", True, out=nodes.data_ptr(), n_leaves=1 << log_n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.merkle_tree(leaves.data_ptr(), "sha256", True, out=nodes.data_ptr(), n_leaves=1 << log_n)
        torch.cuda.synchronize()
        merkle_ms = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    ph = np.zeros(2)
    for _ in range(args.warmup):
        shard.encode(x, rows, n_cols, log_n, y)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        shard.encode(x, rows, n_cols, log_n, y)
        ph += np.array(ctx.last_ntt_phases_ms())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ph /= args.steps
    alg_bytes = rows * (n_cols + (1 << log_n)) * 32
    kern_ms = float(ph.sum())
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None
    if rank == 0:
        emit(({
            "metric": "Ligero Reed-Solomon NTT input coefficients/sec (LigeroPCS over BLS12-381 Fr, 2^24 coeffs)",
            "value": world * rows * n_cols * args.steps / dt, "unit": "coeffs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32 limbs (255-bit Fr modular integer)", "data": "synthetic",
            "config": {"workload": f"{n_rows} x {n_cols} matrix, {n_rows} forward NTTs of size 2^{log_n} (BASELINE configs[4])",
                       "parallelism": "1 GPU" if world == 1 else f"rows sharded over {world} GPUs, no collective"},
            "ntt_phase_ms": {"pass_a": float(ph[0]), "pass_b": float(ph[1])},
            "column_hash_blake2s_ms": hash_ms,   # not part of `value`: the next steps of the commit, device-resident
            "merkle_tree_sha256_ms": merkle_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": None,
                         "kernel": "k_ntt_pass_a + k_ntt_pass_b (one batched NTT = both)",
                         "algorithmic_bytes_per_launch": alg_bytes}}))
    if dist is not None:
        dist.destroy_process_group()


def bench_batch(args):
    """BASELINE configs[2]: 64 polynomials of degree 2^20 over BN254 committed against ONE SRS
    that is split into N contiguous chunks (one per GPU).  Every GPU runs the 64 partial MSMs of
    its chunk as one pipelined batch (pc_hip_msm_batch); the 64 partial points per rank are
    combined with one all_gather (64 x 64 B per rank) + EC adds.  Strong scaling: the job is
    fixed (64 x (2^20 + 1) pairs), per-GPU work shrinks with N."""
    import torch
    import oracle_lib as O
    import poly_commit_amd as pc
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    curve = "bn254" if args.curve == "bls12_381" else args.curve
    from poly_commit_amd import sharded
    total = (1 << args.log_degree) + 1
    lo, hi = sharded.ShardedBatch.chunk_range(total, rank, world)
    n = hi - lo
    ctx = pc.Context(local_rank)
    ctx.set_timing(True)
    job = sharded.ShardedBatch(sharded.HipEngine(ctx, curve), curve, rank, world, dist)
    job.load_srs_chunk(O.gen_bases(curve, n), precompute=bool(args.precompute))   # synthetic chunk (every rank the same points: throughput only)
    polys = [torch.from_numpy(O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0100 + j, n)).view(np.int64)).cuda()
             for j in range(args.polys)]
    lens = [n] * args.polys
    torch.cuda.synchronize()

    def step():
        return job.commit_batch(polys, lens)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        pairs = args.polys * total
        emit(({
            "metric": "MSM G1-scalar-pairs/sec, batched MarlinKZG10<Bn254> commit (64 polys, deg 2^20, SRS sharded)",
            "value": pairs * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 limbs (254-bit modular integer)", "data": "synthetic",
            "config": {"workload": f"{args.polys} x MarlinKZG10<{curve}> commit, deg 2^{args.log_degree}, one SRS in {world} "
                                   f"contiguous chunk(s) (BASELINE configs[2])", "polys_per_s": args.polys * args.steps / dt,
                       "parallelism": "1 GPU" if world == 1 else f"SRS sharded over {world} GPUs, all_gather of {args.polys} partial points"}}))
    if dist is not None:
        dist.destroy_process_group()


def kzg_case(ctx, args, curve, log_degree, steps, warmup, world, rank, dist, with_h2d):
    """One KZG commit+open workload on this rank: timed legs + the post-region blocking MSMs.
    Returns a dict (timings are this rank's; the caller takes the max over ranks)."""
    import collections
    import math
    import torch
    import oracle_lib as O          # synthetic inputs only (never the measured path)
    from poly_commit_amd import sharded

    d = 1 << log_degree
    n = d + 1 if world == 1 else d          # coefficients held by this rank
    # ---- synthetic inputs: bases (i+1)G, coefficients SplitMix64(seed) -------
    eng = sharded.HipEngine(ctx, curve)
    job = sharded.ShardedKzg(eng, curve, rank, world, dist)
    bases = O.gen_bases(curve, n + 1)       # +1: the open of shard r > 0 reaches one base back
    job.load_srs_chunk(bases, precompute=bool(args.precompute))
    del bases
    coeffs_h = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0001 + rank, n))
    coeffs = torch.from_numpy(coeffs_h.view(np.int64)).cuda()
    z_mont = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x2EE7, 1))[0]
    job.set_point(z_mont)
    torch.cuda.synchronize()

    depth = max(0, args.inflight)
    pending = collections.deque()

    def drain():
        if dist is not None and pending and not os.environ.get("PC_DIAG_DIST_NO_EXCHANGE"):
            job.exchange(None, 0, [pending.popleft() for _ in range(len(pending))])
        while pending:
            pending.popleft().result()

    def step_resident(_k):
        # commit and open of one polynomial; up to `depth` results stay in flight so that the
        # latency-bound tail of one MSM overlaps the bucket accumulation of the next
        # (N > 1: the open's exchange step -- shard evaluation + all_gather of one Fr per rank -- runs first, while
        # the previous step's MSMs are still in flight, so the blocking collective does not drain the pipelines)
        if dist is not None and depth > 0 and not os.environ.get("PC_DIAG_DIST_NO_EXCHANGE"):
            # N > 1, pipelined: ONE collective per step -- this step's shard evaluations (the division carries) travel
            # with the partial points of the step that left the pipeline (ShardedKzg.exchange)
            done = [pending.popleft() for _ in range(max(0, len(pending) - 2 * (depth - 1)))]
            carry, _results = job.exchange(coeffs, n, done)
            pending.append(job.commit_async(coeffs, n))
            pending.append(job.open_async(coeffs, n, prepared=True, carry=carry))
            return
        carry = job.open_prepare(coeffs, n)
        pending.append(job.commit_async(coeffs, n))
        if depth == 0:                       # strictly blocking calls: the commitment is back before the open starts
            pending.popleft().result()
        pending.append(job.open_async(coeffs, n, prepared=True, carry=carry))
        while len(pending) > depth:
            pending.popleft().result()

    def timed(step_fn, steps, warmup):
        for k in range(warmup):
            step_fn(k)
        drain()
        eng.phases = []
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step_fn(k)
        drain()                              # every commitment and proof is on the host here
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, list(eng.phases)

    if steps is None and dist is not None:
        steps = 20          # every rank must run the same number of steps
    if steps is None:       # no --steps: long enough for >= 1.2 s of timed region (the driver's gpu_busy sampler needs it)
        t0 = time.perf_counter()
        step_resident(0); drain()
        torch.cuda.synchronize()
        est = time.perf_counter() - t0
        steps = max(10, int(math.ceil(1.2 / max(est, 1e-4))))
    dt, phases = timed(step_resident, steps, warmup)
    ph = np.mean(np.array(phases), axis=0) if phases else np.zeros(8)

    # ---- the same steps with the coefficients handed over as HOST memory (what the Rust shim holds):
    # one pinned H2D copy per polynomial (commit and open share it), double-buffered so that the copy of
    # step k+1 overlaps the MSMs of step k.  Reported beside `value`, never as `value`.
    h2d = None
    if with_h2d and world == 1:
        host = torch.from_numpy(coeffs_h.view(np.int64)).pin_memory()
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
            pending.append(job.commit_async(bufs[k % 3], n))
            if depth == 0:
                pending.popleft().result()
            pending.append(job.open_async(bufs[k % 3], n))
            while len(pending) > min(depth, 2):
                pending.popleft().result()

        dt_h, _ = timed(step_h2d, steps, warmup)
        torch.cuda.synchronize()
        h2d = {"ms_per_step": dt_h / steps * 1e3, "value": (2 * n - 1) * steps / dt_h, "unit": "pairs/s",
               "note": "coefficients start in pinned HOST memory every step: one H2D copy of the polynomial per "
                       "commit+open (32 B/coefficient), triple-buffered on its own stream so it overlaps the previous "
                       "step's MSMs (SURVEY.md 8d: scalars H2D included)"}
        del bufs, host

    # The kernels without a second pipeline competing for the CUs: strictly serial MSMs after the
    # timed region.  msm_phase_ms and roofline.serial come from these (they agree with rocprofv3's
    # per-kernel averages); the timed-region brackets include queueing behind the other pipeline.
    eng.phases = []
    for _ in range(3):
        job.commit_async(coeffs, n).result()
    sp = np.mean(np.array(eng.phases), axis=0) if eng.phases else np.zeros(8)
    t0 = time.perf_counter()
    for _ in range(3):
        job.commit_async(coeffs, n).result()
    blocking_msm_ms = (time.perf_counter() - t0) / 3 * 1e3

    pairs_per_step = world * (2 * n - 1) if world == 1 else world * (2 * n) - 1
    acc_ms, acc_serial_ms = float(ph[3]), float(sp[3])
    pairs_per_launch = (2 * n - 1) / 2.0
    bytes_per_launch = pairs_per_launch * PAIR_BYTES[curve]
    achieved = bytes_per_launch / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else None
    ach_serial = n * PAIR_BYTES[curve] / (acc_serial_ms * 1e-3) / 1e9 if acc_serial_ms > 0 else None
    traffic = traffic_raw = None
    tf = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")       # PMC passes are separate rocprofv3 runs (tools/pmc_summary.py);
    if os.path.exists(tf):                                            # keyed by size and table mode, null when not measured
        try:
            key = f"{curve}:2^{log_degree}:{'table' if args.precompute else 'table-free'}"
            doc = json.load(open(tf))
            traffic = doc.get("accumulate_hbm_bytes_per_launch", {}).get(key)
            traffic_raw = doc.get("accumulate_fetch_raw_plus_write_bytes_per_launch", {}).get(key)
        except Exception:
            traffic = None
    # The bound that actually binds: the kernel is modular arithmetic on the VALU.  One mixed addition per signed
    # digit of every scalar (zero digits, 2^-c of them, skipped) against a memory-free loop of the same additions.
    shape = ctx.last_msm_shape()
    arith = None
    pk = madd_peak(curve) if rank == 0 else None
    if pk and acc_serial_ms > 0:
        adds = float(n) * shape["digits_per_scalar"]
        arith = {"bound": "valu", "unit": "mixed additions/s (XYZZ += affine, 8M + 2S in Fq)",
                 "achieved": adds / (acc_serial_ms * 1e-3), "peak": pk["madd_per_s"],
                 "frac": adds / (acc_serial_ms * 1e-3) / pk["madd_per_s"], "peak_source": pk["source"],
                 "additions_per_launch": adds, "window_bits": shape["window_bits"],
                 "digits_per_scalar": shape["digits_per_scalar"], "buckets": shape["buckets"],
                 "note": "reported beside the prescribed HBM roofline: k_accumulate in the blocking MSMs (roofline.serial) "
                         "against a pure-arithmetic loop of the same addition on this GPU"}
    valu_busy = None                      # SQ counters are a separate rocprofv3 pass (tools/sq_summary.py), like the PMC traffic
    vf = os.path.join(ROOT, "profiles", "r02_valu.json")
    if os.path.exists(vf) and args.precompute:
        try:
            ks = json.load(open(vf)).get("workloads", {}).get(f"kzg_2p{log_degree}", {})
            valu_busy = next((v.get("valu_busy") for k, v in ks.items() if "k_accumulate<" in k), None)
        except Exception:
            valu_busy = None
    if arith is not None:
        arith["valu_busy"] = valu_busy
        arith["valu_busy_note"] = ("SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel time * 2.4 GHz) from profiles/r02_valu.json (separate rocprofv3 "
                                   "--pmc pass of this workload); the nominal clock understates it under sustained VALU load")
    res = {
        "log_degree": log_degree, "steps": steps, "warmup": warmup, "dt": dt, "pairs_per_step": pairs_per_step,
        "value": pairs_per_step * steps / dt, "ms_per_step": dt / steps * 1e3,
        "commit_open_per_s": steps / dt if world == 1 else None,
        "value_h2d_inclusive": h2d,
        "srs_window_table_build_ms": eng.precompute_ms,
        "exchange_host_ms": ({k: (v / max(1, job.exchange_ms["calls"]) if k != "calls" else v) for k, v in job.exchange_ms.items()}
                             if dist is not None else None),
        "blocking_msm_ms": blocking_msm_ms,
        "msm_phase_ms": {k: float(v) for k, v in zip(
            ["digits_hist", "scan", "scatter_fine_sort", "accumulate", "seg_reduce", "bucket_reduce"], sp[:6])},
        # achieved: algorithmic bytes of one launch / the kernel's launch duration, hipEvent brackets on the pipeline's own
        # stream, in blocking MSMs issued right after the timed region -- the duration rocprofv3's per-kernel average
        # reproduces (profiles/).  Inside the pipelined region consecutive accumulations of different pipelines overlap
        # (the next one fills the SIMDs as the previous one's workgroups retire), so a launch's bracket there includes
        # time it shares with its neighbour: reported beside it as `timed_region`, together with the per-launch share of
        # the step time.
        "roofline": {"bound": "hbm", "achieved": ach_serial, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (ach_serial / HBM_PEAK_GBPS) if ach_serial else None, "traffic": traffic,
                     "traffic_note": "PMC FETCH_SIZE (doubled per the gfx950 note of MI355X_MICROARCH.md) + WRITE_SIZE per launch, separate "
                                     "rocprofv3 passes of this workload (profiles/r02_pmc_traffic.json); undoubled: "
                                     + (f"{traffic_raw:.4g} B" if traffic_raw else "n/a") + " -- for this kernel's 16-byte gathers the raw figure is the plausible one",
                     "kernel": "k_accumulate (bucket accumulation): average hipEvent bracket of its launches on the MSM pipeline's stream, "
                               "3 blocking commit MSMs after the timed region (no second pipeline sharing the SIMDs)",
                     "kernel_ms": acc_serial_ms,
                     "algorithmic_bytes_per_launch": n * PAIR_BYTES[curve],
                     "arithmetic": arith,
                     "timed_region": {"bracket_ms": acc_ms, "achieved_from_bracket": achieved,
                                      "ms_per_step_over_launches": dt / steps * 1e3 / 2,
                                      "achieved_from_step_time": (pairs_per_step / world) * PAIR_BYTES[curve] / (dt / steps) / 1e9,
                                      "note": "brackets of overlapping launches inside the timed region; step time / 2 launches is the "
                                              "per-launch time the pipelined run sustains"},
                     # kept under its round-1 name for readers of earlier lines
                     "serial": {"kernel_ms": acc_serial_ms, "achieved": ach_serial,
                                "frac": ach_serial / HBM_PEAK_GBPS if ach_serial else None,
                                "algorithmic_bytes_per_launch": n * PAIR_BYTES[curve]}},
    }
    eng.srs.free()
    del coeffs
    torch.cuda.empty_cache()
    return res


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
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=2,
                    help="commit/open results allowed in flight (0 = strictly sequential, blocking calls)")
    ap.add_argument("--workload", default="kzg", choices=["kzg", "ntt", "batch"],
                    help="kzg (default, BASELINE configs[1]), ntt (configs[4]: Ligero 2^24 coefficients) or "
                         "batch (configs[2]: 64 x MarlinKZG10<Bn254> commits, SRS sharded over the GPUs)")
    ap.add_argument("--polys", type=int, default=64)
    ap.add_argument("--precompute", type=int, default=1,
                    help="1 (default): build the SRS window table in HBM once after the upload "
                         "(pc_hip_srs_precompute; part of SRS residency, outside the timed region); 0: table-free MSM")
    args = ap.parse_args()
    if args.workload == "ntt":
        if args.steps is None:
            args.steps = 100
        return bench_ntt(args)
    if args.workload == "batch":
        if args.steps is None:
            args.steps = 10
        if args.log_degree == 24:
            args.log_degree = 20
        return bench_batch(args)

    import torch
    import poly_commit_amd as pc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PC_BENCH_FORCE_DIST"):      # the latter: exercise the RCCL path on one GPU
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    curve = args.curve
    ctx = pc.Context(local_rank)
    if args.window_bits or args.chunk:
        ctx.set_msm_tuning(args.window_bits, args.chunk)
    ctx.set_timing(True)

    prim = kzg_case(ctx, args, curve, args.log_degree, args.steps, args.warmup, world, rank, dist, with_h2d=not args.no_h2d)
    sec = None
    if world == 1 and args.secondary_log_degree and args.secondary_log_degree != args.log_degree:
        sec = kzg_case(ctx, args, curve, args.secondary_log_degree, None, args.warmup, world, rank, dist,
                       with_h2d=not args.no_h2d)

    if rank == 0:
        def cfg(lg):
            return (f"MarlinKZG10<{curve}> commit+open, dense poly deg 2^{lg} per GPU, SRS resident, hiding off "
                    f"(BASELINE metric sizes: 2^20 = configs[1], 2^24 = north-star target)")
        out = {
            "metric": "MSM G1-scalar-pairs/sec inside KZG commit+open (MarlinKZG10<Bls12_381> shape, hiding off)",
            "value": prim["value"], "unit": "pairs/s", "n_gpus": world, "steps": prim["steps"], "warmup": prim["warmup"],
            "ms_per_step": prim["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Fq / 255-bit Fr modular integer)", "data": "synthetic",
            "parity": "oracle-only (the reference holds no golden vector on this path; its arithmetic crates are not "
                      "buildable here) -- see DESIGN.md section 2",
            "config": {"workload": cfg(args.log_degree), "curve": curve, "log_degree": args.log_degree,
                       "pairs_per_step": prim["pairs_per_step"], "inflight": max(0, args.inflight),
                       "srs_window_table": bool(args.precompute),
                       "srs_window_table_build_ms": prim["srs_window_table_build_ms"],    # once per key, outside the timed region
                       "coefficients": "device-resident when the timed region starts (value); pinned host memory "
                                       "(value_h2d_inclusive)",
                       "parallelism": "1 GPU" if world == 1 else f"SRS/coefficients sharded in {world} contiguous chunks, "
                                                                 f"all_gather of partial points"},
            "commit_open_per_s": prim["commit_open_per_s"],
            "value_h2d_inclusive": prim["value_h2d_inclusive"],
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
                "value_h2d_inclusive": sec["value_h2d_inclusive"], "blocking_msm_ms": sec["blocking_msm_ms"],
                "msm_phase_ms": sec["msm_phase_ms"], "roofline": sec["roofline"]}
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(curve, args.log_degree)
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
