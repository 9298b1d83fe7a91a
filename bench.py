#!/usr/bin/env python3
"""bench.py -- KZG commit+open hot path on MI355X (BASELINE.json configs[1]).

One step = one MarlinKZG10<Bls12_381> commit + one single-point open of one dense
polynomial of degree 2^20 per GPU, hiding off (the shape bench-templates times,
bench-templates/src/lib.rs:69-84,106-138):
    commit : MSM of d+1 pairs over the resident SRS           (kzg10/mod.rs:175-178)
    open   : witness polynomial p/(x-z) on the device          (kzg10/mod.rs:217-240)
             MSM of d pairs                                    (kzg10/mod.rs:255-258)
SRS, coefficients and the evaluation point's quotient stay in HBM; only the two 96-byte
affine results come back to the host.  value = G1 (base, scalar) pairs per second, whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): ONE polynomial of degree
N*2^20 whose SRS and coefficients are sharded in contiguous chunks (weak scaling: fixed pairs
per GPU); each rank runs the full Pippenger on its chunk and the partial commitments /
opening proofs are combined with an all_gather + EC adds (RCCL has no EC reduce op); the
division carry crosses ranks as one Fr element.  See poly-commit_amd/sharded.py.
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

PAIR_BYTES = {"bls12_381": 128, "bn254": 96, "pallas": 96}   # affine base + 32-byte scalar
HBM_PEAK_GBPS = 8000.0                                        # MI355X_MICROARCH.md: 8 TB/s spec


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
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    curve = args.curve
    poly_len = 1 << 24
    n_rows, n_cols, _ = O.ligero_dims(255 if curve != "bn254" else 254, poly_len, 4)
    log_n = (n_cols * 4 - 1).bit_length()
    rows = n_rows // world                      # rows are independent: shard by rows, no collective
    ctx = pc.Context(local_rank)
    ctx.set_timing(True)
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
        ctx.ntt_batch(curve, x.data_ptr(), log_n, out=y.data_ptr(), rows=rows, in_cols=n_cols)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.ntt_batch(curve, x.data_ptr(), log_n, out=y.data_ptr(), rows=rows, in_cols=n_cols)
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
        print(json.dumps({
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
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    curve = "bn254" if args.curve == "bls12_381" else args.curve
    total = (1 << args.log_degree) + 1
    per = (total + world - 1) // world
    lo, hi = rank * per, min(total, (rank + 1) * per)
    n = hi - lo
    ctx = pc.Context(local_rank)
    ctx.set_timing(True)
    bases = O.gen_bases(curve, n)                  # synthetic chunk (every rank the same points: throughput only)
    srs = ctx.upload_srs(curve, bases)
    if args.precompute:
        srs.precompute()
    for _ in range(3):                  # create every pipeline of the SRS (streams + workspace) before anything is timed
        srs.msm(np.zeros((1, 4), dtype=np.uint64), n=0)     # empty MSM: creates the pipeline, launches nothing
    polys = [torch.from_numpy(O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0100 + j, n)).view(np.int64)).cuda()
             for j in range(args.polys)]
    ptrs, lens = [p.data_ptr() for p in polys], [n] * args.polys
    torch.cuda.synchronize()

    def step():
        part = srs.msm_batch(ptrs, lens)
        if dist is None:
            return part
        t = torch.from_numpy(part.reshape(-1).view(np.int64).copy()).cuda()
        out = torch.empty(world * t.numel(), dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(out, t)
        allp = out.cpu().numpy().view(np.uint64).reshape(world, args.polys, -1)
        return np.stack([pc.points_sum(curve, np.ascontiguousarray(allp[:, j])) for j in range(args.polys)])

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
        print(json.dumps({
            "metric": "MSM G1-scalar-pairs/sec, batched MarlinKZG10<Bn254> commit (64 polys, deg 2^20, SRS sharded)",
            "value": pairs * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 limbs (254-bit modular integer)", "data": "synthetic",
            "config": {"workload": f"{args.polys} x MarlinKZG10<{curve}> commit, deg 2^{args.log_degree}, one SRS in {world} "
                                   f"contiguous chunk(s) (BASELINE configs[2])", "polys_per_s": args.polys * args.steps / dt,
                       "parallelism": "1 GPU" if world == 1 else f"SRS sharded over {world} GPUs, all_gather of {args.polys} partial points"}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-degree", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        return bench_ntt(args)
    if args.workload == "batch":
        return bench_batch(args)

    import torch
    import oracle_lib as O          # synthetic inputs + cpu_baseline leg only (never the measured path)
    import poly_commit_amd as pc
    from poly_commit_amd import sharded

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
    d = 1 << args.log_degree
    n = d + 1 if world == 1 else d          # coefficients held by this rank
    ctx = pc.Context(local_rank)
    if args.window_bits or args.chunk:
        ctx.set_msm_tuning(args.window_bits, args.chunk)
    ctx.set_timing(True)

    # ---- synthetic inputs: bases (i+1)G, coefficients SplitMix64(seed) -------
    eng = sharded.HipEngine(ctx, curve)
    job = sharded.ShardedKzg(eng, curve, rank, world, dist)
    bases = O.gen_bases(curve, n + 1)       # +1: the open of shard r > 0 reaches one base back
    job.load_srs_chunk(bases, precompute=bool(args.precompute))
    coeffs_h = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x5EED0001 + rank, n))
    coeffs = torch.from_numpy(coeffs_h.view(np.int64)).cuda()
    z_mont = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x2EE7, 1))[0]
    job.set_point(z_mont)
    torch.cuda.synchronize()

    import collections
    depth = max(0, args.inflight)
    pending = collections.deque()

    def step():
        # commit and open of one polynomial; up to `depth` results stay in flight so that the
        # latency-bound tail of one MSM overlaps the bucket accumulation of the next
        pending.append(job.commit_async(coeffs, n))
        if depth == 0:                       # strictly blocking calls: the commitment is back before the open starts
            pending.popleft().result()
        pending.append(job.open_async(coeffs, n))
        while len(pending) > depth:
            pending.popleft().result()

    def drain():
        while pending:
            pending.popleft().result()

    ph_sum, n_msm = np.zeros(8), 0
    for _ in range(args.warmup):
        step()
    drain()
    eng.phases = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                  # every commitment and proof is on the host here
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    for ph in eng.phases:
        ph_sum += np.array(ph); n_msm += 1
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # The same kernel without a second pipeline competing for the CUs: a few strictly serial MSMs after
    # the timed region (reported beside the timed-region figure, which includes queueing behind the
    # other pipeline's kernels).
    eng.phases = []
    for _ in range(3):
        job.commit_async(coeffs, n).result()
    acc_serial_ms = float(np.mean([ph[3] for ph in eng.phases])) if eng.phases else 0.0

    pairs_per_step = world * (2 * n - 1) if world == 1 else world * (2 * n) - 1
    value = pairs_per_step * args.steps / dt
    ph = ph_sum / max(n_msm, 1)
    # dominant kernel = bucket accumulation (phase index 3); algorithmic bytes = pairs * B/pair
    acc_ms = float(ph[3])
    pairs_per_launch = (2 * n - 1) / 2.0
    achieved = pairs_per_launch * PAIR_BYTES[curve] / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else None

    if rank == 0:
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("accumulate_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "MSM G1-scalar-pairs/sec inside KZG commit+open (MarlinKZG10<Bls12_381> shape, hiding off)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Fq / 255-bit Fr modular integer)", "data": "synthetic",
            "config": {"workload": f"MarlinKZG10<{curve}> commit+open, dense poly deg 2^{args.log_degree} per GPU, "
                                   f"SRS resident, hiding off (BASELINE configs[1])",
                       "curve": curve, "log_degree": args.log_degree, "pairs_per_step": pairs_per_step,
                       "inflight": depth, "srs_window_table": bool(args.precompute),
                       "srs_window_table_build_ms": eng.precompute_ms,    # once per key, outside the timed region
                       "parallelism": "1 GPU" if world == 1 else f"SRS/coefficients sharded in {world} contiguous chunks, "
                                                                 f"all_gather of partial points"},
            "commit_open_per_s": args.steps / dt if world == 1 else None,
            "msm_phase_ms": {k: float(v) for k, v in zip(
                ["digits_hist", "scan", "scatter", "accumulate", "seg_reduce", "bucket_reduce"], ph[:6])},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "kernel": "k_accumulate (bucket accumulation), avg of hipEvent-timed launches on the MSM pipelines' streams",
                         "algorithmic_bytes_per_launch": pairs_per_launch * PAIR_BYTES[curve],
                         "serial": {"kernel_ms": acc_serial_ms,
                                    "achieved": n * PAIR_BYTES[curve] / (acc_serial_ms * 1e-3) / 1e9 if acc_serial_ms > 0 else None,
                                    "frac": n * PAIR_BYTES[curve] / (acc_serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if acc_serial_ms > 0 else None,
                                    "note": "3 blocking commit MSMs after the timed region: no other pipeline on the GPU"}},
        }
        if not args.no_cpu_baseline and world == 1:     # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(curve, args.log_degree)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
