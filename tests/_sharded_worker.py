"""Worker for tests/test_sharded_gloo.py: one rank of a world_size-N gloo job.  Exercises the
index / carry / fold logic of poly_commit_amd/sharded.py with an oracle-backed engine (a test
double for HipEngine -- there is no GPU here) and compares with the single-process oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import torch.distributed as dist

import oracle_lib as O
import pyref as R
import poly_commit_amd as pc
from poly_commit_amd import sharded


class OracleEngine:
    """Same interface as sharded.HipEngine, host buffers, CPU oracle arithmetic (TEST DOUBLE)."""

    def __init__(self, curve):
        self.curve, self.bases, self.phases = curve, None, []

    def load_srs(self, bases):
        self.bases = np.ascontiguousarray(bases)

    def msm(self, scalars, n, base_offset, elem_off=0):
        sc = O.f_from_mont(self.curve, 1, np.ascontiguousarray(scalars[elem_off:elem_off + n]))
        self.phases.append([0.0] * 8)
        return O.msm_pippenger(self.curve, np.ascontiguousarray(self.bases[base_offset:]), sc, 2, 1)

    def div_scan(self, coeffs, n, z, carry_in):
        fr = R.CURVES[self.curve]["fr"]
        p = R.FIELDS[fr]["p"]
        ci = O.fr_from_mont_array(self.curve, np.ascontiguousarray(coeffs[:n]))
        zi = O.fr_from_mont_array(self.curve, np.asarray(z).reshape(1, 4))[0]
        acc = 0 if carry_in is None else O.fr_from_mont_array(self.curve, np.asarray(carry_in).reshape(1, 4))[0]
        out = [0] * n
        for i in range(n - 1, -1, -1):
            acc = (ci[i] + zi * acc) % p
            out[i] = acc
        return O.fr_mont_array(self.curve, out)

    def read_elem(self, buf, idx):
        return np.ascontiguousarray(buf[idx])

    def poly_eval(self, coeffs, n, z):
        return np.ascontiguousarray(self.div_scan(coeffs, n, z, None)[0])

    def points_sum(self, pts):
        return pc.points_sum(self.curve, pts)

    def msm_batch(self, polys, lens):
        return np.stack([self.msm(q, m, 0) for q, m in zip(polys, lens)])

    def ntt_rows(self, rows_buf, n_rows, in_cols, log_n, out_buf):
        return O.ntt_batch(self.curve, np.ascontiguousarray(rows_buf[:n_rows]), log_n, threads=1)

    # chained column digests (ShardedRows.commit): BLAKE2s with an explicit state (tests/_blake2s_state.py)
    def state_buffer(self, n_cols):
        import torch
        return torch.zeros((n_cols, 12), dtype=torch.int32)

    def digest_buffer(self, n_cols):
        import torch
        return torch.zeros((n_cols, 8), dtype=torch.int32)

    def column_hash_part(self, slab, rows, n_cols, rows_total, state, first, last, out, hash_name, col0, cols):
        import _blake2s_state as B
        assert hash_name == "blake2s"
        st = state.numpy().view(np.uint32)
        can = O.fr_from_mont_array(self.curve, np.ascontiguousarray(slab).reshape(-1, 4)) if rows else []
        for j in range(col0, col0 + cols):
            if first:
                (h, t), pending = B.init(), int(rows_total).to_bytes(8, "little")
            else:
                h, t = [int(x) for x in st[j, :8]], int(st[j, 8]) | (int(st[j, 9]) << 32)
                pending = int(st[j, 10]).to_bytes(4, "little") + int(st[j, 11]).to_bytes(4, "little")
            data = b"".join(int(can[r * n_cols + j]).to_bytes(32, "little") for r in range(rows))
            if last:
                out.numpy().view(np.uint8).reshape(-1, 32)[j] = np.frombuffer(B.absorb(h, t, pending, data, True), dtype=np.uint8)
            else:
                h, t, pending = B.absorb(h, t, pending, data, False)
                assert len(pending) == 8
                st[j, :8] = h
                st[j, 8], st[j, 9] = t & 0xFFFFFFFF, t >> 32
                st[j, 10], st[j, 11] = int.from_bytes(pending[:4], "little"), int.from_bytes(pending[4:], "little")

    def merkle_nodes(self, digests, n_leaves, tree_hash):
        leaves = [bytes(x) for x in digests.numpy().view(np.uint8).reshape(-1, 32)[:n_leaves]]
        return np.frombuffer(b"".join(R.merkle_tree(leaves, tree_hash, True)), dtype=np.uint8).reshape(-1, 32)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    # --engine hip: the real device engine, every rank on GPU 0 (gloo carries the 100-byte exchanges);
    # with --precompute the SRS chunk also gets its window table.  Default: the oracle-backed test double.
    use_hip = "--engine" in sys.argv and sys.argv[sys.argv.index("--engine") + 1] == "hip"
    precompute = "--precompute" in sys.argv
    ctx = pc.Context(0) if use_hip else None
    for curve in ("bls12_381", "bn254"):
        n = 3000 if use_hip else 96                # coefficients per rank
        total = n * world
        powers = O.gen_bases(curve, total)         # the "global" SRS
        coeffs = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xC0FFEE, total))
        z = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xBEEF, 1))[0]
        # rank r's chunk: bases[0] = power r*n - 1 (dummy on rank 0), bases[1 + j] = power r*n + j
        lo = rank * n
        chunk = np.zeros((n + 1, powers.shape[1]), dtype=np.uint64)
        chunk[1:] = powers[lo:lo + n]
        chunk[0] = powers[lo - 1] if rank else powers[0]
        mine = np.ascontiguousarray(coeffs[lo:lo + n])
        if use_hip:
            import torch
            job = sharded.ShardedKzg(sharded.HipEngine(ctx, curve), curve, rank, world, dist)
            job.load_srs_chunk(chunk, precompute=precompute)
            mine = torch.from_numpy(mine.view(np.int64)).cuda()
        else:
            job = sharded.ShardedKzg(OracleEngine(curve), curve, rank, world, dist)
            job.load_srs_chunk(chunk)
        job.set_point(z)
        comm = job.commit(mine, n)
        proof = job.open(mine, n)
        rc1, want_c = O.kzg_commit(curve, powers, coeffs, 2)
        rc2, want_w = O.kzg_open(curve, powers, coeffs, z, 2)
        ok &= rc1 == 0 and rc2 == 0 and bool((comm == want_c).all()) and bool((proof == want_w).all())
        # the pipelined form bench.py runs for N > 1: one collective carries the shard evaluations of the polynomial
        # about to be opened together with the partial points of results leaving the pipeline (ShardedKzg.exchange)
        f1 = job.commit_async(mine, n)
        carry, none = job.exchange(mine, n, [])
        f2 = job.open_async(mine, n, prepared=True, carry=carry)
        f3 = job.commit_async(mine, n)
        carry2, (c1, w1) = job.exchange(mine, n, [f1, f2])
        f4 = job.open_async(mine, n, prepared=True, carry=carry2)
        _, (c2, w2) = job.exchange(None, 0, [f3, f4])
        ok &= none == [] and all(bool((a == b).all()) for a, b in ((c1, want_c), (w1, want_w), (c2, want_c), (w2, want_w)))
    # BASELINE configs[2] shape: k polynomials against one SRS in `world` contiguous chunks (bench.py --workload batch)
    # and configs[4] shape: matrix rows split over the ranks, no collective (bench.py --workload ntt)
    for curve in ("bn254",):
        total, k = (700 if use_hip else 101), 5            # deliberately not a multiple of the world size
        powers = O.gen_bases(curve, total)
        polys = [O.f_to_mont(curve, 1, O.gen_scalars(curve, 0xBA7C0 + j, total)) for j in range(k)]
        lo, hi = sharded.ShardedBatch.chunk_range(total, rank, world)
        eng = sharded.HipEngine(ctx, curve) if use_hip else OracleEngine(curve)
        job = sharded.ShardedBatch(eng, curve, rank, world, dist)
        job.load_srs_chunk(np.ascontiguousarray(powers[lo:hi]))
        mine = [np.ascontiguousarray(q[lo:hi]) for q in polys]
        if use_hip:
            import torch
            mine = [torch.from_numpy(q.view(np.int64)).cuda() for q in mine]
        got = job.commit_batch(mine, [hi - lo] * k)
        for j in range(k):
            rc, want = O.kzg_commit(curve, powers, polys[j], 2)
            ok &= rc == 0 and bool((got[j] == want).all())
        n_rows, n_cols, log_n = 7, 16, 6
        mat = O.f_to_mont(curve, 1, O.gen_scalars(curve, 0x2075, n_rows * n_cols)).reshape(n_rows, n_cols, 4)
        rows = sharded.ShardedRows(eng, rank, world)
        r_lo, r_hi = rows.row_range(n_rows)
        want = O.ntt_batch(curve, mat, log_n, threads=1)
        if r_hi > r_lo:
            local = np.ascontiguousarray(mat[r_lo:r_hi])
            if use_hip:
                import torch
                x = torch.from_numpy(local.view(np.int64)).cuda()
                y = torch.empty(((r_hi - r_lo) << log_n, 4), dtype=torch.int64, device="cuda")
                rows.encode(x, r_hi - r_lo, n_cols, log_n, y)
                out = y.cpu().numpy().view(np.uint64).reshape(r_hi - r_lo, 1 << log_n, 4)
            else:
                out = rows.encode(local, r_hi - r_lo, n_cols, log_n)
            ok &= bool((out == want[r_lo:r_hi]).all())
        # the commitment's root with the rows spread over the ranks: chained column digests, tree on the last rank with rows
        fr = R.CURVES[curve]["fr"]
        cols_can = O.fr_from_mont_array(curve, np.ascontiguousarray(want).reshape(-1, 4))
        n_ext = 1 << log_n
        leaves = [R.column_digest(fr, [cols_can[r * n_ext + j] for r in range(n_rows)], "blake2s") for j in range(n_ext)]
        want_root = R.merkle_tree(leaves, "sha256", True)[0]
        if use_hip:
            slab = y if r_hi > r_lo else None
        else:
            slab = out if r_hi > r_lo else None
        for nb in (1, 3):
            root, nodes = rows.commit(slab, n_rows, n_ext, dist, "blake2s", "sha256", blocks=nb)
            ok &= bytes(root) == want_root
        # every row is owned exactly once
        owned = np.zeros(n_rows, dtype=np.int64)
        owned[r_lo:r_hi] = 1
        import torch
        t = torch.from_numpy(owned)
        dist.all_reduce(t)
        ok &= bool((t.numpy() == 1).all())
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {'OK' if ok else 'MISMATCH'}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
