"""KZG commit/open of ONE polynomial whose SRS and coefficients are sharded over the GPUs of a
node in contiguous chunks: rank r holds coefficients
[r*n, (r+1)*n) and the matching powers.  One process per GPU; torch.distributed (RCCL over
xGMI on GPUs, gloo in the CPU tests) carries
  * the partial commitments / proofs: one affine point per rank, all_gather + EC adds
    (RCCL has no elliptic-curve reduce op), and
  * the division carry: one Fr element per rank.
The data path has no other exchange.

Reference semantics restated: KZG10::commit (kzg10/mod.rs:157-210) and KZG10::open
(:287-310 -> compute_witness_polynomial :217-240, open_with_witness_polynomial :243-284),
hiding off.  The engine does the heavy lifting; `HipEngine` is the product engine (HIP
library through the C ABI, device-resident buffers).  Tests substitute an oracle-backed
engine to exercise this file's index/carry/fold logic on CPU-only machines.
"""
import os

import numpy as np

from . import _ffi

# scalar-field moduli (Fr) -- host glue only needs them to compose per-rank carries
FR_MODULUS = {
    "bls12_381": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "pallas": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
}
_R = 1 << 256


def _limbs_to_int(a):
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u8").tobytes(), "little")


def _int_to_limbs(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype="<u8").astype(np.uint64)


def all_gather_u64(dist, world, arr, cache=None):
    """Small uint64 numpy array -> (world, len) array, the same on every rank.  The exchange of this protocol is a few
    hundred bytes per rank (one point / one Fr): one all_gather_into_tensor on persistent buffers -- pinned host staging
    and a device tensor per payload length under RCCL, plain host tensors under gloo."""
    import torch
    flat = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
    if dist is None or world == 1:
        return flat.reshape(1, -1).copy()
    nccl = dist.get_backend() == "nccl"
    key = (flat.size, nccl)
    bufs = cache.get(key) if cache is not None else None
    if bufs is None:
        if nccl:
            bufs = (torch.empty(flat.size, dtype=torch.int64).pin_memory(), torch.empty(flat.size, dtype=torch.int64, device="cuda"),
                    torch.empty(world * flat.size, dtype=torch.int64, device="cuda"), torch.empty(world * flat.size, dtype=torch.int64).pin_memory())
        else:
            bufs = (torch.empty(flat.size, dtype=torch.int64), None, torch.empty(world * flat.size, dtype=torch.int64), None)
        if cache is not None:
            cache[key] = bufs
    h_in, d_in, d_out, h_out = bufs
    h_in.numpy().view(np.uint64)[:] = flat
    if nccl:
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        res = h_out
    else:
        dist.all_gather_into_tensor(d_out, h_in)
        res = d_out
    return res.numpy().view(np.uint64).reshape(world, -1).copy()


class HipEngine:
    """Device engine: everything stays in HBM; only 96-byte points / 32-byte carries return."""

    def __init__(self, ctx, curve):
        self.ctx, self.curve = ctx, curve
        self.srs = None
        self._scratch = None
        self.phases = []
        self.marks = []      # absolute phase boundaries of the same MSMs (Context.last_msm_marks_ms)
        self.blocking = False   # True: every MSM is the BLOCKING pc_hip_msm call (bench.py --inflight 0)

    glv_table = None     # None: the library's policy; True / False: force the GLV (half-size) / the full window table

    def load_srs(self, bases, precompute=False, n=None):
        """bases: host array of affine points, or a device pointer with `n` (a chunk generated on the device)."""
        self.srs = self.ctx.upload_srs(self.curve, bases if isinstance(bases, int) else np.ascontiguousarray(bases), n=n)
        self.precompute_ms = None
        if precompute:          # window table in HBM (pc_hip_srs_precompute): once per key, like the upload
            import time
            t0 = time.perf_counter()
            self.srs.precompute(glv=self.glv_table)
            self.precompute_ms = (time.perf_counter() - t0) * 1e3
        # The SRS's MSM pipelines (streams + workspace, GBs for a large SRS) are created on first use:
        # touch all of them now so that no later call pays for it.
        # (an empty MSM: the pipeline is created, nothing is launched)
        none = np.zeros((1, 4), dtype=np.uint64)
        for _ in range(3):
            self.srs.msm(none, n=0)

    def _ptr(self, buf, elem_off=0):
        if isinstance(buf, np.ndarray):
            raise TypeError("HipEngine works on device-resident buffers (torch cuda tensors)")
        return buf.data_ptr() + 32 * elem_off

    def msm(self, scalars, n, base_offset, elem_off=0):
        return self.msm_async(scalars, n, base_offset, elem_off).wait()

    def msm_async(self, scalars, n, base_offset, elem_off=0):
        """Queue the MSM on one of the SRS's pipelines; .wait() returns the affine point."""
        if self.blocking:
            out, _ = self.srs.msm(self._ptr(scalars, elem_off), n=n, base_offset=base_offset, montgomery=True)

            class _Done:
                def wait(self_inner):
                    return out
            return _Done()
        job = self.srs.msm_async(self._ptr(scalars, elem_off), n=n, base_offset=base_offset, montgomery=True)
        eng = self

        class _Pending:
            def wait(self_inner):
                out, _ = job.wait()
                if job.phases is not None:
                    eng.phases.append(job.phases)
                    eng.marks.append(job.marks)
                return out
        return _Pending()

    def div_scan(self, coeffs, n, z, carry_in):
        import torch
        # a small ring of quotient buffers: a queued open MSM still reads its quotient while the
        # next step's division runs
        if self._scratch is None or self._scratch[0].shape[0] < n:
            self._scratch = [torch.empty((n, 4), dtype=torch.int64, device=coeffs.device) for _ in range(4)]
            self._ring = 0
        buf = self._scratch[self._ring]
        self._ring = (self._ring + 1) % len(self._scratch)
        self.ctx.div_scan(self.curve, self._ptr(coeffs), z, carry_in, out=buf.data_ptr(), n=n)
        return buf

    def read_elem(self, buf, idx):
        return buf[idx].cpu().numpy().view(np.uint64).copy()

    def poly_eval(self, coeffs, n, z):
        """p(z) of this shard's n coefficients (one up-sweep on the device, 32 bytes back)."""
        return self.ctx.poly_eval(self.curve, self._ptr(coeffs), z, n=n)

    def points_sum(self, pts):
        return _ffi.points_sum(self.curve, pts)

    def msm_batch(self, polys, lens):
        """k device-resident scalar vectors against this rank's chunk (pc_hip_msm_batch) -> (k, 2*Fq) points."""
        return self.srs.msm_batch([self._ptr(p) for p in polys], list(lens))

    def ntt_rows(self, rows_buf, n_rows, in_cols, log_n, out_buf):
        self.ctx.ntt_batch(self.curve, self._ptr(rows_buf), log_n, out=self._ptr(out_buf), rows=n_rows, in_cols=in_cols)
        return out_buf

    # ---- column digests of a matrix whose rows are spread over the ranks (ShardedRows.commit) ----
    def state_buffer(self, n_cols):
        import torch
        return torch.zeros((n_cols, 12), dtype=torch.int32, device="cuda")      # pc_hip_column_hash_part: 48 bytes per column

    def digest_buffer(self, n_cols):
        import torch
        return torch.zeros((n_cols, 8), dtype=torch.int32, device="cuda")

    def column_hash_part(self, slab, rows, n_cols, rows_total, state, first, last, out, hash_name, col0, cols):
        self.ctx.column_hash_part(self.curve, self._ptr(slab) if rows else 0, rows, n_cols, rows_total, state.data_ptr(), first, last,
                                  out.data_ptr() if out is not None else 0, hash_name, col0, cols)

    def merkle_nodes(self, digests, n_leaves, tree_hash):
        """(2^h - 1, 32) uint8 node array (root at row 0) over the resident leaf digests."""
        h = max(1, (n_leaves - 1).bit_length())
        nodes = np.zeros(((1 << h) - 1, 32), dtype=np.uint8)
        self.ctx.merkle_tree(digests.data_ptr(), tree_hash, True, out=nodes, n_leaves=n_leaves)
        return nodes


class ShardedKzg:
    def __init__(self, engine, curve, rank=0, world=1, dist=None):
        self.e, self.curve, self.rank, self.world, self.dist = engine, curve, rank, world, dist
        self.p = FR_MODULUS[curve]
        self.z = None
        self.last_phases = []
        self._coll = {}
        self.exchange_ms = {"wait_local_msm": 0.0, "shard_eval": 0.0, "all_gather": 0.0, "calls": 0}

    # bases: n+1 affine points; bases[0] = the power just below this chunk (unused on rank 0),
    # bases[1 + j] = power r*n + j.
    def load_srs_chunk(self, bases, **kw):
        self.e.load_srs(bases, **kw)

    def set_point(self, z_mont):
        self.z = np.ascontiguousarray(z_mont, dtype=np.uint64)

    # ---- collectives -----------------------------------------------------------------------
    def _all_gather(self, arr):
        """arr: small uint64 numpy array -> (world, len) array, same on every rank."""
        if self.dist is None:
            return arr.reshape(1, -1)
        return all_gather_u64(self.dist, self.world, arr, self._coll)

    def _combine_points(self, local_xy):
        pts = self._all_gather(local_xy)
        if self.world == 1:
            return pts[0]
        return self.e.points_sum(pts)

    # ---- KZG10::commit ---------------------------------------------------------------------
    def commit(self, coeffs, n):
        return self.commit_async(coeffs, n).result()

    def commit_async(self, coeffs, n):
        return _Future(self, self._msm_async(coeffs, n, base_offset=1))

    def _msm_async(self, buf, n, base_offset, elem_off=0):
        if hasattr(self.e, "msm_async"):
            return self.e.msm_async(buf, n, base_offset, elem_off)
        val = self.e.msm(buf, n, base_offset, elem_off)      # engines without pipelines (test doubles)

        class _Done:
            def wait(self_inner):
                return val
        return _Done()

    # ---- KZG10::open -----------------------------------------------------------------------
    def open(self, coeffs, n):
        return self.open_async(coeffs, n).result()

    def open_prepare(self, coeffs, n):
        """The exchange step of a sharded open, separated from the enqueue so that a caller can run it while
        earlier MSMs are still in flight (bench.py does: evaluation + all_gather of step k overlap the open
        MSM of step k-1): carry into shard r = composition of the shards above it, c = B_s + z^n * c with
        B_s = p_s(z) (an evaluation-only up-sweep; the full division then runs once, with the carry)."""
        if self.world == 1:
            return None
        return self._carry_from_evals(self._all_gather(self.e.poly_eval(coeffs, n, self.z)), n)

    def _carry_from_evals(self, b, n):
        """carry into this rank's shard from the gathered shard evaluations b[s] = p_s(z) (Montgomery limbs)."""
        z = _limbs_to_int(self.z) * pow(_R, -1, self.p) % self.p
        zn_mont = pow(z, n, self.p) * _R % self.p          # Montgomery form of z^n
        rinv = pow(_R, -1, self.p)
        carry = 0
        for s in range(self.world - 1, self.rank, -1):
            carry = (_limbs_to_int(b[s]) + zn_mont * carry * rinv) % self.p   # Montgomery arithmetic
        return _int_to_limbs(carry) if carry else None

    def exchange(self, coeffs=None, n=0, futures=()):
        """ONE collective for everything a pipelined prover has to exchange at a step boundary: the shard evaluation
        of the polynomial about to be opened (-> this rank's division carry, as open_prepare) and the partial points
        of earlier commits / opens whose local MSMs are awaited here (-> their combined results, as _Future.result).
        Returns (carry or None, [combined point per future])."""
        import time
        futures = list(futures)
        t0 = time.perf_counter()
        local = [f.pending.wait() for f in futures]
        t1 = time.perf_counter()
        if self.world == 1 and self.dist is None:
            return None, local
        parts = []
        if coeffs is not None:
            ev = np.zeros(4, dtype=np.uint64) if (self.world == 1 and os.environ.get("PC_DIAG_SKIP_SHARD_EVAL")) else self.e.poly_eval(coeffs, n, self.z)
            parts.append(np.ascontiguousarray(ev, dtype=np.uint64).reshape(-1))
        t2 = time.perf_counter()
        parts += [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1) for x in local]
        if not parts:
            return None, []
        g = self._all_gather(np.concatenate(parts))
        t3 = time.perf_counter()
        tm = self.exchange_ms                      # host-side time by part (bench.py reports it for N > 1)
        tm["wait_local_msm"] += (t1 - t0) * 1e3; tm["shard_eval"] += (t2 - t1) * 1e3; tm["all_gather"] += (t3 - t2) * 1e3; tm["calls"] += 1
        off, carry = 0, None
        if coeffs is not None:
            carry = self._carry_from_evals(g[:, :4], n) if self.world > 1 else None
            off = 4
        out = []
        for x in local:
            w = x.size
            pts = np.ascontiguousarray(g[:, off:off + w])
            out.append(pts[0] if self.world == 1 else self.e.points_sum(pts))
            off += w
        return carry, out

    def open_async(self, coeffs, n, prepared=False, carry=None):
        if not prepared:
            carry = self.open_prepare(coeffs, n)
        out = self.e.div_scan(coeffs, n, self.z, carry)
        if self.rank == 0:
            pend = self._msm_async(out, n - 1, base_offset=1, elem_off=1)    # q[i-1] = out[i] pairs with power i-1
        else:
            pend = self._msm_async(out, n, base_offset=0)                    # out[j] pairs with power r*n + j - 1
        return _Future(self, pend)


class ShardedBatch:
    """BASELINE configs[2]: k polynomials committed against ONE SRS that is split into `world` contiguous chunks.
    Every rank runs the k partial MSMs of its chunk as one pipelined batch; the k partial points per rank are
    combined with one all_gather (k x 64..96 B per rank) + EC adds.  (MarlinKZG10::commit's loop,
    marlin_pc/mod.rs:192-237; RCCL has no EC reduce op and raw bucket arrays are never exchanged.)"""

    def __init__(self, engine, curve, rank=0, world=1, dist=None):
        self.e, self.curve, self.rank, self.world, self.dist = engine, curve, rank, world, dist
        self._coll = {}

    @staticmethod
    def chunk_range(total, rank, world):
        per = (total + world - 1) // world
        return min(total, rank * per), min(total, (rank + 1) * per)

    def load_srs_chunk(self, bases, **kw):
        self.e.load_srs(bases, **kw)

    def commit_batch(self, polys, lens):
        """polys: this rank's slices of the k coefficient vectors (engine buffers); returns (k, 2*Fq) points."""
        part = self.e.msm_batch(polys, lens)
        if self.dist is None or self.world == 1:
            return part
        allp = all_gather_u64(self.dist, self.world, part, self._coll).reshape(self.world, len(polys), -1)
        return np.stack([self.e.points_sum(np.ascontiguousarray(allp[:, j])) for j in range(len(polys))])


class ShardedRows:
    """BASELINE configs[4]: the rows of Ligero's coefficient matrix are independent (linear_codes/mod.rs:131-135):
    rank r encodes rows [r * per, (r+1) * per); no collective on the data path."""

    def __init__(self, engine, rank=0, world=1):
        self.e, self.rank, self.world = engine, rank, world

    def row_range(self, n_rows, rank=None):
        # an even number of rows per rank (two 32-byte rows fill one 64-byte block of the column digests: commit() below)
        per = (n_rows + self.world - 1) // self.world
        per += per & 1 if self.world > 1 else 0
        r = self.rank if rank is None else rank
        return min(n_rows, r * per), min(n_rows, (r + 1) * per)

    def encode(self, rows_buf, n_rows_local, in_cols, log_n, out_buf=None):
        return self.e.ntt_rows(rows_buf, n_rows_local, in_cols, log_n, out_buf)

    def commit(self, ext_slab, n_rows, n_ext_cols, dist, col_hash="blake2s", tree_hash="sha256", blocks=4):
        """Steps 2-3 of LinearCodePCS::commit (linear_codes/mod.rs:256-277) over an encoded matrix whose ROWS live on different
        ranks (`ext_slab`: this rank's rows of row_range(n_rows), the output of encode()).  A column's digest needs all of its
        rows; instead of transposing the matrix between the devices (2 GiB at BASELINE configs[4]) the digests' chaining states
        travel: rank r absorbs its slab into the states rank r - 1 left (pc_hip_column_hash_part) and hands on 48 bytes per
        column, in `blocks` column ranges so that rank r + 1 works on range b while rank r is on range b + 1.  The last rank
        that holds rows finishes the digests, builds the Merkle tree and broadcasts the root.
        Returns (root: 32 bytes, nodes: the whole node array on the rank that built it, else None)."""
        import torch
        e = self.e
        wire = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"
        if n_rows <= 0:
            raise ValueError("ShardedRows.commit: the matrix has no rows (LinearCodePCS::commit never builds an empty matrix)")
        ranges = [self.row_range(n_rows, r) for r in range(self.world)]
        active = [r for r in range(self.world) if ranges[r][1] > ranges[r][0]]
        lo, hi = ranges[self.rank]
        root = torch.zeros(32, dtype=torch.uint8, device=wire)
        nodes = None
        if self.rank in active:
            k = active.index(self.rank)
            first, last = k == 0, k == len(active) - 1
            state = e.state_buffer(n_ext_cols)
            digests = e.digest_buffer(n_ext_cols) if last else None
            on_device = hasattr(state, "is_cuda") and state.is_cuda
            if on_device:            # torch filled the buffers on ITS stream; the library's kernels run on the context's
                torch.cuda.synchronize()
            nb = max(1, min(blocks, n_ext_cols))
            for b in range(nb):
                c0, c1 = n_ext_cols * b // nb, n_ext_cols * (b + 1) // nb
                if c1 == c0:
                    continue
                if not first:
                    buf = torch.empty((c1 - c0, 12), dtype=torch.int32, device=wire)
                    dist.recv(buf, src=active[k - 1])
                    state[c0:c1].copy_(buf)
                    if on_device:    # (RCCL's receive and the copy are stream-ordered on torch's side only)
                        torch.cuda.synchronize()
                e.column_hash_part(ext_slab, hi - lo, n_ext_cols, n_rows, state, first, last, digests, col_hash, c0, c1 - c0)
                if not last:
                    dist.send(state[c0:c1].to(wire).contiguous(), dst=active[k + 1])
            if last:
                nodes = e.merkle_nodes(digests, n_ext_cols, tree_hash)
                root = torch.from_numpy(nodes[0].copy()).to(wire)
        if dist is not None and self.world > 1:
            dist.broadcast(root, src=active[-1])
        return root.cpu().numpy(), nodes


class _Future:
    """Result of a queued commit/open: .result() waits for the local MSM and folds the ranks."""

    def __init__(self, job, pending):
        self.job, self.pending = job, pending

    def result(self):
        local = self.pending.wait()
        return self.job._combine_points(local)
