"""ctypes binding of include/pc_hip.h.  No fallbacks: if the HIP library is missing or no
GPU is present, construction fails loudly."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CURVES = {"bls12_381": 0, "bn254": 1, "pallas": 2}
FQ_BYTES = {"bls12_381": 48, "bn254": 32, "pallas": 32}
PC_MEM_HOST, PC_MEM_DEVICE = 0, 1
PC_SCALARS_CANONICAL, PC_SCALARS_MONTGOMERY = 0, 1

_lib = None


class PcHipError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__(f"pc_hip status {status}: {msg}")
        self.status = status


def library_path():
    # PC_HIP_LIB: alternative build of the same library (kernel tuning experiments)
    return os.environ.get("PC_HIP_LIB") or os.path.join(HERE, "libpc_hip.so")


def load_library():
    """Load libpc_hip.so (built by poly_commit_amd/build.py).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # The PyTorch wheel bundles its own ROCm runtime; if libpc_hip.so pulls in the system
    # libamdhip64 first, torch later finds "No HIP GPUs".  Let torch bring its runtime up first
    # (harness concern only -- a Rust/C++ consumer links one runtime).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise PcHipError(-100, f"{path} not built: run python -m poly_commit_amd.build "
                               "(or __graft_entry__.build())")
    lib = C.CDLL(path)
    vp, sz, ip = C.c_void_p, C.c_size_t, C.c_int
    lib.pc_hip_device_count.restype = ip
    lib.pc_hip_init.argtypes = [ip, C.POINTER(vp)]
    lib.pc_hip_shutdown.argtypes = [vp]
    lib.pc_hip_shutdown.restype = None
    lib.pc_hip_strerror.argtypes = [ip]
    lib.pc_hip_strerror.restype = C.c_char_p
    lib.pc_hip_last_error.argtypes = [vp]
    lib.pc_hip_last_error.restype = C.c_char_p
    lib.pc_hip_srs_upload.argtypes = [vp, ip, vp, sz, sz, ip, C.POINTER(vp)]
    lib.pc_hip_srs_load_serialized.argtypes = [vp, ip, vp, sz, ip, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    lib.pc_hip_srs_serialize.argtypes = [vp, vp, sz, sz, ip, vp, sz, C.POINTER(sz)]
    lib.pc_hip_universal_params_layout.argtypes = [ip, vp, sz, ip, C.POINTER(sz)]
    lib.pc_hip_srs_free.argtypes = [vp]
    lib.pc_hip_srs_precompute.argtypes = [vp, vp, C.c_uint, sz]
    lib.pc_hip_srs_free.restype = None
    lib.pc_hip_srs_len.argtypes = [vp]
    lib.pc_hip_srs_len.restype = sz
    lib.pc_hip_srs_device_ptr.argtypes = [vp]
    lib.pc_hip_srs_device_ptr.restype = vp
    lib.pc_hip_msm.argtypes = [vp, vp, sz, vp, ip, ip, sz, vp, C.POINTER(ip)]
    lib.pc_hip_msm_batch.argtypes = [vp, vp, C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), sz, ip, ip, vp,
                                     C.POINTER(ip)]
    lib.pc_hip_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.pc_hip_free.argtypes = [vp, vp]
    lib.pc_hip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.pc_hip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    lib.pc_hip_msm_many.argtypes = [vp, vp, sz, vp, ip, ip, sz, sz, vp, C.POINTER(ip)]
    lib.pc_hip_msm_async.argtypes = [vp, vp, sz, vp, ip, ip, sz, vp, C.POINTER(ip), C.POINTER(vp)]
    lib.pc_hip_job_wait.argtypes = [vp, vp]
    lib.pc_hip_set_msm_tuning.argtypes = [vp, C.c_uint, C.c_uint]
    lib.pc_hip_set_timing.argtypes = [vp, ip]
    lib.pc_hip_srs_precompute_ex.argtypes = [vp, vp, C.c_uint, sz, C.c_uint]
    lib.pc_hip_srs_bytes_resident.argtypes = [vp, C.POINTER(sz)]
    lib.pc_hip_ctx_bytes_resident.argtypes = [vp, C.POINTER(sz)]
    lib.pc_hip_ctx_trim.argtypes = [vp]
    lib.pc_hip_last_msm_phases_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pc_hip_last_msm_marks_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pc_hip_last_msm_shape.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.pc_hip_matrix_columns.argtypes = [vp, vp, sz, sz, vp, sz, vp, ip]
    lib.pc_hip_ntt_batch.argtypes = [vp, ip, vp, ip, sz, sz, C.c_uint, vp, ip]
    lib.pc_hip_last_ntt_phases_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pc_hip_witness_poly.argtypes = [vp, ip, vp, ip, sz, vp, vp, ip]
    lib.pc_hip_kzg_open.argtypes = [vp, vp, sz, vp, ip, sz, vp, vp, C.POINTER(C.c_int)]
    lib.pc_hip_column_hash.argtypes = [vp, ip, vp, ip, sz, sz, ip, vp, ip]
    lib.pc_hip_column_hash_part.argtypes = [vp, ip, ip, vp, sz, sz, sz, sz, sz, ip, ip, vp, vp]
    lib.pc_hip_merkle_tree.argtypes = [vp, ip, vp, ip, sz, ip, vp, ip]
    lib.pc_hip_ligero_commit.argtypes = [vp, ip, vp, ip, sz, sz, C.c_uint, ip, ip, ip, vp, ip, vp, vp]
    lib.pc_hip_last_ligero_phases_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pc_hip_fr_lincomb.argtypes = [vp, ip, C.POINTER(vp), ip, C.POINTER(sz), sz, vp, vp, ip, sz]
    lib.pc_hip_fr_fold.argtypes = [vp, ip, vp, vp, sz, vp]
    lib.pc_hip_fr_dot.argtypes = [vp, ip, vp, vp, sz, vp]
    lib.pc_hip_ipa_fold_dots.argtypes = [vp, ip, vp, vp, sz, vp, vp, vp]
    lib.pc_hip_fr_powers.argtypes = [vp, ip, vp, sz, vp]
    lib.pc_hip_ec_fold.argtypes = [vp, vp, sz, vp]
    lib.pc_hip_ec_fold_from.argtypes = [vp, vp, sz, vp, C.POINTER(vp)]
    lib.pc_hip_srs_precompute_fold.argtypes = [vp, vp]
    lib.pc_hip_srs_precompute_fold_ex.argtypes = [vp, vp, C.c_uint, C.c_uint]
    lib.pc_hip_srs_fold_table_info.argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.pc_hip_ec_fold2_from.argtypes = [vp, vp, sz, vp, vp, C.POINTER(vp)]
    lib.pc_hip_ipa_round2_msms.argtypes = [vp, vp, vp, sz, vp, vp, C.POINTER(ip), vp, C.POINTER(ip)]
    lib.pc_hip_ipa_open_rounds.argtypes = [vp, vp, vp, sz, vp, vp, IPA_CHALLENGE_FN, vp, sz, vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pc_hip_ipa_key_scalars.argtypes = [vp, ip, vp, sz, vp, sz, vp, sz, vp, vp]
    lib.pc_hip_srs_read.argtypes = [vp, vp, sz, sz, vp]
    lib.pc_hip_fixed_base_batch_mul.argtypes = [vp, ip, vp, vp, sz, vp]
    lib.pc_hip_poly_eval.argtypes = [vp, ip, vp, ip, sz, vp, vp]
    lib.pc_hip_poly_div_scan.argtypes = [vp, ip, vp, ip, sz, vp, vp, vp, ip]
    lib.pc_hip_points_sum.argtypes = [ip, vp, sz, vp]
    lib.pc_hip_point_mul.argtypes = [ip, vp, vp, vp]
    lib.pc_hip_group_create.argtypes = [C.POINTER(ip), ip, C.POINTER(vp)]
    lib.pc_hip_group_destroy.argtypes = [vp]
    lib.pc_hip_group_destroy.restype = None
    lib.pc_hip_group_size.argtypes = [vp]
    lib.pc_hip_group_ctx.argtypes = [vp, ip]
    lib.pc_hip_group_ctx.restype = vp
    lib.pc_hip_group_srs_upload.argtypes = [vp, ip, vp, sz, sz, ip, C.POINTER(vp)]
    lib.pc_hip_group_srs_free.argtypes = [vp]
    lib.pc_hip_group_srs_free.restype = None
    lib.pc_hip_group_srs_len.argtypes = [vp]
    lib.pc_hip_group_srs_len.restype = sz
    lib.pc_hip_group_msm.argtypes = [vp, vp, sz, vp, ip, sz, vp, C.POINTER(ip)]
    lib.pc_hip_group_msm_batch.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(sz), sz, ip, vp, C.POINTER(ip)]
    lib.pc_hip_group_kzg_open.argtypes = [vp, vp, vp, sz, vp, vp, C.POINTER(ip), vp]
    lib.pc_hip_group_ntt_batch.argtypes = [vp, ip, vp, sz, sz, C.c_uint, vp]
    lib.pc_hip_group_ligero_commit.argtypes = [vp, ip, vp, sz, sz, C.c_uint, ip, ip, ip, vp, vp, vp]
    lib.pc_hip_group_commit_open_async.argtypes = [vp, vp, vp, ip, sz, vp, vp, vp, vp, C.POINTER(vp)]
    lib.pc_hip_group_job_wait.argtypes = [vp, vp]
    _lib = lib
    return lib


# pc_ipa_challenge_fn: void (*)(void* user, const void* l_xy, const void* r_xy, void* out_u_mont)
IPA_CHALLENGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def _ptr(x):
    """host numpy array -> (void*, PC_MEM_HOST); torch cuda tensor / int -> (void*, PC_MEM_DEVICE)."""
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return C.c_void_p(x.ctypes.data), PC_MEM_HOST
    if isinstance(x, int):
        return C.c_void_p(x), PC_MEM_DEVICE
    if hasattr(x, "data_ptr"):   # torch tensor
        assert x.is_contiguous()
        return C.c_void_p(x.data_ptr()), (PC_MEM_DEVICE if x.is_cuda else PC_MEM_HOST)
    raise TypeError(type(x))


class Context:
    """One GPU (pc_ctx)."""

    def __init__(self, device_id=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.pc_hip_init(device_id, C.byref(h))
        if rc != 0:
            raise PcHipError(rc, self.lib.pc_hip_strerror(rc).decode())
        self.h = h
        self.device_id = device_id

    def check(self, rc):
        if rc != 0:
            raise PcHipError(rc, self.lib.pc_hip_strerror(rc).decode() + " / " +
                             self.lib.pc_hip_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.pc_hip_shutdown(self.h)
            self.h = None

    def set_msm_tuning(self, window_bits=0, chunk=0):
        self.check(self.lib.pc_hip_set_msm_tuning(self.h, window_bits, chunk))

    def set_timing(self, on=True):
        self.check(self.lib.pc_hip_set_timing(self.h, 1 if on else 0))

    def last_msm_phases_ms(self):
        out = (C.c_float * 8)()
        self.check(self.lib.pc_hip_last_msm_phases_ms(self.h, out))
        return list(out)

    def last_msm_marks_ms(self):
        """Phase boundaries of the last completed MSM as offsets from the last set_timing(True) (pc_hip_last_msm_marks_ms)."""
        out = (C.c_float * 8)()
        self.check(self.lib.pc_hip_last_msm_marks_ms(self.h, out))
        return list(out)

    def last_msm_shape(self):
        """{window bits, signed digits per scalar, buckets, window table used} of the last completed MSM."""
        out = (C.c_uint32 * 4)()
        self.check(self.lib.pc_hip_last_msm_shape(self.h, out))
        return {"window_bits": out[0], "digits_per_scalar": out[1], "buckets": out[2], "window_table": bool(out[3])}

    def ntt_batch(self, curve, mat, log_n, out=None, rows=None, in_cols=None):
        """rows x in_cols Fr (Montgomery) -> rows x 2^log_n, natural order.  numpy in -> numpy out;
        torch cuda tensors (or raw device pointers with rows/in_cols) stay on the device."""
        pin, win = _ptr(mat)
        if rows is None:
            rows, in_cols = mat.shape[0], mat.shape[1]
        if out is None:
            assert win == PC_MEM_HOST
            out = np.zeros((rows, 1 << log_n, 4), dtype=np.uint64)
        pout, wout = _ptr(out)
        self.check(self.lib.pc_hip_ntt_batch(self.h, CURVES[curve], pin, win, rows, in_cols, log_n, pout, wout))
        return out

    def column_hash(self, curve, ext, hash_name="blake2s", out=None, rows=None, n_cols=None):
        """Digests of the columns of the encoded matrix (rows x n_cols x Fr, Montgomery):
        FieldToBytesColHasher<F, D>, D = 'sha256' | 'blake2s'.  Returns (n_cols, 32) uint8 for host
        input; device buffers (raw pointers + rows/n_cols, out = device pointer) stay on the device."""
        pin, win = _ptr(ext)
        if rows is None:
            rows, n_cols = ext.shape[0], ext.shape[1]
        if out is None:
            assert win == PC_MEM_HOST
            out = np.zeros((n_cols, 32), dtype=np.uint8)
        pout, wout = _ptr(out)
        hid = {"sha256": 0, "blake2s": 1}[hash_name]
        self.check(self.lib.pc_hip_column_hash(self.h, CURVES[curve], pin, win, rows, n_cols, hid, pout, wout))
        return out

    def column_hash_part(self, curve, slab_dev, rows, n_cols, rows_total, state_dev, first, last, out_dev=0, hash_name="blake2s",
                         col0=0, cols=None):
        """One slab of rows absorbed into the per-column chaining states (pc_hip_column_hash_part): device pointers only.
        state_dev: n_cols x 48 bytes; out_dev: n_cols x 32 bytes, written when `last`."""
        hid = {"sha256": 0, "blake2s": 1}[hash_name]
        self.check(self.lib.pc_hip_column_hash_part(self.h, CURVES[curve], hid, C.c_void_p(slab_dev), rows, n_cols, rows_total, col0,
                                                    n_cols - col0 if cols is None else cols, 1 if first else 0, 1 if last else 0,
                                                    C.c_void_p(state_dev), C.c_void_p(out_dev)))

    def matrix_columns(self, mat_dev, rows, n_cols, indices):
        """Columns `indices` of a resident rows x n_cols matrix of 32-byte elements -> (t, rows, 4) uint64 host array."""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        out = np.zeros((len(idx), rows, 4), dtype=np.uint64)
        self.check(self.lib.pc_hip_matrix_columns(self.h, mat_dev, rows, n_cols, idx.ctypes.data, len(idx), out.ctypes.data, PC_MEM_HOST))
        return out

    def merkle_tree(self, digests, hash_name="sha256", len_prefix=True, out=None, n_leaves=None):
        """Inner nodes of the Merkle tree over 32-byte leaf digests (create_merkle_tree,
        linear_codes/mod.rs:506-521), heap order, root at row 0: (2^h - 1, 32) uint8 for host input;
        device pointers (n_leaves given, out = device pointer) stay on the device."""
        pin, win = _ptr(digests)
        if n_leaves is None:
            n_leaves = digests.shape[0]
        h = max(1, (n_leaves - 1).bit_length())
        if out is None:
            assert win == PC_MEM_HOST
            out = np.zeros(((1 << h) - 1, 32), dtype=np.uint8)
        pout, wout = _ptr(out)
        hid = {"sha256": 0, "blake2s": 1}[hash_name]
        self.check(self.lib.pc_hip_merkle_tree(self.h, hid, pin, win, n_leaves, 1 if len_prefix else 0, pout, wout))
        return out

    def fr_lincomb(self, curve, polys, xi, n_out=None, out=None, lens=None):
        """sum_j xi[j] * polys[j] (MarlinKZG10::open's combination, marlin_pc/mod.rs:281-287).
        polys: list of (len_j, 4) uint64 host arrays, or device pointers with `lens`; xi: (k, 4)."""
        k = len(polys)
        ptrs = [_ptr(p) for p in polys]
        where = ptrs[0][1] if k else PC_MEM_HOST
        assert all(w == where for _, w in ptrs)
        if lens is None:
            lens = [p.shape[0] for p in polys]
        if n_out is None:
            n_out = max(lens) if k else 0
        if out is None:
            assert where == PC_MEM_HOST
            out = np.zeros((n_out, 4), dtype=np.uint64)
        pout, wout = _ptr(out)
        arr = (C.c_void_p * max(k, 1))(*[p for p, _ in ptrs])
        larr = (C.c_size_t * max(k, 1))(*lens)
        xi = np.ascontiguousarray(xi, dtype=np.uint64)
        self.check(self.lib.pc_hip_fr_lincomb(self.h, CURVES[curve], arr, where, larr, k, xi.ctypes.data, pout, wout, n_out))
        return out

    def ligero_commit(self, curve, mat, log_n, col_hash="blake2s", tree_hash="sha256", len_prefix=True,
                      rows=None, in_cols=None, ext_out=None, want_leaves=True):
        """LinearCodePCS::commit steps 1-3 (linear_codes/mod.rs:248-277) in one call: returns
        (nodes (2^h - 1, 32) uint8 with the root at row 0, leaves (2^log_n, 32) uint8 or None)."""
        pin, win = _ptr(mat)
        if rows is None:
            rows, in_cols = mat.shape[0], mat.shape[1]
        hid = {"sha256": 0, "blake2s": 1}
        n = 1 << log_n
        nodes = np.zeros(((1 << max(1, log_n)) - 1, 32), dtype=np.uint8)
        leaves = np.zeros((n, 32), dtype=np.uint8) if want_leaves else None
        pext, wext = _ptr(ext_out) if ext_out is not None else (None, PC_MEM_HOST)
        self.check(self.lib.pc_hip_ligero_commit(self.h, CURVES[curve], pin, win, rows, in_cols, log_n, hid[col_hash],
                                                 hid[tree_hash], 1 if len_prefix else 0, pext, wext,
                                                 leaves.ctypes.data if want_leaves else None, nodes.ctypes.data))
        return nodes, leaves

    def last_ligero_phases_ms(self):
        out = (C.c_float * 4)()
        self.check(self.lib.pc_hip_last_ligero_phases_ms(self.h, out))
        return list(out)

    def last_ntt_phases_ms(self):
        out = (C.c_float * 2)()
        self.check(self.lib.pc_hip_last_ntt_phases_ms(self.h, out))
        return list(out)

    # raw device buffers of the library (pc_hip_malloc / pc_hip_free / pc_hip_memcpy_*): what a shim caches a polynomial in
    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.pc_hip_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free_dev(self, ptr):
        self.check(self.lib.pc_hip_free(self.h, C.c_void_p(ptr)))

    def memcpy_h2d(self, dst_dev, src_host):
        src = np.ascontiguousarray(src_host)
        self.check(self.lib.pc_hip_memcpy_h2d(self.h, C.c_void_p(dst_dev), C.c_void_p(src.ctypes.data), src.nbytes))

    def witness_poly(self, curve, coeffs, z, out=None, n=None):
        """q = p / (x - z); coeffs n x 4 uint64 (Montgomery), z 4 x uint64 host array."""
        pin, win = _ptr(coeffs)
        if n is None:
            n = coeffs.shape[0]
        if out is None:
            assert win == PC_MEM_HOST
            out = np.zeros((max(n - 1, 1), 4), dtype=np.uint64)
        pout, wout = _ptr(out)
        z = np.ascontiguousarray(z, dtype=np.uint64)
        self.check(self.lib.pc_hip_witness_poly(self.h, CURVES[curve], pin, win, n, C.c_void_p(z.ctypes.data), pout, wout))
        return out[: max(n - 1, 0)] if isinstance(out, np.ndarray) else out

    def poly_eval(self, curve, coeffs, z, n=None):
        """p(z) of n coefficients (host array or device pointer); returns the Montgomery value, (4,) uint64."""
        pin, win = _ptr(coeffs)
        if n is None:
            n = coeffs.shape[0]
        z = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.check(self.lib.pc_hip_poly_eval(self.h, CURVES[curve], pin, win, n, z.ctypes.data, out.ctypes.data))
        return out

    def div_scan(self, curve, coeffs, z, carry_in=None, out=None, n=None):
        """acc = carry_in; for i = n-1..0: acc = coeffs[i] + z*acc; out[i] = acc  (n outputs)."""
        pin, win = _ptr(coeffs)
        if n is None:
            n = coeffs.shape[0]
        if out is None:
            assert win == PC_MEM_HOST
            out = np.zeros((n, 4), dtype=np.uint64)
        pout, wout = _ptr(out)
        z = np.ascontiguousarray(z, dtype=np.uint64)
        cin = None
        if carry_in is not None:
            carry_in = np.ascontiguousarray(carry_in, dtype=np.uint64)
            cin = C.c_void_p(carry_in.ctypes.data)
        self.check(self.lib.pc_hip_poly_div_scan(self.h, CURVES[curve], pin, win, n, C.c_void_p(z.ctypes.data), cin,
                                                 pout, wout))
        return out

    # ---- IPA round primitives (device-resident vectors; `dev` = raw device pointer) --------------
    def fr_fold(self, curve, lo_dev, hi_dev, n_half, s):
        s = np.ascontiguousarray(s, dtype=np.uint64)
        self.check(self.lib.pc_hip_fr_fold(self.h, CURVES[curve], lo_dev, hi_dev, n_half, C.c_void_p(s.ctypes.data)))

    def fr_dot(self, curve, a_dev, b_dev, n):
        out = np.zeros(4, dtype=np.uint64)
        self.check(self.lib.pc_hip_fr_dot(self.h, CURVES[curve], a_dev, b_dev, n, C.c_void_p(out.ctypes.data)))
        return out

    def ipa_fold_dots(self, curve, coeffs_dev, z_dev, m, u=None, u_inv=None):
        """pc_hip_ipa_fold_dots: optional fold at size 2m by (u, u^-1), then the round's two inner products at size m -> (2, 4) uint64."""
        out = np.zeros((2, 4), dtype=np.uint64)
        pu = pi = None
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.uint64); u_inv = np.ascontiguousarray(u_inv, dtype=np.uint64)
            pu, pi = C.c_void_p(u.ctypes.data), C.c_void_p(u_inv.ctypes.data)
        self.check(self.lib.pc_hip_ipa_fold_dots(self.h, CURVES[curve], coeffs_dev, z_dev, m, pu, pi, C.c_void_p(out.ctypes.data)))
        return out

    def fr_powers(self, curve, z, n, out_dev):
        z = np.ascontiguousarray(z, dtype=np.uint64)
        self.check(self.lib.pc_hip_fr_powers(self.h, CURVES[curve], C.c_void_p(z.ctypes.data), n, out_dev))

    def ipa_key_scalars(self, curve, coeffs_dev, m, s_dev, n0, fold_u=None, fold_m=0, out_l_dev=None, out_r_dev=None):
        """pc_hip_ipa_key_scalars: fold the key factors s by fold_u (optional), build the round's MSM scalars (optional)."""
        fu = None
        if fold_u is not None:
            fold_u = np.ascontiguousarray(fold_u, dtype=np.uint64)
            fu = C.c_void_p(fold_u.ctypes.data)
        self.check(self.lib.pc_hip_ipa_key_scalars(self.h, CURVES[curve], coeffs_dev, m, s_dev, n0, fu, fold_m, out_l_dev, out_r_dev))

    def fixed_base_batch_mul(self, curve, g_xy, scalars_dev, n, out_dev):
        """out[i] = scalars[i] * g (device buffers): the SRS generation of KZG10::setup."""
        g_xy = np.ascontiguousarray(g_xy, dtype=np.uint64)
        self.check(self.lib.pc_hip_fixed_base_batch_mul(self.h, CURVES[curve], C.c_void_p(g_xy.ctypes.data), scalars_dev, n, out_dev))

    def upload_srs(self, curve, bases, n=None, stride_bytes=0):
        return Srs(self, curve, bases, n, stride_bytes)

    def bytes_resident(self):
        """pc_hip_ctx_bytes_resident: device bytes by kind."""
        out = (C.c_size_t * 6)()
        self.check(self.lib.pc_hip_ctx_bytes_resident(self.h, out))
        return dict(zip(("device_total", "keys", "window_tables", "fold_tables", "scratch", "n_keys"), [int(x) for x in out]))

    def trim(self):
        self.check(self.lib.pc_hip_ctx_trim(self.h))

    def load_serialized_srs(self, curve, data, compressed, max_points=0):
        """Resident SRS from the ark-serialize bytes of a Vec<G1Affine> (the head of kzg10::UniversalParams):
        returns (Srs, bytes_consumed)."""
        h, npts, used = C.c_void_p(), C.c_size_t(), C.c_size_t()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        self.check(self.lib.pc_hip_srs_load_serialized(self.h, CURVES[curve], buf, len(data), 1 if compressed else 0, max_points,
                                                       C.byref(h), C.byref(npts), C.byref(used)))
        srs = Srs.__new__(Srs)
        srs.ctx, srs.curve, srs.h, srs.n = self, curve, h, npts.value
        return srs, used.value


class Srs:
    """Resident bases (pc_srs): powers_of_g of a KZG committer key, or an IPA comm_key."""

    def __init__(self, ctx, curve, bases, n=None, stride_bytes=0):
        self.ctx, self.curve = ctx, curve
        p, where = _ptr(bases)
        if n is None:
            n = bases.shape[0]
        h = C.c_void_p()
        ctx.check(ctx.lib.pc_hip_srs_upload(ctx.h, CURVES[curve], p, n, stride_bytes, where, C.byref(h)))
        self.h, self.n = h, n

    def precompute(self, window_bits=0, min_pairs=0, glv=None):
        """Build the window table of this SRS in HBM (pc_hip_srs_precompute).  glv=True: the half-size table over the GLV halves of
        the scalars (pc_hip_srs_precompute_ex, PC_HIP_TABLE_GLV); False: the full table; None: the library's policy."""
        if glv is None:
            self.ctx.check(self.ctx.lib.pc_hip_srs_precompute(self.ctx.h, self.h, window_bits, min_pairs))
        else:
            self.ctx.check(self.ctx.lib.pc_hip_srs_precompute_ex(self.ctx.h, self.h, window_bits, min_pairs, 1 if glv else 0))
        return self

    def free(self):
        if self.h:
            self.ctx.lib.pc_hip_srs_free(self.h)
            self.h = None

    def msm(self, scalars, n=None, base_offset=0, montgomery=False):
        """sum scalars[i] * bases[base_offset + i]; returns (xy uint64 array, is_infinity)."""
        p, where = _ptr(scalars)
        if n is None:
            n = scalars.shape[0]
        out = np.zeros(2 * FQ_BYTES[self.curve] // 8, dtype=np.uint64)
        inf = C.c_int(0)
        self.ctx.check(self.ctx.lib.pc_hip_msm(self.ctx.h, self.h, base_offset, p,
                                               PC_SCALARS_MONTGOMERY if montgomery else PC_SCALARS_CANONICAL,
                                               where, n, C.c_void_p(out.ctypes.data), C.byref(inf)))
        return out, bool(inf.value)

    def kzg_open(self, coeffs, z, n=None, base_offset=0):
        """KZG10::open without hiding as one call (pc_hip_kzg_open): W = MSM(bases[base_offset ..], p / (x - z)); coeffs: (n, 4)
        uint64 host array (Montgomery) or a device pointer with n.  Returns (xy uint64 array, is_infinity)."""
        p, where = _ptr(coeffs)
        if n is None:
            n = coeffs.shape[0]
        z = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros(2 * FQ_BYTES[self.curve] // 8, dtype=np.uint64)
        inf = C.c_int(0)
        self.ctx.check(self.ctx.lib.pc_hip_kzg_open(self.ctx.h, self.h, base_offset, p, where, n, C.c_void_p(z.ctypes.data),
                                                    C.c_void_p(out.ctypes.data), C.byref(inf)))
        return out, bool(inf.value)

    def msm_many(self, scalars, m=None, n_msms=None, base_offset=0, montgomery=False):
        """n_msms MSMs of m pairs over bases[base_offset : base_offset + m] (pc_hip_msm_many; Hyrax's
        one commitment per matrix row).  scalars: (n_msms, m, 4) uint64 host array or a device pointer.
        Returns ((n_msms, 2*Fq limbs) uint64 affine points, (n_msms,) infinity flags)."""
        p, where = _ptr(scalars)
        if m is None:
            n_msms, m = scalars.shape[0], scalars.shape[1]
        out = np.zeros((n_msms, 2 * FQ_BYTES[self.curve] // 8), dtype=np.uint64)
        inf = (C.c_int * max(n_msms, 1))()
        self.ctx.check(self.ctx.lib.pc_hip_msm_many(self.ctx.h, self.h, base_offset, p,
                                                    PC_SCALARS_MONTGOMERY if montgomery else PC_SCALARS_CANONICAL, where, m, n_msms,
                                                    C.c_void_p(out.ctypes.data), inf))
        return out, np.array(list(inf)[:n_msms], dtype=bool)

    def msm_batch(self, scalar_ptrs, lens, base_offsets=None, montgomery=True, host=False):
        """pc_hip_msm_batch: k scalar vectors against this SRS -> (k, 2*Fq) points.  scalar_ptrs: device pointers, or (host=True) host
        numpy arrays / addresses (what MarlinKZG10::commit hands over: the library stages them pass by pass)."""
        k = len(scalar_ptrs)
        if host:
            scalar_ptrs = [p.ctypes.data if isinstance(p, np.ndarray) else p for p in scalar_ptrs]
        ptrs = (C.c_void_p * k)(*scalar_ptrs)
        ns = (C.c_size_t * k)(*lens)
        offs = (C.c_size_t * k)(*base_offsets) if base_offsets is not None else None
        out = np.zeros((k, 2 * FQ_BYTES[self.curve] // 8), dtype=np.uint64)
        infs = (C.c_int * k)()
        self.ctx.check(self.ctx.lib.pc_hip_msm_batch(self.ctx.h, self.h, offs, ptrs, ns, k,
                                                     PC_SCALARS_MONTGOMERY if montgomery else PC_SCALARS_CANONICAL,
                                                     PC_MEM_HOST if host else PC_MEM_DEVICE, C.c_void_p(out.ctypes.data), infs))
        return out

    def ec_fold(self, n_half, u):
        """key[i] = affine(key[i] + u * key[n_half + i]) in place on the resident key."""
        u = np.ascontiguousarray(u, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.pc_hip_ec_fold(self.ctx.h, self.h, n_half, C.c_void_p(u.ctypes.data)))

    def precompute_fold(self, levels=None, naf_width=None):
        """Fold table of this (committer) key: pc_hip_srs_precompute_fold (the library's choice of form), or
        pc_hip_srs_precompute_fold_ex(levels, naf_width) -- levels 1: the upper half (first fold of an opening), 2: the upper three
        quarters (the first two folds in one step, fold2_from); naf_width 2 .. 5; 0 = the library chooses that parameter."""
        if levels is None and naf_width is None:
            self.ctx.check(self.ctx.lib.pc_hip_srs_precompute_fold(self.ctx.h, self.h))
        else:
            self.ctx.check(self.ctx.lib.pc_hip_srs_precompute_fold_ex(self.ctx.h, self.h, levels or 0, naf_width or 0))
        return self

    def fold_table_info(self):
        """(levels, naf_width) of the fold table on this key; (0, 0): none."""
        a, b = C.c_uint(), C.c_uint()
        self.ctx.check(self.ctx.lib.pc_hip_srs_fold_table_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def ipa_open_rounds(self, coeffs_dev, n, point_mont, h_prime_xy, next_challenge, fixed_key_below=0, want_times=False):
        """The whole halving loop of InnerProductArgPC::open on this resident committer key (pc_hip_ipa_open_rounds).
        next_challenge(l_xy, r_xy) -> u as 4 Montgomery uint64 limbs.  Returns (l_vec, r_vec, final_key, c[, round_ms, fold_ms])."""
        lg = n.bit_length() - 1
        w = 2 * FQ_BYTES[self.curve] // 8
        l = np.zeros((max(lg, 1), w), dtype=np.uint64)
        r = np.zeros((max(lg, 1), w), dtype=np.uint64)
        fk, c = np.zeros(w, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        err = []

        def cb(_user, lp, rp, up):
            try:
                lxy = np.frombuffer((C.c_uint64 * w).from_address(lp), dtype=np.uint64).copy()
                rxy = np.frombuffer((C.c_uint64 * w).from_address(rp), dtype=np.uint64).copy()
                u = np.ascontiguousarray(next_challenge(lxy, rxy), dtype=np.uint64)
                C.memmove(up, u.ctypes.data, 32)
            except BaseException as e:      # an exception must not cross the C frames: finish the loop on a dummy challenge, raise afterwards
                err.append(e)
                C.memset(up, 0, 32); C.memset(up, 1, 1)
        fn = IPA_CHALLENGE_FN(cb)
        point = np.ascontiguousarray(point_mont, dtype=np.uint64)
        hp = np.ascontiguousarray(h_prime_xy, dtype=np.uint64)
        rms = (C.c_float * max(lg, 1))()
        fms = (C.c_float * max(lg, 1))()
        p, _ = _ptr(coeffs_dev)
        rc = self.ctx.lib.pc_hip_ipa_open_rounds(self.ctx.h, self.h, p, n, C.c_void_p(point.ctypes.data), C.c_void_p(hp.ctypes.data), fn, None,
                                                 fixed_key_below, C.c_void_p(l.ctypes.data), C.c_void_p(r.ctypes.data), C.c_void_p(fk.ctypes.data),
                                                 C.c_void_p(c.ctypes.data), rms, fms)
        if err:
            raise err[0]
        self.ctx.check(rc)
        out = (l[:lg], r[:lg], fk, c)
        return out + (list(rms)[:lg], list(fms)[:lg]) if want_times else out

    def ipa_round2_msms(self, coeffs_dev, n_quarter, u1):
        """Round 2's two commitments of an opening on this (committer) key by linearity (pc_hip_ipa_round2_msms): (l, r) affine."""
        u1 = np.ascontiguousarray(u1, dtype=np.uint64)
        w = 2 * FQ_BYTES[self.curve] // 8
        l, r = np.zeros(w, dtype=np.uint64), np.zeros(w, dtype=np.uint64)
        p, _ = _ptr(coeffs_dev)
        self.ctx.check(self.ctx.lib.pc_hip_ipa_round2_msms(self.ctx.h, self.h, p, n_quarter, C.c_void_p(u1.ctypes.data), C.c_void_p(l.ctypes.data), None,
                                                           C.c_void_p(r.ctypes.data), None))
        return l, r

    def fold2_from(self, n_quarter, u1, u2):
        """A new resident key of n_quarter points: the key after the first TWO folds (by u1, then u2), straight from this key
        (pc_hip_ec_fold2_from); self is left as it is."""
        u1 = np.ascontiguousarray(u1, dtype=np.uint64)
        u2 = np.ascontiguousarray(u2, dtype=np.uint64)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.pc_hip_ec_fold2_from(self.ctx.h, self.h, n_quarter, C.c_void_p(u1.ctypes.data), C.c_void_p(u2.ctypes.data), C.byref(h)))
        out = Srs.__new__(Srs)
        out.ctx, out.curve, out.h, out.n = self.ctx, self.curve, h, n_quarter
        return out

    def fold_from(self, n_half, u):
        """A new resident key: affine(self[i] + u * self[n_half + i]), i < n_half; self is left as it is (pc_hip_ec_fold_from)."""
        u = np.ascontiguousarray(u, dtype=np.uint64)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.pc_hip_ec_fold_from(self.ctx.h, self.h, n_half, C.c_void_p(u.ctypes.data), C.byref(h)))
        out = Srs.__new__(Srs)
        out.ctx, out.curve, out.h, out.n = self.ctx, self.curve, h, n_half
        return out

    def device_ptr(self):
        """Address of the resident packed point array (pc_hip_srs_device_ptr)."""
        return int(self.ctx.lib.pc_hip_srs_device_ptr(self.h))

    def clone(self):
        """A second resident copy (device-to-device): what a destructive consumer -- the IPA key fold -- works on."""
        return Srs(self.ctx, self.curve, self.device_ptr(), n=self.n)

    def bytes_resident(self):
        out = (C.c_size_t * 4)()
        self.ctx.check(self.ctx.lib.pc_hip_srs_bytes_resident(self.h, out))
        return dict(zip(("bases", "window_tables", "fold_table", "pipelines"), [int(x) for x in out]))

    def serialize(self, offset=0, count=None, compressed=False):
        """ark-serialize bytes of resident points as a Vec<G1Affine> (pc_hip_srs_serialize)."""
        count = self.n - offset if count is None else count
        need = C.c_size_t()
        self.ctx.check(self.ctx.lib.pc_hip_srs_serialize(self.ctx.h, self.h, offset, count, 1 if compressed else 0, None, 0, C.byref(need)))
        buf = (C.c_char * need.value)()
        self.ctx.check(self.ctx.lib.pc_hip_srs_serialize(self.ctx.h, self.h, offset, count, 1 if compressed else 0, buf, need.value, C.byref(need)))
        return bytes(buf)

    def read(self, offset, count):
        out = np.zeros((count, 2 * FQ_BYTES[self.curve] // 8), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.pc_hip_srs_read(self.ctx.h, self.h, offset, count, C.c_void_p(out.ctypes.data)))
        return out

    def msm_async(self, scalars, n=None, base_offset=0, montgomery=False):
        """Queue an MSM; returns a job whose .wait() gives (xy, is_infinity)."""
        p, where = _ptr(scalars)
        if n is None:
            n = scalars.shape[0]
        return MsmJob(self, p, where, n, base_offset, montgomery)


class MsmJob:
    def __init__(self, srs, p, where, n, base_offset, montgomery):
        self.srs = srs
        self.out = np.zeros(2 * FQ_BYTES[srs.curve] // 8, dtype=np.uint64)
        self.inf = C.c_int(0)
        self.h = C.c_void_p()
        ctx = srs.ctx
        ctx.check(ctx.lib.pc_hip_msm_async(ctx.h, srs.h, base_offset, p,
                                           PC_SCALARS_MONTGOMERY if montgomery else PC_SCALARS_CANONICAL, where, n,
                                           C.c_void_p(self.out.ctypes.data), C.byref(self.inf), C.byref(self.h)))
        self.phases = None
        self.marks = None

    def wait(self):
        ctx = self.srs.ctx
        if self.h:
            ctx.check(ctx.lib.pc_hip_job_wait(ctx.h, self.h))
            self.h = None
            self.phases = ctx.last_msm_phases_ms()
            self.marks = ctx.last_msm_marks_ms()
        return self.out, bool(self.inf.value)


def universal_params_layout(curve, data, compressed):
    """Field offsets of a serialized kzg10::UniversalParams (pc_hip_universal_params_layout)."""
    lib = load_library()
    out = (C.c_size_t * 9)()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    rc = lib.pc_hip_universal_params_layout(CURVES[curve], buf, len(data), 1 if compressed else 0, out)
    if rc != 0:
        raise PcHipError(rc, lib.pc_hip_strerror(rc).decode())
    keys = ("powers_of_g", "n_powers_of_g", "powers_of_gamma_g", "n_powers_of_gamma_g", "h", "beta_h", "neg_powers_of_h", "n_neg_powers_of_h", "total")
    return dict(zip(keys, list(out)))


def points_sum(curve, points):
    """Host-side sum of affine points (k x 2*Fq uint64 array) -> one affine point."""
    lib = load_library()
    points = np.ascontiguousarray(points, dtype=np.uint64)
    out = np.zeros(2 * FQ_BYTES[curve] // 8, dtype=np.uint64)
    rc = lib.pc_hip_points_sum(CURVES[curve], C.c_void_p(points.ctypes.data), points.shape[0], C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise PcHipError(rc, lib.pc_hip_strerror(rc).decode())
    return out


def point_mul(curve, point_xy, scalar_mont):
    """Host-side k * P for one affine point and one Montgomery-form Fr."""
    lib = load_library()
    point_xy = np.ascontiguousarray(point_xy, dtype=np.uint64)
    scalar_mont = np.ascontiguousarray(scalar_mont, dtype=np.uint64)
    out = np.zeros(2 * FQ_BYTES[curve] // 8, dtype=np.uint64)
    rc = lib.pc_hip_point_mul(CURVES[curve], C.c_void_p(point_xy.ctypes.data), C.c_void_p(scalar_mont.ctypes.data),
                              C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise PcHipError(rc, lib.pc_hip_strerror(rc).decode())
    return out


class Group:
    """N single-device contexts driven from one process (pc_hip_group_*): one committer key sharded in contiguous
    chunks.  device_ids may repeat a device (tests on a one-GPU box)."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.lib.pc_hip_group_create(ids, len(device_ids), C.byref(h))
        if rc != 0:
            raise PcHipError(rc, self.lib.pc_hip_strerror(rc).decode())
        self.h, self.n_dev = h, len(device_ids)

    def check(self, rc):
        if rc != 0:
            raise PcHipError(rc, self.lib.pc_hip_strerror(rc).decode())

    def close(self):
        if self.h:
            self.lib.pc_hip_group_destroy(self.h)
            self.h = None

    def upload_srs(self, curve, bases, precompute=False):
        return GroupSrs(self, curve, bases, precompute)

    def ntt_batch(self, curve, mat, log_n):
        mat = np.ascontiguousarray(mat, dtype=np.uint64)
        out = np.zeros((mat.shape[0], 1 << log_n, 4), dtype=np.uint64)
        self.check(self.lib.pc_hip_group_ntt_batch(self.h, CURVES[curve], mat.ctypes.data, mat.shape[0], mat.shape[1], log_n, out.ctypes.data))
        return out

    def ligero_commit(self, curve, mat, log_n, col_hash="blake2s", tree_hash="sha256", len_prefix=True):
        """LinearCodePCS::commit steps 1-3 with the rows split over the group's devices (pc_hip_group_ligero_commit): chained
        column digests, tree on the last device.  Returns (nodes (2^h - 1, 32) uint8 with the root at row 0, leaves (2^log_n, 32))."""
        mat = np.ascontiguousarray(mat, dtype=np.uint64)
        hid = {"sha256": 0, "blake2s": 1}
        nodes = np.zeros(((1 << max(1, log_n)) - 1, 32), dtype=np.uint8)
        leaves = np.zeros((1 << log_n, 32), dtype=np.uint8)
        self.check(self.lib.pc_hip_group_ligero_commit(self.h, CURVES[curve], mat.ctypes.data, mat.shape[0], mat.shape[1], log_n, hid[col_hash],
                                                       hid[tree_hash], 1 if len_prefix else 0, None, leaves.ctypes.data, nodes.ctypes.data))
        return nodes, leaves


class GroupJob:
    def __init__(self, gsrs, coeffs, z_mont, n, want_value=True):
        self.g = gsrs.g
        nq = 2 * FQ_BYTES[gsrs.curve] // 8
        self.comm, self.proof, self.val = np.zeros(nq, dtype=np.uint64), np.zeros(nq, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        self.z = np.ascontiguousarray(z_mont, dtype=np.uint64)
        if isinstance(coeffs, np.ndarray):
            self.keep = np.ascontiguousarray(coeffs, dtype=np.uint64)
            ptr, where, n = C.c_void_p(self.keep.ctypes.data), PC_MEM_HOST, self.keep.shape[0]
        else:
            self.keep = (C.c_void_p * len(coeffs))(*coeffs)
            ptr, where = C.cast(self.keep, C.c_void_p), PC_MEM_DEVICE
        self.h = C.c_void_p()
        self.g.check(self.g.lib.pc_hip_group_commit_open_async(self.g.h, gsrs.h, ptr, where, n, self.z.ctypes.data, self.comm.ctypes.data,
                                                              self.proof.ctypes.data, self.val.ctypes.data if want_value else None, C.byref(self.h)))

    def wait(self):
        if self.h:
            self.g.check(self.g.lib.pc_hip_group_job_wait(self.g.h, self.h))
            self.h = None
        return self.comm, self.proof, self.val


class GroupSrs:
    def __init__(self, group, curve, bases, precompute=False):
        self.g, self.curve = group, curve
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        h = C.c_void_p()
        group.check(group.lib.pc_hip_group_srs_upload(group.h, CURVES[curve], bases.ctypes.data, bases.shape[0], 0, 1 if precompute else 0, C.byref(h)))
        self.h, self.n = h, bases.shape[0]

    def free(self):
        if self.h:
            self.g.lib.pc_hip_group_srs_free(self.h)
            self.h = None

    def msm(self, scalars, base_offset=0, montgomery=False):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(2 * FQ_BYTES[self.curve] // 8, dtype=np.uint64)
        inf = C.c_int(0)
        self.g.check(self.g.lib.pc_hip_group_msm(self.g.h, self.h, base_offset, scalars.ctypes.data,
                                                 PC_SCALARS_MONTGOMERY if montgomery else PC_SCALARS_CANONICAL, scalars.shape[0],
                                                 out.ctypes.data, C.byref(inf)))
        return out, bool(inf.value)

    def msm_batch(self, polys, montgomery=True):
        polys = [np.ascontiguousarray(p, dtype=np.uint64) for p in polys]
        k = len(polys)
        ptrs = (C.c_void_p * k)(*[p.ctypes.data for p in polys])
        ns = (C.c_size_t * k)(*[p.shape[0] for p in polys])
        out = np.zeros((k, 2 * FQ_BYTES[self.curve] // 8), dtype=np.uint64)
        self.g.check(self.g.lib.pc_hip_group_msm_batch(self.g.h, self.h, ptrs, ns, k, PC_SCALARS_MONTGOMERY if montgomery else PC_SCALARS_CANONICAL,
                                                       out.ctypes.data, None))
        return out

    def commit_open_async(self, coeffs, z_mont, n=None, want_value=True):
        """pc_hip_group_commit_open_async: coeffs = host (n, 4) uint64 array (Montgomery), or a list of N device pointers with `n`.
        Returns a job; job.wait() -> (commitment xy, proof xy, p(z))."""
        return GroupJob(self, coeffs, z_mont, n, want_value)

    def kzg_open(self, coeffs_mont, z_mont):
        coeffs_mont = np.ascontiguousarray(coeffs_mont, dtype=np.uint64)
        z_mont = np.ascontiguousarray(z_mont, dtype=np.uint64)
        out = np.zeros(2 * FQ_BYTES[self.curve] // 8, dtype=np.uint64)
        val = np.zeros(4, dtype=np.uint64)
        inf = C.c_int(0)
        self.g.check(self.g.lib.pc_hip_group_kzg_open(self.g.h, self.h, coeffs_mont.ctypes.data, coeffs_mont.shape[0], z_mont.ctypes.data,
                                                      out.ctypes.data, C.byref(inf), val.ctypes.data))
        return out, val
