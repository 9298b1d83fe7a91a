"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref  # noqa: E402

CURVES = {"bls12_381": 0, "bn254": 1, "pallas": 2}
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        src_m = max(os.path.getmtime(os.path.join(ROOT, "oracle", f))
                    for f in ("oracle.cpp", "oracle_field.hpp", "oracle_constants.h", "fast_msm.hpp"))
        # several ranks of one node may get here at once (bench.py --gpus N): build under a file lock
        import fcntl
        os.makedirs(os.path.dirname(so), exist_ok=True)
        with open(so + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if not os.path.exists(so) or os.path.getmtime(so) < src_m:
                build()
        _lib = C.CDLL(so)
        _lib.orc_fq_limbs.restype = C.c_int
        _lib.orc_kzg_commit.restype = C.c_int
        _lib.orc_kzg_open.restype = C.c_int
        _lib.orc_on_curve.restype = C.c_int
        _lib.orc_ligero_dims.restype = C.c_int
    return _lib


def p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def fq_limbs(curve):
    return 6 if curve == "bls12_381" else 4


# ---- int <-> limb arrays -----------------------------------------------------------------
def ints_to_limbs(vals, n64):
    out = np.zeros((len(vals), n64), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(n64):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, arr.shape[-1])
    return [sum(int(arr[i, k]) << (64 * k) for k in range(arr.shape[1])) for i in range(arr.shape[0])]


def mont_R(field):
    return 1 << (64 * pyref.FIELDS[field]["limbs64"])


def to_mont_ints(field, vals):
    p = pyref.FIELDS[field]["p"]
    R = mont_R(field)
    return [v * R % p for v in vals]


def from_mont_ints(field, vals):
    p = pyref.FIELDS[field]["p"]
    Ri = pow(mont_R(field), -1, p)
    return [v * Ri % p for v in vals]


def fr_mont_array(curve, vals):
    f = pyref.CURVES[curve]["fr"]
    return ints_to_limbs(to_mont_ints(f, vals), 4)


def fr_from_mont_array(curve, arr):
    f = pyref.CURVES[curve]["fr"]
    return from_mont_ints(f, limbs_to_ints(arr.reshape(-1, 4)))


def points_to_array(curve, pts):
    """list of affine int points (or None) -> (n, 2*N) uint64 Montgomery array."""
    f = pyref.CURVES[curve]["fq"]
    n64 = fq_limbs(curve)
    flat = []
    for P in pts:
        flat += [0, 0] if P is None else [P[0], P[1]]
    m = to_mont_ints(f, flat)
    # infinity stays (0,0)
    return ints_to_limbs(m, n64).reshape(len(pts), 2 * n64)


def array_to_points(curve, arr):
    f = pyref.CURVES[curve]["fq"]
    n64 = fq_limbs(curve)
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 2 * n64)
    vals = from_mont_ints(f, limbs_to_ints(arr.reshape(-1, n64)))
    out = []
    for i in range(arr.shape[0]):
        x, y = vals[2 * i], vals[2 * i + 1]
        out.append(None if (x == 0 and y == 0) else (x, y))
    return out


# ---- oracle wrappers ---------------------------------------------------------------------
def gen_bases(curve, n):
    out = np.zeros((n, 2 * fq_limbs(curve)), dtype=np.uint64)
    lib().orc_gen_bases(CURVES[curve], C.c_size_t(n), p64(out))
    return out


def gen_scalars(curve, seed, n):
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_gen_scalars(CURVES[curve], C.c_uint64(seed), C.c_size_t(n), p64(out))
    return out


def f_to_mont(curve, which, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    out = np.zeros_like(arr)
    n = arr.size // (fq_limbs(curve) if which == 0 else 4)
    lib().orc_f_from_canonical(CURVES[curve], which, p64(arr), p64(out), C.c_size_t(n))
    return out


def f_from_mont(curve, which, arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    out = np.zeros_like(arr)
    n = arr.size // (fq_limbs(curve) if which == 0 else 4)
    lib().orc_f_to_canonical(CURVES[curve], which, p64(arr), p64(out), C.c_size_t(n))
    return out


def msm_naive(curve, bases, scalars):
    n = min(len(bases), len(scalars))
    out = np.zeros(2 * fq_limbs(curve), dtype=np.uint64)
    lib().orc_msm_naive(CURVES[curve], p64(bases), p64(scalars), C.c_size_t(n), p64(out))
    return out


def msm_pippenger(curve, bases, scalars, threads=8, mode=1):
    n = min(len(bases), len(scalars))
    out = np.zeros(2 * fq_limbs(curve), dtype=np.uint64)
    lib().orc_msm_pippenger(CURVES[curve], p64(bases), p64(scalars), C.c_size_t(n), threads, mode, p64(out))
    return out


def ntt_batch(curve, mat, log_n, threads=8):
    rows, in_cols = mat.shape[0], mat.shape[1]
    mat = np.ascontiguousarray(mat, dtype=np.uint64)
    out = np.zeros((rows, 1 << log_n, 4), dtype=np.uint64)
    lib().orc_ntt_batch(CURVES[curve], p64(mat), C.c_size_t(rows), C.c_size_t(in_cols), C.c_uint(log_n),
                        p64(out), threads)
    return out


def root_of_unity(curve, log_n):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_root_of_unity(CURVES[curve], C.c_uint(log_n), p64(out))
    return out


def poly_eval(curve, coeffs, z):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_poly_eval(CURVES[curve], p64(coeffs), C.c_size_t(len(coeffs)), p64(z), p64(out))
    return out


def witness_poly(curve, coeffs, z):
    n = len(coeffs)
    out = np.zeros((max(n - 1, 1), 4), dtype=np.uint64)
    lib().orc_witness_poly(CURVES[curve], p64(coeffs), C.c_size_t(n), p64(z), p64(out))
    return out[: max(n - 1, 0)]


def kzg_commit(curve, powers, coeffs, threads=8):
    out = np.zeros(2 * fq_limbs(curve), dtype=np.uint64)
    rc = lib().orc_kzg_commit(CURVES[curve], p64(powers), C.c_size_t(len(powers)), p64(coeffs),
                              C.c_size_t(len(coeffs)), threads, p64(out))
    return rc, out


def kzg_open(curve, powers, coeffs, z, threads=8):
    out = np.zeros(2 * fq_limbs(curve), dtype=np.uint64)
    rc = lib().orc_kzg_open(CURVES[curve], p64(powers), C.c_size_t(len(powers)), p64(coeffs),
                            C.c_size_t(len(coeffs)), p64(z), threads, p64(out))
    return rc, out


def ipa_rounds(curve, comm_key, coeffs, z, h_prime, challenges, threads=8):
    n = len(coeffs)
    lg = n.bit_length() - 1
    nq = 2 * fq_limbs(curve)
    l = np.zeros((lg, nq), dtype=np.uint64)
    r = np.zeros((lg, nq), dtype=np.uint64)
    fk = np.zeros(nq, dtype=np.uint64)
    c = np.zeros(4, dtype=np.uint64)
    lib().orc_ipa_rounds(CURVES[curve], p64(comm_key), p64(coeffs), C.c_size_t(n), p64(z), p64(h_prime),
                         p64(challenges), threads, p64(l), p64(r), p64(fk), p64(c))
    return l, r, fk, c


def ipa_rounds_fs(curve, comm_key, coeffs, z, h_prime, round_challenge0, threads=8):
    """The halving rounds with the reference's Fiat-Shamir challenges (ipa_pc/mod.rs:681-688), starting from
    the challenge open() derived before the loop.  Returns (l, r, final_key, c, challenges)."""
    n = len(coeffs)
    lg = n.bit_length() - 1
    nq = 2 * fq_limbs(curve)
    l = np.zeros((lg, nq), dtype=np.uint64)
    r = np.zeros((lg, nq), dtype=np.uint64)
    fk = np.zeros(nq, dtype=np.uint64)
    c = np.zeros(4, dtype=np.uint64)
    ch = np.zeros((max(lg, 1), 4), dtype=np.uint64)
    lib().orc_ipa_rounds_fs(CURVES[curve], p64(comm_key), p64(coeffs), C.c_size_t(n), p64(z), p64(h_prime),
                            p64(np.ascontiguousarray(round_challenge0)), threads, p64(l), p64(r), p64(fk), p64(c), p64(ch))
    return l, r, fk, c, ch[:lg]


def ipa_first_challenge(curve, commitment_xy, point, value):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_ipa_first_challenge(CURVES[curve], p64(np.ascontiguousarray(commitment_xy)), p64(np.ascontiguousarray(point)),
                                  p64(np.ascontiguousarray(value)), p64(out))
    return out


def ser_point(curve, xy):
    buf = (C.c_uint8 * 128)()
    lib().orc_ser_point.restype = C.c_size_t
    n = lib().orc_ser_point(CURVES[curve], p64(np.ascontiguousarray(xy)), buf)
    return bytes(buf[:n])


def blake2s(data):
    out = (C.c_uint8 * 32)()
    lib().orc_blake2s(bytes(data), C.c_size_t(len(data)), out)
    return bytes(out)


def ligero_dims(field_bits, poly_len, rho_inv=4, sec_param=128):
    a, b, t = C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = lib().orc_ligero_dims(field_bits, C.c_size_t(poly_len), C.c_size_t(rho_inv), sec_param,
                               C.byref(a), C.byref(b), C.byref(t))
    assert rc == 0
    return a.value, b.value, t.value


def ligero_encode(curve, coeffs, n_rows, n_cols, rho_inv=4, threads=8):
    size = 1
    while size < n_cols * rho_inv:
        size <<= 1
    ext = np.zeros((n_rows, size, 4), dtype=np.uint64)
    lib().orc_ligero_encode(CURVES[curve], p64(coeffs), C.c_size_t(len(coeffs)), C.c_size_t(n_rows),
                            C.c_size_t(n_cols), C.c_size_t(rho_inv), p64(ext), threads)
    return ext
