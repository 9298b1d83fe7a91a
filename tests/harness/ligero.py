"""LinearCodePCS (univariate Ligero) commit / open / check above the C ABI (poly-commit/src/linear_codes/mod.rs), one GPU.

  commit   pc_hip_ligero_commit: the coefficient matrix is encoded row by row (batched NTT), every column of the encoded
           matrix is hashed and the Merkle tree is built on the device; matrix, encoded matrix, leaves and nodes stay in
           the commitment state (mod.rs:228-298)
  open     v = b.M (and r.M with check_well_formedness) as device linear combinations of the resident rows (:343-349,
           generate_proof :523-565), the queried columns gathered out of the resident encoded matrix, their
           authentication paths read out of the node array
  check    the received columns are hashed on the device (pc_hip_column_hash), v is encoded with one NTT, the paths and
           the inner products are checked on the host (:375-503)

The sponge is the caller's: the query indices (get_indices_from_sponge, utils.rs:136-153) and, with
check_well_formedness, the vector r are arguments.  Field elements are Montgomery limbs ((4,) uint64); hashes are the
digests the bench templates use (FieldToBytesColHasher over Blake2s / SHA-256, identity leaf hash, SHA-256 two-to-one).
"""
import hashlib
import math

import numpy as np

from poly_commit_amd.sharded import FR_MODULUS, _R, _limbs_to_int


class InvalidCommitment(ValueError):
    """linear_codes Error::InvalidCommitment: a path, an index or an inner product of the proof does not hold."""


def _ints(curve, arr):
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    raw = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, 4).tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") * rinv % p for i in range(len(raw) // 32)]


def _monts(curve, vals):
    p = FR_MODULUS[curve]
    return np.frombuffer(b"".join((v % p * _R % p).to_bytes(32, "little") for v in vals), dtype="<u8").astype(np.uint64).reshape(-1, 4)


def calculate_t(field_bits, sec_param, distance, codeword_len):
    """linear_codes/utils.rs:156-184."""
    arg = 2.0 ** (-sec_param) - codeword_len / 2.0 ** field_bits
    if not arg > 0:
        raise ValueError("InvalidParameters: the field is not big enough for this codeword length and security parameter")
    nom = math.log2(arg) - 1.0
    denom = math.log2(1.0 - 0.5 * distance[0] / distance[1])
    t = math.ceil(nom / denom)
    return t if t < codeword_len else codeword_len


def compute_dimensions(curve, poly_len, rho_inv=4, sec_param=128):
    """LigeroPCParams::compute_dimensions (ligero.rs:118-128) -> (n_rows, n_cols)."""
    t = calculate_t(FR_MODULUS[curve].bit_length(), sec_param, (rho_inv - 1, rho_inv), poly_len)
    n_rows = 1 << max(0, (math.ceil(math.sqrt(-(-2 * poly_len // t))) - 1).bit_length())
    return n_rows, -(-poly_len // n_rows)


def tensor(curve, z_mont, left, right):
    """UnivariateLigero::tensor (univariate_ligero/mod.rs:70-86) on canonical ints."""
    p = FR_MODULUS[curve]
    z = _limbs_to_int(z_mont) * pow(_R, -1, p) % p
    a, pw = [], 1
    for _ in range(left):
        a.append(pw)
        pw = pw * z % p
    b, q = [], 1
    for _ in range(right):
        b.append(q)
        q = q * pw % p
    return a, b


def tensor_vec(curve, values):
    """linear_codes/utils.rs:240-258 on canonical ints: value i in bit i of the index."""
    p = FR_MODULUS[curve]
    layer = [1]
    for v in values:
        layer = [x * (1 - v) % p for x in layer] + [x * v % p for x in layer]
    return layer


def multilinear_tensor(curve, point_mont, left_len):
    """MultilinearLigero::tensor (multilinear_ligero/mod.rs:70-84): (a, b) for open / check's `tensors` argument; the
    multilinear scheme commits to the evaluations with rho_inv = 2 and always checks well-formedness (:49-55)."""
    pt = _ints(curve, np.ascontiguousarray(point_mont, dtype=np.uint64))
    split = max(0, (left_len - 1).bit_length())
    return tensor_vec(curve, pt[:split]), tensor_vec(curve, pt[split:])


def commit(ctx, curve, coeffs_dev, rho_inv=4, sec_param=128, col_hash="blake2s", tree_hash="sha256"):
    """LinearCodePCS::commit for one polynomial.  coeffs_dev: torch cuda int64 (len, 4), Montgomery.
    Returns (commitment dict {n_rows, n_cols, n_ext_cols, root}, state)."""
    import torch
    total = coeffs_dev.shape[0]
    n_rows, n_cols = compute_dimensions(curve, total, rho_inv, sec_param)
    mat = torch.zeros((n_rows * n_cols, 4), dtype=torch.int64, device=coeffs_dev.device)      # row-major fill, zero padded (utils.rs:72-74)
    mat[:total] = coeffs_dev
    log_n = (n_cols * rho_inv - 1).bit_length()
    ext = torch.empty((n_rows << log_n, 4), dtype=torch.int64, device=coeffs_dev.device)
    nodes, leaves = ctx.ligero_commit(curve, mat.data_ptr(), log_n, col_hash=col_hash, tree_hash=tree_hash, rows=n_rows, in_cols=n_cols,
                                      ext_out=ext.data_ptr())
    com = dict(n_rows=n_rows, n_cols=n_cols, n_ext_cols=1 << log_n, root=bytes(nodes[0]))
    state = dict(mat=mat, ext=ext, leaves=leaves, nodes=nodes, log_n=log_n, col_hash=col_hash, tree_hash=tree_hash, rho_inv=rho_inv, **com)
    return com, state


def _merkle_path(nodes, leaves, index):
    """MerkleTree::generate_proof out of the heap-ordered inner nodes: (sibling leaf digest, siblings bottom-up)."""
    n_inner = nodes.shape[0]
    sib = index ^ 1
    leaf_sibling = bytes(leaves[sib]) if sib < leaves.shape[0] else b""
    node = (n_inner + index + 1) // 2 - 1
    path = []
    while node > 0:
        path.append(bytes(nodes[node + 1 if node % 2 == 1 else node - 1]))
        node = (node - 1) // 2
    return leaf_sibling, path


def _merkle_verify(root, leaf, index, leaf_sibling, path, hash_name):
    conv = lambda b: len(b).to_bytes(8, "little") + b            # noqa: E731  (ByteDigestConverter: ark-serialize of the Vec<u8>)
    l, r = (leaf, leaf_sibling) if index % 2 == 0 else (leaf_sibling, leaf)
    cur = hashlib.new(hash_name, conv(l) + conv(r)).digest()
    index //= 2
    for s in path:
        cur = hashlib.new(hash_name, (cur + s) if index % 2 == 0 else (s + cur)).digest()
        index //= 2
    return cur == root


def open(ctx, curve, state, z_mont, indices, r_mont=None, tensors=None):   # noqa: A001 (the reference's name)
    """LinearCodePCS::open for one polynomial: LinCodePCProof {opening: {paths, v, columns}, well_formedness}.
    tensors = (a, b) replaces the univariate L::tensor(z, ..) (multilinear_tensor for MultilinearLigero)."""
    import torch
    n_rows, n_cols, n_ext = state["n_rows"], state["n_cols"], state["n_ext_cols"]
    mat, ext = state["mat"], state["ext"]
    _, b = tensors if tensors is not None else tensor(curve, z_mont, n_cols, n_rows)
    rows = [mat.data_ptr() + 32 * n_cols * i for i in range(n_rows)]

    def row_mul(coeffs_mont):                                      # Matrix::row_mul (poly-commit/src/utils.rs:120-147)
        out = torch.empty((n_cols, 4), dtype=torch.int64, device=mat.device)
        ctx.fr_lincomb(curve, rows, np.ascontiguousarray(coeffs_mont, dtype=np.uint64), n_out=n_cols, out=out.data_ptr(), lens=[n_cols] * n_rows)
        return out.cpu().numpy().view(np.uint64)
    wf = row_mul(r_mont) if r_mont is not None else None           # :343-349
    v = row_mul(_monts(curve, b))                                  # generate_proof step 1
    cols = ctx.matrix_columns(ext.data_ptr(), n_rows, n_ext, list(indices))                           # (t, n_rows, 4)
    paths = [(int(i),) + _merkle_path(state["nodes"], state["leaves"], int(i)) for i in indices]
    return dict(v=v, columns=cols, paths=paths, well_formedness=wf)


def check(ctx, curve, commitment, z_mont, value_mont, proof, indices, r_mont=None, rho_inv=4, col_hash="blake2s", tree_hash="sha256",
          tensors=None):
    """LinearCodePCS::check for one commitment.  Raises InvalidCommitment where the reference returns Err(InvalidCommitment),
    returns False when only the claimed value is wrong (:494-499)."""
    import torch
    p = FR_MODULUS[curve]
    n_rows, n_cols, n_ext = commitment["n_rows"], commitment["n_cols"], commitment["n_ext_cols"]
    if (r_mont is not None) != (proof["well_formedness"] is not None):
        raise InvalidCommitment("well-formedness proof missing or unexpected")
    # Shape of the proof against the t query indices the sponge produced: the reference indexes
    # proof.opening.columns[transcript_index] and .paths[..] for EVERY one of them (linear_codes/mod.rs:443-489), so a proof
    # with fewer columns or paths can never be accepted there (round-2 advisor finding: zip() stopped at the shorter list).
    indices = [int(i) for i in indices]
    t = len(indices)
    try:
        cols = np.ascontiguousarray(proof["columns"], dtype=np.uint64)                 # (t, n_rows, 4)
        v_arr = np.ascontiguousarray(proof["v"], dtype=np.uint64)
        wf_arr = None if proof["well_formedness"] is None else np.ascontiguousarray(proof["well_formedness"], dtype=np.uint64)
    except (TypeError, ValueError):
        raise InvalidCommitment("proof shape")
    height = max(1, (n_ext - 1).bit_length())
    if (cols.shape != (t, n_rows, 4) or len(proof["paths"]) != t or v_arr.shape != (n_cols, 4)
            or (wf_arr is not None and wf_arr.shape != (n_cols, 4)) or n_ext & (n_ext - 1) or n_ext < n_cols
            or any(not 0 <= i < n_ext for i in indices)
            or any(len(pth) != 3 or len(pth[1]) not in (0, 32) or len(pth[2]) != height - 1 or any(len(s) != 32 for s in pth[2]) for pth in proof["paths"])):
        raise InvalidCommitment("proof shape")
    # 3. hash the received columns on the device: they are the columns of an n_rows x t matrix
    digests = ctx.column_hash(curve, np.ascontiguousarray(cols.transpose(1, 0, 2)), col_hash)       # (t, 32)
    # 4. the paths
    for j, (q_j, (idx, sib, path)) in enumerate(zip(indices, proof["paths"])):
        if idx != q_j or not _merkle_verify(commitment["root"], bytes(digests[j]), idx, sib, path, tree_hash):
            raise InvalidCommitment(f"path of column {q_j}")
    # 5. w = E(v) (and E(well_formedness)): one NTT each
    log_n = n_ext.bit_length() - 1

    def encode(vec_mont):
        x = torch.from_numpy(np.ascontiguousarray(vec_mont, dtype=np.uint64).reshape(1, n_cols, 4).view(np.int64)).cuda()
        y = torch.empty((n_ext, 4), dtype=torch.int64, device="cuda")
        ctx.ntt_batch(curve, x.data_ptr(), log_n, out=y.data_ptr(), rows=1, in_cols=n_cols)
        return y
    sel = torch.tensor(list(indices), dtype=torch.long, device="cuda")
    w = _ints(curve, encode(v_arr)[sel].cpu().numpy().view(np.uint64))
    a, b = tensors if tensors is not None else tensor(curve, z_mont, n_cols, n_rows)
    col_ints = [_ints(curve, cols[j]) for j in range(t)]
    if r_mont is not None:
        wwf = _ints(curve, encode(wf_arr)[sel].cpu().numpy().view(np.uint64))
        r = _ints(curve, r_mont)
        for j in range(t):
            if sum(x * y for x, y in zip(r, col_ints[j])) % p != wwf[j]:
                raise InvalidCommitment(f"well-formedness at column {indices[j]}")
    for j in range(t):
        if sum(x * y for x, y in zip(b, col_ints[j])) % p != w[j]:
            raise InvalidCommitment(f"b.column != w at column {indices[j]}")
    value = _limbs_to_int(value_mont) * pow(_R, -1, p) % p
    return sum(x * y for x, y in zip(_ints(curve, v_arr), a)) % p == value
