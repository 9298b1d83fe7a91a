"""Consumer of tests/golden/ref_arkworks.json -- the file `rust/ref-golden` writes by running the REAL arkworks crates (the
reference checkout + ark-ec / ark-ff / ark-poly / ark-serialize 0.5) on the seeded inputs of oracle/pyref.py.

TEST INFRASTRUCTURE.  Three engines recompute every section of the file from its seeds:

    PyrefEngine    oracle/pyref.py, Python big ints (small cases only)
    OracleEngine   oracle/oracle.cpp through tests/oracle_lib.py
    HipEngine      the product: libpc_hip.so through the C ABI (poly_commit_amd/_ffi.py, poly_commit_amd/ipa.py)

`check_section(engine, section, case)` returns the list of mismatching keys.  While nobody with cargo has run the recipe the file
is absent and the consuming tests skip, loudly; `tools/ref_golden_rehearsal.py` writes a stand-in with the same schema from
pyref + the C++ oracle so that the schema and this plumbing are exercised anyway (a rehearsal pins nothing: the stand-in's
`generator` says so and `is_reference_file` refuses it).
"""
import hashlib
import json
import os

import numpy as np

import oracle_lib as O
import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILE = os.path.join(ROOT, "tests", "golden", "ref_arkworks.json")
SCHEMA = 1
SKIP_REASON = ("tests/golden/ref_arkworks.json is absent: PARITY IS NOT PINNED BY THE REFERENCE.  Produce it on a box with cargo: "
               "`cd rust/ref-golden && cargo run --release` (rust/README.md), commit the file, and these tests compare pyref, the C++ "
               "oracle and the HIP library with the real arkworks outputs.")
FIELD_BITS = {"bls12_381": 255, "bn254": 254, "pallas": 255}


def load(path=None):
    path = path or os.environ.get("PC_REF_GOLDEN") or REF_FILE
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    assert doc.get("schema") == SCHEMA, f"{path}: schema {doc.get('schema')} (this consumer reads {SCHEMA})"
    return doc


def is_reference_file(doc):
    return doc is not None and doc.get("generator", "").startswith("rust/ref-golden")


# ---- value forms ------------------------------------------------------------------------------------------------------------------
def h2i(s):
    return int(s, 16)


def pt_of(js):
    return None if js is None else (h2i(js[0]), h2i(js[1]))


def pt_js(P):
    return None if P is None else [hex(P[0]), hex(P[1])]


def fr_js(v):
    return hex(int(v))


def _norm(v):
    """JSON values in one canonical form for comparison: hex strings -> ints, nested lists kept."""
    if isinstance(v, str) and v.startswith("0x"):
        return int(v, 16)
    if isinstance(v, list):
        return [_norm(x) for x in v]
    return v


# ---- seeded inputs (the same streams the Rust side regenerates) -------------------------------------------------------------------------
def scalars_canonical(curve, seed, n):
    """(n, 4) uint64 canonical residues: pyref.gen_scalars == orc_gen_scalars (tests/test_oracle_cpu.py pins the two together)."""
    return O.gen_scalars(curve, seed, n)


def bases(curve, n):
    return O.gen_bases(curve, n)


def msm_inputs(case):
    curve, n, seed = case["curve"], case["n"], case["seed"]
    b = bases(curve, n).copy()
    s = scalars_canonical(curve, seed, n).copy()
    r = R.FIELDS[R.CURVES[curve]["fr"]]["p"]
    s[0] = 0
    s[1] = O.ints_to_limbs([1], 4)[0]
    s[2] = O.ints_to_limbs([r - 1], 4)[0]
    s[n - 1] = 0
    b[3] = b[4]
    return b, s


def kzg_inputs(case):
    curve, d = case["curve"], case["degree"]
    co = scalars_canonical(curve, case["seed"], d + 1).copy()
    co[:case["zero_low"]] = 0
    z = scalars_canonical(curve, case["z_seed"], 1)[0]
    return bases(curve, d + 1), co, z


# ---- engines ------------------------------------------------------------------------------------------------------------------------
class OracleEngine:
    """oracle/oracle.cpp (+ hashlib / pyref for bytes).  Points travel as Montgomery (x||y) uint64 arrays, Fr as canonical arrays."""
    name = "oracle.cpp"
    max_pairs = 1 << 20

    def msm(self, curve, b, s_canon):
        return O.array_to_points(curve, O.msm_pippenger(curve, np.ascontiguousarray(b), np.ascontiguousarray(s_canon), os.cpu_count() or 8, 1))[0]

    def kzg(self, curve, powers, co_canon, z_canon):
        com = O.f_to_mont(curve, 1, co_canon)
        zm = O.f_to_mont(curve, 1, z_canon.reshape(1, 4))[0]
        rc1, c = O.kzg_commit(curve, np.ascontiguousarray(powers), com)
        rc2, w = O.kzg_open(curve, np.ascontiguousarray(powers), com, zm)
        assert rc1 == 0 and rc2 == 0
        v = O.fr_from_mont_array(curve, O.poly_eval(curve, com, zm).reshape(1, 4))[0]
        return O.array_to_points(curve, c)[0], O.array_to_points(curve, w)[0], v

    def lincomb(self, curve, polys_canon, xi_ints):
        fr = R.CURVES[curve]["fr"]
        return R.fr_lincomb(fr, [O.limbs_to_ints(q) for q in polys_canon], xi_ints)

    def ntt_rows(self, curve, mat_canon, log_n):
        out = O.ntt_batch(curve, O.f_to_mont(curve, 1, mat_canon).reshape(mat_canon.shape), log_n)
        return O.f_from_mont(curve, 1, out).reshape(out.shape)

    def ipa_open(self, curve, comm_key, h_xy, polys_canon, comm_pts, xi_ints, z_int):
        fr = R.CURVES[curve]["fr"]
        n = len(comm_key)
        comb_i = R.fr_lincomb(fr, [O.limbs_to_ints(q) for q in polys_canon], xi_ints)
        comb_i += [0] * (n - len(comb_i))
        ccomm = None
        for P, x in zip(comm_pts, xi_ints):
            ccomm = R.ec_add(curve, ccomm, R.ec_mul(curve, x, P))
        comb = O.fr_mont_array(curve, comb_i)
        point = O.fr_mont_array(curve, [z_int])[0]
        v = O.poly_eval(curve, comb, point)
        rc0 = O.ipa_first_challenge(curve, O.points_to_array(curve, [ccomm])[0], point, v)
        hp = R.ec_mul(curve, O.fr_from_mont_array(curve, rc0.reshape(1, 4))[0], O.array_to_points(curve, h_xy.reshape(1, -1))[0])
        l, r, fk, c, _ = O.ipa_rounds_fs(curve, np.ascontiguousarray(comm_key), comb, point, O.points_to_array(curve, [hp])[0], rc0,
                                         threads=os.cpu_count() or 8)
        return (O.array_to_points(curve, l), O.array_to_points(curve, r), O.array_to_points(curve, fk.reshape(1, -1))[0],
                O.fr_from_mont_array(curve, c.reshape(1, 4))[0])

    def ligero_root(self, curve, co_canon, n_rows, n_cols, log_n):
        flat = np.zeros((n_rows * n_cols, 4), dtype=np.uint64)
        flat[:len(co_canon)] = co_canon
        ext = self.ntt_rows(curve, flat.reshape(n_rows, n_cols, 4), log_n)           # canonical
        raw = np.ascontiguousarray(ext).view(np.uint8).reshape(n_rows, 1 << log_n, 32)
        pre = int(n_rows).to_bytes(8, "little")
        leaves = [hashlib.blake2s(pre + raw[:, j, :].tobytes()).digest() for j in range(1 << log_n)]
        return R.merkle_tree(leaves, "sha256", True)[0]


class PyrefEngine:
    """oracle/pyref.py alone: Python big ints end to end (small cases)."""
    name = "pyref.py"
    max_pairs = 600

    def msm(self, curve, b, s_canon):
        return R.msm(curve, O.array_to_points(curve, b), O.limbs_to_ints(s_canon))

    def kzg(self, curve, powers, co_canon, z_canon):
        fr = R.CURVES[curve]["fr"]
        pw, co, z = O.array_to_points(curve, powers), O.limbs_to_ints(co_canon), O.limbs_to_ints(z_canon.reshape(1, 4))[0]
        return R.kzg_commit(curve, pw, co), R.kzg_open(curve, pw, co, z), R.poly_eval(fr, co, z)

    def lincomb(self, curve, polys_canon, xi_ints):
        return R.fr_lincomb(R.CURVES[curve]["fr"], [O.limbs_to_ints(q) for q in polys_canon], xi_ints)

    def ntt_rows(self, curve, mat_canon, log_n):
        fr = R.CURVES[curve]["fr"]
        rows = [R.ntt(fr, O.limbs_to_ints(row) + [0] * ((1 << log_n) - row.shape[0]), log_n) for row in mat_canon]
        return np.stack([O.ints_to_limbs(r, 4) for r in rows])

    def ipa_open(self, curve, comm_key, h_xy, polys_canon, comm_pts, xi_ints, z_int):
        l, r, fk, c, _ = R.ipa_open(curve, O.array_to_points(curve, comm_key), O.array_to_points(curve, h_xy.reshape(1, -1))[0],
                                    [O.limbs_to_ints(q) for q in polys_canon], comm_pts, z_int, xi_ints)
        return l, r, fk, c

    def ligero_root(self, curve, co_canon, n_rows, n_cols, log_n):
        st = R.ligero_commit(R.CURVES[curve]["fr"], O.limbs_to_ints(co_canon))
        assert (st["n_rows"], st["n_cols"], st["n_ext_cols"]) == (n_rows, n_cols, 1 << log_n)
        return st["root"]


class HipEngine:
    """The product, through the C ABI.  Built lazily: importing this module must not need a GPU."""
    name = "libpc_hip.so"
    max_pairs = 1 << 20

    def __init__(self, ctx):
        self.ctx = ctx

    def msm(self, curve, b, s_canon):
        srs = self.ctx.upload_srs(curve, np.ascontiguousarray(b))
        try:
            out, _ = srs.msm(np.ascontiguousarray(s_canon), montgomery=False)
            srs.precompute(min_pairs=1)                                   # and the window-table path
            out2, _ = srs.msm(O.f_to_mont(curve, 1, s_canon), montgomery=True)
            assert (out == out2).all(), "table-free and window-table MSM differ"
        finally:
            srs.free()
        return O.array_to_points(curve, out)[0]

    def kzg(self, curve, powers, co_canon, z_canon):
        com = O.f_to_mont(curve, 1, co_canon)
        zm = O.f_to_mont(curve, 1, z_canon.reshape(1, 4))[0]
        srs = self.ctx.upload_srs(curve, np.ascontiguousarray(powers))
        try:
            c, _ = srs.msm(com, montgomery=True)                      # KZG10::commit: Montgomery coefficients, zeros skipped in the digit kernel
            w, _ = srs.kzg_open(com, zm)                                  # KZG10::open as one call (division + MSM)
            v = self.ctx.poly_eval(curve, com, zm)
        finally:
            srs.free()
        return O.array_to_points(curve, c)[0], O.array_to_points(curve, np.asarray(w).reshape(1, -1))[0], O.fr_from_mont_array(curve, np.asarray(v).reshape(1, 4))[0]

    def lincomb(self, curve, polys_canon, xi_ints):
        polys = [O.f_to_mont(curve, 1, q) for q in polys_canon]
        out = self.ctx.fr_lincomb(curve, polys, O.fr_mont_array(curve, xi_ints))
        return O.fr_from_mont_array(curve, out)

    def ntt_rows(self, curve, mat_canon, log_n):
        out = self.ctx.ntt_batch(curve, O.f_to_mont(curve, 1, mat_canon).reshape(mat_canon.shape), log_n)
        return O.f_from_mont(curve, 1, out).reshape(out.shape)

    def ipa_open(self, curve, comm_key, h_xy, polys_canon, comm_pts, xi_ints, z_int):
        import torch
        from poly_commit_amd import ipa
        dev = [torch.from_numpy(O.f_to_mont(curve, 1, q).view(np.int64).copy()).cuda() for q in polys_canon]
        comms = [O.points_to_array(curve, [P])[0] for P in comm_pts]
        (l, r, fk, c), _ = ipa.ipa_open(self.ctx, curve, np.ascontiguousarray(comm_key), np.ascontiguousarray(h_xy), [d.data_ptr() for d in dev],
                                        [len(q) for q in polys_canon], comms, O.fr_mont_array(curve, [z_int])[0], O.fr_mont_array(curve, xi_ints))
        return (O.array_to_points(curve, l), O.array_to_points(curve, r), O.array_to_points(curve, np.asarray(fk).reshape(1, -1))[0],
                O.fr_from_mont_array(curve, np.asarray(c).reshape(1, 4))[0])

    def ligero_root(self, curve, co_canon, n_rows, n_cols, log_n):
        flat = np.zeros((n_rows * n_cols, 4), dtype=np.uint64)
        flat[:len(co_canon)] = O.f_to_mont(curve, 1, co_canon)
        nodes, _ = self.ctx.ligero_commit(curve, flat.reshape(n_rows, n_cols, 4), log_n, "blake2s", "sha256", True)
        return bytes(nodes[0])


# ---- sections -------------------------------------------------------------------------------------------------------------------------
def compute_constants(case):
    """Engine-independent: what oracle/pyref.py (and through tools/gen_constants.py every header of the library) assumes."""
    curve = case["curve"]
    c, fr = R.CURVES[curve], R.CURVES[curve]["fr"]
    G = R.generator(curve)
    return {"curve": curve, "generator": pt_js(G), "two_generator": pt_js(R.ec_add(curve, G, G)),
            "fq_modulus": hex(R.FIELDS[c["fq"]]["p"]), "fr_modulus": hex(R.FIELDS[fr]["p"]),
            "fr_multiplicative_generator": hex(R.FIELDS[fr]["gen"]), "fr_two_adicity": R.two_adicity(R.FIELDS[fr]["p"]),
            "fr_two_adic_root_of_unity": hex(R.two_adic_root(fr)), "fr_root_of_unity_2p11": hex(R.root_of_unity(fr, 11)),
            "gen_scalars_seed_0x5eed0001_first4": [hex(v) for v in R.gen_scalars(fr, 0x5EED0001, 4)],
            "gen_bases_first3": [pt_js(P) for P in R.gen_bases(curve, 3)]}


def compute_serialize(case):
    """ark-serialize's byte conventions as oracle/pyref.py restates them (the C++ oracle and the library's decoder are tied to pyref
    by tests/test_oracle_cpu.py and tests/test_srs_gpu.py)."""
    curve = case["curve"]
    fr = R.CURVES[curve]["fr"]
    base = R.gen_bases(curve, 6)
    pts = base + [R.ec_neg(curve, P) for P in base] + [None]
    frs = R.gen_scalars(fr, 0x5E71A11E, 3)
    vec_fr = len(frs).to_bytes(8, "little") + b"".join(R.ser_field(fr, v) for v in frs)
    return {"curve": curve,
            "points": [{"point": pt_js(P), "uncompressed": R.ser_point(curve, P).hex(), "compressed": R.ser_point_compressed(curve, P).hex()} for P in pts],
            "fr": [{"value": hex(v), "bytes": R.ser_field(fr, v).hex()} for v in frs],
            "vec_of_3_points_compressed": R.ser_g1_vec(curve, base[:3], True).hex(),
            "vec_of_3_points_uncompressed": R.ser_g1_vec(curve, base[:3], False).hex(),
            "vec_of_3_fr_compressed": vec_fr.hex()}


def compute_msm(engine, case):
    b, s = msm_inputs(case)
    return {"result": pt_js(engine.msm(case["curve"], b, s))}


def compute_kzg(engine, case):
    powers, co, z = kzg_inputs(case)
    c, w, v = engine.kzg(case["curve"], powers, co, z)
    curve = case["curve"]
    out = {"commitment": pt_js(c), "proof_w": pt_js(w), "value": fr_js(v),
           "commitment_compressed": R.ser_point_compressed(curve, c).hex(),
           # kzg10::Proof {w, random_v: Option<Fr>}: the point, then Option's tag byte 0 for None
           "proof_compressed": (R.ser_point_compressed(curve, w) + b"\x00").hex()}
    return out


def marlin_inputs(case):
    curve, n = case["curve"], case["n"]
    polys = [scalars_canonical(curve, case["seed0"] + j, d + 1) for j, d in enumerate(case["degrees"])]
    z = scalars_canonical(curve, case["z_seed"], 1)[0]
    return bases(curve, n), polys, z


def compute_marlin_open(engine, case):
    """MarlinKZG10::commit per polynomial, open = KZG10::open of sum_j xi_j p_j (marlin_pc/mod.rs:281-309); xi from the file."""
    curve = case["curve"]
    powers, polys, z = marlin_inputs(case)
    xi = [h2i(x) for x in case["opening_challenges"]]
    comms, values = [], []
    for q in polys:
        c, _, v = engine.kzg(curve, powers, q, z)
        comms.append(pt_js(c))
        values.append(fr_js(v))
    comb = O.ints_to_limbs(engine.lincomb(curve, polys, xi), 4)
    _, w, _ = engine.kzg(curve, powers, comb, z)
    return {"commitments": comms, "values": values, "proof_w": pt_js(w)}


def ipa_inputs(case):
    curve, n = case["curve"], 1 << case["log_n"]
    allb = bases(curve, n + 2)
    polys = [scalars_canonical(curve, case["seed0"] + j, d + 1) for j, d in enumerate(case["degrees"])]
    z = O.limbs_to_ints(scalars_canonical(curve, case["z_seed"], 1))[0]
    return np.ascontiguousarray(allb[:n]), np.ascontiguousarray(allb[n]), polys, z


def compute_ipa(engine, case):
    curve = case["curve"]
    key, h, polys, z = ipa_inputs(case)
    xi = [h2i(x) for x in case["opening_challenges"]]
    comm_pts = [engine.msm(curve, key[:len(q)], q) for q in polys]            # cm_commit, hiding off (ipa_pc/mod.rs:54-72)
    l, r, fk, c = engine.ipa_open(curve, key, h, polys, comm_pts, xi, z)
    return {"commitments": [pt_js(P) for P in comm_pts], "l_vec": [pt_js(P) for P in l], "r_vec": [pt_js(P) for P in r],
            "final_comm_key": pt_js(fk), "c": fr_js(c), "hiding_comm_is_none": True}


def compute_reed_solomon(engine, case):
    curve, m = case["field"], case["m"]
    log_n = (m * case["rho_inv"] - 1).bit_length()
    msg = scalars_canonical(curve, case["seed"], m)
    out = engine.ntt_rows(curve, msg.reshape(1, m, 4), log_n)[0]
    return {"output": [hex(v) for v in O.limbs_to_ints(out)]}


def compute_ligero(engine, case):
    curve, n = case["field"], case["poly_len"]
    n_rows, n_cols, _ = O.ligero_dims(FIELD_BITS[curve], n, case["rho_inv"], case["sec_param"])
    log_n = (n_cols * case["rho_inv"] - 1).bit_length()
    root = engine.ligero_root(curve, scalars_canonical(curve, case["seed"], n), n_rows, n_cols, log_n)
    # CanonicalSerialize of LinCodePCCommitment {metadata: {n_rows, n_cols, n_ext_cols}, root: Vec<u8>} (linear_codes/data_structures.rs:84-102)
    blob = b"".join(int(v).to_bytes(8, "little") for v in (n_rows, n_cols, 1 << log_n)) + len(root).to_bytes(8, "little") + root
    return {"commitment_uncompressed": blob.hex()}


SECTIONS = {"msm": compute_msm, "kzg": compute_kzg, "marlin_open": compute_marlin_open, "ipa": compute_ipa,
            "reed_solomon": compute_reed_solomon, "ligero": compute_ligero}
STATIC_SECTIONS = {"constants": compute_constants, "serialize": compute_serialize}


def case_pairs(section, case):
    """Size of a case, for engines that only take small ones."""
    if section == "msm":
        return case["n"]
    if section == "kzg":
        return case["degree"] + 1
    if section == "marlin_open":
        return case["n"]
    if section == "ipa":
        return (1 << case["log_n"]) * (8 if case["log_n"] > 4 else 1)
    if section == "reed_solomon":
        return case["m"] * case["rho_inv"] // 4
    if section == "ligero":
        return case["poly_len"] * 4
    return 0


def diff(got, case):
    """Keys of `got` whose value differs from the file's."""
    return [k for k, v in got.items() if _norm(v) != _norm(case.get(k, "<missing>"))]


def check_section(engine, section, case):
    return diff(SECTIONS[section](engine, case), case)


def check_static(section, case):
    return diff(STATIC_SECTIONS[section](case), case)
