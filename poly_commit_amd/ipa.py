"""InnerProductArgPC::open halving loop (poly-commit/src/ipa_pc/mod.rs:664-711) on one GPU.

All vectors (comm_key, coefficients, powers of z) stay in HBM for the whole proof; per round
only the two 64-byte points L, R come down and one challenge goes up.  The Fiat-Shamir hash
that produces the challenge (`compute_random_oracle_challenge`, :74-87, Blake2s over
ark-serialize bytes) is host work on two points and stays with the caller: it is passed in as
`next_challenge(l_xy, r_xy) -> u (Montgomery Fr limbs)`.
"""
import hashlib
import os

import numpy as np

from . import _ffi
from .sharded import FR_MODULUS, _R, _int_to_limbs, _limbs_to_int

# base-field moduli: needed only to serialise points for the transcript
FQ_MODULUS = {
    "bls12_381": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    "bn254": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "pallas": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
}


# ---- Fiat-Shamir transcript (ipa_pc/mod.rs:74-87, 615-625, 681-688); byte conventions of ark-serialize /
# ark-ff / ark-ec / ark-bls12-381 0.5 restated from their published behaviour (see host/transcript.hpp) ----
def ser_fr(curve, limbs_mont):
    p = FR_MODULUS[curve]
    v = _limbs_to_int(limbs_mont) * pow(_R, -1, p) % p
    return v.to_bytes((p.bit_length() + 7) // 8, "little")


def ser_point(curve, xy_mont):
    q = FQ_MODULUS[curve]
    nq = len(xy_mont) // 2
    rq = 1 << (64 * nq)
    xb, yb = (q.bit_length() + 7) // 8, (q.bit_length() + 2 + 7) // 8
    inf = not np.asarray(xy_mont).any()
    x = _limbs_to_int(xy_mont[:nq]) * pow(rq, -1, q) % q
    y = _limbs_to_int(xy_mont[nq:]) * pow(rq, -1, q) % q
    if curve == "bls12_381":
        if inf:
            return bytes([0x40]) + bytes(2 * xb - 1)
        return x.to_bytes(xb, "big") + y.to_bytes(xb, "big")
    out = bytearray(xb + yb)
    if inf:
        out[-1] |= 0x40
        return bytes(out)
    out[:] = x.to_bytes(xb, "little") + y.to_bytes(yb, "little")
    if y > (q - y) % q:                 # SWFlags::YIsNegative (0x80): y is the larger of {y, -y}; YIsPositive sets no bit
        out[-1] |= 0x80
    return bytes(out)


def random_oracle_challenge(curve, data):
    """compute_random_oracle_challenge with D = Blake2s: returns the challenge as Montgomery limbs."""
    p = FR_MODULUS[curve]
    i = 0
    while True:
        h = hashlib.blake2s(data + i.to_bytes(8, "little")).digest()
        v = int.from_bytes(h, "little") & ((1 << p.bit_length()) - 1)
        if v < p:
            return _int_to_limbs(v * _R % p)
        i += 1


def ipa_open(ctx, curve, comm_key, h_xy, polys_dev, lens, comms, point_mont, opening_challenges, timings=None):
    """InnerProductArgPC::open without hiding / degree bounds (ipa_pc/mod.rs:475-723): combine with the sponge's
    opening challenges (supplied), derive the random-oracle challenges from the transcript, run the rounds.
    polys_dev: device pointers of the coefficient vectors; comms: their commitments (x||y).
    Returns (l_vec, r_vec, final_comm_key, c)."""
    import torch
    n = comm_key.n if isinstance(comm_key, _ffi.Srs) else comm_key.shape[0]
    p = FR_MODULUS[curve]
    xi = np.ascontiguousarray(opening_challenges, dtype=np.uint64)
    comb = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.fr_lincomb(curve, list(polys_dev), xi, n_out=n, out=comb.data_ptr(), lens=list(lens))
    ccomm = _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, c, x) for c, x in zip(comms, xi)]))
    v = ctx.poly_eval(curve, comb.data_ptr(), point_mont, n=n)
    rc = random_oracle_challenge(curve, ser_point(curve, ccomm) + ser_fr(curve, point_mont) + ser_fr(curve, v))
    h_prime = _ffi.point_mul(curve, np.ascontiguousarray(h_xy), rc)
    state = {"rc": rc}

    def next_challenge(l, r):
        state["rc"] = random_oracle_challenge(curve, ser_fr(curve, state["rc"]) + ser_point(curve, l) + ser_point(curve, r))
        return state["rc"]
    return ipa_open_rounds(ctx, curve, comm_key, comb, n, point_mont, h_prime, next_challenge, timings), rc


# rounds with n <= this keep the key and fold per-base factors instead (pc_hip_ipa_key_scalars).  Round 6: 2^16 -- one more ladder fold
# (2^16 elements: ~2 ms, its latency floor) buys 16 rounds of two 2^16-pair MSMs instead of 2^17-pair ones (0.85 against 1.05 ms per
# round): 73.4 against 74.9 ms per opening at 2^22; a window table on the fixed key makes the rounds 0.70-0.76 ms but costs 7-8 ms to
# build per opening (measured, dropped)
FIXED_KEY_BELOW = 1 << 16


def library_fixed_key_below():
    """The default of pc_hip_ipa_open_rounds itself: inside the library the fixed key is a key object with its own window table, refilled
    per opening (rounds of 0.57 ms on 2^17 points, 0.53 on 2^16), and the switch pays one size earlier: 2^17 -- 56.6 ms per opening at
    2^22 against 57.5 (2^16) and 62.2 (2^18).  PC_HIP_IPA_FIXED_TABLE=0 (no such table): 2^16, as in the loop below."""
    return 1 << (16 if os.environ.get("PC_HIP_IPA_FIXED_TABLE") == "0" else 17)


def ipa_open_rounds(ctx, curve, comm_key, coeffs_dev, n, point_mont, h_prime_xy, next_challenge, timings=None,
                    fixed_key_below=None, python_loop=None):
    """comm_key: n x (x||y) host array, or a resident Srs (it is cloned on the device, not consumed); coeffs_dev: torch cuda int64 tensor (n,4), Montgomery,
    CONSUMED (folded in place).  Returns (l_vec, r_vec, final_comm_key, c) as numpy arrays.
    python_loop: False (default; PC_IPA_PY_LOOP=1 flips it) = the library's own loop, pc_hip_ipa_open_rounds; True = the same sequence driven
    from here through the round-by-round entry points (the per-phase `timings` of bench.py's breakdown come from this form)."""
    if python_loop is None:
        python_loop = os.environ.get("PC_IPA_PY_LOOP", "0") == "1"
    if fixed_key_below is None:
        fixed_key_below = FIXED_KEY_BELOW if python_loop else library_fixed_key_below()
    import time
    import torch
    assert n & (n - 1) == 0
    if not python_loop:
        # the loop inside the library (pc_hip_ipa_open_rounds): the same calls in the same order, no host language between the rounds
        resident = isinstance(comm_key, _ffi.Srs)
        srs = comm_key if resident else ctx.upload_srs(curve, np.ascontiguousarray(comm_key))
        try:
            out = srs.ipa_open_rounds(coeffs_dev.data_ptr(), n, point_mont, np.ascontiguousarray(h_prime_xy), next_challenge,
                                      fixed_key_below if fixed_key_below >= 2 else 1, want_times=timings is not None)
        finally:
            if not resident:
                srs.free()
        if timings is not None:
            l, r, fk, c, rms, fms = out
            levels = srs.fold_table_info()[0] if resident else 0
            two = resident and n == srs.n and n >= 8 and n // 2 > fixed_key_below and levels == 2
            timings["per_round_ms"] = [round(x, 3) for x in rms]
            kinds, folds, m = [], [], n
            for k, ms in enumerate(fms):
                h = m // 2
                if m > fixed_key_below:
                    kind = ("deferred" if k == 0 else "table2") if (two and k < 2) else ("table1" if (k == 0 and resident and levels == 1 and srs.n == n) else "ladder")
                    kinds.append(kind); folds.append((h, round(ms, 3)))
                m = h
            timings["ec_fold_per_round_ms"], timings["ec_fold_kind"] = folds, kinds
            timings["ec_fold"] = sum(fms)
            return l, r, fk, c
        return out

    class _T:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            self.t = time.perf_counter()

        def __exit__(self, *a):
            if timings is not None:
                timings[self.name] = timings.get(self.name, 0.0) + (time.perf_counter() - self.t) * 1e3
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    with _T("upload_key"):
        # A resident committer key is never modified and never copied: the first round's MSMs run on it (with its window table, if
        # built), its first fold goes OUT OF PLACE into a new key of half the size (pc_hip_ec_fold_from: from the key's fold table,
        # if built), the later folds act on that key in place.  A key handed over as a host array is uploaded (a private copy).
        resident = isinstance(comm_key, _ffi.Srs)
        srs = comm_key if resident else ctx.upload_srs(curve, np.ascontiguousarray(comm_key))
        owned = not resident
    h_prime_xy = np.ascontiguousarray(h_prime_xy)
    z = torch.empty((n, 4), dtype=torch.int64, device=coeffs_dev.device)
    ctx.fr_powers(curve, point_mont, n, z.data_ptr())
    cptr, zptr = coeffs_dev.data_ptr(), z.data_ptr()
    l_vec, r_vec = [], []
    n0, s_dev, alr = 0, None, None
    # PC_IPA_FIXED_MANY=1: the two MSMs of a fixed-key round as ONE two-row pc_hip_msm_many pass.  Measured slower than two
    # pipelined pc_hip_msm_async calls (1.40 vs 1.20 ms per round at n0 = 2^17, plus 8 ms for the pass's window table): off.
    many = os.environ.get("PC_IPA_FIXED_MANY", "0") == "1"
    # the inner products of the first round; later rounds get theirs from the fused fold (pc_hip_ipa_fold_dots)
    with _T("fold_dots"):
        dots = ctx.ipa_fold_dots(curve, cptr, zptr, n)
    u_prev = None
    # A resident committer key with a TWO-level fold table (pc_hip_srs_precompute_fold_ex): the key is not folded in round 1; round 2's
    # commitments run on the committer key by linearity -- with K' = K_l + u1 K_r the key after one fold and q = n / 4,
    #   MSM(K'[a .. a + q), s) = MSM(K[a .. a + q), s) + u1 MSM(K[a + 2q .. a + 3q), s)
    # -- and the key after BOTH folds then comes out of the table in one step (pc_hip_ec_fold2_from)
    two_level = resident and n >= 8 and n == srs.n and n // 2 > fixed_key_below and srs.fold_table_info()[0] == 2
    u_first, root = None, srs
    while n > 1:
        h = n // 2
        t_round = time.perf_counter()
        if not n0 and n <= fixed_key_below:
            # from here on the resident key key[0..n0) stays fixed; the folds act on the per-base factors s
            n0 = n
            one = _int_to_limbs(_R % p)
            s_dev = torch.empty((n0, 4), dtype=torch.int64, device=coeffs_dev.device)
            ctx.fr_powers(curve, one, n0, s_dev.data_ptr())                # s = (1, 1, ...)
            alr = torch.empty((2 * n0, 4), dtype=torch.int64, device=coeffs_dev.device)   # scalars of l | scalars of r
            u_prev = None                                                  # the key itself carries every fold so far
        # l = cm_commit(key_l, coeffs_r) + h' * <coeffs_r, z_l>;  r = cm_commit(key_r, coeffs_l) + h' * <coeffs_l, z_r>
        with _T("msm_enqueue"):
            if n0:
                # fold of the factors by the previous challenge (size 2n), then this round's scalar vectors (size n): one call
                ctx.ipa_key_scalars(curve, cptr, n, s_dev.data_ptr(), n0, fold_u=u_prev, fold_m=2 * n if u_prev is not None else 0,
                                    out_l_dev=alr.data_ptr(), out_r_dev=alr.data_ptr() + 32 * n0)
                if not many:
                    jl = srs.msm_async(alr.data_ptr(), n=n0, base_offset=0, montgomery=True)
                    jr = srs.msm_async(alr.data_ptr() + 32 * n0, n=n0, base_offset=0, montgomery=True)
            elif u_first is not None:
                # round 2 on the committer key (its window table serves it), by linearity: pc_hip_ipa_round2_msms -- two MSMs of 3q pairs,
                #   l = MSM(K[0 .. 3q), (c_r | 0 | u1 c_r)),  r = MSM(K[q .. 4q), (c_l | 0 | u1 c_l)),  q = h
                # (as four MSMs of q pairs + two host point multiplications the fourth job waits for the first of the key's three
                # pipelines: measured 0.5 ms more)
                ml, mr = srs.ipa_round2_msms(cptr, h, u_first)
            else:
                jl = srs.msm_async(cptr + 32 * h, n=h, base_offset=0, montgomery=True)
                jr = srs.msm_async(cptr, n=h, base_offset=h, montgomery=True)
        with _T("host_point_mul"):
            hl = _ffi.point_mul(curve, h_prime_xy, dots[0])            # h'.mul(inner_product): one point, host
            hr = _ffi.point_mul(curve, h_prime_xy, dots[1])
        with _T("msm_wait"):
            if n0 and many:
                pts, _ = srs.msm_many(alr.data_ptr(), m=n0, n_msms=2, base_offset=0, montgomery=True)
                ml, mr = pts[0], pts[1]
            elif u_first is None or n0:
                ml, mr = jl.wait()[0], jr.wait()[0]
            l = _ffi.points_sum(curve, np.stack([ml, hl]))
            r = _ffi.points_sum(curve, np.stack([mr, hr]))
        l_vec.append(l)
        r_vec.append(r)
        u = np.ascontiguousarray(next_challenge(l, r), dtype=np.uint64)
        ui = pow(_limbs_to_int(u) * rinv % p, -1, p) * _R % p          # u^-1, Montgomery
        with _T("fold_dots"):
            # coeffs_l += u^-1 coeffs_r, z_l += u z_r, and the next round's two inner products in the same pass
            dots = ctx.ipa_fold_dots(curve, cptr, zptr, h, u, _int_to_limbs(ui))
        t_fold = time.perf_counter()
        with _T("ec_fold"):
            if n0:
                u_prev, kind = u, None                                      # applied to the factors at the top of the next round
            elif two_level and u_first is None and srs is root:
                u_first, kind = u, "deferred"                               # round 1: the key stays; round 2 runs on it as well
            elif u_first is not None:
                srs, owned = root.fold2_from(h, u_first, u), True           # round 2: the key after both folds, from the table
                u_first, two_level, kind = None, False, "table2"
            elif owned:
                srs.ec_fold(h, u)                                           # key_l += u key_r, normalised
                kind = "ladder"
            else:
                kind = "table1" if (srs.fold_table_info()[0] == 1 and srs.n == 2 * h) else "ladder"
                srs, owned = srs.fold_from(h, u), True                      # the same fold, out of place: the committer key stays
        if timings is not None and kind is not None:                        # (blocking calls: wall time = the fold's kernels + launch)
            timings.setdefault("ec_fold_per_round_ms", []).append((h, round((time.perf_counter() - t_fold) * 1e3, 3)))
            timings.setdefault("ec_fold_kind", []).append(kind)
        if timings is not None:
            timings.setdefault("per_round_ms", []).append(round((time.perf_counter() - t_round) * 1e3, 3))
        n = h
    if n0 and u_prev is not None:
        ctx.ipa_key_scalars(curve, None, 0, s_dev.data_ptr(), n0, fold_u=u_prev, fold_m=2)     # the last fold (size 2)
    if n0:
        final_key = srs.msm(s_dev.data_ptr(), n=n0, base_offset=0, montgomery=True)[0]   # sum_j s_j K0_j
    else:
        final_key = srs.read(0, 1)[0]
    c = coeffs_dev[0].cpu().numpy().view(np.uint64).copy()
    if owned:
        srs.free()
    return np.stack(l_vec), np.stack(r_vec), final_key, c


def succinct_check_eval(curve, challenges_mont, point_mont):
    """SuccinctCheckPolynomial::evaluate (ipa_pc/data_structures.rs:223-236): prod_i (1 + u_i z^(2^(log_d - i)))
    in O(log d).  Montgomery limbs in, Montgomery limbs out."""
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    z = _limbs_to_int(point_mont) * rinv % p
    log_d = len(challenges_mont)
    prod = 1
    for i, u in enumerate(challenges_mont, start=1):
        prod = prod * (1 + pow(z, 1 << (log_d - i), p) * (_limbs_to_int(u) * rinv % p)) % p
    return _int_to_limbs(prod * _R % p)


def ipa_check(ctx, curve, comm_key, h_xy, comms, point_mont, values_mont, proof, opening_challenges):
    """InnerProductArgPC::check without hiding / degree bounds (ipa_pc/mod.rs:725-773): the succinct check
    (:91-203; O(log d) host point operations) and the verifier's only large computation, the MSM of the committer key
    with the check polynomial's coefficients (:759-765), on the device -- the coefficients never exist on the host:
    they are the per-base factors the key folds would apply (pc_hip_ipa_key_scalars folding ones by every challenge).
    comm_key: n x (x||y) host array or a resident Srs.  proof = (l_vec, r_vec, final_comm_key, c).  Returns bool."""
    import torch
    l_vec, r_vec, final_key, c = proof
    own = not isinstance(comm_key, _ffi.Srs)
    n = comm_key.shape[0] if own else comm_key.n
    log_d = (n - 1).bit_length()
    if len(l_vec) != len(r_vec) or len(l_vec) != log_d:
        raise ValueError(f"IncorrectInputLength: expected proof vectors to be {log_d}, l_vec {len(l_vec)}, r_vec {len(r_vec)}")
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    to_int = lambda a: _limbs_to_int(a) * rinv % p           # noqa: E731
    to_mont = lambda v: _int_to_limbs(v * _R % p)            # noqa: E731
    xi = np.ascontiguousarray(opening_challenges, dtype=np.uint64)
    combined_v = sum(to_int(x) * to_int(v) for x, v in zip(xi, values_mont)) % p
    ccomm = _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, cm, x) for cm, x in zip(comms, xi)]))
    rc = random_oracle_challenge(curve, ser_point(curve, ccomm) + ser_fr(curve, point_mont) + ser_fr(curve, to_mont(combined_v)))
    h_prime = _ffi.point_mul(curve, np.ascontiguousarray(h_xy), rc)
    terms = [ccomm, _ffi.point_mul(curve, h_prime, to_mont(combined_v))]
    chal = []
    for l, r in zip(l_vec, r_vec):
        rc = random_oracle_challenge(curve, ser_fr(curve, rc) + ser_point(curve, l) + ser_point(curve, r))
        chal.append(rc)
        terms.append(_ffi.point_mul(curve, l, to_mont(pow(to_int(rc), -1, p))))
        terms.append(_ffi.point_mul(curve, r, rc))
    round_comm = _ffi.points_sum(curve, np.stack(terms))
    v_prime = to_int(succinct_check_eval(curve, chal, point_mont)) * to_int(c) % p
    check_elem = _ffi.points_sum(curve, np.stack([_ffi.point_mul(curve, final_key, c), _ffi.point_mul(curve, h_prime, to_mont(v_prime))]))
    if not (round_comm == check_elem).all():
        return False
    srs = ctx.upload_srs(curve, np.ascontiguousarray(comm_key)) if own else comm_key
    try:
        s_dev = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.fr_powers(curve, to_mont(1), n, s_dev.data_ptr())
        m = n
        for u in chal:                                       # coefficient j picks up u_i iff bit (log_d - i) of j is set
            ctx.ipa_key_scalars(curve, None, 0, s_dev.data_ptr(), n, fold_u=np.ascontiguousarray(u, dtype=np.uint64), fold_m=m)
            m //= 2
        key = srs.msm(s_dev.data_ptr(), n=n, base_offset=0, montgomery=True)[0]
    finally:
        if own:
            srs.free()
    return bool((key == np.asarray(final_key)).all())


# ---- the general form: hiding and degree bounds (ipa_pc/mod.rs:403-473, 475-723, 91-203) ----------------------------
def _neg_point(curve, xy):
    q = FQ_MODULUS[curve]
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    if not xy.any():
        return xy.copy()
    nq = len(xy) // 2
    out = xy.copy()
    y = _limbs_to_int(xy[nq:])
    out[nq:] = np.frombuffer(((q - y) % q).to_bytes(8 * nq, "little"), dtype="<u8")      # Montgomery form of -y is q - (y R)
    return out


def ipa_commit_general(ctx, curve, comm_key, s_xy, coeffs_dev, n_coeffs, degree_bound=None, rand=None, shifted_rand=None, srs=None):
    """InnerProductArgPC::commit for one polynomial: (comm, shifted_comm or None).  coeffs_dev: device pointer of
    n_coeffs Montgomery coefficients; rand / shifted_rand: Montgomery limbs or None (= Randomness::empty())."""
    own = srs is None
    if own:
        srs = ctx.upload_srs(curve, np.ascontiguousarray(comm_key))
    try:
        d = srs.n - 1
        comm = srs.msm(coeffs_dev, n=n_coeffs, base_offset=0, montgomery=True)[0]
        if rand is not None:
            comm = _ffi.points_sum(curve, np.stack([comm, _ffi.point_mul(curve, s_xy, rand)]))
        shifted = None
        if degree_bound is not None:
            shifted = srs.msm(coeffs_dev, n=n_coeffs, base_offset=d - degree_bound, montgomery=True)[0]
            if shifted_rand is not None:
                shifted = _ffi.points_sum(curve, np.stack([shifted, _ffi.point_mul(curve, s_xy, shifted_rand)]))
    finally:
        if own:
            srs.free()
    return comm, shifted


def ipa_open_general(ctx, curve, comm_key, h_xy, s_xy, polys, point_mont, challenges, hiding_poly_dev=None, hiding_rand=None,
                     timings=None):
    """InnerProductArgPC::open with hiding and degree bounds.  polys: dicts {dev (torch cuda int64 (len, 4)), comm,
    shifted_comm, degree_bound, hiding, rand, shifted_rand}; challenges: the caller's sponge output in squeeze order
    (:502, then :525 and :556 per polynomial), Montgomery limbs; hiding_poly_dev (torch (d+1, 4), CONSUMED) / hiding_rand:
    what the reference draws at :577 / :579.  Returns ((l_vec, r_vec, final_comm_key, c, hiding_comm, rand), first round challenge)."""
    import torch
    n = comm_key.shape[0]
    d = n - 1
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    to_int = lambda a: _limbs_to_int(a) * rinv % p           # noqa: E731
    to_mont = lambda v: _int_to_limbs(v % p * _R % p)        # noqa: E731
    ch = iter([np.ascontiguousarray(c, dtype=np.uint64) for c in challenges])
    vecs, lens, xis, terms = [], [], [], []
    combined_rand, has_hiding = 0, False
    keep = []
    cur = next(ch)
    for q in polys:
        m = q["dev"].shape[0]
        if m - 1 > d:
            raise ValueError("TooManyCoefficients")
        db = q.get("degree_bound")
        if db is not None and (db < m - 1 or db > d):
            raise ValueError("IncorrectDegreeBound")
        vecs.append(q["dev"].data_ptr()); lens.append(m); xis.append(cur)
        terms.append(_ffi.point_mul(curve, q["comm"], cur))
        if q.get("hiding"):
            has_hiding = True
            combined_rand = (combined_rand + to_int(cur) * to_int(q["rand"])) % p
        cur = next(ch)
        if db is not None:
            sh = torch.cat([torch.zeros((d - db, 4), dtype=torch.int64, device=q["dev"].device), q["dev"]])     # shift_polynomial, :230-239
            keep.append(sh)
            vecs.append(sh.data_ptr()); lens.append(sh.shape[0]); xis.append(cur)
            terms.append(_ffi.point_mul(curve, q["shifted_comm"], cur))
            if q.get("hiding"):
                combined_rand = (combined_rand + to_int(cur) * to_int(q["shifted_rand"])) % p
        cur = next(ch)
    comb = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    ctx.fr_lincomb(curve, vecs, np.stack(xis), n_out=n, out=comb.data_ptr(), lens=lens)
    ccomm = _ffi.points_sum(curve, np.stack(terms))
    v = ctx.poly_eval(curve, comb.data_ptr(), point_mont, n=n)
    hiding_comm = None
    if has_hiding:
        hp = hiding_poly_dev
        assert hp is not None and hp.shape[0] == n and hiding_rand is not None
        hv = ctx.poly_eval(curve, hp.data_ptr(), point_mont, n=n)                               # :578
        h0 = hp[0].cpu().numpy().view(np.uint64)
        hp[0] = torch.from_numpy(to_mont(to_int(h0) - to_int(hv)).view(np.int64)).to(hp.device)
        srs = ctx.upload_srs(curve, np.ascontiguousarray(comm_key))
        hiding_comm = _ffi.points_sum(curve, np.stack([srs.msm(hp.data_ptr(), n=n, montgomery=True)[0],
                                                       _ffi.point_mul(curve, s_xy, hiding_rand)]))          # :580-585
        srs.free()
        hc = random_oracle_challenge(curve, ser_point(curve, ccomm) + ser_fr(curve, point_mont) + ser_fr(curve, v) + ser_point(curve, hiding_comm))
        comb2 = torch.empty_like(comb)
        ctx.fr_lincomb(curve, [comb.data_ptr(), hp.data_ptr()], np.stack([to_mont(1), hc]), n_out=n, out=comb2.data_ptr(), lens=[n, n])
        comb = comb2
        combined_rand = (combined_rand + to_int(hc) * to_int(hiding_rand)) % p
        ccomm = _ffi.points_sum(curve, np.stack([ccomm, _ffi.point_mul(curve, hiding_comm, hc),
                                                 _neg_point(curve, _ffi.point_mul(curve, s_xy, to_mont(combined_rand)))]))   # :606-607
    rc = random_oracle_challenge(curve, ser_point(curve, ccomm) + ser_fr(curve, point_mont) + ser_fr(curve, v))
    h_prime = _ffi.point_mul(curve, np.ascontiguousarray(h_xy), rc)
    state = {"rc": rc}

    def next_challenge(l, r):
        state["rc"] = random_oracle_challenge(curve, ser_fr(curve, state["rc"]) + ser_point(curve, l) + ser_point(curve, r))
        return state["rc"]
    l, r, fk, c = ipa_open_rounds(ctx, curve, comm_key, comb, n, point_mont, h_prime, next_challenge, timings)
    return (l, r, fk, c, hiding_comm, (to_mont(combined_rand) if has_hiding else None)), rc


def ipa_check_general(ctx, curve, comm_key, h_xy, s_xy, comms, point_mont, values_mont, proof, challenges):
    """InnerProductArgPC::check with hiding and degree bounds: comms = dicts {comm, shifted_comm, degree_bound};
    proof = (l_vec, r_vec, final_comm_key, c, hiding_comm, rand).  The combination of :116-151 happens here, the rest is
    ipa_check's (succinct equation on the host, final-key MSM on the device)."""
    n = comm_key.shape[0]
    d = n - 1
    p = FR_MODULUS[curve]
    rinv = pow(_R, -1, p)
    to_int = lambda a: _limbs_to_int(a) * rinv % p           # noqa: E731
    to_mont = lambda v: _int_to_limbs(v % p * _R % p)        # noqa: E731
    l_vec, r_vec, final_key, c, hiding_comm, rand = proof
    ch = iter([np.ascontiguousarray(x, dtype=np.uint64) for x in challenges])
    z = to_int(point_mont)
    combined_v, terms = 0, []
    cur = next(ch)
    for cm, value in zip(comms, values_mont):
        combined_v = (combined_v + to_int(cur) * to_int(value)) % p
        terms.append(_ffi.point_mul(curve, cm["comm"], cur))
        cur = next(ch)
        db = cm.get("degree_bound")
        assert (db is not None) == (cm.get("shifted_comm") is not None)
        if db is not None:
            combined_v = (combined_v + to_int(cur) * to_int(value) % p * pow(z, d - db, p)) % p
            terms.append(_ffi.point_mul(curve, cm["shifted_comm"], cur))
        cur = next(ch)
    ccomm = _ffi.points_sum(curve, np.stack(terms))
    assert (hiding_comm is not None) == (rand is not None)
    if hiding_comm is not None:
        hc = random_oracle_challenge(curve, ser_point(curve, ccomm) + ser_fr(curve, point_mont) + ser_fr(curve, to_mont(combined_v))
                                     + ser_point(curve, hiding_comm))
        ccomm = _ffi.points_sum(curve, np.stack([ccomm, _ffi.point_mul(curve, hiding_comm, hc), _neg_point(curve, _ffi.point_mul(curve, s_xy, rand))]))
    # from here on it is the plain check of ONE commitment `ccomm` with value combined_v and opening challenge 1
    return ipa_check(ctx, curve, comm_key, h_xy, [ccomm], point_mont, [to_mont(combined_v)], (l_vec, r_vec, final_key, c), [to_mont(1)])
