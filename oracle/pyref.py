"""Pure-Python big-int reference for the commit/open hot path of arkworks poly-commit.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Nothing in the product path imports this.

This is the *second opinion* behind the C++ oracle (oracle/oracle.cpp): every primitive is
re-derived here with Python arbitrary-precision integers, independently of both the C++
oracle (64-bit limbs) and the HIP code (32-bit limbs).  It also generates the golden
fixtures under tests/golden/ (tools/gen_golden.py).

Reference anchors (relative to /root/reference):
  * MSM semantics            poly-commit/src/kzg10/mod.rs:175-178 (msm_bigint: min(len) pairs)
  * KZG10::commit / open     poly-commit/src/kzg10/mod.rs:157-210, 217-310, 452-470
  * IPA halving rounds       poly-commit/src/ipa_pc/mod.rs:54-72, 664-711
  * reed_solomon / NTT       poly-commit/src/linear_codes/utils.rs:112-127, pinned by
                             test_reed_solomon :303-331 (natural order, arkworks omega)
  * Ligero dimensions        poly-commit/src/linear_codes/ligero.rs:118-128,
                             poly-commit/src/linear_codes/utils.rs:156-184
  * column digests / Merkle  poly-commit/src/linear_codes/mod.rs:256-275, 506-521
  * open's linear combination poly-commit/src/marlin/marlin_pc/mod.rs:281-287
The field/curve arithmetic itself lives in crates.io ark-ff/ark-ec/ark-poly 0.5 (not in
/root/reference); it is restated from the published definitions of the curves.
"""
import math

# ----------------------------------------------------------------------------------------
# Fields.  name -> (modulus, multiplicative generator used by arkworks, two-adicity)
# ----------------------------------------------------------------------------------------
FIELDS = {
    "bls12_381_fq": dict(
        p=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        gen=2, limbs64=6),
    "bls12_381_fr": dict(
        p=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
        gen=7, limbs64=4),
    "bn254_fq": dict(
        p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        gen=3, limbs64=4),
    "bn254_fr": dict(
        p=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        gen=5, limbs64=4),
    # ark-pallas naming: Fq = base field of Pallas, Fr = scalar field of Pallas.
    "pallas_fq": dict(
        p=0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
        gen=5, limbs64=4),
    "pallas_fr": dict(
        p=0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
        gen=5, limbs64=4),
}


def two_adicity(p):
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    return s


def two_adic_root(field):
    """arkworks FftField::TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((p-1)/2^s)."""
    f = FIELDS[field]
    p = f["p"]
    s = two_adicity(p)
    return pow(f["gen"], (p - 1) >> s, p)


def root_of_unity(field, log_n):
    """omega of the radix-2 domain of size 2^log_n (ark-poly Radix2EvaluationDomain::new):
    TWO_ADIC_ROOT_OF_UNITY ^ (2^(s - log_n))."""
    p = FIELDS[field]["p"]
    s = two_adicity(p)
    assert log_n <= s
    return pow(two_adic_root(field), 1 << (s - log_n), p)


# ----------------------------------------------------------------------------------------
# Curves (all short Weierstrass, a = 0):  y^2 = x^3 + b over Fq, scalar field Fr.
# ----------------------------------------------------------------------------------------
CURVES = {
    "bls12_381": dict(
        fq="bls12_381_fq", fr="bls12_381_fr", b=4,
        gx=0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        gy=0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    "bn254": dict(fq="bn254_fq", fr="bn254_fr", b=3, gx=1, gy=2),
    "pallas": dict(
        fq="pallas_fq", fr="pallas_fr", b=5,
        gx=FIELDS["pallas_fq"]["p"] - 1, gy=2),
}

INF = None  # point at infinity


def curve_params(curve):
    c = CURVES[curve]
    return FIELDS[c["fq"]]["p"], FIELDS[c["fr"]]["p"], c["b"]


def on_curve(curve, P):
    if P is INF:
        return True
    p, _, b = curve_params(curve)
    x, y = P
    return (y * y - x * x * x - b) % p == 0


def ec_neg(curve, P):
    if P is INF:
        return INF
    p = curve_params(curve)[0]
    return (P[0], (-P[1]) % p)


def ec_add(curve, P, Q):
    p = curve_params(curve)[0]
    if P is INF:
        return Q
    if Q is INF:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def ec_mul(curve, k, P):
    R = INF
    Q = P
    while k:
        if k & 1:
            R = ec_add(curve, R, Q)
        Q = ec_add(curve, Q, Q)
        k >>= 1
    return R


def generator(curve):
    c = CURVES[curve]
    return (c["gx"], c["gy"])


def msm(curve, bases, scalars):
    """Sum k_i * P_i over min(len) pairs (ark-ec msm_bigint truncation semantics)."""
    acc = INF
    for P, k in zip(bases, scalars):
        acc = ec_add(curve, acc, ec_mul(curve, k, P))
    return acc


def gen_bases(curve, n):
    """Synthetic SRS stand-in P_i = (i+1)*G."""
    G = generator(curve)
    out, cur = [], G
    for _ in range(n):
        out.append(cur)
        cur = ec_add(curve, cur, G)
    return out


# ----------------------------------------------------------------------------------------
# Deterministic scalar generator (SplitMix64, rejection sampled) -- SURVEY.md section 8d.
# ----------------------------------------------------------------------------------------
M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)


def gen_scalars(field, seed, n):
    """n field elements uniform in [0,p): four u64 limbs (LE), top limb masked to the
    modulus bit length, rejection sampled."""
    p = FIELDS[field]["p"]
    nl = FIELDS[field]["limbs64"]
    bits = p.bit_length()
    top_mask = (1 << (bits - 64 * (nl - 1))) - 1
    rng = SplitMix64(seed)
    out = []
    while len(out) < n:
        limbs = [rng.next() for _ in range(nl)]
        limbs[-1] &= top_mask
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < p:
            out.append(v)
    return out


# ----------------------------------------------------------------------------------------
# NTT / Reed-Solomon
# ----------------------------------------------------------------------------------------
def ntt_naive(field, coeffs, log_n):
    """out[j] = sum_i coeffs[i] * omega^(i*j), natural order (O(n*m))."""
    p = FIELDS[field]["p"]
    n = 1 << log_n
    w = root_of_unity(field, log_n)
    out = []
    for j in range(n):
        wj = pow(w, j, p)
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * wj + c) % p
        out.append(acc)
    return out


def ntt(field, coeffs, log_n):
    """Iterative radix-2 DIT, natural in / natural out."""
    p = FIELDS[field]["p"]
    n = 1 << log_n
    a = list(coeffs) + [0] * (n - len(coeffs))
    # bit reversal
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = root_of_unity(field, length.bit_length() - 1)
        for i in range(0, n, length):
            w = 1
            for k in range(length // 2):
                u = a[i + k]
                v = a[i + k + length // 2] * w % p
                a[i + k] = (u + v) % p
                a[i + k + length // 2] = (u - v) % p
                w = w * wl % p
        length <<= 1
    return a


def reed_solomon(field, msg, rho_inv):
    """linear_codes/utils.rs:112-127: domain = next_pow2(len*rho_inv); fft(msg)."""
    m = len(msg)
    size = 1
    while size < m * rho_inv:
        size <<= 1
    return ntt(field, msg, size.bit_length() - 1)


# ----------------------------------------------------------------------------------------
# Ligero shape (linear_codes/utils.rs:156-184, ligero.rs:118-128)
# ----------------------------------------------------------------------------------------
def ceil_div(a, b):
    return (a + b - 1) // b


def ark_log2(x):
    """ark_std::log2 = ceil(log2 x), 0 for x <= 1."""
    if x <= 1:
        return 0
    return (x - 1).bit_length()


def _is_normal(x):
    return x == x and abs(x) != float("inf") and abs(x) >= 2.2250738585072014e-308


def calculate_t(field_bits, sec_param, distance, codeword_len):
    """linear_codes/utils.rs:156-184; Err(InvalidParameters) -> ValueError."""
    residual = codeword_len / 2.0 ** field_bits
    arg = 2.0 ** (-sec_param) - residual
    rhs = math.log2(arg) if arg > 0 else float("nan")          # f64::log2 of <= 0 is NaN / -inf: not normal
    if not _is_normal(rhs):
        raise ValueError("the field is not big enough for this codeword length and security parameter")
    nom = rhs - 1.0
    darg = 1.0 - 0.5 * distance[0] / distance[1]
    denom = math.log2(darg) if darg > 0 else float("nan")
    if not _is_normal(denom):
        raise ValueError("the distance is wrong")
    t = math.ceil(nom / denom)
    return t if t < codeword_len else codeword_len


def ligero_dimensions(field, poly_len, rho_inv=4, sec_param=128):
    bits = FIELDS[field]["p"].bit_length()
    t = calculate_t(bits, sec_param, (rho_inv - 1, rho_inv), poly_len)
    n = 1 << ark_log2(math.ceil(math.sqrt(ceil_div(2 * poly_len, t))))
    m = ceil_div(poly_len, n)
    return n, m, t


# ----------------------------------------------------------------------------------------
# KZG10 commit / open (hiding off), restating kzg10/mod.rs
# ----------------------------------------------------------------------------------------
def skip_leading_zeros(coeffs):
    lz = 0
    while lz < len(coeffs) and coeffs[lz] == 0:
        lz += 1
    return lz, coeffs[lz:]


def kzg_commit(curve, powers_of_g, coeffs):
    lz, plain = skip_leading_zeros(coeffs)
    return msm(curve, powers_of_g[lz:], plain)


def witness_polynomial(field, coeffs, z):
    """Quotient of p(x) by (x - z) (kzg10/mod.rs:217-240); synthetic division."""
    p = FIELDS[field]["p"]
    n = len(coeffs)
    if n <= 1:
        return []
    q = [0] * (n - 1)
    acc = 0
    for i in range(n - 1, 0, -1):
        acc = (coeffs[i] + z * acc) % p
        q[i - 1] = acc
    return q


def kzg_open(curve, powers_of_g, coeffs, z):
    fr = CURVES[curve]["fr"]
    return kzg_commit(curve, powers_of_g, witness_polynomial(fr, coeffs, z))


def poly_eval(field, coeffs, z):
    p = FIELDS[field]["p"]
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * z + c) % p
    return acc


# ----------------------------------------------------------------------------------------
# IPA halving rounds with an externally supplied challenge list (ipa_pc/mod.rs:664-711).
# ----------------------------------------------------------------------------------------
def ipa_rounds(curve, comm_key, coeffs, z_point, h_prime, challenges):
    """Returns (l_vec, r_vec, final_key, final_coeff)."""
    fr = FIELDS[CURVES[curve]["fr"]]["p"]
    n = len(coeffs)
    assert n == len(comm_key) and n & (n - 1) == 0
    zs = [pow(z_point, i, fr) for i in range(n)]
    key = list(comm_key)
    cs = list(coeffs)
    l_vec, r_vec = [], []
    rnd = 0
    while n > 1:
        h = n // 2
        ip_l = sum(a * b for a, b in zip(cs[h:n], zs[:h])) % fr
        ip_r = sum(a * b for a, b in zip(cs[:h], zs[h:n])) % fr
        l = ec_add(curve, msm(curve, key[:h], cs[h:n]), ec_mul(curve, ip_l, h_prime))
        r = ec_add(curve, msm(curve, key[h:n], cs[:h]), ec_mul(curve, ip_r, h_prime))
        l_vec.append(l)
        r_vec.append(r)
        u = challenges[rnd]
        rnd += 1
        ui = pow(u, -1, fr)
        for i in range(h):
            cs[i] = (cs[i] + ui * cs[h + i]) % fr
            zs[i] = (zs[i] + u * zs[h + i]) % fr
            key[i] = ec_add(curve, key[i], ec_mul(curve, u, key[h + i]))
        n = h
    return l_vec, r_vec, key[0], cs[0]


# ----------------------------------------------------------------------------------------
# Fiat-Shamir transcript of InnerProductArgPC::open (ipa_pc/mod.rs:74-87, 615-625, 681-688).
# The byte conventions live in ark-serialize / ark-ff / ark-ec 0.5 (not under /root/reference): restated
# from their published behaviour -- see oracle/oracle.cpp (ser_point) for the statement.
# ----------------------------------------------------------------------------------------
def ser_field(field, v):
    nbytes = (FIELDS[field]["p"].bit_length() + 7) // 8
    return int(v).to_bytes(nbytes, "little")


def ser_point(curve, P):
    """serialize_uncompressed of a G1 affine point."""
    fq = CURVES[curve]["fq"]
    p = FIELDS[fq]["p"]
    xb = (p.bit_length() + 7) // 8
    yb = (p.bit_length() + 2 + 7) // 8
    if curve == "bls12_381":                       # ark-bls12-381: zcash / IETF encoding, big-endian
        if P is None:
            return bytes([0x40]) + bytes(2 * xb - 1)
        return P[0].to_bytes(xb, "big") + P[1].to_bytes(xb, "big")
    if P is None:
        out = bytearray(xb + yb)
        out[-1] |= 0x40
        return bytes(out)
    out = bytearray(P[0].to_bytes(xb, "little") + P[1].to_bytes(yb, "little"))
    if P[1] > (p - P[1]) % p:                      # SWFlags::from_y_coordinate: YIsPositive (no bit) for y <= -y, YIsNegative (0x80) for y > -y
        out[-1] |= 0x80
    return bytes(out)


def ser_point_compressed(curve, P):
    """serialize_compressed of a G1 affine point (ark-ec generic SW / ark-bls12-381's zcash encoding)."""
    fq = CURVES[curve]["fq"]
    p = FIELDS[fq]["p"]
    xb = (p.bit_length() + 7) // 8
    yb = (p.bit_length() + 2 + 7) // 8
    if curve == "bls12_381":
        if P is None:
            return bytes([0xC0]) + bytes(xb - 1)
        out = bytearray(P[0].to_bytes(xb, "big"))
        out[0] |= 0x80
        if P[1] > (p - P[1]) % p:
            out[0] |= 0x20
        return bytes(out)
    out = bytearray(yb)
    if P is None:
        out[-1] |= 0x40
        return bytes(out)
    out[:] = P[0].to_bytes(yb, "little")
    if P[1] > (p - P[1]) % p:                      # YIsNegative: y is the larger of {y, -y}
        out[-1] |= 0x80
    return bytes(out)


def ser_g1_vec(curve, pts, compressed):
    """ark-serialize of Vec<G1Affine>: u64 LE length, then the points -- the head of kzg10::UniversalParams
    (kzg10/data_structures.rs:57-77)."""
    f = ser_point_compressed if compressed else ser_point
    return len(pts).to_bytes(8, "little") + b"".join(f(curve, P) for P in pts)


def from_random_bytes(field, b):
    """Field::from_random_bytes: low 8*N bytes little-endian, bits above the modulus size cleared; None if >= p."""
    p = FIELDS[field]["p"]
    n64 = FIELDS[field]["limbs64"]
    v = int.from_bytes(b[: 8 * n64].ljust(8 * n64, b"\0"), "little")
    v &= (1 << p.bit_length()) - 1
    return v if v < p else None


def random_oracle_challenge(field, data):
    """compute_random_oracle_challenge with D = Blake2s (ipa_pc/mod.rs:74-87)."""
    import hashlib
    i = 0
    while True:
        c = from_random_bytes(field, hashlib.blake2s(data + i.to_bytes(8, "little")).digest())
        if c is not None:
            return c
        i += 1


def ipa_open(curve, comm_key, h, polys, comms, point, opening_challenges):
    """InnerProductArgPC::open without hiding and degree bounds (ipa_pc/mod.rs:475-723).  polys: coefficient
    lists (canonical ints) of at most len(comm_key) coefficients; comms: their commitments; opening_challenges:
    what the caller's sponge squeezes at :502 / :525 / :556, one per polynomial (the sponge is the caller's).
    Returns (l_vec, r_vec, final_comm_key, c, round_challenges)."""
    fr = CURVES[curve]["fr"]
    p = FIELDS[fr]["p"]
    n = len(comm_key)
    combined = [0] * n
    combined_comm = None
    for poly, comm, xi in zip(polys, comms, opening_challenges):
        for i, v in enumerate(poly):
            combined[i] = (combined[i] + xi * v) % p
        combined_comm = ec_add(curve, combined_comm, ec_mul(curve, xi, comm))
    combined_v = poly_eval(fr, combined, point)
    rc = random_oracle_challenge(fr, ser_point(curve, combined_comm) + ser_field(fr, point) + ser_field(fr, combined_v))
    h_prime = ec_mul(curve, rc, h)
    chal = []

    class _FS:
        def __getitem__(self, rnd):
            return chal[rnd]
    # replay ipa_rounds with the transcript-derived challenges
    zs = [pow(point, i, p) for i in range(n)]
    key, cs = list(comm_key), list(combined)
    l_vec, r_vec = [], []
    m = n
    while m > 1:
        hh = m // 2
        ip_l = sum(a * b for a, b in zip(cs[hh:m], zs[:hh])) % p
        ip_r = sum(a * b for a, b in zip(cs[:hh], zs[hh:m])) % p
        l = ec_add(curve, msm(curve, key[:hh], cs[hh:m]), ec_mul(curve, ip_l, h_prime))
        r = ec_add(curve, msm(curve, key[hh:m], cs[:hh]), ec_mul(curve, ip_r, h_prime))
        l_vec.append(l)
        r_vec.append(r)
        rc = random_oracle_challenge(fr, ser_field(fr, rc) + ser_point(curve, l) + ser_point(curve, r))
        chal.append(rc)
        ui = pow(rc, -1, p)
        for i in range(hh):
            cs[i] = (cs[i] + ui * cs[hh + i]) % p
            zs[i] = (zs[i] + rc * zs[hh + i]) % p
            key[i] = ec_add(curve, key[i], ec_mul(curve, rc, key[hh + i]))
        m = hh
    return l_vec, r_vec, key[0], cs[0], chal


def column_digest(field, column_canonical, hash_name):
    """FieldToBytesColHasher<F, D>::evaluate (bench-templates/src/lib.rs:327-337): D over
    to_bytes!(column) = u64 LE length || 32-byte LE canonical residues."""
    import hashlib
    h = hashlib.new(hash_name)
    h.update(len(column_canonical).to_bytes(8, "little"))
    for v in column_canonical:
        h.update(int(v).to_bytes(32, "little"))
    return h.digest()


def merkle_tree(leaf_digests, hash_name="sha256", len_prefix=True):
    """create_merkle_tree (poly-commit/src/linear_codes/mod.rs:506-521) for the Config of
    linear_codes/univariate_ligero/tests.rs:21-37 (identity leaf hash, byte-digest two-to-one
    hash, ByteDigestConverter).  The arithmetic lives in ark-crypto-primitives 0.5 (absent from
    /root/reference; restated from its published behaviour): leaves padded with empty byte strings
    to a power of two >= 2; bottom level D(conv(l) || conv(r)) with conv = ark-serialize of the
    Vec<u8> digest (u64 LE length || bytes) when len_prefix, raw bytes otherwise; upper levels
    D(left || right).  Returns the inner nodes in heap order (root first)."""
    import hashlib
    n = len(leaf_digests)
    assert n >= 1
    h = max(1, (n - 1).bit_length())
    leaves = list(leaf_digests) + [b""] * ((1 << h) - n)
    conv = (lambda b: len(b).to_bytes(8, "little") + b) if len_prefix else (lambda b: b)
    level = [hashlib.new(hash_name, conv(leaves[2 * i]) + conv(leaves[2 * i + 1])).digest() for i in range(1 << (h - 1))]
    levels = [level]
    while len(level) > 1:
        level = [hashlib.new(hash_name, level[2 * i] + level[2 * i + 1]).digest() for i in range(len(level) // 2)]
        levels.append(level)
    nodes = []
    for lv in reversed(levels):
        nodes.extend(lv)
    return nodes


def merkle_path(nodes, leaf_digests, index):
    """Authentication path of leaf `index` read out of the heap-ordered inner nodes (what
    MerkleTree::generate_proof returns, linear_codes/mod.rs:555-557): the sibling leaf digest,
    then the sibling inner node at every level from the bottom up to just below the root."""
    n_inner = len(nodes)
    sib = index ^ 1
    leaf_sibling = leaf_digests[sib] if sib < len(leaf_digests) else b""
    node = (n_inner + index + 1) // 2 - 1      # parent of the leaf: leaves continue the heap numbering at n_inner
    path = []
    while node > 0:
        path.append(nodes[node + 1 if node % 2 == 1 else node - 1])
        node = (node - 1) // 2
    return leaf_sibling, path


def merkle_verify(root, leaf, index, leaf_sibling, path, hash_name="sha256", len_prefix=True):
    import hashlib
    conv = (lambda b: len(b).to_bytes(8, "little") + b) if len_prefix else (lambda b: b)
    l, r = (leaf, leaf_sibling) if index % 2 == 0 else (leaf_sibling, leaf)
    cur = hashlib.new(hash_name, conv(l) + conv(r)).digest()
    index //= 2
    for s in path:
        cur = hashlib.new(hash_name, (cur + s) if index % 2 == 0 else (s + cur)).digest()
        index //= 2
    return cur == root


def fr_lincomb(field, polys, xi):
    """p = sum_j xi_j p_j: MarlinKZG10::open, marlin/marlin_pc/mod.rs:281-287."""
    p = FIELDS[field]["p"]
    n = max((len(q) for q in polys), default=0)
    out = [0] * n
    for c, q in zip(xi, polys):
        for i, v in enumerate(q):
            out[i] = (out[i] + c * v) % p
    return out


# ---------------------------------------------------------------------------------------
# InnerProductArgPC verifier (an independent line of evidence for the prover restated above: a proof made by
# `ipa_open` -- or by the device path -- must satisfy the reference's own check equations)
# ---------------------------------------------------------------------------------------
def succinct_check_coeffs(field, challenges):
    """SuccinctCheckPolynomial::compute_coeffs (ipa_pc/data_structures.rs:204-220)."""
    p = FIELDS[field]["p"]
    log_d = len(challenges)
    coeffs = [1] * (1 << log_d)
    for i, ch in enumerate(challenges, start=1):
        elem_degree = 1 << (log_d - i)
        for start in range(elem_degree, len(coeffs), 2 * elem_degree):
            for off in range(elem_degree):
                coeffs[start + off] = coeffs[start + off] * ch % p
    return coeffs


def succinct_check_eval(field, challenges, point):
    """SuccinctCheckPolynomial::evaluate (ipa_pc/data_structures.rs:223-236)."""
    p = FIELDS[field]["p"]
    log_d = len(challenges)
    prod = 1
    for i, ch in enumerate(challenges, start=1):
        elem = pow(point, 1 << (log_d - i), p)
        prod = prod * (1 + elem * ch) % p
    return prod


def ipa_succinct_check(curve, h, comms, point, values, proof, opening_challenges):
    """InnerProductArgPC::succinct_check without hiding and degree bounds (ipa_pc/mod.rs:91-203).
    proof = (l_vec, r_vec, final_comm_key, c).  Returns the round challenges, or None when the equation fails."""
    fr = CURVES[curve]["fr"]
    p = FIELDS[fr]["p"]
    l_vec, r_vec, final_key, c = proof
    combined_comm, combined_v = None, 0
    for comm, value, xi in zip(comms, values, opening_challenges):                    # :116-131
        combined_v = (combined_v + xi * value) % p
        combined_comm = ec_add(curve, combined_comm, ec_mul(curve, xi, comm))
    rc = random_oracle_challenge(fr, ser_point(curve, combined_comm) + ser_field(fr, point) + ser_field(fr, combined_v))
    h_prime = ec_mul(curve, rc, h)                                                    # :163
    round_comm = ec_add(curve, combined_comm, ec_mul(curve, combined_v, h_prime))     # :165
    chal = []
    for l, r in zip(l_vec, r_vec):                                                    # :170-184
        rc = random_oracle_challenge(fr, ser_field(fr, rc) + ser_point(curve, l) + ser_point(curve, r))
        chal.append(rc)
        round_comm = ec_add(curve, round_comm, ec_add(curve, ec_mul(curve, pow(rc, -1, p), l), ec_mul(curve, rc, r)))
    v_prime = succinct_check_eval(fr, chal, point) * c % p                            # :187
    check_elem = ec_add(curve, ec_mul(curve, c, final_key), ec_mul(curve, v_prime, h_prime))    # :190-195
    if round_comm != check_elem:
        return None
    return chal


def ipa_check(curve, comm_key, h, comms, point, values, proof, opening_challenges):
    """InnerProductArgPC::check (ipa_pc/mod.rs:725-773): the succinct check, then final_comm_key against the MSM of
    the key with the check polynomial's coefficients."""
    log_d = ark_log2(len(comm_key))
    if len(proof[0]) != len(proof[1]) or len(proof[0]) != log_d:
        raise ValueError("IncorrectInputLength")
    chal = ipa_succinct_check(curve, h, comms, point, values, proof, opening_challenges)
    if chal is None:
        return False
    fr = CURVES[curve]["fr"]
    return msm(curve, comm_key, succinct_check_coeffs(fr, chal)) == proof[2]


# ---------------------------------------------------------------------------------------
# HyraxPC (poly-commit/src/hyrax): sqrt(n) row commitments of sqrt(n) pairs -- the batched small-MSM shape
# (SURVEY.md 8f rank 4).  The sponge is the caller's: its challenge c and the prover's random field elements
# are inputs, in the order the reference draws them.
# ---------------------------------------------------------------------------------------
def flat_to_matrix_column_major(flat, n, m):
    """hyrax/utils.rs:13-21: row r = flat[r], flat[n + r], flat[2n + r], ..."""
    assert len(flat) == n * m
    return [[flat[col * n + row] for col in range(m)] for row in range(n)]


def tensor_prime(field, values):
    """hyrax/utils.rs:27-39: all evaluations of eq(i, values), first variable in the top bit."""
    p = FIELDS[field]["p"]
    if not values:
        return [1]
    tail = tensor_prime(field, values[1:])
    val = values[0]
    return [v * (1 - val) % p for v in tail] + [v * val % p for v in tail]


def hyrax_commit(curve, com_key, h, evals, rands):
    """HyraxPC::commit for one polynomial (hyrax/mod.rs:214-252): evals = poly.to_evaluations() (2^n values, n even),
    rands = the dim row randomisers.  Returns (row_coms, matrix rows)."""
    dim = 1 << ((len(evals).bit_length() - 1) // 2)
    assert dim * dim == len(evals) and dim <= len(com_key)
    m = flat_to_matrix_column_major(evals, dim, dim)
    return [ec_add(curve, msm(curve, com_key[:dim], row), ec_mul(curve, r, h)) for row, r in zip(m, rands)], m


def _hyrax_tensors(field, point):
    n = len(point)
    point_rev = list(reversed(point))                       # :297
    return tensor_prime(field, point_rev[n // 2:]), tensor_prime(field, point_rev[:n // 2])    # l (lower), r (upper)


def hyrax_open(curve, com_key, h, mat, rands, point, r_eval, d, r_d, r_b, c):
    """HyraxPC::open for one polynomial (hyrax/mod.rs:287-402).  r_eval, d, r_d, r_b: the random elements in the
    order the reference draws them (:352, :361-362, :367, :371); c: the sponge's challenge (:385).
    Returns (com_eval, com_d, com_b, z, z_d, z_b) and the evaluation."""
    fr = CURVES[curve]["fr"]
    p = FIELDS[fr]["p"]
    dim = len(mat)
    l, r = _hyrax_tensors(fr, point)
    lt = [sum(l[i] * mat[i][j] for i in range(dim)) % p for j in range(dim)]       # t.row_mul(&l)   :341
    r_lt = sum(a * b for a, b in zip(l, rands)) % p                                 # :345-348
    ev = sum(a * b for a, b in zip(lt, r)) % p                                      # :350
    com_eval = ec_add(curve, ec_mul(curve, ev, com_key[0]), ec_mul(curve, r_eval, h))
    b = sum(a * x for a, x in zip(r, d)) % p                                        # :364
    com_d = ec_add(curve, msm(curve, com_key[:dim], d), ec_mul(curve, r_d, h))      # :368
    com_b = ec_add(curve, ec_mul(curve, b, com_key[0]), ec_mul(curve, r_b, h))      # :372
    z = [(x + c * y) % p for x, y in zip(d, lt)]                                    # :387
    return (com_eval, com_d, com_b, z, (c * r_lt + r_d) % p, (c * r_eval + r_b) % p), ev


def hyrax_check(curve, com_key, h, row_coms, point, proof, c):
    """HyraxPC::check for one commitment (hyrax/mod.rs:418-511): equations (14) and (13) of the Hyrax paper."""
    fr = CURVES[curve]["fr"]
    p = FIELDS[fr]["p"]
    n = len(point)
    if n % 2 == 1:
        raise ValueError("InvalidNumberOfVariables")
    if len(row_coms) != 1 << (n // 2):
        raise ValueError("IncorrectCommitmentSize")
    com_eval, com_d, com_b, z, z_d, z_b = proof
    l, r = _hyrax_tensors(fr, point)
    ip = sum(a * b for a, b in zip(r, z)) % p
    com_dp = ec_add(curve, ec_mul(curve, ip, com_key[0]), ec_mul(curve, z_b, h))                # :486
    if com_dp != ec_add(curve, ec_mul(curve, c, com_eval), com_b):
        return False
    t_prime = msm(curve, row_coms, l)                                                           # :495
    com_z_zd = ec_add(curve, msm(curve, com_key[:len(z)], z), ec_mul(curve, z_d, h))            # :498
    return com_z_zd == ec_add(curve, ec_mul(curve, c, t_prime), com_d)


def mle_evaluate(field, evals, point):
    """DenseMultilinearExtension::evaluate (ark-poly 0.5, restated): variable 0 is the least significant index bit."""
    p = FIELDS[field]["p"]
    cur = list(evals)
    for x in point:
        cur = [(cur[2 * i] + x * (cur[2 * i + 1] - cur[2 * i])) % p for i in range(len(cur) // 2)]
    return cur[0]


# ---------------------------------------------------------------------------------------
# InnerProductArgPC with hiding and degree bounds (the general form of commit / open / succinct_check above)
# ---------------------------------------------------------------------------------------
def ipa_commit_general(curve, comm_key, s, coeffs, degree_bound=None, rand=0, shifted_rand=0):
    """InnerProductArgPC::commit for one polynomial (ipa_pc/mod.rs:403-473): comm = MSM(comm_key[..deg+1], coeffs) + s*rand;
    with a degree bound also shifted_comm = MSM(comm_key[supported_degree - bound ..], coeffs) + s*shifted_rand.
    Returns (comm, shifted_comm or None)."""
    d = len(comm_key) - 1
    comm = ec_add(curve, msm(curve, comm_key[:len(coeffs)], coeffs), ec_mul(curve, rand, s))
    shifted = None
    if degree_bound is not None:
        shifted = ec_add(curve, msm(curve, comm_key[d - degree_bound:d - degree_bound + len(coeffs)], coeffs), ec_mul(curve, shifted_rand, s))
    return comm, shifted


def _ipa_challenge_stream(challenges):
    it = iter(challenges)
    return lambda: next(it)


def ipa_open_general(curve, comm_key, h, s, polys, point, challenges, hiding_poly=None, hiding_rand=0):
    """InnerProductArgPC::open (ipa_pc/mod.rs:475-723) with hiding and degree bounds.
    polys: list of dicts {coeffs, comm, shifted_comm, degree_bound (None or int), hiding (bool), rand, shifted_rand};
    challenges: the values the caller's sponge squeezes, in order (:502, then :525 and :556 per polynomial);
    hiding_poly / hiding_rand: what the reference draws from the rng at :577 / :579 when any polynomial hides
    (d + 1 coefficients, before the subtraction of its value at the point).
    Returns (l_vec, r_vec, final_comm_key, c, hiding_comm or None, rand or None, round_challenges)."""
    fr = CURVES[curve]["fr"]
    p = FIELDS[fr]["p"]
    n = len(comm_key)
    d = n - 1
    squeeze = _ipa_challenge_stream(challenges)
    combined = [0] * n
    combined_rand, combined_comm, has_hiding = 0, None, False
    cur = squeeze()
    for q in polys:
        co = q["coeffs"]
        if len(co) - 1 > d:
            raise ValueError("TooManyCoefficients")
        db = q.get("degree_bound")
        if db is not None and (db < len(co) - 1 or db > d):
            raise ValueError("IncorrectDegreeBound")
        for i, v in enumerate(co):
            combined[i] = (combined[i] + cur * v) % p
        combined_comm = ec_add(curve, combined_comm, ec_mul(curve, cur, q["comm"]))
        if q.get("hiding"):
            has_hiding = True
            combined_rand = (combined_rand + cur * q["rand"]) % p
        cur = squeeze()
        if db is not None:
            shift = d - db                                             # shift_polynomial, :230-239
            for i, v in enumerate(co):
                combined[shift + i] = (combined[shift + i] + cur * v) % p
            combined_comm = ec_add(curve, combined_comm, ec_mul(curve, cur, q["shifted_comm"]))
            if q.get("hiding"):
                combined_rand = (combined_rand + cur * q["shifted_rand"]) % p
        cur = squeeze()
    combined_v = poly_eval(fr, combined, point)
    hiding_comm = None
    if has_hiding:
        hp = list(hiding_poly) + [0] * (n - len(hiding_poly))
        hp[0] = (hp[0] - poly_eval(fr, hp, point)) % p                 # :578
        hiding_comm = ec_add(curve, msm(curve, comm_key, hp), ec_mul(curve, hiding_rand, s))       # :580-585
        hc = random_oracle_challenge(fr, ser_point(curve, combined_comm) + ser_field(fr, point) + ser_field(fr, combined_v)
                                     + ser_point(curve, hiding_comm))  # :592-603
        combined = [(a + hc * b) % p for a, b in zip(combined, hp)]
        combined_rand = (combined_rand + hc * hiding_rand) % p
        combined_comm = ec_add(curve, combined_comm, ec_add(curve, ec_mul(curve, hc, hiding_comm),
                                                            ec_neg(curve, ec_mul(curve, combined_rand, s))))      # :606-607
    rc = random_oracle_challenge(fr, ser_point(curve, combined_comm) + ser_field(fr, point) + ser_field(fr, combined_v))
    h_prime = ec_mul(curve, rc, h)
    zs = [pow(point, i, p) for i in range(n)]
    key, cs = list(comm_key), list(combined)
    l_vec, r_vec, chal = [], [], []
    m = n
    while m > 1:
        hh = m // 2
        ip_l = sum(a * b for a, b in zip(cs[hh:m], zs[:hh])) % p
        ip_r = sum(a * b for a, b in zip(cs[:hh], zs[hh:m])) % p
        l = ec_add(curve, msm(curve, key[:hh], cs[hh:m]), ec_mul(curve, ip_l, h_prime))
        r = ec_add(curve, msm(curve, key[hh:m], cs[:hh]), ec_mul(curve, ip_r, h_prime))
        l_vec.append(l)
        r_vec.append(r)
        rc = random_oracle_challenge(fr, ser_field(fr, rc) + ser_point(curve, l) + ser_point(curve, r))
        chal.append(rc)
        ui = pow(rc, -1, p)
        for i in range(hh):
            cs[i] = (cs[i] + ui * cs[hh + i]) % p
            zs[i] = (zs[i] + rc * zs[hh + i]) % p
            key[i] = ec_add(curve, key[i], ec_mul(curve, rc, key[hh + i]))
        m = hh
    return l_vec, r_vec, key[0], cs[0], hiding_comm, (combined_rand if has_hiding else None), chal


def ipa_check_general(curve, comm_key, h, s, comms, point, values, proof, challenges):
    """InnerProductArgPC::check with hiding and degree bounds (ipa_pc/mod.rs:725-773 over succinct_check :91-203).
    comms: list of dicts {comm, shifted_comm, degree_bound}; proof = (l_vec, r_vec, final_comm_key, c, hiding_comm, rand)."""
    fr = CURVES[curve]["fr"]
    p = FIELDS[fr]["p"]
    d = len(comm_key) - 1
    log_d = ark_log2(d + 1)
    l_vec, r_vec, final_key, c, hiding_comm, rand = proof
    if len(l_vec) != len(r_vec) or len(l_vec) != log_d:
        raise ValueError("IncorrectInputLength")
    squeeze = _ipa_challenge_stream(challenges)
    combined_comm, combined_v = None, 0
    cur = squeeze()
    for cm, value in zip(comms, values):
        combined_v = (combined_v + cur * value) % p
        combined_comm = ec_add(curve, combined_comm, ec_mul(curve, cur, cm["comm"]))
        cur = squeeze()
        db = cm.get("degree_bound")
        assert (db is not None) == (cm.get("shifted_comm") is not None)
        if db is not None:
            shift = pow(point, d - db, p)                                                      # :128
            combined_v = (combined_v + cur * value % p * shift) % p
            combined_comm = ec_add(curve, combined_comm, ec_mul(curve, cur, cm["shifted_comm"]))
        cur = squeeze()
    assert (hiding_comm is not None) == (rand is not None)
    if hiding_comm is not None:                                                                # :138-151
        hc = random_oracle_challenge(fr, ser_point(curve, combined_comm) + ser_field(fr, point) + ser_field(fr, combined_v)
                                     + ser_point(curve, hiding_comm))
        combined_comm = ec_add(curve, combined_comm, ec_add(curve, ec_mul(curve, hc, hiding_comm), ec_neg(curve, ec_mul(curve, rand, s))))
    rc = random_oracle_challenge(fr, ser_point(curve, combined_comm) + ser_field(fr, point) + ser_field(fr, combined_v))
    h_prime = ec_mul(curve, rc, h)
    round_comm = ec_add(curve, combined_comm, ec_mul(curve, combined_v, h_prime))
    chal = []
    for l, r in zip(l_vec, r_vec):
        rc = random_oracle_challenge(fr, ser_field(fr, rc) + ser_point(curve, l) + ser_point(curve, r))
        chal.append(rc)
        round_comm = ec_add(curve, round_comm, ec_add(curve, ec_mul(curve, pow(rc, -1, p), l), ec_mul(curve, rc, r)))
    v_prime = succinct_check_eval(fr, chal, point) * c % p
    if round_comm != ec_add(curve, ec_mul(curve, c, final_key), ec_mul(curve, v_prime, h_prime)):
        return False
    return msm(curve, comm_key, succinct_check_coeffs(fr, chal)) == final_key


# ---------------------------------------------------------------------------------------
# LinearCodePCS (univariate Ligero) commit / open / check (linear_codes/mod.rs:228-505, univariate_ligero/mod.rs:68-86).
# The sponge is the caller's: the query indices and (with check_well_formedness) the vector r are inputs.
# ---------------------------------------------------------------------------------------
def ligero_commit(field, coeffs, rho_inv=4, sec_param=128, col_hash="blake2s", tree_hash="sha256"):
    """commit for one polynomial (:248-277): matrix, encoded matrix, column digests, Merkle tree."""
    n_rows, n_cols, _ = ligero_dimensions(field, len(coeffs), rho_inv, sec_param)
    flat = list(coeffs) + [0] * (n_rows * n_cols - len(coeffs))
    mat = [flat[r * n_cols:(r + 1) * n_cols] for r in range(n_rows)]
    ext = [reed_solomon(field, row, rho_inv) for row in mat]
    n_ext = len(ext[0])
    leaves = [column_digest(field, [ext[r][j] for r in range(n_rows)], col_hash) for j in range(n_ext)]
    nodes = merkle_tree(leaves, tree_hash)
    return dict(n_rows=n_rows, n_cols=n_cols, n_ext_cols=n_ext, root=nodes[0], mat=mat, ext=ext, leaves=leaves, nodes=nodes)


def ligero_tensor(field, z, left, right):
    """UnivariateLigero::tensor (univariate_ligero/mod.rs:70-86): ((1, z, .., z^(left-1)), (1, z^left, z^(2 left), ..))."""
    p = FIELDS[field]["p"]
    a, pw = [], 1
    for _ in range(left):
        a.append(pw)
        pw = pw * z % p
    b, q = [], 1
    for _ in range(right):
        b.append(q)
        q = q * pw % p
    return a, b


def ligero_num_queries(field, n_ext_cols, rho_inv=4, sec_param=128):
    return calculate_t(FIELDS[field]["p"].bit_length(), sec_param, (rho_inv - 1, rho_inv), n_ext_cols)


def tensor_vec(field, values):
    """linear_codes/utils.rs:240-258: eq-tensor of the values, value i in bit i of the index."""
    p = FIELDS[field]["p"]
    layer = [1]
    for v in values:
        layer = [x * (1 - v) % p for x in layer] + [x * v % p for x in layer]
    return layer


def ligero_multilinear_tensor(field, point, left_len):
    """MultilinearLigero::tensor (multilinear_ligero/mod.rs:70-84): the point is split at log2(left_len)."""
    split = ark_log2(left_len)
    return tensor_vec(field, point[:split]), tensor_vec(field, point[split:])


def ligero_open(field, st, z, indices, r=None, tensors=None):
    """open for one polynomial (:300-373 with generate_proof :523-565): v = b.M, the queried columns of the encoded
    matrix with their Merkle paths, and r.M when well-formedness is checked.  tensors = (a, b) overrides the univariate
    L::tensor(z, ..) (the multilinear scheme passes ligero_multilinear_tensor)."""
    _, b = tensors if tensors is not None else ligero_tensor(field, z, st["n_cols"], st["n_rows"])
    wf = fr_lincomb(field, st["mat"], r) if r is not None else None
    v = fr_lincomb(field, st["mat"], b)
    columns = [[st["ext"][row][i] for row in range(st["n_rows"])] for i in indices]
    paths = [(i,) + merkle_path(st["nodes"], st["leaves"], i) for i in indices]
    return dict(v=v, columns=columns, paths=paths, well_formedness=wf)


def ligero_check(field, commitment, z, value, proof, indices, r=None, rho_inv=4, col_hash="blake2s", tree_hash="sha256", tensors=None):
    """check for one commitment (:375-503).  commitment: dict with n_rows, n_cols, n_ext_cols, root.
    Raises ValueError("InvalidCommitment") where the reference returns Err, returns False for a wrong value."""
    p = FIELDS[field]["p"]
    n_rows, n_cols, n_ext = commitment["n_rows"], commitment["n_cols"], commitment["n_ext_cols"]
    if (r is not None) != (proof["well_formedness"] is not None):
        raise ValueError("InvalidCommitment")
    # the reference indexes columns[j] / paths[j] for every query index (linear_codes/mod.rs:443-489): a short proof cannot pass
    t = len(indices)
    if (len(proof["columns"]) != t or len(proof["paths"]) != t or len(proof["v"]) != n_cols
            or any(len(c) != n_rows for c in proof["columns"]) or any(not 0 <= i < n_ext for i in indices)
            or (r is not None and (len(proof["well_formedness"]) != n_cols or len(r) != n_rows))):
        raise ValueError("InvalidCommitment")
    col_hashes = [column_digest(field, c, col_hash) for c in proof["columns"]]
    for leaf, q_j, (idx, sib, path) in zip(col_hashes, indices, proof["paths"]):
        if idx != q_j or not merkle_verify(commitment["root"], leaf, idx, sib, path, tree_hash):
            raise ValueError("InvalidCommitment")
    w = reed_solomon(field, proof["v"], rho_inv)
    a, b = tensors if tensors is not None else ligero_tensor(field, z, n_cols, n_rows)
    wwf = reed_solomon(field, proof["well_formedness"], rho_inv) if r is not None else None
    for col, idx in zip(proof["columns"], indices):
        if r is not None and sum(x * y for x, y in zip(r, col)) % p != wwf[idx]:
            raise ValueError("InvalidCommitment")
        if sum(x * y for x, y in zip(b, col)) % p != w[idx]:
            raise ValueError("InvalidCommitment")
    return sum(x * y for x, y in zip(proof["v"], a)) % p == value % p
