"""BLAKE2s-256 with an explicit, exportable chaining state (RFC 7693) -- TEST INFRASTRUCTURE for the oracle-backed engine of
tests/_sharded_worker.py: hashlib's objects cannot be moved between processes, and the chained column digests of
ShardedRows.commit hand exactly that state (h, byte counter, pending bytes) from rank to rank."""
IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
         [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
         [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
         [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
         [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]
M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def compress(h, block, t, last):
    m = [int.from_bytes(block[4 * i:4 * i + 4], "little") for i in range(16)]
    v = list(h) + list(IV)
    v[12] ^= t & M32
    v[13] ^= (t >> 32) & M32
    if last:
        v[14] ^= M32

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M32; v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M32; v[b] = _rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M32; v[d] = _rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M32; v[b] = _rotr(v[b] ^ v[c], 7)
    for r in range(10):
        s = SIGMA[r]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]])
        g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]])
        g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def init():
    h = list(IV)
    h[0] ^= 0x01010020
    return h, 0


def absorb(h, t, pending, data, last):
    """(h, t) + pending bytes + data -> new (h, t, pending) or, when `last`, the 32-byte digest.  All blocks but the final one
    are compressed as soon as MORE input is known to follow; the final block (1..64 bytes, zero padded) carries the flag."""
    buf = pending + data
    while len(buf) > 64:
        t += 64
        h = compress(h, buf[:64], t, False)
        buf = buf[64:]
    if not last:
        return h, t, buf
    t += len(buf)
    h = compress(h, buf + bytes(64 - len(buf)), t, True)
    return b"".join(x.to_bytes(4, "little") for x in h)
