// Column hashing for the linear-code commitments (Ligero): the step right after the row
// encoding in LinearCodePCS::commit (poly-commit/src/linear_codes/mod.rs:256-263),
//     leaves[j] = H::evaluate(col_hash_params, ext_mat.cols()[j]),
// for the byte-digest column hashers the reference's tests and benches use,
// FieldToBytesColHasher<F, D> (bench-templates/src/lib.rs:309-338, D = Blake2s256 / Sha256):
//     D::digest(to_bytes!(column))
// where to_bytes! is ark-serialize's compressed encoding of Vec<F>: the length as u64
// little-endian, then every element as 32 little-endian bytes of its CANONICAL residue.
// One lane per column: consecutive lanes read consecutive columns of the same row, so the
// row-major encoded matrix is streamed with perfectly coalesced 32-byte loads and never has to
// be transposed or leave HBM (2 GiB at 2^24 coefficients -> 4 MiB of digests).
// SHA-256 (FIPS 180-4) and BLAKE2s-256 (RFC 7693) are written as word-streaming states:
// every input item is a multiple of 4 bytes and 4-byte aligned in the message.
#pragma once
#include "fp32.hpp"
#include "hash_constants.h"

namespace pc {

PC_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
PC_HD uint32_t bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }

struct Sha256 {
  uint32_t h[8], buf[16], nbuf; uint64_t bytes;
  PC_HD void init() { PC_UNROLL for (int i = 0; i < 8; i++) h[i] = pc_hash_constants::IV[i]; nbuf = 0; bytes = 0; }
  // chaining state between whole blocks (nbuf == 0): h and the byte counter, 10 words
  PC_HD void export_state(uint32_t* o) const { PC_UNROLL for (int i = 0; i < 8; i++) o[i] = h[i]; o[8] = (uint32_t)bytes; o[9] = (uint32_t)(bytes >> 32); }
  PC_HD void import_state(const uint32_t* o) { PC_UNROLL for (int i = 0; i < 8; i++) h[i] = o[i]; bytes = (uint64_t)o[8] | ((uint64_t)o[9] << 32); nbuf = 0; }
  PC_HD void compress() {
    uint32_t w[16], s[8];
    PC_UNROLL for (int i = 0; i < 16; i++) w[i] = buf[i];
    PC_UNROLL for (int i = 0; i < 8; i++) s[i] = h[i];
    PC_UNROLL for (int t = 0; t < 64; t++) {
      if (t >= 16) {
        uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
        uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
        uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
        w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
      }
      uint32_t S1 = rotr32(s[4], 6) ^ rotr32(s[4], 11) ^ rotr32(s[4], 25);
      uint32_t ch = (s[4] & s[5]) ^ (~s[4] & s[6]);
      uint32_t t1 = s[7] + S1 + ch + pc_hash_constants::K[t] + w[t & 15];
      uint32_t S0 = rotr32(s[0], 2) ^ rotr32(s[0], 13) ^ rotr32(s[0], 22);
      uint32_t maj = (s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]);
      uint32_t t2 = S0 + maj;
      s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1; s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
    }
    PC_UNROLL for (int i = 0; i < 8; i++) h[i] += s[i];
    nbuf = 0;
  }
  // four message bytes, first byte in the low 8 bits
  PC_HD void push_le32(uint32_t w) { buf[nbuf++] = bswap32(w); bytes += 4; if (nbuf == 16) compress(); }
  // Whole 64-byte blocks / a tail of NT words with COMPILE-TIME buffer indices (buf[nbuf++] with a run-time nbuf
  // puts the buffer into scratch memory on the device: for the 2 GiB of config 5 that was 1.9 GB of private-segment
  // writes, rocprofv3 WRITE_SIZE).  Only valid while nbuf == 0.
  PC_HD void absorb_block(const uint32_t* m) {
    PC_UNROLL for (int k = 0; k < 16; k++) buf[k] = bswap32(m[k]);
    bytes += 64; compress();
  }
  template <int NT>
  PC_HD void finish_tail(const uint32_t* tail, uint32_t* out) {
    static_assert(NT <= 13, "tail + padding + length must fit one block");
    bytes += 4 * NT;
    const uint64_t bits = bytes * 8;
    PC_UNROLL for (int k = 0; k < 14; k++) buf[k] = k < NT ? bswap32(tail[k < NT ? k : 0]) : (k == NT ? 0x80000000u : 0u);
    buf[14] = (uint32_t)(bits >> 32); buf[15] = (uint32_t)bits;
    compress();
    PC_UNROLL for (int i = 0; i < 8; i++) out[i] = bswap32(h[i]);
  }
  // digest as 8 words whose little-endian memory image is the 32 digest bytes
  PC_HD void finish(uint32_t* out) {
    const uint64_t bits = bytes * 8;
    buf[nbuf++] = 0x80000000u; if (nbuf == 16) compress();
    while (nbuf != 14) { buf[nbuf++] = 0; if (nbuf == 16) compress(); }
    buf[14] = (uint32_t)(bits >> 32); buf[15] = (uint32_t)bits; nbuf = 16; compress();
    PC_UNROLL for (int i = 0; i < 8; i++) out[i] = bswap32(h[i]);
  }
};

struct Blake2s256 {
  uint32_t h[8], buf[16], nbuf; uint64_t t;
  PC_HD void init() {
    PC_UNROLL for (int i = 0; i < 8; i++) h[i] = pc_hash_constants::IV[i];
    h[0] ^= 0x01010020u;   // digest length 32, no key, fanout 1, depth 1
    nbuf = 0; t = 0;
  }
  PC_HD void export_state(uint32_t* o) const { PC_UNROLL for (int i = 0; i < 8; i++) o[i] = h[i]; o[8] = (uint32_t)t; o[9] = (uint32_t)(t >> 32); }
  PC_HD void import_state(const uint32_t* o) { PC_UNROLL for (int i = 0; i < 8; i++) h[i] = o[i]; t = (uint64_t)o[8] | ((uint64_t)o[9] << 32); nbuf = 0; }
  PC_HD void compress(bool last) {
    uint32_t v[16], m[16];
    PC_UNROLL for (int i = 0; i < 16; i++) m[i] = buf[i];
    PC_UNROLL for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = pc_hash_constants::IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define PC_B2S_G(a, b, c, d, x, y)                                                   \
    v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); v[c] = v[c] + v[d];  \
    v[b] = rotr32(v[b] ^ v[c], 12); v[a] = v[a] + v[b] + (y);                       \
    v[d] = rotr32(v[d] ^ v[a], 8); v[c] = v[c] + v[d]; v[b] = rotr32(v[b] ^ v[c], 7);
    PC_UNROLL for (int r = 0; r < 10; r++) {
      const uint8_t* sg = pc_hash_constants::SIGMA[r];
      PC_B2S_G(0, 4, 8, 12, m[sg[0]], m[sg[1]])
      PC_B2S_G(1, 5, 9, 13, m[sg[2]], m[sg[3]])
      PC_B2S_G(2, 6, 10, 14, m[sg[4]], m[sg[5]])
      PC_B2S_G(3, 7, 11, 15, m[sg[6]], m[sg[7]])
      PC_B2S_G(0, 5, 10, 15, m[sg[8]], m[sg[9]])
      PC_B2S_G(1, 6, 11, 12, m[sg[10]], m[sg[11]])
      PC_B2S_G(2, 7, 8, 13, m[sg[12]], m[sg[13]])
      PC_B2S_G(3, 4, 9, 14, m[sg[14]], m[sg[15]])
    }
#undef PC_B2S_G
    PC_UNROLL for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  }
  PC_HD void push_le32(uint32_t w) {
    if (nbuf == 16) { t += 64; compress(false); nbuf = 0; }   // a full buffer is only compressed once more input arrives
    buf[nbuf++] = w;
  }
  // whole blocks that are NOT the last one / the last NT (>= 1) words, compile-time indices (see Sha256); nbuf == 0
  PC_HD void absorb_block(const uint32_t* m) {
    PC_UNROLL for (int k = 0; k < 16; k++) buf[k] = m[k];
    t += 64; compress(false);
  }
  template <int NT>
  PC_HD void finish_tail(const uint32_t* tail, uint32_t* out) {
    static_assert(NT >= 1 && NT <= 16, "the last block holds 1..16 words");
    t += 4 * NT;
    PC_UNROLL for (int k = 0; k < 16; k++) buf[k] = k < NT ? tail[k < NT ? k : 0] : 0u;
    compress(true);
    PC_UNROLL for (int i = 0; i < 8; i++) out[i] = h[i];
  }
  PC_HD void finish(uint32_t* out) {
    t += 4ull * nbuf;
    while (nbuf < 16) buf[nbuf++] = 0;
    compress(true);
    PC_UNROLL for (int i = 0; i < 8; i++) out[i] = h[i];
  }
};

enum : uint32_t { PC_HASH_SHA256_ID = 0, PC_HASH_BLAKE2S_ID = 1 };

// digest[j] = D(len_u64_le || canonical_le(ext[0][j]) || ... || canonical_le(ext[rows-1][j]))
template <class FrP, class D>
struct ColumnHashBody {
  typedef Fd<FrP> F;
  const uint32_t* ext;   // rows x n_cols elements, row-major, Montgomery
  uint32_t rows, n_cols;
  uint32_t* out;         // n_cols x 8 words
  PC_HD F row(uint32_t r, uint32_t j) const { return F::load(ext + ((size_t)r * n_cols + j) * FrP::N); }
  PC_HD void operator()(uint32_t j) const {
    static_assert(FrP::N == 8, "32-byte scalar-field elements");
    // Message words: [rows, 0] then 8 words per row, i.e. block b = 2 words carried over | row 2b | the first 6 words
    // of row 2b+1; its last 2 words open block b+1.  Two rows per iteration keep every buffer index a constant.
    D d; d.init();
    uint32_t carry[2] = {rows, 0u};                          // Vec length as u64 LE
    // the loads of the next pair are issued before this pair is converted and hashed
    // (one wave per 64 columns leaves 2 waves per SIMD: without this the kernel waits on every load)
    F a = rows > 0 ? row(0, j) : F::zero(), b = rows > 1 ? row(1, j) : F::zero();
    uint32_t r = 0;
    for (; r + 2 <= rows; r += 2) {
      const F ca = a.from_mont(), cb = b.from_mont();
      if (r + 2 < rows) a = row(r + 2, j);
      if (r + 3 < rows) b = row(r + 3, j);
      uint32_t m[16];
      m[0] = carry[0]; m[1] = carry[1];
      PC_UNROLL for (int k = 0; k < 8; k++) m[2 + k] = ca.l[k];
      PC_UNROLL for (int k = 0; k < 6; k++) m[10 + k] = cb.l[k];
      carry[0] = cb.l[6]; carry[1] = cb.l[7];
      d.absorb_block(m);
    }
    uint32_t dig[8];
    if (r < rows) {                                          // odd number of rows: 2 carried words + the last row
      const F ca = a.from_mont();
      uint32_t tail[10];
      tail[0] = carry[0]; tail[1] = carry[1];
      PC_UNROLL for (int k = 0; k < 8; k++) tail[2 + k] = ca.l[k];
      d.template finish_tail<10>(tail, dig);
    } else d.template finish_tail<2>(carry, dig);
    PC_UNROLL for (int k = 0; k < 8; k++) out[(size_t)j * 8 + k] = dig[k];
  }
};

// The same digests when the ROWS of the encoded matrix are spread over several devices (rows are independent for
// the NTT, but a column's digest needs all of its rows).  Instead of transposing 2 GiB between the devices, the digest's CHAINING
// STATE travels: device d absorbs its slab of rows [r0, r0 + rows) into the state device d - 1 left for every column, and hands
// on 48 bytes per column (h, byte counter, the 8 message bytes that straddle the slab edge -- the u64 length prefix shifts every
// 32-byte element by 8 against the 64-byte blocks).  state[j] = 12 words: export_state (10) | carry (2).  `first`: start from
// the IV and the length prefix of the WHOLE column (rows_total); `last`: finish and write the digest.  Slabs of devices that are
// not the last hold an even number of rows (two rows fill one block).
template <class FrP, class D>
struct ColumnHashPartBody {
  typedef Fd<FrP> F;
  const uint32_t* ext;   // rows x n_cols elements of this slab, row-major, Montgomery
  uint32_t rows, n_cols, rows_total, col0, first, last;
  uint32_t* state;       // n_cols x 12 words, read unless `first`, written unless `last`
  uint32_t* out;         // n_cols x 8 words, written when `last`
  PC_HD F row(uint32_t r, uint32_t j) const { return F::load(ext + ((size_t)r * n_cols + j) * FrP::N); }
  PC_HD void operator()(uint32_t lane) const {
    static_assert(FrP::N == 8, "32-byte scalar-field elements");
    const uint32_t j = col0 + lane;
    D d; uint32_t carry[2];
    if (first) { d.init(); carry[0] = rows_total; carry[1] = 0u; }
    else { uint32_t st[12]; PC_UNROLL for (int k = 0; k < 12; k++) st[k] = state[(size_t)j * 12 + k]; d.import_state(st); carry[0] = st[10]; carry[1] = st[11]; }
    F a = rows > 0 ? row(0, j) : F::zero(), b = rows > 1 ? row(1, j) : F::zero();
    uint32_t r = 0;
    for (; r + 2 <= rows; r += 2) {
      const F ca = a.from_mont(), cb = b.from_mont();
      if (r + 2 < rows) a = row(r + 2, j);
      if (r + 3 < rows) b = row(r + 3, j);
      uint32_t m[16];
      m[0] = carry[0]; m[1] = carry[1];
      PC_UNROLL for (int k = 0; k < 8; k++) m[2 + k] = ca.l[k];
      PC_UNROLL for (int k = 0; k < 6; k++) m[10 + k] = cb.l[k];
      carry[0] = cb.l[6]; carry[1] = cb.l[7];
      d.absorb_block(m);
    }
    if (last) {
      uint32_t dig[8];
      if (r < rows) {
        const F ca = a.from_mont();
        uint32_t tail[10];
        tail[0] = carry[0]; tail[1] = carry[1];
        PC_UNROLL for (int k = 0; k < 8; k++) tail[2 + k] = ca.l[k];
        d.template finish_tail<10>(tail, dig);
      } else d.template finish_tail<2>(carry, dig);
      PC_UNROLL for (int k = 0; k < 8; k++) out[(size_t)j * 8 + k] = dig[k];
    } else {                                                    // (rows is even here: checked by the caller)
      uint32_t st[12]; d.export_state(st); st[10] = carry[0]; st[11] = carry[1];
      PC_UNROLL for (int k = 0; k < 12; k++) state[(size_t)j * 12 + k] = st[k];
    }
  }
};

// Merkle tree over the column digests: step 2b of LinearCodePCS::commit,
// create_merkle_tree (poly-commit/src/linear_codes/mod.rs:506-521) ->
// ark_crypto_primitives::merkle_tree::MerkleTree::new with the Config the reference's tests and
// benches use (linear_codes/univariate_ligero/tests.rs:21-37): LeafHash = LeafIdentityHasher
// (the 32-byte column digest IS the leaf digest), TwoToOneHash = a byte digest D,
// LeafInnerDigestConverter = ByteDigestConverter.  The leaf list is padded with empty leaves to a
// power of two (mod.rs:517-518).  One level per launch, one lane per parent:
//   bottom level: parent = D( conv(leaf 2i) || conv(leaf 2i+1) ), conv = the leaf digest either
//                 raw or ark-serialized as Vec<u8> (u64 LE length || bytes) -- `len_prefix`;
//                 a padding leaf is the empty byte string;
//   upper levels: parent = D( left 32 bytes || right 32 bytes ).
// Nodes are written in heap order (root at 0, children of i at 2i+1 / 2i+2), the layout of
// MerkleTree::non_leaf_nodes, so the caller reads authentication paths straight out of it.
template <class D>
struct MerkleLevelBody {
  const uint32_t* child;   // bottom: n_real leaf digests x 8 words; upper: the child level inside `nodes`
  uint32_t* parent;        // this level inside `nodes`
  uint32_t n_real;         // bottom level only: leaves that exist (the rest are padding)
  uint32_t bottom, len_prefix;
  PC_HD void push_leaf(D& d, uint32_t k) const {
    const bool real = k < n_real;
    if (len_prefix) { d.push_le32(real ? 32u : 0u); d.push_le32(0); }
    if (real) { PC_UNROLL for (int w = 0; w < 8; w++) d.push_le32(child[(size_t)k * 8 + w]); }
  }
  PC_HD void operator()(uint32_t i) const {
    D d; d.init();
    if (bottom) { push_leaf(d, 2 * i); push_leaf(d, 2 * i + 1); }
    else { PC_UNROLL for (int w = 0; w < 16; w++) d.push_le32(child[(size_t)i * 16 + w]); }
    uint32_t dig[8]; d.finish(dig);
    PC_UNROLL for (int w = 0; w < 8; w++) parent[(size_t)i * 8 + w] = dig[w];
  }
};

}  // namespace pc
