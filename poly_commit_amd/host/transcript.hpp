// Fiat-Shamir transcript of InnerProductArgPC::open, host side (poly-commit/src/ipa_pc/mod.rs):
//
//   compute_random_oracle_challenge        ipa_pc/mod.rs:74-87    D::digest(bytes || i.to_le_bytes()) until
//                                                                 Field::from_random_bytes accepts the digest
//   bytes hashed before the halving loop   :615-625               ser(combined_commitment) || ser(point) || ser(combined_v)
//   bytes hashed per round                 :681-688               ser(round_challenge) || ser(L) || ser(R)
//
// D = Blake2s (the reference's instantiation: ipa_pc/mod.rs:1056-1064, benches/ipa_times.rs:16); `ser` is
// ark-serialize's serialize_uncompressed.  Those byte conventions live in ark-serialize / ark-ff / ark-ec /
// ark-bls12-381 0.5, which are NOT under /root/reference: they are restated here from the crates' published
// behaviour and must be re-confirmed against the crates the first time a cargo registry is reachable
// (INTEGRATION.md lists them as assertions of the shim):
//   field element     canonical residue, little-endian, ceil(MODULUS_BIT_SIZE / 8) bytes
//   SW affine point   generic short_weierstrass impl (BN254 G1, Pallas): x, then y in ceil((bits + 2) / 8) bytes with
//                     SWFlags in the top bits of the LAST byte -- 0x80 YIsNegative (y > -y: y is the larger root;
//                     SWFlags::from_y_coordinate gives YIsPositive = no bit for y <= -y), 0x40 PointAtInfinity (x = y = 0)
//                     [round 2 had the 0x80 condition inverted; the BN254 generator (1, 2) must end in 0x00, its negation in 0x80]
//   BLS12-381 G1      zcash / IETF encoding: x, y big-endian 48 bytes each; byte 0: 0x80 compressed (clear), 0x40 infinity
//   from_random_bytes the first 8 N bytes little-endian, bits above MODULUS_BIT_SIZE cleared, None if >= modulus
// This is a handful of hashes of < 200 bytes per round: host work, exactly where the reference does it.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "kzg10.hpp"

namespace pc_host {

// SHA-256, FIPS 180-4 (byte-oriented; the verifier's path hashing of LinearCodePCS::check -- a few hundred digests --
// stays on the host, the device code in csrc/hash.hpp does whole trees).
struct Sha256Host {
  static void digest(const uint8_t* msg, size_t len, uint8_t out[32]) {
    static const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
      0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
      0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
      0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
      0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
      0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    auto block = [&](const uint8_t* b) {
      uint32_t w[64];
      for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
      for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
      }
      uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
      for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
      }
      h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    };
    size_t off = 0;
    for (; off + 64 <= len; off += 64) block(msg + off);
    uint8_t tail[128]; memset(tail, 0, sizeof tail);
    const size_t rem = len - off;
    if (rem) memcpy(tail, msg + off, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    block(tail); if (tl == 128) block(tail + 64);
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
  }
};

// BLAKE2s-256, RFC 7693 (unkeyed, byte-oriented; the device code in csrc/hash.hpp streams whole words only).
struct Blake2s {
  static void digest(const uint8_t* msg, size_t len, uint8_t out[32]) {
    static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010020u;
    size_t off = 0;
    uint8_t block[64];
    while (len - off > 64) { memcpy(block, msg + off, 64); off += 64; compress(h, block, off, false); }
    memset(block, 0, 64);
    if (len > off) memcpy(block, msg + off, len - off);
    compress(h, block, len, true);
    for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (8 * k));
  }

 private:
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  static void compress(uint32_t h[8], const uint8_t b[64], uint64_t t, bool last) {
    static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
    // message schedule: sigma_r = the r-th permutation of RFC 7693 section 2.7, generated from sigma_0 by the
    // fixed permutation the specification's table is built from would be obscure -- the table is spelled out
    static const uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    static const uint8_t Q[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15}, {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};
    for (int r = 0; r < 10; r++)
      for (int q = 0; q < 8; q++) {
        uint32_t &a = v[Q[q][0]], &bb = v[Q[q][1]], &c = v[Q[q][2]], &d = v[Q[q][3]];
        a += bb + m[S[r][2 * q]]; d = rotr(d ^ a, 16); c += d; bb = rotr(bb ^ c, 12);
        a += bb + m[S[r][2 * q + 1]]; d = rotr(d ^ a, 8); c += d; bb = rotr(bb ^ c, 7);
      }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
  }
};

template <class E>
struct Transcript {
  typedef FrT<E> Fr;
  typedef pc::host64::F64<typename E::C::FqP> Fq;
  typedef typename E::C::FqP FqP;
  typedef typename E::C::FrP FrP;
  std::vector<uint8_t> bytes;

  // canonical little-endian bytes of a Montgomery-form element
  template <class P>
  static void canonical_le(const uint64_t* mont, size_t nbytes, std::vector<uint8_t>& out) {
    typedef pc::host64::F64<P> F;
    F v; memcpy(v.l, mont, sizeof(v.l));
    F one = F::zero(); one.l[0] = 1;
    F c = v.mul(one);                                   // Montgomery -> canonical
    for (size_t i = 0; i < nbytes; i++) out.push_back(i < sizeof(c.l) ? (uint8_t)(c.l[i / 8] >> (8 * (i % 8))) : 0);
  }
  void append(const Fr& f) { canonical_le<FrP>(f.l, (FrP::BITS + 7) / 8, bytes); }
  void append(const G1Affine<E>& p) {
    const size_t xb = (FqP::BITS + 7) / 8, yb = (FqP::BITS + 2 + 7) / 8;
    if (E::ID == PC_CURVE_BLS12_381) {                  // zcash encoding, big-endian
      const size_t at = bytes.size();
      if (p.infinity) { bytes.insert(bytes.end(), 2 * xb, 0); bytes[at] |= 0x40; return; }
      std::vector<uint8_t> le;
      canonical_le<FqP>(p.x, xb, le); bytes.insert(bytes.end(), le.rbegin(), le.rend());
      le.clear(); canonical_le<FqP>(p.y, xb, le); bytes.insert(bytes.end(), le.rbegin(), le.rend());
      return;
    }
    if (p.infinity) { bytes.insert(bytes.end(), xb + yb, 0); bytes.back() |= 0x40; return; }
    canonical_le<FqP>(p.x, xb, bytes);
    std::vector<uint8_t> y, ny;
    canonical_le<FqP>(p.y, yb, y);
    const G1Affine<E> n = p.neg();
    canonical_le<FqP>(n.y, yb, ny);
    bool y_gt_neg = false;
    for (size_t i = yb; i-- > 0;) if (y[i] != ny[i]) { y_gt_neg = y[i] > ny[i]; break; }
    if (y_gt_neg) y.back() |= 0x80;                     // SWFlags::YIsNegative <=> y > -y (from_y_coordinate: y <= -y is YIsPositive, no bit)
    bytes.insert(bytes.end(), y.begin(), y.end());
  }

  // Field::from_random_bytes for Fr
  static bool from_random_bytes(const uint8_t* b, size_t len, Fr& out) {
    uint64_t c[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < 32 && i < len; i++) c[i / 8] |= (uint64_t)b[i] << (8 * (i % 8));
    const int shave = 256 - FrP::BITS;
    if (shave > 0) c[3] &= ~(uint64_t)0 >> shave;
    typedef pc::host64::F64<FrP> F;
    for (int i = 3; i >= 0; i--) { if (c[i] < F::mod(i)) break; if (c[i] > F::mod(i) || i == 0) return false; }
    F v; memcpy(v.l, c, 32);
    F r2; memcpy(r2.l, FrP::R2, 32);
    F m = v.mul(r2);                                    // canonical -> Montgomery
    memcpy(out.l, m.l, 32);
    return true;
  }
  // compute_random_oracle_challenge(bytes), ipa_pc/mod.rs:74-87
  static Fr compute_random_oracle_challenge(const std::vector<uint8_t>& bytes) {
    std::vector<uint8_t> in(bytes);
    in.resize(bytes.size() + 8);
    for (uint64_t i = 0;; i++) {
      for (int k = 0; k < 8; k++) in[bytes.size() + k] = (uint8_t)(i >> (8 * k));
      uint8_t h[32]; Blake2s::digest(in.data(), in.size(), h);
      Fr out;
      if (from_random_bytes(h, 32, out)) return out;
    }
  }
  Fr challenge() const { return compute_random_oracle_challenge(bytes); }
};

}  // namespace pc_host
