// ark-serialize wire format of G1 points -> resident affine points (x || y, Montgomery, (0,0) = infinity).
//
// kzg10::UniversalParams serialises `powers_of_g: Vec<E::G1Affine>` first (poly-commit/src/kzg10/data_structures.rs:57-77:
// serialize_with_mode of the Vec = u64 little-endian length, then every point), and an IPA key is a `Vec<G>`
// (ipa_pc/data_structures.rs:17-36): this decoder is what lets a prover load a ceremony file / a stored key straight
// into HBM.  The point encodings live in ark-ec / ark-serialize / ark-bls12-381 0.5 (not under /root/reference) and are
// restated from their published behaviour -- the same statement as host/transcript.hpp:
//   generic short Weierstrass (BN254 G1, Pallas)
//     uncompressed  x: ceil(bits/8) bytes LE, y: ceil((bits+2)/8) bytes LE, flags in the top bits of the LAST byte
//     compressed    x: ceil((bits+2)/8) bytes LE with the flags in the top bits of the last byte
//     flags         0x80 YIsNegative (y > -y: the larger of the two roots; no bit = YIsPositive = the smaller), 0x40 PointAtInfinity
//   BLS12-381 G1 (zcash / IETF)
//     uncompressed  x, y: 48 bytes big-endian each; byte 0: 0x80 clear, 0x40 infinity
//     compressed    x: 48 bytes big-endian; byte 0: 0x80 set, 0x40 infinity, 0x20 y is the lexicographically larger root
// Compressed points need a square root: y = (x^3 + b)^((p+1)/4), available for p = 3 (mod 4) (BLS12-381, BN254).
#pragma once
#include "ec.hpp"
#include "msm.hpp"

namespace pc {

template <class C>
struct SrsDecodeBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int FN = Fq::N, AW = 2 * FN;
  static constexpr int XB = (C::FqP::BITS + 7) / 8, YB = (C::FqP::BITS + 2 + 7) / 8;
  const uint8_t* in; uint32_t n; uint32_t compressed; uint32_t zcash;   // zcash: BLS12-381's big-endian encoding
  uint32_t* out;           // n x AW
  uint32_t* bad;           // counter of points that are not on the curve / have no square root
  PC_HD static uint32_t point_bytes(bool compressed, bool zcash) { return zcash ? (compressed ? XB : 2 * XB) : (compressed ? YB : XB + YB); }
  // nbytes little-endian (or big-endian) bytes -> canonical limbs, the top `drop` bits of the most significant byte cleared
  PC_HD static Fq load_bytes(const uint8_t* p, uint32_t nbytes, bool big_endian, uint32_t mask_top) {
    Fq r = Fq::zero();
    for (uint32_t i = 0; i < nbytes && i < 4u * FN; i++) {
      uint32_t b = big_endian ? p[nbytes - 1 - i] : p[i];
      if (i == nbytes - 1) b &= mask_top;
      r.l[i >> 2] |= b << (8 * (i & 3));
    }
    return r;
  }
  PC_HD static bool less_than(const Fq& a, const Fq& b) {       // canonical residues
    for (int i = FN - 1; i >= 0; i--) { if (a.l[i] != b.l[i]) return a.l[i] < b.l[i]; }
    return false;
  }
  PC_HD static bool below_modulus(const Fq& a) {
    for (int i = FN - 1; i >= 0; i--) { if (a.l[i] != C::FqP::MOD[i]) return a.l[i] < C::FqP::MOD[i]; }
    return false;
  }
  PC_HD static Fq curve_rhs(const Fq& x) { Fq b; PC_UNROLL for (int i = 0; i < FN; i++) b.l[i] = C::B_MONT[i]; return x.sqr().mul(x).add(b); }
  PC_HD static Fq pow_p_plus_1_over_4(const Fq& a) {
    uint32_t e[FN];            // (p + 1) / 4
    uint64_t c = 1;
    for (int i = 0; i < FN; i++) { c += C::FqP::MOD[i]; e[i] = (uint32_t)c; c >>= 32; }
    for (int i = 0; i < FN; i++) e[i] = (e[i] >> 2) | (i + 1 < FN ? e[i + 1] << 30 : (uint32_t)c << 30);
    Fq r = Fq::one();
    for (int i = FN * 32 - 1; i >= 0; i--) { r = r.sqr(); if ((e[i >> 5] >> (i & 31)) & 1) r = r.mul(a); }
    return r;
  }
  PC_HD void operator()(uint32_t i) const {
    const uint32_t pbytes = point_bytes(compressed != 0, zcash != 0);
    const uint8_t* p = in + (size_t)i * pbytes;
    bool inf, want_larger = false; Fq xc, yc = Fq::zero();
    if (zcash) {
      inf = (p[0] & 0x40) != 0; want_larger = (p[0] & 0x20) != 0;
      xc = load_bytes(p, XB, true, 0x1f);
      if (!compressed) yc = load_bytes(p + XB, XB, true, 0xff);
    } else if (compressed) {
      inf = (p[YB - 1] & 0x40) != 0; want_larger = (p[YB - 1] & 0x80) != 0;     // YIsNegative (0x80) = the larger root, YIsPositive (no flag) = the smaller
      xc = load_bytes(p, YB, false, 0x3f);
    } else {
      inf = (p[XB + YB - 1] & 0x40) != 0;
      xc = load_bytes(p, XB, false, 0xff);
      yc = load_bytes(p + XB, YB, false, 0x3f);
    }
    AffD<C> a = AffD<C>::infinity();
    if (!inf) {
      bool ok = below_modulus(xc) && below_modulus(yc);
      a.x = xc.to_mont();
      const Fq rhs = curve_rhs(a.x);
      if (compressed) {
        Fq y = pow_p_plus_1_over_4(rhs);
        ok = ok && y.sqr().eq(rhs);
        const Fq yn = y.neg();
        const bool y_is_larger = less_than(yn.from_mont(), y.from_mont());
        a.y = (y_is_larger == want_larger) ? y : yn;
      } else {
        a.y = yc.to_mont();
        ok = ok && a.y.sqr().eq(rhs);
      }
      if (!ok) { atomic_inc_u32(bad); a = AffD<C>::infinity(); }
    }
    a.store(out + (size_t)i * AW);
  }
};

}  // namespace pc
