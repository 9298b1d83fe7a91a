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
// Compressed points need a square root: y = (x^3 + b)^((p+1)/4) for p = 3 (mod 4) (BLS12-381, BN254); Tonelli-Shanks with
// the field's 2^s-th root of unity for Pallas (p = 1 mod 2^32): ~255 + up to s^2/2 squarings per point, once per key.
// SrsEncodeBody is the inverse (resident points -> ark-serialize bytes): what CanonicalSerialize writes for a Vec<G1Affine>.
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
  // square root in Fq (Montgomery in, Montgomery out); false: not a quadratic residue
  PC_HD static bool sqrt(const Fq& a, Fq& out) {
    constexpr int S = C::FqP::TWO_ADICITY;
    if constexpr (S == 1) {
      out = pow_p_plus_1_over_4(a);
      return out.sqr().eq(a);
    } else {
      // Tonelli-Shanks: p - 1 = 2^S t, t odd.  w = a^((t-1)/2), x = a w = a^((t+1)/2), b = x w = a^t; z = a primitive 2^S-th root.
      if (a.is_zero()) { out = Fq::zero(); return true; }
      uint32_t e[FN];            // (p - 1) >> (S + 1) = (t - 1) / 2
      { uint64_t br = 1; for (int i = 0; i < FN; i++) { const uint64_t t = (uint64_t)C::FqP::MOD[i] - br; e[i] = (uint32_t)t; br = t >> 63; } }
      for (int sh = 0; sh < S + 1; sh++) for (int i = 0; i < FN; i++) e[i] = (e[i] >> 1) | (i + 1 < FN ? e[i + 1] << 31 : 0u);
      Fq w = Fq::one();
      for (int i = FN * 32 - 1; i >= 0; i--) { w = w.sqr(); if ((e[i >> 5] >> (i & 31)) & 1) w = w.mul(a); }
      Fq x = a.mul(w), b = x.mul(w), z;
      for (int i = 0; i < FN; i++) z.l[i] = C::FqP::ROOT[i];
      const Fq one = Fq::one();
      int m = S;
      while (!b.eq(one)) {
        int k = 0; Fq t2 = b;
        while (k < m && !t2.eq(one)) { t2 = t2.sqr(); k++; }       // least k with b^(2^k) = 1
        if (k >= m) return false;                                  // b has order 2^m: a is not a square
        Fq wz = z;
        for (int j = 0; j < m - k - 1; j++) wz = wz.sqr();
        x = x.mul(wz); z = wz.sqr(); b = b.mul(z); m = k;
      }
      out = x;
      return true;
    }
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
        Fq y;
        ok = sqrt(rhs, y) && ok;
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

// Resident affine points -> ark-serialize bytes (the image of SrsDecodeBody): `CanonicalSerialize` of each G1Affine of a
// Vec (kzg10/data_structures.rs:57-77 writes `powers_of_g` this way; an IPA key, ipa_pc/data_structures.rs:17-36).
template <class C>
struct SrsEncodeBody {
  typedef Fd<typename C::FqP> Fq;
  typedef SrsDecodeBody<C> D;
  static constexpr int FN = Fq::N, AW = 2 * FN, XB = D::XB, YB = D::YB;
  const uint32_t* pts; uint32_t n; uint32_t compressed; uint32_t zcash; uint8_t* out;
  PC_HD static void store_bytes(const Fq& c, uint8_t* p, uint32_t nbytes, bool big_endian) {      // canonical limbs -> bytes
    for (uint32_t i = 0; i < nbytes; i++) {
      const uint8_t b = i < 4u * FN ? (uint8_t)(c.l[i >> 2] >> (8 * (i & 3))) : 0;
      p[big_endian ? nbytes - 1 - i : i] = b;
    }
  }
  PC_HD void operator()(uint32_t i) const {
    const uint32_t pbytes = D::point_bytes(compressed != 0, zcash != 0);
    uint8_t* p = out + (size_t)i * pbytes;
    const AffD<C> a = AffD<C>::load(pts + (size_t)i * AW);
    for (uint32_t k = 0; k < pbytes; k++) p[k] = 0;
    if (a.is_inf()) {                                     // x = y = 0 with the infinity flag
      if (zcash) p[0] = compressed ? 0xc0 : 0x40; else p[pbytes - 1] = 0x40;
      return;
    }
    const Fq xc = a.x.from_mont(), yc = a.y.from_mont(), ync = a.y.neg().from_mont();
    const bool y_is_larger = D::less_than(ync, yc);       // y > -y: SWFlags::YIsNegative / the zcash sort flag
    if (zcash) {
      store_bytes(xc, p, XB, true);
      if (compressed) p[0] |= 0x80 | (y_is_larger ? 0x20 : 0); else store_bytes(yc, p + XB, XB, true);
    } else if (compressed) {
      store_bytes(xc, p, YB, false);
      if (y_is_larger) p[YB - 1] |= 0x80;
    } else {
      store_bytes(xc, p, XB, false);
      store_bytes(yc, p + XB, YB, false);
      if (y_is_larger) p[XB + YB - 1] |= 0x80;
    }
  }
};

}  // namespace pc
