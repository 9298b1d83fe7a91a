// GLV endomorphism for the j = 0 curves (y^2 = x^3 + b): phi(x, y) = (beta x, y) = [lambda](x, y).
// A full-width scalar k is split on the host into k = k1 + k2*lambda (mod r) with |k1|, |k2| < 2^130,
// and k*P = k1*P + k2*phi(P) is evaluated with one shared ladder of ~130 doublings (Shamir's
// trick, NAF digits) instead of 255.  Used by the IPA key fold (ipa_pc/mod.rs:699-707), where the
// scalar is the round challenge shared by every element.  Constants: glv_constants.h (generated).
#pragma once
#include <string.h>
#include "ec.hpp"
#include "glv_constants.h"

namespace pc {

template <class C> struct GlvOf;
template <> struct GlvOf<pc_curve_bls12_381> { typedef pc_glv_bls12_381 T; };
template <> struct GlvOf<pc_curve_bn254> { typedef pc_glv_bn254 T; };
template <> struct GlvOf<pc_curve_pallas> { typedef pc_glv_pallas T; };

struct GlvSplit { uint32_t k1[5], k2[5]; uint32_t neg1, neg2; };   // sign-magnitude, 160-bit magnitudes

namespace glv_host {
typedef unsigned __int128 u128;
// out[na+nb] = a * b
inline void mul(const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out) {
  for (int i = 0; i < na + nb; i++) out[i] = 0;
  for (int i = 0; i < na; i++) {
    u128 c = 0;
    for (int j = 0; j < nb; j++) { c += (u128)a[i] * b[j] + out[i + j]; out[i + j] = (uint64_t)c; c >>= 64; }
    out[i + nb] = (uint64_t)c;
  }
}
// acc (6 limbs, two's complement) += sign * t (6 limbs, magnitude)
inline void addsub(uint64_t* acc, const uint64_t* t, bool subtract) {
  u128 c = subtract ? 1 : 0;
  for (int i = 0; i < 6; i++) { c += (u128)acc[i] + (subtract ? ~t[i] : t[i]); acc[i] = (uint64_t)c; c >>= 64; }
}
}  // namespace glv_host

// k: canonical scalar, 4 x u64.  Mirrors tools/gen_constants.py::glv_decompose_like_device.
template <class G>
inline GlvSplit glv_decompose(const uint64_t* k) {
  using namespace glv_host;
  uint64_t prod[9], c1[3], c2[3];
  mul(G::G1, 5, k, 4, prod); for (int i = 0; i < 3; i++) c1[i] = prod[6 + i];     // floor(g1 k / 2^384)
  mul(G::G2, 5, k, 4, prod); for (int i = 0; i < 3; i++) c2[i] = prod[6 + i];
  const bool c1neg = G::N1_NEG, c2neg = G::N2_NEG;
  // k1 = k - c1*a1 - c2*a2 ;  k2 = -c1*b1 - c2*b2   (signed, in 384-bit two's complement)
  uint64_t k1[6] = {k[0], k[1], k[2], k[3], 0, 0}, k2[6] = {0, 0, 0, 0, 0, 0}, t[6];
  mul(c1, 3, G::A1, 3, t); addsub(k1, t, !(c1neg ^ (bool)G::A1_NEG));   // subtract when c1*a1 > 0
  mul(c2, 3, G::A2, 3, t); addsub(k1, t, !(c2neg ^ (bool)G::A2_NEG));
  mul(c1, 3, G::B1, 3, t); addsub(k2, t, !(c1neg ^ (bool)G::B1_NEG));
  mul(c2, 3, G::B2, 3, t); addsub(k2, t, !(c2neg ^ (bool)G::B2_NEG));
  GlvSplit s;
  auto fin = [](uint64_t* v, uint32_t* mag, uint32_t& neg) {
    neg = (uint32_t)(v[5] >> 63);
    if (neg) { u128 c = 1; for (int i = 0; i < 6; i++) { c += (u128)(~v[i]); v[i] = (uint64_t)c; c >>= 64; } }
    mag[0] = (uint32_t)v[0]; mag[1] = (uint32_t)(v[0] >> 32); mag[2] = (uint32_t)v[1]; mag[3] = (uint32_t)(v[1] >> 32); mag[4] = (uint32_t)v[2];
  };
  fin(k1, s.k1, s.neg1); fin(k2, s.k2, s.neg2);
  return s;
}

// key[i] = affine(key[i] + k * key[half + i]),  k = (+-k1) + (+-k2) * lambda
template <class C>
struct EcFoldGlvBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int AW = 2 * Fq::N;
  uint32_t* key; uint32_t half;
  NafMasks<5> n1, n2;      // NAF of |k1|, |k2|
  uint32_t neg1, neg2;
  uint32_t beta[Fq::N];
  uint32_t* jac_out = nullptr;   // not null: store (X, Y, Z) of lane i here instead of normalising in the lane
                                 // (JacBatchAffineBody then writes key[i] with one inversion per K points)
  PC_HD void operator()(uint32_t i) const {
    AffD<C> kl = AffD<C>::load(key + (size_t)i * AW), kr = AffD<C>::load(key + (size_t)(half + i) * AW);
    AffD<C> p1 = kr.neg_if(neg1 != 0);
    AffD<C> p2; p2.x = kr.x.mul(Fq::load(beta)); p2.y = kr.y; if (kr.is_inf()) p2 = kr;
    p2 = p2.neg_if(neg2 != 0);
    const AffD<C> m1 = p1.neg_if(true), m2 = p2.neg_if(true);
    JacD<C> acc = JacD<C>::infinity();
    for (int bit = 32 * 6 - 1; bit >= 0; bit--) {
      const uint32_t w = bit >> 5, m = 1u << (bit & 31);
      const uint32_t any = (n1.pos[w] | n1.neg[w] | n2.pos[w] | n2.neg[w]);
      if (!acc.is_inf()) acc = acc.dbl(); else if (!(any & m)) continue;
      if (n1.pos[w] & m) acc.add_affine(p1); else if (n1.neg[w] & m) acc.add_affine(m1);
      if (n2.pos[w] & m) acc.add_affine(p2); else if (n2.neg[w] & m) acc.add_affine(m2);
    }
    acc.add_affine(kl);
    if (jac_out) {
      uint32_t* o = jac_out + (size_t)i * 3 * Fq::N;
      if (acc.is_inf()) { Fq::zero().store(o); Fq::zero().store(o + Fq::N); Fq::zero().store(o + 2 * Fq::N); }
      else { acc.X.store(o); acc.Y.store(o + Fq::N); acc.Z.store(o + 2 * Fq::N); }
    } else acc.to_affine().store(key + (size_t)i * AW);
  }
};

}  // namespace pc
