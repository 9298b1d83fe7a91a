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

// The same decomposition in 32-bit limbs, host AND device (the MSM's GLV table mode splits every scalar in its digit passes):
// k (8 limbs, canonical) -> sign-magnitude halves m[0] = |k1|, m[1] = |k2| (5 limbs each) with k = +-|k1| +- |k2| lambda (mod r).
// Bit for bit the values of glv_decompose above (CPU-stepped test against it and against Python big ints).
// Bound: the basis vectors are below 2^128 (glv_constants.h), Babai rounding with truncated quotients leaves |k_i| below
// (|a_1| + |a_2|) / 2 plus two basis vectors: under 2^130 for the three curves (GLV_HALF_BITS), so MSM_HALF windows of the signed
// radix-2^c recoding cover it (msm_num_windows(GLV_HALF_BITS, c)).
static constexpr uint32_t GLV_HALF_BITS = 130;
template <class C>
struct GlvHalves {
  typedef typename GlvOf<C>::T G;
  uint32_t m[2][5]; uint32_t neg[2];
  static PC_HD uint32_t limb(const uint64_t* a, int i) { return (uint32_t)(a[i >> 1] >> ((i & 1) * 32)); }
  // out[NA + NB] = a * b over 32-bit limbs (a given as 64-bit words)
  template <int NA, int NB, class FA, class FB>
  static PC_HD void mul32(FA fa, FB fb, uint32_t* out) {
    PC_UNROLL for (int i = 0; i < NA + NB; i++) out[i] = 0;
    PC_UNROLL for (int i = 0; i < NA; i++) {
      uint64_t c = 0;
      const uint32_t ai = fa(i);
      PC_UNROLL for (int j = 0; j < NB; j++) { c += (uint64_t)ai * fb(j) + out[i + j]; out[i + j] = (uint32_t)c; c >>= 32; }
      out[i + NB] = (uint32_t)c;
    }
  }
  // acc (12 limbs, two's complement) += / -= t (12 limbs, magnitude)
  static PC_HD void addsub(uint32_t* acc, const uint32_t* t, bool subtract) {
    uint64_t c = subtract ? 1 : 0;
    PC_UNROLL for (int i = 0; i < 12; i++) { c += (uint64_t)acc[i] + (subtract ? ~t[i] : t[i]); acc[i] = (uint32_t)c; c >>= 32; }
  }
  PC_HD void split(const uint32_t* k) {
    uint32_t prod[18], c1[6], c2[6], t[12];
    mul32<10, 8>([](int i) { return limb(G::G1, i); }, [&](int j) { return k[j]; }, prod);
    PC_UNROLL for (int i = 0; i < 6; i++) c1[i] = prod[12 + i];                    // floor(g1 k / 2^384)
    mul32<10, 8>([](int i) { return limb(G::G2, i); }, [&](int j) { return k[j]; }, prod);
    PC_UNROLL for (int i = 0; i < 6; i++) c2[i] = prod[12 + i];
    const bool c1neg = G::N1_NEG != 0, c2neg = G::N2_NEG != 0;
    uint32_t k1[12], k2[12];
    PC_UNROLL for (int i = 0; i < 12; i++) { k1[i] = i < 8 ? k[i] : 0u; k2[i] = 0u; }
    mul32<6, 6>([&](int i) { return c1[i]; }, [](int j) { return limb(G::A1, j); }, t); addsub(k1, t, !(c1neg ^ (G::A1_NEG != 0)));
    mul32<6, 6>([&](int i) { return c2[i]; }, [](int j) { return limb(G::A2, j); }, t); addsub(k1, t, !(c2neg ^ (G::A2_NEG != 0)));
    mul32<6, 6>([&](int i) { return c1[i]; }, [](int j) { return limb(G::B1, j); }, t); addsub(k2, t, !(c1neg ^ (G::B1_NEG != 0)));
    mul32<6, 6>([&](int i) { return c2[i]; }, [](int j) { return limb(G::B2, j); }, t); addsub(k2, t, !(c2neg ^ (G::B2_NEG != 0)));
    fin(k1, 0); fin(k2, 1);
  }
  PC_HD void fin(uint32_t* v, int h) {
    const uint32_t ng = v[11] >> 31;
    if (ng) { uint64_t c = 1; PC_UNROLL for (int i = 0; i < 12; i++) { c += (uint64_t)(~v[i]); v[i] = (uint32_t)c; c >>= 32; } }
    neg[h] = ng;
    PC_UNROLL for (int i = 0; i < 5; i++) m[h][i] = v[i];
  }
};

// key[i] = affine(key[i] + k * key[half + i]),  k = (+-k1) + (+-k2) * lambda
template <class C>
struct EcFoldGlvBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int AW = 2 * Fq::N;
  uint32_t* key; uint32_t half;
  const uint32_t* key_in = nullptr;   // not null: read the 2 * half points from here (out-of-place fold into `key`)
  NafMasks<5> n1, n2;      // NAF of |k1|, |k2|
  uint32_t neg1, neg2;
  uint32_t beta[Fq::N];
  uint32_t* jac_out = nullptr;   // not null: store (X, Y, Z) of lane i here instead of normalising in the lane
                                 // (JacBatchAffineBody then writes key[i] with one inversion per K points)
  PC_HD void operator()(uint32_t i) const {
    const uint32_t* src = key_in ? key_in : key;
    AffD<C> kl = AffD<C>::load(src + (size_t)i * AW), kr = AffD<C>::load(src + (size_t)(half + i) * AW);
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

// ---- the FIRST key fold of an opening from a table of the committer key ---------------------------------------------
// `k_l += k_r * u` (ipa_pc/mod.rs:699-701) multiplies every element of the upper half of the key by the round challenge.  The
// committer key is the same for every opening, so the doublings of that multiplication can be done once per key:
//   T[b][j] = 2^b * K[half + j],  b < FOLD_ROWS,  j < half                    (FOLD_ROWS x the upper half of the key in HBM)
// and u * K[half + j] = sum over the non-zero NAF digits of the GLV halves of u:  +-T[b][j]  (k1)  /  +-phi(T[b][j])  (k2),
// phi(x, y) = (beta x, y): ~86 mixed additions and no doubling instead of ~130 doublings + ~86 additions per element (the
// first fold is half of all fold work of an opening).  The rows are read as coalesced streams (lane i reads T[b][i]).
static constexpr uint32_t FOLD_ROWS = 131;       // |k1|, |k2| <= 2^128 (glv_constants.h): NAF digits at bits 0 .. 129, one spare

// row b of the table from row b - 1: affine doubling of n points with one inversion per K points (Montgomery's trick along a
// lane's run; lambda = 3 x^2 / (2 y))
template <class C>
struct AffineDoubleRowBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int FN = Fq::N, AW = 2 * FN;
  const uint32_t* in; uint32_t* out; uint32_t* scratch;   // scratch: n x Fq (prefix products)
  uint32_t n, K;
  PC_HD void operator()(uint32_t t) const {
    const uint32_t s = t * K, e = (n - s > K) ? s + K : n;
    Fq run = Fq::one();
    for (uint32_t j = s; j < e; j++) {
      run.store(scratch + (size_t)j * FN);
      const AffD<C> p = AffD<C>::load(in + (size_t)j * AW);
      if (!p.is_inf() && !p.y.is_zero()) run = run.mul(p.y.dbl());
    }
    Fq inv = run.inv();
    for (uint32_t j = e; j-- > s;) {
      const AffD<C> p = AffD<C>::load(in + (size_t)j * AW);
      AffD<C> r = AffD<C>::infinity();
      if (!p.is_inf() && !p.y.is_zero()) {
        const Fq d = p.y.dbl();
        const Fq di = inv.mul(Fq::load(scratch + (size_t)j * FN));       // 1 / (2 y)
        inv = inv.mul(d);
        const Fq xx = p.x.sqr(), lam = xx.dbl().add(xx).mul(di);
        r.x = lam.sqr().sub(p.x.dbl());
        r.y = lam.mul(p.x.sub(r.x)).sub(p.y);
      }
      r.store(out + (size_t)j * AW);
    }
  }
};

// row(d + 2, 0) of the fold table from row(d, 0) and row(1, 1): out[j] = a[j] + b[j], affine, one inversion per K points.
// The operands are d P and 2 P of the same point P (d odd, >= 1): in a prime-order group their x coordinates differ unless P is the
// point at infinity, so only that case is handled besides the general chord.
template <class C>
struct AffineAddRowBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int FN = Fq::N, AW = 2 * FN;
  const uint32_t* a; const uint32_t* b; uint32_t* out; uint32_t* scratch;   // scratch: n x Fq (prefix products)
  uint32_t n, K;
  PC_HD void operator()(uint32_t t) const {
    const uint32_t s = t * K, e = (n - s > K) ? s + K : n;
    Fq run = Fq::one();
    for (uint32_t j = s; j < e; j++) {
      run.store(scratch + (size_t)j * FN);
      const AffD<C> p = AffD<C>::load(a + (size_t)j * AW), q = AffD<C>::load(b + (size_t)j * AW);
      const Fq d = q.x.sub(p.x);
      if (!p.is_inf() && !q.is_inf() && !d.is_zero()) run = run.mul(d);
    }
    Fq inv = run.inv();
    for (uint32_t j = e; j-- > s;) {
      const AffD<C> p = AffD<C>::load(a + (size_t)j * AW), q = AffD<C>::load(b + (size_t)j * AW);
      AffD<C> r = p.is_inf() ? q : p;                 // one operand at infinity: the other one
      const Fq d = q.x.sub(p.x);
      if (!p.is_inf() && !q.is_inf()) {
        if (d.is_zero()) r = AffD<C>::infinity();     // (d P = -2 P: not reachable for points of prime order; d P = 2 P neither)
        else {
          const Fq di = inv.mul(Fq::load(scratch + (size_t)j * FN));       // 1 / (x2 - x1)
          inv = inv.mul(d);
          const Fq lam = q.y.sub(p.y).mul(di);
          r.x = lam.sqr().sub(p.x).sub(q.x);
          r.y = lam.mul(p.x.sub(r.x)).sub(p.y);
        }
      }
      r.store(out + (size_t)j * AW);
    }
  }
};

// out_xyzz[i] = K[i] + sum of the listed table entries.  The table holds, for the `row_pts` key points it covers, FOLD_ROWS rows per
// odd multiple d = 1, 3, .., 2^(w-1) - 1:  T[(d >> 1) * FOLD_ROWS + b][j] = d * 2^b * P_j.  An op names one entry for lane i:
//   bits 0..10 row (8 x 131 rows at width 5), bits 11..12 term (the entry is at point term * count + i of its row), bit 14: phi of the entry, bit 15: negate.
// One level (the first fold of an opening, ipa_pc/mod.rs:699-701): one term, the upper half of the key, scalar u.  Two levels: the
// key after TWO folds straight from the committer key,
//   K''[i] = K[i] + u2 K[q + i] + u1 K[2q + i] + (u1 u2) K[3q + i],  q = n / 4,
// three terms over the table of K[q .. 4q) -- 3 x (2 x 130 / (w + 1)) mixed additions per element and no doubling, where the first
// fold from a one-level table plus a ladder for the second cost 86 additions per element of the half and 127 doublings + 94
// additions per element of the quarter.
template <class C>
struct EcFoldTableBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int AW = 2 * Fq::N;
  static constexpr uint32_t MAX_OPS = 448;
  const uint32_t* key_lo;    // K[0 .. count)
  const uint32_t* table;     // rows of row_pts affine points
  uint32_t count, row_pts, n_ops;
  uint16_t ops[MAX_OPS];
  uint32_t beta[Fq::N];
  uint32_t* out_xyzz;        // count x XyzzD::WORDS
  PC_HD void operator()(uint32_t i) const {
    XyzzD<C> acc = XyzzD<C>::from_affine(AffD<C>::load(key_lo + (size_t)i * AW));
    const Fq b = Fq::load(beta);
    for (uint32_t k = 0; k < n_ops; k++) {
      const uint32_t op = ops[k], row = op & 0x7ffu, term = (op >> 11) & 3u;
      AffD<C> p = AffD<C>::load(table + ((size_t)row * row_pts + (size_t)term * count + i) * AW);
      if (p.is_inf()) continue;
      if (op & 0x4000u) p.x = p.x.mul(b);
      acc.add_affine(p.neg_if((op & 0x8000u) != 0));
    }
    acc.store(out_xyzz + (size_t)i * XyzzD<C>::WORDS);
  }
};

// width-w NAF of a 160-bit magnitude (host): digits odd, |digit| < 2^(w-1), at most one non-zero digit in any w consecutive
// positions (w = 2: the plain NAF).  out[bit] in (-2^(w-1), 2^(w-1)); returns the number of positions written.
inline int wnaf_digits(const uint32_t k[5], int w, int8_t out[200]) {
  uint32_t t[6] = {k[0], k[1], k[2], k[3], k[4], 0};
  const uint32_t mask = (1u << w) - 1, half = 1u << (w - 1);
  int len = 0;
  for (int bit = 0; bit < 200; bit++) {
    out[bit] = 0;
    bool any = false; for (int i = 0; i < 6; i++) any |= t[i] != 0;
    if (!any) continue;
    if (t[0] & 1) {
      const uint32_t m = t[0] & mask;
      if (m >= half) {                       // digit m - 2^w < 0: t += 2^w - m
        out[bit] = (int8_t)((int)m - (int)(mask + 1));
        uint64_t c = (uint64_t)(mask + 1 - m); for (int i = 0; i < 6; i++) { c += t[i]; t[i] = (uint32_t)c; c >>= 32; }
      } else { out[bit] = (int8_t)m; t[0] -= m; }
      len = bit + 1;
    }
    for (int i = 0; i < 5; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
    t[5] >>= 1;
  }
  return len;
}

}  // namespace pc
