// Host-side finish of an MSM: the Horner fold of the <= W*levels window/level sums and the
// single inversion to affine.  This is a ~255-doubling dependency chain; one CPU core
// (64-bit limbs, unsigned __int128) runs it in ~0.15 ms where one GPU lane would need
// milliseconds.  Input/outputs are the same little-endian Montgomery bytes the device uses.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "field_constants.h"

namespace pc {
namespace host64 {

typedef unsigned __int128 u128;

template <class P>   // P = one of the 32-bit-limb constant structs, N even
struct F64 {
  static constexpr int N = P::N / 2;
  uint64_t l[N];
  static constexpr uint64_t mod(int i) { return (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32); }
  static constexpr uint64_t inv64() {   // -p^-1 mod 2^64 by Newton iteration
    uint64_t p0 = mod(0), x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - p0 * x;
    return (uint64_t)0 - x;
  }
  static F64 zero() { F64 r; memset(r.l, 0, sizeof(r.l)); return r; }
  static F64 one() { F64 r; memcpy(r.l, P::ONE, sizeof(r.l)); return r; }
  static F64 load(const uint32_t* p) { F64 r; memcpy(r.l, p, sizeof(r.l)); return r; }
  void store(uint32_t* p) const { memcpy(p, l, sizeof(l)); }
  bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  static bool geq(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) { if (a[i] > mod(i)) return true; if (a[i] < mod(i)) return false; }
    return true;
  }
  static void subm(uint64_t* a) {
    u128 br = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - mod(i) - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
  }
  F64 add(const F64& o) const {
    F64 r; u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (geq(r.l)) subm(r.l);
    return r;
  }
  F64 sub(const F64& o) const {
    F64 r; u128 br = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)l[i] - o.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)r.l[i] + mod(i); r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
  }
  F64 dbl() const { return add(*this); }
  F64 mul(const F64& o) const {
    constexpr uint64_t INV = inv64();
    uint64_t t[N + 2];
    for (int i = 0; i < N + 2; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
      u128 c = 0;
      for (int j = 0; j < N; j++) { c += (u128)l[j] * o.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * INV;
      c = (u128)m * mod(0) + t[0]; c >>= 64;
      for (int j = 1; j < N; j++) { c += (u128)m * mod(j) + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);
    }
    F64 r; for (int i = 0; i < N; i++) r.l[i] = t[i];
    if (t[N] || geq(r.l)) subm(r.l);
    return r;
  }
  F64 sqr() const { return mul(*this); }
  F64 inv() const {   // a^(p-2)
    uint64_t e[N]; u128 br = 2;
    for (int i = 0; i < N; i++) { u128 d = (u128)mod(i) - (uint64_t)br; e[i] = (uint64_t)d; br = (d >> 64) & 1; }
    F64 r = one();
    for (int i = N * 64 - 1; i >= 0; i--) { r = r.sqr(); if ((e[i / 64] >> (i % 64)) & 1) r = r.mul(*this); }
    return r;
  }
};

template <class C>
struct Xyzz64 {
  typedef F64<typename C::FqP> Fq;
  static constexpr int FW = C::FqP::N;   // 32-bit words per coordinate
  Fq X, Y, ZZ, ZZZ;
  static Xyzz64 infinity() { Xyzz64 r; r.X = r.Y = r.ZZ = r.ZZZ = Fq::zero(); return r; }
  bool is_inf() const { return ZZ.is_zero(); }
  void store(uint32_t* p) const { X.store(p); Y.store(p + FW); ZZ.store(p + 2 * FW); ZZZ.store(p + 3 * FW); }
  static Xyzz64 load(const uint32_t* p) { Xyzz64 r; r.X = Fq::load(p); r.Y = Fq::load(p + FW); r.ZZ = Fq::load(p + 2 * FW); r.ZZZ = Fq::load(p + 3 * FW); return r; }
  Xyzz64 dbl() const {   // dbl-2008-s-1, a = 0
    if (is_inf() || Y.is_zero()) return infinity();
    Xyzz64 r;
    Fq U = Y.dbl(), V = U.sqr(), W = U.mul(V), S = X.mul(V), xx = X.sqr(), M = xx.dbl().add(xx);
    r.X = M.sqr().sub(S.dbl());
    r.Y = M.mul(S.sub(r.X)).sub(W.mul(Y));
    r.ZZ = V.mul(ZZ); r.ZZZ = W.mul(ZZZ);
    return r;
  }
  void add(const Xyzz64& o) {   // add-2008-s
    if (o.is_inf()) return;
    if (is_inf()) { *this = o; return; }
    Fq U1 = X.mul(o.ZZ), U2 = o.X.mul(ZZ), S1 = Y.mul(o.ZZZ), S2 = o.Y.mul(ZZZ);
    Fq Pp = U2.sub(U1), R = S2.sub(S1);
    if (Pp.is_zero()) { if (R.is_zero()) *this = dbl(); else *this = infinity(); return; }
    Fq PP = Pp.sqr(), PPP = Pp.mul(PP), Q = U1.mul(PP);
    Fq X3 = R.sqr().sub(PPP).sub(Q.dbl());
    Y = R.mul(Q.sub(X3)).sub(S1.mul(PPP));
    X = X3;
    ZZ = ZZ.mul(o.ZZ).mul(PP); ZZZ = ZZZ.mul(o.ZZZ).mul(PPP);
  }
  // affine x||y (Montgomery); (0,0) for infinity
  void store_affine(uint32_t* out) const {
    if (is_inf()) { memset(out, 0, 2 * FW * 4); return; }
    Fq t = ZZ.mul(ZZZ).inv();
    X.mul(t.mul(ZZZ)).store(out); Y.mul(t.mul(ZZ)).store(out + FW);
  }
};

struct WeightedPoint { uint32_t exponent; const uint32_t* xyzz; };

// out = sum_i 2^exponent_i * P_i as an affine point: one descending Horner chain.
template <class C>
void horner_to_affine(std::vector<WeightedPoint>& items, uint32_t* out_affine) {
  std::stable_sort(items.begin(), items.end(), [](const WeightedPoint& a, const WeightedPoint& b) { return a.exponent > b.exponent; });
  Xyzz64<C> acc = Xyzz64<C>::infinity();
  uint32_t cur = items.empty() ? 0 : items[0].exponent;
  for (const auto& it : items) {
    for (; cur > it.exponent; cur--) acc = acc.dbl();
    acc.add(Xyzz64<C>::load(it.xyzz));
  }
  for (; cur > 0; cur--) acc = acc.dbl();
  acc.store_affine(out_affine);
}

// count XYZZ points -> count affine points (x||y Montgomery, (0,0) = infinity) with ONE field
// inversion (Montgomery's trick over the products ZZ*ZZZ).
template <class C>
void batch_to_affine(const uint32_t* xyzz, size_t count, uint32_t* out_affine) {
  typedef Xyzz64<C> P; typedef typename P::Fq Fq;
  constexpr int FW = P::FW;
  std::vector<Fq> prefix(count);
  Fq run = Fq::one();
  for (size_t i = 0; i < count; i++) {
    P p = P::load(xyzz + i * 4 * FW);
    prefix[i] = run;
    if (!p.is_inf()) run = run.mul(p.ZZ.mul(p.ZZZ));
  }
  Fq inv = run.inv();
  for (size_t i = count; i-- > 0;) {
    P p = P::load(xyzz + i * 4 * FW);
    uint32_t* o = out_affine + i * 2 * FW;
    if (p.is_inf()) { memset(o, 0, 2 * FW * 4); continue; }
    Fq t = inv.mul(prefix[i]);                 // 1 / (ZZ * ZZZ) of point i
    inv = inv.mul(p.ZZ.mul(p.ZZZ));
    p.X.mul(t.mul(p.ZZZ)).store(o); p.Y.mul(t.mul(p.ZZ)).store(o + FW);
  }
}

}  // namespace host64
}  // namespace pc
