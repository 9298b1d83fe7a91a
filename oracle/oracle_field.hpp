// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, or called from, the product
// library (poly_commit_amd/csrc).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load the shared object built from this file.
//
// Prime-field and short-Weierstrass (a = 0) arithmetic with 64-bit limbs and
// unsigned __int128 products.  Deliberately a different implementation from the HIP side
// (32-bit limbs, XYZZ buckets): the two only share the moduli in oracle/pyref.py.
//
// What it restates: the arithmetic the reference obtains from crates.io ark-ff / ark-ec
// 0.5 (not vendored under /root/reference):
//   * Fp<MontBackend> : Montgomery residues, 64-bit LE limbs, R = 2^(64 N)
//   * short_weierstrass::{Affine, Projective}: Jacobian coordinates (X/Z^2, Y/Z^3)
// Parity status: "parity unpinned by reference constants" -- the reference holds no golden
// vectors for this path; the oracle is pinned against oracle/pyref.py (Python big ints),
// the curve-order check r*G = O and the recalled arkworks TWO_ADIC_ROOT_OF_UNITY limbs.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "oracle_constants.h"

typedef unsigned __int128 u128;

template <class P>
struct Fp {
  typedef P Params;
  static constexpr int N = P::N;
  uint64_t l[N];

  static Fp zero() { Fp r; memset(r.l, 0, sizeof(r.l)); return r; }
  static Fp one() { Fp r; for (int i = 0; i < N; i++) r.l[i] = P::ONE[i]; return r; }
  static Fp from_raw(const uint64_t* p) { Fp r; memcpy(r.l, p, sizeof(r.l)); return r; }
  void to_raw(uint64_t* p) const { memcpy(p, l, sizeof(l)); }

  bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  bool operator==(const Fp& o) const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i] ^ o.l[i]; return a == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }

  static inline bool geq_mod(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) {
      if (a[i] > P::MOD[i]) return true;
      if (a[i] < P::MOD[i]) return false;
    }
    return true;
  }
  static inline void sub_mod(uint64_t* a) {
    u128 br = 0;
    for (int i = 0; i < N; i++) {
      u128 d = (u128)a[i] - P::MOD[i] - (uint64_t)br;
      a[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }

  Fp operator+(const Fp& o) const {
    Fp r; u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    // all moduli here leave at least one spare bit in the top limb, so c == 0
    if (geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r; u128 br = 0;
    for (int i = 0; i < N; i++) {
      u128 d = (u128)l[i] - o.l[i] - (uint64_t)br;
      r.l[i] = (uint64_t)d; br = (d >> 64) & 1;
    }
    if (br) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)r.l[i] + P::MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
  }
  Fp neg() const { return is_zero() ? *this : zero() - *this; }
  Fp dbl() const { return *this + *this; }

  // CIOS Montgomery product.
  Fp operator*(const Fp& o) const {
    uint64_t t[N + 2];
    for (int i = 0; i < N + 2; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
      u128 c = 0;
      for (int j = 0; j < N; j++) { c += (u128)l[j] * o.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * P::INV;
      c = (u128)m * P::MOD[0] + t[0]; c >>= 64;
      for (int j = 1; j < N; j++) { c += (u128)m * P::MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);
    }
    Fp r; for (int i = 0; i < N; i++) r.l[i] = t[i];
    if (t[N] || geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  Fp sqr() const { return *this * *this; }

  // canonical (non-Montgomery) <-> Montgomery
  static Fp from_canonical(const uint64_t* c) { Fp a = from_raw(c); Fp r2 = from_raw(P::R2); return a * r2; }
  void to_canonical(uint64_t* out) const {
    Fp o; for (int i = 0; i < N; i++) o.l[i] = 0; o.l[0] = 1;   // raw 1 => multiplies by R^-1
    Fp r = *this * o; memcpy(out, r.l, sizeof(r.l));
  }
  static Fp from_u64(uint64_t v) { uint64_t c[N] = {0}; c[0] = v; return from_canonical(c); }

  Fp pow_limbs(const uint64_t* e, int n) const {
    Fp r = one();
    for (int i = n * 64 - 1; i >= 0; i--) { r = r.sqr(); if ((e[i / 64] >> (i % 64)) & 1) r = r * *this; }
    return r;
  }
  Fp pow_u64(uint64_t e) const { return pow_limbs(&e, 1); }
  // Fermat inverse (0 -> 0).
  Fp inv() const {
    uint64_t e[N]; for (int i = 0; i < N; i++) e[i] = P::MOD[i];
    e[0] -= 2;  // all moduli are odd and > 2: no borrow
    return pow_limbs(e, N);
  }
};

// Montgomery-trick batch inversion; zeros stay zero.
template <class F>
static void batch_inverse(F* v, size_t n) {
  std::vector<F> pre(n);
  F acc = F::one();
  for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!v[i].is_zero()) acc = acc * v[i]; }
  F ia = acc.inv();
  for (size_t i = n; i-- > 0;) {
    if (v[i].is_zero()) continue;
    F t = ia * pre[i]; ia = ia * v[i]; v[i] = t;
  }
}

// --------------------------------------------------------------------------------------
// y^2 = x^3 + b, a = 0.
// --------------------------------------------------------------------------------------
template <class C>
struct Aff {
  typedef Fp<typename C::FqP> Fq;
  Fq x, y;  // infinity <=> (0,0) (never on a curve with b != 0)
  static Aff infinity() { Aff a; a.x = Fq::zero(); a.y = Fq::zero(); return a; }
  bool is_inf() const { return x.is_zero() && y.is_zero(); }
  Aff neg() const { Aff a = *this; if (!is_inf()) a.y = y.neg(); return a; }
  static Aff generator() { Aff a; a.x = Fq::from_raw(C::GX); a.y = Fq::from_raw(C::GY); return a; }
  bool on_curve() const {
    if (is_inf()) return true;
    return y.sqr() == x.sqr() * x + Fq::from_raw(C::B_MONT);
  }
  bool operator==(const Aff& o) const { return x == o.x && y == o.y; }
};

template <class C>
struct Jac {
  typedef Fp<typename C::FqP> Fq;
  Fq X, Y, Z;  // infinity <=> Z == 0
  static Jac infinity() { Jac j; j.X = Fq::one(); j.Y = Fq::one(); j.Z = Fq::zero(); return j; }
  static Jac from_affine(const Aff<C>& a) {
    if (a.is_inf()) return infinity();
    Jac j; j.X = a.x; j.Y = a.y; j.Z = Fq::one(); return j;
  }
  bool is_inf() const { return Z.is_zero(); }

  Jac dbl() const {  // dbl-2009-l (a = 0)
    if (is_inf()) return *this;
    Fq A = X.sqr(), B = Y.sqr(), Cc = B.sqr();
    Fq D = ((X + B).sqr() - A - Cc).dbl();
    Fq E = A.dbl() + A, F = E.sqr();
    Jac r;
    r.X = F - D.dbl();
    r.Z = (Y * Z).dbl();
    r.Y = E * (D - r.X) - Cc.dbl().dbl().dbl();
    return r;
  }
  Jac add(const Jac& o) const {  // add-2007-bl with the doubling / inverse cases handled
    if (is_inf()) return o;
    if (o.is_inf()) return *this;
    Fq Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
    Fq U1 = X * Z2Z2, U2 = o.X * Z1Z1;
    Fq S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
    if (U1 == U2) { if (S1 == S2) return dbl(); return infinity(); }
    Fq H = U2 - U1, I = H.dbl().sqr(), J = H * I, r = (S2 - S1).dbl(), V = U1 * I;
    Jac R;
    R.X = r.sqr() - J - V.dbl();
    R.Y = r * (V - R.X) - (S1 * J).dbl();
    R.Z = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return R;
  }
  Jac add_affine(const Aff<C>& o) const {  // madd-2007-bl
    if (o.is_inf()) return *this;
    if (is_inf()) return from_affine(o);
    Fq Z1Z1 = Z.sqr();
    Fq U2 = o.x * Z1Z1, S2 = o.y * Z * Z1Z1;
    if (X == U2) { if (Y == S2) return dbl(); return infinity(); }
    Fq H = U2 - X, HH = H.sqr(), I = HH.dbl().dbl(), J = H * I, r = (S2 - Y).dbl(), V = X * I;
    Jac R;
    R.X = r.sqr() - J - V.dbl();
    R.Y = r * (V - R.X) - (Y * J).dbl();
    R.Z = (Z + H).sqr() - Z1Z1 - HH;
    return R;
  }
  Jac neg() const { Jac r = *this; r.Y = Y.neg(); return r; }
  Aff<C> to_affine() const {
    if (is_inf()) return Aff<C>::infinity();
    Fq zi = Z.inv(), zi2 = zi.sqr();
    Aff<C> a; a.x = X * zi2; a.y = Y * zi2 * zi; return a;
  }
  // scalar given as canonical LE 64-bit limbs
  Jac mul_limbs(const uint64_t* k, int n) const {
    Jac r = infinity();
    for (int i = n * 64 - 1; i >= 0; i--) { r = r.dbl(); if ((k[i / 64] >> (i % 64)) & 1) r = r.add(*this); }
    return r;
  }
};

template <class C>
static void batch_normalize(const Jac<C>* in, Aff<C>* out, size_t n) {
  typedef Fp<typename C::FqP> Fq;
  std::vector<Fq> z(n);
  for (size_t i = 0; i < n; i++) z[i] = in[i].Z;
  batch_inverse(z.data(), n);
  for (size_t i = 0; i < n; i++) {
    if (in[i].is_inf()) { out[i] = Aff<C>::infinity(); continue; }
    Fq zi2 = z[i].sqr();
    out[i].x = in[i].X * zi2; out[i].y = in[i].Y * zi2 * z[i];
  }
}
