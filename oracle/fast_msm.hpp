// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_field.hpp).  The CPU *baseline* leg of bench.py: the same signed-digit bucket
// method as ark-ec 0.5's VariableBaseMSM::msm_bigint (external crate; call sites poly-commit/src/kzg10/mod.rs:175-178, :255-258,
// ipa_pc/mod.rs:64), written to run at the speed a tuned CPU library runs at, so that the GPU/CPU ratio the north star asks for is
// read against a credible number.  It is a PORT, not ark-ec: labelled so wherever it is printed.
//
//   * field: unrolled CIOS Montgomery product on 64-bit limbs through unsigned __int128 (mulx / adc chains under -O3), the
//     "no spare carry word" form that every modulus here allows (top bit clear), dedicated squaring left to the compiler;
//   * buckets: XYZZ coordinates, mixed addition 8M + 2S (ark-ec: Jacobian, 7M + 4S), bucket and base prefetched a few pairs ahead;
//   * digits: d_j = digit_j(k + H) - 2^(c-1) with H = sum_j 2^(jc + c - 1): the same digits as ark-ec's make_digits, computed per
//     (scalar, window) without a pass that stores them;
//   * schedule: (window, chunk) tasks from a shared queue -- ark-ec parallelises over the ~16 windows only, which is why it stops
//     scaling at ~16 cores; chunks keep all cores of the GPU box busy.  threads == 1 with ark-ec's own
//     window rule is the per-core figure.
// Checked against the generic oracle (msm_naive / msm_wnaf) in tests/test_oracle_cpu.py.
#pragma once
#include "oracle_field.hpp"

namespace fastmsm {

template <class P>
struct FF {
  static constexpr int N = P::N;
  uint64_t l[N];
  static inline FF zero() { FF r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
  static inline FF one() { FF r; for (int i = 0; i < N; i++) r.l[i] = P::ONE[i]; return r; }
  inline bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  // a -= p if a >= p (branch-free)
  static inline void reduce_once(uint64_t* a) {
    uint64_t d[N]; u128 br = 0;
    for (int i = 0; i < N; i++) { u128 t = (u128)a[i] - P::MOD[i] - (uint64_t)br; d[i] = (uint64_t)t; br = (t >> 64) & 1; }
    const uint64_t keep = (uint64_t)0 - (uint64_t)br;      // all ones: a < p
    for (int i = 0; i < N; i++) a[i] = (a[i] & keep) | (d[i] & ~keep);
  }
  inline FF add(const FF& o) const {
    FF r; u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    reduce_once(r.l);
    return r;
  }
  inline FF sub(const FF& o) const {
    FF r; u128 br = 0;
    for (int i = 0; i < N; i++) { u128 t = (u128)l[i] - o.l[i] - (uint64_t)br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; }
    const uint64_t m = (uint64_t)0 - (uint64_t)br;
    u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)r.l[i] + (P::MOD[i] & m); r.l[i] = (uint64_t)c; c >>= 64; }
    return r;
  }
  inline FF dbl() const { return add(*this); }
  inline FF neg() const { FF z = zero(); return is_zero() ? z : z.sub(*this); }
  // CIOS without the extra carry words (the top bit of every modulus here is clear: the running value stays below 2p < 2^(64N))
  inline FF mul(const FF& o) const {
    uint64_t t[N];
    for (int i = 0; i < N; i++) t[i] = 0;
#pragma GCC unroll 8
    for (int i = 0; i < N; i++) {
      const uint64_t b = o.l[i];
      u128 c = (u128)l[0] * b + t[0];
      const uint64_t m = (uint64_t)c * P::INV;
      u128 r = (u128)m * P::MOD[0] + (uint64_t)c;
      uint64_t hi1 = (uint64_t)(c >> 64), hi2 = (uint64_t)(r >> 64);
#pragma GCC unroll 8
      for (int j = 1; j < N; j++) {
        c = (u128)l[j] * b + t[j] + hi1;
        hi1 = (uint64_t)(c >> 64);
        r = (u128)m * P::MOD[j] + (uint64_t)c + hi2;
        hi2 = (uint64_t)(r >> 64);
        t[j - 1] = (uint64_t)r;
      }
      t[N - 1] = hi1 + hi2;
    }
    FF r; for (int i = 0; i < N; i++) r.l[i] = t[i];
    reduce_once(r.l);
    return r;
  }
  inline FF sqr() const { return mul(*this); }
};

template <class C>
struct Xyzz {
  typedef FF<typename C::FqP> Fq;
  Fq X, Y, ZZ, ZZZ;
  static inline Xyzz infinity() { Xyzz r; r.X = Fq::zero(); r.Y = Fq::zero(); r.ZZ = Fq::zero(); r.ZZZ = Fq::zero(); return r; }
  inline bool is_inf() const { return ZZ.is_zero(); }
  inline Xyzz dbl() const {      // dbl-2008-s-1, a = 0
    if (is_inf() || Y.is_zero()) return infinity();
    Xyzz r;
    Fq U = Y.dbl(), V = U.sqr(), W = U.mul(V), S = X.mul(V), xx = X.sqr(), M = xx.dbl().add(xx);
    r.X = M.sqr().sub(S.dbl());
    r.Y = M.mul(S.sub(r.X)).sub(W.mul(Y));
    r.ZZ = V.mul(ZZ); r.ZZZ = W.mul(ZZZ);
    return r;
  }
  // this += (x, +-y) affine, madd-2008-s; all special cases
  inline void madd(const Fq& ax, const Fq& ay_in, bool negate) {
    if (ax.is_zero() && ay_in.is_zero()) return;
    const Fq ay = negate ? ay_in.neg() : ay_in;
    if (is_inf()) { X = ax; Y = ay; ZZ = Fq::one(); ZZZ = Fq::one(); return; }
    Fq U2 = ax.mul(ZZ), S2 = ay.mul(ZZZ), Pp = U2.sub(X), R = S2.sub(Y);
    if (Pp.is_zero()) {
      if (!R.is_zero()) { *this = infinity(); return; }
      Xyzz a; a.X = ax; a.Y = ay; a.ZZ = Fq::one(); a.ZZZ = Fq::one();
      *this = a.dbl();
      return;
    }
    Fq PP = Pp.sqr(), PPP = Pp.mul(PP), Q = X.mul(PP);
    Fq X3 = R.sqr().sub(PPP).sub(Q.dbl());
    Y = R.mul(Q.sub(X3)).sub(Y.mul(PPP));
    X = X3; ZZ = ZZ.mul(PP); ZZZ = ZZZ.mul(PPP);
  }
  inline void add(const Xyzz& o) {   // add-2008-s; all special cases
    if (o.is_inf()) return;
    if (is_inf()) { *this = o; return; }
    Fq U1 = X.mul(o.ZZ), U2 = o.X.mul(ZZ), S1 = Y.mul(o.ZZZ), S2 = o.Y.mul(ZZZ), Pp = U2.sub(U1), R = S2.sub(S1);
    if (Pp.is_zero()) { if (R.is_zero()) *this = dbl(); else *this = infinity(); return; }
    Fq PP = Pp.sqr(), PPP = Pp.mul(PP), Q = U1.mul(PP);
    Fq X3 = R.sqr().sub(PPP).sub(Q.dbl());
    Y = R.mul(Q.sub(X3)).sub(S1.mul(PPP));
    X = X3; ZZ = ZZ.mul(o.ZZ).mul(PP); ZZZ = ZZZ.mul(o.ZZZ).mul(PPP);
  }
};

// XYZZ -> the generic oracle's Jacobian form without an inversion: (x, y) = (X/ZZ, Y/ZZZ) and ZZ^3 = ZZZ^2; with Z := ZZ,
// x = (X ZZ) / Z^2 and y = (Y ZZZ) / Z^3.
template <class C>
static Jac<C> to_jac(const Xyzz<C>& p) {
  typedef Fp<typename C::FqP> G; typedef FF<typename C::FqP> Fq;
  if (p.is_inf()) return Jac<C>::infinity();
  const Fq xj = p.X.mul(p.ZZ), yj = p.Y.mul(p.ZZZ);
  Jac<C> j; j.X = G::from_raw(xj.l); j.Y = G::from_raw(yj.l); j.Z = G::from_raw(p.ZZ.l);
  return j;
}

// digit_w(k + H) - 2^(c-1) for one window; kh = k + H as 5 limbs (H added once per scalar and task)
static inline int64_t window_digit(const uint64_t* kh, int w, int c) {
  const int off = w * c, u = off >> 6, b = off & 63;
  uint64_t v = kh[u] >> b;
  if (b + c > 64 && u + 1 < 5) v |= kh[u + 1] << (64 - b);
  return (int64_t)(v & (((uint64_t)1 << c) - 1)) - ((int64_t)1 << (c - 1));
}

// sum over pairs [s, e) of digit_w(k_i) P_i: one bucket set, running-sum reduction
template <class C>
static Xyzz<C> window_task(const uint64_t* bases, const uint64_t* scalars, size_t s, size_t e, int w, int c, const uint64_t* H,
                           std::vector<Xyzz<C>>& buckets) {
  typedef FF<typename C::FqP> Fq;
  constexpr int N = C::FqP::N;
  const size_t nb = (size_t)1 << (c - 1);          // |d| in 1 .. 2^(c-1)
  for (size_t i = 0; i < nb; i++) buckets[i] = Xyzz<C>::infinity();
  constexpr size_t AHEAD = 8;
  auto digit_of = [&](size_t i) {
    uint64_t kh[5]; u128 cy = 0;
    for (int j = 0; j < 4; j++) { cy += (u128)scalars[4 * i + j] + H[j]; kh[j] = (uint64_t)cy; cy >>= 64; }
    kh[4] = (uint64_t)cy + H[4];
    return window_digit(kh, w, c);
  };
  for (size_t i = s; i < e; i++) {
    if (i + AHEAD < e) {
      const int64_t dn = digit_of(i + AHEAD);
      if (dn) __builtin_prefetch(&buckets[(size_t)(dn < 0 ? -dn : dn) - 1], 1, 1);
      __builtin_prefetch(bases + 2 * N * (i + AHEAD), 0, 0);
    }
    const int64_t d = digit_of(i);
    if (!d) continue;
    const Fq* pt = reinterpret_cast<const Fq*>(bases + 2 * N * i);
    buckets[(size_t)(d < 0 ? -d : d) - 1].madd(pt[0], pt[1], d < 0);
  }
  Xyzz<C> run = Xyzz<C>::infinity(), res = Xyzz<C>::infinity();
  for (size_t k = nb; k-- > 0;) { run.add(buckets[k]); res.add(run); }
  return res;
}

// ark-ec's window rule (c = 3 below 32 pairs, else ln_without_floats(n) + 2) for one thread; with more threads the width that
// minimises a task's madd + reduction cost for chunks of n / chunks pairs
static inline int choose_c(size_t n, int bits, size_t chunks_per_window) {
  if (chunks_per_window <= 1) return n < 32 ? 3 : (int)(ark_log2(n) * 69 / 100) + 2;
  int best = 4; double bc = 1e300;
  for (int c = 4; c <= 22; c++) {
    const int W = bits / c + 1;
    const double cost = (double)W * ((double)n + (double)chunks_per_window * 2.8 * (double)((size_t)1 << (c - 1)));
    if (cost < bc) { bc = cost; best = c; }
  }
  return best;
}

template <class C>
static Jac<C> msm(const uint64_t* bases, const uint64_t* scalars, size_t n, int threads) {
  if (n == 0) return Jac<C>::infinity();
  const int bits = C::FrP::BITS;
  if (threads < 1) threads = 1;
  {   // a thread is worth starting for ~2^11 bucket operations (~1 ms); small MSMs stay on few threads
    const int c0 = choose_c(n, bits, 1);
    const size_t ops = n * (size_t)(bits / c0 + 1);
    const size_t worth = ops / ((size_t)1 << 11) + 1;
    if ((size_t)threads > worth) threads = (int)worth;
  }
  // tasks: W windows x `chunks` slices of the pairs, about three per thread
  int c = choose_c(n, bits, 1);
  size_t chunks = 1;
  if (threads > 1) {
    const int W0 = bits / c + 1;
    chunks = std::max<size_t>(1, ((size_t)3 * threads + W0 - 1) / W0);
    if (chunks > n / 256 + 1) chunks = n / 256 + 1;
    c = choose_c(n, bits, chunks);
    const int W1 = bits / c + 1;
    chunks = std::max<size_t>(1, std::min<size_t>(n / 256 + 1, ((size_t)3 * threads + W1 - 1) / W1));
  }
  const int W = bits / c + 1;                      // one more digit than ceil(bits / c) when c | bits: the recoding's carry
  uint64_t H[5] = {0, 0, 0, 0, 0};
  for (int j = 0; j < W; j++) { const int pos = j * c + c - 1; if (pos < 320) H[pos >> 6] |= (uint64_t)1 << (pos & 63); }
  const size_t per = (n + chunks - 1) / chunks;
  std::vector<Xyzz<C>> part((size_t)W * chunks);
  const size_t ntasks = (size_t)W * chunks;
  std::atomic<size_t> next(0);
  auto worker = [&]() {
    std::vector<Xyzz<C>> buckets((size_t)1 << (c - 1));
    for (;;) {
      const size_t t = next.fetch_add(1);
      if (t >= ntasks) break;
      const size_t w = t / chunks, j = t % chunks;
      const size_t s = j * per, e = std::min(n, s + per);
      part[t] = s < e ? window_task<C>(bases, scalars, s, e, (int)w, c, H, buckets) : Xyzz<C>::infinity();
    }
  };
  const int nt = (int)std::min<size_t>((size_t)threads, ntasks);
  if (nt <= 1) worker();
  else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(worker); for (auto& t : th) t.join(); }
  Xyzz<C> total = Xyzz<C>::infinity();
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c && w != W - 1; k++) total = total.dbl();
    for (size_t j = 0; j < chunks; j++) total.add(part[(size_t)w * chunks + j]);
  }
  return to_jac<C>(total);
}

}  // namespace fastmsm
