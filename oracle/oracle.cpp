// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_field.hpp header and oracle/README.md).
//
// CPU restatement of the reference's commit/open hot path, exported as a flat C API for
// ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Each function cites
// the reference lines it follows (paths relative to /root/reference).
//
// Data conventions at this API (they mirror arkworks' in-memory forms):
//   * field elements ("mont"): Montgomery residues, 64-bit LE limbs (4 per Fr; 6 per
//     BLS12-381 Fq, 4 per BN254/Pallas Fq)
//   * bigint scalars ("canon"): canonical residues, 4 x u64 LE  (F::into_bigint())
//   * affine points: x || y in Montgomery form; the point at infinity is (0, 0)
//
// Parity status: parity unpinned by reference constants (the reference has no golden
// vectors for this path and cannot be built here -- no Rust toolchain, arithmetic lives
// in un-vendored ark-ec/ark-poly 0.5).  Pinned instead against oracle/pyref.py.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include "oracle_field.hpp"

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
template <class Fn>
static void parallel_for(size_t n, int threads, Fn fn) {
  if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<size_t> next(0);
  std::vector<std::thread> th;
  int nt = (int)std::min<size_t>(threads, n);
  for (int t = 0; t < nt; t++)
    th.emplace_back([&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= n) break; fn(i); } });
  for (auto& t : th) t.join();
}

template <class C> static Aff<C> load_aff(const uint64_t* p) {
  constexpr int N = C::FqP::N; Aff<C> a;
  a.x = Fp<typename C::FqP>::from_raw(p); a.y = Fp<typename C::FqP>::from_raw(p + N); return a;
}
template <class C> static void store_aff(const Aff<C>& a, uint64_t* p) {
  constexpr int N = C::FqP::N; a.x.to_raw(p); a.y.to_raw(p + N);
}

static uint32_t ark_log2(size_t x) {  // ark_std::log2 = ceil(log2 x), 0 for x <= 1
  if (x <= 1) return 0;
  uint32_t b = 0; size_t v = x - 1; while (v) { b++; v >>= 1; } return b;
}

// ---------------------------------------------------------------------------------------
// MSM.  ark-ec 0.5 VariableBaseMSM::msm_bigint (external crate; call sites
// poly-commit/src/kzg10/mod.rs:175,255 and ipa_pc/mod.rs:64).
// ---------------------------------------------------------------------------------------
template <class C>
static Jac<C> msm_naive(const uint64_t* bases, const uint64_t* scalars, size_t n) {
  constexpr int N = C::FqP::N;
  Jac<C> acc = Jac<C>::infinity();
  for (size_t i = 0; i < n; i++) {
    Jac<C> b = Jac<C>::from_affine(load_aff<C>(bases + 2 * N * i));
    acc = acc.add(b.mul_limbs(scalars + 4 * i, 4));
  }
  return acc;
}

// ark-ec `make_digits`: signed radix-2^w digits in [-2^(w-1), 2^(w-1)), last digit absorbs
// the carry.
static void make_digits(const uint64_t* s, int w, int num_bits, int64_t* out) {
  uint64_t radix = 1ull << w, mask = radix - 1, carry = 0;
  int count = (num_bits + w - 1) / w;
  for (int i = 0; i < count; i++) {
    int off = i * w, u = off / 64, b = off % 64;
    uint64_t buf;
    if (b < 64 - w || u == 3) buf = s[u] >> b;
    else buf = (s[u] >> b) | (s[u + 1] << (64 - b));
    uint64_t coef = carry + (buf & mask);
    carry = (coef + radix / 2) >> w;
    int64_t d = (int64_t)coef - (int64_t)(carry << w);
    if (i == count - 1) d += (int64_t)(carry << w);
    out[i] = d;
  }
}

// One window-parallel signed-digit Pippenger pass over [bases, scalars) -- the schedule of
// ark-ec's msm_bigint_wnaf: c = 3 if n < 32 else ln_without_floats(n) + 2, one task per
// window, 2^c bucket slots, running-sum reduction, fold windows with c doublings.
template <class C>
static Jac<C> msm_wnaf(const uint64_t* bases, const uint64_t* scalars, size_t n, int threads) {
  constexpr int N = C::FqP::N;
  if (n == 0) return Jac<C>::infinity();
  int c = n < 32 ? 3 : (int)(ark_log2(n) * 69 / 100) + 2;
  int num_bits = C::FrP::BITS;
  int W = (num_bits + c - 1) / c;
  std::vector<int64_t> digits(n * (size_t)W);
  parallel_for((n + 4095) / 4096, threads, [&](size_t blk) {
    size_t e = std::min(n, (blk + 1) * 4096);
    for (size_t i = blk * 4096; i < e; i++) make_digits(scalars + 4 * i, c, num_bits, &digits[i * W]);
  });
  std::vector<Jac<C>> wsum(W);
  parallel_for((size_t)W, threads, [&](size_t w) {
    // ark-ec sizes the bucket vector 1 << c: the last digit absorbs the carry un-recentred,
    // so |d| - 1 can exceed 2^(c-1) - 1 when the top window is a full c bits wide.
    std::vector<Jac<C>> buckets((size_t)1 << c, Jac<C>::infinity());
    for (size_t i = 0; i < n; i++) {
      int64_t d = digits[i * W + w];
      if (d == 0) continue;
      Aff<C> b = load_aff<C>(bases + 2 * N * i);
      if (d > 0) buckets[d - 1] = buckets[d - 1].add_affine(b);
      else buckets[-d - 1] = buckets[-d - 1].add_affine(b.neg());
    }
    Jac<C> run = Jac<C>::infinity(), res = Jac<C>::infinity();
    for (size_t k = buckets.size(); k-- > 0;) { run = run.add(buckets[k]); res = res.add(run); }
    wsum[w] = res;
  });
  Jac<C> total = Jac<C>::infinity();
  for (int w = W - 1; w >= 1; w--) {
    total = total.add(wsum[w]);
    for (int k = 0; k < c; k++) total = total.dbl();
  }
  return total.add(wsum[0]);
}

#include "fast_msm.hpp"

// mode 0: windows are the parallel unit (ark-ec msm_bigint_wnaf under rayon);
// mode 1: input split into `threads` chunks, each a full sequential msm_wnaf, partial
//         results added (the chunk-parallel schedule of later ark-ec releases);
// mode 2: the tuned port (fast_msm.hpp: same bucket method, XYZZ buckets, (window, chunk) tasks) -- the CPU baseline of bench.py.
template <class C>
static Jac<C> msm_pippenger(const uint64_t* bases, const uint64_t* scalars, size_t n, int threads, int mode) {
  constexpr int N = C::FqP::N;
  if (mode == 2) return fastmsm::msm<C>(bases, scalars, n, threads);
  if (mode == 0 || threads <= 1 || n < (size_t)threads * 64) return msm_wnaf<C>(bases, scalars, n, threads);
  size_t chunk = (n + threads - 1) / threads;
  size_t nch = (n + chunk - 1) / chunk;
  std::vector<Jac<C>> part(nch);
  parallel_for(nch, threads, [&](size_t k) {
    size_t s = k * chunk, e = std::min(n, s + chunk);
    part[k] = msm_wnaf<C>(bases + 2 * N * s, scalars + 4 * s, e - s, 1);
  });
  Jac<C> acc = Jac<C>::infinity();
  for (auto& p : part) acc = acc.add(p);
  return acc;
}

// ---------------------------------------------------------------------------------------
// NTT.  ark-poly 0.5 Radix2EvaluationDomain::fft (external), call site
// poly-commit/src/linear_codes/utils.rs:119-126; behaviour pinned by test_reed_solomon
// utils.rs:303-331: out[j] = p(omega^j), natural order,
// omega = TWO_ADIC_ROOT_OF_UNITY^(2^(s - log_n)).
// ---------------------------------------------------------------------------------------
template <class P>
static Fp<P> omega(unsigned log_n) {
  Fp<P> w = Fp<P>::from_raw(P::ROOT);
  for (unsigned i = log_n; i < (unsigned)P::TWO_ADICITY; i++) w = w.sqr();
  return w;
}

template <class P>
static void ntt_inplace(Fp<P>* a, unsigned log_n) {
  typedef Fp<P> F;
  size_t n = (size_t)1 << log_n;
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j |= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (unsigned s = 1; s <= log_n; s++) {
    size_t len = (size_t)1 << s, half = len >> 1;
    F wl = omega<P>(s);
    std::vector<F> tw(half);
    F w = F::one();
    for (size_t k = 0; k < half; k++) { tw[k] = w; w = w * wl; }
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < half; k++) {
        F u = a[i + k], v = a[i + k + half] * tw[k];
        a[i + k] = u + v; a[i + k + half] = u - v;
      }
  }
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
#define CURVE_SWITCH(curve, ...)                              \
  switch (curve) {                                            \
    case 0: { typedef pc_curve_bls12_381 C; __VA_ARGS__; } break; \
    case 1: { typedef pc_curve_bn254 C; __VA_ARGS__; } break;     \
    case 2: { typedef pc_curve_pallas C; __VA_ARGS__; } break;    \
    default: abort();                                         \
  }

extern "C" {

int orc_fq_limbs(int curve) { int r = 0; CURVE_SWITCH(curve, r = C::FqP::N); return r; }
int orc_fr_bits(int curve) { int r = 0; CURVE_SWITCH(curve, r = C::FrP::BITS); return r; }

// ---- field unit ops (which: 0 = Fq, 1 = Fr), Montgomery in / out ----------------------
#define FIELD_OP(NAME, EXPR)                                                              \
  void NAME(int curve, int which, const uint64_t* a, const uint64_t* b, uint64_t* out) {  \
    CURVE_SWITCH(curve, {                                                                 \
      if (which == 0) { typedef Fp<C::FqP> F; F x = F::from_raw(a), y = F::from_raw(b); (void)y; F r = EXPR; r.to_raw(out); } \
      else { typedef Fp<C::FrP> F; F x = F::from_raw(a), y = F::from_raw(b); (void)y; F r = EXPR; r.to_raw(out); }            \
    });                                                                                   \
  }
FIELD_OP(orc_f_mul, x * y)
FIELD_OP(orc_f_add, x + y)
FIELD_OP(orc_f_sub, x - y)
FIELD_OP(orc_f_inv, x.inv())

void orc_f_from_canonical(int curve, int which, const uint64_t* in, uint64_t* out, size_t n) {
  CURVE_SWITCH(curve, {
    if (which == 0) { constexpr int N = C::FqP::N; for (size_t i = 0; i < n; i++) Fp<C::FqP>::from_canonical(in + N * i).to_raw(out + N * i); }
    else { constexpr int N = C::FrP::N; for (size_t i = 0; i < n; i++) Fp<C::FrP>::from_canonical(in + N * i).to_raw(out + N * i); }
  });
}
void orc_f_to_canonical(int curve, int which, const uint64_t* in, uint64_t* out, size_t n) {
  CURVE_SWITCH(curve, {
    if (which == 0) { constexpr int N = C::FqP::N; for (size_t i = 0; i < n; i++) Fp<C::FqP>::from_raw(in + N * i).to_canonical(out + N * i); }
    else { constexpr int N = C::FrP::N; for (size_t i = 0; i < n; i++) Fp<C::FrP>::from_raw(in + N * i).to_canonical(out + N * i); }
  });
}

// ---- synthetic inputs ------------------------------------------
// bases P_i = (i+1) G by repeated affine addition with batched inversion.
void orc_gen_bases(int curve, size_t n, uint64_t* out) {
  CURVE_SWITCH(curve, {
    constexpr int N = C::FqP::N; typedef Fp<C::FqP> Fq;
    if (n == 0) return;
    // doubling ladder: P_{2^k} blocks; simple approach: P_{i+B} = P_i + B*G in batches.
    Aff<C> G = Aff<C>::generator();
    std::vector<Aff<C>> pts(n);
    pts[0] = G;
    size_t have = 1;
    Jac<C> jG = Jac<C>::from_affine(G);
    while (have < n) {
      // step = have * G ; new points pts[have + i] = pts[i] + step, i < min(have, n - have)
      uint64_t k[4] = {have, 0, 0, 0};
      Aff<C> step = jG.mul_limbs(k, 1).to_affine();
      size_t m = std::min(have, n - have);
      std::vector<Fq> den(m);
      for (size_t i = 0; i < m; i++) den[i] = step.x - pts[i].x;   // 0 only for i == have-1 (pts[i] == step)
      batch_inverse(den.data(), m);
      for (size_t i = 0; i < m; i++) {
        if (i == have - 1) { pts[have + i] = Jac<C>::from_affine(step).dbl().to_affine(); continue; }
        Fq lam = (step.y - pts[i].y) * den[i];
        Fq x3 = lam.sqr() - pts[i].x - step.x;
        Fq y3 = lam * (pts[i].x - x3) - pts[i].y;
        pts[have + i].x = x3; pts[have + i].y = y3;
      }
      have += m;
    }
    for (size_t i = 0; i < n; i++) store_aff<C>(pts[i], out + 2 * N * i);
  });
}

static inline uint64_t splitmix64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// n canonical Fr scalars uniform in [0, r): SplitMix64(seed), rejection sampled.
void orc_gen_scalars(int curve, uint64_t seed, size_t n, uint64_t* out) {
  CURVE_SWITCH(curve, {
    typedef C::FrP P;
    uint64_t s = seed;
    uint64_t top_mask = (P::BITS % 64) ? ((1ull << (P::BITS % 64)) - 1) : ~0ull;
    size_t i = 0;
    while (i < n) {
      uint64_t l[4];
      for (int k = 0; k < 4; k++) l[k] = splitmix64(s);
      l[3] &= top_mask;
      if (Fp<P>::geq_mod(l)) continue;
      memcpy(out + 4 * i, l, 32); i++;
    }
  });
}

// ---- EC unit ops ----------------------------------------------------------------------
void orc_ec_add(int curve, const uint64_t* p, const uint64_t* q, uint64_t* out) {
  CURVE_SWITCH(curve, {
    Jac<C> a = Jac<C>::from_affine(load_aff<C>(p));
    store_aff<C>(a.add_affine(load_aff<C>(q)).to_affine(), out);
  });
}
void orc_ec_mul(int curve, const uint64_t* p, const uint64_t* k, uint64_t* out) {
  CURVE_SWITCH(curve, {
    store_aff<C>(Jac<C>::from_affine(load_aff<C>(p)).mul_limbs(k, 4).to_affine(), out);
  });
}
int orc_on_curve(int curve, const uint64_t* p) {
  int r = 0; CURVE_SWITCH(curve, r = load_aff<C>(p).on_curve()); return r;
}
void orc_generator(int curve, uint64_t* out) { CURVE_SWITCH(curve, store_aff<C>(Aff<C>::generator(), out)); }

// ---- MSM ------------------------------------------------------------------------------
void orc_msm_naive(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t* out) {
  CURVE_SWITCH(curve, store_aff<C>(msm_naive<C>(bases, scalars, n).to_affine(), out));
}
void orc_msm_pippenger(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int threads,
                       int mode, uint64_t* out) {
  CURVE_SWITCH(curve, store_aff<C>(msm_pippenger<C>(bases, scalars, n, threads, mode).to_affine(), out));
}

// ---- NTT ------------------------------------------------------------------------------
void orc_root_of_unity(int curve, unsigned log_n, uint64_t* out) {
  CURVE_SWITCH(curve, omega<C::FrP>(log_n).to_raw(out));
}
// rows x in_cols (row-major, Montgomery) -> rows x 2^log_n; each row zero-padded then
// forward NTT, natural order (reed_solomon, linear_codes/utils.rs:112-127).
void orc_ntt_batch(int curve, const uint64_t* in, size_t rows, size_t in_cols, unsigned log_n,
                   uint64_t* out, int threads) {
  CURVE_SWITCH(curve, {
    typedef Fp<C::FrP> F;
    size_t n = (size_t)1 << log_n;
    parallel_for(rows, threads, [&](size_t r) {
      std::vector<F> a(n, F::zero());
      for (size_t i = 0; i < in_cols && i < n; i++) a[i] = F::from_raw(in + 4 * (r * in_cols + i));
      ntt_inplace<C::FrP>(a.data(), log_n);
      for (size_t i = 0; i < n; i++) a[i].to_raw(out + 4 * (r * n + i));
    });
  });
}
void orc_poly_eval(int curve, const uint64_t* coeffs, size_t n, const uint64_t* z, uint64_t* out) {
  CURVE_SWITCH(curve, {
    typedef Fp<C::FrP> F;
    F acc = F::zero(), zz = F::from_raw(z);
    for (size_t i = n; i-- > 0;) acc = acc * zz + F::from_raw(coeffs + 4 * i);
    acc.to_raw(out);
  });
}

// ---- KZG10 (hiding off) ---------------------------------------------------------------
// compute_witness_polynomial, kzg10/mod.rs:217-240: quotient of p by (x - z).
void orc_witness_poly(int curve, const uint64_t* coeffs, size_t n, const uint64_t* z, uint64_t* out) {
  CURVE_SWITCH(curve, {
    typedef Fp<C::FrP> F;
    if (n <= 1) return;
    F acc = F::zero(), zz = F::from_raw(z);
    for (size_t i = n - 1; i >= 1; i--) { acc = F::from_raw(coeffs + 4 * i) + zz * acc; acc.to_raw(out + 4 * (i - 1)); }
  });
}

// KZG10::commit, kzg10/mod.rs:157-210 with hiding_bound = None:
//   check_degree_is_too_large :393-403 (degree = index of last non-zero coeff),
//   skip_leading_zeros_and_convert_to_bigints :452-461, msm_bigint :175-178, into affine :209.
// returns 0, or -1 for Error::TooManyCoefficients.
int orc_kzg_commit(int curve, const uint64_t* powers, size_t n_powers, const uint64_t* coeffs, size_t n,
                   int threads, uint64_t* out) {
  int rc = 0;
  CURVE_SWITCH(curve, {
    typedef Fp<C::FrP> F; constexpr int N = C::FqP::N;
    size_t deg_p1 = n;
    while (deg_p1 > 0 && F::from_raw(coeffs + 4 * (deg_p1 - 1)).is_zero()) deg_p1--;
    size_t num_coeffs = deg_p1 == 0 ? 1 : deg_p1;   // degree 0 for the zero polynomial
    if (num_coeffs > n_powers) { rc = -1; }
    else {
      size_t lz = 0;
      while (lz < n && F::from_raw(coeffs + 4 * lz).is_zero()) lz++;
      size_t m = n - lz;
      std::vector<uint64_t> big(4 * (m ? m : 1));
      for (size_t i = 0; i < m; i++) F::from_raw(coeffs + 4 * (lz + i)).to_canonical(&big[4 * i]);
      size_t nb = n_powers > lz ? n_powers - lz : 0;
      size_t pairs = std::min(nb, m);   // msm_bigint uses min(len)
      store_aff<C>(msm_pippenger<C>(powers + 2 * N * lz, big.data(), pairs, threads, 2).to_affine(), out);
    }
  });
  return rc;
}
// KZG10::open, kzg10/mod.rs:287-310 -> open_with_witness_polynomial :243-284 (no hiding).
int orc_kzg_open(int curve, const uint64_t* powers, size_t n_powers, const uint64_t* coeffs, size_t n,
                 const uint64_t* z, int threads, uint64_t* out) {
  std::vector<uint64_t> q(4 * (n > 1 ? n - 1 : 1), 0);
  orc_witness_poly(curve, coeffs, n, z, q.data());
  return orc_kzg_commit(curve, powers, n_powers, q.data(), n > 1 ? n - 1 : 0, threads, out);
}

// ---- BLAKE2s-256 (RFC 7693), byte-oriented: the digest D of InnerProductArgPC<G, D, P> in the reference's
//      tests and benches (ipa_pc/mod.rs:1056-1064, benches/ipa_times.rs:16).  Pinned against Python's hashlib
//      in tests/test_oracle_cpu.py. ------------------------------------------------------------------------
namespace b2s {
static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t SIGMA[10][16] = {
  {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
  {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
  {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
  {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
  {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void compress(uint32_t h[8], const uint8_t block[64], uint64_t t, bool last) {
  uint32_t m[16], v[16];
  for (int i = 0; i < 16; i++) m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) | ((uint32_t)block[4 * i + 3] << 24);
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
  v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] = ~v[14];
  auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
    v[a] = v[a] + v[b] + x; v[d] = ror(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = ror(v[b] ^ v[c], 12);
    v[a] = v[a] + v[b] + y; v[d] = ror(v[d] ^ v[a], 8); v[c] = v[c] + v[d]; v[b] = ror(v[b] ^ v[c], 7);
  };
  for (int r = 0; r < 10; r++) {
    const uint8_t* s = SIGMA[r];
    G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
    G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
static void digest(const uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t h[8];
  for (int i = 0; i < 8; i++) h[i] = IV[i];
  h[0] ^= 0x01010020u;                                   // digest length 32, no key, fanout 1, depth 1
  size_t off = 0;
  while (len - off > 64) { compress(h, msg + off, off + 64, false); off += 64; }
  uint8_t block[64]; memset(block, 0, 64);
  memcpy(block, msg + off, len - off);
  compress(h, block, len, true);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)h[i]; out[4 * i + 1] = (uint8_t)(h[i] >> 8); out[4 * i + 2] = (uint8_t)(h[i] >> 16); out[4 * i + 3] = (uint8_t)(h[i] >> 24); }
}
}  // namespace b2s
extern "C" void orc_blake2s(const uint8_t* msg, size_t len, uint8_t* out32) { b2s::digest(msg, len, out32); }
extern "C++" {
// ---- Transcript bytes: ark-serialize's serialize_uncompressed of the items InnerProductArgPC::open hashes
//      (ipa_pc/mod.rs:615-623, 681-688).  ark-serialize / ark-ec 0.5 are not under /root/reference; restated from
//      their published behaviour:
//        field element            canonical residue, little-endian, ceil(bits / 8) bytes
//        short-Weierstrass point  (generic impl: BN254, Pallas)  x as above, then y in ceil((bits + 2) / 8) bytes
//                                 with the SWFlags in the top bits of the LAST byte: 0x80 = YIsNegative = y is the larger of
//                                 {y, -y} (y > -y; YIsPositive, y <= -y, sets no bit), 0x40 = point at infinity (x = y = 0)
//        BLS12-381 G1             ark-bls12-381 overrides the generic impl with the zcash / IETF encoding:
//                                 x, y big-endian, 48 bytes each; top bits of byte 0: 0x80 compressed (clear here),
//                                 0x40 infinity (all other bytes zero), 0x20 unused in the uncompressed form
template <class F>
static void ser_field(const F& v, size_t nbytes, std::vector<uint8_t>& out) {
  uint64_t c[F::N]; v.to_canonical(c);
  for (size_t i = 0; i < nbytes; i++) out.push_back(i < 8 * (size_t)F::N ? (uint8_t)(c[i / 8] >> (8 * (i % 8))) : 0);
}
template <class C>
static void ser_point(int curve, const Aff<C>& p, std::vector<uint8_t>& out) {
  typedef Fp<typename C::FqP> Fq;
  const size_t xb = (C::FqP::BITS + 7) / 8, yb = (C::FqP::BITS + 2 + 7) / 8;
  if (curve == 0) {                       // zcash encoding
    const size_t at = out.size();
    std::vector<uint8_t> le;
    if (p.is_inf()) { out.insert(out.end(), 2 * xb, 0); out[at] |= 0x40; return; }
    ser_field(p.x, xb, le); out.insert(out.end(), le.rbegin(), le.rend());
    le.clear(); ser_field(p.y, xb, le); out.insert(out.end(), le.rbegin(), le.rend());
    return;
  }
  if (p.is_inf()) { out.insert(out.end(), xb + yb, 0); out.back() |= 0x40; return; }
  ser_field(p.x, xb, out);
  ser_field(p.y, yb, out);
  // SWFlags::from_y_coordinate: y <= -y -> YIsPositive (no bit); y > -y -> YIsNegative (0x80)
  uint64_t a[Fq::N], b[Fq::N]; p.y.to_canonical(a); p.y.neg().to_canonical(b);
  bool y_gt_neg = false;
  for (int i = Fq::N - 1; i >= 0; i--) if (a[i] != b[i]) { y_gt_neg = a[i] > b[i]; break; }
  if (y_gt_neg) out.back() |= 0x80;
}
// Field::from_random_bytes (ark-ff): the first 8 N bytes as a little-endian integer with the bits above
// MODULUS_BIT_SIZE cleared; None if that integer is >= the modulus.
template <class F>
static bool from_random_bytes(const uint8_t* bytes, size_t len, F& out) {
  uint64_t c[F::N];
  for (int i = 0; i < F::N; i++) { c[i] = 0; for (int k = 0; k < 8; k++) { size_t idx = 8 * i + k; if (idx < len) c[i] |= (uint64_t)bytes[idx] << (8 * k); } }
  const int shave = 64 * F::N - F::Params::BITS;
  if (shave > 0) c[F::N - 1] &= ~(uint64_t)0 >> shave;
  for (int i = F::N - 1; i >= 0; i--) { if (c[i] < F::Params::MOD[i]) break; if (c[i] > F::Params::MOD[i] || i == 0) return false; }
  out = F::from_canonical(c);
  return true;
}
// compute_random_oracle_challenge, ipa_pc/mod.rs:74-87
template <class F>
static F random_oracle_challenge(const std::vector<uint8_t>& bytes) {
  for (uint64_t i = 0;; i++) {
    std::vector<uint8_t> in(bytes);
    for (int k = 0; k < 8; k++) in.push_back((uint8_t)(i >> (8 * k)));
    uint8_t h[32]; b2s::digest(in.data(), in.size(), h);
    F out;
    if (from_random_bytes<F>(h, 32, out)) return out;
  }
}
}  // extern "C++"

// ---- IPA halving rounds, ipa_pc/mod.rs:664-711 ------------------------------------------
// comm_key: n affine points; coeffs: n Fr (mont); z: evaluation point (mont); h_prime: 1 affine.
// challenges != null: log2(n) Fr (mont) supplied by the caller.  challenges == null: Fiat-Shamir as in the
// reference -- round_challenge = RO(ser(round_challenge) || ser(L) || ser(R)) (:681-688), starting from
// *rc0 (the challenge open() derived before the loop, :615-625); the challenges used are written to ch_out.
// Outputs: l_vec / r_vec (log2 n affine points each), final_comm_key (1 affine), c (1 Fr mont).
extern "C++" {
template <class C>
static void ipa_rounds_impl(int curve, const uint64_t* comm_key, const uint64_t* coeffs_in, size_t n, const uint64_t* zpt,
                            const uint64_t* h_prime, const uint64_t* challenges, const uint64_t* rc0, int threads, uint64_t* l_out,
                            uint64_t* r_out, uint64_t* final_key, uint64_t* c_out, uint64_t* ch_out) {
    typedef Fp<typename C::FrP> F; constexpr int N = C::FqP::N;
    std::vector<F> cs(n), zs(n);
    F cur = F::one(), zz = F::from_raw(zpt);
    for (size_t i = 0; i < n; i++) { cs[i] = F::from_raw(coeffs_in + 4 * i); zs[i] = cur; cur = cur * zz; }
    std::vector<uint64_t> key(comm_key, comm_key + 2 * N * n);
    Jac<C> hp = Jac<C>::from_affine(load_aff<C>(h_prime));
    size_t rnd = 0;
    std::vector<uint64_t> big(4 * n);
    F round_challenge = rc0 ? F::from_raw(rc0) : F::zero();
    auto cm_commit = [&](const uint64_t* k, const F* s, size_t m) {   // ipa_pc/mod.rs:54-72
      for (size_t i = 0; i < m; i++) s[i].to_canonical(&big[4 * i]);
      return msm_pippenger<C>(k, big.data(), m, threads, 1);
    };
    auto inner = [&](const F* a, const F* b, size_t m) {              // utils.rs:150-155
      F acc = F::zero(); for (size_t i = 0; i < m; i++) acc = acc + a[i] * b[i]; return acc;
    };
    while (n > 1) {
      size_t h = n / 2;
      uint64_t ipc[4];
      inner(&cs[h], &zs[0], h).to_canonical(ipc);
      Jac<C> l = cm_commit(key.data(), &cs[h], h).add(hp.mul_limbs(ipc, 4));
      inner(&cs[0], &zs[h], h).to_canonical(ipc);
      Jac<C> r = cm_commit(key.data() + 2 * N * h, &cs[0], h).add(hp.mul_limbs(ipc, 4));
      Aff<C> la = l.to_affine(), ra = r.to_affine();
      store_aff<C>(la, l_out + 2 * N * rnd);
      store_aff<C>(ra, r_out + 2 * N * rnd);
      F u;
      if (challenges) u = F::from_raw(challenges + 4 * rnd);
      else {
        std::vector<uint8_t> bytes;
        ser_field(round_challenge, (C::FrP::BITS + 7) / 8, bytes); ser_point<C>(curve, la, bytes); ser_point<C>(curve, ra, bytes);
        u = round_challenge = random_oracle_challenge<F>(bytes);
      }
      if (ch_out) u.to_raw(ch_out + 4 * rnd);
      F ui = u.inv();
      rnd++;
      uint64_t uc[4]; u.to_canonical(uc);
      for (size_t i = 0; i < h; i++) { cs[i] = cs[i] + ui * cs[h + i]; zs[i] = zs[i] + u * zs[h + i]; }
      std::vector<Jac<C>> kp(h);
      parallel_for((h + 63) / 64, threads, [&](size_t blk) {
        size_t e = std::min(h, (blk + 1) * 64);
        for (size_t i = blk * 64; i < e; i++) {
          Jac<C> kr = Jac<C>::from_affine(load_aff<C>(&key[2 * N * (h + i)])).mul_limbs(uc, 4);
          kp[i] = kr.add_affine(load_aff<C>(&key[2 * N * i]));
        }
      });
      std::vector<Aff<C>> ka(h);
      batch_normalize<C>(kp.data(), ka.data(), h);
      for (size_t i = 0; i < h; i++) store_aff<C>(ka[i], &key[2 * N * i]);
      n = h;
    }
    memcpy(final_key, key.data(), 2 * N * 8);
    cs[0].to_raw(c_out);
}
}  // extern "C++"
void orc_ipa_rounds(int curve, const uint64_t* comm_key, const uint64_t* coeffs_in, size_t n,
                    const uint64_t* zpt, const uint64_t* h_prime, const uint64_t* challenges, int threads,
                    uint64_t* l_out, uint64_t* r_out, uint64_t* final_key, uint64_t* c_out) {
  CURVE_SWITCH(curve, ipa_rounds_impl<C>(curve, comm_key, coeffs_in, n, zpt, h_prime, challenges, nullptr, threads, l_out, r_out, final_key, c_out, nullptr));
}
void orc_ipa_rounds_fs(int curve, const uint64_t* comm_key, const uint64_t* coeffs_in, size_t n, const uint64_t* zpt,
                       const uint64_t* h_prime, const uint64_t* round_challenge0, int threads, uint64_t* l_out, uint64_t* r_out,
                       uint64_t* final_key, uint64_t* c_out, uint64_t* challenges_out) {
  CURVE_SWITCH(curve, ipa_rounds_impl<C>(curve, comm_key, coeffs_in, n, zpt, h_prime, nullptr, round_challenge0, threads, l_out, r_out, final_key, c_out, challenges_out));
}
// The challenge open() derives before the loop (ipa_pc/mod.rs:615-625): RO(ser(combined_commitment) || ser(point) ||
// ser(combined_v)); and the generic transcript pieces, for the tests.
void orc_ipa_first_challenge(int curve, const uint64_t* commitment_xy, const uint64_t* point, const uint64_t* value, uint64_t* out) {
  CURVE_SWITCH(curve, {
    typedef Fp<C::FrP> F;
    std::vector<uint8_t> bytes;
    ser_point<C>(curve, load_aff<C>(commitment_xy), bytes);
    ser_field(F::from_raw(point), (C::FrP::BITS + 7) / 8, bytes);
    ser_field(F::from_raw(value), (C::FrP::BITS + 7) / 8, bytes);
    random_oracle_challenge<F>(bytes).to_raw(out);
  });
}
size_t orc_ser_point(int curve, const uint64_t* xy, uint8_t* out) {
  size_t n = 0;
  CURVE_SWITCH(curve, { std::vector<uint8_t> b; ser_point<C>(curve, load_aff<C>(xy), b); memcpy(out, b.data(), b.size()); n = b.size(); });
  return n;
}

// ---- Ligero shape + encode ------------------------------------------------------------
// calculate_t, linear_codes/utils.rs:156-184; compute_dimensions, ligero.rs:118-128.
int orc_ligero_dims(int field_bits, size_t poly_len, size_t rho_inv, int sec_param, size_t* n_rows,
                    size_t* n_cols, size_t* t_out) {
  double residual = (double)poly_len / pow(2.0, field_bits);
  double rhs = log2(pow(2.0, -sec_param) - residual);
  if (!isnormal(rhs)) return -1;
  double nom = rhs - 1.0;
  double denom = log2(1.0 - 0.5 * (double)(rho_inv - 1) / (double)rho_inv);
  if (!isnormal(denom)) return -1;
  size_t t = (size_t)ceil(nom / denom);
  if (t >= poly_len) t = poly_len;
  size_t q = (2 * poly_len + t - 1) / t;
  size_t sq = (size_t)ceil(sqrt((double)q));
  size_t n = (size_t)1 << ark_log2(sq);
  size_t m = (poly_len + n - 1) / n;
  *n_rows = n; *n_cols = m; *t_out = t;
  return 0;
}
// compute_matrices, linear_codes/mod.rs:118-138: pad to n_rows*n_cols, row-major, encode
// each row with reed_solomon (utils.rs:112-127).  ext must hold n_rows * next_pow2(n_cols*rho_inv).
void orc_ligero_encode(int curve, const uint64_t* coeffs, size_t len, size_t n_rows, size_t n_cols,
                       size_t rho_inv, uint64_t* ext, int threads) {
  size_t size = 1; unsigned lg = 0;
  while (size < n_cols * rho_inv) { size <<= 1; lg++; }
  std::vector<uint64_t> mat(4 * n_rows * n_cols, 0);
  memcpy(mat.data(), coeffs, 32 * std::min(len, n_rows * n_cols));
  orc_ntt_batch(curve, mat.data(), n_rows, n_cols, lg, ext, threads);
}

}  // extern "C"
