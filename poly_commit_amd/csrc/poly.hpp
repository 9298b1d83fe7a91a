// Scalar-field polynomial kernels on the open path.
//
// witness polynomial: quotient of p(x) by (x - z) -- KZG10::compute_witness_polynomial,
// poly-commit/src/kzg10/mod.rs:217-240 (ark-poly's `&p / &divisor`).  It is the linear
// recurrence acc = p[i] + z*acc, q[i-1] = acc (i = n-1 .. 1), evaluated as a chunked scan of
// affine maps: each lane folds a chunk of G elements (up-sweep, carry-in 0), the per-chunk
// results are folded again with factor z^G, ... and the carries are pushed back down.
#pragma once
#include <vector>
#include "fp32.hpp"

namespace pc {

// out[u] = Horner(x[u*G .. min(count,(u+1)*G)), high index first, factor f, carry-in 0)
template <class FrP>
struct ScanUpBody {
  static constexpr bool LATENCY_BOUND = true;   // raised wave priority beside an accumulation (hip_backend.hpp)
  typedef Fd<FrP> F;
  const uint32_t* x; uint32_t count; uint32_t G; F f; uint32_t* out;
  const uint32_t* extra;      // optional virtual element x[count - 1] (the caller's carry-in), else null
  PC_HD F elem(uint32_t j) const { return (extra && j + 1 == count) ? F::load(extra) : F::load(x + (size_t)j * FrP::N); }
  PC_HD void operator()(uint32_t u) const {
    uint32_t s = u * G, e = (count - s > G) ? s + G : count;
    F acc = F::zero();
    for (uint32_t j = e; j-- > s;) acc = elem(j).add(f.mul(acc));
    acc.store(out + (size_t)u * FrP::N);
  }
};

// group u: acc = carry_in[u] (or 0); for j high..low: [pre] out[j] = acc; acc = x[j] + f*acc; [post] out[j] = acc
template <class FrP>
struct ScanDownBody {
  static constexpr bool LATENCY_BOUND = true;   // raised wave priority beside an accumulation (hip_backend.hpp)
  typedef Fd<FrP> F;
  const uint32_t* x; uint32_t count; uint32_t G; F f;
  const uint32_t* carry_in;   // one per group, may be null
  uint32_t* out;              // one per element
  uint32_t post;              // 0: store the carry INTO element j; 1: store the value AFTER element j
  const uint32_t* extra;      // optional virtual top element (see ScanUpBody); never stored
  PC_HD void operator()(uint32_t u) const {
    uint32_t s = u * G, e = (count - s > G) ? s + G : count;
    F acc = carry_in ? F::load(carry_in + (size_t)u * FrP::N) : F::zero();
    for (uint32_t j = e; j-- > s;) {
      const bool virt = extra && j + 1 == count;
      if (!post && !virt) acc.store(out + (size_t)j * FrP::N);
      acc = (virt ? F::load(extra) : F::load(x + (size_t)j * FrP::N)).add(f.mul(acc));
      if (post && !virt) acc.store(out + (size_t)j * FrP::N);
    }
  }
};

// Chunked scan of the division recurrence over x[0..count):
//   acc = carry_in (or 0);  for j = count-1 .. 0:  acc = x[j] + z*acc;  out[j] = acc.
// carry_in_host: one Montgomery Fr on the host, or null.  x / out are device pointers.
template <class FrP, class Backend>
void div_scan(Backend& be, const uint32_t* x0, size_t count_in, const uint32_t* z_host, const uint32_t* carry_in_host,
              uint32_t* out, uint32_t G = 16, uint32_t G0 = 8) {
  typedef Fd<FrP> F;
  if (count_in == 0) return;
  // A carry-in c is the same as one more (virtual) coefficient on top: acc = c + z*0 = c.
  uint32_t count0 = (uint32_t)count_in + (carry_in_host ? 1 : 0);
  F z = F::load(z_host);
  // Level 0 touches every coefficient: short chunks (G0 = 8 -> 256 contiguous bytes per lane,
  // count0/8 lanes) keep it bandwidth-bound instead of latency-bound; upper levels are tiny and
  // latency-bound: fan-in G = 16 keeps every serial chain short (74 instead of 168 dependent
  // steps per sweep at 2^20).
  std::vector<uint32_t> counts, fan; std::vector<F> factors;
  counts.push_back(count0); factors.push_back(z);
  do {
    const uint32_t g = counts.size() == 1 ? G0 : G;
    F f = factors.back(), fg = F::one();
    for (uint32_t i = 0; i < g; i++) fg = fg.mul(f);      // f^g
    fan.push_back(g);
    counts.push_back((counts.back() + g - 1) / g); factors.push_back(fg);
  } while (counts.back() > 1);
  const size_t L = counts.size() - 1;                       // number of up-sweeps
  size_t total = 0; for (size_t k = 1; k <= L; k++) total += counts[k];
  uint32_t* buf = (uint32_t*)be.workspace((2 * total + 2) * (size_t)FrP::N * 4);   // cached: no per-call hipMalloc/hipFree
  std::vector<uint32_t*> B(L + 1), Cc(L + 1);
  uint32_t* cur = buf;
  for (size_t k = 1; k <= L; k++) { B[k] = cur; cur += (size_t)counts[k] * FrP::N; }
  for (size_t k = 1; k <= L; k++) { Cc[k] = cur; cur += (size_t)counts[k] * FrP::N; }
  uint32_t* extra = nullptr;
  if (carry_in_host) { extra = cur; be.copy_h2d(extra, carry_in_host, (size_t)FrP::N * 4); }
  B[0] = const_cast<uint32_t*>(x0);
  for (size_t k = 0; k < L; k++) {
    ScanUpBody<FrP> b{B[k], counts[k], fan[k], factors[k], B[k + 1], k == 0 ? extra : nullptr};
    be.launch(b, counts[k + 1]);
  }
  // down-sweep: the single element of level L has carry-in 0
  for (size_t k = L; k-- > 0;) {
    const uint32_t* cin = (k + 1 == L) ? nullptr : Cc[k + 1];
    if (k > 0) {
      ScanDownBody<FrP> b{B[k], counts[k], fan[k], factors[k], cin, Cc[k], 0, nullptr};
      be.launch(b, counts[k + 1]);
    } else {
      ScanDownBody<FrP> b{B[0], counts[0], fan[0], factors[0], cin, out, 1, extra};
      be.launch(b, counts[1]);
    }
  }
  be.sync();
}

// p(z) = sum_i x[i] z^i: the up-sweep of the division scan alone (no carries pushed back down, no
// output polynomial) -- what a shard of a polynomial contributes to the shards below it.
template <class FrP, class Backend>
void poly_eval(Backend& be, const uint32_t* x0, size_t count_in, const uint32_t* z_host, uint32_t* out_host, uint32_t G = 16,
               uint32_t G0 = 8) {
  typedef Fd<FrP> F;
  if (count_in == 0) { F::zero().store(out_host); return; }
  std::vector<uint32_t> counts, fan; std::vector<F> factors;
  counts.push_back((uint32_t)count_in); factors.push_back(F::load(z_host));
  do {
    const uint32_t g = counts.size() == 1 ? G0 : G;
    F f = factors.back(), fg = F::one();
    for (uint32_t i = 0; i < g; i++) fg = fg.mul(f);
    fan.push_back(g);
    counts.push_back((counts.back() + g - 1) / g); factors.push_back(fg);
  } while (counts.back() > 1);
  const size_t L = counts.size() - 1;
  size_t total = 0; for (size_t k = 1; k <= L; k++) total += counts[k];
  uint32_t* buf = (uint32_t*)be.workspace((total + 1) * (size_t)FrP::N * 4);
  const uint32_t* src = x0; uint32_t* dst = buf;
  for (size_t k = 0; k < L; k++) {
    ScanUpBody<FrP> b{src, counts[k], fan[k], factors[k], dst, nullptr};
    be.launch(b, counts[k + 1]);
    src = dst; dst += (size_t)counts[k + 1] * FrP::N;
  }
  be.copy_d2h(out_host, src, (size_t)FrP::N * 4);      // the single element of the last level
}

// q (n-1 elements) = p (n elements) / (x - z): q[i-1] = value after element i, i = n-1 .. 1.
template <class FrP, class Backend>
void witness_polynomial(Backend& be, const uint32_t* p, size_t n, const uint32_t* z_host, uint32_t* q, uint32_t G = 16,
                        uint32_t G0 = 8) {
  if (n <= 1) return;
  div_scan<FrP>(be, p + FrP::N, n - 1, z_host, nullptr, q, G, G0);
}

// p = sum_j xi_j * p_j, coefficient-wise: the combination MarlinKZG10::open forms before the
// single witness division (marlin/marlin_pc/mod.rs:281-287, `p += (challenge_j, polynomial)`),
// and the shifted twin for degree-bounded polynomials (:291-301).  Polynomials may have
// different lengths (missing high coefficients are zero).  One lane per output coefficient:
// consecutive lanes read consecutive coefficients of the same polynomial (coalesced), the
// xi_j and the pointer table are wave-uniform loads.  64 x 2^20 Fr: 2 GiB read, 32 MiB written.
template <class FrP>
struct FrLinCombBody {
  typedef Fd<FrP> F;
  const uint64_t* polys;   // k device addresses
  const uint32_t* lens;    // k lengths
  const uint32_t* xi;      // k Fr, Montgomery
  uint32_t k; uint32_t* out;
  PC_HD void operator()(uint32_t i) const {
    F acc = F::zero();
    for (uint32_t j = 0; j < k; j++) {
      if (i >= lens[j]) continue;
      const uint32_t* p = reinterpret_cast<const uint32_t*>(polys[j]);
      acc = acc.add(F::load(xi + (size_t)j * FrP::N).mul(F::load(p + (size_t)i * FrP::N)));
    }
    acc.store(out + (size_t)i * FrP::N);
  }
};

}  // namespace pc
