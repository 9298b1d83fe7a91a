// Variable-base MSM for gfx950: sum_i k_i * P_i over a resident SRS.
//
// Replaces <E::G1 as VariableBaseMSM>::msm_bigint at its call sites in the reference
// (poly-commit/src/kzg10/mod.rs:175-178, :255-258; ipa_pc/mod.rs:64).  The result is
// returned as an affine point, which is canonical, so it is bit-identical to ark-ec's
// result after into_affine() whatever the internal schedule.
//
// Pipeline (all on one HIP stream, no host round trip until the final download):
//   1-3. bucket sort     signed radix-2^c digits of every scalar (optional Montgomery->canonical
//                        fused) -> entries (base index | sign<<31) grouped by (window, bucket) plus
//                        CSR offsets.  HIP: two-level LDS radix sort (msm_sort.hpp).  Reference
//                        version kept below (histogram atomics / scan / cursor scatter): it is what
//                        the CPU stepping tests run and what PC_HIP_SORT=atomic selects.
//   4. accumulate        the flat, bucket-sorted entry array is cut into equal chunks of T
//                        entries, one chunk per lane: every lane executes exactly T mixed
//                        additions (no load imbalance whatever the scalar distribution).
//                        Runs that lie wholly inside a chunk are written to their bucket;
//                        the (at most two) runs cut by a chunk edge go to a partial list --
//                        on HIP after neighbouring lanes merged the two halves of a cut run
//                        (k_accumulate, msm_coop.hpp), which finishes almost every bucket.
//   5. seg-reduce        the partial list (sorted by bucket by construction) is reduced
//                        level by level with the same chunk rule until every bucket is whole
//   6. bucket reduce     sum_j (j+1) B_j per window, recursively Red(X) = sum(Tw) + K * Red(S):
//                        wide levels by lane-serial running sums (fan-in 4), small levels as
//                        workgroup-cooperative "bits" levels (fan-in up to 256 in log2 K dependent
//                        additions); older partial sums ride along as plain fan-in-K trees
//   7. host tail         the last level's <= W * (#arrays) points are downloaded (pinned, async) and
//                        folded by one descending Horner chain (255 dependent doublings: one CPU
//                        core beats one GPU lane 40x here) + one inversion to affine
// The window width c is chosen per call from the call's n (plan_geometry); a plan is sized for
// every n <= n_max.  enqueue() queues one MSM, finish() waits for it: several plans (pipelines)
// per SRS overlap the latency-bound steps 5-7 of one MSM with step 4 of the next.
//
// The kernel bodies are functors templated on nothing but the curve; the orchestration is a
// template over a Backend that provides launch/alloc/scan.  The product instantiates it
// with HipBackend only (hip_backend.hpp); tests/emu instantiates the same code with a
// single-threaded CPU stepping backend to validate the indexing logic without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include <stdexcept>
#include <vector>
#include "ec.hpp"
#include "glv.hpp"
#include "host_tail.hpp"

// "these values must have arrived": an empty asm that reads one register of every 16-byte load of an affine point (and
// an index word) makes the compiler place its s_waitcnt here
#if defined(__HIP_DEVICE_COMPILE__)
#define PC_ARRIVED(PT, IDX)                                                                                             \
  do {                                                                                                                  \
    constexpr int PC_N_ = sizeof((PT).x.l) / 4;                                                                         \
    for (int pc_i_ = 0; pc_i_ < PC_N_; pc_i_ += 4) asm volatile("" ::"v"((PT).x.l[pc_i_]), "v"((PT).y.l[pc_i_]));     \
    asm volatile("" ::"v"(IDX));                                                                                        \
  } while (0)
#define PC_ARRIVED_WORD(W) asm volatile("" ::"v"(W))
#else
#define PC_ARRIVED(PT, IDX) ((void)0)
#define PC_ARRIVED_WORD(W) ((void)0)
#endif

// measurement switch (never set in the product build): PC_ACC_DEBUG_IDX_MASK=m gathers table point (index & m) instead of the entry's --
// wrong results, but the accumulate kernel then runs with its gathers in cache: the bound on what a deeper prefetch could still buy
#ifndef PC_ACC_DEBUG_IDX_MASK
#define PC_ACC_DEBUG_IDX_MASK 0x7fffffffu
#endif

namespace pc {

#if defined(__HIP_DEVICE_COMPILE__)
PC_D uint32_t atomic_inc_u32(uint32_t* p) { return atomicAdd(p, 1u); }
#else
inline uint32_t atomic_inc_u32(uint32_t* p) { uint32_t o = *p; *p = o + 1; return o; }
#endif

static constexpr uint32_t KEY_INVALID = 0xffffffffu;

// a call that does not fit the workspace of its plan (reported as PC_ERR_TOO_LARGE, nothing launched)
struct MsmCapacityError : std::runtime_error { using std::runtime_error::runtime_error; };

struct MsmGeom {
  uint32_t n;          // pairs in this call
  uint32_t c;          // window bits
  uint32_t W;          // bucket sets: one per window, or 1 when the windows share buckets (table mode)
  uint32_t nb_win;     // buckets per window = 2^(c-1)
  uint32_t NB;         // total buckets = W * nb_win
  uint32_t base_off;   // offset of this call's first base inside the resident SRS
  uint32_t from_mont;  // scalars arrive as Montgomery residues
  uint32_t T;          // level-0 chunk length
  uint32_t T2;         // chunk length of seg-reduce level 1
  uint32_t T2b;        // chunk length of the deeper (sparser) seg-reduce levels
  uint32_t Wd;         // signed digits per scalar = bits / c + 1
  uint32_t tbl_stride; // 0: classic (digit w goes to bucket set w and adds base i);  else the bases are a
                       // precomputed table T[w][i] = 2^(c w) P_i of `tbl_stride` points per window: digit w
                       // adds T[w][i] into the ONE shared bucket set (W == 1)
  uint32_t pt_stride;  // words between consecutive points of the array the accumulate kernel gathers from
  uint32_t m_sub;      // 0: one MSM.  else the n scalars are n / m_sub independent MSMs of m_sub pairs over the SAME
                       // bases (table mode only): sub-MSM s owns bucket set s, scalar i adds table[w][i mod m_sub]
  const uint64_t* scalar_tab;   // many-MSM mode, optional: device array of one scalar-vector address per sub-MSM (the
                                // polynomials of a batch lie in separate buffers); null: one contiguous n x Fr array
  uint32_t glv;        // table mode only: the table holds Wd / 2 windows (2^(c w) P_i for the 130-bit halves of the GLV split
                       // k = k1 + k2 lambda); the digits of k1 go to bucket set 0, those of k2 to bucket set 1 with the SAME table
                       // point, and phi (x -> beta x) is applied once to the reduced sum of set 1 (phi is a homomorphism):
                       // half the table, the same number of additions.  Wd then counts the digits of BOTH halves.
  // address of scalar i = (sub, j)
  PC_HD const uint32_t* scalar_at(const uint32_t* scalars, uint32_t i, uint32_t sub, uint32_t j, int fr_words) const {
    return scalar_tab ? reinterpret_cast<const uint32_t*>(scalar_tab[sub]) + (size_t)j * fr_words : scalars + (size_t)i * fr_words;
  }
  PC_HD uint32_t sets_per_msm() const { return glv ? 2u : 1u; }
  PC_HD uint32_t windows() const { return glv ? Wd / 2 : Wd; }      // windows a (half-)scalar is cut into
  PC_HD uint32_t key_window(uint32_t w, uint32_t sub, uint32_t h = 0) const { return m_sub ? sub * sets_per_msm() + h : tbl_stride ? h : w; }
  PC_HD uint32_t base_index(uint32_t w, uint32_t j) const { return tbl_stride ? w * tbl_stride + base_off + j : base_off + j; }
  // scalar i -> (sub-MSM, position inside it)
  PC_HD void split(uint32_t i, uint32_t& sub, uint32_t& j) const { if (m_sub) { sub = i / m_sub; j = i - sub * m_sub; } else { sub = 0; j = i; } }
};

PC_HD uint32_t msm_num_windows(uint32_t bits, uint32_t c) { return bits / c + 1; }

// Signed radix-2^c recoding: digit in [-(2^(c-1)-1), 2^(c-1)].
// Returns key+1 (0 for a zero digit) and the sign.
template <class FrP>
struct ScalarDigits {
  uint32_t s[FrP::N];
  PC_HD void load(const uint32_t* p, bool from_mont) {
    Fd<FrP> f = Fd<FrP>::load(p);
    if (from_mont) f = f.from_mont();
    PC_UNROLL for (int i = 0; i < FrP::N; i++) s[i] = f.l[i];
  }
  PC_HD uint32_t bits_at(uint32_t off, uint32_t c) const {
    uint32_t w = off >> 5, b = off & 31;
    if (w >= (uint32_t)FrP::N) return 0;
    uint64_t v = s[w];
    if (w + 1 < (uint32_t)FrP::N) v |= (uint64_t)s[w + 1] << 32;
    return (uint32_t)(v >> b) & ((1u << c) - 1u);
  }
  // f(w, bits [w c, (w+1) c)) for w = 0 .. Wd-1, in order.  The limbs are fed through a 64-bit shift register with
  // COMPILE-TIME limb indices: bits_at()'s run-time index into s[] puts the scalar into scratch memory on the
  // device (36 bytes per lane in the sort passes: a store and several loads per scalar through the private segment).
  template <class F>
  PC_HD void for_each_window(uint32_t c, uint32_t Wd, F f) const {
    uint64_t buf = 0; uint32_t have = 0, w = 0;
    const uint32_t mask = (1u << c) - 1u;
    // preconditions (asserted where the geometry is made, plan_geometry): 2 <= c <= 24 and Wd * c > 32 * (N - 1), i.e. the
    // windows consume every limb but possibly the last; a caller that asked for fewer windows stops the limb loop instead
    // of letting `have` grow past the register (round-2 advisor finding)
    PC_UNROLL for (int i = 0; i < FrP::N; i++) {
      if (w >= Wd) break;
      buf |= (uint64_t)s[i] << have; have += 32;                 // have < c <= 24 before: no bit is lost
      while (have >= c && w < Wd) { f(w, (uint32_t)buf & mask); buf >>= c; have -= c; w++; }
    }
    while (w < Wd) { f(w, (uint32_t)buf & mask); buf >>= c; w++; }   // the (short) top window(s), zero-extended
  }
};

// the windows of a little-endian limb array (same shift register as ScalarDigits::for_each_window, any limb count)
template <int NL, class F>
PC_HD void for_each_window_limbs(const uint32_t* s, uint32_t c, uint32_t Wd, F f) {
  uint64_t buf = 0; uint32_t have = 0, w = 0;
  const uint32_t mask = (1u << c) - 1u;
  PC_UNROLL for (int i = 0; i < NL; i++) {
    if (w >= Wd) break;
    buf |= (uint64_t)s[i] << have; have += 32;
    while (have >= c && w < Wd) { f(w, (uint32_t)buf & mask); buf >>= c; have -= c; w++; }
  }
  while (w < Wd) { f(w, (uint32_t)buf & mask); buf >>= c; w++; }
}

// Every non-zero signed digit of one scalar: f(half, window, magnitude in [1, 2^(c-1)], negative).  Plain mode: half = 0, the
// scalar's Wd windows.  GLV table mode (Geo::glv): the scalar is split into k1, k2 (GlvHalves), each half is recoded over
// Wd / 2 windows and a half's own sign flips its digits' signs.  Geo: MsmGeom or SortGeom.
// GLV: 0 plain, 1 the split done here, 2 `scalar` points at a PRE-SPLIT record (GlvPresplit below: the halves' magnitudes, canonical,
// their signs in the top bits) -- the form the LDS sort's two passes read after k_glv_presplit ran once.
static constexpr int GLV_SPLIT_WORDS = 10;            // 2 x 5 limbs per pre-split scalar (40 bytes)
template <class C, int GLV, class Geo, class F>
PC_HD void for_each_signed_digit_t(const Geo& g, const uint32_t* scalar, F f) {
  typedef typename C::FrP FrP;
  const uint32_t half = 1u << (g.c - 1);
  if constexpr (GLV == 0) {
    ScalarDigits<FrP> sd; sd.load(scalar, g.from_mont);
    uint32_t carry = 0;
    sd.for_each_window(g.c, g.Wd, [&](uint32_t w, uint32_t bits) {
      const uint32_t raw = bits + carry;
      carry = raw > half;
      const uint32_t mag = carry ? (2 * half - raw) : raw;
      if (mag) f(0u, w, mag, carry);
    });
  } else {
    uint32_t m[2][5], neg[2];
    if constexpr (GLV == 1) {
      ScalarDigits<FrP> sd; sd.load(scalar, g.from_mont);
      GlvHalves<C> hv; hv.split(sd.s);
      PC_UNROLL for (int h = 0; h < 2; h++) { neg[h] = hv.neg[h]; PC_UNROLL for (int k = 0; k < 5; k++) m[h][k] = hv.m[h][k]; }
    } else {
      PC_UNROLL for (int h = 0; h < 2; h++) {
        PC_UNROLL for (int k = 0; k < 5; k++) m[h][k] = scalar[5 * h + k];
        neg[h] = m[h][4] >> 31; m[h][4] &= 0x7fffffffu;
      }
    }
    PC_UNROLL for (int h = 0; h < 2; h++) {
      uint32_t carry = 0;
      const uint32_t ng = neg[h];
      for_each_window_limbs<5>(m[h], g.c, g.Wd / 2, [&](uint32_t w, uint32_t bits) {
        const uint32_t raw = bits + carry;
        carry = raw > half;
        const uint32_t mag = carry ? (2 * half - raw) : raw;
        if (mag) f((uint32_t)h, w, mag, carry ^ ng);
      });
    }
  }
}
// One scalar -> its pre-split record (the sort's GLV passes then cost what the plain ones do: the split's temporaries tripled
// their registers, 33 -> 83-94 VGPRs, which kept them from running beside an accumulation -- round 4: pipelined step 76.8 vs 69.3 ms)
template <class C>
struct GlvPresplitBody {
  typedef typename C::FrP FrP;
  MsmGeom g;
  const uint32_t* scalars;
  uint32_t* split;          // n x GLV_SPLIT_WORDS
  PC_HD void operator()(uint32_t i) const {
    uint32_t sub, j; g.split(i, sub, j);
    ScalarDigits<FrP> sd; sd.load(g.scalar_at(scalars, i, sub, j, FrP::N), g.from_mont);
    GlvHalves<C> hv; hv.split(sd.s);
    uint32_t* o = split + (size_t)i * GLV_SPLIT_WORDS;
    PC_UNROLL for (int h = 0; h < 2; h++) {
      PC_UNROLL for (int k = 0; k < 4; k++) o[5 * h + k] = hv.m[h][k];
      o[5 * h + 4] = hv.m[h][4] | (hv.neg[h] << 31);          // |k_h| < 2^130: limb 4 holds two bits
    }
  }
};
// (the mode as a run-time flag: the CPU-stepped bodies and the atomic reference sort; the LDS sort's kernels take it as a template
// parameter -- the split's temporaries would triple the registers of the plain passes, 33 -> 94 VGPRs, and halve their occupancy)
template <class C, class Geo, class F>
PC_HD void for_each_signed_digit(const Geo& g, const uint32_t* scalar, F f) {
  if (g.glv) for_each_signed_digit_t<C, 1>(g, scalar, f); else for_each_signed_digit_t<C, 0>(g, scalar, f);
}

// ---------------------------------------------------------------------------------------
// 1. digits + histogram
// ---------------------------------------------------------------------------------------
template <class C>
struct DigitsHistBody {
  typedef typename C::FrP FrP;
  MsmGeom g;
  const uint32_t* scalars;   // n x FrP::N
  uint32_t* hist;            // NB counters (zeroed)
  PC_HD void operator()(uint32_t i) const {
    uint32_t sub, j; g.split(i, sub, j);
    for_each_signed_digit<C>(g, g.scalar_at(scalars, i, sub, j, FrP::N), [&](uint32_t h, uint32_t w, uint32_t mag, uint32_t) {
      atomic_inc_u32(hist + (size_t)g.key_window(w, sub, h) * g.nb_win + (mag - 1));
    });
  }
};

// ---------------------------------------------------------------------------------------
// 3. scatter
// ---------------------------------------------------------------------------------------
template <class C>
struct ScatterBody {
  typedef typename C::FrP FrP;
  MsmGeom g;
  const uint32_t* scalars;
  uint32_t* cursor;          // NB, initialised to the bucket offsets
  uint32_t* entries;         // M = offsets[NB] slots
  PC_HD void operator()(uint32_t i) const {
    uint32_t sub, j; g.split(i, sub, j);
    for_each_signed_digit<C>(g, g.scalar_at(scalars, i, sub, j, FrP::N), [&](uint32_t h, uint32_t w, uint32_t mag, uint32_t neg) {
      const uint32_t pos = atomic_inc_u32(cursor + (size_t)g.key_window(w, sub, h) * g.nb_win + (mag - 1));
      entries[pos] = g.base_index(w, j) | (neg << 31);
    });
  }
};

// first index k in [0, n] with a[k] > v, minus one: a[k] <= v < a[k+1]
PC_HD uint32_t find_bucket(const uint32_t* offs, uint32_t nb, uint32_t v) {
  uint32_t lo = 0, hi = nb;   // invariant: offs[lo] <= v, answer in [lo, hi)
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (offs[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

// Slot range [lo, hi] that the partials of bucket `key` occupy at reduction level `level`
// (level 1 = output of the accumulate kernel).  See the header comment, step 5.
PC_HD void partial_slot_range(const uint32_t* offs, uint32_t key, uint32_t T, uint32_t T2, uint32_t T2b, uint32_t level,
                              uint32_t& lo, uint32_t& hi) {
  lo = 2 * (offs[key] / T);
  hi = 2 * ((offs[key + 1] - 1) / T);
  for (uint32_t l = 2; l <= level; l++) { const uint32_t d = (l == 2) ? T2 : T2b; lo = 2 * (lo / d); hi = 2 * (hi / d); }
}

// ---------------------------------------------------------------------------------------
// 4. accumulate: one chunk of T sorted entries per lane
// ---------------------------------------------------------------------------------------
template <class C>
struct AccumulateBody {
  typedef XyzzD<C> Pt;
  static constexpr int AW = 2 * Fd<typename C::FqP>::N;   // words per affine base
  MsmGeom g;
  const uint32_t* bases;     // resident SRS, AW words per point
  const uint32_t* entries;
  const uint32_t* offsets;   // NB + 1
  uint32_t* buckets;         // NB x Pt::WORDS (zeroed = infinity)
  uint32_t* pkeys;           // 2 slots per lane
  uint32_t* ppts;            // 2 x Pt::WORDS per lane
  // complete: the whole run [offsets[k], offsets[k+1]) lies inside this lane's chunk
  // the running sum is kept with lazily reduced coordinates where the field allows it (ec.hpp, add_affine_lz): canonical
  // again wherever it leaves the lane
  static constexpr bool LAZY = Pt::Fq::LAZY_OK;
  // ... and may leave it that way only where the consumers' first operations absorb coordinates below 2p (fp32.hpp, LAZY_STORE_OK:
  // R >= 9p, BLS12-381); BN254 (R = 5.3p) canonicalises at the flush
  static constexpr bool LAZY_STORE = LAZY && Pt::Fq::LAZY_STORE_OK;
  // the form in which a running sum leaves the lane (bucket store, partial list, the register copy the in-workgroup joins use)
  static PC_HD Pt store_form(const Pt& acc) { if constexpr (LAZY && !LAZY_STORE) return acc.canonical(); else return acc; }
  PC_HD void flush(const Pt& acc_lz, uint32_t k, bool complete, uint32_t t, bool first, uint32_t& k0, uint32_t& k1) const {
    // (LAZY_STORE: the sum leaves the lane lazily reduced, coordinates in [0, 2p): every consumer -- the joins of k_accumulate, the
    // segmented and the bucket reduction, the host tail -- feeds loaded coordinates into multiplications first (XyzzD::add), or doubles Y
    // once and squares (XyzzD::dbl: 2Y < 3p after its conditional subtraction, 9p^2 < pR); infinity is the exact ZZ == 0 either way and
    // Y = 0 (mod p) cannot occur (no 2-torsion on these curves).  Saves four conditional subtractions per flush: 1.3 % of the kernel)
    const Pt& acc = acc_lz;
    if (complete) { acc.store(buckets + (size_t)k * Pt::WORDS); return; }
    uint32_t slot = first ? 2 * t : 2 * t + 1;
    acc.store(ppts + (size_t)slot * Pt::WORDS);
    if (first) k0 = k; else k1 = k;
  }
  // One lane's chunk.  Returns the keys of its (at most two) partial runs -- their points are in
  // slots 2t / 2t+1 -- and, in `last`, the register copy of the last run's sum (valid iff k1 is).
  PC_HD void chunk(uint32_t t, uint32_t& k0, uint32_t& k1, Pt& last) const {
    const uint32_t M = offsets[g.NB];
    const uint64_t s64 = (uint64_t)t * g.T;
    k0 = KEY_INVALID; k1 = KEY_INVALID;
    if (s64 < M) {
      const uint32_t s = (uint32_t)s64;
      const uint32_t e = (M - s > g.T) ? s + g.T : M;
      uint32_t k = find_bucket(offsets, g.NB, s);
      // the run of bucket k is [run_lo, boundary); `next_boundary` = offsets[k + 2] is fetched one bucket
      // ahead so that a boundary costs no dependent load (it is hit on nearly every iteration by some lane)
      uint32_t run_lo = offsets[k], boundary = offsets[k + 1];
      uint32_t next_boundary = offsets[k + 2 <= g.NB ? k + 2 : g.NB];
      Pt acc = Pt::infinity();
      bool first = true;
      uint32_t val = entries[s];
      uint32_t nval = (s + 1 < e) ? entries[s + 1] : val;
      AffD<C> pt = AffD<C>::load(bases + (size_t)(val & PC_ACC_DEBUG_IDX_MASK) * g.pt_stride);
      for (uint32_t p = s; p < e; p++) {
        // Everything in flight here (the base, the index and the bucket offset gathered during the previous addition) has had a
        // whole addition to arrive: wait for it NOW, before the boundary block below issues its bucket stores.  The memory counter is
        // in order, so the compiler's own wait -- at the first use of `pt`, behind that block -- became vmcnt(0) over the just-issued
        // stores: every wave sat out a store round trip on the iterations in which one of its lanes crosses a bucket boundary.
        PC_ARRIVED(pt, nval);
        PC_ARRIVED_WORD(next_boundary);
        // Bucket boundary.  NOTHING in this block may wait on memory: some lane of a wave crosses a boundary in 50 % (96 entries per
        // bucket) to 90 % (26) of the iterations.  Until round 4 the look-ahead offsets[k + 2] was loaded HERE -- the compiler
        // copied the loaded value into the loop-carried register at the end of the block, i.e. waited for it, and (in-order counter)
        // for the bucket stores issued just before: a full memory round trip per boundary iteration, ~3 us of a 13 us (254-bit) or
        // 22 us (381-bit) iteration -- the 24 % / 12 % of wave cycles in s_waitcnt that the SQ counters showed.  The look-ahead now
        // rides in the prefetch group below (one more 4-byte load per iteration, nearly always the same cached line).
        if (p == boundary) {
          flush(store_form(acc), k, run_lo >= s, t, first, k0, k1);       // its end, p, is inside the chunk
          first = false; acc = Pt::infinity();
          k++; run_lo = p; boundary = next_boundary;
          while (boundary <= p) { k++; boundary = offsets[k + 1]; }      // empty buckets (rare): all start at p
        }
        // Software pipeline, issued right before the long addition: the base of entry p+1 (its index arrived an iteration ago), the
        // index of entry p+2 and the end of the NEXT bucket's run, offsets[k + 2] (clamped to offsets[NB] = M behind the last bucket:
        // no position of the chunk reaches it)
        // (all three UNCONDITIONAL, with clamped positions -- behind the chunk's end `nval` repeats a valid index: under `if (p + 1 < e)`
        // the compiler gathered into scratch registers and assembled `npt` from them with copies, i.e. waited for the gather it had
        // just issued -- s_waitcnt vmcnt(3) / vmcnt(2) right behind the four loads of the 8-limb kernels, a memory round trip in
        // EVERY iteration)
        const AffD<C> npt = AffD<C>::load(bases + (size_t)(nval & PC_ACC_DEBUG_IDX_MASK) * g.pt_stride);
        const uint32_t nnval = entries[p + 2 < e ? p + 2 : e - 1];
        next_boundary = offsets[k + 2 <= g.NB ? k + 2 : g.NB];
        if constexpr (LAZY) acc.add_affine_lz(pt, (val >> 31) != 0); else acc.add_affine(pt.neg_if(val >> 31));
        val = nval; nval = nnval; pt = npt;
      }
      acc = store_form(acc);
      flush(acc, k, run_lo >= s && boundary <= e, t, first, k0, k1);
      last = acc;
    }
  }
  PC_HD void operator()(uint32_t t) const {
    uint32_t k0, k1; Pt last;
    chunk(t, k0, k1, last);
    pkeys[2 * t] = k0; pkeys[2 * t + 1] = k1;
  }
};

// ---------------------------------------------------------------------------------------
// 5. segmented reduction of the partial list, level >= 1 -> level + 1
// ---------------------------------------------------------------------------------------
template <class C>
struct SegReduceBody {
  static constexpr bool LATENCY_BOUND = true;   // raised wave priority beside an accumulation (hip_backend.hpp)
  typedef XyzzD<C> Pt;
  MsmGeom g;
  uint32_t level;            // level of the INPUT slots (>= 1)
  uint32_t n_in;             // input slots
  const uint32_t* in_keys; const uint32_t* in_pts;
  const uint32_t* offsets;
  uint32_t* buckets;
  uint32_t* out_keys; uint32_t* out_pts;   // 2 slots per lane
  PC_HD void flush(const Pt& acc, uint32_t key, uint32_t cs, uint32_t ce, uint32_t u, bool first, uint32_t& k0, uint32_t& k1) const {
    uint32_t lo, hi; partial_slot_range(offsets, key, g.T, g.T2, g.T2b, level, lo, hi);
    if (lo >= cs && hi < ce) { acc.store(buckets + (size_t)key * Pt::WORDS); return; }
    uint32_t slot = first ? 2 * u : 2 * u + 1;
    acc.store(out_pts + (size_t)slot * Pt::WORDS);
    if (first) k0 = key; else k1 = key;
  }
  PC_HD void operator()(uint32_t u) const {
    const uint32_t T2l = level == 1 ? g.T2 : g.T2b;
    const uint32_t cs = u * T2l;
    const uint32_t ce = (n_in - cs > T2l) ? cs + T2l : n_in;
    uint32_t k0 = KEY_INVALID, k1 = KEY_INVALID;
    uint32_t cur = KEY_INVALID; bool first = true;
    Pt acc = Pt::infinity();
    for (uint32_t p = cs; p < ce; p++) {
      uint32_t key = in_keys[p];
      if (key == KEY_INVALID) continue;
      if (key != cur) {
        if (cur != KEY_INVALID) { flush(acc, cur, cs, ce, u, first, k0, k1); first = false; }
        cur = key; acc = Pt::infinity();
      }
      acc.add(Pt::load(in_pts + (size_t)p * Pt::WORDS));
    }
    if (cur != KEY_INVALID) flush(acc, cur, cs, ce, u, first, k0, k1);
    out_keys[2 * u] = k0; out_keys[2 * u + 1] = k1;
  }
};

// ---------------------------------------------------------------------------------------
// 6. bucket reduction level:  X (wb x m) -> S (wb x m/K), Tw (wb x m/K);  plain fan-in-K sums
//    of the older Tw arrays ride along (which = 1 + j).
// ---------------------------------------------------------------------------------------
// One launch per level: lanes [0, cnt) do the weighted running sums of this level; lanes
// [cnt*(1+a), cnt*(2+a)) fold older plain array a (fan-in K).  Serial depth per level =
// one launch instead of (1 + l).
template <class C>
struct BucketLevelBody {
  static constexpr bool LATENCY_BOUND = true;   // raised wave priority beside an accumulation (hip_backend.hpp)
  typedef XyzzD<C> Pt;
  uint32_t K, weight_off, cnt, n_old;
  const uint32_t* x;           // weighted input, cnt * K points
  const uint32_t* old_in;      // n_old arrays of cnt * K points each (previous level's arrays 1..n_old)
  uint32_t* out;               // this level's region: [S][Tw][old 0 reduced]...[old n_old-1 reduced], cnt points each
  PC_HD void operator()(uint32_t lane) const {
    const uint32_t a = lane / cnt, gidx = lane % cnt;
    const size_t stride = (size_t)cnt * Pt::WORDS;
    // (the next point is loaded while the current one is added: the lanes of a wave read K * 192 bytes apart, so every
    // load is a DRAM/L2 round trip of its own, and with one wave per SIMD nothing else hides it -- the SQ counters
    // showed 47 % of this kernel's wave cycles in s_waitcnt.  Unconditional with a clamped index: guarded by `if (j > 0)` the
    // compiler loaded into scratch registers and copied, waiting for part of the prefetch right behind its issue)
    if (a == 0) {
      const uint32_t* base = x + (size_t)gidx * K * Pt::WORDS;
      Pt run = Pt::infinity(), acc = Pt::infinity();
      Pt nxt = Pt::load(base + (size_t)(K - 1) * Pt::WORDS);
      for (uint32_t j = K; j-- > 0;) {
        const Pt cur = nxt;
        nxt = Pt::load(base + (size_t)(j > 0 ? j - 1 : 0) * Pt::WORDS);      // unconditional (clamped): see AccumulateBody::chunk
        run.add(cur);
        if (j + weight_off > 0) acc.add(run);
      }
      run.store(out + (size_t)gidx * Pt::WORDS);
      acc.store(out + stride + (size_t)gidx * Pt::WORDS);
    } else {
      const uint32_t* base = old_in + (size_t)(a - 1) * cnt * K * Pt::WORDS + (size_t)gidx * K * Pt::WORDS;
      Pt acc = Pt::infinity();
      Pt nxt = Pt::load(base);
      for (uint32_t j = 0; j < K; j++) {
        const Pt cur = nxt;
        nxt = Pt::load(base + (size_t)(j + 1 < K ? j + 1 : j) * Pt::WORDS);
        acc.add(cur);
      }
      acc.store(out + (size_t)(1 + a) * stride + (size_t)gidx * Pt::WORDS);
    }
  }
};

// "Bits" level (K >= 16): the weighted sum of a group is returned as log2(K) plain partial sums
//   A_b = sum of the X_l whose index l has bit b set,   sum_l l*X_l = sum_b 2^b A_b,
// whose weights 2^b are applied by the host's Horner chain.  Workgroup-cooperatively (msm_coop.hpp)
// all of them and S fall out of ONE binary tree of log2(K) dependent additions; this lane-serial
// form is what the CPU stepping backend runs.  Output arrays: [0] = S, [1+b] = A_b,
// [1+log2 K] = copy of S when the level's weights are l+1 (weight_off), then the older arrays.
template <class C>
struct BucketLevelBitsBody {
  typedef XyzzD<C> Pt;
  uint32_t K, lgK, weight_off, cnt, n_old;
  const uint32_t* x; const uint32_t* old_in; uint32_t* out;
  PC_HD void operator()(uint32_t lane) const {
    const uint32_t a = lane / cnt, gidx = lane % cnt;
    const size_t stride = (size_t)cnt * Pt::WORDS;
    const uint32_t nw = lgK + weight_off;
    if (a == 0) {
      const uint32_t* base = x + (size_t)gidx * K * Pt::WORDS;
      Pt S = Pt::infinity();
      for (uint32_t l = 0; l < K; l++) S.add(Pt::load(base + (size_t)l * Pt::WORDS));
      S.store(out + (size_t)gidx * Pt::WORDS);
      if (weight_off) S.store(out + (size_t)(1 + lgK) * stride + (size_t)gidx * Pt::WORDS);
      for (uint32_t b = 0; b < lgK; b++) {
        Pt A = Pt::infinity();
        for (uint32_t l = 0; l < K; l++) if ((l >> b) & 1) A.add(Pt::load(base + (size_t)l * Pt::WORDS));
        A.store(out + (size_t)(1 + b) * stride + (size_t)gidx * Pt::WORDS);
      }
    } else {
      const uint32_t* base = old_in + (size_t)(a - 1) * cnt * K * Pt::WORDS + (size_t)gidx * K * Pt::WORDS;
      Pt acc = Pt::infinity();
      for (uint32_t j = 0; j < K; j++) acc.add(Pt::load(base + (size_t)j * Pt::WORDS));
      acc.store(out + (size_t)(nw + a) * stride + (size_t)gidx * Pt::WORDS);
    }
  }
};

// All remaining seg-reduce levels, one after the other (the CPU stepping backend's version of
// HipBackend::seg_reduce_tail, which runs the same loop inside one workgroup).
template <class C, class Backend>
void seg_reduce_tail_serial(Backend& be, const MsmGeom& g, uint32_t level, uint32_t slots, uint32_t* const* pk, uint32_t* const* pp,
                            int cur, const uint32_t* offsets, uint32_t* buckets) {
  for (;;) {
    const uint32_t T2l = level == 1 ? g.T2 : g.T2b;
    uint32_t lanes2 = (slots + T2l - 1) / T2l;
    SegReduceBody<C> b{g, level, slots, pk[cur], pp[cur], offsets, buckets, pk[cur ^ 1], pp[cur ^ 1]};
    be.launch(b, lanes2);
    if (lanes2 == 1) break;
    slots = 2 * lanes2; level++; cur ^= 1;
  }
}

// Steps 1-3 with device-scope atomics (histogram, scan, cursor scatter).  Used by the CPU
// stepping backend and available on HIP (PC_HIP_SORT=atomic) as the simple reference sort.
template <class C, class Backend>
void sort_entries_atomic(Backend& be, const MsmGeom& g, const uint32_t* scalars_dev, uint32_t* hist, uint32_t* offsets,
                         uint32_t* cursor, uint32_t* entries) {
  be.memset(hist, 0, ((size_t)g.NB + 1) * 4);
  { DigitsHistBody<C> b{g, scalars_dev, hist}; be.launch(b, g.n); }
  be.mark();   // 1: digits + histogram
  be.exclusive_scan_u32(hist, offsets, (size_t)g.NB + 1);
  be.copy_d2d(cursor, offsets, ((size_t)g.NB + 1) * 4);
  be.mark();   // 2: scan
  { ScatterBody<C> b{g, scalars_dev, cursor, entries}; be.launch(b, g.n); }
  be.mark();   // 3: scatter
}

// ---------------------------------------------------------------------------------------
// Window table of a resident SRS: table[w][i] = 2^(c w) P_i for w < Wd (affine, (0,0) = infinity).
// With it digit w of scalar i adds table[w][i] into ONE bucket set shared by all windows:
//   sum_i k_i P_i = sum_{w,i} d_{w,i} (2^(c w) P_i) = sum_b b * B[b],
// so the bucket count no longer multiplies with the window count and c can grow (c = 20 at
// n = 2^20: 13 digits per scalar instead of 16, the same 2^19 buckets in total, no window fold).
// Built once per SRS (like the upload, outside any commit/open): one lane per base walks the
// doubling chain in Jacobian coordinates and normalises once per window.
// ---------------------------------------------------------------------------------------
template <class C>
struct WindowTableBody {
  static constexpr int AW = 2 * Fd<typename C::FqP>::N;
  const uint32_t* bases; uint32_t n, c, Wd; uint32_t* table;
  uint32_t stride;           // words per table entry (>= AW; 32 = one 128-byte line per 96-byte point)
  PC_HD void operator()(uint32_t i) const {
    AffD<C> p = AffD<C>::load(bases + (size_t)i * AW);
    p.store(table + (size_t)i * stride);
    JacD<C> acc = JacD<C>::infinity(); acc.add_affine(p);
    for (uint32_t w = 1; w < Wd; w++) {
      for (uint32_t k = 0; k < c; k++) acc = acc.dbl();
      acc.to_affine().store(table + ((size_t)w * n + i) * stride);
    }
  }
};

// The same table with one inversion per 16 points instead of one per point: the doubling chains of all bases advance
// window by window (Jacobian state in a scratch buffer), and each window's n points are normalised by
// JacBatchAffineBody (Montgomery's trick along a lane's run).  ~1750 instead of ~5700 field products per base at
// c = 22: 2.05 s -> 0.7 s for the 19 GB table of a 2^24-point BLS12-381 key.
template <class C>
struct TableInitBody {          // table[0][i] = P_i, state[i] = P_i (Jacobian)
  static constexpr int FN = Fd<typename C::FqP>::N, AW = 2 * FN;
  const uint32_t* bases; uint32_t* table; uint32_t stride; uint32_t* state;
  PC_HD void operator()(uint32_t i) const {
    const AffD<C> p = AffD<C>::load(bases + (size_t)i * AW);
    p.store(table + (size_t)i * stride);
    uint32_t* o = state + (size_t)i * 3 * FN;
    if (p.is_inf()) { Fd<typename C::FqP>::zero().store(o); Fd<typename C::FqP>::zero().store(o + FN); Fd<typename C::FqP>::zero().store(o + 2 * FN); }
    else { p.x.store(o); p.y.store(o + FN); Fd<typename C::FqP>::one().store(o + 2 * FN); }
  }
};
template <class C>
struct TableDoubleBody {        // state[i] = 2^c * state[i]
  typedef Fd<typename C::FqP> Fq;
  static constexpr int FN = Fq::N;
  uint32_t* state; uint32_t c;
  PC_HD void operator()(uint32_t i) const {
    uint32_t* o = state + (size_t)i * 3 * FN;
    JacD<C> a; a.X = Fq::load(o); a.Y = Fq::load(o + FN); a.Z = Fq::load(o + 2 * FN);
    for (uint32_t k = 0; k < c; k++) a = a.dbl();
    if (a.is_inf()) { Fq::zero().store(o); Fq::zero().store(o + FN); Fq::zero().store(o + 2 * FN); }
    else { a.X.store(o); a.Y.store(o + FN); a.Z.store(o + 2 * FN); }
  }
};
template <class C, class Backend>
void build_window_table_batched(Backend& be, const uint32_t* bases, uint32_t n, uint32_t c, uint32_t Wd, uint32_t* table, uint32_t stride,
                                uint32_t K = 16) {
  constexpr int FN = Fd<typename C::FqP>::N;
  if (!n) return;
  uint32_t* state = (uint32_t*)be.alloc((size_t)n * 4 * FN * 4);      // Jacobian state | prefix products
  try {
    uint32_t* scratch = state + (size_t)n * 3 * FN;
    { TableInitBody<C> b{bases, table, stride, state}; be.launch(b, n); }
    for (uint32_t w = 1; w < Wd; w++) {
      { TableDoubleBody<C> b{state, c}; be.launch(b, n); }
      JacBatchAffineBody<C> nb{state, scratch, table + (size_t)w * n * stride, n, K, stride};
      be.launch(nb, (n + K - 1) / K);
    }
    be.sync();
  } catch (...) { be.free(state); throw; }
  be.free(state);
}

// The same table for SMALL keys in one normalisation: the Jacobian states of all windows are kept (window w from window w - 1 by c
// doublings, a launch each) and ONE JacBatchAffineBody pass normalises the (Wd - 1) n points together -- one inversion chain in the
// whole build instead of one per window, and the backend's grow-only workspace instead of an allocation.  What the window-by-window
// build costs a small key is latency: 15 x (16 doublings + an inversion chain) and a hipMalloc / hipFree pair, ~5 ms for the 2^16
// points of an IPA opening's fixed key, against ~1 ms here.  Affine coordinates are canonical: the table is the same, bit for bit.
template <class C>
struct TableDoubleToBody {      // out[i] = 2^c * in[i]
  typedef Fd<typename C::FqP> Fq;
  static constexpr int FN = Fq::N;
  const uint32_t* in; uint32_t* out; uint32_t c;
  PC_HD void operator()(uint32_t i) const {
    const uint32_t* s = in + (size_t)i * 3 * FN;
    uint32_t* o = out + (size_t)i * 3 * FN;
    JacD<C> a; a.X = Fq::load(s); a.Y = Fq::load(s + FN); a.Z = Fq::load(s + 2 * FN);
    for (uint32_t k = 0; k < c; k++) a = a.dbl();
    if (a.is_inf()) { Fq::zero().store(o); Fq::zero().store(o + FN); Fq::zero().store(o + 2 * FN); }
    else { a.X.store(o); a.Y.store(o + FN); a.Z.store(o + 2 * FN); }
  }
};
static constexpr size_t TABLE_ONESHOT_MAX_POINTS = (size_t)1 << 21;      // (Wd - 1) n up to here: 268 MB of workspace for the 8-limb curves
template <class C, class Backend>
void build_window_table_oneshot(Backend& be, const uint32_t* bases, uint32_t n, uint32_t c, uint32_t Wd, uint32_t* table, uint32_t stride,
                                uint32_t K = 16) {
  constexpr int FN = Fd<typename C::FqP>::N;
  if (!n) return;
  const size_t pts = (size_t)(Wd > 1 ? Wd - 1 : 0) * n;
  uint32_t* st0 = (uint32_t*)be.workspace(((size_t)n * 3 * FN + pts * 4 * FN) * 4);      // state of window 0 | states of windows 1 .. | prefix products
  uint32_t* sts = st0 + (size_t)n * 3 * FN;
  uint32_t* scratch = sts + pts * 3 * FN;
  { TableInitBody<C> b{bases, table, stride, st0}; be.launch(b, n); }
  for (uint32_t w = 1; w < Wd; w++) {
    TableDoubleToBody<C> b{w == 1 ? st0 : sts + (size_t)(w - 2) * n * 3 * FN, sts + (size_t)(w - 1) * n * 3 * FN, c};
    be.launch(b, n);
  }
  if (pts) {
    JacBatchAffineBody<C> nb{sts, scratch, table + (size_t)n * stride, (uint32_t)pts, K, stride};
    be.launch(nb, (pts + K - 1) / K);
  }
  be.sync();
}

// window width for the table mode: n * (bits/c + 1) mixed adds against ~3 * 2^(c-1) reduction adds
inline uint32_t msm_choose_table_c(size_t n, uint32_t scalar_bits = 255, uint32_t min_top_bits = 5, bool glv = false) {
  uint32_t best = 8; double best_cost = 1e300;
  if (glv) {      // two halves of ~128 significant bits (recoded over GLV_HALF_BITS), two bucket sets
    for (uint32_t c = 8; c <= 23; c++) {
      const uint32_t Wh = GLV_HALF_BITS / c + 1;
      if (128 > (Wh - 1) * c ? 128 - (Wh - 1) * c < min_top_bits : true) continue;     // top digit of fewer than 5 (or no) significant bits
      const double cost = 2.0 * (double)n * Wh + 6.0 * (double)((size_t)1 << (c - 1));
      if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
  }
  for (uint32_t c = 8; c <= 23; c++) {
    // a top digit of fewer than 5 bits lands in <= 2^4 buckets with n / 2^4 .. n / 2 entries each: chains of hundreds of chunks
    // (c = 11, 12, 14 at 255 bits; 2^14 pairs with c = 14: 8 buckets of 2048 entries, 7 scan steps in k_accumulate).  The many-MSM
    // pass asks for no minimum: with a bucket set per sub-MSM the bucket count is what costs (Hyrax, 1024 x 1025 pairs: c = 13
    // instead of 11 is 4 x the buckets, commit 4.2 -> 5.2 ms)
    const uint32_t Wd = scalar_bits / c + 1;
    if (scalar_bits - (Wd - 1) * c < min_top_bits) continue;
    double cost = (double)n * Wd + 3.0 * (double)((size_t)1 << (c - 1));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// Many-MSM mode: after the bucket reduction every sub-MSM s holds `narr` partial sums P_a[s] with
// weights 2^exp[a] (no window fold: the table put all windows into one bucket set).  One lane per
// sub-MSM folds them with a short Horner chain (<= c doublings) -- B results, B lanes, one launch.
// `order` lists the arrays by descending exponent.
template <class C>
struct SubFoldBody {
  typedef XyzzD<C> Pt;
  const uint32_t* arrays;    // narr arrays of W points each (array a at arrays + a * W * Pt::WORDS)
  uint32_t W, narr;
  uint32_t exp[32], order[32];
  uint32_t* out;             // W XYZZ points
  PC_HD void operator()(uint32_t s) const {
    Pt acc = Pt::infinity();
    uint32_t cur = narr ? exp[order[0]] : 0;
    for (uint32_t t = 0; t < narr; t++) {
      const uint32_t a = order[t];
      for (; cur > exp[a]; cur--) acc = acc.dbl();
      acc.add(Pt::load(arrays + ((size_t)a * W + s) * Pt::WORDS));
    }
    for (; cur > 0; cur--) acc = acc.dbl();
    acc.store(out + (size_t)s * Pt::WORDS);
  }
};

// A[k] += B[k] over two bucket arrays (an MSM run in parts: every part after the first accumulates into a second bucket array,
// which is folded into the first before the one bucket reduction; MsmPlan::add_part)
template <class C>
struct BucketMergeBody {
  typedef XyzzD<C> Pt;
  uint32_t* a; const uint32_t* b;
  PC_HD void operator()(uint32_t k) const {
    const Pt y = Pt::load(b + (size_t)k * Pt::WORDS);
    if (y.is_inf()) return;
    Pt x = Pt::load(a + (size_t)k * Pt::WORDS);
    x.add(y);
    x.store(a + (size_t)k * Pt::WORDS);
  }
};

// ---------------------------------------------------------------------------------------
// Orchestration
// ---------------------------------------------------------------------------------------
struct MsmConfig {
  uint32_t c = 0;            // 0 = choose from n
  uint32_t T = 0;            // 0 = choose from n*W
  // seg-reduce chunk lengths.  Round 1 measured 4 < 8 < 64 when every workgroup edge left a partial to reduce; since the
  // edge merge (k_accumulate_edges) the list is empty for uniformly distributed scalars and the levels are launches without
  // work: chunks of 8 halve their number (skewed scalars pay chains of up to 8 instead of 4 additions per level)
  uint32_t T2 = 8;
  uint32_t T2b = 8;          // deeper levels
  uint32_t K0 = 8;           // bucket-reduce group size of the wide levels (serial chains; measured 8 < 4 << 16)
  uint32_t K1 = 256;         // group size of the later, latency-bound levels (workgroup-cooperative on HIP)
  uint32_t target_lanes = 1u << 18;
  uint32_t seg_tail_lanes = 256;         // seg-reduce levels with at most this many lanes run inside one launch
  uint32_t coop2_max_points = 1u << 15;  // cooperative levels of at most this many points (all arrays) run with two lanes per point
                                         // (msm_coop.hpp): latency-bound chains; their groups are then at most 128 points, i.e.
                                         // workgroups of 256 lanes that fit beside a resident accumulation workgroup (512-lane
                                         // groups made the pipelined 2^20 step 9 % slower: they need a whole CU to themselves)
  size_t coop2_max_entries = (size_t)1 << 22;   // ... and only in MSMs of at most this many entries (n x digits): at 2^20 pairs the
                                         // blocking MSM gains 3.7 % (3.83 -> 3.69 ms) but the pipelined step loses 1.5 % (5.61 -> 5.70 ms:
                                         // twice the lanes and one more launch compete with the next MSM's accumulation)
  uint32_t coop_max_points = 1u << 17;   // levels with more points than this use the serial fan-in K0 (cooperative levels from the
                                         // first level on -- 7 + 6 + 6 instead of 16 + 8 + 8 dependent additions at 2^19 buckets --
                                         // measured slower: bucket reduction 1.82 vs 1.55 ms on the same box; PC_HIP_COOP_MAX_LOG2)
  // Precomputed window table of a resident SRS (pc_hip_srs_precompute): calls of at least tbl_min_n
  // pairs run with window width tbl_c against tbl[w][i] = 2^(tbl_c w) P_i and ONE shared bucket set.
  const uint32_t* tbl = nullptr;
  uint32_t tbl_c = 0, tbl_stride = 0;
  uint32_t tbl_pt_stride = 0;            // words per table entry
  bool tbl_glv = false;                  // the table covers the GLV_HALF_BITS-bit halves of the scalar split (MsmGeom::glv)
  size_t tbl_min_n = 0;
  // The shared bucket set is denser where the short top digit lands (56 instead of 24 entries per
  // bucket at n = 2^20, c = 20): chunks of M / 2^18 = 52 entries would cut those buckets twice, which
  // the in-workgroup neighbour merge cannot repair (seg-reduce 0.37 -> 0.76 ms).  One round of 512
  // workgroups (chunks of 104) keeps every bucket within two chunks at the same accumulate time.
  uint32_t tbl_target_lanes = 1u << 17;
  // ... doubled (up to tbl_max_lanes) while the chunks stay at least tbl_chunk entries long.  512 workgroups are exactly ONE round of
  // the chip at two workgroups per CU: a blocking MSM then lasts as long as its slowest workgroup (accumulate 34.5 ms at 2^24); with
  // 2048 workgroups (chunks of 384) the rounds rebalance: 32.0 ms.  Only whole multiples of a round (768 or 832 workgroups: 1.5 rounds
  // of work in the time of 2), and not below ~64 entries per chunk: the joins of the cut buckets then cost more than the balance
  // gains, and the pipelined 2^20 step lost 4 % with chunks of 52.  (BN254 batch pass 94.1 -> 91.0 ms, Pallas 2^22 4.23 -> 4.13 ms;
  // the pipelined 2^24 step is unchanged: consecutive accumulations already fill each other's tails.)  0: fixed tbl_target_lanes.
  uint32_t tbl_chunk = 64;
  uint32_t tbl_max_lanes = 1u << 19;
  uint32_t tbl_K0 = 8;                  // first reduction level of the 2^(c-1)-bucket set (measured 8 < 4 << 16)
};

PC_HD uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

inline uint32_t msm_choose_c(size_t n, uint32_t scalar_bits = 255) {
  // window width: balances n*W mixed adds against ~2*W*2^(c-1) full adds of the reduction
  if (n < 32) return 3;
  uint32_t lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
  uint32_t c = lg > 4 ? lg - 4 : 2;
  if (c < 4) c = 4;
  if (c > 20) c = 20;
  if (n >= (1u << 16)) {
    // Avoid a degenerate top window.  With W = bits/c + 1 windows the last one holds only
    // bits - (W-1)*c scalar bits; when that is ~0-3 (c = 15, 17, 18 for 255-bit scalars) its few
    // buckets each receive n/8 .. n/2 entries: one workgroup of the fine sort then walks a
    // giant bin alone (measured 0.17 -> 1.3 ms at 2^20).  Take the nearest width with >= 7 top bits.
    for (uint32_t d = 0; d <= 3; d++) {
      const uint32_t cand[2] = {c - d, c + d};
      for (uint32_t k = 0; k < 2; k++) {
        const uint32_t cc = cand[k];
        if (cc < 8 || cc > 20) continue;
        const uint32_t W = scalar_bits / cc + 1, tb = scalar_bits - (W - 1) * cc;
        if (tb >= 7) return cc;
      }
    }
  }
  return c;
}

template <class C, class Backend>
class MsmPlan {
 public:
  typedef XyzzD<C> Pt;
  typedef typename C::FrP FrP;
  static constexpr int AW = 2 * Fd<typename C::FqP>::N;

  // subs == 0: one MSM of up to n_max pairs per call.  subs == B > 0: every call is B independent MSMs of
  // n_max / B pairs over the same bases (cfg.tbl must hold their window table); see enqueue().
  MsmPlan(Backend& be, size_t n_max, const MsmConfig& cfg, uint32_t subs = 0) : be_(be), cfg_(cfg), n_max_(n_max), subs_(subs) {
    if (subs_ && (!cfg_.tbl || !cfg_.tbl_c || n_max % subs_)) throw std::runtime_error("MsmPlan: many-MSM mode needs a window table");
    if (cfg_.T2 < 4) cfg_.T2 = 4;       // each level must shrink the list: 2*ceil(s/T2) < s
    if (cfg_.T2b < 4) cfg_.T2b = 4;
    // plan_geometry divides the bucket count by these and turns them into Horner exponents: powers of two only
    // (K1 / the cooperative levels: at most one workgroup of 256 lanes)
    auto pow2_floor = [](uint32_t v, uint32_t lo, uint32_t hi) { uint32_t p = lo; while (p * 2 <= v && p * 2 <= hi) p *= 2; return p; };
    cfg_.K0 = pow2_floor(cfg_.K0, 2, 1u << 12);
    cfg_.tbl_K0 = pow2_floor(cfg_.tbl_K0, 2, 1u << 12);
    cfg_.K1 = pow2_floor(cfg_.K1, 2, 256);
    if (cfg_.T2 > 4096) cfg_.T2 = 4096;
    if (cfg_.T2b > 4096) cfg_.T2b = 4096;
    min_T_ = cfg_.T ? cfg_.T : 16;
    // The window width is chosen per call from the call's n (a resident SRS serves MSMs of many
    // lengths: KZG opens, IPA halving rounds).  Size every buffer for the worst call n <= n_max.
    // The width is a step function of n (msm_choose_c: one value per power-of-two bracket, plus the n < 32 and
    // table thresholds), while the entry count n * Wd grows with n inside a bracket: evaluate both ends of every
    // bracket -- a call just below a power of two has the bracket's (smaller) c, hence more digits per
    // scalar than any power-of-two n (KZG opens have n - 1 pairs over a 2^k SRS).
    size_t NBmax = 1, Mmax = 1, red_max = 1, res_max = 1;
    std::vector<size_t> sizes;
    if (subs_) sizes.push_back(n_max);
    else {
      for (size_t p2 = 1; p2 <= n_max; p2 <<= 1) { sizes.push_back(p2); if (p2 > 1) sizes.push_back(p2 - 1); if (p2 > (n_max >> 1)) break; }
      sizes.push_back(31); sizes.push_back(32); sizes.push_back(n_max);
      if (cfg_.tbl && cfg_.tbl_min_n) { sizes.push_back(cfg_.tbl_min_n); if (cfg_.tbl_min_n > 1) sizes.push_back(cfg_.tbl_min_n - 1); }
    }
    for (size_t n : sizes) {
      if (n < 1 || n > n_max) continue;
      plan_geometry(n);
      NBmax = std::max<size_t>(NBmax, g_.NB);
      Mmax = std::max<size_t>(Mmax, n * (size_t)g_.Wd);
      size_t total = 0;
      for (uint32_t l = 0; l < n_levels_; l++) total += (size_t)(1 + lvl_narr_[l]) * g_.W * lvl_m_[l];
      red_max = std::max(red_max, total);
      res_max = std::max<size_t>(res_max, (size_t)g_.W * (n_levels_ ? lvl_narr_[n_levels_ - 1] : 1));
    }
    entries_cap_ = Mmax; nb_cap_ = NBmax;
    // plans that never see more than 2^17 entries (keys of up to ~2^12 points) are pure latency: chunks of 8 halve the
    // accumulation's dependent chain, and the in-workgroup scan of k_accumulate joins the longer chains of cut buckets in one
    // more step (blocking MSM of 2^12 pairs 0.79 -> 0.70 ms, 2^8 0.64 -> 0.57; slower from 2^14 on, where the lanes fill the GPU)
    if (!cfg_.T && Mmax <= ((size_t)1 << 17)) min_T_ = 8;
    hist_ = offsets_ = cursor_ = entries_ = buckets_ = scalars_ = red_ = nullptr; pk_[0] = pk_[1] = pp_[0] = pp_[1] = nullptr;
    try {
    hist_ = (uint32_t*)be_.alloc((NBmax + 1) * 4);
    offsets_ = (uint32_t*)be_.alloc((NBmax + 1) * 4);
    cursor_ = (uint32_t*)be_.alloc((NBmax + 1) * 4);
    entries_ = (uint32_t*)be_.alloc(Mmax * 4);
    buckets_ = (uint32_t*)be_.alloc(NBmax * Pt::WORDS * 4);
    scalars_ = (uint32_t*)be_.alloc((n_max ? n_max : 1) * (size_t)FrP::N * 4);
    // partial levels
    // chunk length is max(16, M / target_lanes) per call (or the fixed cfg.T): at most this many lanes
    size_t lanes0 = ceil_div_u32(Mmax, min_T_);
    if (!cfg_.T) {
      // T = clamp(floor(M / target_lanes), 16, 4096)  =>  lanes = ceil(M / T) <= max(17/16 target_lanes + 2, M / 4096 + 1)
      const size_t tl = std::max(cfg_.target_lanes, cfg_.tbl ? (cfg_.tbl_chunk ? std::max(cfg_.tbl_max_lanes, cfg_.tbl_target_lanes) : cfg_.tbl_target_lanes) : 0u);
      size_t bound = std::max<size_t>(tl * 17 / 16 + 2, Mmax / 4096 + 1);
      if (lanes0 > bound) lanes0 = bound;
    }
    size_t slots = 2 * lanes0;
    part_slots_ = slots;
    for (int i = 0; i < 2; i++) {
      pk_[i] = (uint32_t*)be_.alloc(slots * 4);
      pp_[i] = (uint32_t*)be_.alloc(slots * (size_t)Pt::WORDS * 4);
      slots = 2 * (size_t)ceil_div_u32(slots, cfg_.T2);
    }
    red_ = (uint32_t*)be_.alloc((red_max + 2 * (size_t)subs_) * (size_t)Pt::WORDS * 4);       // + the folded results of every bucket set
    result_host_ = (uint32_t*)be_.alloc_host(std::max<size_t>(res_max, 2 * (size_t)subs_) * Pt::WORDS * 4);
    red_points_ = red_max;
    } catch (...) { release(); throw; }   // a failed hipMalloc must not leak the earlier buffers
    plan_geometry(n_max);
  }
  ~MsmPlan() { release(); }
  void release() {
    void* ps[] = {hist_, offsets_, cursor_, entries_, buckets_, scalars_, pk_[0], pk_[1], pp_[0], pp_[1], red_, entries2_, offsets2_, buckets2_, hist2_, cursor2_};
    entries2_ = offsets2_ = buckets2_ = hist2_ = cursor2_ = nullptr;
    for (void* p : ps) be_.free(p);
    be_.free_host(result_host_);
    be_.free(tab_); be_.free_host(tab_host_); tab_ = nullptr; tab_host_ = nullptr;
    hist_ = offsets_ = cursor_ = entries_ = buckets_ = scalars_ = red_ = nullptr; pk_[0] = pk_[1] = pp_[0] = pp_[1] = nullptr;
    result_host_ = nullptr;
  }

  // What a host call in parts added on first use (second sort output, offsets, bucket array: ~1.2 GB on a 2^24 BLS12-381 key) goes back;
  // begin_parts re-creates it.  Only on an idle plan (the caller synchronised the pipeline).
  void trim() {
    if (in_parts_) return;
    void* ps[] = {entries2_, offsets2_, buckets2_, hist2_, cursor2_};
    entries2_ = offsets2_ = buckets2_ = hist2_ = cursor2_ = nullptr;
    for (void* p : ps) be_.free(p);
  }

  const MsmGeom& geom() const { return g_; }
  uint32_t* scalar_staging() { return scalars_; }
  const MsmGeom& last_geom() const { return g_; }

  // bases_dev: resident SRS; scalars_dev: n x FrP::N words on the device.
  // Writes the affine result (AW words, Montgomery; (0,0) = infinity) to out_host.
  void run(const uint32_t* bases_dev, uint32_t base_off, const uint32_t* scalars_dev, size_t n, bool from_mont,
           uint32_t* out_host) {
    enqueue(bases_dev, base_off, scalars_dev, n, from_mont);
    finish(out_host);
  }

  // All device work of one MSM plus the asynchronous download of the <= W*levels partial
  // sums; returns as soon as everything is queued on the backend's stream.
  // many-MSM mode with the scalar vectors in separate device buffers: `count` <= subs vectors of m scalars each
  // (sub-MSM k reads ptrs[k]); the bucket sets of the missing ones stay empty.
  void enqueue_vectors(const uint32_t* bases_dev, uint32_t base_off, const uint64_t* ptrs_host, size_t count, size_t m, bool from_mont) {
    if (!subs_ || count > subs_ || count * m > n_max_) throw MsmCapacityError("MsmPlan: batch exceeds the plan");
    if (!tab_) { tab_ = (uint64_t*)be_.alloc((size_t)subs_ * 8); tab_host_ = (uint64_t*)be_.alloc_host((size_t)subs_ * 8); }
    for (size_t k = 0; k < count; k++) tab_host_[k] = ptrs_host[k];
    be_.copy_h2d(tab_, tab_host_, count * 8);
    enqueue(bases_dev, base_off, nullptr, count * m, from_mont, m, tab_);
  }

  void enqueue(const uint32_t* bases_dev, uint32_t base_off, const uint32_t* scalars_dev, size_t n, bool from_mont,
               size_t m_sub = 0, const uint64_t* scalar_tab = nullptr) {
    pending_empty_ = (n == 0);
    if (n == 0) return;
    plan_geometry(n);
    parts_k_ = 0; in_parts_ = false;
    part_body(bases_dev, base_off, scalars_dev, n, from_mont, m_sub, scalar_tab, /*first=*/true, /*last=*/true);
    reduce_and_download();
  }

  // ---- one MSM in PARTS (host scalars: a blocking call cannot hide the PCIe copy of its scalars behind anything but its own work) ----
  // The MSM is a sum over index ranges: part k sorts and accumulates scalars [first_k, first_k + n_k) against bases base_off + first_k ..
  // with the geometry (window width, buckets, reduction plan) of the WHOLE call; every part after the first accumulates into a second
  // bucket array that is folded into the first one (BucketMergeBody: 2^(c-1) additions), and ONE bucket reduction + host tail close
  // the call.  The sort of part k + 1 (and, in the runner, the copy of its scalars) runs on the backend's auxiliary queue beside the
  // accumulation of part k (sort outputs are double-buffered), so what a blocking call adds to the resident MSM is the copy and sort
  // of its FIRST part and the merges -- not the whole copy (round 4: two half-size MSMs on two pipelines, each with its own bucket
  // reduction and tail: the second half's copy hidden, 47.8 ms for a commit of 2^24 host coefficients against 38.3 ms resident).
  void begin_parts(size_t n_total) {
    if (subs_) throw std::runtime_error("MsmPlan: parts are for single MSMs");
    pending_empty_ = (n_total == 0);
    if (n_total == 0) return;
    if (n_total > n_max_) throw MsmCapacityError("MsmPlan: call exceeds the workspace this plan was sized for");
    plan_geometry(n_total);
    parts_k_ = 0; in_parts_ = true; parts_n_ = 0;
    if (!entries2_) {        // second sort output + second bucket array, on first use (freed with the plan)
      entries2_ = (uint32_t*)be_.alloc(entries_cap_ * 4);          // (full size: any weights of the parts)
      offsets2_ = (uint32_t*)be_.alloc((nb_cap_ + 1) * 4);
      buckets2_ = (uint32_t*)be_.alloc(nb_cap_ * Pt::WORDS * 4);
      hist2_ = (uint32_t*)be_.alloc((nb_cap_ + 1) * 4);
      cursor2_ = (uint32_t*)be_.alloc((nb_cap_ + 1) * 4);
    }
    set_tok_[0] = set_tok_[1] = -1;
  }
  // scalars_dev: the part's n scalars on the device; ready_tok: a backend token (aux_token / -1) after which they may be read
  void add_part(const uint32_t* bases_dev, uint32_t base_off, const uint32_t* scalars_dev, size_t n, bool from_mont, int ready_tok, bool last) {
    if (pending_empty_ || n == 0) { if (last && !pending_empty_) { if (parts_k_ == 0) { pending_empty_ = true; return; } reduce_and_download(); } return; }
    part_body(bases_dev, base_off, scalars_dev, n, from_mont, 0, nullptr, parts_k_ == 0, last, ready_tok);
    parts_n_ += n;
    if (last) reduce_and_download();
  }

 private:
  void part_body(const uint32_t* bases_dev, uint32_t base_off, const uint32_t* scalars_dev, size_t n, bool from_mont, size_t m_sub,
                 const uint64_t* scalar_tab, bool first, bool last, int ready_tok = -1) {
    MsmGeom g = g_;
    g.n = (uint32_t)n; g.base_off = base_off; g.from_mont = from_mont ? 1 : 0;
    g.m_sub = subs_ ? (uint32_t)(m_sub ? m_sub : n / subs_) : 0u;
    g.scalar_tab = scalar_tab;
    const size_t Mmax = n * g.Wd;
    uint32_t want_lanes = g.tbl_stride ? cfg_.tbl_target_lanes : cfg_.target_lanes;
    if (g.tbl_stride && cfg_.tbl_chunk)
      while ((uint64_t)want_lanes * 2 <= cfg_.tbl_max_lanes && Mmax / ((size_t)want_lanes * 2) >= cfg_.tbl_chunk) want_lanes *= 2;
    // ceil: at most want_lanes lanes, i.e. whole rounds of the chip.  (The floor of round 4 gave a call of 2^24 - 1 pairs -- every KZG open --
    // chunks of 383 instead of 384 entries and 2054 workgroups instead of 2048: a fifth, nearly empty round, accumulate 36.0 instead of 32.0 ms)
    uint32_t T = cfg_.T ? cfg_.T : (uint32_t)((Mmax + want_lanes - 1) / want_lanes);
    if (T < min_T_) T = min_T_;
    if (T > 4096) T = 4096;
    g.T = T; g.T2 = cfg_.T2; g.T2b = cfg_.T2b;
    // capacity checks BEFORE anything is launched (the buffers were sized in the constructor for every n <= n_max)
    const size_t lanes = ceil_div_u32(Mmax, T);
    if (n > n_max_ || Mmax > entries_cap_ || g.NB > nb_cap_ || 2 * lanes > part_slots_)
      throw MsmCapacityError("MsmPlan: call exceeds the workspace this plan was sized for");
    // sort outputs: part k of a call in parts uses set k & 1, and sorts on the auxiliary queue once the accumulation that last read
    // that set is through; the accumulation then waits for the sort
    const int set = in_parts_ ? (int)(parts_k_ & 1u) : 0;
    uint32_t* entries = set ? entries2_ : entries_; uint32_t* offsets = set ? offsets2_ : offsets_;
    uint32_t* hist = set ? hist2_ : hist_; uint32_t* cursor = set ? cursor2_ : cursor_;
    uint32_t* buckets = (in_parts_ && !first) ? buckets2_ : buckets_;
    const bool marks = !in_parts_ || first;              // phase marks: the sort of the first part; accumulate / seg-reduce marks of the last
    be_.memset(buckets, 0, (size_t)g.NB * Pt::WORDS * 4);
    if (in_parts_) {
      int tok = -1;
      {
        // (a throw inside -- a failed workspace allocation -- must not leave the backend on its auxiliary queue)
        struct AuxScope { Backend& b; bool marks_was; int* tok; ~AuxScope() { b.timing_marks(marks_was); try { *tok = b.aux_end(); } catch (...) { *tok = -1; } } };
        be_.aux_begin(set_tok_[set], ready_tok);
        AuxScope scope{be_, be_.timing_marks(marks), &tok};
        be_.template sort_entries<C>(g, scalars_dev, hist, offsets, cursor, entries);
      }
      be_.wait_token(tok);
    } else {
      // steps 1-3: entries grouped by bucket + CSR offsets (backend chooses the sort)
      be_.template sort_entries<C>(g, scalars_dev, hist, offsets, cursor, entries);
    }

    { AccumulateBody<C> b{g, g.tbl_stride ? cfg_.tbl : bases_dev, entries, offsets, buckets, pk_[0], pp_[0]}; be_.template accumulate<C>(b, lanes); }
    if (!in_parts_ || last) be_.mark();   // 4: accumulate
    // the reductions below are latency-bound: they go to the pipeline's low-priority queue (HIP backend)
    // (only while the reductions are a sizeable share of the MSM: 20-30 % faster steps up to 2^18, 10-15 % at
    // 2^20, but 8 % slower at 2^21 and beyond, where a delayed reduction stalls the caller's pipeline)
    const bool tail_queue = !in_parts_ && n <= ((size_t)3 << 19);
    // (a throw in the launches below must not leave the backend on its low-priority queue: the guard hands the open tail over to
    // reduce_and_download -- tail_open_ -- only once they are all queued)
    struct TailGuard { Backend& b; bool armed; ~TailGuard() { if (armed) b.end_tail(); } } tail_guard{be_, false};
    if (tail_queue) { be_.begin_tail(); tail_guard.armed = true; }
    size_t slots = 2 * lanes; uint32_t level = 1; int cur = 0;
    for (;;) {
      size_t lanes2 = ceil_div_u32(slots, level == 1 ? g.T2 : g.T2b);
      if (lanes2 <= (cfg_.seg_tail_lanes ? cfg_.seg_tail_lanes : 1u)) {   // lanes2 == 1 always ends the walk
        // the remaining levels are tiny: one workgroup walks them all in a single launch
        be_.template seg_reduce_tail<C>(g, level, (uint32_t)slots, pk_, pp_, cur, offsets, buckets);
        break;
      }
      SegReduceBody<C> b{g, level, (uint32_t)slots, pk_[cur], pp_[cur], offsets, buckets, pk_[cur ^ 1], pp_[cur ^ 1]};
      be_.launch(b, lanes2);
      slots = 2 * lanes2; level++; cur ^= 1;
    }
    if (in_parts_) {
      if (!first) { BucketMergeBody<C> m{buckets_, buckets2_}; be_.launch(m, g.NB); }
      set_tok_[set] = be_.main_token();
      parts_k_++;
    }
    last_g_ = g;
    tail_open_ = tail_queue; tail_guard.armed = false;
  }

  void reduce_and_download() {
    const MsmGeom& g = last_g_;
    struct TailScope { Backend& b; bool on; ~TailScope() { if (on) b.end_tail(); } } tail_scope{be_, tail_open_};
    tail_open_ = false;
    be_.mark();   // 5: segmented reduction of partials
    // bucket reduction
    // layout of red_: per level l: [S_l][this level's weighted arrays][older arrays, folded], each W * lvl_m_[l] points
    const uint32_t* x = buckets_;
    uint32_t* lvl_base = red_;
    uint32_t* prev_base = nullptr;
    for (uint32_t l = 0; l < n_levels_; l++) {
      const uint32_t K = lvl_K_[l], m_out = lvl_m_[l];
      const size_t cnt = (size_t)g.W * m_out;           // groups at this level
      const size_t stride = cnt * Pt::WORDS;
      const uint32_t n_old = l ? lvl_narr_[l - 1] : 0;  // previous level's arrays after its S array, contiguous
      const uint32_t* old_in = l ? prev_base + (size_t)g.W * lvl_m_[l - 1] * Pt::WORDS : nullptr;
      be_.template bucket_level<C>(K, l == 0 ? 1u : 0u, (uint32_t)cnt, n_old, (int)lvl_bits_[l], x, old_in, lvl_base);
      x = lvl_base; prev_base = lvl_base; lvl_base += (size_t)(1 + lvl_narr_[l]) * stride;
    }
    be_.mark();   // 6: bucket reduction
    // download: the last level holds W points per array, S first
    if (subs_) {
      // many-MSM mode: fold each sub-MSM's few weighted sums on the device, download one XYZZ point each
      SubFoldBody<C> f;
      f.arrays = n_levels_ ? prev_base + (size_t)g.W * Pt::WORDS : buckets_;
      f.W = g.W; f.narr = n_levels_ ? (uint32_t)arr_exp_.size() : 1u;
      if (f.narr > 32) throw std::runtime_error("MsmPlan: too many reduction arrays");
      for (uint32_t a = 0; a < f.narr; a++) { f.exp[a] = n_levels_ ? arr_exp_[a] : 0u; f.order[a] = a; }
      std::stable_sort(f.order, f.order + f.narr, [&](uint32_t x, uint32_t y) { return f.exp[x] > f.exp[y]; });
      f.out = red_ + red_points_ * Pt::WORDS;
      be_.launch(f, g.W);
      be_.copy_d2h_async(result_host_, f.out, (size_t)g.W * Pt::WORDS * 4);
    }
    else if (n_levels_ == 0) be_.copy_d2h_async(result_host_, buckets_, (size_t)g.W * Pt::WORDS * 4);
    else be_.copy_d2h_async(result_host_, prev_base + (size_t)g.W * Pt::WORDS, (size_t)g.W * lvl_narr_[n_levels_ - 1] * Pt::WORDS * 4);
    be_.record_done();
    in_parts_ = false;
  }

 public:
  // Host-side state of enqueue() for a call whose device work is replayed from a captured graph (same n as the captured
  // call): what finish() and its Horner fold read.
  void prepare_replay(size_t n) { pending_empty_ = (n == 0); if (n) plan_geometry(n); }

  // Wait for the queued MSM and fold its partial sums on the host (Horner) into one affine point.
  void finish(uint32_t* out_host) {
    if (subs_) {      // subs_ affine points: batch-normalise the folded XYZZ results (one inversion in all)
      if (pending_empty_) { for (size_t i = 0; i < (size_t)subs_ * AW; i++) out_host[i] = 0; return; }
      be_.wait_done();
      if (g_.glv) {      // sub-MSM k = set 2k + phi(set 2k + 1)
        typedef host64::Xyzz64<C> P64;
        std::vector<uint32_t> sum((size_t)subs_ * Pt::WORDS);
        for (uint32_t k = 0; k < subs_; k++) {
          P64 a = P64::load(result_host_ + (size_t)(2 * k) * Pt::WORDS), b = P64::load(result_host_ + (size_t)(2 * k + 1) * Pt::WORDS);
          b.X = b.X.mul(P64::Fq::load(GlvOf<C>::T::BETA_MONT));
          a.add(b);
          a.store(&sum[(size_t)k * Pt::WORDS]);
        }
        host64::batch_to_affine<C>(sum.data(), subs_, out_host);
        return;
      }
      host64::batch_to_affine<C>(result_host_, subs_, out_host);
      return;
    }
    if (pending_empty_) { for (int i = 0; i < AW; i++) out_host[i] = 0; return; }
    if (in_parts_) {
      // a call in parts that never reached its last part (a failure between two parts): there is no result to wait for -- result_host_
      // is the previous call's -- and work of the parts that were queued may still run on the main and auxiliary queues
      be_.quiesce();
      in_parts_ = false; tail_open_ = false;
      throw std::runtime_error("MsmPlan: a call in parts was abandoned before its last part");
    }
    be_.wait_done();
    host_tail(out_host);
  }

 private:
  // window width, bucket counts and the reduction-level plan for a call of n pairs
  void plan_geometry(size_t n) {
    const bool tbl = subs_ || (cfg_.tbl && cfg_.tbl_c && n >= cfg_.tbl_min_n);
    uint32_t c = tbl ? cfg_.tbl_c : cfg_.c ? cfg_.c : msm_choose_c(n, FrP::BITS);
    if (c < 2 || c > 24) throw std::runtime_error("MsmPlan: window width out of range (2..24)");   // ScalarDigits' shift register, the sort passes
    const bool glv = tbl && cfg_.tbl_glv;
    g_.glv = glv ? 1u : 0u;
    g_.c = c; g_.Wd = glv ? 2 * msm_num_windows(GLV_HALF_BITS, c) : msm_num_windows(FrP::BITS, c);
    g_.W = subs_ ? subs_ * (glv ? 2u : 1u) : tbl ? (glv ? 2u : 1u) : g_.Wd; g_.tbl_stride = tbl ? cfg_.tbl_stride : 0u; g_.m_sub = 0;
    g_.nb_win = 1u << (c - 1); g_.NB = g_.W * g_.nb_win;
    g_.pt_stride = tbl ? cfg_.tbl_pt_stride : (uint32_t)AW;
    g_.n = (uint32_t)n; g_.base_off = 0; g_.from_mont = 0; g_.T = 0; g_.T2 = cfg_.T2; g_.T2b = cfg_.T2b; g_.scalar_tab = nullptr;
    uint32_t m = g_.nb_win; n_levels_ = 0;
    uint32_t kbits = 0; arr_exp_.clear();
    while (m > 1) {
      // wide levels are throughput-bound: short serial chains (K0).  Once a level holds few enough
      // points the chain length is all that matters: workgroup-cooperative "bits" levels (K1).
      uint32_t K;
      // points this level reads: groups x K x arrays (its weighted array and the older plain ones)
      const size_t level_pts = (size_t)n * g_.Wd <= cfg_.coop2_max_entries ? (size_t)g_.W * m * (1 + (n_levels_ ? lvl_narr_[n_levels_ - 1] : 0))
                                                                           : (size_t)-1;      // (no two-lane levels in large MSMs)
      if ((size_t)g_.W * m > cfg_.coop_max_points) K = tbl ? cfg_.tbl_K0 : cfg_.K0;
      else {
        // the remaining log2(m) bits are split evenly over the fewest cooperative levels of at most log2(K1) bits each:
        // the chain is one addition per bit, so 19 bits cost 7 + 6 + 6 dependent additions, not 8 + 8 + a serial tail
        uint32_t rem = 0; while ((1u << rem) < m) rem++;
        uint32_t lg1 = 0; while ((1u << lg1) < cfg_.K1) lg1++;
        if (level_pts <= cfg_.coop2_max_points && lg1 > 7) lg1 = 7;      // two lanes per point: groups of at most 128
        const uint32_t nl = (rem + lg1 - 1) / lg1;
        K = 1u << ((rem + nl - 1) / nl);
      }
      if (K > m) K = m;
      uint32_t lgK = 0; while ((1u << lgK) < K) lgK++;
      const bool bits = K >= 16;
      const uint32_t woff = n_levels_ == 0 ? 1u : 0u;
      const uint32_t nw = bits ? lgK + woff : 1u;
      lvl_K_[n_levels_] = K; m /= K; lvl_m_[n_levels_] = m;   // m = elements per window AFTER this level
      lvl_bits_[n_levels_] = !bits ? 0u : (K <= 128 && level_pts <= cfg_.coop2_max_points) ? 2u : 1u;
      lvl_narr_[n_levels_] = nw + (n_levels_ ? lvl_narr_[n_levels_ - 1] : 0);
      // weights of the new arrays (as powers of two), then the older arrays keep theirs
      std::vector<uint32_t> e;
      if (bits) { for (uint32_t b = 0; b < lgK; b++) e.push_back(kbits + b); if (woff) e.push_back(kbits); }
      else e.push_back(kbits);
      e.insert(e.end(), arr_exp_.begin(), arr_exp_.end());
      arr_exp_.swap(e);
      kbits += lgK;
      n_levels_++;
    }
  }

  // Horner over (level, window) on the host.  P_j[w] has weight 2^(c*w + k_0 + ... + k_{j-1}).
  void host_tail(uint32_t* out_host) {
    const uint32_t L = n_levels_, W = g_.W;
    // classic mode: bucket set w is window w, weight 2^(c w); table mode: the sets share the weights (one set, or the two of the
    // GLV split: the points of set 1 go through phi -- X * beta in XYZZ coordinates -- before the one Horner chain)
    const uint32_t cw = g_.tbl_stride ? 0u : g_.c;
    const size_t narr = L == 0 ? 1 : arr_exp_.size();
    if (g_.glv) {
      typedef host64::Xyzz64<C> P64;
      for (size_t a = 0; a < narr; a++) {
        uint32_t* p = &result_host_[(a * W + 1) * Pt::WORDS];
        P64::Fq::load(p).mul(P64::Fq::load(GlvOf<C>::T::BETA_MONT)).store(p);
      }
    }
    std::vector<host64::WeightedPoint> items;
    if (L == 0) {   // c == 1: one bucket per window, weight 1
      for (uint32_t w = 0; w < W; w++) items.push_back({cw * w, &result_host_[(size_t)w * Pt::WORDS]});
    } else {
      // arrays after S of the last level, weights 2^(c*w + arr_exp_[a])
      for (size_t a = 0; a < arr_exp_.size(); a++)
        for (uint32_t w = 0; w < W; w++)
          items.push_back({cw * w + arr_exp_[a], &result_host_[(a * W + w) * Pt::WORDS]});
    }
    host64::horner_to_affine<C>(items, out_host);
  }

  Backend& be_;
  MsmConfig cfg_;
  size_t n_max_;
  uint32_t subs_ = 0;
  size_t red_points_ = 0;
  MsmGeom g_;
  uint32_t min_T_;
  size_t part_slots_ = 0, entries_cap_ = 0, nb_cap_ = 0;
  uint32_t *hist_, *offsets_, *cursor_, *entries_, *buckets_, *scalars_, *red_;
  uint32_t* pk_[2]; uint32_t* pp_[2];
  uint32_t n_levels_; uint32_t lvl_K_[32]; uint32_t lvl_m_[32]; uint32_t lvl_bits_[32]; uint32_t lvl_narr_[32];
  std::vector<uint32_t> arr_exp_;   // log2 weight of every array after S at the last level
  uint32_t* result_host_ = nullptr;   // pinned
  uint64_t* tab_ = nullptr; uint64_t* tab_host_ = nullptr;   // scalar-vector addresses of enqueue_vectors (device / pinned)
  bool pending_empty_ = true;
  // a call in parts (begin_parts / add_part): second sort outputs and bucket array, the part counter, backend tokens of the last
  // accumulation that read each sort-output set
  uint32_t *entries2_ = nullptr, *offsets2_ = nullptr, *buckets2_ = nullptr, *hist2_ = nullptr, *cursor2_ = nullptr;
  uint32_t parts_k_ = 0; size_t parts_n_ = 0; bool in_parts_ = false, tail_open_ = false;
  int set_tok_[2] = {-1, -1};
  MsmGeom last_g_;
};

}  // namespace pc
