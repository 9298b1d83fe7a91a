// Prime-field arithmetic for gfx950: Montgomery residues held as 32-bit limbs so that every
// partial product is one v_mad_u64_u32 (32x32+64 -> 64).  Template parameter P is one of
// the generated pc_<field> structs in field_constants.h (N limbs, MOD, ONE, R2, INV).
//
// The byte layout of an element (N little-endian 32-bit limbs, Montgomery form, R = 2^(32N))
// is identical to arkworks' Fp<MontBackend<_, N/2>> ([u64; N/2] little-endian limbs), so SRS
// points and polynomial coefficients cross the C ABI without conversion
// (reference call sites: poly-commit/src/kzg10/mod.rs:175-178, :463-470).
//
// Everything is PC_HD (host + device) so that the same arithmetic is unit-tested on the CPU
// (tests/emu) against the independent 64-bit-limb oracle before it ever runs on a GPU.
#pragma once
#include <stdint.h>
#include "field_constants.h"

// tuning experiment switch (poly_commit_amd/build.py, PC_HIP_CXXFLAGS): 0 keeps the 8-limb fields on the canonical path
#ifndef PC_LAZY_N8
#define PC_LAZY_N8 1
#endif
// tuning experiment switch: 1 closes a Montgomery column with four simple instructions where p = 1 (mod 2^32) (close_column below).
// Measured slower (round 5, profiles/EXPERIMENTS.md): the pair v_mul_lo_u32 + v_mad_u64_u32 it replaces costs less than the five
// instructions the compiler makes of it (BLS12-381 Fr NTT 5.20 -> 5.31 ms, Pallas Fq product 157 -> 154 G/s): off.
#ifndef PC_P0_ONE
#define PC_P0_ONE 0
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PC_HD __host__ __device__ __forceinline__
#define PC_D __device__ __forceinline__
#else
#define PC_HD inline
#define PC_D inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PC_UNROLL _Pragma("unroll")
#else
#define PC_UNROLL
#endif

namespace pc {

template <class P>
struct Fd {
  static constexpr int N = P::N;
  uint32_t l[N];

  static PC_HD Fd zero() { Fd r; PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
  static PC_HD Fd one() { Fd r; PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = P::ONE[i]; return r; }
  static PC_HD Fd load(const uint32_t* p) { Fd r; PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = p[i]; return r; }
  PC_HD void store(uint32_t* p) const { PC_UNROLL for (int i = 0; i < N; i++) p[i] = l[i]; }

  PC_HD bool is_zero() const { uint32_t a = 0; PC_UNROLL for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  PC_HD bool eq(const Fd& o) const { uint32_t a = 0; PC_UNROLL for (int i = 0; i < N; i++) a |= l[i] ^ o.l[i]; return a == 0; }

  // r = a - MOD if a >= MOD else a   (a < 2*MOD, possibly with an extra carry word `hi`)
  // Carry chains: on the device clang's add/sub-with-carry builtins lower to one v_addc / v_subb per limb
  // (the portable 64-bit formulation costs ~5 instructions per limb there); the host keeps the portable form.
#if defined(__HIP_DEVICE_COMPILE__)
  static __device__ __forceinline__ uint32_t adc(uint32_t a, uint32_t b, uint32_t& c) { unsigned co; uint32_t r = __builtin_addc(a, b, c, &co); c = co; return r; }
  static __device__ __forceinline__ uint32_t sbb(uint32_t a, uint32_t b, uint32_t& br) { unsigned bo; uint32_t r = __builtin_subc(a, b, br, &bo); br = bo; return r; }
#else
  static inline uint32_t adc(uint32_t a, uint32_t b, uint32_t& c) { uint64_t t = (uint64_t)a + b + c; c = (uint32_t)(t >> 32); return (uint32_t)t; }
  static inline uint32_t sbb(uint32_t a, uint32_t b, uint32_t& br) { uint64_t t = (uint64_t)a - b - br; br = (uint32_t)(t >> 63); return (uint32_t)t; }
#endif
  static PC_HD void cond_sub(uint32_t* a, uint32_t hi) {
    uint32_t d[N];
    uint32_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) d[i] = sbb(a[i], P::MOD[i], br);
    // a >= MOD  <=>  no final borrow, or the carry word absorbs it
    bool ge = (hi != 0) || (br == 0);
    PC_UNROLL for (int i = 0; i < N; i++) a[i] = ge ? d[i] : a[i];
  }

  PC_HD Fd add(const Fd& o) const {
    Fd r; uint32_t c = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = adc(l[i], o.l[i], c);
    cond_sub(r.l, c);   // every modulus here leaves >= 1 spare top bit, so c == 0
    return r;
  }
  PC_HD Fd sub(const Fd& o) const {
    Fd r; uint32_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = sbb(l[i], o.l[i], br);
    uint32_t mask = (uint32_t)0 - br;   // add MOD back on borrow
    uint32_t c = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = adc(r.l[i], P::MOD[i] & mask, c);
    return r;
  }
  PC_HD Fd dbl() const { return add(*this); }
  PC_HD Fd neg() const {
    // MOD - a, or 0 for a == 0
    Fd r; uint32_t br = 0, nz = 0;
    PC_UNROLL for (int i = 0; i < N; i++) { r.l[i] = sbb(P::MOD[i], l[i], br); nz |= l[i]; }
    uint32_t mask = nz ? 0xffffffffu : 0u;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] &= mask;
    return r;
  }

  // Montgomery product r = a * b * R^-1 mod p.
  //
  // Device: product scanning (FIPS).  Column k gathers every a_i*b_{k-i} and m_i*p_{k-i} into a
  // 96-bit accumulator; each partial product is exactly two VALU instructions,
  //     v_mad_u64_u32  acc[0:1], vcc, a, b, acc[0:1]     ; 32x32+64 with carry-out
  //     v_addc_co_u32  acc2, vcc, 0, acc2, vcc           ; fold the carry into the top word
  // (hipcc's own lowering of the C expression spends ~2.5 v_mov per product on zero-extension
  // and register-pair alignment; 1350 -> ~720 instructions per 381-bit product).
  // Host (tests, the Horner tail): portable CIOS.
#if defined(__HIP_DEVICE_COMPILE__)
#define PC_MAC1(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
  // first product of a column: the carry word is written, not accumulated (0 + 0 + carry), so it needs
  // no zero-initialised register
#define PC_MAC1F(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc\n\t"
  // k partial products per asm statement (hipcc pads every statement with an s_nop).  FIRST: the
  // statement opens a column (`hi` is an output only).
#define PC_MAC_FNS(NAME, M0, HI, YC)                                                                                    \
  static __device__ __forceinline__ void NAME##1(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y) {   \
    asm(M0("%2", "%3") : "+v"(acc), HI(hi) : "v"(x[0]), YC(y[0]) : "vcc");                                               \
  }                                                                                                                     \
  static __device__ __forceinline__ void NAME##2(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y) {   \
    asm(M0("%2", "%3") PC_MAC1("%4", "%5")                                                                               \
        : "+v"(acc), HI(hi) : "v"(x[0]), YC(y[0]), "v"(x[1]), YC(y[1]) : "vcc");                                         \
  }                                                                                                                     \
  static __device__ __forceinline__ void NAME##4(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y) {   \
    asm(M0("%2", "%3") PC_MAC1("%4", "%5") PC_MAC1("%6", "%7") PC_MAC1("%8", "%9")                                       \
        : "+v"(acc), HI(hi)                                                                                              \
        : "v"(x[0]), YC(y[0]), "v"(x[1]), YC(y[1]), "v"(x[2]), YC(y[2]), "v"(x[3]), YC(y[3]) : "vcc");                   \
  }                                                                                                                     \
  static __device__ __forceinline__ void NAME##8(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y) {   \
    asm(M0("%2", "%3") PC_MAC1("%4", "%5") PC_MAC1("%6", "%7") PC_MAC1("%8", "%9")                                       \
        PC_MAC1("%10", "%11") PC_MAC1("%12", "%13") PC_MAC1("%14", "%15") PC_MAC1("%16", "%17")                          \
        : "+v"(acc), HI(hi)                                                                                              \
        : "v"(x[0]), YC(y[0]), "v"(x[1]), YC(y[1]), "v"(x[2]), YC(y[2]), "v"(x[3]), YC(y[3]),                            \
          "v"(x[4]), YC(y[4]), "v"(x[5]), YC(y[5]), "v"(x[6]), YC(y[6]), "v"(x[7]), YC(y[7]) : "vcc");                   \
  }
#define PC_HI_INOUT(h) "+v"(h)
#define PC_HI_OUT(h) "=&v"(h)
#define PC_Y_VGPR(v) "v"(v)
#define PC_Y_SGPR(v) "s"(v)
  PC_MAC_FNS(mac, PC_MAC1, PC_HI_INOUT, PC_Y_VGPR)
  PC_MAC_FNS(macf, PC_MAC1F, PC_HI_OUT, PC_Y_VGPR)
  // the same with the second factor in a scalar register: the modulus limbs of the reduction terms m_i * p_j are
  // constants (one SGPR operand per VOP3 instruction is allowed), which keeps N vector registers free
  PC_MAC_FNS(macs, PC_MAC1, PC_HI_INOUT, PC_Y_SGPR)
  PC_MAC_FNS(macsf, PC_MAC1F, PC_HI_OUT, PC_Y_SGPR)
  template <int CNT, bool FIRST = false>
  static __device__ __forceinline__ void mac_n(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y) {
    if constexpr (CNT >= 8) { if constexpr (FIRST) macf8(acc, hi, x, y); else mac8(acc, hi, x, y); mac_n<CNT - 8>(acc, hi, x + 8, y + 8); }
    else if constexpr (CNT >= 4) { if constexpr (FIRST) macf4(acc, hi, x, y); else mac4(acc, hi, x, y); mac_n<CNT - 4>(acc, hi, x + 4, y + 4); }
    else if constexpr (CNT >= 2) { if constexpr (FIRST) macf2(acc, hi, x, y); else mac2(acc, hi, x, y); mac_n<CNT - 2>(acc, hi, x + 2, y + 2); }
    else if constexpr (CNT == 1) { if constexpr (FIRST) macf1(acc, hi, x, y); else mac1(acc, hi, x, y); }
  }
  template <int CNT, bool FIRST = false>
  static __device__ __forceinline__ void mac_ns(uint64_t& acc, uint32_t& hi, const uint32_t* x, const uint32_t* y) {
    if constexpr (CNT >= 8) { if constexpr (FIRST) macsf8(acc, hi, x, y); else macs8(acc, hi, x, y); mac_ns<CNT - 8>(acc, hi, x + 8, y + 8); }
    else if constexpr (CNT >= 4) { if constexpr (FIRST) macsf4(acc, hi, x, y); else macs4(acc, hi, x, y); mac_ns<CNT - 4>(acc, hi, x + 4, y + 4); }
    else if constexpr (CNT >= 2) { if constexpr (FIRST) macsf2(acc, hi, x, y); else macs2(acc, hi, x, y); mac_ns<CNT - 2>(acc, hi, x + 2, y + 2); }
    else if constexpr (CNT == 1) { if constexpr (FIRST) macsf1(acc, hi, x, y); else macs1(acc, hi, x, y); }
  }
  // number of non-zero modulus limbs among MOD[lo..hi]
  static constexpr int nz_mod(int lo, int hi_) { int c = 0; for (int i = lo; i <= hi_; i++) c += P::MOD[i] != 0; return c; }
  // reduction terms of column K: m_i * p_{K-i} for i = I0..I1 (zero limbs of p skipped), the limbs of p as scalar constants
  template <int K, int I0, int I1, bool FIRST>
  static __device__ __forceinline__ void red_terms(const uint32_t* m, uint64_t& acc, uint32_t& hi) {
    constexpr int CR = I1 >= I0 ? nz_mod(K - I1, K - I0) : 0;
    if constexpr (CR > 0) {
      uint32_t x[CR], y[CR];
      int c = 0;
      PC_UNROLL for (int i = I0; i <= I1; i++) if (P::MOD[K - i] != 0) { x[c] = m[i]; y[c] = P::MOD[K - i]; c++; }
      mac_ns<CR, FIRST>(acc, hi, x, y);
    }
  }
  static __device__ __forceinline__ void red_last(const uint32_t* mk, uint64_t& acc, uint32_t& hi) {
    const uint32_t p0 = P::MOD[0];
    macs1(acc, hi, mk, &p0);
  }
  // Moduli with p = 1 (mod 2^32) -- BLS12-381 Fr, both Pasta fields -- have INV = -p^-1 = -1 (mod 2^32): the Montgomery factor of a
  // column is m = -acc_lo, and m * p_0 = m only clears the low word of the accumulator and carries iff that word was not zero.
  // Closes a column (m, the term m * p_0 and the shift to the next column) in four simple instructions instead of v_mul_lo_u32 +
  // v_mad_u64_u32 + v_addc (the two multiplier instructions are several times the issue cost of an addition on this part).
  static constexpr bool P0_ONE = PC_P0_ONE && P::MOD[0] == 1u && P::INV == 0xffffffffu;
  static __device__ __forceinline__ void close_column(uint32_t& mk, uint64_t& acc, uint32_t& hi) {
    if constexpr (P0_ONE) {
      const uint32_t lo = (uint32_t)acc;
      mk = 0u - lo;
      acc = ((acc >> 32) | ((uint64_t)hi << 32)) + (uint64_t)(lo != 0u);      // < 2^64: the 96-bit column sum carries at most into its own top words
    } else {
      mk = (uint32_t)acc * P::INV;
      red_last(&mk, acc, hi);
      acc = (acc >> 32) | ((uint64_t)hi << 32);
    }
  }

  // UNIT: the second operand is the raw integer 1 (Montgomery -> canonical conversion): the only
  // a_i * b_j left in column K is a_K * 1, so a column is one product plus the reduction terms.
  template <int K, bool UNIT>
  __device__ __forceinline__ void column_lo(const Fd& o, uint32_t* m, const uint32_t* mod, uint64_t& acc, uint32_t& hi) const {
    // column K < N: a_i*b_{K-i} (i = 0..K), m_i*p_{K-i} (i = 0..K-1, p_{K-i} != 0)
    constexpr int CNT = UNIT ? 1 : K + 1;
    uint32_t x[CNT], y[CNT];
    int c = 0;
    if constexpr (UNIT) { x[c] = l[K]; y[c] = o.l[0]; c++; }
    else { PC_UNROLL for (int i = 0; i <= K; i++) { x[c] = l[i]; y[c] = o.l[K - i]; c++; } }
    mac_n<CNT, true>(acc, hi, x, y);
    red_terms<K, 0, K - 1, false>(m, acc, hi);
    close_column(m[K], acc, hi);
    if constexpr (K + 1 < N) column_lo<K + 1, UNIT>(o, m, mod, acc, hi);
  }
  template <int K, bool UNIT>
  __device__ __forceinline__ void column_hi(const Fd& o, const uint32_t* m, const uint32_t* mod, uint64_t& acc, uint32_t& hi, uint32_t* t) const {
    // column K >= N: i = K-N+1 .. N-1
    constexpr int CNT = UNIT ? 0 : 2 * N - 1 - K;
    uint32_t x[CNT > 0 ? CNT : 1], y[CNT > 0 ? CNT : 1];
    int c = 0;
    if constexpr (!UNIT) { PC_UNROLL for (int i = K - N + 1; i < N; i++) { x[c] = l[i]; y[c] = o.l[K - i]; c++; } }
    mac_n<CNT, true>(acc, hi, x, y);      // (the last column has no products: its stale carry word only reaches discarded bits)
    red_terms<K, K - N + 1, N - 1, CNT == 0>(m, acc, hi);
    t[K - N] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)hi << 32);
    if constexpr (K + 1 < 2 * N) column_hi<K + 1, UNIT>(o, m, mod, acc, hi, t);
  }
  template <bool UNIT, bool REDUCE = true>
  __device__ __forceinline__ Fd mul_impl(const Fd& o) const {
    uint32_t m[N], t[N + 1];
    uint32_t mod[N];
    PC_UNROLL for (int i = 0; i < N; i++) mod[i] = P::MOD[i];
    uint64_t acc = 0; uint32_t hi = 0;
    column_lo<0, UNIT>(o, m, mod, acc, hi);
    column_hi<N, UNIT>(o, m, mod, acc, hi, t);
    Fd r;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = t[i];
    if constexpr (REDUCE) cond_sub(r.l, (uint32_t)acc);
    return r;
  }
  __device__ __forceinline__ Fd mul(const Fd& o) const { return mul_impl<false>(o); }
  __device__ __forceinline__ Fd sqr() const { return sqr_inl(); }
  __device__ __forceinline__ Fd mul_add_mul(const Fd& b, const Fd& c, const Fd& d) const { return mul_add_mul_inl(b, c, d); }
  // Lazy forms (see LAZY_OK below): operands in [0, 2p], result in [0, 2p); no final conditional subtraction (the fused pair
  // keeps its one where R < 8p)
  __device__ __forceinline__ Fd mul_lz(const Fd& o) const { return mul_impl<false, false>(o); }
  __device__ __forceinline__ Fd sqr_lz() const { return sqr_inl<false>(); }
  __device__ __forceinline__ Fd mul_add_mul_lz(const Fd& b, const Fd& c, const Fd& d) const { return mul_add_mul_inl<!LAZY_FUSED_OK>(b, c, d); }
  // (the three products as real functions -- operands by value, in registers -- were measured against the ~50 KB of
  // inlined code of a mixed addition: accumulate 36.8 -> 44.0 ms; the instruction cache is not what limits it)

  // a*b + c*d with ONE reduction: the columns of both products are gathered into the same accumulator before the
  // m_i * p terms, 3 N^2 instead of 4 N^2 partial products for the pair (the sum stays below 2 p^2 < p R, so the
  // result is below 2 p and the single conditional subtraction of mul() still canonicalises it).
  template <int K>
  __device__ __forceinline__ void dual_column_lo(const Fd& b, const Fd& c, const Fd& d, uint32_t* m, const uint32_t* mod, uint64_t& acc, uint32_t& hi) const {
    constexpr int CNT = 2 * (K + 1);
    uint32_t x[CNT], y[CNT];
    int n = 0;
    PC_UNROLL for (int i = 0; i <= K; i++) { x[n] = l[i]; y[n] = b.l[K - i]; n++; }
    PC_UNROLL for (int i = 0; i <= K; i++) { x[n] = c.l[i]; y[n] = d.l[K - i]; n++; }
    mac_n<CNT, true>(acc, hi, x, y);
    red_terms<K, 0, K - 1, false>(m, acc, hi);
    close_column(m[K], acc, hi);
    if constexpr (K + 1 < N) dual_column_lo<K + 1>(b, c, d, m, mod, acc, hi);
  }
  template <int K>
  __device__ __forceinline__ void dual_column_hi(const Fd& b, const Fd& c, const Fd& d, const uint32_t* m, const uint32_t* mod, uint64_t& acc, uint32_t& hi,
                                                 uint32_t* t) const {
    constexpr int CNT = 2 * (2 * N - 1 - K);
    uint32_t x[CNT > 0 ? CNT : 1], y[CNT > 0 ? CNT : 1];
    int n = 0;
    PC_UNROLL for (int i = K - N + 1; i < N; i++) { x[n] = l[i]; y[n] = b.l[K - i]; n++; }
    PC_UNROLL for (int i = K - N + 1; i < N; i++) { x[n] = c.l[i]; y[n] = d.l[K - i]; n++; }
    mac_n<CNT, true>(acc, hi, x, y);
    red_terms<K, K - N + 1, N - 1, CNT == 0>(m, acc, hi);
    t[K - N] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)hi << 32);
    if constexpr (K + 1 < 2 * N) dual_column_hi<K + 1>(b, c, d, m, mod, acc, hi, t);
  }
  template <bool REDUCE = true>
  __device__ __forceinline__ Fd mul_add_mul_inl(const Fd& b, const Fd& c, const Fd& d) const {
    static_assert(P::BITS < 32 * N, "the fused pair assumes 2 p <= R");
    uint32_t m[N], t[N + 1], mod[N];
    PC_UNROLL for (int i = 0; i < N; i++) mod[i] = P::MOD[i];
    uint64_t acc = 0; uint32_t hi = 0;
    dual_column_lo<0>(b, c, d, m, mod, acc, hi);
    dual_column_hi<N>(b, c, d, m, mod, acc, hi, t);
    Fd r;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = t[i];
    if constexpr (REDUCE) cond_sub(r.l, (uint32_t)acc);
    return r;
  }

  // Dedicated squaring: a^2 = sum_i a_i^2 2^(64 i) + sum_{i<j} a_i (2 a_j) 2^(32 (i+j)).  The doubled cross terms are
  // taken from d = a << 1 (it fits N limbs: every modulus here leaves a spare top bit): row i multiplies a_i by the limbs of
  // ((a >> 32 (i+1)) << 1), i.e. d_j for j >= i+2 and d_{i+1} without the bit that a_i shifted in.  N (N-1)/2 + N partial
  // products instead of N^2 (78 vs 144 at 12 limbs), the same interleaved reduction, no doubling pass over the columns.
  template <int K>
  __device__ __forceinline__ void sq_column_lo(const uint32_t* d, const uint32_t* dm, uint32_t* m, const uint32_t* mod, uint64_t& acc, uint32_t& hi) const {
    constexpr int NCROSS = (K + 1) / 2;                      // i < j, i + j = K, i = 0 .. NCROSS-1
    constexpr int CNT = NCROSS + ((K & 1) ? 0 : 1);
    uint32_t x[CNT > 0 ? CNT : 1], y[CNT > 0 ? CNT : 1];
    int c = 0;
    PC_UNROLL for (int i = 0; i < NCROSS; i++) { x[c] = l[i]; y[c] = (K - i == i + 1) ? dm[K - i] : d[K - i]; c++; }
    if constexpr ((K & 1) == 0) { x[c] = l[K / 2]; y[c] = l[K / 2]; c++; }
    mac_n<CNT, true>(acc, hi, x, y);
    red_terms<K, 0, K - 1, CNT == 0>(m, acc, hi);
    close_column(m[K], acc, hi);
    if constexpr (K + 1 < N) sq_column_lo<K + 1>(d, dm, m, mod, acc, hi);
  }
  template <int K>
  __device__ __forceinline__ void sq_column_hi(const uint32_t* d, const uint32_t* dm, const uint32_t* m, const uint32_t* mod, uint64_t& acc, uint32_t& hi,
                                               uint32_t* t) const {
    constexpr int I0 = K - N + 1;                            // i = I0 .. N-1 with j = K - i; cross terms: i < j  <=>  2 i < K
    constexpr int NCROSS = (K + 1) / 2 - I0 > 0 ? (K + 1) / 2 - I0 : 0;
    constexpr int DIAG = ((K & 1) == 0 && K / 2 < N) ? 1 : 0;
    constexpr int CNT = NCROSS + DIAG;
    uint32_t x[CNT > 0 ? CNT : 1], y[CNT > 0 ? CNT : 1];
    int c = 0;
    PC_UNROLL for (int i = I0; i < I0 + NCROSS; i++) { x[c] = l[i]; y[c] = (K - i == i + 1) ? dm[K - i] : d[K - i]; c++; }
    if constexpr (DIAG) { x[c] = l[K / 2]; y[c] = l[K / 2]; c++; }
    mac_n<CNT, true>(acc, hi, x, y);
    red_terms<K, I0, N - 1, CNT == 0>(m, acc, hi);
    t[K - N] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)hi << 32);
    if constexpr (K + 1 < 2 * N) sq_column_hi<K + 1>(d, dm, m, mod, acc, hi, t);
  }
  template <bool REDUCE = true>
  __device__ __forceinline__ Fd sqr_inl() const {
    static_assert(P::BITS < 32 * N, "squaring assumes that 2a fits N limbs");
    static_assert(REDUCE || P::BITS + 2 <= 32 * N, "the lazy squaring doubles operands up to 2p: 4p must fit N limbs");
    uint32_t d[N], dm[N], m[N], t[N + 1], mod[N];
    PC_UNROLL for (int i = 0; i < N; i++) { mod[i] = P::MOD[i]; d[i] = (l[i] << 1) | (i ? l[i - 1] >> 31 : 0u); dm[i] = l[i] << 1; }
    uint64_t acc = 0; uint32_t hi = 0;
    sq_column_lo<0>(d, dm, m, mod, acc, hi);
    sq_column_hi<N>(d, dm, m, mod, acc, hi, t);
    Fd r;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = t[i];
    if constexpr (REDUCE) cond_sub(r.l, (uint32_t)acc);
    return r;
  }
#else
  PC_HD Fd mul(const Fd& o) const {
    uint32_t t[N + 1];
    PC_UNROLL for (int i = 0; i <= N; i++) t[i] = 0;
    PC_UNROLL for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      const uint32_t bi = o.l[i];
      PC_UNROLL for (int j = 0; j < N; j++) {
        c = (uint64_t)l[j] * bi + t[j] + c;      // <= 2^64 - 1, never overflows
        t[j] = (uint32_t)c; c >>= 32;
      }
      c += t[N];
      t[N] = (uint32_t)c;
      const uint32_t top = (uint32_t)(c >> 32);
      const uint32_t m = t[0] * P::INV;
      c = (uint64_t)m * P::MOD[0] + t[0];
      c >>= 32;
      PC_UNROLL for (int j = 1; j < N; j++) {
        c = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)c; c >>= 32;
      }
      c += t[N];
      t[N - 1] = (uint32_t)c;
      t[N] = top + (uint32_t)(c >> 32);
    }
    Fd r;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = t[i];
    cond_sub(r.l, t[N]);
    return r;
  }
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
  PC_HD Fd sqr() const { return mul(*this); }
  PC_HD Fd mul_add_mul(const Fd& b, const Fd& c, const Fd& d) const { return mul(b).add(c.mul(d)); }
  // host: (a b [+ c d] + m p) / R by rows with ONE interleaved reduction and, for `reduce == false`, no final subtraction --
  // Montgomery's m is determined by the sum alone, so these are the device's lazy values bit for bit (the CPU-stepped
  // tests then exercise the real [0, 2p) ranges, not canonicalised stand-ins)
  static inline Fd mont_rows_host(const Fd& a, const Fd& b, const Fd* c, const Fd* d, bool reduce) {
    uint32_t t[N + 2];
    for (int i = 0; i < N + 2; i++) t[i] = 0;
    for (int i = 0; i < N; i++) {
      for (int pass = 0; pass < (c ? 2 : 1); pass++) {
        const Fd& x = pass ? *c : a; const uint32_t yi = pass ? d->l[i] : b.l[i];
        uint64_t cy = 0;
        for (int j = 0; j < N; j++) { cy += (uint64_t)x.l[j] * yi + t[j]; t[j] = (uint32_t)cy; cy >>= 32; }
        cy += t[N]; t[N] = (uint32_t)cy; t[N + 1] += (uint32_t)(cy >> 32);
      }
      const uint32_t m = t[0] * P::INV;
      uint64_t cy = (uint64_t)m * P::MOD[0] + t[0];
      cy >>= 32;
      for (int j = 1; j < N; j++) { cy += (uint64_t)m * P::MOD[j] + t[j]; t[j - 1] = (uint32_t)cy; cy >>= 32; }
      cy += t[N]; t[N - 1] = (uint32_t)cy; cy >>= 32;
      cy += t[N + 1]; t[N] = (uint32_t)cy; t[N + 1] = (uint32_t)(cy >> 32);
    }
    Fd r;
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    if (reduce) cond_sub(r.l, t[N]);
    return r;
  }
  PC_HD Fd mul_lz(const Fd& o) const { return mont_rows_host(*this, o, nullptr, nullptr, false); }
  PC_HD Fd sqr_lz() const { return mont_rows_host(*this, *this, nullptr, nullptr, false); }
  PC_HD Fd mul_add_mul_lz(const Fd& b, const Fd& c, const Fd& d) const { return mont_rows_host(*this, b, &c, &d, !LAZY_FUSED_OK); }
#endif

  // ---- lazy reduction (Walter's bound) ---------------------------------------------------------------------------
  // With R >= 4p a Montgomery product of operands in [0, 2p] is below (4p^2 + pR)/R <= 2p WITHOUT the final conditional
  // subtraction: inside a long chain of multiplications (the mixed addition of the bucket accumulation: 9 multiplier calls)
  // values are kept in [0, 2p) and canonicalised only where they leave the chain; additive operations then work modulo 2p
  // (their results, up to 4p, must fit N limbs: the same bound).  LAZY_OK: BLS12-381 Fq (381 bits in 384) and BN254 Fq
  // (254 in 256; R = 5.29p); Pallas' p = 2^254 + ... has 4p > 2^256 and keeps the canonical path.
  // The fused a*b + c*d is below (8p^2 + pR)/R: below 2p only when R >= 8p (LAZY_FUSED_OK: BLS12-381); with 4p <= R < 8p it is
  // below 3p and keeps its ONE conditional subtraction, which brings it below 2p again (not necessarily below p).
  // LAZY_STORE_OK (R >= 9p): lazily reduced coordinates may also LEAVE the accumulation as they are -- the consumers' formulas
  // (XyzzD::add / dbl, canonical multiplier) start with products of loaded coordinates, except dbl()'s U = 2Y (one conditional
  // subtraction: below 3p for Y < 2p), whose square needs 9p^2 < pR to come out canonical.
  static constexpr bool LAZY_OK = P::BITS + 2 <= 32 * N && (N > 8 || PC_LAZY_N8);
  static constexpr bool LAZY_FUSED_OK = P::BITS + 3 <= 32 * N;
  static constexpr bool LAZY_STORE_OK = LAZY_FUSED_OK && ((uint64_t)P::MOD[N - 1] + 1) * 9 <= ((uint64_t)1 << 32);
  static PC_HD uint32_t mod2(int i) { return (P::MOD[i] << 1) | (i ? P::MOD[i - 1] >> 31 : 0u); }      // limbs of 2p
  // (a - b) mod 2p for a, b in [0, 2p]: in [0, 2p] (in [0, 2p) when a < 2p)
  PC_HD Fd sub_lz(const Fd& o) const {
    Fd r; uint32_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = sbb(l[i], o.l[i], br);
    const uint32_t mask = (uint32_t)0 - br;
    uint32_t c = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = adc(r.l[i], mod2(i) & mask, c);
    return r;
  }
  // 2 (a) mod 2p-representation: a in [0, 2p) -> [0, 2p)
  PC_HD Fd dbl_lz() const {
    Fd r; uint32_t c = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = adc(l[i], l[i], c);
    uint32_t d[N]; uint32_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) d[i] = sbb(r.l[i], mod2(i), br);
    const bool ge = br == 0;                     // 4p < 2^(32 N): no carry word
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = ge ? d[i] : r.l[i];
    return r;
  }
  // 2p - a in [0, 2p] (no special case for zero: 2p is a valid representative of it)
  PC_HD Fd neg_lz() const {
    Fd r; uint32_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = sbb(mod2(i), l[i], br);
    return r;
  }
  // p - a for a canonical a (table points): in (0, p]
  PC_HD Fd neg_lz_canonical() const {
    Fd r; uint32_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] = sbb(P::MOD[i], l[i], br);
    return r;
  }
  // value = 0 (mod p) for a value in [0, 2p]
  PC_HD bool is_zero_lz() const {
    uint32_t z = 0, e1 = 0, e2 = 0;
    PC_UNROLL for (int i = 0; i < N; i++) { z |= l[i]; e1 |= l[i] ^ P::MOD[i]; e2 |= l[i] ^ mod2(i); }
    return z == 0 || e1 == 0 || e2 == 0;
  }
  // [0, 2p) -> [0, p): one conditional subtraction (multiplier outputs and sub_lz results are strictly below 2p)
  PC_HD Fd canon1() const {
    Fd r = *this;
    cond_sub(r.l, 0);
    return r;
  }
  // [0, 2p] -> [0, p)
  PC_HD Fd canon() const {
    Fd r = *this;
    cond_sub(r.l, 0);
    cond_sub(r.l, 0);          // (only 2p itself needs the second one)
    return r;
  }

  // Montgomery <-> canonical
  PC_HD Fd from_mont() const {   // multiply by raw 1 => a * R^-1
    Fd o = zero(); o.l[0] = 1;
#if defined(__HIP_DEVICE_COMPILE__)
    return mul_impl<true>(o);    // reduction only: N^2 + N partial products instead of 2 N^2 + N
#else
    return mul(o);
#endif
  }
  PC_HD Fd to_mont() const { Fd r2; PC_UNROLL for (int i = 0; i < N; i++) r2.l[i] = P::R2[i]; return mul(r2); }

  // Fermat inverse a^(p-2); 0 -> 0.  Used once per MSM / per batch, never per element.
  PC_HD Fd inv() const {
    // exponent p - 2, derived word by word from the constant modulus (a local copy of it indexed by the loop counter
    // would live in scratch memory on the device); Pallas' p = ...00000001 borrows out of word 0
    Fd r = one();
    for (int w = N - 1; w >= 0; w--) {
      uint32_t e = 0; uint64_t br = 2;
      for (int i = 0; i <= w; i++) { const uint64_t t = (uint64_t)P::MOD[i] - br; e = (uint32_t)t; br = t >> 63; }
      for (int b = 31; b >= 0; b--) {
        r = r.sqr();
        if ((e >> b) & 1) r = r.mul(*this);
      }
    }
    return r;
  }
};

}  // namespace pc
