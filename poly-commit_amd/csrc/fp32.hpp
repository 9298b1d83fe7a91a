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
  static PC_HD void cond_sub(uint32_t* a, uint32_t hi) {
    uint32_t d[N];
    uint64_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)a[i] - P::MOD[i] - br;
      d[i] = (uint32_t)t; br = (t >> 63);
    }
    // a >= MOD  <=>  no final borrow, or the carry word absorbs it
    bool ge = (hi != 0) || (br == 0);
    PC_UNROLL for (int i = 0; i < N; i++) a[i] = ge ? d[i] : a[i];
  }

  PC_HD Fd add(const Fd& o) const {
    Fd r; uint64_t c = 0;
    PC_UNROLL for (int i = 0; i < N; i++) { c += (uint64_t)l[i] + o.l[i]; r.l[i] = (uint32_t)c; c >>= 32; }
    cond_sub(r.l, (uint32_t)c);   // every modulus here leaves >= 1 spare top bit, so c == 0
    return r;
  }
  PC_HD Fd sub(const Fd& o) const {
    Fd r; uint64_t br = 0;
    PC_UNROLL for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)l[i] - o.l[i] - br;
      r.l[i] = (uint32_t)t; br = (t >> 63);
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;   // add MOD back on borrow
    uint64_t c = 0;
    PC_UNROLL for (int i = 0; i < N; i++) { c += (uint64_t)r.l[i] + (P::MOD[i] & mask); r.l[i] = (uint32_t)c; c >>= 32; }
    return r;
  }
  PC_HD Fd dbl() const { return add(*this); }
  PC_HD Fd neg() const {
    // MOD - a, or 0 for a == 0
    Fd r; uint64_t br = 0; uint32_t nz = 0;
    PC_UNROLL for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)P::MOD[i] - l[i] - br;
      r.l[i] = (uint32_t)t; br = (t >> 63); nz |= l[i];
    }
    uint32_t mask = nz ? 0xffffffffu : 0u;
    PC_UNROLL for (int i = 0; i < N; i++) r.l[i] &= mask;
    return r;
  }

  // CIOS Montgomery product: r = a * b * R^-1 mod p.
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
  PC_HD Fd sqr() const { return mul(*this); }

  // Montgomery <-> canonical
  PC_HD Fd from_mont() const {   // multiply by raw 1 => a * R^-1
    Fd o = zero(); o.l[0] = 1; return mul(o);
  }
  PC_HD Fd to_mont() const { Fd r2; PC_UNROLL for (int i = 0; i < N; i++) r2.l[i] = P::R2[i]; return mul(r2); }

  // Fermat inverse a^(p-2); 0 -> 0.  Used once per MSM / per batch, never per element.
  PC_HD Fd inv() const {
    uint32_t e[N];   // p - 2
    uint64_t br = 2;
    PC_UNROLL for (int i = 0; i < N; i++) {
      uint64_t t = (uint64_t)P::MOD[i] - br;
      e[i] = (uint32_t)t; br = (t >> 63);
    }
    Fd r = one();
    for (int i = N * 32 - 1; i >= 0; i--) {
      r = r.sqr();
      if ((e[i >> 5] >> (i & 31)) & 1) r = r.mul(*this);
    }
    return r;
  }
};

}  // namespace pc
