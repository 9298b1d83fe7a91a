// Batched forward radix-2 NTT over the scalar field: the Reed-Solomon encoder under Ligero.
//
// Replaces GeneralEvaluationDomain::<F>::new(m * rho_inv).fft(msg) at
// poly-commit/src/linear_codes/utils.rs:119-126 (called per matrix row from
// linear_codes/mod.rs:131-135).  Semantics pinned by test_reed_solomon (utils.rs:303-331):
//   out[j] = sum_i in[i] * omega^(i*j),  natural order in and out,
//   omega = TWO_ADIC_ROOT_OF_UNITY^(2^(s - log_n))   (arkworks' FftField constants).
// Each row is zero-padded from in_cols to N = 2^log_n (the padding is never read).
//
// Four-step decomposition N = N1 * N2, two kernels, both staging a tile in LDS (limb-pair-major with a
// bank swizzle, LdsTile below: 8-byte accesses, conflict-free in every phase):
//   pass A  tile = C adjacent columns i2 of the N1 x N2 view; N1-point NTT down each column
//           (bit-reversed on the way into LDS, DIT butterflies), then * omega_N^(i2*j1)
//   pass B  tile = R adjacent rows j1; N2-point NTT along each row; transposed store
//           X[j1 + N1*j2] (R*32 B contiguous runs)
// Twiddles come from one table W[j] = omega_N^j, j < N (<= 4 MiB at 2^17: L2 resident); the butterfly
// stages of a pass read theirs from a copy in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "fp32.hpp"
#include "hip_backend.hpp"

namespace pc {

// Lanes and elements per tile, per pass.  The LDS tile, not the VGPRs (85), bounds the workgroups per CU, so the occupancy is set here.
// Pass A: 512 lanes on 2048 elements (64 KiB): 2 workgroups = 4 waves per SIMD; its tile is C = 4 adjacent columns, i.e. 128-byte runs in
// global memory (256 lanes left 2 waves per SIMD: 6.01 -> 5.48 ms for the 2^24-coefficient batch in round 3; half tiles with 256 lanes
// would keep 4 waves per SIMD but read 64-byte runs).  Pass B: 256 lanes on 1024 elements: 4 independent workgroups per CU instead of 2
// (a barrier then spans 4 waves, and a workgroup's load / store phases overlap three others' butterflies); its R = 4 rows still store
// 128-byte runs (round 5: pass B 2.33 -> 2.23 ms).  The kernels take their lane count from blockDim.
#ifndef PC_NTT_THREADS
#define PC_NTT_THREADS 512
#endif
#ifndef PC_NTT_TILE
#define PC_NTT_TILE 2048
#endif
#ifndef PC_NTT_THREADS_B
#define PC_NTT_THREADS_B 256
#endif
#ifndef PC_NTT_TILE_B
#define PC_NTT_TILE_B 1024
#endif
static constexpr int NTT_THREADS_MAX = 512;
static constexpr uint32_t NTT_TILE_MAX = PC_NTT_TILE;   // elements per LDS tile of pass A (64 KiB of 32-byte elements)

template <class FrP>
struct PowTable { uint32_t w[32][FrP::N]; };   // w[k] = omega_N^(2^k)

template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_twiddles(uint32_t* W, uint32_t n, PowTable<FrP> pt) {
  typedef Fd<FrP> F;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  F acc = F::one();
  for (uint32_t k = 0; (j >> k) != 0; k++)
    if ((j >> k) & 1) acc = acc.mul(F::load(pt.w[k]));
  acc.store(W + (size_t)j * FrP::N);
}

PC_HD uint32_t bitrev(uint32_t v, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

// An LDS tile of field elements, limb-PAIR-major: element `pos` keeps limbs (2m, 2m + 1) at base[m * stride + phys(pos)], so an
// element is N / 2 ds_read_b64 / ds_write_b64 (half the LDS instructions and, for reads, half the LDS cycles of dword accesses).
//
// phys() permutes the low five bits of a position as a function of its higher bits -- a bijection, so any position may be used
// anywhere -- chosen so that each 32-lane group of every access pattern of the two passes lands in 32 distinct 8-byte slots (the LDS
// services 32 lanes per cycle and serialises lanes that share a bank; round 4 ran the butterfly stages 1 .. 5 of a tile with 2- to
// 4-way conflicts on every data access and up to 8-way on the twiddles):
//   DATA   the positions of a radix-4 group of stage s are p | {0, 1, 2, 3} << (s - 1): consecutive lanes vary bits 0 .. s-2 and
//          s+1 .. 6 of p, bits s-1 and s (both below 5 for s <= 4) are zero in every lane.  bank = low5 ^ T(bit 5, bit 6) with
//          T(1, 0) = 10101b, T(0, 1) = 11010b: over GF(2) the two vectors project onto every pair of adjacent bank bits (0,1), (1,2),
//          (2,3), (3,4) as a basis, so bits 5, 6 fill whichever two bank bits the stage leaves empty; stages >= 6 vary bits 0 .. 4.
//          Bits 7 .. 10 (constant inside a group of a stage) are folded in as well, onto bank bits 0, 2, 3, 4: they are what varies in
//          the bit-reversed fill and in the transposed read-out of the tile.  Lines start at multiples of the line length (no padding
//          word: an added offset would carry into the bits the argument is about).
//   TWID   a stage reads omega^(j << sh) for consecutive j: five consecutive bits sh .. sh+4 vary.  bank = XOR of the 5-bit digits of
//          the position: every window of five consecutive bits hits each bank bit exactly once.
// tuning experiment switch: 0 keeps the positions as they are (the conflicts of round 4, with 8-byte accesses)
#ifndef PC_NTT_SWIZZLE
#define PC_NTT_SWIZZLE 1
#endif
enum { LDS_PLAIN = 0, LDS_DATA = PC_NTT_SWIZZLE ? 1 : 0, LDS_TWID = PC_NTT_SWIZZLE ? 2 : 3 };
template <class FrP, int MODE>
struct LdsTile {
  typedef Fd<FrP> F;
  static_assert(FrP::N % 2 == 0, "limb pairs");
  uint2* base; uint32_t stride;   // 8-byte stride between limb pairs (= tile elements)
  static __device__ __forceinline__ uint32_t phys(uint32_t pos) {
    if constexpr (MODE == 1) {
      const uint32_t x = pos >> 5;
      return pos ^ ((0u - (x & 1u)) & 21u) ^ ((0u - ((x >> 1) & 1u)) & 26u) ^ ((x >> 2) & 1u) ^ (((x >> 3) & 7u) << 2);
    } else if constexpr (MODE == 2) {
      return pos ^ (((pos >> 5) ^ (pos >> 10)) & 31u);
    } else return pos;
  }
  // (phys is linear over GF(2) with phys(0) = 0: phys(a ^ b) = phys(a) ^ phys(b).  The four positions of a radix-4 group differ in two
  // bits that are zero in the first one, p + k h = p ^ k h, so a group costs ONE phys() and three XORs with wave-uniform constants)
  __device__ __forceinline__ F get_at(uint32_t p) const {      // p: a physical position
    F r;
#pragma unroll
    for (int m = 0; m < FrP::N / 2; m++) { const uint2 v = base[m * stride + p]; r.l[2 * m] = v.x; r.l[2 * m + 1] = v.y; }
    return r;
  }
  __device__ __forceinline__ void put_at(uint32_t p, const F& v) const {
#pragma unroll
    for (int m = 0; m < FrP::N / 2; m++) base[m * stride + p] = make_uint2(v.l[2 * m], v.l[2 * m + 1]);
  }
  __device__ __forceinline__ F get(uint32_t pos) const { return get_at(phys(pos)); }
  __device__ __forceinline__ void put(uint32_t pos, const F& v) const { put_at(phys(pos), v); }
};

// DIT butterfly stages over `lines` independent length-2^lg sequences laid out in LDS at
// pos = line * len + p (bit-reversed input, natural output).
// The 2^(lg-1) twiddles omega_{2^lg}^j of these stages are staged in LDS first (`tw`, limb-major like
// the tile): every butterfly then reads its twiddle from LDS instead of gathering 32 bytes from the
// global table (5.7e8 L2 gathers per 2^24-coefficient batch).
template <class FrP>
__device__ __forceinline__ void lds_ntt_stages(const LdsTile<FrP, LDS_DATA>& t, uint32_t lines, uint32_t lg, const uint32_t* W,
                                               uint32_t log_n_total, const LdsTile<FrP, LDS_TWID>& tw, uint32_t first_stage = 1) {
  typedef Fd<FrP> F;
  const uint32_t len = 1u << lg, halfs = len >> 1;
  for (uint32_t j = threadIdx.x; j < halfs; j += blockDim.x)
    tw.put(j, F::load(W + ((size_t)j << (log_n_total - lg)) * FrP::N));      // omega_{2^lg}^j = W[j << (log_n - lg)]
  __syncthreads();
  // every size is a power of two: index arithmetic is shifts and masks (an integer divide by a
  // run-time value costs ~30 VALU instructions on this ISA, four of them per butterfly)
  uint32_t s = first_stage;
  if (((lg - first_stage + 1) & 1) && s <= lg) {     // odd number of stages: one radix-2 stage first
    const uint32_t h = 1u << (s - 1);
    const uint32_t tw_shift = lg - s;                // omega_{2^s}^j = tw[j << (lg - s)]
    const uint32_t dh = t.phys(h);
    for (uint32_t b = threadIdx.x; b < lines * halfs; b += blockDim.x) {
      uint32_t line = b >> (lg - 1), k = b & (halfs - 1);
      uint32_t g = k >> (s - 1), j = k & (h - 1);
      const uint32_t q0 = t.phys(line * len + (g << s) + j), q1 = q0 ^ dh;
      F u = t.get_at(q0), v = t.get_at(q1);
      if (j) v = v.mul(tw.get(j << tw_shift));
      t.put_at(q0, u.add(v)); t.put_at(q1, u.sub(v));
    }
    __syncthreads();
    s++;
  }
  // Two stages per LDS round trip: a lane holds the four elements p, p+h, p+2h, p+3h of a group in
  // registers, applies stage s to (p, p+h) and (p+2h, p+3h) -- both with omega_{2^s}^j -- and stage
  // s+1 to (p, p+2h) with omega_{2^(s+1)}^j and (p+h, p+3h) with omega_{2^(s+1)}^(j+h): the same four
  // products as two radix-2 stages, half the LDS traffic and half the barriers.
  const uint32_t quarters = len >> 2;
  for (; s + 1 <= lg; s += 2) {
    const uint32_t h = 1u << (s - 1);
    const uint32_t sh1 = lg - s, sh2 = lg - s - 1;
    const uint32_t d1 = t.phys(h), d2 = t.phys(2 * h), d3 = d1 ^ d2, dw = tw.phys(h << sh2);      // wave-uniform
    for (uint32_t b = threadIdx.x; b < lines * quarters; b += blockDim.x) {
      uint32_t line = b >> (lg - 2), k = b & (quarters - 1);
      uint32_t g = k >> (s - 1), j = k & (h - 1);
      const uint32_t q0 = t.phys(line * len + (g << (s + 1)) + j);      // bits s-1 and s of the position are zero: + k h = ^ k h
      F x0 = t.get_at(q0), x1 = t.get_at(q0 ^ d1), x2 = t.get_at(q0 ^ d2), x3 = t.get_at(q0 ^ d3);
      const uint32_t qw = tw.phys(j << sh2);                             // (j + h) << sh2 = (j << sh2) ^ (h << sh2) for j < h
      if (j) { F w1 = tw.get(j << sh1); x1 = x1.mul(w1); x3 = x3.mul(w1); }
      F a0 = x0.add(x1), a1 = x0.sub(x1), a2 = x2.add(x3), a3 = x2.sub(x3);
      if (j) a2 = a2.mul(tw.get_at(qw));
      a3 = a3.mul(tw.get_at(qw ^ dw));
      t.put_at(q0, a0.add(a2)); t.put_at(q0 ^ d2, a0.sub(a2));
      t.put_at(q0 ^ d1, a1.add(a3)); t.put_at(q0 ^ d3, a1.sub(a3));
    }
    __syncthreads();
  }
}

template <class FrP>
__global__ void __launch_bounds__(NTT_THREADS_MAX) k_ntt_pass_a(const uint32_t* in, uint32_t in_cols, uint32_t* tmp, const uint32_t* W,
                                                           uint32_t log_n, uint32_t lg1, uint32_t C, uint32_t zskip) {
  typedef Fd<FrP> F;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t lg2 = log_n - lg1, N1 = 1u << lg1, N2 = 1u << lg2, N = 1u << log_n;
  const uint32_t lgC = 31 - __builtin_clz(C), tile_bits = lg2 - lgC;
  const uint32_t row = blockIdx.x >> tile_bits, tile = blockIdx.x & ((1u << tile_bits) - 1);
  LdsTile<FrP, LDS_DATA> t{(uint2*)smem, C * N1};
  LdsTile<FrP, LDS_TWID> tw{(uint2*)(smem + (size_t)FrP::N * C * N1), N1 / 2 > 1 ? N1 / 2 : 1};
  const uint32_t* rin = in + (size_t)row * in_cols * FrP::N;
  // Zero padding: if only the first N >> z coefficients can be non-zero (rho_inv = 4 -> z = 2), a
  // column holds data only at i1 < N1 >> z, i.e. (bit-reversed) at LDS positions = 0 mod 2^z, and
  // the first z DIT stages just replicate each value over its group of 2^z: skip them.
  const uint32_t zpow = 1u << zskip, n1nz = N1 >> zskip;
  for (uint32_t idx = threadIdx.x; idx < C * n1nz; idx += blockDim.x) {
    uint32_t c = idx & (C - 1), i1 = idx >> lgC;
    uint32_t i = i1 * N2 + tile * C + c;
    F v = (i < in_cols) ? F::load(rin + (size_t)i * FrP::N) : F::zero();
    uint32_t pos = c * N1 + bitrev(i1, lg1);
    for (uint32_t r = 0; r < zpow; r++) t.put(pos + r, v);
  }
  __syncthreads();
  lds_ntt_stages<FrP>(t, C, lg1, W, log_n, tw, zskip + 1);
  uint32_t* rout = tmp + (size_t)row * N * FrP::N;
  for (uint32_t idx = threadIdx.x; idx < C * N1; idx += blockDim.x) {
    uint32_t c = idx & (C - 1), j1 = idx >> lgC;
    uint32_t i2 = tile * C + c;
    F v = t.get(c * N1 + j1);
    uint32_t e = i2 * j1;                       // < N
    if (e) v = v.mul(F::load(W + (size_t)e * FrP::N));
    v.store(rout + ((size_t)j1 * N2 + i2) * FrP::N);
  }
}

template <class FrP>
__global__ void __launch_bounds__(NTT_THREADS_MAX) k_ntt_pass_b(const uint32_t* tmp, uint32_t* out, const uint32_t* W, uint32_t log_n,
                                                           uint32_t lg1, uint32_t R) {
  typedef Fd<FrP> F;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t lg2 = log_n - lg1, N1 = 1u << lg1, N2 = 1u << lg2, N = 1u << log_n;
  const uint32_t lgR = 31 - __builtin_clz(R), tile_bits = lg1 - lgR;
  const uint32_t row = blockIdx.x >> tile_bits, tile = blockIdx.x & ((1u << tile_bits) - 1);
  LdsTile<FrP, LDS_DATA> t{(uint2*)smem, R * N2};
  LdsTile<FrP, LDS_TWID> tw{(uint2*)(smem + (size_t)FrP::N * R * N2), N2 / 2 > 1 ? N2 / 2 : 1};
  const uint32_t* rin = tmp + ((size_t)row * N + (size_t)tile * R * N2) * FrP::N;
  for (uint32_t idx = threadIdx.x; idx < R * N2; idx += blockDim.x) {
    uint32_t r = idx >> lg2, i2 = idx & (N2 - 1);
    t.put(r * N2 + bitrev(i2, lg2), F::load(rin + (size_t)idx * FrP::N));
  }
  __syncthreads();
  lds_ntt_stages<FrP>(t, R, lg2, W, log_n, tw);
  uint32_t* rout = out + (size_t)row * N * FrP::N;
  for (uint32_t idx = threadIdx.x; idx < R * N2; idx += blockDim.x) {
    uint32_t r = idx & (R - 1), j2 = idx >> lgR;
    F v = t.get(r * N2 + j2);
    v.store(rout + ((size_t)(tile * R + r) + (size_t)N1 * j2) * FrP::N);
  }
}

template <class FrP>
class NttPlan {
 public:
  typedef Fd<FrP> F;
  NttPlan(HipBackend& be, unsigned log_n) : be_(be), log_n_(log_n) {
    const uint32_t N = 1u << log_n;
    lg1_ = (log_n + 1) / 2;
    W_ = (uint32_t*)be_.alloc((size_t)N * FrP::N * 4);
    PowTable<FrP> pt;
    F w = F::load(FrP::ROOT);
    for (unsigned i = log_n; i < (unsigned)FrP::TWO_ADICITY; i++) w = w.sqr();   // omega_N
    for (unsigned k = 0; k < 32; k++) { w.store(pt.w[k]); w = w.sqr(); }
    hipLaunchKernelGGL(k_ntt_twiddles<FrP>, dim3((N + 255) / 256), dim3(256), 0, be_.stream, W_, N, pt);
    PC_HIP_CHECK(hipGetLastError());
    be_.sync();
  }
  ~NttPlan() { be_.free(W_); be_.free(tmp_); }
  unsigned log_n() const { return log_n_; }

  // in: rows x in_cols, out: rows x 2^log_n, device pointers, Montgomery form.
  void run(const uint32_t* in, size_t rows, size_t in_cols, uint32_t* out) {
    if (rows == 0) return;
    const uint32_t N = 1u << log_n_, lg2 = log_n_ - lg1_, N1 = 1u << lg1_, N2 = 1u << lg2;
    size_t need = rows * (size_t)N * FrP::N * 4;
    if (need > tmp_bytes_) { be_.sync(); be_.free(tmp_); tmp_ = (uint32_t*)be_.alloc(need); tmp_bytes_ = need; }
    uint32_t C = 8; while (C > N2) C >>= 1; while (C > 1 && C * N1 > NTT_TILE_MAX) C >>= 1;
    uint32_t R = 8; while (R > N1) R >>= 1; while (R > 1 && R * N2 > (uint32_t)PC_NTT_TILE_B) R >>= 1;
    // tile + the stage twiddles of the pass (N1/2 resp. N2/2 elements)
    size_t lds_a = ((size_t)C * N1 + (N1 / 2 > 1 ? N1 / 2 : 1)) * FrP::N * 4, lds_b = ((size_t)R * N2 + (N2 / 2 > 1 ? N2 / 2 : 1)) * FrP::N * 4;
    if (lds_a > 160 * 1024 || lds_b > 160 * 1024) throw std::runtime_error("NTT size exceeds the LDS tile");
    if (lds_a > 64 * 1024) PC_HIP_CHECK(hipFuncSetAttribute((const void*)k_ntt_pass_a<FrP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    if (lds_b > 64 * 1024) PC_HIP_CHECK(hipFuncSetAttribute((const void*)k_ntt_pass_b<FrP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    uint32_t zskip = 0; while (zskip < lg1_ && in_cols <= ((size_t)N >> (zskip + 1))) zskip++;
    // Optional row groups (PC_HIP_NTT_GROUP = rows per group): pass B of a group runs right after its pass A, while the
    // group's intermediate could still sit in the 256 MB memory-side cache.  Measured on config 5 (32 / 64 rows per group):
    // 6.3 / 6.2 ms against 6.0 ms for the whole batch in one pair of launches -- the kernels are bound by the multiplier,
    // not by the 1 TB/s they move, and the shorter launches only add tail effects.  Off by default.
    static const size_t group_env = []() { const char* e = getenv("PC_HIP_NTT_GROUP"); return e ? (size_t)atol(e) : (size_t)0; }();
    size_t group = group_env;
    if (group == 0 || group > rows) group = rows;
    const bool single = group == rows;      // phase brackets: [pass A, pass B] for one group; [whole batch, 0] when grouped
    be_.mark();
    for (size_t r0 = 0; r0 < rows; r0 += group) {
      const size_t nr = std::min(group, rows - r0);
      const uint32_t* gin = in + r0 * in_cols * FrP::N;
      uint32_t* gtmp = tmp_ + r0 * (size_t)N * FrP::N;
      uint32_t* gout = out + r0 * (size_t)N * FrP::N;
      hipLaunchKernelGGL(k_ntt_pass_a<FrP>, dim3((unsigned)(nr * (N2 / C))), dim3(PC_NTT_THREADS), lds_a, be_.stream, gin,
                         (uint32_t)in_cols, gtmp, W_, log_n_, lg1_, C, zskip);
      PC_HIP_CHECK(hipGetLastError());
      if (single) be_.mark();
      hipLaunchKernelGGL(k_ntt_pass_b<FrP>, dim3((unsigned)(nr * (N1 / R))), dim3(PC_NTT_THREADS_B), lds_b, be_.stream, gtmp, gout, W_,
                         log_n_, lg1_, R);
      PC_HIP_CHECK(hipGetLastError());
    }
    if (!single) be_.mark();
    be_.mark();
  }

 private:
  HipBackend& be_;
  unsigned log_n_, lg1_;
  uint32_t* W_ = nullptr;
  uint32_t* tmp_ = nullptr;
  size_t tmp_bytes_ = 0;
};

}  // namespace pc
