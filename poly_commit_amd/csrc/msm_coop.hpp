// Workgroup-cooperative bucket-reduction level (HIP only).
//
// Contract of BucketLevelBitsBody (msm.hpp) for one group of K points, with the K points spread
// over K lanes of one workgroup and combined through LDS in log2(K) dependent EC additions (a
// lane-serial running sum needs 2*K): the later levels of the reduction are pure latency (a few
// thousand points, one dependent chain per launch), so the chain length is what counts.
#pragma once
#include <hip/hip_runtime.h>
#include "msm.hpp"

namespace pc {

template <class C>
struct LdsPoints {
  typedef XyzzD<C> Pt;
  uint32_t* base; uint32_t lanes;
  __device__ __forceinline__ void put(uint32_t lane, const Pt& p) const {
    uint32_t w[Pt::WORDS]; p.store(w);
#pragma unroll
    for (int k = 0; k < Pt::WORDS; k++) base[k * lanes + lane] = w[k];
  }
  __device__ __forceinline__ Pt get(uint32_t lane) const {
    uint32_t w[Pt::WORDS];
#pragma unroll
    for (int k = 0; k < Pt::WORDS; k++) w[k] = base[k * lanes + lane];
    return Pt::load(w);
  }
};

// grid = cnt * (1 + n_old) workgroups of K lanes.
// Weighted groups (a == 0): ONE binary tree yields S and every A_b (BucketLevelBitsBody's contract).
// At step d lane l adds the value of lane l + 2^d iff  l mod 2^(d+1)  is 0 (the S chain) or a power
// of two below 2^d (the tree of A_p, p = position of that bit); a lane whose lowest set bit is d is
// read but keeps its value, which seeds A_d.  After log2(K) steps lane 0 holds S, lane 2^b holds A_b.
template <class C>
__global__ void __launch_bounds__(256) k_bucket_level_coop(uint32_t K, uint32_t lgK, uint32_t weight_off, uint32_t cnt, uint32_t n_old,
                                                          const uint32_t* x, const uint32_t* old_in, uint32_t* out) {
  PC_LATENCY_KERNEL();
  typedef XyzzD<C> Pt;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  LdsPoints<C> lds{smem, K};
  const uint32_t a = blockIdx.x / cnt, gidx = blockIdx.x % cnt, l = threadIdx.x;
  const size_t stride = (size_t)cnt * Pt::WORDS;
  const uint32_t nw = lgK + weight_off;
  const uint32_t* src = (a == 0) ? x + ((size_t)gidx * K + l) * Pt::WORDS
                                 : old_in + ((size_t)(a - 1) * cnt * K + (size_t)gidx * K + l) * Pt::WORDS;
  Pt v = Pt::load(src);
  if (a == 0) {
    for (uint32_t d = 0; d < lgK; d++) {
      lds.put(l, v);
      __syncthreads();
      const uint32_t low = l & ((2u << d) - 1u);
      if ((low & (low - 1u)) == 0 && low < (1u << d)) v.add(lds.get(l + (1u << d)));
      __syncthreads();
    }
    uint32_t* o = out + (size_t)gidx * Pt::WORDS;
    if (l == 0) { v.store(o); if (weight_off) v.store(o + (size_t)(1 + lgK) * stride); }
    else if ((l & (l - 1u)) == 0) { uint32_t b = 31 - __builtin_clz(l); v.store(o + (size_t)(1 + b) * stride); }
  } else {
    for (uint32_t d = K >> 1; d >= 1; d >>= 1) {     // plain tree sum of an older array
      lds.put(l, v);
      __syncthreads();
      if (l < d) v.add(lds.get(l + d));
      __syncthreads();
    }
    if (l == 0) v.store(out + (size_t)(nw + a) * stride + (size_t)gidx * Pt::WORDS);
  }
}

// ---- the same level with TWO lanes per point ---------------------------------------------------------------------------
// A cooperative level is a chain of log2(K) dependent additions on a lone wave per SIMD: its time is the latency of one XYZZ
// addition (12M + 2S, ~17 us) times the number of bits.  Here an even lane holds (X, ZZ) and its odd neighbour (Y, ZZZ) of the
// same point, and the addition is laid out so that both run the SAME instruction stream on their halves:
//   slot 1  A1*B2          U1            | S1              4  B1*B2     ZZ1*ZZ2      | ZZZ1*ZZZ2
//        2  A2*B1          U2            | S2              5  D*DD      PPP          | (unused)
//           D = T2 - T1    P             | R                  exchange: even <- RR, odd <- PPP
//        3  D^2            PP            | RR              6  U1*PP = Q              | ZZZ12*PPP = ZZZ3
//                                                             X3 = RR - PPP - 2Q, exchange: odd <- Q - X3
//                                                          7  ZZ12*PP = ZZ3          | R*(Q - X3) - S1*PPP = Y3   (fused pair)
// 7.3 multiplication times instead of 13, three neighbour exchanges (DPP quad_perm [1,0,3,2]).  Same formulas, same
// intermediate values as XyzzD::add: the results are bit-identical.  Equal x (doubling / P + (-P)) gathers the whole point into
// both lanes and runs XyzzD::dbl (rare).  Used for the levels that are latency-bound (few points); wide levels keep one lane
// per point (twice the lanes would only add work there).
template <class Fq>
__device__ __forceinline__ Fq lane_xchg(const Fq& v) {       // the value of the neighbour lane (lane ^ 1)
  Fq r;
#pragma unroll
  for (int i = 0; i < Fq::N; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], 0xB1, 0xF, 0xF, true);
  return r;
}

// p += o on a lane pair (ec.hpp: HalfPt, HalfAdd -- the phases between the exchanges)
template <class C>
__device__ __forceinline__ void half_add(HalfPt<C>& p, const HalfPt<C>& o, bool odd) {
  typedef Fd<typename C::FqP> Fq;
  if (o.b.is_zero()) return;                      // infinity has ZZ = ZZZ = 0: both lanes of a pair agree
  if (p.b.is_zero()) { p = o; return; }
  HalfAdd<C> h;
  const int dz = h.p1(p, o) ? 1 : 0, dz_nb = __builtin_amdgcn_mov_dpp(dz, 0xB1, 0xF, 0xF, true);
  const bool pz = odd ? dz_nb != 0 : dz != 0, rz = odd ? dz != 0 : dz_nb != 0;
  if (pz) {                                       // same x: doubling or P + (-P) -- the whole point into both lanes, XyzzD::dbl
    const Fq na = lane_xchg(p.a), nb = lane_xchg(p.b);
    XyzzD<C> f;
    f.X = fq_sel(odd, na, p.a); f.Y = fq_sel(odd, p.a, na); f.ZZ = fq_sel(odd, nb, p.b); f.ZZZ = fq_sel(odd, p.b, nb);
    p = HalfPt<C>::of(rz ? f.dbl() : XyzzD<C>::infinity(), odd);
    return;
  }
  const Fq rc1 = lane_xchg(h.p2(p, o, odd));
  const Fq rc2 = lane_xchg(h.p3(odd, rc1));
  h.p4(p, odd, rc2);
}

// grid = cnt * (1 + n_old) workgroups of 2 K <= 256 lanes; contract of k_bucket_level_coop
template <class C>
__global__ void __launch_bounds__(256) k_bucket_level_coop2(uint32_t K, uint32_t lgK, uint32_t weight_off, uint32_t cnt, uint32_t n_old,
                                                           const uint32_t* x, const uint32_t* old_in, uint32_t* out) {
  PC_LATENCY_KERNEL();
  typedef XyzzD<C> Pt;
  typedef Fd<typename C::FqP> Fq;
  constexpr int FN = Fq::N;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const uint32_t a = blockIdx.x / cnt, gidx = blockIdx.x % cnt, tid = threadIdx.x, l = tid >> 1, L = 2 * K;
  const bool odd = (tid & 1u) != 0;
  const uint32_t oa = odd ? FN : 0, ob = odd ? 3 * FN : 2 * FN;               // where this lane's halves sit in a stored point
  const size_t stride = (size_t)cnt * Pt::WORDS;
  const uint32_t nw = lgK + weight_off;
  const uint32_t* src = (a == 0) ? x + ((size_t)gidx * K + l) * Pt::WORDS
                                 : old_in + ((size_t)(a - 1) * cnt * K + (size_t)gidx * K + l) * Pt::WORDS;
  HalfPt<C> v; v.a = Fq::load(src + oa); v.b = Fq::load(src + ob);
  auto put = [&]() {
#pragma unroll
    for (int k = 0; k < FN; k++) { smem[k * L + tid] = v.a.l[k]; smem[(FN + k) * L + tid] = v.b.l[k]; }
  };
  auto get = [&](uint32_t t2) {
    HalfPt<C> r;
#pragma unroll
    for (int k = 0; k < FN; k++) { r.a.l[k] = smem[k * L + t2]; r.b.l[k] = smem[(FN + k) * L + t2]; }
    return r;
  };
  auto store = [&](uint32_t* dst) { v.a.store(dst + oa); v.b.store(dst + ob); };
  if (a == 0) {
    for (uint32_t d = 0; d < lgK; d++) {
      put();
      __syncthreads();
      const uint32_t low = l & ((2u << d) - 1u);
      if ((low & (low - 1u)) == 0 && low < (1u << d)) half_add<C>(v, get(tid + (2u << d)), odd);
      __syncthreads();
    }
    uint32_t* o = out + (size_t)gidx * Pt::WORDS;
    if (l == 0) { store(o); if (weight_off) store(o + (size_t)(1 + lgK) * stride); }
    else if ((l & (l - 1u)) == 0) { uint32_t b = 31 - __builtin_clz(l); store(o + (size_t)(1 + b) * stride); }
  } else {
    for (uint32_t d = K >> 1; d >= 1; d >>= 1) {
      put();
      __syncthreads();
      if (l < d) half_add<C>(v, get(tid + 2 * d), odd);
      __syncthreads();
    }
    if (l == 0) store(out + (size_t)(nw + a) * stride + (size_t)gidx * Pt::WORDS);
  }
}

// Bucket accumulation with an in-workgroup merge of the runs that chunk edges cut.
// AccumulateBody::chunk leaves, per lane, a partial for its first run when that run is not complete inside the chunk (slot 2t,
// key k0) and one for its last run when that is a different run and continues in the next chunk (slot 2t + 1, key k1).  A bucket
// cut by chunk edges is a chain of consecutive lanes: a START lane (its piece begins inside the chunk and runs off its end: the
// tail, or a head that begins at the chunk's first entry), any number of TRANSPARENT lanes (the whole chunk is one run of that
// bucket) and a CLOSING lane (its head ends inside the chunk).  With ~96 entries per bucket and chunks of 1500 (n = 2^24)
// every chain is start + close: lane t hands its last-run sum to lane t + 1 through LDS.  With chunks of 16 and ~20-32 entries
// per bucket (n <= 2^18) most chains have one or two transparent lanes, and before round 3 all of those went through the
// level-by-level segmented reduction (0.4-0.7 ms of a 1.1-2.0 ms MSM): now the pieces that flow right are combined by a
// segmented Hillis-Steele scan over the workgroup's lanes (one step per doubling of the longest chain, none when no lane is
// transparent), and the closing lane adds the scanned sum of its left neighbour.  Chains that leave the workgroup keep ONE
// partial per side (k_accumulate_edges joins those); only chains longer than that still reach the partial list.
// Occupancy the register allocator is held to (waves per SIMD; 1 = unconstrained): the 12-limb kernel needs ~212 VGPRs (2 waves);
// the 8-limb kernels ~150 (3 waves), PC_ACC_WAVES_N8 = 4 holds them to 128
#ifndef PC_ACC_WAVES_PER_EU
#define PC_ACC_WAVES_PER_EU 1
#endif
#ifndef PC_ACC_WAVES_N8
#define PC_ACC_WAVES_N8 1
#endif
template <class C> struct AccTune { static constexpr int WAVES = Fd<typename C::FqP>::N <= 8 ? PC_ACC_WAVES_N8 : PC_ACC_WAVES_PER_EU; };
#define PC_ACC_BOUNDS __launch_bounds__(256, AccTune<C>::WAVES)
template <class C>
__global__ void PC_ACC_BOUNDS k_accumulate(AccumulateBody<C> b, uint32_t lanes) {
  typedef XyzzD<C> Pt;
  __shared__ uint32_t xch[Pt::WORDS * 256];
  __shared__ uint32_t key_out[256], key_first[256], flags[256];
  const uint32_t tid = threadIdx.x, t = blockIdx.x * 256 + tid;
  const bool valid = t < lanes;
  uint32_t k0 = KEY_INVALID, k1 = KEY_INVALID;
  Pt V = Pt::infinity();                       // the lane's last run (AccumulateBody::chunk), then the scanned sum of its chain
  if (valid) b.chunk(t, k0, k1, V);
  const uint32_t M = b.offsets[b.g.NB];
  const uint64_t s64 = (uint64_t)t * b.g.T;
  const uint32_t cs = s64 < M ? (uint32_t)s64 : M;
  const uint32_t ce = (M - cs > b.g.T) ? cs + b.g.T : M;
  // the piece that flows to the right: the tail, or a head that runs off the chunk's end
  const bool head_left = k0 != KEY_INVALID && b.offsets[k0] < cs;                       // begins in an earlier chunk
  const bool head_right = k0 != KEY_INVALID && k1 == KEY_INVALID && b.offsets[k0 + 1] > ce;   // continues in the next chunk
  const bool transparent = head_left && head_right;
  const uint32_t okey = k1 != KEY_INVALID ? k1 : head_right ? k0 : KEY_INVALID;
  LdsPoints<C> lds{xch, 256};
  // flags: bit 0 = the sum reaches back to the chain's start (or to lane 0), bit 1 = it stopped at lane 0 without one
  uint32_t fl = (okey == KEY_INVALID || !transparent) ? 1u : 0u;
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const bool need = !(fl & 1u);
    if (!__syncthreads_or(need)) break;
    if (okey != KEY_INVALID) lds.put(tid, V);
    flags[tid] = fl;
    __syncthreads();
    if (need) {
      if (tid < d) fl = 3u;                      // lanes [0, tid] are all transparent: the chain began in an earlier workgroup
      else { V.add(lds.get(tid - d)); fl = flags[tid - d]; }
    }
    __syncthreads();
  }
  key_out[tid] = okey; key_first[tid] = head_left ? k0 : KEY_INVALID; flags[tid] = fl;
  if (okey != KEY_INVALID) lds.put(tid, V);
  __syncthreads();
  const bool take = head_left && tid > 0 && key_out[tid - 1] == k0;
  const bool give = okey != KEY_INVALID && tid < 255 && key_first[tid + 1] == okey;
  if (transparent) {
    // V already holds everything of this bucket from the chain's start (or lane 0) to here
    if (give) k0 = KEY_INVALID;
    else V.store(b.ppts + (size_t)(2 * t) * Pt::WORDS);      // last lane of the workgroup: one partial for the chain so far
  } else {
    if (take) {
      uint32_t* slot = b.ppts + (size_t)(2 * t) * Pt::WORDS;
      Pt f = Pt::load(slot);
      f.add(lds.get(tid - 1));
      // complete iff the chain began inside this workgroup (the head ends inside this chunk: it is not `head_right`)
      if (!(flags[tid - 1] & 2u) && !head_right) { f.store(b.buckets + (size_t)k0 * Pt::WORDS); k0 = KEY_INVALID; }
      else f.store(slot);
    }
    if (give) { if (k1 != KEY_INVALID) k1 = KEY_INVALID; else k0 = KEY_INVALID; }
  }
  if (valid) { b.pkeys[2 * t] = k0; b.pkeys[2 * t + 1] = k1; }
}

// The chains that a WORKGROUP edge cuts (last lane of workgroup k / first lanes of workgroup k + 1): one lane per edge joins the
// piece the left workgroup kept for its last lane (its tail, or the scanned sum of a chain that ran off the workgroup) with the
// closing partial on the right -- lane 0 of workgroup k + 1, or, when that lane and its neighbours were transparent and handed
// their pieces on, the first lane behind them that still holds the bucket (looked for among the first EDGE_REACH lanes).
// With uniformly distributed scalars this leaves the partial list empty, so the level-by-level segmented reduction behind
// it has launches but no additions left (it still handles longer chains: skewed scalars).
static constexpr uint32_t EDGE_REACH = 8;
template <class C>
__global__ void __launch_bounds__(64) k_accumulate_edges(AccumulateBody<C> b, uint32_t lanes) {
  PC_LATENCY_KERNEL();
  typedef XyzzD<C> Pt;
  const uint32_t k = blockIdx.x * 64 + threadIdx.x;
  const uint64_t t64 = (uint64_t)(k + 1) * 256;           // first lane of workgroup k + 1
  if (t64 >= lanes) return;
  const uint32_t t = (uint32_t)t64, a = t - 1;
  const uint32_t M = b.offsets[b.g.NB];
  // the left piece: the tail of lane a, or its head when that runs off the chunk (then slot 2a + 1 is unused)
  uint32_t aslot = 2 * a + 1, key = b.pkeys[aslot];
  if (key == KEY_INVALID) {
    aslot = 2 * a; key = b.pkeys[aslot];
    if (key == KEY_INVALID || (uint64_t)b.offsets[key + 1] <= t64 * b.g.T) return;   // ends inside lane a's chunk: not cut by this edge
  }
  // the right piece
  uint32_t u = t;
  for (;; u++) {
    if (u >= lanes || u - t >= EDGE_REACH) return;
    const uint32_t ku = b.pkeys[2 * u];
    if (ku == key) break;
    if (ku != KEY_INVALID) return;
  }
  const uint64_t us = (uint64_t)u * b.g.T;
  const uint32_t ue = (M - (uint32_t)us > b.g.T) ? (uint32_t)us + b.g.T : M;
  uint32_t* slot = b.ppts + (size_t)(2 * u) * Pt::WORDS;
  Pt f = Pt::load(slot);
  f.add(Pt::load(b.ppts + (size_t)aslot * Pt::WORDS));
  b.pkeys[aslot] = KEY_INVALID;
  // complete iff the bucket begins inside the left workgroup (whose scan collected all of it into lane a's piece) and ends
  // inside lane u's chunk
  const uint64_t ws = (uint64_t)(a - 255u) * b.g.T;
  if (b.offsets[key] >= ws && b.offsets[key + 1] <= ue) { f.store(b.buckets + (size_t)key * Pt::WORDS); b.pkeys[2 * u] = KEY_INVALID; }
  else f.store(slot);
}

// The tail of the segmented reduction: levels whose lane count fits one workgroup are walked
// inside a single launch (lane u of level L = thread u, u + 1024, ...), with a workgroup barrier
// between levels instead of a kernel boundary (~60 us each on an otherwise idle stream).
template <class C>
__global__ void __launch_bounds__(256) k_seg_reduce_tail(MsmGeom g, uint32_t level, uint32_t slots, uint32_t* pk0, uint32_t* pk1,
                                                         uint32_t* pp0, uint32_t* pp1, int cur, const uint32_t* offsets,
                                                         uint32_t* buckets) {
  PC_LATENCY_KERNEL();
  uint32_t* pk[2] = {pk0, pk1}; uint32_t* pp[2] = {pp0, pp1};
  for (;;) {
    const uint32_t T2l = level == 1 ? g.T2 : g.T2b;
    const uint32_t lanes2 = (slots + T2l - 1) / T2l;
    SegReduceBody<C> b{g, level, slots, pk[cur], pp[cur], offsets, buckets, pk[cur ^ 1], pp[cur ^ 1]};
    for (uint32_t u = threadIdx.x; u < lanes2; u += blockDim.x) b(u);
    if (lanes2 == 1) break;
    __threadfence_block();
    __syncthreads();
    slots = 2 * lanes2; level++; cur ^= 1;
  }
}

}  // namespace pc
