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

// Bucket accumulation with an in-workgroup merge of the runs that chunk edges cut in two.
// AccumulateBody::chunk leaves, per lane, a partial for its first run (bucket began in an earlier
// chunk) and one for its last run (bucket continues in the next chunk).  With ~32 entries per
// bucket and 64 per chunk nearly every bucket is cut exactly once, i.e. its two halves sit in
// neighbouring lanes: lane t hands its last-run sum to lane t+1 through LDS, lane t+1 adds it to
// its first-run partial and, if the bucket ends inside its chunk, writes the finished bucket.
// Only buckets spanning >= 3 chunks or a workgroup edge still go through the partial list.
#ifndef PC_ACC_WAVES_PER_EU
#define PC_ACC_WAVES_PER_EU 0
#endif
#if PC_ACC_WAVES_PER_EU
#define PC_ACC_BOUNDS __launch_bounds__(256, PC_ACC_WAVES_PER_EU)
#else
#define PC_ACC_BOUNDS __launch_bounds__(256)
#endif
template <class C>
__global__ void PC_ACC_BOUNDS k_accumulate(AccumulateBody<C> b, uint32_t lanes) {
  typedef XyzzD<C> Pt;
  __shared__ uint32_t xch[Pt::WORDS * 256];
  __shared__ uint32_t key_offer[256], key_first[256];
  const uint32_t tid = threadIdx.x, t = blockIdx.x * 256 + tid;
  const bool valid = t < lanes;
  uint32_t k0 = KEY_INVALID, k1 = KEY_INVALID;
  Pt last = Pt::infinity();
  if (valid) b.chunk(t, k0, k1, last);
  LdsPoints<C> lds{xch, 256};
  key_offer[tid] = k1; key_first[tid] = k0;
  if (k1 != KEY_INVALID) lds.put(tid, last);
  __syncthreads();
  const bool give = k1 != KEY_INVALID && tid < 255 && key_first[tid + 1] == k1;
  const bool take = k0 != KEY_INVALID && tid > 0 && key_offer[tid - 1] == k0;
  if (take) {
    uint32_t* slot = b.ppts + (size_t)(2 * t) * Pt::WORDS;
    Pt f = Pt::load(slot);
    f.add(lds.get(tid - 1));
    // the merged sum covers the bucket from its start (inside lane t-1's chunk) to where lane t's
    // first run stopped: complete iff the bucket ends inside this chunk
    const uint32_t M = b.offsets[b.g.NB];
    const uint64_t s64 = (uint64_t)t * b.g.T;
    const uint32_t e = (M - (uint32_t)s64 > b.g.T) ? (uint32_t)s64 + b.g.T : M;
    if (b.offsets[k0 + 1] <= e) { f.store(b.buckets + (size_t)k0 * Pt::WORDS); k0 = KEY_INVALID; }
    else f.store(slot);
  }
  if (give) k1 = KEY_INVALID;
  if (valid) { b.pkeys[2 * t] = k0; b.pkeys[2 * t + 1] = k1; }
}

// The runs that a WORKGROUP edge cut (lane 255 of workgroup k / lane 0 of workgroup k+1) are the only cut runs the
// in-workgroup merge above cannot see: one lane per edge does the same merge right after the accumulation.  With
// uniformly distributed scalars this leaves the partial list empty, so the level-by-level segmented reduction behind
// it has launches but no additions left (it still handles buckets spanning three or more chunks: skewed scalars).
template <class C>
__global__ void __launch_bounds__(64) k_accumulate_edges(AccumulateBody<C> b, uint32_t lanes) {
  PC_LATENCY_KERNEL();
  typedef XyzzD<C> Pt;
  const uint32_t k = blockIdx.x * 64 + threadIdx.x;
  const uint64_t t64 = (uint64_t)(k + 1) * 256;           // first lane of workgroup k + 1
  if (t64 >= lanes) return;
  const uint32_t t = (uint32_t)t64, a = t - 1;
  const uint32_t k1 = b.pkeys[2 * a + 1], k0 = b.pkeys[2 * t];
  if (k1 == KEY_INVALID || k1 != k0) return;
  uint32_t* slot = b.ppts + (size_t)(2 * t) * Pt::WORDS;
  Pt f = Pt::load(slot);
  f.add(Pt::load(b.ppts + (size_t)(2 * a + 1) * Pt::WORDS));
  const uint32_t M = b.offsets[b.g.NB];
  const uint64_t s64 = (uint64_t)t * b.g.T;
  const uint32_t e = (M - (uint32_t)s64 > b.g.T) ? (uint32_t)s64 + b.g.T : M;
  b.pkeys[2 * a + 1] = KEY_INVALID;
  if (b.offsets[k0 + 1] <= e) { f.store(b.buckets + (size_t)k0 * Pt::WORDS); b.pkeys[2 * t] = KEY_INVALID; }
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
