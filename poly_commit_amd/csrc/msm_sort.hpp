// Bucket sort of the MSM's (window, bucket) entries without global atomics (HIP only).
//
// A two-level MSD radix sort whose counters live in LDS:
//   count    one workgroup per chunk of scalars: signed digits -> LDS histogram over the COARSE
//            bins (window, bucket >> fine_bits); the histogram row is stored to a table G[block][bin]
//   binscan  per bin, exclusive prefix over the blocks; bin totals -> exclusive scan = bin bases
//   scatter  same chunks again: LDS cursors = bin base + this block's prefix; every digit claims
//            its slot with an LDS atomic and writes an 8-byte record {entry, fine bucket}
//   fine     one workgroup per coarse bin: LDS histogram over the <= 1024 fine buckets, LDS scan
//            (which also yields the CSR offsets of those buckets), LDS-atomic placement of the
//            4-byte entries
// Every entry costs two LDS atomics instead of two device-scope atomics (which on this part
// are served at the memory side, ~10 G/s: 2.3 ms of a 8.6 ms MSM at 2^20, 36 of 94 ms at 2^24).
// The result -- `entries` grouped by bucket, `offsets` = CSR row pointers -- is what the
// accumulate kernel consumes; the order inside a bucket is irrelevant (the sum is commutative).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "msm.hpp"

namespace pc {

struct SortGeom {
  uint32_t n, c, W, nb_win, NB, base_off, from_mont, Wd, tbl_stride, m_sub, glv;
  uint32_t fine_bits, cb, ncw /* coarse bins per window */, NC /* total coarse bins */, S /* scalars per block */, nblocks;
  uint32_t top_w, top_fine_bits;   // the last window only uses 2^(tb-1) buckets: it gets its own (smaller) fine width
  const uint64_t* scalar_tab;      // see MsmGeom
};

inline SortGeom make_sort_geom(const MsmGeom& g, uint32_t scalar_bits) {
  SortGeom s;
  s.n = g.n; s.c = g.c; s.W = g.W; s.nb_win = g.nb_win; s.NB = g.NB; s.base_off = g.base_off; s.from_mont = g.from_mont;
  s.Wd = g.Wd; s.tbl_stride = g.tbl_stride; s.m_sub = g.m_sub; s.scalar_tab = g.scalar_tab; s.glv = g.glv;
  uint32_t bbits = g.c - 1;                               // bucket bits per window
  uint32_t cb_max = 0; while ((2u << cb_max) * g.W <= 32768u) cb_max++;
  // fine width: buckets per coarse bin = 2^fine (<= 2048, the LDS histogram of the fine pass).  Fewer, larger coarse bins
  // make the runs a workgroup appends to a bin longer (its 8-byte records then complete 32-byte sectors while they are
  // still in L2); PC_HIP_FINE_BITS overrides (tuning).
  static const uint32_t fine_target = []() { const char* e = getenv("PC_HIP_FINE_BITS"); int v = e ? atoi(e) : 8; return (uint32_t)(v < 4 ? 4 : v > 11 ? 11 : v); }();
  // (two bucket sets -- the GLV table -- with the common fine width would double the coarse bins: 16384 instead of 8192 at c = 22, i.e.
  // half-length runs of 8-byte records per (block, bin) in the scatter pass: one more fine bit keeps the bin count of the plain form)
  const uint32_t ft = fine_target + ((g.glv && !g.m_sub && fine_target < 11) ? 1u : 0u);
  uint32_t cb = bbits > ft ? bbits - ft : 0;
  if (cb > cb_max) cb = cb_max;
  s.cb = cb; s.fine_bits = bbits - cb; s.ncw = 1u << cb; s.NC = g.W * s.ncw;
  // fine_bits can exceed 11 (k_sort_fine's LDS histogram holds 2^11 buckets) when cb clamps to cb_max with many bucket sets
  // (c = 24 with W >= 11; 64 sub-MSMs at c >= 22): HipBackend::sort_entries checks fine_bits / NC and takes the atomic sort then
  // The top window holds tb = bits - (W-1)*c scalar bits, i.e. only 2^tb of its 2^(c-1) buckets are reachable.
  // With the common fine width all of its n entries would land in one or two coarse bins (one
  // workgroup of the fine sort walking n records alone); give it fine width (tb-1) - cb instead so
  // that its buckets spread over up to 2^cb bins.
  s.top_w = g.tbl_stride ? 0xffffffffu : g.W - 1;       // shared buckets (table mode): no window is special
  s.top_fine_bits = s.fine_bits;
  if (!g.tbl_stride) {
    uint32_t tb = scalar_bits > (g.W - 1) * g.c ? scalar_bits - (g.W - 1) * g.c : 0;
    uint32_t used = tb;                                    // its digits are unsigned in [0, 2^tb]: buckets 0 .. 2^tb - 1
    s.top_fine_bits = used > cb ? used - cb : 0;
    if (s.top_fine_bits > s.fine_bits) s.top_fine_bits = s.fine_bits;
  }
  uint32_t nb = (g.n + 2047) / 2048; if (nb > 512) nb = 512; if (nb == 0) nb = 1;
  s.nblocks = nb; s.S = (g.n + nb - 1) / nb;
  return s;
}

// GLV: 0 plain scalars; 2 `scalars` is the pre-split array of k_glv_presplit (GLV_SPLIT_WORDS words per scalar, in call order:
// sub-MSM addressing already resolved)
template <class C, bool SCATTER, int GLV = 0>
__global__ void __launch_bounds__(1024) k_sort_pass(SortGeom sg, const uint32_t* scalars, uint32_t* G, const uint32_t* binbase,
                                                  uint2* records) {
  typedef typename C::FrP FrP;
  extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];
  uint32_t* row = G + (size_t)blockIdx.x * sg.NC;
  for (uint32_t k = threadIdx.x; k < sg.NC; k += blockDim.x) cnt[k] = SCATTER ? binbase[k] + row[k] : 0u;
  __syncthreads();
  const uint32_t lo = blockIdx.x * sg.S;
  const uint32_t hi = (sg.n - lo > sg.S) ? lo + sg.S : sg.n;
  constexpr uint32_t sets = GLV ? 2u : 1u;
  static_assert(GLV == 0 || GLV == 2, "the LDS sort reads plain or pre-split scalars");
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    uint32_t sub = 0, j = i;
    if (sg.m_sub) { sub = i / sg.m_sub; j = i - sub * sg.m_sub; }      // many-MSM mode: bucket set(s) of `sub`
    const uint32_t* sp = GLV == 2 ? scalars + (size_t)i * GLV_SPLIT_WORDS
                                  : sg.scalar_tab ? reinterpret_cast<const uint32_t*>(sg.scalar_tab[sub]) + (size_t)j * FrP::N : scalars + (size_t)i * FrP::N;
    for_each_signed_digit_t<C, GLV>(sg, sp, [&](uint32_t h, uint32_t w, uint32_t mag, uint32_t neg) {
      const uint32_t fb = (w == sg.top_w) ? sg.top_fine_bits : sg.fine_bits;
      uint32_t b = mag - 1, cbin = b >> fb;
      if (cbin >= sg.ncw) cbin = sg.ncw - 1;           // (top window, magnitude 2^(tb-1): one past its range)
      const uint32_t set = sg.m_sub ? sub * sets + h : sg.tbl_stride ? h : w;      // MsmGeom::key_window
      const uint32_t bin = set * sg.ncw + cbin;
      uint32_t pos = atomicAdd(&cnt[bin], 1u);
      const uint32_t base = sg.tbl_stride ? w * sg.tbl_stride + sg.base_off + j : sg.base_off + j;
      if (SCATTER) records[pos] = make_uint2(base | (neg << 31), b - (cbin << fb));
    });
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < sg.NC; k += blockDim.x) row[k] = cnt[k];
  }
}

// per coarse bin: exclusive prefix over the blocks (in place), total -> bintotal[bin].
// A workgroup covers 16 bins x 16 segments of the block range: each thread sums its segment, the 16
// segment sums of a bin are scanned through LDS, then the segment is walked again to write the
// prefixes -- a dependent chain of 2 * ceil(nblocks / 16) loads instead of nblocks (0.12 -> 0.02 ms).
static __global__ void __launch_bounds__(256) k_sort_binscan(uint32_t* G, uint32_t nblocks, uint32_t NC, uint32_t* bintotal) {
  __shared__ uint32_t seg_sum[16][17];
  const uint32_t kb = threadIdx.x & 15, seg = threadIdx.x >> 4;
  const uint32_t k = blockIdx.x * 16 + kb;
  const uint32_t L = (nblocks + 15) / 16;
  const uint32_t b0 = seg * L, b1 = (b0 + L < nblocks) ? b0 + L : nblocks;
  uint32_t sum = 0;
  if (k < NC) for (uint32_t b = b0; b < b1; b++) sum += G[(size_t)b * NC + k];
  seg_sum[seg][kb] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (uint32_t t = 0; t < seg; t++) run += seg_sum[t][kb];
  if (k < NC) {
    for (uint32_t b = b0; b < b1; b++) {
      uint32_t v = G[(size_t)b * NC + k];
      G[(size_t)b * NC + k] = run;
      run += v;
    }
    if (seg == 15) bintotal[k] = run;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) bintotal[NC] = 0;
}

// 512 lanes: a bin of 24.5 k records (2^24 pairs, c = 22) is 12 iterations of four records per lane; measured 2.39 ->
// 1.7 ms for the fine pass at 2^24 against 256 lanes
#ifndef PC_SORT_FINE_THREADS
#define PC_SORT_FINE_THREADS 512
#endif
static constexpr uint32_t FT = PC_SORT_FINE_THREADS;      // lanes of the fine pass (one workgroup per coarse bin)
static __global__ void __launch_bounds__(PC_SORT_FINE_THREADS) k_sort_fine(SortGeom sg, const uint32_t* binbase, const uint2* records, uint32_t* entries,
                                                  uint32_t* offsets) {
  __shared__ uint32_t h[4096];        // [0, F): counts / cursors; [F, 2F): scan ping-pong   (F <= 2048)
  const uint32_t k = blockIdx.x, start = binbase[k], end = binbase[k + 1];
  const uint32_t w = k / sg.ncw, cbin = k % sg.ncw;
  const uint32_t fb = (w == sg.top_w) ? sg.top_fine_bits : sg.fine_bits;
  const uint32_t F = 1u << fb;
  for (uint32_t f = threadIdx.x; f < F; f += FT) h[f] = 0;
  __syncthreads();
  // (four records in flight per lane: one load per iteration left the pass waiting on DRAM latency -- a bin is walked by
  // one workgroup, ~100 iterations per lane at 2^24)
  {
    uint32_t r = start + threadIdx.x;
    for (; r + 3 * FT < end; r += 4 * FT) {
      const uint32_t f0 = records[r].y, f1 = records[r + FT].y, f2 = records[r + 2 * FT].y, f3 = records[r + 3 * FT].y;
      atomicAdd(&h[f0], 1u); atomicAdd(&h[f1], 1u); atomicAdd(&h[f2], 1u); atomicAdd(&h[f3], 1u);
    }
    for (; r < end; r += FT) atomicAdd(&h[records[r].y], 1u);
  }
  __syncthreads();
  // inclusive Hillis-Steele scan over F counters, ping-pong between h[0..F) and h[F..2F)
  uint32_t src = 0;
  for (uint32_t d = 1; d < F; d <<= 1) {
    for (uint32_t f = threadIdx.x; f < F; f += FT) {
      uint32_t v = h[src + f];
      if (f >= d) v += h[src + f - d];
      h[(src ^ F) + f] = v;
    }
    __syncthreads();
    src ^= F;
  }
  // exclusive offsets -> CSR row pointers of this bin's buckets, and the placement cursors
  const uint32_t key0 = w * sg.nb_win + (cbin << fb);
  uint32_t excl[2048 / FT];           // F / FT values per lane
  uint32_t q = 0;
  for (uint32_t f = threadIdx.x; f < F; f += FT, q++) excl[q] = start + (f ? h[src + f - 1] : 0u);
  __syncthreads();
  q = 0;
  for (uint32_t f = threadIdx.x; f < F; f += FT, q++) { h[f] = excl[q]; offsets[key0 + f] = excl[q]; }
  if (cbin + 1 == sg.ncw) {
    // buckets of this window beyond the last bin's range (top window only) are empty: their row
    // pointers equal the end of the window
    for (uint32_t kk = key0 + F + threadIdx.x; kk < (w + 1) * sg.nb_win; kk += FT) offsets[kk] = end;
    if (k + 1 == sg.NC && threadIdx.x == 0) offsets[sg.NB] = end;
  }
  __syncthreads();
  {
    uint32_t r = start + threadIdx.x;
    for (; r + 3 * FT < end; r += 4 * FT) {
      const uint2 a = records[r], b = records[r + FT], c = records[r + 2 * FT], d = records[r + 3 * FT];
      const uint32_t pa = atomicAdd(&h[a.y], 1u), pb = atomicAdd(&h[b.y], 1u), pc = atomicAdd(&h[c.y], 1u), pd = atomicAdd(&h[d.y], 1u);
      entries[pa] = a.x; entries[pb] = b.x; entries[pc] = c.x; entries[pd] = d.x;
    }
    for (; r < end; r += FT) {
      uint2 rec = records[r];
      uint32_t pos = atomicAdd(&h[rec.y], 1u);
      entries[pos] = rec.x;
    }
  }
}

}  // namespace pc
