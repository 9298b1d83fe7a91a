// HipBackend's MSM-specific launchers (templated on the curve): LDS radix sort, accumulate with the
// neighbour merge, the one-launch seg-reduce tail and the cooperative bucket-reduction levels.
#pragma once
#include <stdlib.h>
#include <string.h>
#include "hip_backend.hpp"
#include "msm.hpp"
#include "msm_coop.hpp"
#include "msm_sort.hpp"

namespace pc {
template <class C>
void HipBackend::sort_entries(const MsmGeom& g, const uint32_t* scalars, uint32_t* hist, uint32_t* offsets, uint32_t* cursor,
                              uint32_t* entries) {
  if (sort_mode < 0) { const char* e = getenv("PC_HIP_SORT"); sort_mode = (e && !strcmp(e, "atomic")) ? 0 : 1; }
  if (sort_mode == 0) { sort_entries_atomic<C>(*this, g, scalars, hist, offsets, cursor, entries); return; }
  SortGeom sg = make_sort_geom(g, C::FrP::BITS);
  if (sg.fine_bits > 11 || sg.NC > 16384) {   // wider than the fine pass's LDS histogram, or more bucket sets than the coarse one holds
    sort_entries_atomic<C>(*this, g, scalars, hist, offsets, cursor, entries); return;
  }
  // workspace: G[nblocks][NC] | bintotal[NC+1] | binbase[NC+1] | records[n*W] (8 B each)
  const size_t gw = (size_t)sg.nblocks * sg.NC, nb1 = (size_t)sg.NC + 1;
  const size_t rec_off = ((gw + 2 * nb1) * 4 + 15) & ~(size_t)15;
  const size_t split_off = rec_off + (size_t)g.n * g.Wd * 8;                       // GLV table mode: the pre-split scalars behind the records
  const size_t need = split_off + (g.glv ? (size_t)g.n * GLV_SPLIT_WORDS * 4 : 0);
  if (need > sort_ws_bytes) {
    if (sort_ws) { PC_HIP_CHECK(hipStreamSynchronize(stream)); free(sort_ws); sort_ws = nullptr; sort_ws_bytes = 0; }
    sort_ws = alloc(need); sort_ws_bytes = need;
  }
  uint32_t* G = (uint32_t*)sort_ws; uint32_t* bintotal = G + gw; uint32_t* binbase = bintotal + nb1;
  uint2* records = (uint2*)((char*)sort_ws + rec_off);
  const size_t lds = (size_t)sg.NC * 4;
  uint32_t* split = (uint32_t*)((char*)sort_ws + split_off);
  if (g.glv) {      // k = k1 + k2 lambda once per scalar (Montgomery -> canonical fused): both passes below read the 40-byte records
    GlvPresplitBody<C> b{g, scalars, split};
    launch(b, g.n);
  }
  auto pass = [&](auto scatter_tag, const uint32_t* bb, uint2* rec, int threads) {
    constexpr bool SC = decltype(scatter_tag)::value;
    if (g.glv) {
      if (lds > 64 * 1024) PC_HIP_CHECK(hipFuncSetAttribute((const void*)k_sort_pass<C, SC, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_sort_pass<C, SC, 2>), dim3(sg.nblocks), dim3(threads), lds, stream, sg, (const uint32_t*)split, G, bb, rec);
    } else {
      if (lds > 64 * 1024) PC_HIP_CHECK(hipFuncSetAttribute((const void*)k_sort_pass<C, SC, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_sort_pass<C, SC, 0>), dim3(sg.nblocks), dim3(threads), lds, stream, sg, scalars, G, bb, rec);
    }
    PC_HIP_CHECK(hipGetLastError());
  };
  // 512 workgroups cover the chip twice at most: with 256 lanes each that is 2 waves per SIMD for two passes that are bound
  // by DRAM latency (a scalar load, ~13 LDS atomics and as many 8-byte stores per lane and iteration).  16 waves per workgroup
  // give 8 per SIMD; the LDS histogram (<= 64 KiB) still allows two workgroups per CU.  PC_HIP_SORT_THREADS overrides (tuning).
  static const int threads_env = []() { const char* e = getenv("PC_HIP_SORT_THREADS"); return e ? atoi(e) : 0; }();
  const int sort_threads = threads_env ? threads_env : (lds > 32 * 1024 || g.n >= 65536) ? 1024 : 256;
  pass(std::false_type{}, (const uint32_t*)nullptr, (uint2*)nullptr, sort_threads);
  mark();   // 1: digits + coarse histogram
  hipLaunchKernelGGL(k_sort_binscan, dim3((sg.NC + 15) / 16), dim3(256), 0, stream, G, sg.nblocks, sg.NC, bintotal);
  PC_HIP_CHECK(hipGetLastError());
  exclusive_scan_u32(bintotal, binbase, nb1);
  mark();   // 2: scans
  pass(std::true_type{}, (const uint32_t*)binbase, records, sort_threads);
  hipLaunchKernelGGL(k_sort_fine, dim3(sg.NC), dim3(FT), 0, stream, sg, (const uint32_t*)binbase, (const uint2*)records, entries, offsets);
  PC_HIP_CHECK(hipGetLastError());
  mark();   // 3: coarse scatter + fine sort
}

template <class C>
void HipBackend::accumulate(const AccumulateBody<C>& body, size_t lanes) {
  if (lanes == 0) return;
  const unsigned wgs = (unsigned)((lanes + 255) / 256);
  hipLaunchKernelGGL(k_accumulate<C>, dim3(wgs), dim3(256), 0, stream, body, (uint32_t)lanes);
  PC_HIP_CHECK(hipGetLastError());
  if (wgs > 1) {
    hipLaunchKernelGGL(k_accumulate_edges<C>, dim3((wgs - 1 + 63) / 64), dim3(64), 0, stream, body, (uint32_t)lanes);
    PC_HIP_CHECK(hipGetLastError());
  }
}

template <class C>
void HipBackend::seg_reduce_tail(const MsmGeom& g, uint32_t level, uint32_t slots, uint32_t* const* pk, uint32_t* const* pp, int cur,
                                 const uint32_t* offsets, uint32_t* buckets) {
  hipLaunchKernelGGL(k_seg_reduce_tail<C>, dim3(1), dim3(256), 0, stream, g, level, slots, pk[0], pk[1], pp[0], pp[1], cur, offsets, buckets);
  PC_HIP_CHECK(hipGetLastError());
}

template <class C>
void HipBackend::bucket_level(uint32_t K, uint32_t weight_off, uint32_t cnt, uint32_t n_old, int mode, const uint32_t* x,
                              const uint32_t* old_in, uint32_t* out) {
  if (mode) {   // 16 <= K <= 256: one workgroup per group; mode 2: two lanes per point (K <= 128, latency-bound levels)
    uint32_t lgK = 0; while ((1u << lgK) < K) lgK++;
    size_t lds = (size_t)K * XyzzD<C>::WORDS * 4;
    if (mode == 2 && K <= 128)
      hipLaunchKernelGGL(k_bucket_level_coop2<C>, dim3(cnt * (1 + n_old)), dim3(2 * K), lds, stream, K, lgK, weight_off, cnt, n_old, x,
                         old_in, out);
    else
      hipLaunchKernelGGL(k_bucket_level_coop<C>, dim3(cnt * (1 + n_old)), dim3(K), lds, stream, K, lgK, weight_off, cnt, n_old, x,
                         old_in, out);
    PC_HIP_CHECK(hipGetLastError());
  } else {
    BucketLevelBody<C> b{K, weight_off, cnt, n_old, x, old_in, out};
    launch(b, (size_t)cnt * (1 + n_old));
  }
}
}  // namespace pc
