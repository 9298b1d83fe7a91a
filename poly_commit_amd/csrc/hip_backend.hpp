// HIP execution backend for the orchestration templates (msm.hpp, ...): one stream (plus an optional
// low-priority tail stream), every kernel body launched through a single generic __global__ wrapper,
// the u32 scan as three small kernels of its own.
#pragma once
#include <type_traits>
#include <string.h>
#include <hip/hip_runtime.h>
#include <stdexcept>
#include <string>
#include <vector>
#include <mutex>
#include <unordered_map>

namespace pc {

template <class C> struct AccumulateBody;

struct HipError : std::runtime_error {
  hipError_t code;
  HipError(hipError_t c, const char* what) : std::runtime_error(std::string(what) + ": " + hipGetErrorString(c)), code(c) {}
};
#define PC_HIP_CHECK(expr)                                        \
  do {                                                            \
    hipError_t _e = (expr);                                       \
    if (_e != hipSuccess) throw ::pc::HipError(_e, #expr);        \
  } while (0)

// Every hipMalloc / hipFree of the library goes through these two: per-device totals of what it holds (pc_hip_ctx_bytes_resident).
struct DevMemLedger {
  std::mutex mu; std::unordered_map<void*, std::pair<int, size_t>> live; std::unordered_map<int, size_t> total;
};
inline DevMemLedger& dev_mem_ledger() { static DevMemLedger l; return l; }
inline hipError_t dev_malloc(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) {
    int dev = 0; (void)hipGetDevice(&dev);
    DevMemLedger& l = dev_mem_ledger(); std::lock_guard<std::mutex> lk(l.mu);
    l.live[*p] = {dev, bytes}; l.total[dev] += bytes;
  }
  return e;
}
inline size_t dev_free(void* p) {        // returns the size that was recorded for p
  size_t bytes = 0;
  if (!p) return 0;
  { DevMemLedger& l = dev_mem_ledger(); std::lock_guard<std::mutex> lk(l.mu);
    auto it = l.live.find(p);
    if (it != l.live.end()) { bytes = it->second.second; l.total[it->second.first] -= bytes; l.live.erase(it); } }
  (void)hipFree(p);
  return bytes;
}
inline size_t dev_bytes_held(int device) { DevMemLedger& l = dev_mem_ledger(); std::lock_guard<std::mutex> lk(l.mu); auto it = l.total.find(device); return it == l.total.end() ? 0 : it->second; }

// The short, latency-bound kernels around the bucket accumulation (division scan levels, bucket / segmented reduction
// levels, cooperative reductions; a k_run body opts in, see body_latency_bound) usually share the chip with an accumulation of another pipeline whose waves issue
// multiply-adds back to back: they raise their waves' issue priority so that they finish while the accumulation runs
// instead of crawling beside it (pipelined trace at 2^20 before: division 2.6 ms instead of 0.6, reduction levels 3 ms
// instead of 0.3, and the open's sort waiting for the division): commit+open 6.58 -> 5.78 ms at 2^20, 64 x 2^20 batch
// 100.7 -> 93.9 ms, 2^24 unchanged.  The sort passes stay at the default priority: raising them too cost 4 % at 2^24
// (74.4 vs 71.6 ms) and gained nothing at 2^20.
#if defined(__HIP_DEVICE_COMPILE__)
#define PC_LATENCY_KERNEL() __builtin_amdgcn_s_setprio(3)
#else
#define PC_LATENCY_KERNEL() ((void)0)
#endif

// a kernel body opts in with `static constexpr bool LATENCY_BOUND = true;` (the heavy streaming bodies -- digits, histogram,
// scatter of the table-free sort, table builds, hashes -- must not: prioritised, they starve the accumulation: the table-free
// 2^24 step went from 108.6 to 138.4 ms when every k_run kernel raised its priority)
template <class B, class = void> struct body_latency_bound : std::false_type {};
template <class B> struct body_latency_bound<B, std::void_t<decltype(B::LATENCY_BOUND)>> : std::bool_constant<B::LATENCY_BOUND> {};

template <class Body>
__global__ void __launch_bounds__(256) k_run(Body body, uint32_t lanes) {
  if constexpr (body_latency_bound<Body>::value) PC_LATENCY_KERNEL();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < lanes) body(i);
}

// ---- exclusive scan of u32 (three small kernels; n up to 2^32) --------------------------------
// tile = 1024 elements per workgroup (256 lanes x 4): tile sums -> scan of the tile sums (recursive
// for > 1M tiles, which never happens here) -> per-tile exclusive scan seeded with the tile base.
static constexpr uint32_t SCAN_TILE = 1024;
static __device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t* lds /* 256 */, uint32_t& total) {
  const uint32_t tid = threadIdx.x;
  lds[tid] = v;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t x = tid >= d ? lds[tid - d] : 0u;
    __syncthreads();
    lds[tid] += x;
    __syncthreads();
  }
  total = lds[255];
  return lds[tid] - v;
}
static __global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t* in, size_t n, uint32_t* tile_sums) {
  __shared__ uint32_t lds[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  uint32_t s = 0;
  for (int k = 0; k < 4; k++) if (base + k < n) s += in[base + k];
  uint32_t total; (void)wg_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
static __global__ void __launch_bounds__(256) k_scan_small(uint32_t* v, uint32_t n) {   // in place, n <= 2^20, one workgroup
  __shared__ uint32_t lds[256];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += SCAN_TILE) {
    uint32_t a[4]; uint32_t s = 0;
    for (int k = 0; k < 4; k++) { uint32_t i = base + threadIdx.x * 4 + k; a[k] = i < n ? v[i] : 0u; s += a[k]; }
    uint32_t total; uint32_t ex = wg_exclusive_scan(s, lds, total) + carry;
    for (int k = 0; k < 4; k++) { uint32_t i = base + threadIdx.x * 4 + k; if (i < n) v[i] = ex; ex += a[k]; }
    carry += total;
    __syncthreads();
  }
}
static __global__ void __launch_bounds__(256) k_scan_tiles(const uint32_t* in, size_t n, const uint32_t* tile_base, uint32_t* out) {
  __shared__ uint32_t lds[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  uint32_t a[4]; uint32_t s = 0;
  for (int k = 0; k < 4; k++) { a[k] = base + k < n ? in[base + k] : 0u; s += a[k]; }
  uint32_t total; uint32_t ex = wg_exclusive_scan(s, lds, total) + tile_base[blockIdx.x];
  for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = ex; ex += a[k]; }
}

struct HipBackend {
  hipStream_t stream = nullptr;
  void* scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  // optional phase timing
  static constexpr int MAX_EV = 16;
  hipEvent_t ev[MAX_EV];
  hipEvent_t done = nullptr;
  int n_ev = 0;
  bool timing = false;

  // cu_part / cu_parts: restrict this backend's stream to one interleaved share of the CUs
  // (every cu_parts-th CU).  Pipelines that each own a share run side by side, so the
  // latency-bound tail of one MSM overlaps the bucket accumulation of the others even though a
  // 218-VGPR accumulate wave leaves no room for a second kernel on the same SIMD.
  void init(int cu_part = 0, int cu_parts = 1) {
    if (cu_parts > 1) {
      hipDeviceProp_t prop; int dev = 0;
      PC_HIP_CHECK(hipGetDevice(&dev));
      PC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
      for (int cu = 0; cu < ncu; cu++)
        if ((cu / 2) % cu_parts == cu_part) mask[cu / 32] |= 1u << (cu % 32);   // pairs of CUs (a WGP-like unit) stay together
      PC_HIP_CHECK(hipExtStreamCreateWithCUMask(&stream, (uint32_t)mask.size(), mask.data()));
    } else if (tail_split) {
      // Two queues per pipeline: sort + accumulate at normal priority, the
      // latency-bound reductions at low priority, so that a bucket accumulation arriving at a busy chip gets the CUs first.
      int lo = 0, hi = 0;
      PC_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      (void)hi;    // (raising the main queues to `hi` as well measured the same at 2^20 and 4 % slower at 2^22)
      int tail_prio = lo, main_prio = 0;
      if (const char* e = getenv("PC_HIP_TAIL_PRIO")) tail_prio = !strcmp(e, "hi") ? hi : !strcmp(e, "lo") ? lo : 0;   // tuning experiments
      if (const char* e = getenv("PC_HIP_MAIN_PRIO")) main_prio = !strcmp(e, "hi") ? hi : !strcmp(e, "lo") ? lo : 0;
      PC_HIP_CHECK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, main_prio));
      PC_HIP_CHECK(hipStreamCreateWithPriority(&tail_stream, hipStreamNonBlocking, tail_prio));
      main_prio_ = main_prio; tail_prio_ = tail_prio;
      PC_HIP_CHECK(hipEventCreateWithFlags(&tail_ev, hipEventDisableTiming));
      main_stream = stream;
    } else {
      PC_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    for (int i = 0; i < MAX_EV; i++) PC_HIP_CHECK(hipEventCreate(&ev[i]));
    PC_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  }
  // After a stream capture that did not end cleanly (curve_ops_impl.hpp: a captured call that threw): a queue the runtime still
  // counts as capturing refuses every further launch ("operation failed due to a previous error during capture").  Such a queue
  // is replaced; its real work, if any, finishes on the old one (hipStreamDestroy releases it when idle).
  void replace_capturing_streams() {
    auto stuck = [](hipStream_t q) {
      if (!q) return false;
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      const hipError_t e = hipStreamIsCapturing(q, &st);
      (void)hipGetLastError();
      return e != hipSuccess || st != hipStreamCaptureStatusNone;
    };
    if (!tail_split) return;
    if (stuck(tail_stream)) { (void)hipStreamDestroy(tail_stream); tail_stream = nullptr; PC_HIP_CHECK(hipStreamCreateWithPriority(&tail_stream, hipStreamNonBlocking, tail_prio_)); }
    if (stuck(main_stream)) {
      const bool on_main = stream == main_stream;
      (void)hipStreamDestroy(main_stream); main_stream = nullptr;
      PC_HIP_CHECK(hipStreamCreateWithPriority(&main_stream, hipStreamNonBlocking, main_prio_));
      if (on_main) stream = main_stream;
    }
    if (stream != main_stream && stream != tail_stream && stream != aux_stream) stream = main_stream;
  }
  int main_prio_ = 0, tail_prio_ = 0;
  void destroy() {
    trim();
    for (int i = 0; i < MAX_EV; i++) (void)hipEventDestroy(ev[i]);
    if (done) (void)hipEventDestroy(done);
    if (main_stream) stream = main_stream;
    if (tail_stream) (void)hipStreamDestroy(tail_stream);
    if (tail_ev) (void)hipEventDestroy(tail_ev);
    if (aux_saved) { stream = aux_saved; aux_saved = nullptr; }
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
    for (int i = 0; i < N_TOK; i++) if (tok_ev[i]) (void)hipEventDestroy(tok_ev[i]);
    if (stream) (void)hipStreamDestroy(stream);
  }
  // release every grow-only scratch buffer (they come back on demand); the stream must be idle
  void trim() {
    free(scan_tmp); scan_tmp = nullptr; scan_tmp_bytes = 0;
    free(sort_ws); sort_ws = nullptr; sort_ws_bytes = 0;
    free(work_ws); work_ws = nullptr; work_ws_bytes = 0;
    for (int i = 0; i < 2; i++) { free(stage_ws[i]); stage_ws[i] = nullptr; stage_bytes[i] = 0; }
  }
  size_t scratch_bytes() const { return scan_tmp_bytes + sort_ws_bytes + work_ws_bytes + stage_bytes[0] + stage_bytes[1]; }
  void mark() { if (timing && marks_on && n_ev < MAX_EV) PC_HIP_CHECK(hipEventRecord(ev[n_ev++], stream)); }
  bool marks_on = true;
  bool timing_marks(bool on) { const bool was = marks_on; marks_on = on; return was; }      // suppress the phase marks of inner steps (an MSM in parts)

  // ---- auxiliary queue of a pipeline (MsmPlan::add_part): the sort of the next part of an MSM, and the copy of its scalars, run here
  // beside the accumulation of the current part on the main queue.  Tokens are events from a small ring; -1 = nothing to wait for.
  static constexpr int N_TOK = 16;
  hipStream_t aux_stream = nullptr, aux_saved = nullptr;
  hipEvent_t tok_ev[N_TOK] = {};
  int tok_next = 0;
  int new_token(hipStream_t s) {
    const int t = tok_next; tok_next = (tok_next + 1) % N_TOK;
    if (!tok_ev[t]) PC_HIP_CHECK(hipEventCreateWithFlags(&tok_ev[t], hipEventDisableTiming));
    PC_HIP_CHECK(hipEventRecord(tok_ev[t], s));
    return t;
  }
  void aux_begin(int wait_a, int wait_b) {      // everything queued until aux_end() goes to the auxiliary queue, after the two tokens
    if (!aux_stream) PC_HIP_CHECK(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
    if (wait_a >= 0) PC_HIP_CHECK(hipStreamWaitEvent(aux_stream, tok_ev[wait_a], 0));
    if (wait_b >= 0) PC_HIP_CHECK(hipStreamWaitEvent(aux_stream, tok_ev[wait_b], 0));
    aux_saved = stream; stream = aux_stream;      // LAST: a failing wait above must not leave the backend on the auxiliary queue (the callers' scope guards are built after this call)
  }
  // a call that failed half way (between the parts of an MSM in parts): nothing of it may still run when the pipeline is used again
  void quiesce() {
    if (aux_saved) { stream = aux_saved; aux_saved = nullptr; }
    if (tail_stream) stream = main_stream;
    if (aux_stream) (void)hipStreamSynchronize(aux_stream);
    if (tail_stream) (void)hipStreamSynchronize(tail_stream);
    (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
  }
  int aux_end() { stream = aux_saved; aux_saved = nullptr; return new_token(aux_stream); }      // (the main queue is back before anything can throw)
  void wait_token(int t) { if (t >= 0) PC_HIP_CHECK(hipStreamWaitEvent(stream, tok_ev[t], 0)); }
  int main_token() { return new_token(stream); }

  size_t bytes_live = 0;      // device bytes currently allocated through THIS backend (a pipeline's workspace, a context's buffers)
  void* alloc(size_t bytes) { void* p = nullptr; PC_HIP_CHECK(dev_malloc(&p, bytes ? bytes : 4)); bytes_live += bytes ? bytes : 4; return p; }
  void free(void* p) { if (p) { bytes_live -= dev_free(p); free_epoch++; } }
  // bumped by every device free through this backend: a captured hipGraph (curve_ops_impl.hpp) holds the addresses of the grow-only
  // scratch buffers (sort, scan, workspace) as they were at capture time; once ANY of them was given back the graph must not be
  // replayed (round 6: a memory access fault in the second opening of a process, where the pipelines meet other call sizes in
  // another order and the sort scratch of one grew after a smaller call had been captured on it)
  uint64_t free_epoch = 0;
  // Grow-only scratch for the short kernels of one call (division scan levels): hipMalloc/hipFree per
  // call would synchronise the whole device and drain the MSM pipelines running on other streams.
  void* workspace(size_t bytes) {
    if (bytes > work_ws_bytes) {
      if (work_ws) { PC_HIP_CHECK(hipStreamSynchronize(stream)); free(work_ws); work_ws = nullptr; work_ws_bytes = 0; }
      work_ws = alloc(bytes); work_ws_bytes = bytes;
    }
    return work_ws;
  }
  // Grow-only device copies of the host arguments / results of one call (slot 0: input, 1: output): the entry points that take
  // host memory (the trait-shaped open hands its polynomial over on the host) paid a hipMalloc + a device-synchronising hipFree
  // per call.  Kept up to STAGE_KEEP bytes per slot; larger requests stay transient (caller allocates).
  static constexpr size_t STAGE_KEEP = (size_t)1 << 30;
  void* stage(int slot, size_t bytes) {
    if (bytes > stage_bytes[slot]) {
      if (stage_ws[slot]) { PC_HIP_CHECK(hipStreamSynchronize(stream)); free(stage_ws[slot]); stage_ws[slot] = nullptr; stage_bytes[slot] = 0; }
      stage_ws[slot] = alloc(bytes ? bytes : 4); stage_bytes[slot] = bytes ? bytes : 4;
    }
    return stage_ws[slot];
  }
  void memset(void* p, int v, size_t bytes) { PC_HIP_CHECK(hipMemsetAsync(p, v, bytes, stream)); }
  void copy_d2d(void* d, const void* s, size_t bytes) { PC_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, stream)); }
  void copy_h2d(void* d, const void* s, size_t bytes) { PC_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream)); }
  void copy_d2h(void* d, const void* s, size_t bytes) {
    PC_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream));
    PC_HIP_CHECK(hipStreamSynchronize(stream));
  }
  void sync() { PC_HIP_CHECK(hipStreamSynchronize(stream)); }
  void copy_d2h_async(void* d, const void* s, size_t bytes) { PC_HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream)); }
  void* alloc_host(size_t bytes) { void* p = nullptr; PC_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault)); return p; }
  void free_host(void* p) { if (p) (void)hipHostFree(p); }
  // everything launched between begin_tail() and end_tail() goes to the low-priority queue, ordered
  // after what was queued before
  void begin_tail() {
    if (!tail_stream) return;
    PC_HIP_CHECK(hipEventRecord(tail_ev, main_stream));
    PC_HIP_CHECK(hipStreamWaitEvent(tail_stream, tail_ev, 0));
    stream = tail_stream;
  }
  void end_tail() { if (tail_stream) stream = main_stream; }
  void record_done() { PC_HIP_CHECK(hipEventRecord(done, stream)); }
  void wait_done() { PC_HIP_CHECK(hipEventSynchronize(done)); }

  void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
    if (n == 0) return;
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles > (1u << 20)) throw std::runtime_error("exclusive_scan_u32: input too large");
    if (tiles * 4 > scan_tmp_bytes) {
      if (scan_tmp) { PC_HIP_CHECK(hipStreamSynchronize(stream)); free(scan_tmp); scan_tmp = nullptr; }
      scan_tmp = alloc(tiles * 4); scan_tmp_bytes = tiles * 4;
    }
    uint32_t* sums = (uint32_t*)scan_tmp;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)tiles), dim3(256), 0, stream, in, n, sums);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, stream, sums, (uint32_t)tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(256), 0, stream, in, n, (const uint32_t*)sums, out);
    PC_HIP_CHECK(hipGetLastError());
  }

  // steps 1-3 of the MSM: LDS radix sort (msm_sort.hpp) or the atomic reference sort
  template <class C>
  void sort_entries(const struct MsmGeom& g, const uint32_t* scalars, uint32_t* hist, uint32_t* offsets, uint32_t* cursor,
                    uint32_t* entries);
  void* sort_ws = nullptr; size_t sort_ws_bytes = 0;
  void* work_ws = nullptr; size_t work_ws_bytes = 0;
  void* stage_ws[2] = {nullptr, nullptr}; size_t stage_bytes[2] = {0, 0};
  int sort_mode = -1;   // -1 = read PC_HIP_SORT on first use; 0 = atomic; 1 = LDS radix
  bool tail_split = false; hipStream_t main_stream = nullptr, tail_stream = nullptr; hipEvent_t tail_ev = nullptr;

  // bucket accumulation with the neighbour merge of cut runs (msm_coop.hpp)
  template <class C>
  void accumulate(const struct AccumulateBody<C>& body, size_t lanes);

  // all remaining (small) levels of the segmented reduction in one launch
  template <class C>
  void seg_reduce_tail(const struct MsmGeom& g, uint32_t level, uint32_t slots, uint32_t* const* pk, uint32_t* const* pp, int cur,
                       const uint32_t* offsets, uint32_t* buckets);

  // one level of the bucket reduction (see BucketLevelBody / k_bucket_level_coop)
  template <class C>
  void bucket_level(uint32_t K, uint32_t weight_off, uint32_t cnt, uint32_t n_old, int mode, const uint32_t* x,
                    const uint32_t* old_in, uint32_t* out);

  template <class Body>
  void launch(const Body& body, size_t lanes, int block = 256) {
    if (lanes == 0) return;
    dim3 grid((unsigned)((lanes + block - 1) / block));
    hipLaunchKernelGGL(k_run<Body>, grid, dim3(block), 0, stream, body, (uint32_t)lanes);
    PC_HIP_CHECK(hipGetLastError());
  }
};

}  // namespace pc
