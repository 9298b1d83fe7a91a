// HIP execution backend for the orchestration templates (msm.hpp, ...): one stream, every
// kernel body launched through a single generic __global__ wrapper, scans through hipCUB.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdexcept>
#include <string>
#include <vector>

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

template <class Body>
__global__ void __launch_bounds__(256) k_run(Body body, uint32_t lanes) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < lanes) body(i);
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
    } else {
      PC_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    for (int i = 0; i < MAX_EV; i++) PC_HIP_CHECK(hipEventCreate(&ev[i]));
    PC_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  }
  void destroy() {
    if (scan_tmp) (void)hipFree(scan_tmp);
    if (sort_ws) (void)hipFree(sort_ws);
    for (int i = 0; i < MAX_EV; i++) (void)hipEventDestroy(ev[i]);
    if (done) (void)hipEventDestroy(done);
    if (stream) (void)hipStreamDestroy(stream);
  }
  void mark() { if (timing && n_ev < MAX_EV) PC_HIP_CHECK(hipEventRecord(ev[n_ev++], stream)); }

  void* alloc(size_t bytes) { void* p = nullptr; PC_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 4)); return p; }
  void free(void* p) { if (p) (void)hipFree(p); }
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
  void record_done() { PC_HIP_CHECK(hipEventRecord(done, stream)); }
  void wait_done() { PC_HIP_CHECK(hipEventSynchronize(done)); }

  void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
    size_t need = 0;
    PC_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, (int)n, stream));
    if (need > scan_tmp_bytes) {
      if (scan_tmp) { PC_HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(scan_tmp); }
      PC_HIP_CHECK(hipMalloc(&scan_tmp, need)); scan_tmp_bytes = need;
    }
    PC_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, need, in, out, (int)n, stream));
  }

  // steps 1-3 of the MSM: LDS radix sort (msm_sort.hpp) or the atomic reference sort
  template <class C>
  void sort_entries(const struct MsmGeom& g, const uint32_t* scalars, uint32_t* hist, uint32_t* offsets, uint32_t* cursor,
                    uint32_t* entries);
  void* sort_ws = nullptr; size_t sort_ws_bytes = 0;
  int sort_mode = -1;   // -1 = read PC_HIP_SORT on first use; 0 = atomic; 1 = LDS radix

  // bucket accumulation with the neighbour merge of cut runs (msm_coop.hpp)
  template <class C>
  void accumulate(const struct AccumulateBody<C>& body, size_t lanes);

  // all remaining (small) levels of the segmented reduction in one launch
  template <class C>
  void seg_reduce_tail(const struct MsmGeom& g, uint32_t level, uint32_t slots, uint32_t* const* pk, uint32_t* const* pp, int cur,
                       const uint32_t* offsets, uint32_t* buckets);

  // one level of the bucket reduction (see BucketLevelBody / k_bucket_level_coop)
  template <class C>
  void bucket_level(uint32_t K, uint32_t weight_off, uint32_t cnt, uint32_t n_old, const uint32_t* x, const uint32_t* old_in,
                    uint32_t* out);

  template <class Body>
  void launch(const Body& body, size_t lanes, int block = 256) {
    if (lanes == 0) return;
    dim3 grid((unsigned)((lanes + block - 1) / block));
    hipLaunchKernelGGL(k_run<Body>, grid, dim3(block), 0, stream, body, (uint32_t)lanes);
    PC_HIP_CHECK(hipGetLastError());
  }
};

}  // namespace pc
