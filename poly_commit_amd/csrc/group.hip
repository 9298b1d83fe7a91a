// Multi-GPU entry points of the C ABI (include/pc_hip.h, "pc_hip_group_*"): one committer key sharded over the
// GPUs of a node in contiguous chunks, driven from ONE process -- what a Rust prover holding a
// single CommitterKey needs.  A group is N single-device contexts; every call fans out to one host thread per
// device, each of which runs the complete single-device path on its chunk (full Pippenger -> ONE affine point),
// and the N partial points are added on the host (pc_hip_points_sum): the exchange is N * 96 bytes, so no
// device-to-device collective is involved -- partial BUCKET arrays are never moved (75 MB per GPU at 2^20).
// The one-process-per-GPU form of the same protocol (torch.distributed / RCCL all_gather of the partial
// points) is poly_commit_amd/sharded.py, which bench.py --gpus N uses.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <string.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include "../../include/pc_hip.h"
#include "host_tail.hpp"

// Persistent threads per device, each with a FIFO of tasks (round 2 spawned and joined N std::threads per call): the WORKER queues
// work on the device (every group call hands its per-device part to it), the COPIER brings a job's host coefficients in, the
// REAPER waits for queued MSMs.  Tasks of one queue run in order; different queues and devices run concurrently.
struct DeviceWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  void start() {
    th = std::thread([this]() {
      for (;;) {
        std::function<void()> fn;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [this]() { return stop || !q.empty(); });
          if (q.empty()) return;
          fn = std::move(q.front()); q.pop_front();
        }
        fn();
      }
    });
  }
  void push(std::function<void()> fn) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(fn)); } cv.notify_one(); }
  // ahead of everything queued: the next phase of an OLDER job goes before a younger job's first phase
  void push_front(std::function<void()> fn) { { std::lock_guard<std::mutex> lk(mu); q.push_front(std::move(fn)); } cv.notify_one(); }
  void shutdown() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_one(); if (th.joinable()) th.join(); }
};

// Grow-only device buffers of one device for the commit+open jobs: a ring of coefficient shards and quotients, so that a
// queued MSM can still read its scalars while the next job's shard is being copied in (a per-call hipMalloc / hipFree
// synchronises the whole device and drains the other pipelines).
static constexpr int GROUP_RING = 3;
struct DeviceBufs {
  void* c[GROUP_RING] = {nullptr, nullptr, nullptr};
  void* q[GROUP_RING] = {nullptr, nullptr, nullptr};
  size_t cap[GROUP_RING] = {0, 0, 0};
};

struct pc_group {
  std::vector<pc_ctx*> ctx;
  std::vector<int> dev;                                    // HIP ordinal of context d
  std::vector<std::unique_ptr<DeviceWorker>> worker;
  // a second thread per device for the host -> device copy of a job's shard: a 512 MB copy from pageable memory keeps its
  // thread for ~10 ms, and on the worker it sat between an older job's division scan / open MSM and the device
  std::vector<std::unique_ptr<DeviceWorker>> copier;
  // ... and a third that only WAITS for queued MSMs (phase C of a job): on the worker such a wait (tens of ms) sat in front of the
  // next job's enqueues
  std::vector<std::unique_ptr<DeviceWorker>> reaper;
  std::vector<hipStream_t> copy_stream;                    // created by the copier thread at its first copy
  std::vector<DeviceBufs> bufs;
  std::mutex jobs_mu;
  std::deque<struct pc_group_job*> inflight;       // commit+open jobs not yet waited for, oldest first
  uint64_t seq = 0;
  std::string last_error;
};

struct pc_group_srs {
  pc_group* g = nullptr;
  pc_curve curve = PC_CURVE_BLS12_381;
  size_t n = 0, per = 0, pb = 0;          // points, points per chunk, bytes per packed point
  std::vector<pc_srs*> chunk;             // chunk d holds bases [lo(d) - halo(d), hi(d)),  halo(d) = d > 0
  size_t lo(size_t d) const { return std::min(n, d * per); }
  size_t hi(size_t d) const { return std::min(n, (d + 1) * per); }
  size_t halo(size_t d) const { return d > 0 && lo(d) < n ? 1 : 0; }
};

namespace {

size_t fq_bytes(pc_curve c) { return c == PC_CURVE_BLS12_381 ? 48 : 32; }

// run fn(d) for every device on that device's worker thread and wait for all of them; returns the first non-OK status
template <class Fn>
int fan_out(pc_group* g, Fn fn) {
  const size_t n_dev = g->ctx.size();
  std::vector<int> rc(n_dev, PC_OK);
  std::mutex mu; std::condition_variable cv; size_t left = n_dev;
  for (size_t d = 0; d < n_dev; d++)
    g->worker[d]->push([&, d]() {
      rc[d] = fn(d);
      std::lock_guard<std::mutex> lk(mu);
      if (--left == 0) cv.notify_one();
    });
  { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return left == 0; }); }
  for (int r : rc) if (r != PC_OK) return r;
  return PC_OK;
}

// Fr arithmetic for the division carries (a handful of elements per call)
template <class FrP>
struct FrHost {
  typedef pc::host64::F64<FrP> F;
  static F load(const void* p) { F f; memcpy(f.l, p, 32); return f; }
  // carry into shard d = value of the shards above it: c_d = sum_{s > d} B_s z^((s - d - 1) per'), composed top down
  static void carries(const std::vector<F>& B, const std::vector<size_t>& len, const F& z, std::vector<F>& carry) {
    const size_t N = B.size();
    carry.assign(N, F::zero());
    F acc = F::zero();
    for (size_t d = N; d-- > 0;) {
      carry[d] = acc;
      // acc = B_d + z^len_d * acc
      F zp = F::one(), base = z;
      for (size_t e = len[d]; e; e >>= 1) { if (e & 1) zp = zp.mul(base); base = base.mul(base); }
      acc = B[d].add(zp.mul(acc));
    }
  }
};

template <class FrP>
int open_carries(const std::vector<std::vector<uint64_t>>& evals, const std::vector<size_t>& len, const void* z_host,
                 std::vector<std::vector<uint64_t>>& carry, uint64_t* value_out) {
  typedef FrHost<FrP> H; typedef typename H::F F;
  std::vector<F> B; for (auto& e : evals) B.push_back(H::load(e.data()));
  std::vector<F> c;
  H::carries(B, len, H::load(z_host), c);
  carry.resize(B.size());
  for (size_t d = 0; d < B.size(); d++) { carry[d].assign(4, 0); memcpy(carry[d].data(), c[d].l, 32); }
  if (value_out) {   // p(z) = B_0 + z^len_0 * c_0
    F zp = F::one(), base = H::load(z_host);
    for (size_t e = len[0]; e; e >>= 1) { if (e & 1) zp = zp.mul(base); base = base.mul(base); }
    F v = B[0].add(zp.mul(c[0]));
    memcpy(value_out, v.l, 32);
  }
  return PC_OK;
}

}  // namespace

extern "C" {

int pc_hip_group_create(const int* device_ids, int n_devices, pc_group** out) {
  if (!out || !device_ids || n_devices <= 0) return PC_ERR_INVALID_ARG;
  *out = nullptr;
  pc_group* g = new (std::nothrow) pc_group();
  if (!g) return PC_ERR_OOM;
  for (int i = 0; i < n_devices; i++) {
    pc_ctx* c = nullptr;
    int rc = pc_hip_init(device_ids[i], &c);
    if (rc != PC_OK) { pc_hip_group_destroy(g); return rc; }
    g->ctx.push_back(c); g->dev.push_back(device_ids[i]);
  }
  g->bufs.resize(g->ctx.size());
  g->copy_stream.assign(g->ctx.size(), nullptr);
  for (size_t d = 0; d < g->ctx.size(); d++) {
    g->worker.emplace_back(new DeviceWorker()); g->worker.back()->start();
    g->copier.emplace_back(new DeviceWorker()); g->copier.back()->start();
    g->reaper.emplace_back(new DeviceWorker()); g->reaper.back()->start();
  }
  *out = g;
  return PC_OK;
}

void pc_hip_group_destroy(pc_group* g) {
  if (!g) return;
  while (!g->inflight.empty()) (void)pc_hip_group_job_wait(g, g->inflight.front());      // nothing may still be queued on the workers
  for (auto& w : g->copier) w->shutdown();
  for (auto& w : g->worker) w->shutdown();
  for (auto& w : g->reaper) w->shutdown();
  for (size_t d = 0; d < g->copy_stream.size(); d++)
    if (g->copy_stream[d] && hipSetDevice(g->dev[d]) == hipSuccess) (void)hipStreamDestroy(g->copy_stream[d]);
  for (size_t d = 0; d < g->ctx.size(); d++)
    if (d < g->bufs.size()) for (int k = 0; k < GROUP_RING; k++) { pc_hip_free(g->ctx[d], g->bufs[d].c[k]); pc_hip_free(g->ctx[d], g->bufs[d].q[k]); }
  for (pc_ctx* c : g->ctx) pc_hip_shutdown(c);
  delete g;
}

int pc_hip_group_size(const pc_group* g) { return g ? (int)g->ctx.size() : 0; }
pc_ctx* pc_hip_group_ctx(pc_group* g, int i) { return (g && i >= 0 && (size_t)i < g->ctx.size()) ? g->ctx[i] : nullptr; }

int pc_hip_group_srs_upload(pc_group* g, pc_curve curve, const void* bases_host, size_t n, size_t stride_bytes, int precompute,
                            pc_group_srs** out) {
  if (!g || !out || (!bases_host && n) || (int)curve < 0 || (int)curve > 2) return PC_ERR_INVALID_ARG;
  *out = nullptr;
  const size_t pb = 2 * fq_bytes(curve);
  if (stride_bytes == 0) stride_bytes = pb;
  if (stride_bytes < pb) return PC_ERR_INVALID_ARG;
  pc_group_srs* s = new (std::nothrow) pc_group_srs();
  if (!s) return PC_ERR_OOM;
  const size_t N = g->ctx.size();
  s->g = g; s->curve = curve; s->n = n; s->pb = pb; s->per = (n + N - 1) / N; if (!s->per) s->per = 1;
  s->chunk.assign(N, nullptr);
  int rc = fan_out(g, [&](size_t d) {
    const size_t lo = s->lo(d) - s->halo(d), hi = s->hi(d);
    int r = pc_hip_srs_upload(g->ctx[d], curve, (const char*)bases_host + lo * stride_bytes, hi - lo, stride_bytes, PC_MEM_HOST, &s->chunk[d]);
    if (r == PC_OK && precompute && hi > lo) r = pc_hip_srs_precompute(g->ctx[d], s->chunk[d], 0, 0);
    return r;
  });
  if (rc != PC_OK) { pc_hip_group_srs_free(s); return rc; }
  *out = s;
  return PC_OK;
}

void pc_hip_group_srs_free(pc_group_srs* s) {
  if (!s) return;
  for (pc_srs* c : s->chunk) pc_hip_srs_free(c);
  delete s;
}

size_t pc_hip_group_srs_len(const pc_group_srs* s) { return s ? s->n : 0; }

// out = sum_{i < n} scalars[i] * bases[base_offset + i]
int pc_hip_group_msm(pc_group* g, const pc_group_srs* s, size_t base_offset, const void* scalars_host, pc_scalar_form form, size_t n,
                     void* out_xy, int* out_is_infinity) {
  if (!g || !s || s->g != g || !out_xy || base_offset > s->n || (n && !scalars_host)) return PC_ERR_INVALID_ARG;
  if (n > s->n - base_offset) n = s->n - base_offset;                 // msm_bigint: min(bases.len(), scalars.len())
  const size_t N = g->ctx.size();
  std::vector<uint8_t> parts(N * s->pb, 0);
  int rc = fan_out(g, [&](size_t d) {
    const size_t a = std::max(base_offset, s->lo(d)), b = std::min(base_offset + n, s->hi(d));
    if (a >= b) return (int)PC_OK;                                     // partial = infinity (zeros)
    return pc_hip_msm(g->ctx[d], s->chunk[d], a - s->lo(d) + s->halo(d), (const char*)scalars_host + (a - base_offset) * 32, form,
                      PC_MEM_HOST, b - a, parts.data() + d * s->pb, nullptr);
  });
  if (rc != PC_OK) return rc;
  rc = pc_hip_points_sum(s->curve, parts.data(), N, out_xy);
  if (rc == PC_OK && out_is_infinity) { uint8_t acc = 0; for (size_t i = 0; i < s->pb; i++) acc |= ((const uint8_t*)out_xy)[i]; *out_is_infinity = acc == 0; }
  return rc;
}

// MarlinKZG10::commit's loop over polynomials (marlin_pc/mod.rs:192-237) against the sharded key: every device runs the k
// partial MSMs of its chunk as one pipelined batch; k * N partial points are folded on the host.
int pc_hip_group_msm_batch(pc_group* g, const pc_group_srs* s, const void* const* scalars_host, const size_t* n, size_t k,
                           pc_scalar_form form, void* out_xy, int* out_is_infinity) {
  if (!g || !s || s->g != g || !out_xy || (k && (!scalars_host || !n))) return PC_ERR_INVALID_ARG;
  const size_t N = g->ctx.size();
  std::vector<uint8_t> parts(N * k * s->pb, 0);
  int rc = fan_out(g, [&](size_t d) {
    std::vector<const void*> ptr(k); std::vector<size_t> len(k), off(k);
    for (size_t j = 0; j < k; j++) {
      const size_t nj = std::min(n[j], s->n), a = std::min(nj, s->lo(d)), b = std::min(nj, s->hi(d));
      ptr[j] = (const char*)scalars_host[j] + a * 32; len[j] = b - a; off[j] = s->halo(d);
    }
    return pc_hip_msm_batch(g->ctx[d], s->chunk[d], off.data(), ptr.data(), len.data(), k, form, PC_MEM_HOST, parts.data() + d * k * s->pb, nullptr);
  });
  if (rc != PC_OK) return rc;
  std::vector<uint8_t> col(N * s->pb);
  for (size_t j = 0; j < k && rc == PC_OK; j++) {
    for (size_t d = 0; d < N; d++) memcpy(&col[d * s->pb], &parts[(d * k + j) * s->pb], s->pb);
    uint8_t* o = (uint8_t*)out_xy + j * s->pb;
    rc = pc_hip_points_sum(s->curve, col.data(), N, o);
    if (out_is_infinity) { uint8_t acc = 0; for (size_t i = 0; i < s->pb; i++) acc |= o[i]; out_is_infinity[j] = acc == 0; }
  }
  return rc;
}

// KZG10::open (kzg10/mod.rs:287-310) of one polynomial of n coefficients against the sharded key, hiding off:
// witness polynomial p / (x - z) (:217-240) as one division scan per device with the carry of the shards above it
// (composed on the host from one evaluation per shard), then the MSM of the quotient chunk (:255-258).
// out_value_host (optional): p(z).
int pc_hip_group_kzg_open(pc_group* g, const pc_group_srs* s, const void* coeffs_host, size_t n, const void* z_host, void* out_proof_xy,
                          int* out_is_infinity, void* out_value_host) {
  if (!g || !s || s->g != g || !out_proof_xy || !z_host || (n && !coeffs_host) || n > s->n) return PC_ERR_INVALID_ARG;
  const size_t N = g->ctx.size();
  std::vector<void*> cdev(N, nullptr), qdev(N, nullptr);
  std::vector<size_t> len(N, 0);
  std::vector<std::vector<uint64_t>> evals(N, std::vector<uint64_t>(4, 0)), carry;
  for (size_t d = 0; d < N; d++) len[d] = std::min(n, s->hi(d)) - std::min(n, s->lo(d));
  auto cleanup = [&]() { for (size_t d = 0; d < N; d++) { pc_hip_free(g->ctx[d], cdev[d]); pc_hip_free(g->ctx[d], qdev[d]); } };
  // 1. shards to the devices, one evaluation each
  int rc = fan_out(g, [&](size_t d) {
    if (!len[d]) return (int)PC_OK;
    int r = pc_hip_malloc(g->ctx[d], len[d] * 32, &cdev[d]);
    if (r == PC_OK) r = pc_hip_malloc(g->ctx[d], len[d] * 32, &qdev[d]);
    if (r == PC_OK) r = pc_hip_memcpy_h2d(g->ctx[d], cdev[d], (const char*)coeffs_host + std::min(n, s->lo(d)) * 32, len[d] * 32);
    if (r == PC_OK) r = pc_hip_poly_eval(g->ctx[d], s->curve, cdev[d], PC_MEM_DEVICE, len[d], z_host, evals[d].data());
    return r;
  });
  // 2. carries (host: N field elements)
  if (rc == PC_OK) {
    switch (s->curve) {
      case PC_CURVE_BLS12_381: rc = open_carries<pc_bls12_381_fr>(evals, len, z_host, carry, (uint64_t*)out_value_host); break;
      case PC_CURVE_BN254: rc = open_carries<pc_bn254_fr>(evals, len, z_host, carry, (uint64_t*)out_value_host); break;
      default: rc = open_carries<pc_pallas_fr>(evals, len, z_host, carry, (uint64_t*)out_value_host); break;
    }
  }
  // 3. division with the carry, MSM of the quotient chunk: out[j] (coefficient j of this shard) pairs with power j - 1
  std::vector<uint8_t> parts(N * s->pb, 0);
  if (rc == PC_OK) rc = fan_out(g, [&](size_t d) {
    if (!len[d]) return (int)PC_OK;
    bool nz = false; for (uint64_t w : carry[d]) nz |= w != 0;
    int r = pc_hip_poly_div_scan(g->ctx[d], s->curve, cdev[d], PC_MEM_DEVICE, len[d], z_host, nz ? carry[d].data() : nullptr, qdev[d], PC_MEM_DEVICE);
    if (r != PC_OK) return r;
    if (d == 0) {                     // q[i - 1] = out[i], i >= 1: skips out[0] = p(z), powers from 0
      if (len[0] < 2) return (int)PC_OK;
      return pc_hip_msm(g->ctx[0], s->chunk[0], 0, (const char*)qdev[0] + 32, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, len[0] - 1, parts.data(), nullptr);
    }
    return pc_hip_msm(g->ctx[d], s->chunk[d], 0, qdev[d], PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, len[d], parts.data() + d * s->pb, nullptr);   // halo base first
  });
  cleanup();
  if (rc != PC_OK) return rc;
  rc = pc_hip_points_sum(s->curve, parts.data(), N, out_proof_xy);
  if (rc == PC_OK && out_is_infinity) { uint8_t acc = 0; for (size_t i = 0; i < s->pb; i++) acc |= ((const uint8_t*)out_proof_xy)[i]; *out_is_infinity = acc == 0; }
  return rc;
}

}  // extern "C"

// ---- commit + open of one polynomial as ONE asynchronous job ----------------------------------------------------------
// Copy (host coefficients only, on the device's copier thread): shard -> its ring slot; one copy serves commit and open.
// Phase A (per device, on its worker): commit MSM queued on an SRS pipeline, p_shard(z).  The device that finishes phase A last composes the carries on the host and queues phase B on
// every worker: division scan with the carry, open MSM queued.  Phase C reaps both MSMs.  Tasks of consecutive jobs interleave
// in the workers' FIFOs (A_k, A_k+1, B_k, C_k, ...), so job k+1's copy and sort overlap job k's accumulation -- the schedule
// bench.py's `value` runs through the per-pipeline API.
struct pc_group_job {
  pc_group* g = nullptr; const pc_group_srs* s = nullptr;
  size_t n = 0; int slot = 0; pc_mem where = PC_MEM_HOST;
  const void* coeffs = nullptr;                    // host array (PC_MEM_HOST) or array of N device pointers (PC_MEM_DEVICE)
  std::vector<const void*> dev_ptrs;
  uint64_t z[4] = {0, 0, 0, 0};
  void* out_commit = nullptr; void* out_proof = nullptr; void* out_value = nullptr;
  std::vector<size_t> len;
  std::vector<std::vector<uint64_t>> evals, carry;
  std::vector<uint8_t> part_c, part_w;            // N partial points each
  std::vector<pc_job*> jc, jw;
  std::vector<int> rc;
  std::mutex mu; std::condition_variable cv;
  size_t left_a = 0, left_c = 0; bool done = false;
  double ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // PC_HIP_GROUP_TRACE: submit, copy done, A done, B start, B done, C start, C done (device 0)
};

namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool group_trace() { static const bool on = getenv("PC_HIP_GROUP_TRACE") != nullptr; return on; }

int ensure_ring(pc_group* g, size_t d, int slot, size_t elems) {
  DeviceBufs& b = g->bufs[d];
  if (b.cap[slot] >= elems) return PC_OK;
  pc_hip_free(g->ctx[d], b.c[slot]); pc_hip_free(g->ctx[d], b.q[slot]); b.c[slot] = b.q[slot] = nullptr; b.cap[slot] = 0;
  int r = pc_hip_malloc(g->ctx[d], elems * 32, &b.c[slot]);
  if (r == PC_OK) r = pc_hip_malloc(g->ctx[d], elems * 32, &b.q[slot]);
  if (r == PC_OK) b.cap[slot] = elems;
  return r;
}

void job_phase_c(pc_group_job* j, size_t d) {
  pc_group* g = j->g;
  if (d == 0) j->ts[5] = now_ms();
  if (j->jc[d]) { int r = pc_hip_job_wait(g->ctx[d], j->jc[d]); if (j->rc[d] == PC_OK) j->rc[d] = r; }
  if (j->jw[d]) { int r = pc_hip_job_wait(g->ctx[d], j->jw[d]); if (j->rc[d] == PC_OK) j->rc[d] = r; }
  if (d == 0) j->ts[6] = now_ms();
  std::lock_guard<std::mutex> lk(j->mu);
  if (--j->left_c == 0) { j->done = true; j->cv.notify_all(); }
}

void job_phase_b(pc_group_job* j, size_t d) {
  pc_group* g = j->g; const pc_group_srs* s = j->s;
  if (d == 0) j->ts[3] = now_ms();
  if (j->rc[d] == PC_OK && j->len[d]) {
    const void* cdev = j->where == PC_MEM_DEVICE ? j->dev_ptrs[d] : g->bufs[d].c[j->slot];
    void* qdev = g->bufs[d].q[j->slot];
    bool nz = false; for (uint64_t w : j->carry[d]) nz |= w != 0;
    int r = pc_hip_poly_div_scan(g->ctx[d], s->curve, cdev, PC_MEM_DEVICE, j->len[d], j->z, nz ? j->carry[d].data() : nullptr, qdev, PC_MEM_DEVICE);
    if (r == PC_OK) {
      if (d == 0) {                   // q[i - 1] = out[i], i >= 1: skips out[0] = p(z), powers from 0
        if (j->len[0] >= 2) r = pc_hip_msm_async(g->ctx[0], s->chunk[0], 0, (const char*)qdev + 32, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, j->len[0] - 1,
                                                 j->part_w.data(), nullptr, &j->jw[0]);
      } else r = pc_hip_msm_async(g->ctx[d], s->chunk[d], 0, qdev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, j->len[d], j->part_w.data() + d * s->pb, nullptr, &j->jw[d]);   // halo base first
    }
    j->rc[d] = r;
  }
  if (d == 0) j->ts[4] = now_ms();
  g->reaper[d]->push([j, d]() { job_phase_c(j, d); });
}

// the shard of a host polynomial -> its ring slot, on the device's COPIER thread and a stream of its own (the slot's previous user
// is two jobs back and waited for: pc_hip_group_commit_open_async throttles); nothing of the context is locked while the copy runs
int job_copy_in(pc_group_job* j, size_t d) {
  pc_group* g = j->g; const pc_group_srs* s = j->s;
  int r = ensure_ring(g, d, j->slot, j->len[d]);
  if (r != PC_OK) return r;
  if (hipSetDevice(g->dev[d]) != hipSuccess) return PC_ERR_HIP;
  if (!g->copy_stream[d] && hipStreamCreateWithFlags(&g->copy_stream[d], hipStreamNonBlocking) != hipSuccess) return PC_ERR_HIP;
  const size_t lo = std::min(j->n, s->lo(d));
  if (hipMemcpyAsync(g->bufs[d].c[j->slot], (const char*)j->coeffs + lo * 32, j->len[d] * 32, hipMemcpyHostToDevice, g->copy_stream[d]) != hipSuccess) return PC_ERR_HIP;
  const int rc = hipStreamSynchronize(g->copy_stream[d]) == hipSuccess ? PC_OK : PC_ERR_HIP;
  if (d == 0) j->ts[1] = now_ms();
  return rc;
}

void job_phase_a(pc_group_job* j, size_t d, int r) {
  pc_group* g = j->g; const pc_group_srs* s = j->s;
  if (j->len[d]) {
    const void* cdev = nullptr;
    if (r == PC_OK) r = ensure_ring(g, d, j->slot, j->len[d]);
    if (r == PC_OK) cdev = j->where == PC_MEM_DEVICE ? j->dev_ptrs[d] : g->bufs[d].c[j->slot];
    // commit: coefficient lo + i pairs with power lo + i = chunk base halo(d) + i
    if (r == PC_OK) r = pc_hip_msm_async(g->ctx[d], s->chunk[d], s->halo(d), cdev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, j->len[d],
                                         j->part_c.data() + d * s->pb, nullptr, &j->jc[d]);
    // p_shard(z): feeds the division carries of the shards below (and p(z)); a single shard with no value requested needs none
    if (r == PC_OK && (g->ctx.size() > 1 || j->out_value)) r = pc_hip_poly_eval(g->ctx[d], s->curve, cdev, PC_MEM_DEVICE, j->len[d], j->z, j->evals[d].data());
  }
  j->rc[d] = r;
  if (d == 0) j->ts[2] = now_ms();
  bool last;
  { std::lock_guard<std::mutex> lk(j->mu); last = --j->left_a == 0; }
  if (!last) return;
  // every shard's p_s(z) is here: the carries (N field elements, host), then phase B everywhere
  int rc = PC_OK;
  for (int x : j->rc) if (x != PC_OK) rc = x;
  if (rc == PC_OK) {
    switch (s->curve) {
      case PC_CURVE_BLS12_381: rc = open_carries<pc_bls12_381_fr>(j->evals, j->len, j->z, j->carry, (uint64_t*)j->out_value); break;
      case PC_CURVE_BN254: rc = open_carries<pc_bn254_fr>(j->evals, j->len, j->z, j->carry, (uint64_t*)j->out_value); break;
      default: rc = open_carries<pc_pallas_fr>(j->evals, j->len, j->z, j->carry, (uint64_t*)j->out_value); break;
    }
  }
  if (rc != PC_OK) for (int& x : j->rc) if (x == PC_OK) x = rc;
  // phase B goes to the FRONT of every worker's queue: the open MSM of this job is queued on the device before a younger
  // job's copy and commit (phase C, which only reaps, goes to the back)
  for (size_t e = 0; e < g->ctx.size(); e++) g->worker[e]->push_front([j, e]() { job_phase_b(j, e); });
}

}  // namespace

extern "C" {

int pc_hip_group_commit_open_async(pc_group* g, const pc_group_srs* s, const void* coeffs, pc_mem where, size_t n, const void* z_host,
                                   void* out_commit_xy, void* out_proof_xy, void* out_value_host, pc_group_job** out_job) {
  if (!g || !s || s->g != g || !out_commit_xy || !out_proof_xy || !z_host || !out_job || (n && !coeffs) || n > s->n) return PC_ERR_INVALID_ARG;
  *out_job = nullptr;
  const size_t N = g->ctx.size();
  // at most GROUP_RING - 1 jobs in flight: a job's ring slot is reused two jobs later
  for (;;) {
    pc_group_job* oldest = nullptr;
    { std::lock_guard<std::mutex> lk(g->jobs_mu); if (g->inflight.size() >= (size_t)GROUP_RING - 1) oldest = g->inflight.front(); }
    if (!oldest) break;
    std::unique_lock<std::mutex> lk(oldest->mu);
    oldest->cv.wait(lk, [&]() { return oldest->done; });
    lk.unlock();
    std::lock_guard<std::mutex> lk2(g->jobs_mu);
    if (!g->inflight.empty() && g->inflight.front() == oldest) g->inflight.pop_front();      // finished; its owner still calls job_wait
  }
  pc_group_job* j = new (std::nothrow) pc_group_job();
  if (!j) return PC_ERR_OOM;
  j->g = g; j->s = s; j->n = n; j->where = where; j->coeffs = coeffs;
  if (where == PC_MEM_DEVICE) j->dev_ptrs.assign((const void* const*)coeffs, (const void* const*)coeffs + N);
  memcpy(j->z, z_host, 32);
  j->out_commit = out_commit_xy; j->out_proof = out_proof_xy; j->out_value = out_value_host;
  j->len.assign(N, 0);
  for (size_t d = 0; d < N; d++) j->len[d] = std::min(n, s->hi(d)) - std::min(n, s->lo(d));
  j->evals.assign(N, std::vector<uint64_t>(4, 0));
  j->part_c.assign(N * s->pb, 0); j->part_w.assign(N * s->pb, 0);
  j->jc.assign(N, nullptr); j->jw.assign(N, nullptr); j->rc.assign(N, PC_OK);
  j->left_a = N; j->left_c = N;
  j->ts[0] = now_ms();
  { std::lock_guard<std::mutex> lk(g->jobs_mu); j->slot = (int)(g->seq++ % GROUP_RING); g->inflight.push_back(j); }
  for (size_t d = 0; d < N; d++) {
    if (where == PC_MEM_HOST && j->len[d]) g->copier[d]->push([j, d]() { const int r = job_copy_in(j, d); j->g->worker[d]->push([j, d, r]() { job_phase_a(j, d, r); }); });
    else g->worker[d]->push([j, d]() { job_phase_a(j, d, PC_OK); });
  }
  *out_job = j;
  return PC_OK;
}

int pc_hip_group_job_wait(pc_group* g, pc_group_job* j) {
  if (!g || !j || j->g != g) return PC_ERR_INVALID_ARG;
  { std::unique_lock<std::mutex> lk(j->mu); j->cv.wait(lk, [&]() { return j->done; }); }
  { std::lock_guard<std::mutex> lk(g->jobs_mu); for (auto it = g->inflight.begin(); it != g->inflight.end(); ++it) if (*it == j) { g->inflight.erase(it); break; } }
  if (group_trace())
    fprintf(stderr, "[group job] submit %.2f  copy +%.2f  A +%.2f  B %.2f..%.2f  C %.2f..%.2f  wait-returns +%.2f\n", j->ts[0], j->ts[1] ? j->ts[1] - j->ts[0] : 0.0,
            j->ts[2] - j->ts[0], j->ts[3] - j->ts[0], j->ts[4] - j->ts[0], j->ts[5] - j->ts[0], j->ts[6] - j->ts[0], now_ms() - j->ts[0]);
  int rc = PC_OK;
  for (int x : j->rc) if (x != PC_OK) rc = x;
  if (rc == PC_OK) rc = pc_hip_points_sum(j->s->curve, j->part_c.data(), g->ctx.size(), j->out_commit);
  if (rc == PC_OK) rc = pc_hip_points_sum(j->s->curve, j->part_w.data(), g->ctx.size(), j->out_proof);
  delete j;
  return rc;
}

}  // extern "C"

extern "C" {

// LinearCodePCS::commit steps 1-3 (linear_codes/mod.rs:248-277) with the rows of the coefficient matrix split over the devices:
// every device encodes its slab (the rows are independent), the column digests are chained through the devices -- device d
// absorbs its slab into the per-column states device d - 1 left (pc_hip_column_hash_part; 48 bytes per column and hop through a
// host buffer, in column ranges so that the devices overlap) -- and the last device with rows builds the tree.  The encoded
// slabs stay resident when out_ext_slabs is given (one device pointer per device; freed by the caller with pc_hip_free).
int pc_hip_group_ligero_commit(pc_group* g, pc_curve field_of, const void* mat_host, size_t rows, size_t in_cols, unsigned log_n,
                               pc_hash col_hash, pc_hash tree_hash, int len_prefix, void** out_ext_slabs, void* leaves_out_host,
                               void* nodes_out_host) {
  if (!g || !rows || !in_cols || !mat_host || !nodes_out_host || log_n > 32 || in_cols > ((size_t)1 << log_n)) return PC_ERR_INVALID_ARG;
  const size_t N = g->ctx.size(), NC = (size_t)1 << log_n;
  size_t per = (rows + N - 1) / N; per += per & 1;                     // even slabs: two rows fill one block of the digests
  std::vector<size_t> lo(N), hi(N);
  size_t last_dev = 0;
  for (size_t d = 0; d < N; d++) { lo[d] = std::min(rows, d * per); hi[d] = std::min(rows, (d + 1) * per); if (hi[d] > lo[d]) last_dev = d; }
  std::vector<void*> ext(N, nullptr), state(N, nullptr);
  void* leaves = nullptr; void* nodes = nullptr;
  // 1. encode: all devices at once
  int rc = fan_out(g, [&](size_t d) {
    if (hi[d] == lo[d]) return (int)PC_OK;
    int r = pc_hip_malloc(g->ctx[d], (hi[d] - lo[d]) * NC * 32, &ext[d]);
    if (r == PC_OK) r = pc_hip_malloc(g->ctx[d], NC * 48, &state[d]);
    if (r == PC_OK) r = pc_hip_ntt_batch(g->ctx[d], field_of, (const char*)mat_host + lo[d] * in_cols * 32, PC_MEM_HOST, hi[d] - lo[d], in_cols, log_n,
                                         ext[d], PC_MEM_DEVICE);
    return r;
  });
  // 2. digests: range b of device d waits for range b of device d - 1 (a counter per device, advanced under one mutex)
  if (rc == PC_OK) rc = pc_hip_malloc(g->ctx[last_dev], NC * 32, &leaves);
  const size_t NB = std::min<size_t>(8, NC);
  std::vector<uint8_t> wire(NC * 48);                                  // the states between two devices (host)
  std::vector<size_t> done(N, 0);
  std::mutex mu; std::condition_variable cv; int chain_rc = PC_OK;
  if (rc == PC_OK) rc = fan_out(g, [&](size_t d) {
    if (hi[d] == lo[d]) return (int)PC_OK;
    const bool first = d == 0, last = d == last_dev;
    for (size_t b = 0; b < NB; b++) {
      const size_t c0 = NC * b / NB, c1 = NC * (b + 1) / NB;
      int r = PC_OK;
      if (!first) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return done[d - 1] > b || chain_rc != PC_OK; });
        if (chain_rc != PC_OK) return chain_rc;
        lk.unlock();
        r = pc_hip_memcpy_h2d(g->ctx[d], (char*)state[d] + c0 * 48, wire.data() + c0 * 48, (c1 - c0) * 48);
      }
      if (r == PC_OK) r = pc_hip_column_hash_part(g->ctx[d], field_of, col_hash, ext[d], hi[d] - lo[d], NC, rows, c0, c1 - c0, first, last, state[d], leaves);
      if (r == PC_OK && !last) r = pc_hip_memcpy_d2h(g->ctx[d], wire.data() + c0 * 48, (char*)state[d] + c0 * 48, (c1 - c0) * 48);
      { std::lock_guard<std::mutex> lk(mu); if (r != PC_OK) chain_rc = r; else done[d] = b + 1; }
      cv.notify_all();
      if (r != PC_OK) return r;
    }
    return (int)PC_OK;
  });
  // 3. tree on the last device
  unsigned h = 1; while (((size_t)1 << h) < NC) h++;
  if (rc == PC_OK) rc = pc_hip_malloc(g->ctx[last_dev], ((size_t)1 << h) * 32, &nodes);
  if (rc == PC_OK) rc = pc_hip_merkle_tree(g->ctx[last_dev], tree_hash, leaves, PC_MEM_DEVICE, NC, len_prefix, nodes, PC_MEM_DEVICE);
  if (rc == PC_OK) rc = pc_hip_memcpy_d2h(g->ctx[last_dev], nodes_out_host, nodes, (((size_t)1 << h) - 1) * 32);
  if (rc == PC_OK && leaves_out_host) rc = pc_hip_memcpy_d2h(g->ctx[last_dev], leaves_out_host, leaves, NC * 32);
  pc_hip_free(g->ctx[last_dev], leaves); pc_hip_free(g->ctx[last_dev], nodes);
  for (size_t d = 0; d < N; d++) {
    pc_hip_free(g->ctx[d], state[d]);
    if (rc == PC_OK && out_ext_slabs) out_ext_slabs[d] = ext[d]; else pc_hip_free(g->ctx[d], ext[d]);
  }
  return rc;
}

// LinearEncode::compute_matrices' rows (linear_codes/mod.rs:131-135) are independent: rows split over the devices,
// no exchange at all.
int pc_hip_group_ntt_batch(pc_group* g, pc_curve field_of, const void* in_host, size_t rows, size_t in_cols, unsigned log_n, void* out_host) {
  if (!g || (rows && (!in_host || !out_host))) return PC_ERR_INVALID_ARG;
  const size_t N = g->ctx.size(), per = (rows + N - 1) / N;
  const size_t out_cols = (size_t)1 << log_n;
  return fan_out(g, [&](size_t d) {
    const size_t a = std::min(rows, d * per), b = std::min(rows, (d + 1) * per);
    if (a >= b) return (int)PC_OK;
    return pc_hip_ntt_batch(g->ctx[d], field_of, (const char*)in_host + a * in_cols * 32, PC_MEM_HOST, b - a, in_cols, log_n,
                            (char*)out_host + a * out_cols * 32, PC_MEM_HOST);
  });
}

}  // extern "C"
