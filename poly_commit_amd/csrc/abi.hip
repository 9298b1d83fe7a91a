// C ABI of the gfx950 backend (include/pc_hip.h).  Thin glue: context/SRS lifetime, buffer
// staging, error translation.  The kernels live in the per-curve / per-field translation units
// (curve_*.hip, field_*.hip) and are reached through the ops tables of pc_internal.hpp.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "pc_internal.hpp"

using pc::MsmRunner;
using pc::NttRunner;

namespace {

// One independent MSM pipeline: own stream, own workspace.  Several lanes per SRS let the
// latency-bound tail of one MSM (segmented / bucket reduction, download, host Horner) overlap
// the bucket accumulation of the next.
struct MsmLane {
  pc::HipBackend be;
  MsmRunner* runner = nullptr;
  struct pc_job* inflight = nullptr;
  ~MsmLane() { delete runner; be.destroy(); }
};

}  // namespace

struct pc_ctx {
  std::map<std::pair<int, unsigned>, std::unique_ptr<NttRunner>> ntt_plans;
  float ntt_phases[2] = {0, 0};
  float ligero_phases[4] = {0, 0, 0, 0};
  int device = 0;
  pc::HipBackend be;
  std::recursive_mutex mu;   // recursive: the fused entry points call the single-step ones
  std::string last_error;
  pc::MsmConfig msm_cfg;
  float phases[8] = {0};
  float marks[8] = {0};          // the same marks as offsets from `epoch` (pc_hip_last_msm_marks_ms)
  hipEvent_t epoch = nullptr;    // recorded by pc_hip_set_timing(on)
  uint32_t shape[4] = {0};
  // pc_hip_ligero_commit in row slabs: the slab buffers (grow-only up to LIGERO_KEEP, pc_hip_ctx_trim frees them) and the queue of the way out
  void* lig_arena = nullptr; size_t lig_bytes = 0; hipStream_t lig_out_q[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<struct pc_srs*> keys;   // every key object of this context that is alive (pc_hip_ctx_bytes_resident, pc_hip_ctx_trim)
  // pc_hip_ipa_open_rounds: the powers of z, the per-base factors of the fixed key and the two scalar vectors of its rounds (grow-only, pc_hip_ctx_trim frees them)
  void* ipa_buf[3] = {nullptr, nullptr, nullptr}; size_t ipa_bytes[3] = {0, 0, 0};
};

// Independent pipelines per SRS (stream + workspace each), used round-robin; a pipeline that still
// holds a job is drained before it is reused.
static constexpr int PC_MSM_LANES = 3;
struct pc_srs {
  pc_ctx* ctx = nullptr;
  pc_curve curve = PC_CURVE_BLS12_381;
  size_t n = 0;
  uint32_t* bases = nullptr;     // packed x||y
  uint32_t* table = nullptr;     // precomputed window table (pc_hip_srs_precompute), or null
  uint32_t* fold_tbl = nullptr;  // fold table (pc_hip_srs_precompute_fold[_ex]) of the key points [fold_half, fold_half + fold_pts), or null
  size_t fold_half = 0;          // points of the key the table leaves out = size of the key it folds to: n / 2 (one level) or n / 4 (two)
  size_t fold_pts = 0;           // points per table row: n / 2 or 3 n / 4
  uint32_t fold_levels = 0, fold_w = 2;   // folds the table serves in one step; width of the NAF digits it holds the odd multiples for
  // pc_hip_ec_fold_from: the half-size working key of an opening keeps its buffers and pipelines across openings -- freeing it
  // hands it back to the committer key it was folded from (a fresh key cost ~4 ms of pipeline workspace allocation per opening)
  pc_srs* parent = nullptr;      // the key this one was folded from (while that key is alive)
  pc_srs* work_cache = nullptr;  // a returned working key, ready for reuse
  pc_srs* fixed_cache = nullptr; // pc_hip_ipa_open_rounds: the key object of the late rounds' FIXED key (n0 points, its window table, its pipelines), refilled by every opening
  pc_srs* work_out = nullptr;    // the working key currently handed out
  int aw = 0;                    // words per affine point
  pc::MsmConfig cfg;
  MsmLane* lanes[PC_MSM_LANES] = {nullptr, nullptr, nullptr};
  int next_lane = 0;
  // pc_hip_msm_many: window table of bases[base_offset .. base_offset + m) and the pipeline sized for B x m
  struct Many { size_t base_offset = 0, m = 0, B = 0; uint32_t* table = nullptr; MsmLane* lane = nullptr; } many;
  // pc_hip_msm_batch over the window table: G polynomials of m coefficients per pass, one bucket set each (two pipelines)
  struct BatchMany { size_t m = 0, G = 0; MsmLane* lanes[2] = {nullptr, nullptr}; uint32_t* stage[2] = {nullptr, nullptr}; } bm;      // stage: device copies of one pass's HOST polynomials
};
struct pc_job {
  pc_srs* srs = nullptr; int lane = 0;
  uint32_t* out_xy = nullptr; int* out_inf = nullptr;
  bool done = false; int status = 0;
};

static int fq_bytes(pc_curve c) { return c == PC_CURVE_BLS12_381 ? 48 : 32; }

template <class Fn>
static int guarded(pc_ctx* ctx, Fn fn) {
  try {
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PC_ERR_HIP; }
    return fn();
  } catch (const pc::HipError& e) {
    ctx->last_error = e.what();
    return e.code == hipErrorOutOfMemory ? PC_ERR_OOM : PC_ERR_HIP;
  } catch (const pc::MsmCapacityError& e) {
    ctx->last_error = e.what(); return PC_ERR_TOO_LARGE;
  } catch (const std::bad_alloc&) {
    ctx->last_error = "host allocation failed"; return PC_ERR_OOM;
  } catch (const std::exception& e) {
    ctx->last_error = e.what(); return PC_ERR_HIP;
  }
}

static MsmLane* srs_lane(pc_srs* srs, int i) {
  if (srs->lanes[i]) return srs->lanes[i];
  MsmLane* L = new MsmLane();
  try {
    // CU-partitioned pipelines are opt-in (PC_HIP_SPLIT_CUS=1): on this part a masked stream lost far
    // more throughput than the share of CUs it gave up (accumulate 3.65 ms on 256 CUs, 6.8 ms on 240).
    static const bool split = []() { const char* e = getenv("PC_HIP_SPLIT_CUS"); return e && e[0] == '1'; }();
    // default on; PC_HIP_TAIL_PRIO=0 puts a pipeline back on one queue (measured 8.5-9.0 -> 7.7 ms/step at 2^20)
    static const bool tsplit = []() { const char* e = getenv("PC_HIP_TAIL_PRIO"); return !(e && e[0] == '0'); }();
    // (pipelines of a large SRS keep one plain queue: the split only pays up to ~2^20 pairs per call, and
    // priority-created streams measured 5 % slower at 2^22 even with the split unused)
    L->be.tail_split = tsplit && srs->n <= ((size_t)3 << 19);
    if (i == 0 || !split) L->be.init(); else L->be.init(i - 1, PC_MSM_LANES - 1);
    L->runner = pc::curve_ops(srs->curve).make_runner(L->be, srs->n, srs->cfg, 0);
  } catch (...) { delete L; throw; }
  srs->lanes[i] = L;
  return L;
}

// Finish the job occupying a lane: wait for its stream, host tail, outputs, phase times.
static void complete_job(pc_ctx* ctx, pc_job* job) {
  pc_srs* srs = job->srs;
  MsmLane* L = srs->lanes[job->lane];
  // Whatever happens below (finish() may throw on a HIP error), the lane must not keep a pointer to this
  // job: it may live on the caller's stack (pc_hip_msm, pc_hip_msm_batch) or be deleted by pc_hip_job_wait.
  L->inflight = nullptr; job->done = true; job->status = PC_ERR_HIP;
  L->runner->finish(job->out_xy);
  if (job->out_inf) {
    uint32_t acc = 0;
    for (int i = 0; i < srs->aw; i++) acc |= job->out_xy[i];
    *job->out_inf = acc == 0;
  }
  for (int i = 0; i < 8; i++) ctx->phases[i] = 0;
  L->runner->shape(ctx->shape);
  if (L->be.timing) for (int i = 0; i + 1 < L->be.n_ev && i < 8; i++) (void)hipEventElapsedTime(&ctx->phases[i], L->be.ev[i], L->be.ev[i + 1]);
  for (int i = 0; i < 8; i++) ctx->marks[i] = -1.0f;
  if (L->be.timing && ctx->epoch) for (int i = 0; i < L->be.n_ev && i < 8; i++) (void)hipEventElapsedTime(&ctx->marks[i], ctx->epoch, L->be.ev[i]);
  job->status = PC_OK;
}

// Queue one MSM on the next lane (completing whatever that lane still holds).
static int enqueue_job(pc_ctx* ctx, pc_srs* srs, size_t base_offset, const void* scalars, pc_scalar_form form, pc_mem where,
                       size_t n, void* out_xy, int* out_is_infinity, pc_job* job, bool pipelined) {
  if (base_offset > srs->n) return PC_ERR_INVALID_ARG;
  size_t avail = srs->n - base_offset;     // msm_bigint semantics: min(bases.len(), scalars.len()) pairs
  if (n > avail) n = avail;
  if (n && !scalars) return PC_ERR_INVALID_ARG;
  (void)pipelined;
  int li = srs->next_lane; srs->next_lane = (srs->next_lane + 1) % PC_MSM_LANES;
  MsmLane* L = srs_lane(srs, li);
  if (L->inflight) complete_job(ctx, L->inflight);
  L->be.timing = ctx->be.timing;
  job->srs = srs; job->lane = li; job->out_xy = (uint32_t*)out_xy; job->out_inf = out_is_infinity; job->done = false;
  L->runner->enqueue(srs->bases, (uint32_t)base_offset, scalars, where, n, form == PC_SCALARS_MONTGOMERY);
  L->inflight = job;
  return PC_OK;
}

// fan-in of the upper levels of the division scan (tuning hook)
static uint32_t scan_fan() {
  static const uint32_t g = []() { const char* e = getenv("PC_HIP_SCAN_G"); int v = e ? atoi(e) : 0; return (uint32_t)(v >= 2 ? v : 16); }();
  return g;
}

static void drop_many(pc_srs* srs) {
  delete srs->many.lane; srs->many.lane = nullptr;
  if (srs->many.table) srs->ctx->be.free(srs->many.table);
  srs->many.table = nullptr; srs->many.m = srs->many.B = srs->many.base_offset = 0;
}

static void drop_batch_many(pc_srs* srs) {
  for (int i = 0; i < 2; i++) {
    if (srs->bm.stage[i] && srs->bm.lanes[i]) srs->bm.lanes[i]->be.free(srs->bm.stage[i]);
    srs->bm.stage[i] = nullptr;
    delete srs->bm.lanes[i]; srs->bm.lanes[i] = nullptr;
  }
  srs->bm.m = srs->bm.G = 0;
}

// Forget the window table of an SRS (and the pipelines sized for it).  No job may be in flight.
static void drop_table(pc_srs* srs) {
  drop_batch_many(srs);
  if (!srs->table) return;
  for (int i = 0; i < PC_MSM_LANES; i++) { delete srs->lanes[i]; srs->lanes[i] = nullptr; }
  srs->ctx->be.free(srs->table); srs->table = nullptr;
  srs->cfg.tbl = nullptr; srs->cfg.tbl_c = 0; srs->cfg.tbl_stride = 0; srs->cfg.tbl_min_n = 0; srs->cfg.tbl_glv = false;
}

extern "C" {

int pc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pc_hip_init(int device_id, pc_ctx** out) {
  if (!out) return PC_ERR_INVALID_ARG;
  *out = nullptr;
  int n = pc_hip_device_count();
  if (n <= 0) return PC_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return PC_ERR_INVALID_ARG;
  pc_ctx* ctx = new (std::nothrow) pc_ctx();
  if (!ctx) return PC_ERR_OOM;
  ctx->device = device_id;
  int rc = guarded(ctx, [&]() { ctx->be.init(); return (int)PC_OK; });
  if (rc != PC_OK) { delete ctx; return rc; }
  *out = ctx;
  return PC_OK;
}

static void srs_release_device(pc_srs* srs);
void pc_hip_shutdown(pc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  {
    // Keys that outlive their context (Drop order of an Arc<ResidentKey> against the context, a Python object collected late): their
    // device memory and pipelines go now, the host object stays behind as a tombstone (ctx = nullptr) that a later pc_hip_srs_free
    // only deletes -- it must never lock a mutex inside the context deleted below.  Cached working keys are held by nobody: deleted.
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    std::vector<pc_srs*> alive = ctx->keys, cached;
    for (pc_srs* s : alive) if (s->work_cache) { cached.push_back(s->work_cache); s->work_cache = nullptr; }
    for (pc_srs* s : alive) if (s->fixed_cache) { cached.push_back(s->fixed_cache); s->fixed_cache = nullptr; }
    for (pc_srs* s : alive) { s->parent = nullptr; s->work_out = nullptr; }
    for (pc_srs* s : alive) { srs_release_device(s); s->ctx = nullptr; }
    for (pc_srs* s : cached) delete s;
    ctx->keys.clear();
  }
  ctx->ntt_plans.clear();
  if (ctx->epoch) (void)hipEventDestroy(ctx->epoch);
  for (hipStream_t q : ctx->lig_out_q) if (q) (void)hipStreamDestroy(q);
  ctx->be.free(ctx->lig_arena);
  for (int i = 0; i < 3; i++) ctx->be.free(ctx->ipa_buf[i]);
  ctx->be.destroy();
  delete ctx;
}

const char* pc_hip_strerror(int status) {
  switch (status) {
    case PC_OK: return "ok";
    case PC_ERR_INVALID_ARG: return "invalid argument";
    case PC_ERR_OOM: return "out of memory";
    case PC_ERR_HIP: return "HIP runtime error";
    case PC_ERR_NO_DEVICE: return "no HIP device";
    case PC_ERR_TOO_LARGE: return "problem too large for this build";
    case PC_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
  }
}
const char* pc_hip_last_error(const pc_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int pc_hip_set_msm_tuning(pc_ctx* ctx, unsigned window_bits, unsigned chunk) {
  if (!ctx || window_bits == 1 || window_bits > 24) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->msm_cfg.c = window_bits; ctx->msm_cfg.T = chunk;
  return PC_OK;
}

int pc_hip_srs_upload(pc_ctx* ctx, pc_curve curve, const void* bases, size_t n, size_t stride_bytes, pc_mem where,
                      pc_srs** out) {
  if (!ctx || !out || (!bases && n) || (int)curve < 0 || (int)curve > 2) return PC_ERR_INVALID_ARG;
  const size_t pb = 2 * (size_t)fq_bytes(curve);
  if (stride_bytes == 0) stride_bytes = pb;
  if (stride_bytes < pb) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 31)) return PC_ERR_TOO_LARGE;
  if (where == PC_MEM_DEVICE && stride_bytes != pb) return PC_ERR_UNSUPPORTED;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  *out = nullptr;
  pc_srs* srs = new (std::nothrow) pc_srs();
  if (!srs) return PC_ERR_OOM;
  srs->ctx = ctx; srs->curve = curve; srs->n = n; srs->aw = (int)(pb / 4);
  ctx->keys.push_back(srs);
  int rc = guarded(ctx, [&]() {
    srs->bases = (uint32_t*)ctx->be.alloc((n ? n : 1) * pb);
    if (n) {
      if (where == PC_MEM_DEVICE) {
        ctx->be.copy_d2d(srs->bases, bases, n * pb);
      } else if (stride_bytes == pb) {
        ctx->be.copy_h2d(srs->bases, bases, n * pb);
      } else {
        // Rust Affine{x, y, infinity}: repack, mapping the flag to the (0,0) encoding
        std::vector<uint8_t> packed(n * pb);
        const uint8_t* src = (const uint8_t*)bases;
        for (size_t i = 0; i < n; i++) {
          const uint8_t* p = src + i * stride_bytes;
          if (p[pb]) memset(&packed[i * pb], 0, pb); else memcpy(&packed[i * pb], p, pb);
        }
        ctx->be.copy_h2d(srs->bases, packed.data(), n * pb);
        ctx->be.sync();
      }
      ctx->be.sync();
    }
    srs->cfg = ctx->msm_cfg;
    if (const char* e = getenv("PC_HIP_SEG_TAIL")) srs->cfg.seg_tail_lanes = (uint32_t)atoi(e);   // tuning experiments
    if (const char* e = getenv("PC_HIP_T2")) srs->cfg.T2 = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_T2B")) srs->cfg.T2b = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_K0")) srs->cfg.K0 = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_TBL_K0")) srs->cfg.tbl_K0 = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_TBL_LANES")) srs->cfg.tbl_target_lanes = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_TBL_CHUNK")) srs->cfg.tbl_chunk = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_TBL_MAX_LANES")) srs->cfg.tbl_max_lanes = (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_COOP_MAX_LOG2")) srs->cfg.coop_max_points = 1u << (uint32_t)atoi(e);
    if (const char* e = getenv("PC_HIP_COOP2_MAX_LOG2")) { int v = atoi(e); srs->cfg.coop2_max_points = v < 0 ? 0u : 1u << (uint32_t)(v > 30 ? 30 : v); }
    srs_lane(srs, 0);   // allocate the first pipeline now so that OOM surfaces at upload
    return (int)PC_OK;
  });
  if (rc != PC_OK) { pc_hip_srs_free(srs); return rc; }
  *out = srs;
  return PC_OK;
}

int pc_hip_srs_load_serialized(pc_ctx* ctx, pc_curve curve, const void* bytes, size_t n_bytes, int compressed, size_t max_points,
                               pc_srs** out, size_t* out_points, size_t* out_bytes_consumed) {
  if (!ctx || !out || !bytes || (int)curve < 0 || (int)curve > 2) return PC_ERR_INVALID_ARG;
  *out = nullptr;
  if (n_bytes < 8) return PC_ERR_INVALID_ARG;
  const size_t fb = (size_t)fq_bytes(curve);
  const size_t bits = curve == PC_CURVE_BLS12_381 ? 381 : curve == PC_CURVE_BN254 ? 254 : 255;
  const size_t yb = (bits + 2 + 7) / 8;
  const size_t pbytes = curve == PC_CURVE_BLS12_381 ? (compressed ? fb : 2 * fb) : (compressed ? yb : fb + yb);
  uint64_t len = 0; memcpy(&len, bytes, 8);                                    // Vec<T>: u64 little-endian length
  if (len > (n_bytes - 8) / pbytes) return PC_ERR_INVALID_ARG;                 // truncated input
  const size_t n = max_points && max_points < len ? max_points : (size_t)len;
  if (n >= (1ull << 31)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  void* raw = nullptr; void* pts = nullptr;
  uint32_t bad = 0;
  int rc = guarded(ctx, [&]() {
    raw = ctx->be.alloc(n * pbytes); pts = ctx->be.alloc((n ? n : 1) * 2 * fb);
    if (n) {
      ctx->be.copy_h2d(raw, (const char*)bytes + 8, n * pbytes);
      bad = pc::curve_ops(curve).srs_decode(ctx->be, (const uint8_t*)raw, n, compressed, (uint32_t*)pts);
    }
    return (int)PC_OK;
  });
  if (rc == PC_OK && bad) { ctx->last_error = std::to_string(bad) + " serialized point(s) are not on the curve"; rc = PC_ERR_INVALID_ARG; }
  if (rc == PC_OK) rc = pc_hip_srs_upload(ctx, curve, pts, n, 0, PC_MEM_DEVICE, out);
  (void)guarded(ctx, [&]() { ctx->be.free(raw); ctx->be.free(pts); return (int)PC_OK; });
  if (rc == PC_OK) { if (out_points) *out_points = n; if (out_bytes_consumed) *out_bytes_consumed = 8 + (size_t)len * pbytes; }
  return rc;
}

static size_t g1_point_bytes(pc_curve curve, int compressed) {
  const size_t fb = (size_t)fq_bytes(curve);
  const size_t bits = curve == PC_CURVE_BLS12_381 ? 381 : curve == PC_CURVE_BN254 ? 254 : 255;
  const size_t yb = (bits + 2 + 7) / 8;
  return curve == PC_CURVE_BLS12_381 ? (compressed ? fb : 2 * fb) : (compressed ? yb : fb + yb);
}

int pc_hip_srs_serialize(pc_ctx* ctx, const pc_srs* srs, size_t offset, size_t count, int compressed, void* out_bytes_host, size_t capacity,
                         size_t* out_written) {
  if (!ctx || !srs || srs->ctx != ctx || offset > srs->n || count > srs->n - offset || !out_written) return PC_ERR_INVALID_ARG;
  const size_t pbytes = g1_point_bytes(srs->curve, compressed), need = 8 + count * pbytes;
  *out_written = need;
  if (!out_bytes_host || capacity < need) return out_bytes_host ? PC_ERR_INVALID_ARG : PC_OK;      // NULL buffer: size query
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    const uint64_t len = count;
    memcpy(out_bytes_host, &len, 8);                                                                // Vec<T>: u64 little-endian length
    if (!count) return (int)PC_OK;
    void* dev = ctx->be.alloc(count * pbytes);
    try {
      pc::curve_ops(srs->curve).srs_encode(ctx->be, srs->bases + offset * (size_t)srs->aw, count, compressed, (uint8_t*)dev);
      ctx->be.copy_d2h((char*)out_bytes_host + 8, dev, count * pbytes);
    } catch (...) { ctx->be.free(dev); throw; }
    ctx->be.free(dev);
    return (int)PC_OK;
  });
}

int pc_hip_universal_params_layout(pc_curve curve, const void* bytes, size_t n_bytes, int compressed, size_t out[9]) {
  // kzg10::UniversalParams, CanonicalSerialize order (kzg10/data_structures.rs:57-77):
  //   powers_of_g: Vec<G1Affine> | powers_of_gamma_g: BTreeMap<usize, G1Affine> | h: G2Affine | beta_h: G2Affine |
  //   neg_powers_of_h: BTreeMap<usize, G2Affine>        (Vec / BTreeMap: u64 LE length first; map entries: u64 LE key, value)
  if (!bytes || !out || ((int)curve != PC_CURVE_BLS12_381 && (int)curve != PC_CURVE_BN254)) return PC_ERR_INVALID_ARG;   // pairing curves only
  const size_t g1 = g1_point_bytes(curve, compressed);
  // G2 over Fq2: BLS12-381 (zcash): 96 / 192 bytes; BN254 (generic SW over Fq2, flags in the spare bits of the last byte): 64 / 128
  const size_t g2 = curve == PC_CURVE_BLS12_381 ? (compressed ? 96 : 192) : (compressed ? 64 : 128);
  const uint8_t* p = (const uint8_t*)bytes;
  size_t at = 0;
  auto take_len = [&](uint64_t& v) { if (n_bytes - at < 8) return false; memcpy(&v, p + at, 8); at += 8; return true; };
  uint64_t n_g = 0, n_gg = 0, n_neg = 0;
  out[0] = at; if (!take_len(n_g) || n_g > (n_bytes - at) / g1) return PC_ERR_INVALID_ARG;
  out[1] = (size_t)n_g; at += (size_t)n_g * g1;
  out[2] = at; if (!take_len(n_gg) || n_gg > (n_bytes - at) / (8 + g1)) return PC_ERR_INVALID_ARG;
  out[3] = (size_t)n_gg; at += (size_t)n_gg * (8 + g1);
  if (n_bytes - at < 2 * g2) return PC_ERR_INVALID_ARG;
  out[4] = at; at += g2;                                   // h
  out[5] = at; at += g2;                                   // beta_h
  out[6] = at; if (!take_len(n_neg) || n_neg > (n_bytes - at) / (8 + g2)) return PC_ERR_INVALID_ARG;
  out[7] = (size_t)n_neg; at += (size_t)n_neg * (8 + g2);
  out[8] = at;                                             // total size of the structure
  return PC_OK;
}

// Every path mutates shared context state (ctx->keys, the backend's byte ledger, a parent's work cache) and is reached from arbitrary
// threads (Drop of the last Arc<ResidentKey>, device::release, the LRU eviction of the Rust shim) while other threads may be inside
// pc_hip_srs_upload / pc_hip_ctx_trim / any alloc: the context lock is held for the whole call (recursive: pc_hip_ctx_trim and the
// work-cache recursion below re-enter).  The mutex lives in the context; pc_hip_shutdown releases every key still alive and leaves it with ctx == nullptr, so a key freed after its context never touches that mutex.
static void srs_free_locked(pc_srs* srs);
void pc_hip_srs_free(pc_srs* srs) {
  if (!srs) return;
  if (srs->ctx) { std::lock_guard<std::recursive_mutex> lk(srs->ctx->mu); srs_free_locked(srs); }
  else srs_free_locked(srs);
}
static void srs_free_locked(pc_srs* srs) {
  if (srs->parent) {                                   // a working key goes back to its committer key (see pc_srs)
    pc_srs* par = srs->parent;
    for (int i = 0; i < PC_MSM_LANES; i++)             // nothing of it may still be queued
      if (srs->lanes[i] && srs->lanes[i]->inflight) { try { complete_job(srs->ctx, srs->lanes[i]->inflight); } catch (...) {} }
    if (par->work_out == srs) par->work_out = nullptr;
    if (!par->work_cache) { par->work_cache = srs; return; }
    srs->parent = nullptr;                             // the cache is taken: a real free
  }
  if (srs->work_cache) { srs->work_cache->parent = nullptr; srs_free_locked(srs->work_cache); srs->work_cache = nullptr; }
  if (srs->fixed_cache) { pc_srs* f = srs->fixed_cache; srs->fixed_cache = nullptr; f->parent = nullptr; srs_free_locked(f); }
  if (srs->work_out) { srs->work_out->parent = nullptr; srs->work_out = nullptr; }      // still held by the caller: it frees it
  srs_release_device(srs);
  delete srs;
}
// everything a key holds on the device and in its context's books; the host object is left empty (a key whose context was shut down
// under it has ctx == nullptr and nothing left to release)
static void srs_release_device(pc_srs* srs) {
  if (!srs->ctx) return;
  (void)hipSetDevice(srs->ctx->device);
  for (int i = 0; i < PC_MSM_LANES; i++) {
    if (srs->lanes[i] && srs->lanes[i]->inflight) {   // abandon: let the stream drain, mark the job failed
      (void)hipStreamSynchronize(srs->lanes[i]->be.stream);
      if (srs->lanes[i]->be.tail_stream) (void)hipStreamSynchronize(srs->lanes[i]->be.tail_stream);
      srs->lanes[i]->inflight->done = true; srs->lanes[i]->inflight->status = PC_ERR_INVALID_ARG;
    }
    delete srs->lanes[i]; srs->lanes[i] = nullptr;
  }
  { auto& ks = srs->ctx->keys; ks.erase(std::remove(ks.begin(), ks.end(), srs), ks.end()); }
  if (srs->bases) srs->ctx->be.free(srs->bases);
  if (srs->fold_tbl) srs->ctx->be.free(srs->fold_tbl);
  srs->fold_tbl = nullptr;
  drop_batch_many(srs);
  if (srs->table) srs->ctx->be.free(srs->table);
  drop_many(srs);
  srs->bases = srs->fold_tbl = srs->table = nullptr; srs->n = 0;
}
// windows of the key's table: the 255-bit scalar's, or those of its 130-bit GLV halves
static uint32_t table_windows(const pc_srs* srs, uint32_t c, bool glv) {
  return pc::msm_num_windows(glv ? pc::GLV_HALF_BITS : pc::curve_ops(srs->curve).scalar_bits, c);
}
int pc_hip_srs_precompute_ex(pc_ctx* ctx, pc_srs* srs, unsigned window_bits, size_t min_pairs, unsigned flags) {
  if (!ctx || !srs || srs->ctx != ctx || window_bits == 1 || window_bits > 23 || (flags & ~(unsigned)(PC_HIP_TABLE_GLV | PC_HIP_TABLE_GLV_IF_TIGHT | PC_HIP_TABLE_GLV_IF_LARGE))) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    for (int i = 0; i < PC_MSM_LANES; i++)
      if (srs->lanes[i] && srs->lanes[i]->inflight) complete_job(ctx, srs->lanes[i]->inflight);
    drop_table(srs);
    if (!srs->n) return (int)PC_OK;
    const uint32_t bits = pc::curve_ops(srs->curve).scalar_bits;
    // 96-byte points (BLS12-381) are padded to one 128-byte line each: a gather then touches one
    // DRAM line instead of 1.5 on average (the table no longer fits the 256 MB MALL)
    uint32_t pt_stride = srs->aw == 24 ? 32u : (uint32_t)srs->aw;
    if (const char* e = getenv("PC_HIP_TBL_PAD")) { if (!atoi(e)) pt_stride = (uint32_t)srs->aw; }      // =0: packed 96-byte rows
    bool glv = (flags & PC_HIP_TABLE_GLV) != 0;
    auto geometry = [&](bool g, uint32_t& c, uint32_t& Wt, size_t& bytes) {
      c = window_bits ? window_bits : pc::msm_choose_table_c(srs->n, bits, 5, g);
      if (const char* e = getenv("PC_HIP_TBL_C")) { if (!window_bits && atoi(e) >= 4 && atoi(e) <= 24) c = (uint32_t)atoi(e); }      // measurements only
      Wt = table_windows(srs, c, g);
      bytes = (size_t)Wt * srs->n * pt_stride * 4;
    };
    uint32_t c, Wt; size_t bytes;
    geometry(glv, c, Wt, bytes);
    if (!glv && (flags & PC_HIP_TABLE_GLV_IF_LARGE)) {
      // large keys: half the table (a 2^24-point BLS12-381 key: 12.9 instead of 25.8 GB) for one more bucket set to reduce; small keys
      // keep the full table (at 2^20 the second set's reduction costs 20 % of an MSM, the table only 1.6 GB)
      static const size_t large = []() { const char* e = getenv("PC_HIP_TABLE_GLV_LARGE_MB"); return (size_t)(e ? atol(e) : 4096) << 20; }();
      if (bytes > large) { glv = true; geometry(glv, c, Wt, bytes); }
    }
    if (!glv && (flags & PC_HIP_TABLE_GLV_IF_TIGHT)) {
      // the full table (bits / c + 1 copies of the key: 25.8 GB for 2^24 BLS12-381 points) only when it leaves half of the free
      // memory to everything else; otherwise the GLV form (half the windows: the same additions, one more bucket set to reduce)
      size_t free_b = 0, total_b = 0;
      PC_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
      if (bytes > free_b / 2) { glv = true; geometry(glv, c, Wt, bytes); }
    }
    if ((uint64_t)Wt * srs->n >= (1ull << 31)) return (int)PC_ERR_TOO_LARGE;     // entry = 31-bit table index + sign
    uint32_t* table = (uint32_t*)ctx->be.alloc(bytes);
    try {
      pc::curve_ops(srs->curve).window_table(ctx->be, srs->bases, (uint32_t)srs->n, c, Wt, table, pt_stride);
    } catch (...) { ctx->be.free(table); throw; }
    for (int i = 0; i < PC_MSM_LANES; i++) { delete srs->lanes[i]; srs->lanes[i] = nullptr; }
    srs->table = table;
    srs->cfg.tbl = table; srs->cfg.tbl_c = c; srs->cfg.tbl_stride = (uint32_t)srs->n; srs->cfg.tbl_pt_stride = pt_stride; srs->cfg.tbl_glv = glv;
    srs->cfg.tbl_min_n = min_pairs ? min_pairs : (srs->n + 3) / 4;
    try { srs_lane(srs, 0); }                     // workspace for the table geometry; on failure fall back
    catch (...) { drop_table(srs); srs_lane(srs, 0); throw; }
    return (int)PC_OK;
  });
}
int pc_hip_srs_precompute(pc_ctx* ctx, pc_srs* srs, unsigned window_bits, size_t min_pairs) {
  // PC_HIP_TABLE_GLV=1: every table in the GLV form; =0: never; =large: for keys whose full table exceeds 4 GiB; unset: the full table
  // unless device memory is tight (round 5: the GLV form costs 7 % of a pipelined 2^24 step -- split 0.6 ms, second bucket set 1.1 ms --
  // for 12.9 GB less: speed is the default, memory the option)
  static const unsigned flags = []() { const char* e = getenv("PC_HIP_TABLE_GLV"); return !e ? (unsigned)PC_HIP_TABLE_GLV_IF_TIGHT : !strcmp(e, "large") ? (unsigned)(PC_HIP_TABLE_GLV_IF_TIGHT | PC_HIP_TABLE_GLV_IF_LARGE) : atoi(e) ? (unsigned)PC_HIP_TABLE_GLV : 0u; }();
  return pc_hip_srs_precompute_ex(ctx, srs, window_bits, min_pairs, flags);
}
size_t pc_hip_srs_len(const pc_srs* srs) { return srs ? srs->n : 0; }
void* pc_hip_srs_device_ptr(const pc_srs* srs) { return srs ? srs->bases : nullptr; }

// Scalars that arrive in HOST memory are copied inside the call (the trait hands over &[F]: 512 MB at degree 2^24, ~9 ms of PCIe).
// Every step of one MSM needs all of its scalars, so nothing of that MSM can hide the copy -- but the MSM is a sum: from
// host_split_min() pairs on (2^21: measured gains 8 % / 17 % / 17 % / 14 % of a commit at 2^21 / 2^22 / 2^23 / 2^24, a loss at 2^20),
// the call runs in parts over index ranges (host_part_cuts below).
// a job on this call's stack must not outlive it inside a pipeline (an exception between two enqueues would leave the lane with a
// dangling pointer): completed on scope exit if it still is in flight
struct StackJob {
  pc_ctx* ctx; pc_job job;
  explicit StackJob(pc_ctx* c) : ctx(c) {}
  ~StackJob() { if (job.srs && !job.done) { try { complete_job(ctx, &job); } catch (...) {} } }
};
static size_t host_split_min() {
  static const size_t v = []() { const char* e = getenv("PC_HIP_HOST_SPLIT_LOG2"); int lg = e ? atoi(e) : 21; return lg <= 0 ? (size_t)-1 : (size_t)1 << (lg > 40 ? 40 : lg); }();
  return v;
}
// The two half-size jobs of a split call: pc_hip_last_msm_phases_ms then reports the SUM of both jobs' phase brackets (the halves run
// one after the other on the device where it matters: two accumulations never share the chip usefully), pc_hip_last_msm_marks_ms and
// pc_hip_last_msm_shape the second job's (one set of marks cannot describe two pipelines).  A job that an enqueue already completed
// (lane reuse) contributed its phases then; they are lost to the sum -- with PC_MSM_LANES = 3 pipelines that never happens for two jobs.
static void complete_two(pc_ctx* ctx, pc_job* a, pc_job* b) {
  float ph[8] = {0};
  if (!a->done) { complete_job(ctx, a); for (int i = 0; i < 8; i++) ph[i] = ctx->phases[i]; }
  if (!b->done) { complete_job(ctx, b); for (int i = 0; i < 8; i++) ctx->phases[i] += ph[i]; }
}
// Host scalars of at least host_split_min() pairs: ONE MSM in PC_HIP_HOST_PARTS parts (default 4; 0 = the two half-size MSMs on two
// pipelines of round 4) on one pipeline -- MsmPlan::begin_parts: the copy and sort of part k + 1 run beside the accumulation of part k,
// all parts share one bucket reduction and one host tail.
// PC_HIP_HOST_PARTS: a part count (equal parts) or a comma list of relative weights (default "1,2,5,8": a short first part, so that the
// first copy and sort -- the only ones nothing hides -- are short; measured at 2^24 BLS12-381: commit of host coefficients 40.3 ms against
// 38.5 resident and 46.8 as two half-size MSMs, open 41.7 against 39.5 / 46.8; four equal parts 44.0 / 45.3, "1,3,4,8" 40.9 / 42.4).
static const std::vector<double>& host_part_cuts() {      // cumulative fractions: cuts[0] = 0 < ... < cuts[K] = 1; empty = no parts
  static const std::vector<double> cuts = []() {
    std::vector<double> w;
    const char* e = getenv("PC_HIP_HOST_PARTS");
    std::string spec = e ? e : "1,2,5,8";
    if (spec.find(',') == std::string::npos) { int k = atoi(spec.c_str()); if (k > 8) k = 8; for (int i = 0; i < k; i++) w.push_back(1.0); }
    else { size_t at = 0; while (at <= spec.size() && w.size() < 8) { size_t c = spec.find(',', at); if (c == std::string::npos) c = spec.size(); double v = atof(spec.substr(at, c - at).c_str()); if (v > 0) w.push_back(v); at = c + 1; } }
    std::vector<double> out;
    if (w.size() < 2) return out;
    double tot = 0; for (double v : w) tot += v;
    double acc = 0; out.push_back(0.0);
    for (double v : w) { acc += v; out.push_back(acc / tot); }
    out.back() = 1.0;
    return out;
  }();
  return cuts;
}
static size_t host_parts() { return host_part_cuts().empty() ? 0 : host_part_cuts().size() - 1; }
static size_t part_cut(size_t n, size_t k) { const auto& c = host_part_cuts(); return k + 1 >= c.size() ? n : (size_t)((double)n * c[k]); }
// claim a pipeline of the key for a job in parts (completing what it still holds).  Always the first one: a blocking call gains nothing from
// rotating, and only the pipeline that runs parts grows the second sort output and bucket array (1.2 GB at 2^24 BLS12-381 points).
static MsmLane* claim_lane(pc_ctx* ctx, pc_srs* srs, void* out_xy, int* out_is_infinity, pc_job* job) {
  const int li = 0;
  MsmLane* L = srs_lane(srs, li);
  if (L->inflight) complete_job(ctx, L->inflight);
  L->be.timing = ctx->be.timing;
  job->srs = srs; job->lane = li; job->out_xy = (uint32_t*)out_xy; job->out_inf = out_is_infinity; job->done = false;
  return L;
}
// sum of two affine results into out_xy / out_is_infinity
static void fold_two(pc_srs* srs, const uint32_t* a, const uint32_t* b, void* out_xy, int* out_is_infinity) {
  std::vector<uint32_t> two(2 * (size_t)srs->aw);
  memcpy(two.data(), a, (size_t)srs->aw * 4); memcpy(two.data() + srs->aw, b, (size_t)srs->aw * 4);
  pc::curve_ops(srs->curve).points_sum(two.data(), 2, (uint32_t*)out_xy);
  if (out_is_infinity) { uint32_t acc = 0; for (int i = 0; i < srs->aw; i++) acc |= ((const uint32_t*)out_xy)[i]; *out_is_infinity = acc == 0; }
}

int pc_hip_msm(pc_ctx* ctx, const pc_srs* srs_c, size_t base_offset, const void* scalars, pc_scalar_form form,
               pc_mem where, size_t n, void* out_xy, int* out_is_infinity) {
  pc_srs* srs = const_cast<pc_srs*>(srs_c);
  if (!ctx || !srs || !out_xy || srs->ctx != ctx) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    const size_t avail = base_offset <= srs->n ? srs->n - base_offset : 0;
    const size_t ne = n < avail ? n : avail;
    if (where == PC_MEM_HOST && scalars && ne >= host_split_min() && base_offset <= srs->n && host_parts() >= 2) {
      const size_t K = host_parts();
      StackJob j(ctx);
      MsmLane* L = claim_lane(ctx, srs, out_xy, out_is_infinity, &j.job);
      L->runner->begin_parts(ne);
      L->inflight = &j.job;                      // from here on the pipeline holds work of this job (StackJob completes it on any exit)
      std::vector<std::pair<size_t, size_t>> parts;        // (first, count), empty parts dropped
      for (size_t k = 0; k < K; k++) { const size_t first = part_cut(ne, k), cnt = part_cut(ne, k + 1) - first; if (cnt) parts.push_back({first, cnt}); }
      for (size_t k = 0; k < parts.size(); k++)
        L->runner->add_part(srs->bases, (uint32_t)base_offset, parts[k].first, (const uint8_t*)scalars + parts[k].first * 32, PC_MEM_HOST, parts[k].second,
                            form == PC_SCALARS_MONTGOMERY, k + 1 == parts.size());
      complete_job(ctx, &j.job);
      return (int)PC_OK;
    }
    if (where == PC_MEM_HOST && scalars && ne >= host_split_min() && base_offset <= srs->n) {
      const size_t h = ne / 2;
      std::vector<uint32_t> r1(srs->aw), r2(srs->aw);
      StackJob j1(ctx), j2(ctx);
      int rc = enqueue_job(ctx, srs, base_offset, scalars, form, where, h, r1.data(), nullptr, &j1.job, true);
      if (rc != PC_OK) return rc;
      rc = enqueue_job(ctx, srs, base_offset + h, (const uint8_t*)scalars + h * 32, form, where, ne - h, r2.data(), nullptr, &j2.job, true);
      if (rc != PC_OK) return rc;
      complete_two(ctx, &j1.job, &j2.job);
      fold_two(srs, r1.data(), r2.data(), out_xy, out_is_infinity);
      return (int)PC_OK;
    }
    pc_job job;
    int rc = enqueue_job(ctx, srs, base_offset, scalars, form, where, n, out_xy, out_is_infinity, &job, false);
    if (rc != PC_OK) return rc;
    complete_job(ctx, &job);
    return (int)PC_OK;
  });
}

int pc_hip_msm_async(pc_ctx* ctx, const pc_srs* srs_c, size_t base_offset, const void* scalars, pc_scalar_form form,
                     pc_mem where, size_t n, void* out_xy, int* out_is_infinity, pc_job** out_job) {
  pc_srs* srs = const_cast<pc_srs*>(srs_c);
  if (!ctx || !srs || !out_xy || !out_job || srs->ctx != ctx) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  *out_job = nullptr;
  pc_job* job = new (std::nothrow) pc_job();
  if (!job) return PC_ERR_OOM;
  int rc = guarded(ctx, [&]() { return enqueue_job(ctx, srs, base_offset, scalars, form, where, n, out_xy, out_is_infinity, job, true); });
  if (rc != PC_OK) { delete job; return rc; }
  *out_job = job;
  return PC_OK;
}

int pc_hip_job_wait(pc_ctx* ctx, pc_job* job) {
  if (!ctx || !job) return PC_ERR_INVALID_ARG;
  // The wait for the device happens WITHOUT the context's lock: other threads (the per-device workers of pc_hip_group_*)
  // keep queueing work on this context meanwhile -- with the lock held for the whole wait a reaper serialised them behind
  // every MSM it waited for.  The bookkeeping behind it (host tail, phase times) is under the lock as before; a job that
  // another call completed in between (enqueue_job reusing its lane) is simply found done.
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!job->done && job->srs) { MsmLane* L = job->srs->lanes[job->lane]; if (L && L->inflight == job) ev = L->be.done; }
  }
  if (ev && hipSetDevice(ctx->device) == hipSuccess) (void)hipEventSynchronize(ev);      // an error surfaces in complete_job below
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  int rc = job->status;
  if (!job->done) rc = guarded(ctx, [&]() { complete_job(ctx, job); return job->status; });
  delete job;
  return rc;
}

int pc_hip_msm_batch(pc_ctx* ctx, const pc_srs* srs_c, const size_t* base_offsets, const void* const* scalars,
                     const size_t* n, size_t n_polys, pc_scalar_form form, pc_mem where, void* out_xy,
                     int* out_is_infinity) {
  pc_srs* srs = const_cast<pc_srs*>(srs_c);
  if (!ctx || !srs || !out_xy || srs->ctx != ctx || (n_polys && (!scalars || !n))) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    // Equal-length polynomials against the window table of the key (MarlinKZG10::commit of a batch: config 3): G of them per
    // pass through ONE sort / accumulate / reduce pipeline with a bucket set each (the many-MSM machinery of
    // pc_hip_msm_many, over the key's own table).  The latency-bound reductions and the host tail are then paid once per G
    // polynomials and are wide enough to be throughput-bound; two such pipelines alternate.
    {
      // (HOST polynomials -- what MarlinKZG10::commit hands over -- take the same passes: the G polynomials of a pass are copied to a
      // staging buffer on the pipeline that will run the pass, i.e. beside the other pipeline's pass: 64 x 2^20 host polynomials cost
      // one exposed copy of 8, not 2 GiB of PCIe in front of the batch)
      bool same = n_polys >= 2 && srs->table && n[0] >= ((size_t)1 << 14) && n[0] >= srs->cfg.tbl_min_n;
      const size_t b0 = base_offsets ? base_offsets[0] : 0;
      for (size_t k = 0; same && k < n_polys; k++) same = n[k] == n[0] && (base_offsets ? base_offsets[k] : 0) == b0 && scalars[k];
      if (same && b0 <= srs->n && n[0] <= srs->n - b0) {
        static const size_t Gmax = []() { const char* e = getenv("PC_HIP_BATCH_G"); int v = e ? atoi(e) : 8; return (size_t)(v < 0 ? 0 : v); }();
        const size_t m = n[0];
        size_t G = std::min(Gmax, n_polys);
        const uint32_t sets = srs->cfg.tbl_glv ? 2u : 1u;
        const uint32_t Wd = sets * table_windows(srs, srs->cfg.tbl_c, srs->cfg.tbl_glv);      // digits per scalar
        while (G >= 2 && ((uint64_t)G * m * Wd >= (1ull << 32) || ((uint64_t)G * sets << (srs->cfg.tbl_c - 1)) >= (1ull << 31))) G /= 2;
        // HOST polynomials are staged on the device, G of them per pipeline: that copy has a budget (BATCH_STAGE_MAX per pipeline; 8 x
        // 2^24 coefficients would be 2 x 4 GiB beside the passes' own workspace).  G shrinks to fit; below two polynomials per pass
        // the call takes the per-polynomial pipeline further down, which stages one polynomial at a time.
        static const size_t BATCH_STAGE_MAX = []() { const char* e = getenv("PC_HIP_BATCH_STAGE_MAX_MB"); long v = e ? atol(e) : 1024; return (size_t)(v < 0 ? 0 : v) << 20; }();
        if (where == PC_MEM_HOST) while (G >= 2 && (uint64_t)G * m * 32 > BATCH_STAGE_MAX) G /= 2;
        if (G >= 2) {
          pc_srs::BatchMany& B = srs->bm;
          if (B.m != m || B.G != G) {
            drop_batch_many(srs);
            for (int i = 0; i < 2; i++) {
              MsmLane* L = new MsmLane();
              try { L->be.init(); L->runner = pc::curve_ops(srs->curve).make_runner(L->be, G * m, srs->cfg, (uint32_t)G); }
              catch (...) { delete L; drop_batch_many(srs); throw; }
              B.lanes[i] = L;
            }
            B.m = m; B.G = G;
          }
          // both staging buffers before any pass is queued: when the device cannot give them, nothing is in flight yet and the
          // call goes on through the per-polynomial pipeline instead of failing (host inputs took that road before this path existed)
          bool staged_ok = true;
          if (where == PC_MEM_HOST)
            for (int i = 0; i < 2 && staged_ok; i++)
              if (!B.stage[i]) {
                try { B.stage[i] = (uint32_t*)B.lanes[i]->be.alloc(G * m * 32); }
                catch (const std::exception&) {
                  staged_ok = false;
                  (void)hipGetLastError();
                  for (int j = 0; j < 2; j++) if (B.stage[j]) { B.lanes[j]->be.free(B.stage[j]); B.stage[j] = nullptr; }
                }
              }
          if (staged_ok) {
          for (int i = 0; i < PC_MSM_LANES; i++)       // nothing of the single-MSM pipelines may be in flight on this key's outputs
            if (srs->lanes[i] && srs->lanes[i]->inflight) complete_job(ctx, srs->lanes[i]->inflight);
          const size_t pb = (size_t)srs->aw * 4;
          std::vector<uint32_t> tmp[2]; tmp[0].resize(G * srs->aw); tmp[1].resize(G * srs->aw);
          size_t pending_first[2] = {0, 0}, pending_cnt[2] = {0, 0};
          // phase brackets of the passes (pc_hip_last_msm_phases_ms after a batch): [0..5] summed over the passes, [6] the union of
          // the passes' accumulate intervals (consecutive passes overlap on the two pipelines), [7] the number of passes
          float ph_sum[8] = {0}; std::vector<std::pair<float, float>> acc_iv;
          auto drain = [&](int li) {
            if (!pending_cnt[li]) return;
            B.lanes[li]->runner->finish(tmp[li].data());
            pc::HipBackend& lbe = B.lanes[li]->be;
            if (lbe.timing && lbe.n_ev >= 5) {
              for (int i = 0; i + 1 < lbe.n_ev && i < 6; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, lbe.ev[i], lbe.ev[i + 1]); ph_sum[i] += ms; }
              if (ctx->epoch) { float a = 0, b = 0; (void)hipEventElapsedTime(&a, ctx->epoch, lbe.ev[3]); (void)hipEventElapsedTime(&b, ctx->epoch, lbe.ev[4]); acc_iv.push_back({a, b}); }
              ph_sum[7] += 1.0f;
            }
            for (size_t k = 0; k < pending_cnt[li]; k++) {
              uint8_t* o = (uint8_t*)out_xy + (pending_first[li] + k) * pb;
              memcpy(o, tmp[li].data() + k * srs->aw, pb);
              if (out_is_infinity) { uint32_t acc = 0; for (int w = 0; w < srs->aw; w++) acc |= tmp[li][k * srs->aw + w]; out_is_infinity[pending_first[li] + k] = acc == 0; }
            }
            pending_cnt[li] = 0;
          };
          int li = 0;
          for (size_t first = 0; first < n_polys; first += G, li ^= 1) {
            drain(li);
            const size_t cnt = std::min(G, n_polys - first);
            std::vector<uint64_t> ptrs(cnt);
            if (where == PC_MEM_HOST) {
              pc::HipBackend& lbe = B.lanes[li]->be;
              for (size_t k = 0; k < cnt; k++) {
                uint32_t* dst = B.stage[li] + k * m * 8;
                lbe.copy_h2d(dst, scalars[first + k], m * 32);
                ptrs[k] = (uint64_t)(uintptr_t)dst;
              }
            } else
              for (size_t k = 0; k < cnt; k++) ptrs[k] = (uint64_t)(uintptr_t)scalars[first + k];
            B.lanes[li]->be.timing = ctx->be.timing;
            B.lanes[li]->runner->enqueue_vectors(srs->bases, (uint32_t)b0, ptrs.data(), cnt, m, form == PC_SCALARS_MONTGOMERY);
            pending_first[li] = first; pending_cnt[li] = cnt;
          }
          drain(li); drain(li ^ 1);
          if (ctx->be.timing) {
            std::sort(acc_iv.begin(), acc_iv.end());
            float tot = 0, end = -1e30f;
            for (auto& iv : acc_iv) { if (iv.second <= end) continue; tot += iv.second - std::max(iv.first, end); end = iv.second; }
            ph_sum[6] = tot;
            for (int i = 0; i < 8; i++) ctx->phases[i] = ph_sum[i];
            B.lanes[0]->runner->shape(ctx->shape);
          }
          // staging above the keep threshold is transient, as the single-call buffers are (CallBuf / STAGE_KEEP)
          static constexpr size_t BATCH_STAGE_KEEP = (size_t)256 << 20;
          if (where == PC_MEM_HOST && G * m * 32 > BATCH_STAGE_KEEP)
            for (int i = 0; i < 2; i++) if (B.stage[i]) { B.lanes[i]->be.free(B.stage[i]); B.stage[i] = nullptr; }
          return (int)PC_OK;
          }      // staged_ok
        }
      }
    }
    // software pipeline over the lanes: polynomial k+1 accumulates while k's tail drains
    std::vector<pc_job> jobs(n_polys);
    for (size_t k = 0; k < n_polys; k++) {
      int rc = enqueue_job(ctx, srs, base_offsets ? base_offsets[k] : 0, scalars[k], form, where, n[k],
                           (uint8_t*)out_xy + k * (size_t)srs->aw * 4, out_is_infinity ? out_is_infinity + k : nullptr, &jobs[k], n_polys > 1);
      if (rc != PC_OK) { for (size_t j = 0; j < k; j++) if (!jobs[j].done) complete_job(ctx, &jobs[j]); return rc; }
    }
    for (size_t k = 0; k < n_polys; k++) if (!jobs[k].done) complete_job(ctx, &jobs[k]);
    return (int)PC_OK;
  });
}

int pc_hip_malloc(pc_ctx* ctx, size_t bytes, void** out_dev) {
  if (!ctx || !out_dev) return PC_ERR_INVALID_ARG;
  *out_dev = nullptr;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() { *out_dev = ctx->be.alloc(bytes); return (int)PC_OK; });
}
int pc_hip_free(pc_ctx* ctx, void* dev) {
  if (!ctx) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() { ctx->be.free(dev); return (int)PC_OK; });
}
int pc_hip_memcpy_h2d(pc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!ctx || (bytes && (!dst_dev || !src_host))) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() { if (bytes) { ctx->be.copy_h2d(dst_dev, src_host, bytes); ctx->be.sync(); } return (int)PC_OK; });
}
int pc_hip_memcpy_d2h(pc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (!ctx || (bytes && (!dst_host || !src_dev))) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() { if (bytes) ctx->be.copy_d2h(dst_host, src_dev, bytes); return (int)PC_OK; });
}

int pc_hip_msm_many(pc_ctx* ctx, pc_srs* srs, size_t base_offset, const void* scalars, pc_scalar_form form, pc_mem where,
                    size_t m, size_t n_msms, void* out_xy, int* out_is_infinity) {
  if (!ctx || !srs || srs->ctx != ctx || !out_xy || (m && n_msms && !scalars)) return PC_ERR_INVALID_ARG;
  if (base_offset > srs->n || m > srs->n - base_offset) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    const size_t pb = (size_t)srs->aw * 4;
    if (!n_msms) return (int)PC_OK;
    if (!m) { memset(out_xy, 0, n_msms * pb); if (out_is_infinity) for (size_t k = 0; k < n_msms; k++) out_is_infinity[k] = 1; return (int)PC_OK; }
    const uint32_t bits = srs->curve == PC_CURVE_BN254 ? 254u : 255u;
    const uint32_t c = pc::msm_choose_table_c(m, bits, 0), Wd = pc::msm_num_windows(bits, c);
    if ((uint64_t)n_msms * m * Wd >= (1ull << 31) || ((uint64_t)n_msms << (c - 1)) >= (1ull << 31)) return (int)PC_ERR_TOO_LARGE;
    pc_srs::Many& M = srs->many;
    if (!M.lane || M.base_offset != base_offset || M.m != m || M.B != n_msms) {
      drop_many(srs);
      uint32_t* table = (uint32_t*)ctx->be.alloc((size_t)Wd * m * pb);
      MsmLane* L = nullptr;
      try {
        const uint32_t* b0 = srs->bases + base_offset * srs->aw;
        pc::curve_ops(srs->curve).window_table(ctx->be, b0, (uint32_t)m, c, Wd, table, (uint32_t)srs->aw);
        pc::MsmConfig cfg = srs->cfg;
        cfg.c = 0; cfg.T = 0; cfg.tbl = table; cfg.tbl_c = c; cfg.tbl_stride = (uint32_t)m; cfg.tbl_pt_stride = (uint32_t)srs->aw; cfg.tbl_min_n = 0;
        cfg.tbl_glv = false;      // this pass's own table is the full one: c, Wd and the capacity checks above are the plain form's, whatever the key's table is
        L = new MsmLane();
        L->be.init();
        L->runner = pc::curve_ops(srs->curve).make_runner(L->be, n_msms * m, cfg, (uint32_t)n_msms);
      } catch (...) { delete L; ctx->be.free(table); throw; }
      M.table = table; M.lane = L; M.base_offset = base_offset; M.m = m; M.B = n_msms;
    }
    MsmLane* L = M.lane;
    L->be.timing = ctx->be.timing;
    L->runner->enqueue(srs->bases, 0, scalars, where, n_msms * m, form == PC_SCALARS_MONTGOMERY);
    L->runner->finish((uint32_t*)out_xy);
    if (out_is_infinity) {
      const uint32_t* o = (const uint32_t*)out_xy;
      for (size_t k = 0; k < n_msms; k++) { uint32_t acc = 0; for (int i = 0; i < srs->aw; i++) acc |= o[k * srs->aw + i]; out_is_infinity[k] = acc == 0; }
    }
    for (int i = 0; i < 8; i++) ctx->phases[i] = 0;
  L->runner->shape(ctx->shape);
    if (L->be.timing) for (int i = 0; i + 1 < L->be.n_ev && i < 8; i++) (void)hipEventElapsedTime(&ctx->phases[i], L->be.ev[i], L->be.ev[i + 1]);
    return (int)PC_OK;
  });
}

static size_t lane_bytes(const MsmLane* L) { return L ? L->be.bytes_live : 0; }
static void srs_bytes(const pc_srs* s, size_t out[4]) {
  const size_t pb = (size_t)s->aw * 4;
  out[0] = (s->n ? s->n : 1) * pb;
  out[1] = 0;
  if (s->table) out[1] = (size_t)table_windows(s, s->cfg.tbl_c, s->cfg.tbl_glv) * s->n * s->cfg.tbl_pt_stride * 4;
  if (s->many.table) { const uint32_t bits = pc::curve_ops(s->curve).scalar_bits; out[1] += (size_t)pc::msm_num_windows(bits, pc::msm_choose_table_c(s->many.m, bits, 0)) * s->many.m * pb; }
  out[2] = s->fold_tbl ? ((size_t)pc::curve_ops(s->curve).fold_rows << (s->fold_w - 2)) * s->fold_pts * pb : 0;
  out[3] = 0;
  for (int i = 0; i < PC_MSM_LANES; i++) out[3] += lane_bytes(s->lanes[i]);
  out[3] += lane_bytes(s->many.lane) + lane_bytes(s->bm.lanes[0]) + lane_bytes(s->bm.lanes[1]);
}
int pc_hip_srs_bytes_resident(const pc_srs* srs, size_t out[4]) {
  if (!srs || !out) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(srs->ctx->mu);
  srs_bytes(srs, out);
  return PC_OK;
}
int pc_hip_ctx_bytes_resident(pc_ctx* ctx, size_t out[6]) {
  if (!ctx || !out) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  for (int i = 0; i < 6; i++) out[i] = 0;
  out[0] = pc::dev_bytes_held(ctx->device);
  for (const pc_srs* s : ctx->keys) {
    size_t b[4]; srs_bytes(s, b);
    out[1] += b[0]; out[2] += b[1]; out[3] += b[2];
  }
  out[4] = ctx->be.scratch_bytes();
  out[5] = ctx->keys.size();
  return PC_OK;
}
int pc_hip_ctx_trim(pc_ctx* ctx) {
  if (!ctx) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    ctx->be.sync();
    ctx->be.trim();
    ctx->ntt_plans.clear();
    ctx->be.free(ctx->lig_arena); ctx->lig_arena = nullptr; ctx->lig_bytes = 0;      // pc_hip_ligero_commit's slab buffers
    for (int i = 0; i < 3; i++) { ctx->be.free(ctx->ipa_buf[i]); ctx->ipa_buf[i] = nullptr; ctx->ipa_bytes[i] = 0; }      // pc_hip_ipa_open_rounds' vectors
    // working keys that an opening handed back (pc_hip_ec_fold_from keeps one per committer key, with its three pipelines)
    std::vector<pc_srs*> cached;
    for (pc_srs* s : ctx->keys) if (s->work_cache) { cached.push_back(s->work_cache); s->work_cache = nullptr; }
    for (pc_srs* s : ctx->keys) if (s->fixed_cache) { cached.push_back(s->fixed_cache); s->fixed_cache = nullptr; }
    for (pc_srs* w : cached) { w->parent = nullptr; pc_hip_srs_free(w); }
    // idle pipelines give their sort / scan scratch back (the plan's own workspace stays: it is what makes the next call cheap)
    for (pc_srs* s : ctx->keys)
      for (int i = 0; i < PC_MSM_LANES; i++)
        if (s->lanes[i] && !s->lanes[i]->inflight) { s->lanes[i]->be.sync(); s->lanes[i]->be.trim(); if (s->lanes[i]->runner) s->lanes[i]->runner->trim(); }
    // the staging copies of HOST polynomials in the batch pipelines (pc_hip_msm_batch: 2 x 8 polynomials)
    for (pc_srs* s : ctx->keys)
      for (int i = 0; i < 2; i++)
        if (s->bm.stage[i] && s->bm.lanes[i]) { s->bm.lanes[i]->be.sync(); s->bm.lanes[i]->be.free(s->bm.stage[i]); s->bm.stage[i] = nullptr; }
    return (int)PC_OK;
  });
}

int pc_hip_set_timing(pc_ctx* ctx, int on) {
  if (!ctx) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    ctx->be.timing = on != 0;
    if (on) {     // the reference point of pc_hip_last_msm_marks_ms
      if (!ctx->epoch) PC_HIP_CHECK(hipEventCreate(&ctx->epoch));
      PC_HIP_CHECK(hipEventRecord(ctx->epoch, ctx->be.stream));
      PC_HIP_CHECK(hipEventSynchronize(ctx->epoch));
    }
    return (int)PC_OK;
  });
}

int pc_hip_last_msm_marks_ms(const pc_ctx* ctx, float out[8]) {
  if (!ctx || !out) return PC_ERR_INVALID_ARG;
  for (int i = 0; i < 8; i++) out[i] = ctx->marks[i];
  return PC_OK;
}

int pc_hip_last_msm_phases_ms(const pc_ctx* ctx, float out[8]) {
  if (!ctx || !out) return PC_ERR_INVALID_ARG;
  for (int i = 0; i < 8; i++) out[i] = ctx->phases[i];
  return PC_OK;
}

int pc_hip_last_msm_shape(const pc_ctx* ctx, uint32_t out[4]) {
  if (!ctx || !out) return PC_ERR_INVALID_ARG;
  for (int i = 0; i < 4; i++) out[i] = ctx->shape[i];
  return PC_OK;
}


// Stage a host buffer on the device (or pass a device pointer through).
struct Staged {
  pc::HipBackend& be; void* dev = nullptr; bool owned = false;
  // slot 0 / 1: the context's grow-only staging buffers (input / output of the call); -1 or a large request: transient
  Staged(pc::HipBackend& b, const void* p, pc_mem where, size_t bytes, bool copy_in, int slot = -1) : be(b) {
    if (where == PC_MEM_DEVICE) { dev = const_cast<void*>(p); return; }
    if (slot >= 0 && bytes <= pc::HipBackend::STAGE_KEEP) dev = be.stage(slot, bytes);
    else { dev = be.alloc(bytes); owned = true; }
    if (copy_in && bytes) be.copy_h2d(dev, p, bytes);
  }
  ~Staged() { if (owned) be.free(dev); }
};

static size_t ligero_slab_rows(size_t rows, size_t N);
static int ligero_commit_streamed(pc_ctx* ctx, pc_curve field_of, const char* mat, size_t rows, size_t in_cols, unsigned log_n, size_t S,
                                  pc_hash col_hash, pc_hash tree_hash, int len_prefix, char* ext_out, void* leaves_out_host, void* nodes_out_host);
int pc_hip_ntt_batch(pc_ctx* ctx, pc_curve field_of, const void* in, pc_mem where_in, size_t rows, size_t in_cols,
                     unsigned log_n, void* out, pc_mem where_out) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || (rows && (!in || !out))) return PC_ERR_INVALID_ARG;
  const unsigned max_lg = field_of == PC_CURVE_BN254 ? 28 : 32;
  if (log_n > max_lg) return PC_ERR_TOO_LARGE;
  if (log_n > PC_HIP_NTT_MAX_LOG_N) return PC_ERR_UNSUPPORTED;   // two LDS-staged passes: one factor must fit the 160 KB LDS
  if (in_cols > ((size_t)1 << log_n)) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  // host -> host (LinearEncode::encode over a whole matrix, the shim's encode_matrix): in slabs of rows, the encoded slabs on their way
  // back beside the kernels of the next ones -- pc_hip_ligero_commit's road without the digests
  if (where_in == PC_MEM_HOST && where_out == PC_MEM_HOST && rows && in_cols && rows < (1ull << 32))
    if (const size_t S = ligero_slab_rows(rows, (size_t)1 << log_n)) {
      ctx->ntt_phases[0] = ctx->ntt_phases[1] = 0;
      return ligero_commit_streamed(ctx, field_of, (const char*)in, rows, in_cols, log_n, S, PC_HASH_SHA256, PC_HASH_SHA256, 0, (char*)out, nullptr, nullptr);
    }
  return guarded(ctx, [&]() {
    if (rows == 0) return (int)PC_OK;
    auto key = std::make_pair((int)field_of, log_n);
    auto it = ctx->ntt_plans.find(key);
    if (it == ctx->ntt_plans.end()) {
      std::unique_ptr<NttRunner> r(pc::field_ops(field_of).make_ntt(ctx->be, log_n));
      it = ctx->ntt_plans.emplace(key, std::move(r)).first;
    }
    const size_t N = (size_t)1 << log_n;
    Staged sin(ctx->be, in, where_in, rows * in_cols * 32, true, 0);
    Staged sout(ctx->be, out, where_out, rows * N * 32, false, 1);
    ctx->be.n_ev = 0;
    it->second->run((const uint32_t*)sin.dev, rows, in_cols, (uint32_t*)sout.dev);
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out, sout.dev, rows * N * 32); else ctx->be.sync();
    ctx->ntt_phases[0] = ctx->ntt_phases[1] = 0;
    if (ctx->be.timing && ctx->be.n_ev >= 3) {
      (void)hipEventElapsedTime(&ctx->ntt_phases[0], ctx->be.ev[0], ctx->be.ev[1]);
      (void)hipEventElapsedTime(&ctx->ntt_phases[1], ctx->be.ev[1], ctx->be.ev[2]);
    }
    return (int)PC_OK;
  });
}

int pc_hip_last_ntt_phases_ms(const pc_ctx* ctx, float out[2]) {
  if (!ctx || !out) return PC_ERR_INVALID_ARG;
  out[0] = ctx->ntt_phases[0]; out[1] = ctx->ntt_phases[1];
  return PC_OK;
}

int pc_hip_poly_eval(pc_ctx* ctx, pc_curve field_of, const void* coeffs, pc_mem where_in, size_t n, const void* z_host,
                     void* out_host) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !z_host || !out_host || (n && !coeffs)) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    Staged sin(ctx->be, coeffs, where_in, n * 32, true, 0);
    const uint32_t* z = (const uint32_t*)z_host;
    pc::field_ops(field_of).poly_eval(ctx->be, (const uint32_t*)sin.dev, n, z, (uint32_t*)out_host, scan_fan());
    return (int)PC_OK;
  });
}

int pc_hip_poly_div_scan(pc_ctx* ctx, pc_curve field_of, const void* coeffs, pc_mem where_in, size_t n, const void* z_host,
                         const void* carry_in_host, void* out, pc_mem where_out) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !z_host || (n && (!coeffs || !out))) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (n == 0) return (int)PC_OK;
    Staged sin(ctx->be, coeffs, where_in, n * 32, true, 0);
    Staged sout(ctx->be, out, where_out, n * 32, false, 1);
    const uint32_t* z = (const uint32_t*)z_host; const uint32_t* cin = (const uint32_t*)carry_in_host;
    pc::field_ops(field_of).div_scan(ctx->be, (const uint32_t*)sin.dev, n, z, cin, (uint32_t*)sout.dev, scan_fan());
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out, sout.dev, n * 32);
    return (int)PC_OK;
  });
}

int pc_hip_points_sum(pc_curve curve, const void* points_xy, size_t count, void* out_xy) {
  if ((int)curve < 0 || (int)curve > 2 || !out_xy || (count && !points_xy)) return PC_ERR_INVALID_ARG;
  pc::curve_ops(curve).points_sum((const uint32_t*)points_xy, count, (uint32_t*)out_xy);
  return PC_OK;
}

int pc_hip_column_hash(pc_ctx* ctx, pc_curve field_of, const void* ext_mat, pc_mem where_in, size_t rows, size_t n_cols,
                       pc_hash hash, void* out_digests, pc_mem where_out) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || ((int)hash != PC_HASH_SHA256 && (int)hash != PC_HASH_BLAKE2S) ||
      (rows && n_cols && (!ext_mat || !out_digests))) return PC_ERR_INVALID_ARG;
  if (rows >= (1ull << 32) || n_cols >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (!n_cols) return (int)PC_OK;
    Staged sin(ctx->be, ext_mat, where_in, rows * n_cols * 32, true, 0);
    Staged sout(ctx->be, out_digests, where_out, n_cols * 32, false, 1);
    const uint32_t* e = (const uint32_t*)sin.dev; uint32_t* o = (uint32_t*)sout.dev;
    ctx->be.n_ev = 0; ctx->be.mark();
    pc::field_ops(field_of).column_hash(ctx->be, (int)hash, e, (uint32_t)rows, (uint32_t)n_cols, o);
    ctx->be.mark();
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out_digests, sout.dev, n_cols * 32); else ctx->be.sync();
    ctx->ntt_phases[0] = ctx->ntt_phases[1] = 0;
    if (ctx->be.timing && ctx->be.n_ev >= 2) (void)hipEventElapsedTime(&ctx->ntt_phases[0], ctx->be.ev[0], ctx->be.ev[1]);
    return (int)PC_OK;
  });
}

int pc_hip_column_hash_part(pc_ctx* ctx, pc_curve field_of, pc_hash hash, const void* ext_slab_dev, size_t rows, size_t n_cols, size_t rows_total,
                            size_t col0, size_t cols, int first, int last, void* state_dev, void* out_digests_dev) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || ((int)hash != PC_HASH_SHA256 && (int)hash != PC_HASH_BLAKE2S)) return PC_ERR_INVALID_ARG;
  if (rows >= (1ull << 32) || n_cols >= (1ull << 32) || rows_total >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  if (col0 > n_cols || cols > n_cols - col0 || rows > rows_total || (rows && cols && !ext_slab_dev)) return PC_ERR_INVALID_ARG;
  if (cols && ((!(first && last) && !state_dev) || (last && !out_digests_dev))) return PC_ERR_INVALID_ARG;
  if (!last && (rows & 1)) return PC_ERR_UNSUPPORTED;             // two rows fill one block: only the last slab may be odd
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (!cols) return (int)PC_OK;
    pc::field_ops(field_of).column_hash_part(ctx->be, (int)hash, (const uint32_t*)ext_slab_dev, (uint32_t)rows, (uint32_t)n_cols, (uint32_t)rows_total,
                                             (uint32_t)col0, (uint32_t)cols, first, last, (uint32_t*)state_dev, (uint32_t*)out_digests_dev);
    ctx->be.sync();
    return (int)PC_OK;
  });
}

int pc_hip_witness_poly(pc_ctx* ctx, pc_curve field_of, const void* coeffs, pc_mem where_in, size_t n, const void* z_host,
                        void* out, pc_mem where_out) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !z_host || (n && !coeffs) || (n > 1 && !out)) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (n <= 1) return (int)PC_OK;
    Staged sin(ctx->be, coeffs, where_in, n * 32, true, 0);
    Staged sout(ctx->be, out, where_out, (n - 1) * 32, false, 1);
    const uint32_t* z = (const uint32_t*)z_host;
    pc::field_ops(field_of).witness(ctx->be, (const uint32_t*)sin.dev, n, z, (uint32_t*)sout.dev, scan_fan());
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out, sout.dev, (n - 1) * 32);
    return (int)PC_OK;
  });
}

// KZG10::open without hiding (poly-commit/src/kzg10/mod.rs:287-310: compute_witness_polynomial :217-240, then
// open_with_witness_polynomial's MSM :255-258) as ONE call: W = sum_j q[j] * powers[base_offset + j], q = p / (x - z).
// The quotient never leaves the device.  Host coefficients of at least host_split_min() elements run in parts, top part first (its
// quotient needs nothing from below): copy + division of a part on the context's queue beside the accumulation of the part above on
// the key's pipeline, every further division with the carry q[hi] of the part above; ONE MSM over all parts (MsmPlan::begin_parts).
int pc_hip_kzg_open(pc_ctx* ctx, const pc_srs* srs_c, size_t base_offset, const void* coeffs, pc_mem where, size_t n, const void* z_host,
                    void* out_xy, int* out_is_infinity) {
  pc_srs* srs = const_cast<pc_srs*>(srs_c);
  if (!ctx || !srs || srs->ctx != ctx || !out_xy || !z_host || (n && !coeffs)) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  if (base_offset > srs->n || (n > 1 && n - 1 > srs->n - base_offset)) return PC_ERR_INVALID_ARG;     // the reference checks the degree before (kzg10/mod.rs:393-407)
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (n <= 1) { memset(out_xy, 0, (size_t)srs->aw * 4); if (out_is_infinity) *out_is_infinity = 1; return (int)PC_OK; }
    const pc::FieldOps& F = pc::field_ops(srs->curve);
    const uint32_t* z = (const uint32_t*)z_host;
    const size_t m = n - 1;                                       // quotient length; x[j] = p[j + 1]
    // the quotient (and, in the split path, the shifted coefficients): the context's grow-only staging up to STAGE_KEEP, transient
    // buffers above it -- one open of a 2^26-coefficient polynomial would otherwise pin 2 x 2 GiB until pc_hip_ctx_trim.  The transient
    // ones are freed when the call returns: every job that reads them is complete by then (StackJob / complete_job below).
    struct CallBuf {
      pc::HipBackend& be; void* dev; bool owned;
      CallBuf(pc::HipBackend& b, int slot, size_t bytes) : be(b), owned(bytes > pc::HipBackend::STAGE_KEEP) { dev = owned ? be.alloc(bytes) : be.stage(slot, bytes); }
      ~CallBuf() { if (owned) { (void)hipStreamSynchronize(be.stream); be.free(dev); } }
    };
    CallBuf qbuf(ctx->be, 1, m * 32);
    uint32_t* q = (uint32_t*)qbuf.dev;
    if (where == PC_MEM_DEVICE || n < host_split_min()) {
      Staged sin(ctx->be, coeffs, where, n * 32, true, 0);
      F.witness(ctx->be, (const uint32_t*)sin.dev, n, z, q, scan_fan());
      pc_job job;
      int rc = enqueue_job(ctx, srs, base_offset, q, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, m, out_xy, out_is_infinity, &job, false);
      if (rc != PC_OK) return rc;
      complete_job(ctx, &job);
      return (int)PC_OK;
    }
    CallBuf xbuf(ctx->be, 0, m * 32);
    uint32_t* x = (uint32_t*)xbuf.dev;
    const uint8_t* src = (const uint8_t*)coeffs + 32;
    if (host_parts() >= 2) {
      // the quotient in parts, TOP part first (its scan needs nothing from below; every further part takes the carry q[hi] of the one
      // above): copy + division of part t + 1 on the context's queue beside the accumulation of part t on the key's pipeline; one MSM
      const size_t K = host_parts();
      std::vector<uint32_t> carry(8);
      StackJob j(ctx);
      MsmLane* L = claim_lane(ctx, srs, out_xy, out_is_infinity, &j.job);
      L->runner->begin_parts(m);
      L->inflight = &j.job;
      std::vector<std::pair<size_t, size_t>> parts;        // (lo, hi) from the top; part t of the weights counted from the top
      for (size_t t = 0; t < K; t++) { const size_t hi = m - part_cut(m, t), lo = m - part_cut(m, t + 1); if (hi > lo) parts.push_back({lo, hi}); }
      for (size_t t = 0; t < parts.size(); t++) {
        const size_t lo = parts[t].first, hi = parts[t].second, len = hi - lo;
        ctx->be.copy_h2d(x + lo * 8, src + lo * 32, len * 32);
        if (t) ctx->be.copy_d2h(carry.data(), q + hi * 8, 32);
        F.div_scan(ctx->be, x + lo * 8, len, z, t ? carry.data() : nullptr, q + lo * 8, scan_fan());      // (returns with the stream drained)
        L->runner->add_part(srs->bases, (uint32_t)base_offset, lo, q + lo * 8, PC_MEM_DEVICE, len, true, t + 1 == parts.size());
      }
      complete_job(ctx, &j.job);
      return (int)PC_OK;
    }
    const size_t h = m / 2;
    std::vector<uint32_t> r1(srs->aw), r2(srs->aw), carry(8);
    StackJob j1(ctx), j2(ctx);
    ctx->be.copy_h2d(x + h * 8, src + h * 32, (m - h) * 32);
    F.div_scan(ctx->be, x + h * 8, m - h, z, nullptr, q + h * 8, scan_fan());          // (returns with the stream drained)
    int rc = enqueue_job(ctx, srs, base_offset + h, q + h * 8, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, m - h, r1.data(), nullptr, &j1.job, true);
    if (rc != PC_OK) return rc;
    ctx->be.copy_h2d(x, src, h * 32);
    ctx->be.copy_d2h(carry.data(), q + h * 8, 32);
    F.div_scan(ctx->be, x, h, z, carry.data(), q, scan_fan());
    rc = enqueue_job(ctx, srs, base_offset, q, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, r2.data(), nullptr, &j2.job, true);
    if (rc != PC_OK) return rc;
    complete_two(ctx, &j1.job, &j2.job);
    fold_two(srs, r1.data(), r2.data(), out_xy, out_is_infinity);
    return (int)PC_OK;
  });
}


// ---- IPA round kernels --------------------------------------------------------------------
int pc_hip_fr_fold(pc_ctx* ctx, pc_curve field_of, void* lo_dev, const void* hi_dev, size_t n_half, const void* s_host) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !s_host || (n_half && (!lo_dev || !hi_dev))) return PC_ERR_INVALID_ARG;
  if (n_half >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (n_half) pc::field_ops(field_of).fr_fold(ctx->be, (uint32_t*)lo_dev, (const uint32_t*)hi_dev, n_half, (const uint32_t*)s_host);
    return (int)PC_OK;
  });
}
int pc_hip_fr_dot(pc_ctx* ctx, pc_curve field_of, const void* a_dev, const void* b_dev, size_t n, void* out_host) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !out_host || (n && (!a_dev || !b_dev))) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    pc::field_ops(field_of).fr_dot(ctx->be, (const uint32_t*)a_dev, (const uint32_t*)b_dev, n, (uint32_t*)out_host);
    return (int)PC_OK;
  });
}
int pc_hip_ipa_fold_dots(pc_ctx* ctx, pc_curve field_of, void* coeffs_dev, void* z_dev, size_t m, const void* u_host, const void* u_inv_host,
                         void* out_dots_host) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !coeffs_dev || !z_dev || !out_dots_host || !m || (m & (m - 1))) return PC_ERR_INVALID_ARG;
  if ((u_host != nullptr) != (u_inv_host != nullptr)) return PC_ERR_INVALID_ARG;
  if (m >= (1ull << 31)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    pc::field_ops(field_of).ipa_fold_dots(ctx->be, (uint32_t*)coeffs_dev, (uint32_t*)z_dev, m, (const uint32_t*)u_host, (const uint32_t*)u_inv_host,
                                          (uint32_t*)out_dots_host);
    return (int)PC_OK;
  });
}
int pc_hip_fr_powers(pc_ctx* ctx, pc_curve field_of, const void* z_host, size_t n, void* out_dev) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !z_host || (n && !out_dev)) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (n) pc::field_ops(field_of).fr_powers(ctx->be, (const uint32_t*)z_host, n, (uint32_t*)out_dev);
    return (int)PC_OK;
  });
}
int pc_hip_ipa_key_scalars(pc_ctx* ctx, pc_curve field_of, const void* coeffs_dev, size_t m, void* s_dev, size_t n0,
                           const void* fold_u_host, size_t fold_m, void* out_l_dev, void* out_r_dev) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || !s_dev || !n0 || (n0 & (n0 - 1))) return PC_ERR_INVALID_ARG;
  if (fold_u_host && (fold_m < 2 || (fold_m & (fold_m - 1)) || fold_m > n0)) return PC_ERR_INVALID_ARG;
  if ((out_l_dev != nullptr) != (out_r_dev != nullptr)) return PC_ERR_INVALID_ARG;
  if (out_l_dev && (!coeffs_dev || m < 2 || (m & (m - 1)) || m > n0)) return PC_ERR_INVALID_ARG;
  if (n0 >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    pc::field_ops(field_of).ipa_key_scalars(ctx->be, (const uint32_t*)coeffs_dev, m, (uint32_t*)s_dev, n0, (const uint32_t*)fold_u_host, fold_m,
                                            (uint32_t*)out_l_dev, (uint32_t*)out_r_dev);
    return (int)PC_OK;
  });
}
// pc_hip_ligero_commit with the matrix AND the encoded matrix on the host (what LinearCodePCS::commit hands over and keeps,
// linear_codes/mod.rs:248-268): the encoded matrix is 2^log_n / in_cols times the input and its way back over PCIe is the longest
// leg of the call by far (config 5: 2 GiB, ~37 ms, against 9 ms in and 7 ms of kernels).  The rows are independent
// (compute_matrices, mod.rs:131-135) and the column digests chain over row slabs (pc_hip_column_hash_part), so the call runs in slabs of
// consecutive rows: slab s is copied in and encoded + absorbed on the context's queue while helper threads copy the slabs before it
// out on queues of their own.  helpers + 1 slab buffers each way instead of the whole encoded matrix in HBM.
//
// The caller's matrices are pageable memory, and a pageable copy blocks its calling thread while the runtime pins the pages (or finds
// them in its cache of recent pins), moves them by DMA and lets go of them.  With ONE helper the call took 41 ms as long as that cache
// hit -- the same buffers call after call in a quiet process -- and 80-82 ms whenever it did not (measured: from the moment a key with
// its tables had been freed, for as long as the probe ran): pinning and unpinning 2 GiB costs about as much host time as moving them
// takes, and one thread does the two one after the other.  So several helpers take the slabs in turn, one pinning while another's
// DMA runs (after a key was freed: 82 / 53 / 43-46 / 52 ms with 1 / 2 / 3 / 4 helpers; quiet: 40-42 ms with any).  The way IN stays with
// the runtime too: with one helper it was the slow side after a key was freed (a bounce-buffer memcpy at 11 GB/s of the calling
// thread), with three it hides under the way out, and registering the coefficient matrix's pages from the calling thread instead
// (page-aligned pieces just ahead of the copies, released behind them: built in round 5 as PC_HIP_LIGERO_PIN=1) measured 2-3 ms slower in
// both states (42.3 vs 40.0 ms quiet, 46.0 vs 42.8 ms after a key was freed: releasing a registration waits for the device) and was
// REMOVED in round 6: the library maps no caller memory into the device's address space (EXPERIMENTS 00).
// The whole-matrix path of the same call: 58-60 ms in either state.
// PC_HIP_LIGERO_SLAB_MB: encoded bytes per slab (default 32; 0 = the whole-matrix path), PC_HIP_LIGERO_HELPERS (default 3, at most 4),
// PC_HIP_LIGERO_TRACE=1: where the threads spent the call, on stderr;
// all read per call.  tools/ligero_stream_probe.py sweeps them in both states of the process.
static constexpr int LIG_MAX_HELPERS = 4;
static int lig_helpers() { const char* e = getenv("PC_HIP_LIGERO_HELPERS"); const int h = e ? atoi(e) : 3; return h < 1 ? 1 : h > LIG_MAX_HELPERS ? LIG_MAX_HELPERS : h; }
static size_t ligero_slab_rows(size_t rows, size_t N) {
  const char* e = getenv("PC_HIP_LIGERO_SLAB_MB");
  const double mb = e ? atof(e) : 32.0;
  if (!(mb > 0)) return 0;
  size_t s = (size_t)(mb * 1048576.0 / ((double)N * 32.0));
  s &= ~(size_t)1;                                     // every slab but the last holds an even number of rows (two rows fill a block)
  if (s < 2) s = 2;
  return s * 2 <= rows ? s : 0;                        // fewer than two slabs: nothing to overlap
}

static int ligero_commit_streamed(pc_ctx* ctx, pc_curve field_of, const char* mat, size_t rows, size_t in_cols, unsigned log_n, size_t S,
                                  pc_hash col_hash, pc_hash tree_hash, int len_prefix, char* ext_out, void* leaves_out_host, void* nodes_out_host) {
  const size_t N = (size_t)1 << log_n, n_slabs = (rows + S - 1) / S;
  const size_t in_row = in_cols * 32, ext_row = N * 32;
  const bool with_digests = nodes_out_host != nullptr;      // pc_hip_ntt_batch host -> host takes the same road without them
  const int LIG_HELPERS = lig_helpers(), LIG_BUFS = LIG_HELPERS + 1;      // one slab under the kernels, one with every helper
  void* in_dev[LIG_MAX_HELPERS + 1] = {}; void* ext_dev[LIG_MAX_HELPERS + 1] = {};
  void* state = nullptr; void* leaves = nullptr; void* nodes = nullptr; void* transient = nullptr;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t a_in = up(S * in_row), a_ext = up(S * ext_row), a_state = up(N * 48), a_leaves = up(N * 32), a_nodes = up((N > 1 ? N : 2) * 32);
  const size_t arena_bytes = LIG_BUFS * (a_in + a_ext) + a_state + a_leaves + a_nodes;
  static constexpr size_t LIGERO_KEEP = (size_t)512 << 20;
  std::vector<hipEvent_t> done(n_slabs, nullptr);
  // caller -> helpers: slabs whose kernels are queued (their event is recorded); helpers -> caller: slabs that have arrived
  std::mutex mu; std::condition_variable cv;
  size_t queued = 0; std::vector<char> arrived(n_slabs, 0); bool stop = false; int helper_rc = PC_OK; std::string helper_err;
  std::thread helpers[LIG_MAX_HELPERS];
  double tr_out[LIG_MAX_HELPERS] = {}, tr_in[3] = {0, 0, 0};      // PC_HIP_LIGERO_TRACE: helpers [copies out], caller [input buffer free, pin + copy in, slab buffer free]
  auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  int rc = guarded(ctx, [&]() {
    char* a;
    if (arena_bytes <= LIGERO_KEEP) {
      if (arena_bytes > ctx->lig_bytes) {
        ctx->be.sync(); ctx->be.free(ctx->lig_arena); ctx->lig_arena = nullptr; ctx->lig_bytes = 0;
        ctx->lig_arena = ctx->be.alloc(arena_bytes); ctx->lig_bytes = arena_bytes;
      }
      a = (char*)ctx->lig_arena;
    } else {
      a = (char*)(transient = ctx->be.alloc(arena_bytes));
    }
    for (int b = 0; b < LIG_BUFS; b++) { in_dev[b] = a; a += a_in; ext_dev[b] = a; a += a_ext; }
    state = a; a += a_state; leaves = a; a += a_leaves; nodes = a;
    for (int h = 0; h < LIG_HELPERS; h++)
      if (!ctx->lig_out_q[h]) PC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->lig_out_q[h], hipStreamNonBlocking));
    for (auto& e : done) PC_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto key = std::make_pair((int)field_of, log_n);
    auto it = ctx->ntt_plans.find(key);
    if (it == ctx->ntt_plans.end()) {
      std::unique_ptr<NttRunner> r(pc::field_ops(field_of).make_ntt(ctx->be, log_n));
      it = ctx->ntt_plans.emplace(key, std::move(r)).first;
    }
    NttRunner* ntt = it->second.get();
    for (int h = 0; h < LIG_HELPERS; h++)
      helpers[h] = std::thread([&, h]() {
        try {
          PC_HIP_CHECK(hipSetDevice(ctx->device));
          hipStream_t q = ctx->lig_out_q[h];
          for (size_t s = (size_t)h; s < n_slabs; s += LIG_HELPERS) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return queued > s || stop; }); if (queued <= s) return; }
            const size_t r0 = s * S, nr = std::min(S, rows - r0);
            const double t_a = now_ms();
            PC_HIP_CHECK(hipStreamWaitEvent(q, done[s], 0));
            PC_HIP_CHECK(hipMemcpyAsync(ext_out + r0 * ext_row, ext_dev[s % LIG_BUFS], nr * ext_row, hipMemcpyDeviceToHost, q));
            PC_HIP_CHECK(hipStreamSynchronize(q));
            tr_out[h] += now_ms() - t_a;
            { std::lock_guard<std::mutex> lk(mu); arrived[s] = 1; }
            cv.notify_all();
          }
        } catch (const std::exception& e) {
          { std::lock_guard<std::mutex> lk(mu); helper_rc = PC_ERR_HIP; helper_err = e.what(); std::fill(arrived.begin(), arrived.end(), 1); }   // releases the caller
          cv.notify_all();
        }
      });
    const bool marks = ctx->be.timing_marks(false);
    struct Restore { pc::HipBackend& be; bool m; ~Restore() { be.timing_marks(m); } } restore{ctx->be, marks};
    for (size_t s = 0; s < n_slabs; s++) {
      const int b = (int)(s % LIG_BUFS);
      const size_t r0 = s * S, nr = std::min(S, rows - r0);
      const double t_a = now_ms();
      if (s >= (size_t)LIG_BUFS) PC_HIP_CHECK(hipEventSynchronize(done[s - LIG_BUFS]));        // in_dev[b] has been read
      const double t_b = now_ms();
      {
        ctx->be.copy_h2d(in_dev[b], mat + r0 * in_row, nr * in_row);
      }
      const double t_c = now_ms();
      if (s >= (size_t)LIG_BUFS) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return arrived[s - LIG_BUFS] != 0; }); }   // ext_dev[b] is on the host
      tr_in[0] += t_b - t_a; tr_in[1] += t_c - t_b; tr_in[2] += now_ms() - t_c;
      { std::lock_guard<std::mutex> lk(mu); if (helper_rc != PC_OK) break; }
      ntt->run((const uint32_t*)in_dev[b], nr, in_cols, (uint32_t*)ext_dev[b]);
      if (with_digests)
        pc::field_ops(field_of).column_hash_part(ctx->be, (int)col_hash, (const uint32_t*)ext_dev[b], (uint32_t)nr, (uint32_t)N, (uint32_t)rows, 0u,
                                                 (uint32_t)N, s == 0, s + 1 == n_slabs, (uint32_t*)state, (uint32_t*)leaves);
      PC_HIP_CHECK(hipEventRecord(done[s], ctx->be.stream));
      { std::lock_guard<std::mutex> lk(mu); queued = s + 1; }
      cv.notify_all();
    }
    return (int)PC_OK;
  });
  if (rc == PC_OK && helper_rc == PC_OK && with_digests) {      // the tree and the small downloads run beside the last slabs' way out
    rc = pc_hip_merkle_tree(ctx, tree_hash, leaves, PC_MEM_DEVICE, N, len_prefix, nodes, PC_MEM_DEVICE);
    if (rc == PC_OK) rc = guarded(ctx, [&]() {
      unsigned h = 1; while (((size_t)1 << h) < N) h++;
      ctx->be.copy_d2h(nodes_out_host, nodes, (((size_t)1 << h) - 1) * 32);
      if (leaves_out_host) ctx->be.copy_d2h(leaves_out_host, leaves, N * 32);
      return (int)PC_OK;
    });
  }
  { std::lock_guard<std::mutex> lk(mu); stop = true; }
  cv.notify_all();
  for (auto& t : helpers) if (t.joinable()) t.join();
  if (rc == PC_OK && helper_rc != PC_OK) { ctx->last_error = helper_err; rc = helper_rc; }
  (void)guarded(ctx, [&]() {
    (void)hipStreamSynchronize(ctx->be.stream);
    for (int h = 0; h < LIG_HELPERS; h++) if (ctx->lig_out_q[h]) (void)hipStreamSynchronize(ctx->lig_out_q[h]);
    ctx->be.free(transient);
    for (auto e : done) if (e) (void)hipEventDestroy(e);
    return (int)PC_OK;
  });
  if (getenv("PC_HIP_LIGERO_TRACE"))
    fprintf(stderr, "[pc_hip] ligero slabs %zu x %zu rows: helpers' copies out %.1f / %.1f ms | caller in-buffer %.1f copy in %.1f out-buffer %.1f ms\n",
            n_slabs, S, tr_out[0], tr_out[LIG_HELPERS - 1], tr_in[0], tr_in[1], tr_in[2]);
  const float ph[4] = {0, 0, 0, with_digests ? ctx->ntt_phases[0] : 0.f};      // the slabs' kernels overlap the copies: only the tree has a bracket of its own
  memcpy(ctx->ligero_phases, ph, sizeof ph);
  return rc;
}

int pc_hip_ligero_commit(pc_ctx* ctx, pc_curve field_of, const void* mat, pc_mem where_in, size_t rows, size_t in_cols,
                         unsigned log_n, pc_hash col_hash, pc_hash tree_hash, int len_prefix, void* ext_out,
                         pc_mem where_ext, void* leaves_out_host, void* nodes_out_host) {
  if (!ctx || !rows || !in_cols || !mat || !nodes_out_host || log_n > 32 || in_cols > ((size_t)1 << log_n))
    return PC_ERR_INVALID_ARG;
  if (log_n > PC_HIP_NTT_MAX_LOG_N) return PC_ERR_UNSUPPORTED;
  auto known = [](pc_hash h) { return (int)h == PC_HASH_SHA256 || (int)h == PC_HASH_BLAKE2S; };
  if ((int)field_of < 0 || (int)field_of > 2 || !known(col_hash) || !known(tree_hash)) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  const size_t N = (size_t)1 << log_n;
  if (where_in == PC_MEM_HOST && ext_out && where_ext == PC_MEM_HOST && rows < (1ull << 32))
    if (const size_t S = ligero_slab_rows(rows, N))
      return ligero_commit_streamed(ctx, field_of, (const char*)mat, rows, in_cols, log_n, S, col_hash, tree_hash, len_prefix, (char*)ext_out,
                                    leaves_out_host, nodes_out_host);
  void* ext = nullptr; void* leaves = nullptr; void* nodes = nullptr;
  const bool own_ext = !(ext_out && where_ext == PC_MEM_DEVICE);
  int rc = guarded(ctx, [&]() {
    ext = own_ext ? ctx->be.alloc(rows * N * 32) : ext_out;
    leaves = ctx->be.alloc(N * 32);
    nodes = ctx->be.alloc((N > 1 ? N : 2) * 32);
    return (int)PC_OK;
  });
  float ph[4] = {0, 0, 0, 0};
  if (rc == PC_OK) rc = pc_hip_ntt_batch(ctx, field_of, mat, where_in, rows, in_cols, log_n, ext, PC_MEM_DEVICE);
  if (rc == PC_OK) { ph[0] = ctx->ntt_phases[0]; ph[1] = ctx->ntt_phases[1]; }
  if (rc == PC_OK) rc = pc_hip_column_hash(ctx, field_of, ext, PC_MEM_DEVICE, rows, N, col_hash, leaves, PC_MEM_DEVICE);
  if (rc == PC_OK) ph[2] = ctx->ntt_phases[0];
  if (rc == PC_OK) rc = pc_hip_merkle_tree(ctx, tree_hash, leaves, PC_MEM_DEVICE, N, len_prefix, nodes, PC_MEM_DEVICE);
  if (rc == PC_OK) ph[3] = ctx->ntt_phases[0];
  if (rc == PC_OK) rc = guarded(ctx, [&]() {
    unsigned h = 1; while (((size_t)1 << h) < N) h++;
    ctx->be.copy_d2h(nodes_out_host, nodes, (((size_t)1 << h) - 1) * 32);
    if (leaves_out_host) ctx->be.copy_d2h(leaves_out_host, leaves, N * 32);
    if (ext_out && where_ext == PC_MEM_HOST) ctx->be.copy_d2h(ext_out, ext, rows * N * 32);
    return (int)PC_OK;
  });
  (void)guarded(ctx, [&]() {
    if (own_ext && ext) ctx->be.free(ext);
    if (leaves) ctx->be.free(leaves);
    if (nodes) ctx->be.free(nodes);
    return (int)PC_OK;
  });
  memcpy(ctx->ligero_phases, ph, sizeof ph);
  return rc;
}
int pc_hip_last_ligero_phases_ms(const pc_ctx* ctx, float out[4]) {
  if (!ctx || !out) return PC_ERR_INVALID_ARG;
  memcpy(out, ctx->ligero_phases, sizeof ctx->ligero_phases);
  return PC_OK;
}

int pc_hip_merkle_tree(pc_ctx* ctx, pc_hash hash, const void* leaf_digests, pc_mem where_in, size_t n_leaves,
                       int len_prefix, void* out_nodes, pc_mem where_out) {
  if (!ctx || ((int)hash != PC_HASH_SHA256 && (int)hash != PC_HASH_BLAKE2S) || !n_leaves || !leaf_digests || !out_nodes)
    return PC_ERR_INVALID_ARG;
  if (n_leaves > (1ull << 31)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    unsigned h = 1; while (((size_t)1 << h) < n_leaves) h++;      // padded leaf count 2^h >= 2
    const size_t n_nodes = ((size_t)1 << h) - 1;
    Staged sin(ctx->be, leaf_digests, where_in, n_leaves * 32, true, 0);
    Staged sout(ctx->be, out_nodes, where_out, n_nodes * 32, false, 1);
    uint32_t* nodes = (uint32_t*)sout.dev;
    ctx->be.n_ev = 0; ctx->be.mark();
    for (int d = (int)h - 1; d >= 0; d--) {
      const bool bottom = d == (int)h - 1;
      const size_t cnt = (size_t)1 << d;
      const uint32_t* child = bottom ? (const uint32_t*)sin.dev : nodes + (((size_t)2 << d) - 1) * 8;
      uint32_t* parent = nodes + (cnt - 1) * 8;
      pc::merkle_level(ctx->be, (int)hash, child, parent, (uint32_t)n_leaves, bottom ? 1u : 0u, len_prefix ? 1u : 0u, cnt);
    }
    ctx->be.mark();
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out_nodes, sout.dev, n_nodes * 32); else ctx->be.sync();
    ctx->ntt_phases[0] = ctx->ntt_phases[1] = 0;
    if (ctx->be.timing && ctx->be.n_ev >= 2) (void)hipEventElapsedTime(&ctx->ntt_phases[0], ctx->be.ev[0], ctx->be.ev[1]);
    return (int)PC_OK;
  });
}

int pc_hip_matrix_columns(pc_ctx* ctx, const void* mat_dev, size_t rows, size_t n_cols, const uint32_t* indices_host, size_t t,
                          void* out, pc_mem where_out) {
  if (!ctx || !mat_dev || !rows || !n_cols || (t && (!indices_host || !out))) return PC_ERR_INVALID_ARG;
  if (rows * (uint64_t)t >= (1ull << 31) || n_cols >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  for (size_t j = 0; j < t; j++) if (indices_host[j] >= n_cols) return PC_ERR_INVALID_ARG;
  if (!t) return PC_OK;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    Staged sidx(ctx->be, indices_host, PC_MEM_HOST, t * 4, true, 0);
    Staged sout(ctx->be, out, where_out, rows * t * 32, false, 1);
    pc::gather_columns(ctx->be, (const uint32_t*)mat_dev, rows, n_cols, (const uint32_t*)sidx.dev, t, (uint32_t*)sout.dev);
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out, sout.dev, rows * t * 32); else ctx->be.sync();
    return (int)PC_OK;
  });
}

int pc_hip_fr_lincomb(pc_ctx* ctx, pc_curve field_of, const void* const* polys, pc_mem where_in, const size_t* lens,
                      size_t k, const void* xi_host, void* out, pc_mem where_out, size_t n_out) {
  if (!ctx || (int)field_of < 0 || (int)field_of > 2 || (k && (!polys || !lens || !xi_host)) || (n_out && !out))
    return PC_ERR_INVALID_ARG;
  if (n_out >= (1ull << 32) || k >= (1ull << 20)) return PC_ERR_TOO_LARGE;
  size_t total = 0;
  for (size_t j = 0; j < k; j++) {
    if (lens[j] >= (1ull << 32)) return PC_ERR_TOO_LARGE;
    if (lens[j] && !polys[j]) return PC_ERR_INVALID_ARG;
    total += lens[j];
  }
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (!n_out) return (int)PC_OK;
    // host polynomials are staged back to back in one device buffer
    Staged stage(ctx->be, nullptr, PC_MEM_HOST, where_in == PC_MEM_HOST ? total * 32 : 0, false, 0);
    std::vector<uint64_t> addr(k ? k : 1, 0); std::vector<uint32_t> len32(k ? k : 1, 0);
    size_t off = 0;
    for (size_t j = 0; j < k; j++) {
      len32[j] = (uint32_t)lens[j];
      if (where_in == PC_MEM_HOST) {
        if (lens[j]) ctx->be.copy_h2d((char*)stage.dev + off * 32, polys[j], lens[j] * 32);
        addr[j] = (uint64_t)(uintptr_t)((char*)stage.dev + off * 32); off += lens[j];
      } else addr[j] = (uint64_t)(uintptr_t)polys[j];
    }
    // the three small argument arrays side by side in the context's grow-only scratch (three hipMalloc / hipFree pairs per call before)
    const size_t kk = k ? k : 1, o_len = kk * 8, o_xi = (o_len + kk * 4 + 31) & ~(size_t)31;
    char* args = (char*)ctx->be.workspace(o_xi + kk * 32);
    ctx->be.copy_h2d(args, addr.data(), kk * 8);
    ctx->be.copy_h2d(args + o_len, len32.data(), kk * 4);
    if (k) ctx->be.copy_h2d(args + o_xi, xi_host, k * 32);
    Staged sout(ctx->be, out, where_out, n_out * 32, false, 1);
    ctx->be.n_ev = 0; ctx->be.mark();
    pc::field_ops(field_of).fr_lincomb(ctx->be, args, args + o_len, args + o_xi, k, sout.dev, n_out);
    ctx->be.mark();
    if (where_out == PC_MEM_HOST) ctx->be.copy_d2h(out, sout.dev, n_out * 32); else ctx->be.sync();
    ctx->ntt_phases[0] = ctx->ntt_phases[1] = 0;
    if (ctx->be.timing && ctx->be.n_ev >= 2) (void)hipEventElapsedTime(&ctx->ntt_phases[0], ctx->be.ev[0], ctx->be.ev[1]);
    return (int)PC_OK;
  });
}
static void drop_fold_table(pc_srs* srs);
int pc_hip_ec_fold(pc_ctx* ctx, pc_srs* srs, size_t n_half, const void* u_host) {
  if (!ctx || !srs || srs->ctx != ctx || !u_host || 2 * n_half > srs->n) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    for (int i = 0; i < PC_MSM_LANES; i++)      // queued MSMs still read the old key
      if (srs->lanes[i] && srs->lanes[i]->inflight) complete_job(ctx, srs->lanes[i]->inflight);
    if (!n_half) return (int)PC_OK;
    drop_table(srs);                            // the key changes: its window tables are stale
    drop_many(srs);
    drop_fold_table(srs);
    pc::curve_ops(srs->curve).ec_fold(ctx->be, srs->bases, n_half, (const uint32_t*)u_host);
    return (int)PC_OK;
  });
}
static void drop_fold_table(pc_srs* srs) {
  if (srs->fold_tbl) srs->ctx->be.free(srs->fold_tbl);
  srs->fold_tbl = nullptr; srs->fold_half = srs->fold_pts = 0; srs->fold_levels = 0; srs->fold_w = 2;
}
int pc_hip_srs_precompute_fold_ex(pc_ctx* ctx, pc_srs* srs, unsigned levels, unsigned naf_width) {
  if (!ctx || !srs || srs->ctx != ctx || levels > 2 || (naf_width && (naf_width < 2 || naf_width > 5))) return PC_ERR_INVALID_ARG;
  if (srs->n < 2 || (srs->n & 1) || (levels == 2 && (srs->n & 3))) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    drop_fold_table(srs);
    const size_t pb = (size_t)srs->aw * 4;
    const pc::CurveOps& ops = pc::curve_ops(srs->curve);
    // Refused above a share of the device's FREE memory (PC_HIP_FOLD_TABLE_MAX_FRAC, default 0.5) instead of driving a shared GPU out
    // of memory; the opening then runs the GLV ladder.  levels / naf_width 0: the largest form that fits that share -- two levels
    // from 2^16 points on (below, the second fold is a latency-bound ladder either way), digits as wide as the memory allows:
    //   rows = 131 * 2^(w-2), points per row = n / 2 or 3 n / 4:  a 2^22-point Pallas key: 17.6 GB (1, 2) .. 26 / 53 / 106 GB (2, 2 / 3 / 4)
    static const double frac = []() { const char* e = getenv("PC_HIP_FOLD_TABLE_MAX_FRAC"); double v = e ? atof(e) : 0.5; return v < 0 ? 0.0 : v > 1 ? 1.0 : v; }();
    size_t free_b = 0, total_b = 0;
    PC_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    auto bytes_of = [&](unsigned L, unsigned w) { return ((size_t)ops.fold_rows << (w - 2)) * (srs->n - (srs->n >> L)) * pb; };
    // the forms in the order of the opening times measured on a 2^22-point Pallas key (EXPERIMENTS 00): (2,4) 63.6 ms, (2,3) 66.8,
    // (1,4) 69.6, (2,2) 70.9, (1,3), (1,2) 73.7; the first one the caller's choice allows and the memory share holds
    static const unsigned order[6][2] = {{2, 4}, {2, 3}, {1, 4}, {2, 2}, {1, 3}, {1, 2}};
    const bool two_ok = srs->n >= ((size_t)1 << 16) && !(srs->n & 3);      // (below 2^16 points the second fold is a latency-bound ladder either way)
    unsigned L = 0, w = 0; size_t need = 0;
    for (const auto& f : order) {
      if (levels ? f[0] != levels : (f[0] == 2 && !two_ok)) continue;
      if (naf_width && f[1] != naf_width) continue;
      need = bytes_of(f[0], f[1]);
      if ((double)need <= frac * (double)free_b) { L = f[0]; w = f[1]; break; }
    }
    if (!L && naf_width == 5)      // width 5 (43 additions per term, twice the rows of width 4) only on request
      for (unsigned l : {2u, 1u}) {
        if (levels ? l != levels : (l == 2 && !two_ok)) continue;
        need = bytes_of(l, 5);
        if ((double)need <= frac * (double)free_b) { L = l; w = 5; break; }
      }
    if (!L) {
      ctx->last_error = "fold table of " + std::to_string(need >> 20) + " MiB exceeds " + std::to_string(frac) + " of the free device memory (" + std::to_string(free_b >> 20) + " MiB)";
      return (int)PC_ERR_UNSUPPORTED;
    }
    const size_t q = srs->n >> L, pts = srs->n - q;
    uint32_t* t = (uint32_t*)ctx->be.alloc(need);
    try { ops.fold_table_build(ctx->be, srs->bases + q * (size_t)srs->aw, pts, w, t); }
    catch (...) { ctx->be.free(t); throw; }
    srs->fold_tbl = t; srs->fold_half = q; srs->fold_pts = pts; srs->fold_levels = L; srs->fold_w = w;
    // The working key the first opening on this key will fold into (q points and three pipelines: ~10 ms of allocations) is made now,
    // with the table, instead of inside that opening; it waits in the key's cache like one an opening handed back (pc_hip_ctx_trim
    // releases it, pc_hip_ec_fold[2]_from re-creates it on demand).
    if (!srs->work_cache && !srs->parent && q >= 2) {
      pc_srs* wk = new (std::nothrow) pc_srs();
      if (wk) {
        wk->ctx = ctx; wk->curve = srs->curve; wk->n = q; wk->aw = srs->aw; wk->cfg = ctx->msm_cfg;
        ctx->keys.push_back(wk);
        try {
          wk->bases = (uint32_t*)ctx->be.alloc(q * pb);
          ctx->be.memset(wk->bases, 0, q * pb);                  // points at infinity until an opening folds into it
          for (int i = 0; i < PC_MSM_LANES; i++) srs_lane(wk, i);
          ctx->be.sync();
          wk->parent = srs; srs->work_cache = wk;
        } catch (...) { wk->parent = nullptr; srs_free_locked(wk); (void)hipGetLastError(); }      // no memory for it now: the opening will try again
      }
    }
    return (int)PC_OK;
  });
}
int pc_hip_srs_precompute_fold(pc_ctx* ctx, pc_srs* srs) {
  // PC_HIP_FOLD_TABLE="levels,width" (e.g. "1,2": the one-level table of plain NAF digits of rounds 3-5); unset: the library's choice
  static const std::pair<unsigned, unsigned> form = []() {
    const char* e = getenv("PC_HIP_FOLD_TABLE"); unsigned l = 0, w = 0;
    if (e) { l = (unsigned)atoi(e); const char* c = strchr(e, ','); if (c) w = (unsigned)atoi(c + 1); }
    return std::make_pair(l > 2 ? 0u : l, (w && (w < 2 || w > 5)) ? 0u : w);
  }();
  return pc_hip_srs_precompute_fold_ex(ctx, srs, form.first, form.second);
}
int pc_hip_srs_fold_table_info(const pc_srs* srs, unsigned* out_levels, unsigned* out_naf_width) {
  if (!srs) return PC_ERR_INVALID_ARG;
  if (out_levels) *out_levels = srs->fold_tbl ? srs->fold_levels : 0u;
  if (out_naf_width) *out_naf_width = srs->fold_tbl ? srs->fold_w : 0u;
  return PC_OK;
}

// the working key of `count` points an opening folds the committer key `par` into: the one the last opening handed back, or a new one
static pc_srs* working_key(pc_ctx* ctx, pc_srs* par, size_t count, bool* fresh) {
  pc_srs* dst = nullptr;
  if (par->work_cache && par->work_cache->n == count) { dst = par->work_cache; par->work_cache = nullptr; }      // buffers and pipelines of the last opening
  *fresh = dst == nullptr;
  if (*fresh) {
    dst = new (std::nothrow) pc_srs();
    if (!dst) return nullptr;
    dst->ctx = ctx; dst->curve = par->curve; dst->n = count; dst->aw = par->aw;
    ctx->keys.push_back(dst);
  }
  return dst;
}
int pc_hip_ec_fold_from(pc_ctx* ctx, const pc_srs* src, size_t n_half, const void* u_host, pc_srs** out) {
  if (!ctx || !src || src->ctx != ctx || !u_host || !out || !n_half || 2 * n_half > src->n) return PC_ERR_INVALID_ARG;
  *out = nullptr;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  pc_srs* par = const_cast<pc_srs*>(src);
  bool fresh = false;
  pc_srs* dst = working_key(ctx, par, n_half, &fresh);
  if (!dst) return PC_ERR_OOM;
  int rc = guarded(ctx, [&]() {
    if (fresh) { dst->bases = (uint32_t*)ctx->be.alloc(n_half * (size_t)src->aw * 4); dst->cfg = ctx->msm_cfg; }
    else { drop_table(dst); drop_many(dst); }
    const uint32_t* tbl = (src->fold_tbl && src->fold_levels == 1 && src->fold_half == n_half) ? src->fold_tbl : nullptr;
    pc::curve_ops(src->curve).ec_fold_to(ctx->be, src->bases, dst->bases, n_half, (const uint32_t*)u_host, tbl, src->fold_w);
    if (fresh) for (int i = 0; i < PC_MSM_LANES; i++) srs_lane(dst, i);      // all pipelines now: the next rounds' MSMs find them ready
    return (int)PC_OK;
  });
  if (rc != PC_OK) { dst->parent = nullptr; pc_hip_srs_free(dst); return rc; }
  if (!par->work_out) { dst->parent = par; par->work_out = dst; } else dst->parent = nullptr;
  *out = dst;
  return PC_OK;
}
int pc_hip_ec_fold2_from(pc_ctx* ctx, const pc_srs* src, size_t n_quarter, const void* u1_host, const void* u2_host, pc_srs** out) {
  if (!ctx || !src || src->ctx != ctx || !u1_host || !u2_host || !out || !n_quarter || 4 * n_quarter > src->n) return PC_ERR_INVALID_ARG;
  *out = nullptr;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  pc_srs* par = const_cast<pc_srs*>(src);
  const bool have = src->fold_tbl && src->fold_levels == 2 && src->fold_half == n_quarter;
  bool fresh = false;
  pc_srs* dst = working_key(ctx, par, n_quarter, &fresh);
  if (!dst) return PC_ERR_OOM;
  int rc = guarded(ctx, [&]() {
    const pc::CurveOps& ops = pc::curve_ops(src->curve);
    const size_t aw = (size_t)src->aw;
    if (fresh) { dst->bases = (uint32_t*)ctx->be.alloc(n_quarter * aw * 4); dst->cfg = ctx->msm_cfg; }
    else { drop_table(dst); drop_many(dst); }
    bool done = false;
    if (have) {
      // terms in the order of the table's points: K[q .. 2q) by u2, K[2q .. 3q) by u1, K[3q .. 4q) by u1 u2
      uint32_t u12[8];
      ops.fr_mul((const uint32_t*)u1_host, (const uint32_t*)u2_host, u12);
      const uint32_t* us[3] = {(const uint32_t*)u2_host, (const uint32_t*)u1_host, u12};
      done = ops.ec_fold_table(ctx->be, src->bases, dst->bases, n_quarter, src->fold_pts, 3, us, src->fold_w, src->fold_tbl);
    }
    if (!done) {
      // no two-level table on this key (or a split beyond its rows): the two folds one after the other, through a scratch half key
      uint32_t* tmp = (uint32_t*)ctx->be.alloc(2 * n_quarter * aw * 4);
      try {
        const uint32_t* tbl1 = (src->fold_tbl && src->fold_levels == 1 && src->fold_half == 2 * n_quarter) ? src->fold_tbl : nullptr;
        ops.ec_fold_to(ctx->be, src->bases, tmp, 2 * n_quarter, (const uint32_t*)u1_host, tbl1, src->fold_w);
        ops.ec_fold_to(ctx->be, tmp, tmp, n_quarter, (const uint32_t*)u2_host, nullptr, 2);
        ctx->be.copy_d2d(dst->bases, tmp, n_quarter * aw * 4);
        ctx->be.sync();
      } catch (...) { ctx->be.free(tmp); throw; }
      ctx->be.free(tmp);
    }
    if (fresh) for (int i = 0; i < PC_MSM_LANES; i++) srs_lane(dst, i);
    return (int)PC_OK;
  });
  if (rc != PC_OK) { dst->parent = nullptr; pc_hip_srs_free(dst); return rc; }
  if (!par->work_out) { dst->parent = par; par->work_out = dst; } else dst->parent = nullptr;
  *out = dst;
  return PC_OK;
}

int pc_hip_ipa_round2_msms(pc_ctx* ctx, const pc_srs* srs_c, const void* coeffs_dev, size_t n_quarter, const void* u1_host,
                           void* out_l_xy, int* out_l_is_infinity, void* out_r_xy, int* out_r_is_infinity) {
  pc_srs* srs = const_cast<pc_srs*>(srs_c);
  if (!ctx || !srs || srs->ctx != ctx || !coeffs_dev || !u1_host || !out_l_xy || !out_r_xy || !n_quarter || 4 * n_quarter > srs->n) return PC_ERR_INVALID_ARG;
  if (3 * n_quarter >= (1ull << 31)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    const size_t q = n_quarter, fw = 8;                                    // scalars: 8 words each
    const size_t bytes = 2 * 3 * q * fw * 4;
    // scalar vectors (c_r | 0 | u1 c_r) and (c_l | 0 | u1 c_l): the context's grow-only call buffer up to STAGE_KEEP, transient above it
    const bool owned = bytes > pc::HipBackend::STAGE_KEEP;
    uint32_t* buf = (uint32_t*)(owned ? ctx->be.alloc(bytes) : ctx->be.stage(0, bytes));
    struct Release { pc::HipBackend& be; void* p; ~Release() { if (p) { (void)hipStreamSynchronize(be.stream); be.free(p); } } } release{ctx->be, owned ? buf : nullptr};
    uint32_t* sl = buf; uint32_t* sr = buf + 3 * q * fw;
    const uint32_t* c = (const uint32_t*)coeffs_dev;
    ctx->be.memset(buf, 0, bytes);
    ctx->be.copy_d2d(sl, c + q * fw, q * fw * 4);                          // c_r = coeffs[q .. 2q)
    ctx->be.copy_d2d(sr, c, q * fw * 4);                                   // c_l = coeffs[0 .. q)
    const pc::FieldOps& fo = pc::field_ops(srs->curve);
    fo.fr_fold(ctx->be, sl + 2 * q * fw, c + q * fw, q, (const uint32_t*)u1_host);      // 0 + u1 c_r
    fo.fr_fold(ctx->be, sr + 2 * q * fw, c, q, (const uint32_t*)u1_host);               // 0 + u1 c_l
    ctx->be.sync();                                                        // the pipelines run on queues of their own
    StackJob jl(ctx), jr(ctx);
    int rc = enqueue_job(ctx, srs, 0, sl, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, 3 * q, out_l_xy, out_l_is_infinity, &jl.job, true);
    if (rc != PC_OK) return rc;
    rc = enqueue_job(ctx, srs, q, sr, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, 3 * q, out_r_xy, out_r_is_infinity, &jr.job, true);
    if (rc != PC_OK) return rc;
    if (!jl.job.done) complete_job(ctx, &jl.job);
    if (!jr.job.done) complete_job(ctx, &jr.job);
    return jl.job.status != PC_OK ? jl.job.status : jr.job.status;
  });
}

// The halving loop of InnerProductArgPC::open (ipa_pc/mod.rs:664-711) as ONE call: everything the round-by-round entry points above do,
// in the order poly_commit_amd/ipa.py and host/ipa_pc.hpp drive them, without a host language between the rounds (measured: 61.1-61.9 ms
// against 62.2 ms driven from Python at 2^22 -- the rounds are bound by the device's dependency chain; what the call buys a binding is
// one entry point instead of ~150 calls).  The transcript stays the caller's: `next_challenge`
// gets the round's l and r (affine, Montgomery x || y; all zeros = infinity) and returns the challenge u (Montgomery Fr).
static bool ipa_fixed_table() {      // PC_HIP_IPA_FIXED_TABLE=0: the late rounds run table-free on the working key (round 5's form)
  static const bool on = []() { const char* e = getenv("PC_HIP_IPA_FIXED_TABLE"); return !(e && !strcmp(e, "0")); }();
  return on;
}
static void* ipa_buffer(pc_ctx* ctx, int i, size_t bytes) {
  if (bytes > ctx->ipa_bytes[i]) {
    if (ctx->ipa_buf[i]) { ctx->be.sync(); ctx->be.free(ctx->ipa_buf[i]); ctx->ipa_buf[i] = nullptr; ctx->ipa_bytes[i] = 0; }
    ctx->ipa_buf[i] = ctx->be.alloc(bytes); ctx->ipa_bytes[i] = bytes;
  }
  return ctx->ipa_buf[i];
}
int pc_hip_ipa_open_rounds(pc_ctx* ctx, const pc_srs* comm_key, void* coeffs_dev, size_t n, const void* point_host, const void* h_prime_xy_host,
                           pc_ipa_challenge_fn next_challenge, void* user, size_t fixed_key_below,
                           void* out_l_vec_xy, void* out_r_vec_xy, void* out_final_key_xy, void* out_c_host, float* out_round_ms, float* out_fold_ms) {
  pc_srs* root = const_cast<pc_srs*>(comm_key);
  if (!ctx || !root || root->ctx != ctx || !coeffs_dev || !n || (n & (n - 1)) || n > root->n || !point_host || !h_prime_xy_host || !next_challenge ||
      !out_final_key_xy || !out_c_host || (n > 1 && (!out_l_vec_xy || !out_r_vec_xy))) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 31)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  const pc_curve curve = root->curve;
  const pc::CurveOps& ops = pc::curve_ops(curve);
  const size_t pb = (size_t)root->aw * 4;                       // bytes of an affine point
  // the default: 2^17 when the fixed key gets its window table (56.6 ms at 2^22 against 57.5 with 2^16 and 62.2 with 2^18), 2^16 without
  // one (EXPERIMENTS 00); any value below 2: the key is folded in every round
  if (!fixed_key_below) fixed_key_below = (size_t)1 << (ipa_fixed_table() ? 17 : 16);
  pc_srs* srs = root; bool owned = false;
  pc_srs* fixed = nullptr;                                      // the fixed key's own object (window table), or null: the working key serves the late rounds
  struct KeyGuard { pc_srs*& k; bool& owned; ~KeyGuard() { if (owned && k) pc_hip_srs_free(k); } } key_guard{srs, owned};
  void* z = nullptr; void* s_dev = nullptr; char* alr = nullptr;
  int rc = guarded(ctx, [&]() { z = ipa_buffer(ctx, 0, n * 32); return (int)PC_OK; });
  if (rc != PC_OK) return rc;
  char* c = (char*)coeffs_dev;
  rc = pc_hip_fr_powers(ctx, curve, point_host, n, z);                                          // z = (1, point, point^2, ..)   :641-649
  uint32_t dots[2][8];
  if (rc == PC_OK) rc = pc_hip_ipa_fold_dots(ctx, curve, c, z, n, nullptr, nullptr, dots);      // the inner products of the first round
  size_t n0 = 0;
  uint32_t u_prev[8], u_first[8], u[8], ui[8], one[8];
  bool have_u_prev = false, have_u_first = false;
  ops.fr_one(one);
  // a committer key with a two-level fold table (pc_hip_srs_precompute_fold_ex): round 1 leaves the key alone, round 2 runs on it by
  // linearity (pc_hip_ipa_round2_msms), the key after both folds comes out of the table in one step (pc_hip_ec_fold2_from)
  bool two_level = n == root->n && n >= 8 && n / 2 > fixed_key_below && root->fold_tbl && root->fold_levels == 2;
  std::vector<uint32_t> pts(4 * (size_t)root->aw);               // ml | hl | mr | hr
  size_t round = 0;
  using clk = std::chrono::steady_clock;
  for (size_t m = n; rc == PC_OK && m > 1; m /= 2, round++) {
    const auto t_round = clk::now();
    const size_t h = m / 2;
    if (!n0 && m <= fixed_key_below) {                                                          // from here on key[0 .. n0) stays fixed
      n0 = m;
      rc = guarded(ctx, [&]() { s_dev = ipa_buffer(ctx, 1, n0 * 32); alr = (char*)ipa_buffer(ctx, 2, 2 * n0 * 32); return (int)PC_OK; });
      if (rc == PC_OK) rc = pc_hip_fr_powers(ctx, curve, one, n0, s_dev);                        // s = (1, 1, ..)
      if (rc != PC_OK) break;
      have_u_prev = false;                                                                      // the key itself carries every fold so far
      // The fixed key serves 2 log2(n0) + 1 MSMs of n0 pairs: with a window table (one shared bucket set, fewer digits) each costs
      // ~0.15 ms less.  The table lives in a key object that belongs to the committer key and is REFILLED by every opening (points
      // copied on the device, table rebuilt in place: no allocation, the pipelines and their captured launch graphs stay): building a
      // new key with its table per opening cost 7-8 ms (EXPERIMENTS 00), refilling one costs what its kernels take.
      if (ipa_fixed_table() && srs == root && root->table && n0 >= root->cfg.tbl_min_n) {
        fixed = root;                                                                           // no fold yet and the committer key has its window table: it IS the fixed key, nothing to copy or refill
      } else if (ipa_fixed_table() && n0 >= ((size_t)1 << 12) && n0 <= ((size_t)1 << 18)) {
        const int frc = guarded(ctx, [&]() {
          pc_srs* fk = root->fixed_cache;
          if (fk && fk->n != n0) { root->fixed_cache = nullptr; srs_free_locked(fk); fk = nullptr; }
          if (!fk) {
            fk = new pc_srs();
            fk->ctx = ctx; fk->curve = curve; fk->n = n0; fk->aw = root->aw; fk->cfg = ctx->msm_cfg;
            ctx->keys.push_back(fk);
            root->fixed_cache = fk;
            fk->bases = (uint32_t*)ctx->be.alloc(n0 * pb);
          }
          ctx->be.copy_d2d(fk->bases, srs->bases, n0 * pb);
          static const unsigned fixed_c = []() { const char* e = getenv("PC_HIP_IPA_FIXED_C"); int v = e ? atoi(e) : 0; return (unsigned)(v >= 4 && v <= 22 ? v : 0); }();      // measurements: the table's window width
          if (!fk->table) return pc_hip_srs_precompute_ex(ctx, fk, fixed_c, 1, 0);                    // first opening: table, pipelines (full form: the key is small)
          for (int i = 0; i < PC_MSM_LANES; i++)
            if (fk->lanes[i] && fk->lanes[i]->inflight) complete_job(ctx, fk->lanes[i]->inflight);
          ops.window_table(ctx->be, fk->bases, (uint32_t)n0, fk->cfg.tbl_c, table_windows(fk, fk->cfg.tbl_c, false), fk->table, fk->cfg.tbl_pt_stride);
          return (int)PC_OK;
        });
        if (frc == PC_OK) fixed = root->fixed_cache;                                            // (on any failure the rounds run table-free on the working key, as before)
        else {                                                                                  // a half-made object is not kept: the next opening starts over
          (void)hipGetLastError();
          if (pc_srs* fk = root->fixed_cache) { root->fixed_cache = nullptr; (void)guarded(ctx, [&]() { srs_free_locked(fk); return (int)PC_OK; }); }
        }
      }
    }
    uint32_t* ml = pts.data(); uint32_t* hl = ml + root->aw; uint32_t* mr = hl + root->aw; uint32_t* hr = mr + root->aw;
    pc_job* jl = nullptr; pc_job* jr = nullptr;
    // l = cm_commit(key_l, coeffs_r) + h' <coeffs_r, z_l>;  r = cm_commit(key_r, coeffs_l) + h' <coeffs_l, z_r>          :666-675
    if (n0) {
      rc = pc_hip_ipa_key_scalars(ctx, curve, c, m, s_dev, n0, have_u_prev ? u_prev : nullptr, have_u_prev ? 2 * m : 0, alr, alr + 32 * n0);
      if (rc == PC_OK) rc = pc_hip_msm_async(ctx, fixed ? fixed : srs, 0, alr, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, ml, nullptr, &jl);
      if (rc == PC_OK) rc = pc_hip_msm_async(ctx, fixed ? fixed : srs, 0, alr + 32 * n0, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, mr, nullptr, &jr);
    } else if (have_u_first) {
      rc = pc_hip_ipa_round2_msms(ctx, srs, c, h, u_first, ml, nullptr, mr, nullptr);
    } else {
      rc = pc_hip_msm_async(ctx, srs, 0, c + 32 * h, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, ml, nullptr, &jl);
      if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, h, c, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, mr, nullptr, &jr);
    }
    if (rc == PC_OK) {                                                                          // beside the MSMs: h' * <.., ..>, one point each
      ops.point_mul((const uint32_t*)h_prime_xy_host, dots[0], hl);
      ops.point_mul((const uint32_t*)h_prime_xy_host, dots[1], hr);
    }
    const int w1 = jl ? pc_hip_job_wait(ctx, jl) : PC_OK, w2 = jr ? pc_hip_job_wait(ctx, jr) : PC_OK;      // always reap queued jobs
    if (rc == PC_OK) rc = w1 != PC_OK ? w1 : w2;
    if (rc != PC_OK) break;
    uint32_t* l = (uint32_t*)((char*)out_l_vec_xy + round * pb); uint32_t* r = (uint32_t*)((char*)out_r_vec_xy + round * pb);
    ops.points_sum(ml, 2, l);
    ops.points_sum(mr, 2, r);
    next_challenge(user, l, r, u);                                                              // :681-689, the caller's transcript
    ops.fr_inv(u, ui);
    rc = pc_hip_ipa_fold_dots(ctx, curve, c, z, h, u, ui, dots);                                // :691-697 + the next round's inner products
    if (rc != PC_OK) break;
    const auto t_fold = clk::now();
    bool folded = false;
    if (n0) { memcpy(u_prev, u, 32); have_u_prev = true; }                                      // applied to the factors at the top of the next round
    else if (two_level && !have_u_first && srs == root) { memcpy(u_first, u, 32); have_u_first = true; }
    else if (have_u_first) {
      pc_srs* work = nullptr;
      rc = pc_hip_ec_fold2_from(ctx, root, h, u_first, u, &work);
      if (rc == PC_OK) { srs = work; owned = true; }
      have_u_first = false; two_level = false; folded = true;
    } else if (owned) { rc = pc_hip_ec_fold(ctx, srs, h, u); folded = true; }                   // key_l += u key_r, normalised          :699-707
    else {
      pc_srs* work = nullptr;
      rc = pc_hip_ec_fold_from(ctx, srs, h, u, &work);                                          // the same fold, out of place: the committer key stays
      if (rc == PC_OK) { srs = work; owned = true; }
      folded = true;
    }
    const auto t_end = clk::now();
    if (out_fold_ms) out_fold_ms[round] = folded ? std::chrono::duration<float, std::milli>(t_end - t_fold).count() : 0.0f;
    if (out_round_ms) out_round_ms[round] = std::chrono::duration<float, std::milli>(t_end - t_round).count();
  }
  if (rc == PC_OK && n0 && have_u_prev) rc = pc_hip_ipa_key_scalars(ctx, curve, nullptr, 0, s_dev, n0, u_prev, 2, nullptr, nullptr);      // the last fold (size 2)
  if (rc == PC_OK) rc = n0 ? pc_hip_msm(ctx, fixed ? fixed : srs, 0, s_dev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, out_final_key_xy, nullptr)           // sum_j s_j K0_j
                           : pc_hip_srs_read(ctx, srs, 0, 1, out_final_key_xy);
  if (rc == PC_OK) rc = pc_hip_memcpy_d2h(ctx, out_c_host, coeffs_dev, 32);
  return rc;
}

int pc_hip_point_mul(pc_curve curve, const void* point_xy, const void* scalar_mont, void* out_xy) {
  if ((int)curve < 0 || (int)curve > 2 || !point_xy || !scalar_mont || !out_xy) return PC_ERR_INVALID_ARG;
  pc::curve_ops(curve).point_mul((const uint32_t*)point_xy, (const uint32_t*)scalar_mont, (uint32_t*)out_xy);
  return PC_OK;
}
int pc_hip_fixed_base_batch_mul(pc_ctx* ctx, pc_curve curve, const void* g_xy_host, const void* scalars_dev, size_t n,
                                void* out_points_dev) {
  if (!ctx || (int)curve < 0 || (int)curve > 2 || !g_xy_host || (n && (!scalars_dev || !out_points_dev))) return PC_ERR_INVALID_ARG;
  if (n >= (1ull << 32)) return PC_ERR_TOO_LARGE;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (!n) return (int)PC_OK;
    pc::curve_ops(curve).fixed_base(ctx->be, (const uint32_t*)g_xy_host, (const uint32_t*)scalars_dev, n, (uint32_t*)out_points_dev);
    return (int)PC_OK;
  });
}
int pc_hip_srs_read(pc_ctx* ctx, const pc_srs* srs, size_t offset, size_t count, void* out_xy) {
  if (!ctx || !srs || srs->ctx != ctx || offset + count > srs->n || (count && !out_xy)) return PC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  return guarded(ctx, [&]() {
    if (count) ctx->be.copy_d2h(out_xy, srs->bases + offset * (size_t)srs->aw, count * (size_t)srs->aw * 4);
    return (int)PC_OK;
  });
}

}  // extern "C"
