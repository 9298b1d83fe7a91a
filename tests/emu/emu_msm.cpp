// TEST-ONLY single-threaded stepping backend for the MSM orchestration in
// poly_commit_amd/csrc/msm.hpp.  It runs every kernel body as a plain loop over the lane
// index so that the indexing logic (chunking, partial lists, reduction levels, host tail)
// can be validated against the oracle on a machine without a GPU.  It is NOT part of the
// product library (libpc_hip.so is built from csrc/ only and instantiates HipBackend only).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../poly_commit_amd/csrc/msm.hpp"
#include "../../poly_commit_amd/csrc/poly.hpp"
#include "../../poly_commit_amd/csrc/ipa.hpp"
#include "../../poly_commit_amd/csrc/hash.hpp"
#include "../../poly_commit_amd/csrc/glv.hpp"
#include "../../poly_commit_amd/csrc/serialize.hpp"
#include "../../poly_commit_amd/csrc/fold_table.hpp"

struct CpuStepBackend {
  void* alloc(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
  void free(void* p) { if (p) ::free(p); }
  std::vector<uint8_t> ws;
  void* workspace(size_t bytes) { if (ws.size() < bytes) ws.assign(bytes, 0); return ws.data(); }
  void memset(void* p, int v, size_t bytes) { ::memset(p, v, bytes); }
  void mark() {}
  bool timing_marks(bool) { return true; }
  void aux_begin(int, int) {}
  int aux_end() { return -1; }
  void wait_token(int) {}
  int main_token() { return -1; }
  void record_done() {}
  void begin_tail() {}
  void end_tail() {}
  void quiesce() {}
  void wait_done() {}
  void* alloc_host(size_t b) { return calloc(1, b ? b : 1); }
  void free_host(void* p) { if (p) ::free(p); }
  void copy_d2h_async(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); }
  void sync() {}
  void copy_d2d(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); }
  void copy_d2h(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); }
  void copy_h2d(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); }
  void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n) {
    uint32_t acc = 0;
    for (size_t i = 0; i < n; i++) { uint32_t v = in[i]; out[i] = acc; acc += v; }
  }
  template <class C>
  void sort_entries(const pc::MsmGeom& g, const uint32_t* scalars, uint32_t* hist, uint32_t* offsets, uint32_t* cursor,
                    uint32_t* entries) {
    pc::sort_entries_atomic<C>(*this, g, scalars, hist, offsets, cursor, entries);
  }
  template <class C>
  void accumulate(const pc::AccumulateBody<C>& body, size_t lanes) { launch(body, lanes); }
  template <class C>
  void seg_reduce_tail(const pc::MsmGeom& g, uint32_t level, uint32_t slots, uint32_t* const* pk, uint32_t* const* pp, int cur,
                       const uint32_t* offsets, uint32_t* buckets) {
    pc::seg_reduce_tail_serial<C>(*this, g, level, slots, pk, pp, cur, offsets, buckets);
  }
  template <class C>
  void bucket_level(uint32_t K, uint32_t weight_off, uint32_t cnt, uint32_t n_old, bool bits, const uint32_t* x, const uint32_t* old_in,
                    uint32_t* out) {
    if (bits) {
      uint32_t lgK = 0; while ((1u << lgK) < K) lgK++;
      pc::BucketLevelBitsBody<C> b{K, lgK, weight_off, cnt, n_old, x, old_in, out};
      launch(b, (size_t)cnt * (1 + n_old));
    } else {
      pc::BucketLevelBody<C> b{K, weight_off, cnt, n_old, x, old_in, out};
      launch(b, (size_t)cnt * (1 + n_old));
    }
  }
  template <class B> void launch(const B& body, size_t lanes) {
    for (size_t i = 0; i < lanes; i++) body((uint32_t)i);
  }
  template <class B> void launch(const B& body, size_t lanes, int) { launch(body, lanes); }
};

template <class C>
static void run(const uint32_t* bases, const uint32_t* scalars, size_t n, uint32_t base_off, int c, int T, int T2, int K0,
                int from_mont, uint32_t* out) {
  CpuStepBackend be;
  pc::MsmConfig cfg; cfg.c = c; cfg.T = T; if (T2) { cfg.T2 = T2; cfg.T2b = T2 == 4 ? 6 : T2; } if (K0) { cfg.K0 = K0; cfg.K1 = K0 == 2 ? 4 : K0; cfg.coop_max_points = 64; cfg.seg_tail_lanes = (T2 == 5) ? 1 : 3; }
  pc::MsmPlan<C, CpuStepBackend> plan(be, n, cfg);
  plan.run(bases, base_off, scalars, n, from_mont != 0, out);
}

extern "C" void emu_msm(int curve, const uint32_t* bases, const uint32_t* scalars, size_t n, uint32_t base_off, int c,
                        int T, int T2, int K0, int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: run<pc_curve_bls12_381>(bases, scalars, n, base_off, c, T, T2, K0, from_mont, out); break;
    case 1: run<pc_curve_bn254>(bases, scalars, n, base_off, c, T, T2, K0, from_mont, out); break;
    case 2: run<pc_curve_pallas>(bases, scalars, n, base_off, c, T, T2, K0, from_mont, out); break;
  }
}

// A plan sized for an SRS of n_srs points serving a shorter call (KZG open: n_srs - 1 pairs; IPA rounds): the
// call's window width differs from the one of n_srs.  Returns 0, or 1 if the plan refused the call.
template <class C>
static int run_sized(const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off, int from_mont, uint32_t* out) {
  CpuStepBackend be;
  pc::MsmConfig cfg;
  pc::MsmPlan<C, CpuStepBackend> plan(be, n_srs, cfg);
  try { plan.run(bases, base_off, scalars, n, from_mont != 0, out); } catch (const pc::MsmCapacityError&) { return 1; }
  return 0;
}
extern "C" int emu_msm_sized(int curve, const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off,
                             int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: return run_sized<pc_curve_bls12_381>(bases, n_srs, scalars, n, base_off, from_mont, out);
    case 1: return run_sized<pc_curve_bn254>(bases, n_srs, scalars, n, base_off, from_mont, out);
    default: return run_sized<pc_curve_pallas>(bases, n_srs, scalars, n, base_off, from_mont, out);
  }
}

// table mode (pc_hip_srs_precompute): window table built by the same body the GPU runs, then the
// plan with one shared bucket set.  n_srs bases, MSM over [base_off, base_off + n).
template <class C>
static void run_table(const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off, int c, int K0,
                      int from_mont, uint32_t* out, bool glv = false) {
  CpuStepBackend be;
  constexpr int AW = 2 * pc::Fd<typename C::FqP>::N;
  const uint32_t Wd = pc::msm_num_windows(glv ? pc::GLV_HALF_BITS : (uint32_t)C::FrP::BITS, (uint32_t)c);      // GLV: the table of the 130-bit halves
  const uint32_t stride = AW + 8;             // padded entries, as the 128-byte-aligned BLS12-381 table
  std::vector<uint32_t> table((size_t)Wd * n_srs * stride);
  { pc::WindowTableBody<C> b{bases, (uint32_t)n_srs, (uint32_t)c, Wd, table.data(), stride}; be.launch(b, n_srs); }
  pc::MsmConfig cfg; cfg.tbl = table.data(); cfg.tbl_c = (uint32_t)c; cfg.tbl_stride = (uint32_t)n_srs; cfg.tbl_pt_stride = stride; cfg.tbl_min_n = 1;
  cfg.tbl_glv = glv;
  if (K0) { cfg.K0 = K0; cfg.K1 = K0 == 2 ? 4 : K0; cfg.coop_max_points = 64; cfg.seg_tail_lanes = 3; }
  pc::MsmPlan<C, CpuStepBackend> plan(be, n_srs, cfg);
  plan.run(bases, base_off, scalars, n, from_mont != 0, out);
}
extern "C" void emu_msm_table_glv(int curve, const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off,
                                  int c, int K0, int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: run_table<pc_curve_bls12_381>(bases, n_srs, scalars, n, base_off, c, K0, from_mont, out, true); break;
    case 1: run_table<pc_curve_bn254>(bases, n_srs, scalars, n, base_off, c, K0, from_mont, out, true); break;
    case 2: run_table<pc_curve_pallas>(bases, n_srs, scalars, n, base_off, c, K0, from_mont, out, true); break;
  }
}
// ONE MSM in `parts` parts (MsmPlan::begin_parts / add_part: pc_hip_msm and pc_hip_kzg_open on host memory): with the window table (c > 0:
// plain or GLV form) or table-free (c == 0); parts of unequal lengths, the last one possibly a single scalar.
template <class C>
static int run_parts(const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off, int c, int glv, int parts,
                     int from_mont, uint32_t* out) {
  CpuStepBackend be;
  constexpr int AW = 2 * pc::Fd<typename C::FqP>::N;
  pc::MsmConfig cfg;
  std::vector<uint32_t> table;
  if (c > 0) {
    const uint32_t Wd = pc::msm_num_windows(glv ? pc::GLV_HALF_BITS : (uint32_t)C::FrP::BITS, (uint32_t)c);
    const uint32_t stride = AW;
    table.resize((size_t)Wd * n_srs * stride);
    { pc::WindowTableBody<C> b{bases, (uint32_t)n_srs, (uint32_t)c, Wd, table.data(), stride}; be.launch(b, n_srs); }
    cfg.tbl = table.data(); cfg.tbl_c = (uint32_t)c; cfg.tbl_stride = (uint32_t)n_srs; cfg.tbl_pt_stride = stride; cfg.tbl_min_n = 1; cfg.tbl_glv = glv != 0;
  }
  cfg.coop_max_points = 64; cfg.seg_tail_lanes = 3;
  pc::MsmPlan<C, CpuStepBackend> plan(be, n_srs, cfg);
  try {
    plan.begin_parts(n);
    for (int k = 0; k < parts; k++) {
      // unequal cuts: thirds of what is left, the last part takes the rest
      const size_t first = n * (size_t)k / (size_t)parts + (k ? 1 : 0) * (k < parts ? 0 : 0);
      const size_t end = k + 1 == parts ? n : n * (size_t)(k + 1) / (size_t)parts;
      plan.add_part(bases, base_off + (uint32_t)first, scalars + first * (size_t)C::FrP::N, end - first, from_mont != 0, -1, k + 1 == parts);
    }
    plan.finish(out);
  } catch (const pc::MsmCapacityError&) { return 1; }
  return 0;
}
extern "C" int emu_msm_parts(int curve, const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off, int c, int glv,
                             int parts, int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: return run_parts<pc_curve_bls12_381>(bases, n_srs, scalars, n, base_off, c, glv, parts, from_mont, out);
    case 1: return run_parts<pc_curve_bn254>(bases, n_srs, scalars, n, base_off, c, glv, parts, from_mont, out);
    default: return run_parts<pc_curve_pallas>(bases, n_srs, scalars, n, base_off, c, glv, parts, from_mont, out);
  }
}
// the GLV split in 32-bit limbs (the device's) against the 64-bit host version: out = |k1| (5 limbs) | |k2| (5) | neg1 | neg2; returns 1 if they agree
template <class C> static int glv_split_check(const uint32_t* k, uint32_t* out) {
  pc::GlvHalves<C> hv; hv.split(k);
  uint64_t k64[4]; memcpy(k64, k, 32);
  const pc::GlvSplit ref = pc::glv_decompose<typename pc::GlvOf<C>::T>(k64);
  memcpy(out, hv.m[0], 20); memcpy(out + 5, hv.m[1], 20); out[10] = hv.neg[0]; out[11] = hv.neg[1];
  return memcmp(hv.m[0], ref.k1, 20) == 0 && memcmp(hv.m[1], ref.k2, 20) == 0 && hv.neg[0] == ref.neg1 && hv.neg[1] == ref.neg2;
}
extern "C" int emu_glv_split(int curve, const uint32_t* k, uint32_t* out) {
  switch (curve) {
    case 0: return glv_split_check<pc_curve_bls12_381>(k, out);
    case 1: return glv_split_check<pc_curve_bn254>(k, out);
    default: return glv_split_check<pc_curve_pallas>(k, out);
  }
}
extern "C" void emu_msm_table(int curve, const uint32_t* bases, size_t n_srs, const uint32_t* scalars, size_t n, uint32_t base_off,
                              int c, int K0, int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: run_table<pc_curve_bls12_381>(bases, n_srs, scalars, n, base_off, c, K0, from_mont, out); break;
    case 1: run_table<pc_curve_bn254>(bases, n_srs, scalars, n, base_off, c, K0, from_mont, out); break;
    case 2: run_table<pc_curve_pallas>(bases, n_srs, scalars, n, base_off, c, K0, from_mont, out); break;
  }
}

// many-MSM mode (pc_hip_msm_many): B independent MSMs of m pairs over the same m bases.
template <class C>
static void run_many(const uint32_t* bases, size_t m, const uint32_t* scalars, size_t B, int c, int K0, int from_mont, uint32_t* out) {
  CpuStepBackend be;
  constexpr int AW = 2 * pc::Fd<typename C::FqP>::N;
  const uint32_t Wd = pc::msm_num_windows(C::FrP::BITS, (uint32_t)c);
  std::vector<uint32_t> table((size_t)Wd * m * AW);
  { pc::WindowTableBody<C> b{bases, (uint32_t)m, (uint32_t)c, Wd, table.data(), (uint32_t)AW}; be.launch(b, m); }
  pc::MsmConfig cfg; cfg.tbl = table.data(); cfg.tbl_c = (uint32_t)c; cfg.tbl_stride = (uint32_t)m; cfg.tbl_pt_stride = AW; cfg.tbl_min_n = 1;
  if (K0) { cfg.tbl_K0 = K0; cfg.K1 = K0 == 2 ? 4 : 16; cfg.coop_max_points = 64; cfg.seg_tail_lanes = 3; }
  pc::MsmPlan<C, CpuStepBackend> plan(be, B * m, cfg, (uint32_t)B);
  plan.run(bases, 0, scalars, B * m, from_mont != 0, out);
}
static bool g_many_glv = false;      // emu_msm_many_vectors: the key's table is the GLV (half-scalar) one
extern "C" void emu_set_many_glv(int on) { g_many_glv = on != 0; }
// the same with the scalar vectors in separate buffers and fewer vectors than bucket sets (pc_hip_msm_batch's fast path)
template <class C>
static void run_many_vectors(const uint32_t* bases, size_t n_srs, size_t base_off, size_t m, const uint32_t* const* vecs, size_t count, size_t B, int c,
                             int from_mont, uint32_t* out) {
  CpuStepBackend be;
  constexpr int AW = 2 * pc::Fd<typename C::FqP>::N;
  const uint32_t Wd = pc::msm_num_windows(C::FrP::BITS, (uint32_t)c);
  std::vector<uint32_t> table((size_t)Wd * n_srs * AW);
  { pc::WindowTableBody<C> b{bases, (uint32_t)n_srs, (uint32_t)c, Wd, table.data(), (uint32_t)AW}; be.launch(b, n_srs); }
  pc::MsmConfig cfg; cfg.tbl = table.data(); cfg.tbl_c = (uint32_t)c; cfg.tbl_stride = (uint32_t)n_srs; cfg.tbl_pt_stride = AW; cfg.tbl_min_n = 1;
  cfg.tbl_glv = g_many_glv;
  cfg.tbl_K0 = 4; cfg.K1 = 16; cfg.coop_max_points = 64; cfg.seg_tail_lanes = 3;
  pc::MsmPlan<C, CpuStepBackend> plan(be, B * m, cfg, (uint32_t)B);
  std::vector<uint64_t> ptrs(count);
  for (size_t k = 0; k < count; k++) ptrs[k] = (uint64_t)(uintptr_t)vecs[k];
  plan.enqueue_vectors(bases, (uint32_t)base_off, ptrs.data(), count, m, from_mont != 0);
  plan.finish(out);
}
extern "C" void emu_msm_many_vectors(int curve, const uint32_t* bases, size_t n_srs, size_t base_off, size_t m, const uint32_t* const* vecs, size_t count,
                                     size_t B, int c, int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: run_many_vectors<pc_curve_bls12_381>(bases, n_srs, base_off, m, vecs, count, B, c, from_mont, out); break;
    case 1: run_many_vectors<pc_curve_bn254>(bases, n_srs, base_off, m, vecs, count, B, c, from_mont, out); break;
    case 2: run_many_vectors<pc_curve_pallas>(bases, n_srs, base_off, m, vecs, count, B, c, from_mont, out); break;
  }
}

extern "C" void emu_msm_many(int curve, const uint32_t* bases, size_t m, const uint32_t* scalars, size_t B, int c, int K0,
                             int from_mont, uint32_t* out) {
  switch (curve) {
    case 0: run_many<pc_curve_bls12_381>(bases, m, scalars, B, c, K0, from_mont, out); break;
    case 1: run_many<pc_curve_bn254>(bases, m, scalars, B, c, K0, from_mont, out); break;
    case 2: run_many<pc_curve_pallas>(bases, m, scalars, B, c, K0, from_mont, out); break;
  }
}

// field / curve unit hooks (32-bit limb code vs the 64-bit limb oracle)
template <class P> static void fop(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  typedef pc::Fd<P> F; F x = F::load(a), y = F::load(b), r;
  switch (op) {
    case 0: r = x.mul(y); break; case 1: r = x.add(y); break; case 2: r = x.sub(y); break;
    case 3: r = x.inv(); break; case 4: r = x.neg(); break; case 5: r = x.from_mont(); break;
    // the lazy (mod 2p) helpers of the bucket accumulation: raw outputs, inputs anywhere in [0, 2p]
    case 10: r = x.sub_lz(y); break; case 11: r = x.dbl_lz(); break; case 12: r = x.neg_lz(); break;
    case 13: r = x.neg_lz_canonical(); break; case 14: r = x.canon(); break;
    case 15: r = F::zero(); r.l[0] = x.is_zero_lz() ? 1u : 0u; break;
    case 16: r = x.mul_lz(y).canon(); break; case 17: r = x.sqr_lz().canon(); break; case 18: r = x.mul_add_mul_lz(y, y, x).canon(); break;
    // ... and their RAW values (the host forms skip the final subtraction exactly where the device's do): must stay below 2p
    case 19: r = x.mul_lz(y); break; case 20: r = x.sqr_lz(); break; case 21: r = x.mul_add_mul_lz(y, y, x); break;
    default: r = x.to_mont(); break;
  }
  r.store(out);
}
extern "C" void emu_fop(int curve, int which, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  switch (curve * 2 + which) {
    case 0: fop<pc_bls12_381_fq>(op, a, b, out); break; case 1: fop<pc_bls12_381_fr>(op, a, b, out); break;
    case 2: fop<pc_bn254_fq>(op, a, b, out); break;     case 3: fop<pc_bn254_fr>(op, a, b, out); break;
    case 4: fop<pc_pallas_fq>(op, a, b, out); break;    case 5: fop<pc_pallas_fr>(op, a, b, out); break;
  }
}
template <class C> static void ecop(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  typedef pc::XyzzD<C> Pt; typedef pc::AffD<C> A;
  A pa = A::load(a), pb = A::load(b);
  Pt r = Pt::from_affine(pa);
  switch (op) {
    case 0: r.add_affine(pb); break;
    case 4: r.add_affine_lz(pb, false); r = r.canonical(); break;                      // the lazy mixed addition, both signs
    case 5: r.add_affine_lz(pb.neg_if(true), true); r = r.canonical(); break;
    case 6: { Pt q = Pt::from_affine(pb); q = q.dbl(); r = q; r.add_affine_lz(pa, false); r.add_affine_lz(pb, true); r = r.canonical(); } break;  // 2 Pb + Pa - Pb, lazy chain
    case 1: { Pt q = Pt::from_affine(pb); q = q.dbl(); q.add_affine(pb.neg_if(true)); r.add(q); } break;  // via full add, non-trivial ZZ
    case 2: r = r.dbl(); break;
    case 3: r = Pt::dbl_affine(pa); break;
  }
  r.to_affine().store(out);
}
// The two-lane addition of k_bucket_level_coop2 (ec.hpp: HalfPt / HalfAdd), both lanes of the pair stepped on the host: the phases are
// the device's own code, the exchanges between them (DPP on the device) are plain assignments here.  mode bit 0 / 1: give the first /
// second operand a non-trivial ZZ (2P - P instead of P).
template <class C> static void half_add_pair(int mode, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  typedef pc::XyzzD<C> Pt; typedef pc::AffD<C> A; typedef pc::HalfPt<C> H; typedef pc::Fd<typename C::FqP> Fq;
  auto lift = [](const A& q, bool twist) { Pt r = Pt::from_affine(q); if (twist && !q.is_inf()) { r = r.dbl(); r.add_affine(q.neg_if(true)); } return r; };
  const Pt P = lift(A::load(a), mode & 1), Q = lift(A::load(b), mode & 2);
  H p[2] = {H::of(P, false), H::of(P, true)}, o[2] = {H::of(Q, false), H::of(Q, true)};
  if (o[0].b.is_zero()) { /* + infinity */ }
  else if (p[0].b.is_zero()) { p[0] = o[0]; p[1] = o[1]; }
  else {
    pc::HalfAdd<C> h[2];
    const bool dz[2] = {h[0].p1(p[0], o[0]), h[1].p1(p[1], o[1])};
    const bool pz = dz[0], rz = dz[1];                                   // even lane: P == 0, odd lane: R == 0
    if (pz) {
      Pt f; f.X = p[0].a; f.ZZ = p[0].b; f.Y = p[1].a; f.ZZZ = p[1].b;
      const Pt r = rz ? f.dbl() : Pt::infinity();
      p[0] = H::of(r, false); p[1] = H::of(r, true);
    } else {
      const Fq s0 = h[0].p2(p[0], o[0], false), s1 = h[1].p2(p[1], o[1], true);
      const Fq t0 = h[0].p3(false, s1), t1 = h[1].p3(true, s0);
      h[0].p4(p[0], false, t1); h[1].p4(p[1], true, t0);
    }
  }
  Pt r; r.X = p[0].a; r.ZZ = p[0].b; r.Y = p[1].a; r.ZZZ = p[1].b;
  // bit-identical to the one-lane addition, coordinate by coordinate
  Pt w = P; w.add(Q);
  uint32_t x[Pt::WORDS], y[Pt::WORDS]; r.store(x); w.store(y);
  out[2 * Fq::N] = memcmp(x, y, sizeof x) == 0 ? 1u : 0u;
  r.to_affine().store(out);
}
extern "C" void emu_half_add(int curve, int mode, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  switch (curve) {
    case 0: half_add_pair<pc_curve_bls12_381>(mode, a, b, out); break;
    case 1: half_add_pair<pc_curve_bn254>(mode, a, b, out); break;
    case 2: half_add_pair<pc_curve_pallas>(mode, a, b, out); break;
  }
}
extern "C" void emu_ecop(int curve, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  switch (curve) {
    case 0: ecop<pc_curve_bls12_381>(op, a, b, out); break;
    case 1: ecop<pc_curve_bn254>(op, a, b, out); break;
    case 2: ecop<pc_curve_pallas>(op, a, b, out); break;
  }
}

extern "C" void emu_witness(int curve, const uint32_t* p, size_t n, const uint32_t* z, uint32_t* q, uint32_t G) {
  CpuStepBackend be;
  switch (curve) {
    case 0: pc::witness_polynomial<pc_bls12_381_fr>(be, p, n, z, q, G, G == 64 ? 8 : G); break;
    case 1: pc::witness_polynomial<pc_bn254_fr>(be, p, n, z, q, G, 5); break;
    case 2: pc::witness_polynomial<pc_pallas_fr>(be, p, n, z, q, G, G); break;
  }
}

extern "C" void emu_div_scan(int curve, const uint32_t* x, size_t n, const uint32_t* z, const uint32_t* carry, uint32_t* out, uint32_t G) {
  CpuStepBackend be;
  switch (curve) {
    case 0: pc::div_scan<pc_bls12_381_fr>(be, x, n, z, carry, out, G); break;
    case 1: pc::div_scan<pc_bn254_fr>(be, x, n, z, carry, out, G); break;
    case 2: pc::div_scan<pc_pallas_fr>(be, x, n, z, carry, out, G); break;
  }
}

extern "C" void emu_poly_eval(int curve, const uint32_t* x, size_t n, const uint32_t* z, uint32_t* out, uint32_t G) {
  CpuStepBackend be;
  switch (curve) {
    case 0: pc::poly_eval<pc_bls12_381_fr>(be, x, n, z, out, G); break;
    case 1: pc::poly_eval<pc_bn254_fr>(be, x, n, z, out, G); break;
    case 2: pc::poly_eval<pc_pallas_fr>(be, x, n, z, out, G); break;
  }
}

// IPA round bodies, stepped
template <class C>
static void ipa_bodies(uint32_t* key, size_t half, const uint32_t* u_canon, uint32_t* lo, const uint32_t* hi, const uint32_t* s_mont,
                       uint32_t* dot_out, const uint32_t* z_mont, uint32_t* pow_out, size_t npow) {
  typedef typename C::FrP FrP; typedef pc::Fd<FrP> F;
  CpuStepBackend be;
  { pc::EcFoldBody<C> b; b.key = key; b.half = (uint32_t)half; b.naf.from_scalar(u_canon); be.launch(b, half); }
  { uint32_t lanes = 7; std::vector<uint32_t> part(lanes * FrP::N);
    pc::FrDotBody<FrP> b{lo, hi, (uint32_t)half, lanes, part.data()}; be.launch(b, lanes);
    F acc = F::zero(); for (uint32_t t = 0; t < lanes; t++) acc = acc.add(F::load(&part[t * FrP::N])); acc.store(dot_out); }
  { pc::FrFoldBody<FrP> b{lo, hi, F::load(s_mont)}; be.launch(b, half); }
  { pc::FrPowersBody<FrP> b; b.out = pow_out; F w = F::load(z_mont); for (int k = 0; k < 32; k++) { w.store(b.pt.w[k]); w = w.sqr(); } be.launch(b, npow); }
}
extern "C" void emu_ipa_bodies(int curve, uint32_t* key, size_t half, const uint32_t* u_canon, uint32_t* lo, const uint32_t* hi,
                               const uint32_t* s_mont, uint32_t* dot_out, const uint32_t* z_mont, uint32_t* pow_out, size_t npow) {
  switch (curve) {
    case 0: ipa_bodies<pc_curve_bls12_381>(key, half, u_canon, lo, hi, s_mont, dot_out, z_mont, pow_out, npow); break;
    case 1: ipa_bodies<pc_curve_bn254>(key, half, u_canon, lo, hi, s_mont, dot_out, z_mont, pow_out, npow); break;
    case 2: ipa_bodies<pc_curve_pallas>(key, half, u_canon, lo, hi, s_mont, dot_out, z_mont, pow_out, npow); break;
  }
}

// the fixed-key late rounds (IpaKeyScalarUpdateBody / IpaFixedKeyScalarsBody) and the batched normalisation
template <class FrP>
static void key_scalars(const uint32_t* c, uint32_t m, uint32_t* s, uint32_t n0, const uint32_t* fold_u, uint32_t fold_m, uint32_t* out_l, uint32_t* out_r) {
  CpuStepBackend be;
  if (fold_u) { pc::IpaKeyScalarUpdateBody<FrP> b{s, fold_m, pc::Fd<FrP>::load(fold_u)}; be.launch(b, n0); }
  if (out_l) { pc::IpaFixedKeyScalarsBody<FrP> b{c, s, m, out_l, out_r}; be.launch(b, n0); }
}
extern "C" void emu_ipa_key_scalars(int curve, const uint32_t* c, uint32_t m, uint32_t* s, uint32_t n0, const uint32_t* fold_u, uint32_t fold_m,
                                    uint32_t* out_l, uint32_t* out_r) {
  switch (curve) {
    case 0: key_scalars<pc_bls12_381_fr>(c, m, s, n0, fold_u, fold_m, out_l, out_r); break;
    case 1: key_scalars<pc_bn254_fr>(c, m, s, n0, fold_u, fold_m, out_l, out_r); break;
    case 2: key_scalars<pc_pallas_fr>(c, m, s, n0, fold_u, fold_m, out_l, out_r); break;
  }
}
template <class C>
static void glv_fold_batched(uint32_t* key, size_t half, const uint64_t* k_canon, uint32_t K) {
  typedef typename pc::GlvOf<C>::T G;
  constexpr int FN = C::FqP::N;
  pc::GlvSplit sp = pc::glv_decompose<G>(k_canon);
  std::vector<uint32_t> ws(half * 4 * FN);
  pc::EcFoldGlvBody<C> body; body.key = key; body.half = (uint32_t)half; body.jac_out = ws.data();
  body.n1.from_scalar(sp.k1); body.n2.from_scalar(sp.k2); body.neg1 = sp.neg1; body.neg2 = sp.neg2;
  for (int i = 0; i < FN; i++) body.beta[i] = G::BETA_MONT[i];
  CpuStepBackend be; be.launch(body, half);
  pc::JacBatchAffineBody<C> nb{ws.data(), ws.data() + half * 3 * FN, key, (uint32_t)half, K};
  be.launch(nb, (half + K - 1) / K);
}
extern "C" void emu_glv_fold_batched(int curve, uint32_t* key, size_t half, const uint64_t* k_canon, uint32_t K) {
  switch (curve) {
    case 0: glv_fold_batched<pc_curve_bls12_381>(key, half, k_canon, K); break;
    case 1: glv_fold_batched<pc_curve_bn254>(key, half, k_canon, K); break;
    case 2: glv_fold_batched<pc_curve_pallas>(key, half, k_canon, K); break;
  }
}

// the batched window-table build against the one-lane-per-base build (both are what the GPU runs)
template <class C>
static int table_builds_agree(const uint32_t* bases, uint32_t n, uint32_t c, uint32_t stride, uint32_t K) {
  const uint32_t Wd = pc::msm_num_windows(C::FrP::BITS, c);
  std::vector<uint32_t> a((size_t)Wd * n * stride, 0xabababab), b((size_t)Wd * n * stride, 0xabababab);
  CpuStepBackend be;
  { pc::WindowTableBody<C> body{bases, n, c, Wd, a.data(), stride}; be.launch(body, n); }
  pc::build_window_table_batched<C>(be, bases, n, c, Wd, b.data(), stride, K);
  std::vector<uint32_t> d((size_t)Wd * n * stride, 0xabababab);      // and the one-normalisation build of small keys
  pc::build_window_table_oneshot<C>(be, bases, n, c, Wd, d.data(), stride, K);
  return a == b && a == d;
}
extern "C" int emu_table_builds_agree(int curve, const uint32_t* bases, uint32_t n, uint32_t c, uint32_t stride, uint32_t K) {
  switch (curve) {
    case 0: return table_builds_agree<pc_curve_bls12_381>(bases, n, c, stride, K);
    case 1: return table_builds_agree<pc_curve_bn254>(bases, n, c, stride, K);
    default: return table_builds_agree<pc_curve_pallas>(bases, n, c, stride, K);
  }
}

// fixed-base window-table multiplication + XYZZ batch normalisation, stepped (table built with the device bodies' host twins)
template <class C>
static void fixed_base_table(const uint32_t* g, const uint32_t* scalars_mont, size_t n, uint32_t K, uint32_t* out) {
  constexpr int AW = 2 * pc::Fd<typename C::FqP>::N, XW = pc::XyzzD<C>::WORDS, FW = pc::Fd<typename C::FqP>::N;
  const uint32_t Wd = pc::msm_num_windows(C::FrP::BITS, pc::FIXED_BASE_C), half = 1u << (pc::FIXED_BASE_C - 1);
  std::vector<uint32_t> tbl((size_t)Wd * half * AW), res(n * XW), scr(n * FW);
  pc::XyzzD<C> base = pc::XyzzD<C>::from_affine(pc::AffD<C>::load(g));
  for (uint32_t w = 0; w < Wd; w++) {
    pc::XyzzD<C> cur = base;
    for (uint32_t d = 0; d < half; d++) { cur.to_affine().store(&tbl[((size_t)w * half + d) * AW]); cur.add(base); }
    for (uint32_t k = 0; k < pc::FIXED_BASE_C; k++) base = base.dbl();
  }
  CpuStepBackend be;
  pc::FixedBaseTableMulBody<C> body{scalars_mont, tbl.data(), Wd, res.data()};
  be.launch(body, n);
  pc::XyzzBatchAffineBody<C> nb{res.data(), scr.data(), out, (uint32_t)n, K};
  be.launch(nb, (n + K - 1) / K);
}
extern "C" void emu_fixed_base_table(int curve, const uint32_t* g, const uint32_t* scalars_mont, size_t n, uint32_t K, uint32_t* out) {
  switch (curve) {
    case 0: fixed_base_table<pc_curve_bls12_381>(g, scalars_mont, n, K, out); break;
    case 1: fixed_base_table<pc_curve_bn254>(g, scalars_mont, n, K, out); break;
    case 2: fixed_base_table<pc_curve_pallas>(g, scalars_mont, n, K, out); break;
  }
}

// ark-serialize bytes -> affine points (SrsDecodeBody), stepped
template <class C>
static uint32_t srs_decode(const uint8_t* in, uint32_t n, int compressed, uint32_t* out) {
  uint32_t bad = 0;
  pc::SrsDecodeBody<C> b{in, n, compressed ? 1u : 0u, C::FqP::BITS == 381 ? 1u : 0u, out, &bad};
  CpuStepBackend be; be.launch(b, n);
  return bad;
}
extern "C" uint32_t emu_srs_decode(int curve, const uint8_t* in, uint32_t n, int compressed, uint32_t* out) {
  switch (curve) {
    case 0: return srs_decode<pc_curve_bls12_381>(in, n, compressed, out);
    case 1: return srs_decode<pc_curve_bn254>(in, n, compressed, out);
    default: return srs_decode<pc_curve_pallas>(in, n, compressed, out);
  }
}

// affine points -> ark-serialize bytes (SrsEncodeBody), stepped
template <class C>
static void srs_encode(const uint32_t* pts, uint32_t n, int compressed, uint8_t* out) {
  pc::SrsEncodeBody<C> b{pts, n, compressed ? 1u : 0u, C::FqP::BITS == 381 ? 1u : 0u, out};
  CpuStepBackend be; be.launch(b, n);
}
extern "C" void emu_srs_encode(int curve, const uint32_t* pts, uint32_t n, int compressed, uint8_t* out) {
  switch (curve) {
    case 0: srs_encode<pc_curve_bls12_381>(pts, n, compressed, out); break;
    case 1: srs_encode<pc_curve_bn254>(pts, n, compressed, out); break;
    default: srs_encode<pc_curve_pallas>(pts, n, compressed, out); break;
  }
}

extern "C" void emu_fixed_base(int curve, const uint32_t* g, const uint32_t* scalars_mont, size_t n, uint32_t* out) {
  CpuStepBackend be;
  switch (curve) {
    case 0: { pc::FixedBaseMulBody<pc_curve_bls12_381> b; b.scalars = scalars_mont; b.out = out; for (int i = 0; i < 24; i++) b.g[i] = g[i]; be.launch(b, n); } break;
    case 1: { pc::FixedBaseMulBody<pc_curve_bn254> b; b.scalars = scalars_mont; b.out = out; for (int i = 0; i < 16; i++) b.g[i] = g[i]; be.launch(b, n); } break;
    case 2: { pc::FixedBaseMulBody<pc_curve_pallas> b; b.scalars = scalars_mont; b.out = out; for (int i = 0; i < 16; i++) b.g[i] = g[i]; be.launch(b, n); } break;
  }
}

template <class FrP>
static void colhash(const uint32_t* ext, uint32_t rows, uint32_t n_cols, int hash_id, uint32_t* out) {
  CpuStepBackend be;
  if (hash_id == 0) { pc::ColumnHashBody<FrP, pc::Sha256> b{ext, rows, n_cols, out}; be.launch(b, n_cols); }
  else { pc::ColumnHashBody<FrP, pc::Blake2s256> b{ext, rows, n_cols, out}; be.launch(b, n_cols); }
}
extern "C" void emu_column_hash(int curve, const uint32_t* ext, uint32_t rows, uint32_t n_cols, int hash_id, uint32_t* out) {
  switch (curve) {
    case 0: colhash<pc_bls12_381_fr>(ext, rows, n_cols, hash_id, out); break;
    case 1: colhash<pc_bn254_fr>(ext, rows, n_cols, hash_id, out); break;
    case 2: colhash<pc_pallas_fr>(ext, rows, n_cols, hash_id, out); break;
  }
}

template <class C>
static void glv_fold(uint32_t* key, size_t half, const uint64_t* k_canon, uint32_t* split_out) {
  typedef typename pc::GlvOf<C>::T G;
  pc::GlvSplit sp = pc::glv_decompose<G>(k_canon);
  memcpy(split_out, &sp, sizeof(sp));
  pc::EcFoldGlvBody<C> body; body.key = key; body.half = (uint32_t)half;
  body.n1.from_scalar(sp.k1); body.n2.from_scalar(sp.k2); body.neg1 = sp.neg1; body.neg2 = sp.neg2;
  for (int i = 0; i < C::FqP::N; i++) body.beta[i] = G::BETA_MONT[i];
  CpuStepBackend be; be.launch(body, half);
}
extern "C" void emu_glv_fold(int curve, uint32_t* key, size_t half, const uint64_t* k_canon, uint32_t* split_out) {
  switch (curve) {
    case 0: glv_fold<pc_curve_bls12_381>(key, half, k_canon, split_out); break;
    case 1: glv_fold<pc_curve_bn254>(key, half, k_canon, split_out); break;
    case 2: glv_fold<pc_curve_pallas>(key, half, k_canon, split_out); break;
  }
}
extern "C" void emu_glv_lambda(int curve, uint64_t* out) {
  switch (curve) {
    case 0: memcpy(out, pc_glv_bls12_381::LAMBDA, 32); break;
    case 1: memcpy(out, pc_glv_bn254::LAMBDA, 32); break;
    case 2: memcpy(out, pc_glv_pallas::LAMBDA, 32); break;
  }
}

// the fold table of an IPA committer key (fold_table.hpp): built for `levels` folds and width-w NAF digits over key[n >> levels .. n), then
// the fold(s) out of it -- one term (u1) or three (u2, u1, u1 u2) -- exactly the calls pc_hip_srs_precompute_fold_ex / pc_hip_ec_fold_from /
// pc_hip_ec_fold2_from make.  Returns 0 if the split did not fit the table (the library then takes the ladder).
template <class C>
static int fold_table(const uint32_t* key, uint32_t n, uint32_t levels, uint32_t w, const uint32_t* u1_mont, const uint32_t* u2_mont, uint32_t* out) {
  typedef pc::Fd<typename C::FrP> Fr;
  constexpr int AW = 2 * C::FqP::N;
  const size_t q = n >> levels, pts = n - q;
  std::vector<uint32_t> table(((size_t)pc::FOLD_ROWS << (w - 2)) * pts * AW, 0xabababab);
  CpuStepBackend be;
  pc::fold_table_build_run<C>(be, key + q * AW, pts, w, table.data());
  uint32_t u12[C::FrP::N];
  Fr::load(u1_mont).mul(Fr::load(u2_mont)).store(u12);
  const uint32_t* us1[1] = {u1_mont};
  const uint32_t* us3[3] = {u2_mont, u1_mont, u12};
  return pc::ec_fold_table_run<C>(be, key, out, q, pts, levels == 2 ? 3u : 1u, levels == 2 ? us3 : us1, w, table.data()) ? 1 : 0;
}
extern "C" int emu_fold_table(int curve, const uint32_t* key, uint32_t n, uint32_t levels, uint32_t w, const uint32_t* u1_mont, const uint32_t* u2_mont,
                              uint32_t* out) {
  switch (curve) {
    case 0: return fold_table<pc_curve_bls12_381>(key, n, levels, w, u1_mont, u2_mont, out);
    case 1: return fold_table<pc_curve_bn254>(key, n, levels, w, u1_mont, u2_mont, out);
    case 2: return fold_table<pc_curve_pallas>(key, n, levels, w, u1_mont, u2_mont, out);
  }
  return -1;
}
extern "C" int emu_wnaf(const uint32_t* k5, int w, int8_t* out200) { return pc::wnaf_digits(k5, w, out200); }
