// Instantiation of everything templated on one curve; each curve_<name>.hip includes this header and
// defines one CurveOps table (pc_internal.hpp).
#pragma once
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "pc_internal.hpp"
#include "hip_backend_msm.hpp"
#include "ipa.hpp"
#include "glv.hpp"
#include "fold_table.hpp"
#include "serialize.hpp"

namespace pc {

// One MSM pipeline.  A call that repeats an earlier one exactly (same resident bases, same device scalar buffer, same length -- the
// late rounds of an IPA opening, a prover that keeps its buffers) replays the ~25 launches of the MSM from a captured hipGraph: the
// first occurrence runs normally (it also grows every lazily sized workspace), the second is captured, later ones are one
// hipGraphLaunch.  On for calls of at most 2^18 pairs (PC_HIP_GRAPHS=0: never, =1: every size): there the host-side cost of queueing
// the launches IS the latency -- a kernel trace of the fixed-key rounds of a Pallas opening showed the second MSM of a round starting
// 160 us after the first (the time the host takes to queue 14 launches) in a round of 970 us; with replays the 16 rounds take 13.8
// instead of 16.2 ms (open 62.0 against 65.5 ms at 2^22).  Large calls gain nothing (2^20 commit+open 5.79-5.82 vs 5.81-5.84 ms:
// their launches are issued behind running kernels) and stay on plain launches.
template <class C>
struct MsmRunnerT : MsmRunner {
  HipBackend& be;
  MsmPlan<C, HipBackend> plan;
  struct CallKey {
    const uint32_t* bases; uint32_t base_off; const uint32_t* scalars; size_t n; bool from_mont;
    bool operator==(const CallKey& o) const { return bases == o.bases && base_off == o.base_off && scalars == o.scalars && n == o.n && from_mont == o.from_mont; }
  };
  // epoch: the backend's free_epoch when the graph was captured; plain_epoch: after the key's last plain run (which sized every scratch
  // buffer for it) -- a capture is only attempted while nothing was freed since, i.e. while it cannot meet an allocation
  struct GraphSlot { CallKey key; int seen = 0; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; uint64_t stamp = 0; uint64_t epoch = 0, plain_epoch = ~0ull; };
  static constexpr int GRAPH_SLOTS = 4;
  GraphSlot slots[GRAPH_SLOTS];
  uint64_t clock = 0;
  bool graphs_on;
  size_t graph_max_n;
  hipEvent_t join_ev = nullptr;
  MsmRunnerT(HipBackend& b, size_t n, const MsmConfig& cfg, uint32_t subs = 0) : be(b), plan(b, n, cfg, subs) {
    const char* e = getenv("PC_HIP_GRAPHS");
    graphs_on = subs == 0 && !(e && !strcmp(e, "0"));
    graph_max_n = (e && !strcmp(e, "1")) ? (size_t)-1 : (size_t)1 << 18;
  }
  ~MsmRunnerT() override {
    for (auto& s : slots) drop(s);
    if (join_ev) (void)hipEventDestroy(join_ev);
  }
  static void drop(GraphSlot& s) {
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (s.graph) (void)hipGraphDestroy(s.graph);
    s.exec = nullptr; s.graph = nullptr; s.seen = 0;
  }
  GraphSlot* slot_for(const CallKey& k) {
    GraphSlot* lru = &slots[0];
    for (auto& s : slots) {
      if (s.seen && s.key == k) {
        if (s.exec && s.epoch != be.free_epoch) { drop(s); s.key = k; }      // a buffer the graph may point into was freed since: start over
        return &s;
      }
      if (s.stamp < lru->stamp) lru = &s;
    }
    drop(*lru); lru->key = k;
    return lru;
  }
  // capture the launch sequence of plan.enqueue() on the pipeline's queues into slot s; false: capture unavailable
  bool capture(GraphSlot& s, const uint32_t* bases, uint32_t base_off, const uint32_t* sdev, size_t n, bool from_mont) {
    hipStream_t origin = be.stream;
    if (hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ok = true;
    try {
      plan.enqueue(bases, base_off, sdev, n, from_mont);
      if (be.tail_stream) {     // the reductions forked to the tail queue: join them back before the capture ends
        if (!join_ev) PC_HIP_CHECK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
        PC_HIP_CHECK(hipEventRecord(join_ev, be.tail_stream));
        PC_HIP_CHECK(hipStreamWaitEvent(origin, join_ev, 0));
      }
    } catch (...) { ok = false; }
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(origin, &g) != hipSuccess || !g) { (void)hipGetLastError(); ok = false; }
    if (ok && hipGraphInstantiate(&s.exec, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); s.exec = nullptr; ok = false; }
    if (!ok) {
      if (g) (void)hipGraphDestroy(g);
      be.stream = origin;
      try { be.replace_capturing_streams(); } catch (...) {}      // a queue left in capture mode would refuse the plain launches that follow
      return false;
    }
    s.graph = g; s.epoch = be.free_epoch;
    return true;
  }
  void enqueue(const uint32_t* bases, uint32_t base_off, const void* scalars, pc_mem where, size_t n, bool from_mont) override {
    const uint32_t* sdev = (const uint32_t*)scalars;
    be.n_ev = 0; be.mark();
    if (where == PC_MEM_HOST && n) {
      be.copy_h2d(plan.scalar_staging(), scalars, n * (size_t)C::FrP::N * 4);
      sdev = plan.scalar_staging();
    }
    if (graphs_on && !be.timing && where == PC_MEM_DEVICE && n >= 32 && n <= graph_max_n) {      // (host scalars through the staging buffer as well: measured, 1.12 vs 1.15 ms for a blocking commit+open at 2^10 -- a blocking call's launches are queued beside its own kernels -- not taken)
      GraphSlot* s = slot_for(CallKey{bases, base_off, sdev, n, from_mont});
      s->stamp = ++clock;
      if (s->seen >= 1 && !s->exec && s->plain_epoch == be.free_epoch && !capture(*s, bases, base_off, sdev, n, from_mont)) graphs_on = false;   // this runner stays on plain launches
      if (s->exec) {
        plan.prepare_replay(n);
        PC_HIP_CHECK(hipGraphLaunch(s->exec, be.stream));
        be.record_done();
        s->seen++;
        return;
      }
      s->seen++;
      plan.enqueue(bases, base_off, sdev, n, from_mont);
      s->plain_epoch = be.free_epoch;      // whatever this run had to grow is grown: the next occurrence may be captured
      return;
    }
    plan.enqueue(bases, base_off, sdev, n, from_mont);
  }
  void begin_parts(size_t n_total) override { be.n_ev = 0; be.mark(); plan.begin_parts(n_total); }
  void add_part(const uint32_t* bases, uint32_t base_off, size_t first, const void* scalars_part, pc_mem where, size_t n, bool from_mont, bool last) override {
    const uint32_t* sdev = (const uint32_t*)scalars_part;
    int tok = -1;
    if (where == PC_MEM_HOST && n) {      // the copy rides on the auxiliary queue (in order before this part's sort), beside the previous part's accumulation
      uint32_t* dst = plan.scalar_staging() + first * (size_t)C::FrP::N;
      {
        struct AuxScope { HipBackend& b; int* tok; ~AuxScope() { try { *tok = b.aux_end(); } catch (...) { *tok = -1; } } };
        be.aux_begin(-1, -1);
        AuxScope scope{be, &tok};
        be.copy_h2d(dst, scalars_part, n * (size_t)C::FrP::N * 4);
      }
      sdev = dst;
    }
    plan.add_part(bases, base_off + (uint32_t)first, sdev, n, from_mont, tok, last);
  }
  void enqueue_vectors(const uint32_t* bases, uint32_t base_off, const uint64_t* ptrs_host, size_t count, size_t m, bool from_mont) override {
    be.n_ev = 0; be.mark();
    plan.enqueue_vectors(bases, base_off, ptrs_host, count, m, from_mont);
  }
  void finish(uint32_t* out_host) override { plan.finish(out_host); }
  void shape(uint32_t out[4]) const override {
    const MsmGeom& g = plan.last_geom();
    out[0] = g.c; out[1] = g.Wd; out[2] = g.NB; out[3] = g.tbl_stride ? 1u : 0u;
  }
  void trim() override { plan.trim(); }
};

template <class C>
void build_window_table(HipBackend& be, const uint32_t* bases, uint32_t n, uint32_t c, uint32_t Wd, uint32_t* table, uint32_t stride) {
  // PC_HIP_TABLE_BUILD=serial: one lane per base walking its chain and inverting at every window (the round-1 build)
  static const bool serial = []() { const char* e = getenv("PC_HIP_TABLE_BUILD"); return e && !strcmp(e, "serial"); }();
  if (serial) { WindowTableBody<C> b{bases, n, c, Wd, table, stride}; be.launch(b, n, 64); be.sync(); }
  else if ((size_t)(Wd > 1 ? Wd - 1 : 0) * n <= TABLE_ONESHOT_MAX_POINTS) build_window_table_oneshot<C>(be, bases, n, c, Wd, table, stride);
  else build_window_table_batched<C>(be, bases, n, c, Wd, table, stride);
}

// key[i] = affine(key[i] + u * key[half + i]): GLV split of the shared challenge on the host, one ladder per lane
template <class C>
void ec_fold_run(HipBackend& be, uint32_t* key, size_t half, const uint32_t* u_mont) {
  typedef typename GlvOf<C>::T G;
  Fd<typename C::FrP> u = Fd<typename C::FrP>::load(u_mont).from_mont();
  uint64_t k[4]; memcpy(k, u.l, 32);
  GlvSplit sp = glv_decompose<G>(k);
  EcFoldGlvBody<C> body; body.key = key; body.half = (uint32_t)half;
  body.n1.from_scalar(sp.k1); body.n2.from_scalar(sp.k2); body.neg1 = sp.neg1; body.neg2 = sp.neg2;
  for (int i = 0; i < C::FqP::N; i++) body.beta[i] = G::BETA_MONT[i];
  constexpr int FN = C::FqP::N;
  if (half >= 4096) {
    // ladders leave Jacobian results; one inversion per K of them (normalize_batch, ipa_pc/mod.rs:706-708)
    uint32_t* ws = (uint32_t*)be.workspace(half * (size_t)4 * FN * 4);
    body.jac_out = ws;
    be.launch(body, half, 64);
    const uint32_t K = half >= ((size_t)1 << 20) ? 16 : half >= ((size_t)1 << 17) ? 8 : 4;     // >= 2^14 lanes while it matters
    JacBatchAffineBody<C> nb{ws, ws + half * (size_t)3 * FN, key, (uint32_t)half, K};
    be.launch(nb, (half + K - 1) / K, 64);
  } else be.launch(body, half, 64);
  be.sync();
}

// out[i] = affine(in[i] + u * in[half + i]), i < half (out may be `in` itself: the in-place fold).  table != null: the first
// fold of an opening from the committer key's (one-level) fold table of width-w NAF digits, else the GLV ladder per element.
template <class C>
void ec_fold_to_run(HipBackend& be, const uint32_t* in, uint32_t* out, size_t half, const uint32_t* u_mont, const uint32_t* table, uint32_t w) {
  typedef typename GlvOf<C>::T G;
  constexpr int FN = C::FqP::N;
  if (table) {
    const uint32_t* us[1] = {u_mont};
    if (ec_fold_table_run<C>(be, in, out, half, half, 1, us, w, table)) return;
    // a split that does not fit the table's rows (never seen: |k1|, |k2| <= 2^128): the ladder below
  }
  Fd<typename C::FrP> u = Fd<typename C::FrP>::load(u_mont).from_mont();
  uint64_t k[4]; memcpy(k, u.l, 32);
  GlvSplit sp = glv_decompose<G>(k);
  NafMasks<5> n1, n2; n1.from_scalar(sp.k1); n2.from_scalar(sp.k2);
  EcFoldGlvBody<C> body; body.key = out; body.key_in = in == out ? nullptr : in; body.half = (uint32_t)half;
  body.n1 = n1; body.n2 = n2; body.neg1 = sp.neg1; body.neg2 = sp.neg2;
  for (int i = 0; i < FN; i++) body.beta[i] = G::BETA_MONT[i];
  if (half >= 4096) {
    uint32_t* ws = (uint32_t*)be.workspace(half * (size_t)4 * FN * 4);
    body.jac_out = ws;
    be.launch(body, half, 64);
    const uint32_t K = half >= ((size_t)1 << 20) ? 16 : half >= ((size_t)1 << 17) ? 8 : 4;
    JacBatchAffineBody<C> nb{ws, ws + half * (size_t)3 * FN, out, (uint32_t)half, K};
    be.launch(nb, (half + K - 1) / K, 64);
  } else be.launch(body, half, 64);
  be.sync();
}

template <class C>
struct CurveOpsImpl {
  static constexpr int AW = 2 * C::FqP::N;
  static MsmRunner* make_runner(HipBackend& be, size_t n_max, const MsmConfig& cfg, uint32_t subs) {
    return new MsmRunnerT<C>(be, n_max, cfg, subs);
  }
  static void window_table(HipBackend& be, const uint32_t* bases, uint32_t n, uint32_t c, uint32_t Wd, uint32_t* table, uint32_t stride) {
    build_window_table<C>(be, bases, n, c, Wd, table, stride);
  }
  static void ec_fold(HipBackend& be, uint32_t* key, size_t half, const uint32_t* u_mont) {
    ec_fold_run<C>(be, key, half, u_mont);
  }
  static void ec_fold_to(HipBackend& be, const uint32_t* in, uint32_t* out, size_t half, const uint32_t* u_mont, const uint32_t* table, uint32_t w) {
    ec_fold_to_run<C>(be, in, out, half, u_mont, table, w);
  }
  static bool ec_fold_table(HipBackend& be, const uint32_t* key_lo, uint32_t* out, size_t count, size_t row_pts, uint32_t terms,
                            const uint32_t* const* u_monts, uint32_t w, const uint32_t* table) {
    return ec_fold_table_run<C>(be, key_lo, out, count, row_pts, terms, u_monts, w, table);
  }
  static void fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    typedef Fd<typename C::FrP> Fr;
    Fr::load(a).mul(Fr::load(b)).store(out);
  }
  static void fr_inv(const uint32_t* a, uint32_t* out) { Fd<typename C::FrP>::load(a).inv().store(out); }
  static void fr_one(uint32_t* out) { Fd<typename C::FrP>::one().store(out); }
  static void fold_table_build(HipBackend& be, const uint32_t* pts, size_t count, uint32_t w, uint32_t* table) { fold_table_build_run<C>(be, pts, count, w, table); }
  static void fixed_base(HipBackend& be, const uint32_t* g, const uint32_t* scalars, size_t n, uint32_t* out) {
    if (n < 4096) {          // a handful of scalars: the per-lane ladder, no table
      FixedBaseMulBody<C> body; body.scalars = scalars; body.out = out;
      for (int i = 0; i < AW; i++) body.g[i] = g[i];
      be.launch(body, n, 64); be.sync();
      return;
    }
    // window table of the fixed base on the host: T[w][d-1] = d * 2^(8 w) * g, d = 1..128 (one inversion for all of it)
    typedef host64::Xyzz64<C> P64;
    constexpr int FW = C::FqP::N, XW = XyzzD<C>::WORDS;
    const uint32_t Wd = msm_num_windows(C::FrP::BITS, FIXED_BASE_C), half = 1u << (FIXED_BASE_C - 1);
    std::vector<uint32_t> xyzz((size_t)Wd * half * XW), tbl((size_t)Wd * half * AW);
    P64 base = P64::infinity();
    bool inf = true; for (int i = 0; i < AW; i++) inf &= g[i] == 0;
    if (!inf) { base.X = P64::Fq::load(g); base.Y = P64::Fq::load(g + FW); base.ZZ = P64::Fq::one(); base.ZZZ = P64::Fq::one(); }
    for (uint32_t w = 0; w < Wd; w++) {
      P64 cur = base;
      for (uint32_t d = 0; d < half; d++) {
        cur.X.store(&xyzz[((size_t)w * half + d) * XW]); cur.Y.store(&xyzz[((size_t)w * half + d) * XW + FW]);
        cur.ZZ.store(&xyzz[((size_t)w * half + d) * XW + 2 * FW]); cur.ZZZ.store(&xyzz[((size_t)w * half + d) * XW + 3 * FW]);
        cur.add(base);
      }
      for (uint32_t k = 0; k < FIXED_BASE_C; k++) base = base.dbl();
    }
    host64::batch_to_affine<C>(xyzz.data(), (size_t)Wd * half, tbl.data());
    // device: table | XYZZ results | prefix products
    const size_t tb = tbl.size() * 4, rb = n * (size_t)XW * 4, sb = n * (size_t)FW * 4;
    uint8_t* ws = (uint8_t*)be.workspace(tb + rb + sb);
    uint32_t* dtbl = (uint32_t*)ws; uint32_t* dres = (uint32_t*)(ws + tb); uint32_t* dscr = (uint32_t*)(ws + tb + rb);
    be.copy_h2d(dtbl, tbl.data(), tb);
    be.sync();                                   // the host vectors go out of scope below
    FixedBaseTableMulBody<C> body{scalars, dtbl, Wd, dres};
    be.launch(body, n, 64);
    const uint32_t K = 16;
    XyzzBatchAffineBody<C> nb{dres, dscr, out, (uint32_t)n, K};
    be.launch(nb, (n + K - 1) / K, 64);
    be.sync();
  }
  static uint32_t srs_decode(HipBackend& be, const uint8_t* bytes_dev, size_t n, int compressed, uint32_t* out) {
    uint32_t* bad = (uint32_t*)be.workspace(4);
    be.memset(bad, 0, 4);
    SrsDecodeBody<C> b{bytes_dev, (uint32_t)n, compressed ? 1u : 0u, C::FqP::BITS == 381 ? 1u : 0u, out, bad};
    be.launch(b, n, 64);
    uint32_t h = 0; be.copy_d2h(&h, bad, 4);
    return h;
  }
  static void srs_encode(HipBackend& be, const uint32_t* pts_dev, size_t n, int compressed, uint8_t* out_dev) {
    SrsEncodeBody<C> b{pts_dev, (uint32_t)n, compressed ? 1u : 0u, C::FqP::BITS == 381 ? 1u : 0u, out_dev};
    be.launch(b, n, 64);
  }
  static void points_sum(const uint32_t* pts, size_t count, uint32_t* out) {
    XyzzD<C> acc = XyzzD<C>::infinity();
    for (size_t i = 0; i < count; i++) acc.add_affine(AffD<C>::load(pts + i * AW));
    acc.to_affine().store(out);
  }
  static void point_mul(const uint32_t* pt, const uint32_t* k_mont, uint32_t* out) {
    typedef host64::Xyzz64<C> P64;
    constexpr int FW = C::FqP::N;
    Fd<typename C::FrP> k = Fd<typename C::FrP>::load(k_mont).from_mont();
    bool inf = true; for (int i = 0; i < 2 * FW; i++) inf &= pt[i] == 0;
    P64 base = P64::infinity();
    if (!inf) { base.X = P64::Fq::load(pt); base.Y = P64::Fq::load(pt + FW); base.ZZ = P64::Fq::one(); base.ZZZ = P64::Fq::one(); }
    P64 acc = P64::infinity();
    for (int bit = C::FrP::N * 32 - 1; bit >= 0; bit--) {
      acc = acc.dbl();
      if ((k.l[bit >> 5] >> (bit & 31)) & 1) acc.add(base);
    }
    acc.store_affine(out);
  }
  static CurveOps table() {
    return CurveOps{AW, (uint32_t)C::FrP::BITS, &make_runner, &window_table, &ec_fold, &ec_fold_to, &ec_fold_table, &fold_table_build, (uint32_t)FOLD_ROWS, &fixed_base, &srs_decode, &srs_encode, &points_sum, &point_mul, &fr_mul, &fr_inv, &fr_one};
  }
};

}  // namespace pc
