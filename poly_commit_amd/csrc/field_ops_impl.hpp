// Instantiation of everything templated on one scalar field; each field_<name>.hip includes this header
// and defines one FieldOps table (pc_internal.hpp).
#pragma once
#include <vector>
#include "pc_internal.hpp"
#include "ntt.hpp"
#include "poly.hpp"
#include "ipa.hpp"
#include "hash.hpp"

namespace pc {

template <class FrP>
struct NttRunnerT : NttRunner {
  NttPlan<FrP> plan;
  NttRunnerT(HipBackend& be, unsigned log_n) : plan(be, log_n) {}
  void run(const uint32_t* in, size_t rows, size_t in_cols, uint32_t* out) override { plan.run(in, rows, in_cols, out); }
};

template <class FrP>
struct FieldOpsImpl {
  typedef Fd<FrP> F;
  static NttRunner* make_ntt(HipBackend& be, unsigned log_n) { return new NttRunnerT<FrP>(be, log_n); }
  static void poly_eval_f(HipBackend& be, const uint32_t* x, size_t n, const uint32_t* z, uint32_t* out_host, uint32_t fan) {
    poly_eval<FrP>(be, x, n, z, out_host, fan);
  }
  static void div_scan_f(HipBackend& be, const uint32_t* x, size_t n, const uint32_t* z, const uint32_t* cin, uint32_t* out, uint32_t fan) {
    div_scan<FrP>(be, x, n, z, cin, out, fan);
  }
  static void witness_f(HipBackend& be, const uint32_t* p, size_t n, const uint32_t* z, uint32_t* q, uint32_t fan) {
    witness_polynomial<FrP>(be, p, n, z, q, fan);
  }
  static void fr_fold(HipBackend& be, uint32_t* lo, const uint32_t* hi, size_t n, const uint32_t* s) {
    FrFoldBody<FrP> b{lo, hi, F::load(s)};
    be.launch(b, n); be.sync();
  }
  static void fr_dot(HipBackend& be, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
    // stage 1: up to 2^17 strided partial products-sums (wide); stage 2: 256 strided sums of those;
    // the host folds the last 256 (~15 us)
    const uint32_t wide = n < (1u << 17) ? (uint32_t)(n ? n : 1) : (1u << 17);
    const uint32_t lanes = wide < 256 ? wide : 256;
    static thread_local std::vector<uint32_t> h;
    h.resize((size_t)lanes * FrP::N);
    const size_t need = ((size_t)wide + lanes) * FrP::N * 4;
    if (need > be.scan_tmp_bytes) {      // reuse the backend's small scratch buffer
      if (be.scan_tmp) { be.sync(); be.free(be.scan_tmp); be.scan_tmp = nullptr; be.scan_tmp_bytes = 0; }
      be.scan_tmp = be.alloc(need); be.scan_tmp_bytes = need;
    }
    uint32_t* part1 = (uint32_t*)be.scan_tmp; uint32_t* part = part1 + (size_t)wide * FrP::N;
    FrDotBody<FrP> body{a, b, (uint32_t)n, wide, part1};
    be.launch(body, wide);
    FrSumBody<FrP> body2{part1, wide, lanes, part};
    be.launch(body2, lanes);
    be.copy_d2h(h.data(), part, h.size() * 4);
    F acc = F::zero();
    for (uint32_t t = 0; t < lanes; t++) acc = acc.add(F::load(&h[(size_t)t * FrP::N]));
    acc.store(out);
  }
  static void ipa_fold_dots(HipBackend& be, uint32_t* c, uint32_t* z, size_t m, const uint32_t* u, const uint32_t* ui, uint32_t* out_host) {
    const uint32_t q = (uint32_t)(m >> 1);
    uint32_t blocks = (q + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks == 0) blocks = 1;
    const size_t need = ((size_t)blocks + 1) * 2 * FrP::N * 4;
    if (need > be.scan_tmp_bytes) {      // reuse the backend's small scratch buffer (as fr_dot does)
      if (be.scan_tmp) { be.sync(); be.free(be.scan_tmp); be.scan_tmp = nullptr; be.scan_tmp_bytes = 0; }
      be.scan_tmp = be.alloc(need); be.scan_tmp_bytes = need;
    }
    uint32_t* partial = (uint32_t*)be.scan_tmp; uint32_t* fin = partial + (size_t)blocks * 2 * FrP::N;
    const F fu = u ? F::load(u) : F::zero(), fui = ui ? F::load(ui) : F::zero();
    hipLaunchKernelGGL(k_ipa_fold_dots<FrP>, dim3(blocks), dim3(256), 0, be.stream, c, z, (uint32_t)m, u ? 1u : 0u, fu, fui, partial);
    PC_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_ipa_dots_final<FrP>, dim3(1), dim3(256), 0, be.stream, (const uint32_t*)partial, blocks, fin);
    PC_HIP_CHECK(hipGetLastError());
    be.copy_d2h(out_host, fin, (size_t)2 * FrP::N * 4);
  }
  static void fr_powers(HipBackend& be, const uint32_t* z, size_t n, uint32_t* out) {
    FrPowersBody<FrP> body; body.out = out;
    F w = F::load(z);
    for (int k = 0; k < 32; k++) { w.store(body.pt.w[k]); w = w.sqr(); }
    be.launch(body, n); be.sync();
  }
  static void ipa_key_scalars(HipBackend& be, const uint32_t* c, size_t m, uint32_t* s, size_t n0, const uint32_t* fold_u, size_t fold_m,
                              uint32_t* out_l, uint32_t* out_r) {
    if (fold_u) { IpaKeyScalarUpdateBody<FrP> b{s, (uint32_t)fold_m, F::load(fold_u)}; be.launch(b, n0); }
    if (out_l) { IpaFixedKeyScalarsBody<FrP> b{c, s, (uint32_t)m, out_l, out_r}; be.launch(b, n0); }
    be.sync();
  }
  static void fr_lincomb(HipBackend& be, const void* addr, const void* lens, const void* xi, size_t k, void* out, size_t n_out) {
    FrLinCombBody<FrP> b{(const uint64_t*)addr, (const uint32_t*)lens, (const uint32_t*)xi, (uint32_t)k, (uint32_t*)out};
    be.launch(b, n_out, 256);
  }
  static void column_hash(HipBackend& be, int hash, const uint32_t* e, uint32_t rows, uint32_t n_cols, uint32_t* o) {
    if (hash == PC_HASH_SHA256) { ColumnHashBody<FrP, Sha256> b{e, rows, n_cols, o}; be.launch(b, n_cols, 64); }
    else { ColumnHashBody<FrP, Blake2s256> b{e, rows, n_cols, o}; be.launch(b, n_cols, 64); }
  }
  static void column_hash_part(HipBackend& be, int hash, const uint32_t* e, uint32_t rows, uint32_t n_cols, uint32_t rows_total, uint32_t col0,
                               uint32_t cols, int first, int last, uint32_t* state, uint32_t* o) {
    if (hash == PC_HASH_SHA256) { ColumnHashPartBody<FrP, Sha256> b{e, rows, n_cols, rows_total, col0, (uint32_t)first, (uint32_t)last, state, o}; be.launch(b, cols, 64); }
    else { ColumnHashPartBody<FrP, Blake2s256> b{e, rows, n_cols, rows_total, col0, (uint32_t)first, (uint32_t)last, state, o}; be.launch(b, cols, 64); }
  }
  static FieldOps table() {
    return FieldOps{&make_ntt, &poly_eval_f, &div_scan_f, &witness_f, &fr_fold, &fr_dot, &ipa_fold_dots, &fr_powers, &ipa_key_scalars, &fr_lincomb, &column_hash,
                    &column_hash_part};
  }
};

}  // namespace pc
