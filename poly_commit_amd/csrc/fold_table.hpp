// The fold table of an IPA committer key and the folds that read it (glv.hpp: AffineDoubleRowBody, AffineAddRowBody, EcFoldTableBody):
// host-side drivers, generic over the backend (HipBackend in the library, the CPU stepping backend of tests/emu).
//   InnerProductArgPC::open, `k_l += k_r * round_challenge` + normalize_batch (poly-commit/src/ipa_pc/mod.rs:699-707), rounds 1 and 2
#pragma once
#include <algorithm>
#include <vector>
#include "glv.hpp"
#include "ipa.hpp"

namespace pc {

// out[i] = affine(key_lo[i] + sum_t u_t * P_t[i]), i < count, from the key's fold table (glv.hpp: EcFoldTableBody): term t covers table
// points [t * count, (t + 1) * count) of rows of row_pts points, built for width-w NAF digits.  false: a digit beyond the table's rows
// or more ops than the kernel's list holds (the caller falls back to the ladder).
template <class C, class Backend>
bool ec_fold_table_run(Backend& be, const uint32_t* key_lo, uint32_t* out, size_t count, size_t row_pts, uint32_t terms,
                       const uint32_t* const* u_monts, uint32_t w, const uint32_t* table) {
  typedef typename GlvOf<C>::T G;
  constexpr int FN = C::FqP::N;
  EcFoldTableBody<C> body; body.key_lo = key_lo; body.table = table; body.count = (uint32_t)count; body.row_pts = (uint32_t)row_pts; body.n_ops = 0;
  for (int i = 0; i < FN; i++) body.beta[i] = G::BETA_MONT[i];
  // ops ordered by row: consecutive additions of a lane then walk the table row after row (every row is one coalesced stream)
  struct Op { uint32_t row; uint16_t code; };
  std::vector<Op> ops;
  for (uint32_t t = 0; t < terms; t++) {
    Fd<typename C::FrP> u = Fd<typename C::FrP>::load(u_monts[t]).from_mont();
    uint64_t k[4]; memcpy(k, u.l, 32);
    const GlvSplit sp = glv_decompose<G>(k);
    for (int which = 0; which < 2; which++) {
      int8_t dg[200];
      const int len = wnaf_digits(which ? sp.k2 : sp.k1, (int)w, dg);
      const uint32_t sgn = which ? sp.neg2 : sp.neg1;
      for (int bit = 0; bit < len; bit++) {
        if (!dg[bit]) continue;
        if ((uint32_t)bit >= FOLD_ROWS) return false;
        const uint32_t mag = (uint32_t)(dg[bit] < 0 ? -dg[bit] : dg[bit]);          // odd, < 2^(w-1)
        const uint32_t row = (mag >> 1) * FOLD_ROWS + (uint32_t)bit;
        const bool negate = (dg[bit] < 0) != (sgn != 0);
        if (row > 0x7ffu) return false;
        ops.push_back(Op{row, (uint16_t)(row | (t << 11) | (which ? 0x4000u : 0u) | (negate ? 0x8000u : 0u))});
      }
    }
  }
  if (ops.size() > EcFoldTableBody<C>::MAX_OPS) return false;
  std::stable_sort(ops.begin(), ops.end(), [](const Op& a, const Op& b) { return a.row < b.row; });
  for (const Op& o : ops) body.ops[body.n_ops++] = o.code;
  // XYZZ sums | prefix products, then one inversion per K points into `out`
  uint32_t* ws = (uint32_t*)be.workspace(count * (size_t)5 * FN * 4);
  body.out_xyzz = ws;
  be.launch(body, count, 64);
  const uint32_t K = count >= ((size_t)1 << 20) ? 16 : count >= ((size_t)1 << 17) ? 8 : 4;
  XyzzBatchAffineBody<C> nb{ws, ws + count * (size_t)4 * FN, out, (uint32_t)count, K};
  be.launch(nb, (count + K - 1) / K, 64);
  be.sync();
  return true;
}

// T[(d >> 1) * FOLD_ROWS + b][j] = d * 2^b * pts[j] for the odd d < 2^(w-1) (glv.hpp): row (1, 0) is a copy, rows (d, 0) are batched
// affine additions of 2 P, every further row one batched affine doubling of the row before it
template <class C, class Backend>
void fold_table_build_run(Backend& be, const uint32_t* pts, size_t count, uint32_t w, uint32_t* table) {
  constexpr int FN = C::FqP::N, AW = 2 * FN;
  const uint32_t D = 1u << (w - 2);
  const size_t row = count * (size_t)AW;
  be.copy_d2d(table, pts, row * 4);
  uint32_t* scratch = (uint32_t*)be.workspace(count * (size_t)FN * 4);
  const uint32_t K = 64;
  const uint32_t blocks = (uint32_t)((count + K - 1) / K);
  auto dbl = [&](uint32_t d_idx, uint32_t b) {
    uint32_t* base = table + (size_t)d_idx * FOLD_ROWS * row;
    AffineDoubleRowBody<C> body{base + (size_t)(b - 1) * row, base + (size_t)b * row, scratch, (uint32_t)count, K};
    be.launch(body, blocks, 64);
  };
  if (D > 1) dbl(0, 1);                                        // 2 P = row (1, 1)
  for (uint32_t di = 1; di < D; di++) {                        // (2 di + 1) P = (2 di - 1) P + 2 P
    AffineAddRowBody<C> body{table + (size_t)(di - 1) * FOLD_ROWS * row, table + row, table + (size_t)di * FOLD_ROWS * row, scratch, (uint32_t)count, K};
    be.launch(body, blocks, 64);
  }
  for (uint32_t di = 0; di < D; di++)
    for (uint32_t b = (di == 0 && D > 1) ? 2 : 1; b < FOLD_ROWS; b++) dbl(di, b);
  be.sync();
}

}  // namespace pc
