// Test driver: HyraxPC::commit / open / check of the C++ host mirror (poly_commit_amd/host/hyrax.hpp) on inputs read from
// a file; commitment and proof written to a file -- tests/test_hyrax_gpu.py compares them with the oracle's restatement.
//   file in : u32 curve, u32 n_vars | com_key dim*xy | h xy | evals 2^n Fr | rands dim Fr | point n Fr | r_eval | d dim Fr | r_d | r_b | c
//   file out: row_coms dim*xy | com_eval, com_d, com_b xy | z dim Fr | z_d | z_b | eval
#include <stdio.h>
#include <stdlib.h>
#include "../../poly_commit_amd/host/hyrax.hpp"
using namespace pc_host;

template <class E>
static int run(pc_ctx* ctx, FILE* in, uint32_t n_vars, const char* out_path) {
  typedef FrT<E> Fr;
  auto rd = [&](void* p, size_t b) { if (fread(p, 1, b, in) != b) { printf("short input\n"); exit(2); } };
  auto rd_pt = [&]() { uint64_t xy[2 * E::NQ]; rd(xy, sizeof xy); bool inf = true; for (int i = 0; i < 2 * E::NQ; i++) inf &= xy[i] == 0; return G1Affine<E>::from_xy(xy, inf); };
  auto rd_fr = [&](size_t k) { std::vector<Fr> v(k); if (k) rd(v.data(), k * 32); return v; };
  const size_t dim = (size_t)1 << (n_vars / 2);
  HyraxCommitterKey<E> ck;
  for (size_t i = 0; i < dim; i++) ck.com_key.push_back(rd_pt());
  ck.h = rd_pt();
  std::vector<Fr> evals = rd_fr((size_t)1 << n_vars), rands = rd_fr(dim), point = rd_fr(n_vars);
  Fr r_eval = rd_fr(1)[0];
  std::vector<Fr> d = rd_fr(dim);
  Fr r_d = rd_fr(1)[0], r_b = rd_fr(1)[0], c = rd_fr(1)[0];

  std::vector<G1Affine<E>> row_coms; HyraxCommitmentState<E> state;
  if (Error e = HyraxPC<E>::commit(ctx, ck, evals, rands, row_coms, state)) { printf("commit: kind %d %s\n", (int)e.kind, e.msg.c_str()); return 1; }
  HyraxProof<E> proof; Fr eval;
  if (Error e = HyraxPC<E>::open(ctx, ck, state, point, r_eval, d, r_d, r_b, c, proof, &eval)) { printf("open: kind %d %s\n", (int)e.kind, e.msg.c_str()); return 1; }
  bool ok = false;
  if (Error e = HyraxPC<E>::check(ctx, ck, row_coms, point, proof, c, ok)) { printf("check: kind %d %s\n", (int)e.kind, e.msg.c_str()); return 1; }
  if (!ok) { printf("check rejected an honest proof\n"); return 1; }
  { HyraxProof<E> bad = proof; bad.z_b = bad.z_b + Fr::one();
    if (HyraxPC<E>::check(ctx, ck, row_coms, point, bad, c, ok) || ok) { printf("check accepted an altered z_b (equation 14)\n"); return 1; } }
  { HyraxProof<E> bad = proof; bad.z_d = bad.z_d + Fr::one();
    if (HyraxPC<E>::check(ctx, ck, row_coms, point, bad, c, ok) || ok) { printf("check accepted an altered z_d (equation 13)\n"); return 1; } }
  if (dim > 1) {
    std::vector<G1Affine<E>> fewer(row_coms.begin(), row_coms.end() - 1);
    if (HyraxPC<E>::check(ctx, ck, fewer, point, proof, c, ok).kind != Error::IncorrectCommitmentSize) { printf("short commitment not reported as IncorrectCommitmentSize\n"); return 1; }
  }
  { std::vector<Fr> odd(point.begin(), point.end() - 1);
    if (HyraxPC<E>::check(ctx, ck, row_coms, odd, proof, c, ok).kind != Error::InvalidNumberOfVariables) { printf("odd point not reported as InvalidNumberOfVariables\n"); return 1; }
    std::vector<Fr> half(evals.begin(), evals.begin() + evals.size() / 2);
    std::vector<G1Affine<E>> rc2; HyraxCommitmentState<E> st2;
    if (HyraxPC<E>::commit(ctx, ck, half, rands, rc2, st2).kind != Error::InvalidNumberOfVariables) { printf("odd polynomial not reported as InvalidNumberOfVariables\n"); return 1; } }
  FILE* out = fopen(out_path, "wb");
  auto wr_pt = [&](const G1Affine<E>& p) { uint64_t xy[2 * E::NQ]; p.to_xy(xy); fwrite(xy, 1, sizeof xy, out); };
  for (auto& p : row_coms) wr_pt(p);
  wr_pt(proof.com_eval); wr_pt(proof.com_d); wr_pt(proof.com_b);
  fwrite(proof.z.data(), 32, proof.z.size(), out);
  fwrite(proof.z_d.l, 1, 32, out); fwrite(proof.z_b.l, 1, 32, out); fwrite(eval.l, 1, 32, out);
  fclose(out);
  printf("hyrax commit/open/check OK (dim %zu)\n", dim);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: hyrax_driver in out\n"); return 2; }
  FILE* in = fopen(argv[1], "rb");
  if (!in) { printf("cannot open %s\n", argv[1]); return 2; }
  uint32_t hdr[2]; if (fread(hdr, 4, 2, in) != 2) return 2;
  pc_ctx* ctx = nullptr;
  int rc = pc_hip_init(0, &ctx);
  if (rc != PC_OK) { printf("pc_hip_init failed: %s\n", pc_hip_strerror(rc)); return rc == PC_ERR_NO_DEVICE ? 77 : 1; }
  int r = 1;
  switch (hdr[0]) {
    case 0: r = run<Bls12_381>(ctx, in, hdr[1], argv[2]); break;
    case 1: r = run<Bn254>(ctx, in, hdr[1], argv[2]); break;
    case 2: r = run<Pallas>(ctx, in, hdr[1], argv[2]); break;
  }
  fclose(in);
  pc_hip_shutdown(ctx);
  return r;
}
