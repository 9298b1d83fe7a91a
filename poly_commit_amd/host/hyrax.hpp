// C++ host mirror of HyraxPC's commit / open / check (poly-commit/src/hyrax/mod.rs), above the C ABI:
//
//   commit   hyrax/mod.rs:214-252   dim = 2^(n/2) Pedersen commitments, one per row of the dim x dim evaluation matrix
//                                   (flat_to_matrix_column_major, utils.rs:13-21), each + h * r_i           -> pc_hip_msm_many
//   open     hyrax/mod.rs:287-402   lt = t.row_mul(l), eval = <lt, r>, com_eval, the dot-product argument (d, com_d, com_b),
//                                   z = d + c lt, z_d, z_b                                                  -> pc_hip_fr_lincomb, pc_hip_msm
//   check    hyrax/mod.rs:418-511   equation (14) on the host, t' = MSM(row_coms, l) and com(z, z_d) on the device (13)
//
// Every row commitment is MSM(com_key[..dim] || h, row_i || r_i): the hiding term rides in the same many-MSM pass as one more
// base.  The sponge is the caller's (a generic CryptographicSponge absorbs key, row commitments, point and the three
// commitments, :335-341 / :377-385): its challenge c is an argument, and so are the prover's random field elements, in the
// order the reference draws them (rows' r_i inside the row loop; then r_eval :352, d :361-362, r_d :367, r_b :371).
#pragma once
#include "kzg10.hpp"

namespace pc_host {

template <class E>
struct HyraxCommitterKey {                          // hyrax/data_structures.rs: HyraxCommitterKey { com_key, h } (= verifier key)
  std::vector<G1Affine<E>> com_key;
  G1Affine<E> h = G1Affine<E>::zero();
};

template <class E>
struct HyraxCommitmentState {                       // HyraxCommitmentState { randomness, mat }
  std::vector<FrT<E>> randomness;                   // dim
  std::vector<FrT<E>> mat;                          // dim x dim, row-major
  size_t dim = 0;
};

template <class E>
struct HyraxProof {                                 // HyraxProof { com_eval, com_d, com_b, z, z_d, z_b }
  G1Affine<E> com_eval = G1Affine<E>::zero(), com_d = G1Affine<E>::zero(), com_b = G1Affine<E>::zero();
  std::vector<FrT<E>> z;
  FrT<E> z_d = FrT<E>::zero(), z_b = FrT<E>::zero();
};

template <class E>
struct HyraxPC {
  typedef FrT<E> Fr;
  static Error backend_error(pc_ctx* ctx, int rc) {
    Error e; e.kind = Error::Backend; e.msg = std::string(pc_hip_strerror(rc)) + ": " + pc_hip_last_error(ctx); return e;
  }
  static Error invalid_vars(size_t n) { Error e; e.kind = Error::InvalidNumberOfVariables; e.a = n; return e; }
  static G1Affine<E> from_out(const uint64_t* xy) {
    bool inf = true; for (int i = 0; i < 2 * E::NQ; i++) inf &= xy[i] == 0;
    return G1Affine<E>::from_xy(xy, inf);
  }
  // tensor_prime (hyrax/utils.rs:27-39): all evaluations of eq(i, values), first variable in the top bit
  static std::vector<Fr> tensor_prime(const Fr* values, size_t k) {
    std::vector<Fr> out(1, Fr::one());
    for (size_t v = k; v-- > 0;) {
      std::vector<Fr> nx(out.size() * 2);
      const Fr one_minus = Fr::one() - values[v];
      for (size_t i = 0; i < out.size(); i++) { nx[i] = out[i] * one_minus; nx[out.size() + i] = out[i] * values[v]; }
      out.swap(nx);
    }
    return out;
  }
  static void tensors(const std::vector<Fr>& point, std::vector<Fr>& l, std::vector<Fr>& r) {
    const size_t n = point.size();
    std::vector<Fr> rev(point.rbegin(), point.rend());                       // :297
    l = tensor_prime(rev.data() + n / 2, n - n / 2);                         // point_lower
    r = tensor_prime(rev.data(), n / 2);                                     // point_upper
  }
  static Fr inner_product(const std::vector<Fr>& a, const std::vector<Fr>& b) {
    Fr s = Fr::zero(); for (size_t i = 0; i < a.size() && i < b.size(); i++) s = s + a[i] * b[i]; return s;
  }
  // com_key[..dim] || h resident on the device
  static int upload_ext(pc_ctx* ctx, const HyraxCommitterKey<E>& ck, size_t dim, pc_srs** srs) {
    std::vector<G1Affine<E>> ext(ck.com_key.begin(), ck.com_key.begin() + dim);
    ext.push_back(ck.h);
    return pc_hip_srs_upload(ctx, E::ID, ext.data(), ext.size(), sizeof(G1Affine<E>), PC_MEM_HOST, srs);
  }
  // MSM(com_key[..dim], v) + h * r through the extended key
  static int pedersen_hiding(pc_ctx* ctx, pc_srs* ext, const std::vector<Fr>& v, const Fr& r, G1Affine<E>& out) {
    std::vector<Fr> sc(v); sc.push_back(r);
    uint64_t xy[2 * E::NQ]; int inf = 0;
    int rc = pc_hip_msm(ctx, ext, 0, sc.data(), PC_SCALARS_MONTGOMERY, PC_MEM_HOST, sc.size(), xy, &inf);
    if (rc == PC_OK) out = from_out(xy);
    return rc;
  }

  // commit for one polynomial: evals = poly.to_evaluations() (2^n values), rands = the dim row randomisers.
  static Error commit(pc_ctx* ctx, const HyraxCommitterKey<E>& ck, const std::vector<Fr>& evals, const std::vector<Fr>& rands,
                      std::vector<G1Affine<E>>& row_coms, HyraxCommitmentState<E>& state) {
    size_t n = 0; while (((size_t)1 << n) < evals.size()) n++;
    if (((size_t)1 << n) != evals.size() || n % 2 == 1 || n > ck.com_key.size()) return invalid_vars(n);      // :220-228
    const size_t dim = (size_t)1 << (n / 2);
    if (dim > ck.com_key.size() || rands.size() != dim) return invalid_vars(n);
    state.dim = dim; state.randomness = rands; state.mat.resize(dim * dim);
    std::vector<Fr> ext_rows(dim * (dim + 1));
    for (size_t row = 0; row < dim; row++) {
      for (size_t col = 0; col < dim; col++) { state.mat[row * dim + col] = evals[col * dim + row]; ext_rows[row * (dim + 1) + col] = evals[col * dim + row]; }
      ext_rows[row * (dim + 1) + dim] = rands[row];
    }
    pc_srs* ext = nullptr;
    int rc = upload_ext(ctx, ck, dim, &ext);
    std::vector<uint64_t> out(dim * 2 * E::NQ);
    if (rc == PC_OK) rc = pc_hip_msm_many(ctx, ext, 0, ext_rows.data(), PC_SCALARS_MONTGOMERY, PC_MEM_HOST, dim + 1, dim, out.data(), nullptr);
    pc_hip_srs_free(ext);
    if (rc != PC_OK) return backend_error(ctx, rc);
    row_coms.clear();
    for (size_t row = 0; row < dim; row++) row_coms.push_back(from_out(&out[row * 2 * E::NQ]));
    return Error();
  }

  static Error open(pc_ctx* ctx, const HyraxCommitterKey<E>& ck, const HyraxCommitmentState<E>& state, const std::vector<Fr>& point,
                    const Fr& r_eval, const std::vector<Fr>& d, const Fr& r_d, const Fr& r_b, const Fr& c, HyraxProof<E>& proof, Fr* eval_out = nullptr) {
    const size_t n = point.size(), dim = state.dim;
    if (n % 2 == 1 || ((size_t)1 << (n / 2)) != dim || d.size() != dim) return invalid_vars(n);
    std::vector<Fr> l, r; tensors(point, l, r);
    // lt = t.row_mul(&l): sum_i l_i * row_i, on the device                                  :341
    std::vector<const void*> rows(dim); std::vector<size_t> lens(dim, dim);
    for (size_t i = 0; i < dim; i++) rows[i] = &state.mat[i * dim];
    std::vector<Fr> lt(dim);
    int rc = pc_hip_fr_lincomb(ctx, E::ID, rows.data(), PC_MEM_HOST, lens.data(), dim, l.data(), lt.data(), PC_MEM_HOST, dim);
    if (rc != PC_OK) return backend_error(ctx, rc);
    const Fr r_lt = inner_product(l, state.randomness);                                      // :345-348
    const Fr eval = inner_product(lt, r);                                                    // :350
    if (eval_out) *eval_out = eval;
    proof.com_eval = ck.com_key[0].mul(eval).add(ck.h.mul(r_eval));                          // :353
    const Fr b = inner_product(r, d);                                                        // :364
    pc_srs* ext = nullptr;
    rc = upload_ext(ctx, ck, dim, &ext);
    if (rc == PC_OK) rc = pedersen_hiding(ctx, ext, d, r_d, proof.com_d);                    // :368
    pc_hip_srs_free(ext);
    if (rc != PC_OK) return backend_error(ctx, rc);
    proof.com_b = ck.com_key[0].mul(b).add(ck.h.mul(r_b));                                   // :372
    proof.z.resize(dim);
    for (size_t j = 0; j < dim; j++) proof.z[j] = d[j] + c * lt[j];                          // :387
    proof.z_d = c * r_lt + r_d;
    proof.z_b = c * r_eval + r_b;
    return Error();
  }

  static Error check(pc_ctx* ctx, const HyraxCommitterKey<E>& vk, const std::vector<G1Affine<E>>& row_coms, const std::vector<Fr>& point,
                     const HyraxProof<E>& proof, const Fr& c, bool& ok) {
    ok = false;
    const size_t n = point.size();
    if (n % 2 == 1) return invalid_vars(n);                                                  // :431-435
    const size_t dim = (size_t)1 << (n / 2);
    if (row_coms.size() != dim) { Error e; e.kind = Error::IncorrectCommitmentSize; e.a = row_coms.size(); e.b = dim; return e; }   // :461-466
    if (proof.z.size() != dim || dim > vk.com_key.size()) return invalid_vars(n);
    std::vector<Fr> l, r; tensors(point, l, r);
    // equation (14): com(<r, z>, z_b) == c * com_eval + com_b                               :486-489
    const G1Affine<E> com_dp = vk.com_key[0].mul(inner_product(r, proof.z)).add(vk.h.mul(proof.z_b));
    if (!(com_dp == proof.com_eval.mul(c).add(proof.com_b))) return Error();
    // equation (13): com(z, z_d) == c * t' + com_d with t' = MSM(row_coms, l)                :492-501
    pc_srs* rows = nullptr; pc_srs* ext = nullptr;
    uint64_t xy[2 * E::NQ]; int inf = 0;
    int rc = pc_hip_srs_upload(ctx, E::ID, row_coms.data(), dim, sizeof(G1Affine<E>), PC_MEM_HOST, &rows);
    if (rc == PC_OK) rc = pc_hip_msm(ctx, rows, 0, l.data(), PC_SCALARS_MONTGOMERY, PC_MEM_HOST, dim, xy, &inf);
    pc_hip_srs_free(rows);
    if (rc != PC_OK) return backend_error(ctx, rc);
    const G1Affine<E> t_prime = from_out(xy);
    G1Affine<E> com_z_zd;
    rc = upload_ext(ctx, vk, dim, &ext);
    if (rc == PC_OK) rc = pedersen_hiding(ctx, ext, proof.z, proof.z_d, com_z_zd);
    pc_hip_srs_free(ext);
    if (rc != PC_OK) return backend_error(ctx, rc);
    ok = com_z_zd == t_prime.mul(c).add(proof.com_d);
    return Error();
  }
};

}  // namespace pc_host
