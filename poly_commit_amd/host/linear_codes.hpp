// C++ host mirror of the Ligero encoder glue (poly-commit/src/linear_codes/), above the C ABI.
//
//   calculate_t                         linear_codes/utils.rs:156-184
//   LigeroPCParams::compute_dimensions  linear_codes/ligero.rs:118-128
//   reed_solomon                        linear_codes/utils.rs:112-127
//   LinearEncode::compute_matrices      linear_codes/mod.rs:118-138
//   LinearCodePCS::commit               linear_codes/mod.rs:234-297 (steps 1-4, one polynomial at a time)
//   generate_proof step 1 (row_mul)     linear_codes/mod.rs:539, poly-commit/src/utils.rs:120-147
// Matrix<F> is the reference's dense row-major matrix (utils.rs:49-147) flattened.
#pragma once
#include <array>
#include <math.h>
#include "kzg10.hpp"
#include "transcript.hpp"

namespace pc_host {

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
inline uint32_t ark_log2(size_t x) { if (x <= 1) return 0; uint32_t b = 0; size_t v = x - 1; while (v) { b++; v >>= 1; } return b; }   // ark_std::log2

// Ok(t) or -1 (Error::InvalidParameters)
inline long calculate_t(int field_bits, size_t sec_param, size_t dist_num, size_t dist_den, size_t codeword_len) {
  double residual = (double)codeword_len / pow(2.0, field_bits);
  double rhs = log2(pow(2.0, -(double)sec_param) - residual);
  if (!isnormal(rhs)) return -1;
  double nom = rhs - 1.0;
  double denom = log2(1.0 - 0.5 * (double)dist_num / (double)dist_den);
  if (!isnormal(denom)) return -1;
  size_t t = (size_t)ceil(nom / denom);
  return (long)(t < codeword_len ? t : codeword_len);
}

struct LigeroPCParams {              // linear_codes/ligero.rs:22-39 (the fields the encoder needs)
  size_t sec_param = 128, rho_inv = 4;
  // (n_rows, n_cols) for a polynomial of poly_len coefficients
  // The reference unwraps calculate_t here and aborts with Error::InvalidParameters (ligero.rs:124); the mirror
  // returns that error instead of committing with a meaningless shape.
  template <class E>
  Error compute_dimensions(size_t poly_len, size_t& n, size_t& m) const {
    long t = calculate_t(E::C::FrP::BITS, sec_param, rho_inv - 1, rho_inv, poly_len);
    if (t < 0) { Error e; e.kind = Error::InvalidParameters; e.msg = "calculate_t: the field is not big enough for this codeword length / security parameter, or the distance is wrong"; return e; }
    if (t == 0) { Error e; e.kind = Error::InvalidParameters; e.msg = "calculate_t: empty polynomial"; return e; }
    n = (size_t)1 << ark_log2((size_t)ceil(sqrt((double)ceil_div(2 * poly_len, (size_t)t))));
    m = ceil_div(poly_len, n);
    return Error();
  }
};

template <class E>
struct Matrix { size_t n = 0, m = 0; std::vector<FrT<E>> entries; FrT<E>& at(size_t r, size_t c) { return entries[r * m + c]; } };

template <class E>
struct LinearEncode {
  // reed_solomon(msg, rho_inv): one row -> next_pow2(len * rho_inv) evaluations
  static Error reed_solomon(pc_ctx* ctx, const std::vector<FrT<E>>& msg, size_t rho_inv, std::vector<FrT<E>>& out) {
    Matrix<E> in; in.n = 1; in.m = msg.size(); in.entries = msg;
    Matrix<E> ext; Error e = encode_rows(ctx, in, rho_inv, ext); out = ext.entries; return e;
  }
  static Error encode_rows(pc_ctx* ctx, const Matrix<E>& mat, size_t rho_inv, Matrix<E>& ext) {
    size_t size = 1; unsigned lg = 0; while (size < mat.m * rho_inv) { size <<= 1; lg++; }
    ext.n = mat.n; ext.m = size; ext.entries.assign(mat.n * size, FrT<E>::zero());
    int rc = pc_hip_ntt_batch(ctx, E::ID, mat.entries.data(), PC_MEM_HOST, mat.n, mat.m, lg, ext.entries.data(), PC_MEM_HOST);
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    return Error();
  }
  // compute_matrices: pad the coefficients to n_rows*n_cols, row-major, encode every row
  static Error compute_matrices(pc_ctx* ctx, const DensePolynomial<E>& polynomial, const LigeroPCParams& param, Matrix<E>& mat, Matrix<E>& ext_mat) {
    std::vector<FrT<E>> coeffs = polynomial.coeffs;
    std::pair<size_t, size_t> dims;
    if (Error e = param.compute_dimensions<E>(coeffs.size(), dims.first, dims.second)) return e;
    coeffs.resize(dims.first * dims.second, FrT<E>::zero());
    mat.n = dims.first; mat.m = dims.second; mat.entries = coeffs;
    return encode_rows(ctx, mat, param.rho_inv, ext_mat);
  }
};

// Metadata / LinCodePCCommitment / LinCodePCCommitmentState (linear_codes/data_structures.rs:84-125);
// the state additionally keeps the inner Merkle nodes so that open does not rebuild the tree
// (the reference rebuilds it from `leaves` in open, mod.rs:329-336)
struct Metadata { size_t n_rows = 0, n_cols = 0, n_ext_cols = 0; };
struct LinCodePCCommitment { Metadata metadata; uint8_t root[32] = {0}; };
template <class E>
struct LinCodePCCommitmentState {
  Matrix<E> mat, ext_mat;
  std::vector<uint8_t> leaves;     // n_ext_cols x 32: column digests
  std::vector<uint8_t> nodes;      // inner Merkle nodes, heap order, root first (MerkleTree::non_leaf_nodes)
};

template <class E>
struct LinearCodePCS {
  pc_hash col_hash = PC_HASH_BLAKE2S;      // FieldToBytesColHasher<F, Blake2s256>
  pc_hash tree_hash = PC_HASH_SHA256;      // TwoToOneHash = Sha256
  bool len_prefix = true;                  // ByteDigestConverter
  // commit to one polynomial: matrix -> encode -> column digests -> Merkle tree, chained on the device
  Error commit(pc_ctx* ctx, const DensePolynomial<E>& polynomial, const LigeroPCParams& param, LinCodePCCommitment& com,
               LinCodePCCommitmentState<E>& state, bool keep_ext_mat = true) const {
    std::vector<FrT<E>> coeffs = polynomial.coeffs;
    std::pair<size_t, size_t> dims;
    if (Error e = param.compute_dimensions<E>(coeffs.size(), dims.first, dims.second)) return e;
    coeffs.resize(dims.first * dims.second, FrT<E>::zero());
    state.mat.n = dims.first; state.mat.m = dims.second; state.mat.entries = coeffs;
    size_t size = 1; unsigned lg = 0; while (size < dims.second * param.rho_inv) { size <<= 1; lg++; }
    state.ext_mat.n = dims.first; state.ext_mat.m = size;
    if (keep_ext_mat) state.ext_mat.entries.assign(dims.first * size, FrT<E>::zero()); else state.ext_mat.entries.clear();
    state.leaves.assign(size * 32, 0);
    const size_t n_nodes = (size > 1 ? size : 2) - 1;
    state.nodes.assign(n_nodes * 32, 0);
    int rc = pc_hip_ligero_commit(ctx, E::ID, state.mat.entries.data(), PC_MEM_HOST, dims.first, dims.second, lg, col_hash,
                                  tree_hash, len_prefix ? 1 : 0, keep_ext_mat ? state.ext_mat.entries.data() : nullptr,
                                  PC_MEM_HOST, state.leaves.data(), state.nodes.data());
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    com.metadata.n_rows = dims.first; com.metadata.n_cols = dims.second; com.metadata.n_ext_cols = size;
    memcpy(com.root, state.nodes.data(), 32);
    return Error();
  }
  // v = b^T * mat (generate_proof step 1)
  static Error row_mul(pc_ctx* ctx, const Matrix<E>& mat, const std::vector<FrT<E>>& b, std::vector<FrT<E>>& v) {
    if (b.size() != mat.n) { Error e; e.kind = Error::Backend; e.msg = "Invalid row multiplication: vector length differs from the number of rows"; return e; }
    std::vector<const void*> rows(mat.n); std::vector<size_t> lens(mat.n, mat.m);
    for (size_t r = 0; r < mat.n; r++) rows[r] = mat.entries.data() + r * mat.m;
    v.assign(mat.m, FrT<E>::zero());
    int rc = pc_hip_fr_lincomb(ctx, E::ID, rows.data(), PC_MEM_HOST, lens.data(), mat.n, b.data(), v.data(), PC_MEM_HOST, mat.m);
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    return Error();
  }
  // Merkle authentication path of column `index` out of the commitment state (col_tree.generate_proof,
  // mod.rs:555-557): the sibling leaf digest, then the sibling inner nodes from the bottom up.
  static void merkle_path(const LinCodePCCommitmentState<E>& st, size_t index, uint8_t leaf_sibling[32],
                          std::vector<uint8_t>& path) {
    const size_t n_inner = st.nodes.size() / 32;
    memcpy(leaf_sibling, st.leaves.data() + (index ^ 1) * 32, 32);
    path.clear();
    for (size_t node = (n_inner + index + 1) / 2 - 1; node > 0; node = (node - 1) / 2) {
      const size_t sib = (node & 1) ? node + 1 : node - 1;
      path.insert(path.end(), st.nodes.begin() + sib * 32, st.nodes.begin() + (sib + 1) * 32);
    }
  }

  // ---- open / check (linear_codes/mod.rs:300-503, generate_proof :523-565).  The sponge is the caller's: the t query
  //      indices (get_indices_from_sponge, utils.rs:136-153) and, with check_well_formedness, the vector r are arguments.
  struct ProofSingle {                               // LinCodePCProof { opening: LinCodePCProofSingle { paths, v, columns }, well_formedness }
    std::vector<size_t> leaf_index;
    std::vector<std::array<uint8_t, 32>> leaf_sibling;
    std::vector<std::vector<uint8_t>> paths;         // per query: sibling inner nodes, bottom-up
    std::vector<FrT<E>> v;
    std::vector<std::vector<FrT<E>>> columns;
    bool has_well_formedness = false;
    std::vector<FrT<E>> well_formedness;
  };
  // UnivariateLigero::tensor (univariate_ligero/mod.rs:70-86): ((1, z, .., z^(left-1)), (1, z^left, z^(2 left), ..))
  static void tensor(const FrT<E>& z, size_t left, size_t right, std::vector<FrT<E>>& a, std::vector<FrT<E>>& b) {
    a.clear(); b.clear();
    FrT<E> pw = FrT<E>::one();
    for (size_t i = 0; i < left; i++) { a.push_back(pw); pw = pw * z; }
    FrT<E> q = FrT<E>::one();
    for (size_t i = 0; i < right; i++) { b.push_back(q); q = q * pw; }
  }
  static Error invalid_commitment(const char* what) { Error e; e.kind = Error::InvalidCommitment; e.msg = what; return e; }
  void two_to_one(const uint8_t* data, size_t len, uint8_t out[32]) const {
    if (tree_hash == PC_HASH_SHA256) Sha256Host::digest(data, len, out); else Blake2s::digest(data, len, out);
  }
  // Path::verify for the byte-digest Config: bottom level D(conv(l) || conv(r)), upper levels D(left || right)
  bool verify_path(const uint8_t root[32], const uint8_t leaf[32], size_t index, const uint8_t sibling[32], const std::vector<uint8_t>& path) const {
    uint8_t buf[96], cur[32]; size_t n = 0;
    auto put = [&](const uint8_t* d) { if (len_prefix) { uint64_t l = 32; memcpy(buf + n, &l, 8); n += 8; } memcpy(buf + n, d, 32); n += 32; };
    if (index % 2 == 0) { put(leaf); put(sibling); } else { put(sibling); put(leaf); }
    two_to_one(buf, n, cur);
    index /= 2;
    for (size_t k = 0; k + 32 <= path.size(); k += 32, index /= 2) {
      if (index % 2 == 0) { memcpy(buf, cur, 32); memcpy(buf + 32, path.data() + k, 32); } else { memcpy(buf, path.data() + k, 32); memcpy(buf + 32, cur, 32); }
      two_to_one(buf, 64, cur);
    }
    return memcmp(cur, root, 32) == 0;
  }
  Error open(pc_ctx* ctx, const LinCodePCCommitment& com, const LinCodePCCommitmentState<E>& st, const FrT<E>& z,
             const std::vector<size_t>& indices, const std::vector<FrT<E>>* r, ProofSingle& proof) const {
    if (st.ext_mat.entries.empty()) { Error e; e.kind = Error::Backend; e.msg = "open needs the encoded matrix in the commitment state"; return e; }
    std::vector<FrT<E>> a, b; tensor(z, com.metadata.n_cols, com.metadata.n_rows, a, b);
    proof = ProofSingle();
    if (r) { if (Error e = row_mul(ctx, st.mat, *r, proof.well_formedness)) return e; proof.has_well_formedness = true; }     // :343-349
    if (Error e = row_mul(ctx, st.mat, b, proof.v)) return e;                                                              // generate_proof 1.
    for (size_t idx : indices) {                                                                                            // 3.
      if (idx >= st.ext_mat.m) return invalid_commitment("query index out of range");
      std::vector<FrT<E>> col(st.ext_mat.n);
      for (size_t row = 0; row < st.ext_mat.n; row++) col[row] = st.ext_mat.entries[row * st.ext_mat.m + idx];
      proof.columns.push_back(col);
      std::array<uint8_t, 32> sib; std::vector<uint8_t> path;
      merkle_path(st, idx, sib.data(), path);
      proof.leaf_index.push_back(idx); proof.leaf_sibling.push_back(sib); proof.paths.push_back(path);
    }
    return Error();
  }
  // ok == false: the claimed value is wrong (:494-499); Error::InvalidCommitment: a path, an index or an inner product fails
  Error check(pc_ctx* ctx, const LinCodePCCommitment& com, const LigeroPCParams& param, const FrT<E>& z, const FrT<E>& value,
              const ProofSingle& proof, const std::vector<size_t>& indices, const std::vector<FrT<E>>* r, bool& ok) const {
    ok = false;
    const size_t n_rows = com.metadata.n_rows, n_cols = com.metadata.n_cols, n_ext = com.metadata.n_ext_cols, t = indices.size();
    if ((r != nullptr) != proof.has_well_formedness) return invalid_commitment("well-formedness proof missing or unexpected");
    // every container the loops below index is sized against the t query indices first (an untrusted prover's proof must end in
    // InvalidCommitment, never in an out-of-bounds read; the reference indexes all of them per query, linear_codes/mod.rs:443-489)
    unsigned height = 1; while (((size_t)1 << height) < n_ext) height++;
    if (proof.columns.size() != t || proof.paths.size() != t || proof.leaf_index.size() != t || proof.leaf_sibling.size() != t ||
        proof.v.size() != n_cols || (r && proof.well_formedness.size() != n_cols) || (r && r->size() != n_rows) ||
        n_ext == 0 || (n_ext & (n_ext - 1)) || n_ext < n_cols)
      return invalid_commitment("proof shape");
    for (size_t j = 0; j < t; j++)
      if (indices[j] >= n_ext || proof.paths[j].size() != (size_t)32 * (height - 1)) return invalid_commitment("proof shape");
    // 3. hash the received columns (device: they are the columns of an n_rows x t matrix)
    std::vector<FrT<E>> m(n_rows * t);
    for (size_t j = 0; j < t; j++) { if (proof.columns[j].size() != n_rows) return invalid_commitment("column length"); for (size_t row = 0; row < n_rows; row++) m[row * t + j] = proof.columns[j][row]; }
    std::vector<uint8_t> digests(t * 32);
    int rc = t ? pc_hip_column_hash(ctx, E::ID, m.data(), PC_MEM_HOST, n_rows, t, col_hash, digests.data(), PC_MEM_HOST) : PC_OK;
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    for (size_t j = 0; j < t; j++)                                                                                          // 4.
      if (proof.leaf_index[j] != indices[j] || !verify_path(com.root, &digests[j * 32], indices[j], proof.leaf_sibling[j].data(), proof.paths[j]))
        return invalid_commitment("Merkle path");
    std::vector<FrT<E>> w, wwf;                                                                                             // 5.
    if (Error e = LinearEncode<E>::reed_solomon(ctx, proof.v, param.rho_inv, w)) return e;
    if (r) if (Error e = LinearEncode<E>::reed_solomon(ctx, proof.well_formedness, param.rho_inv, wwf)) return e;
    if (w.size() != n_ext) return invalid_commitment("encoding length");
    std::vector<FrT<E>> a, b; tensor(z, n_cols, n_rows, a, b);                                                               // 6.
    auto ip = [](const std::vector<FrT<E>>& x, const std::vector<FrT<E>>& y) { FrT<E> acc = FrT<E>::zero(); for (size_t i = 0; i < x.size() && i < y.size(); i++) acc = acc + x[i] * y[i]; return acc; };
    for (size_t j = 0; j < t; j++) {                                                                                         // 7.
      if (r && !(ip(*r, proof.columns[j]) == wwf[indices[j]])) return invalid_commitment("well-formedness inner product");
      if (!(ip(b, proof.columns[j]) == w[indices[j]])) return invalid_commitment("b . column != w");
    }
    ok = ip(proof.v, a) == value;
    return Error();
  }
};

// omega of the size-2^lg domain (arkworks FftField constants), host side
template <class E>
FrT<E> domain_generator(unsigned lg) {
  FrT<E> w; memcpy(w.l, E::C::FrP::ROOT, 32);
  for (unsigned i = lg; i < (unsigned)E::C::FrP::TWO_ADICITY; i++) w = w * w;
  return w;
}

}  // namespace pc_host
