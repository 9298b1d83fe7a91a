// C++ host mirror of MarlinKZG10's commit/open glue (poly-commit/src/marlin/marlin_pc/mod.rs)
// on top of pc_host::KZG10 (kzg10.hpp), i.e. on top of the C ABI.
//
//   CommitterKey::{powers, shifted_powers}       marlin_pc/data_structures.rs:48-83
//   trim (committer half)                          marlin_pc/mod.rs:80-169
//   shift_polynomial                               marlin_pc/mod.rs:34-53
//   check_degrees_and_bounds                       kzg10/mod.rs:424-449
//   MarlinKZG10::commit                            marlin_pc/mod.rs:172-242
//   MarlinKZG10::open                              marlin_pc/mod.rs:245-336
// The Fiat-Shamir sponge is the caller's (`ChallengeSource`): the reference squeezes one
// 128-bit challenge per polynomial (and one more per degree-bounded polynomial) at
// marlin_pc/mod.rs:282,299.  setup/check stay with the reference (verifier side).
#pragma once
#include <algorithm>
#include <optional>
#include "kzg10.hpp"

namespace pc_host {

template <class E>
struct LabeledPolynomial {          // data_structures.rs:109-180
  std::string label;
  DensePolynomial<E> polynomial;
  std::optional<size_t> degree_bound, hiding_bound;
};

template <class E> struct MarlinCommitment { G1Affine<E> comm = G1Affine<E>::zero(); std::optional<G1Affine<E>> shifted_comm; };   // :227-235
template <class E> struct MarlinRandomness { Randomness<E> rand; std::optional<Randomness<E>> shifted_rand; };                     // :304-311

template <class E>
struct ChallengeSource { virtual ~ChallengeSource() {} virtual FrT<E> squeeze_challenge() = 0; };

template <class E>
struct CommitterKey {               // marlin_pc/data_structures.rs:26-44
  std::vector<G1Affine<E>> powers, powers_of_gamma_g;
  std::optional<std::vector<G1Affine<E>>> shifted_powers_vec;
  std::optional<std::vector<size_t>> enforced_degree_bounds;
  size_t max_degree = 0;
  // HBM residents (uploaded once in trim)
  pc_ctx* ctx = nullptr; pc_srs* srs_powers = nullptr; pc_srs* srs_shifted = nullptr; pc_srs* srs_gamma = nullptr;

  size_t supported_degree() const { return powers.size() - 1; }
  Powers<E> powers_view() const {                                                    // powers()
    Powers<E> p; p.ctx = ctx; p.powers_of_g = powers.data(); p.n_powers = powers.size();
    p.powers_of_gamma_g = powers_of_gamma_g.data(); p.n_gamma = powers_of_gamma_g.size();
    p.srs_g = srs_powers; p.srs_gamma = srs_gamma; p.g_offset = 0; return p;
  }
  // shifted_powers(degree_bound): the slice shifted_powers[(max_bound - degree_bound)..]
  std::optional<Powers<E>> shifted_powers(std::optional<size_t> degree_bound) const {
    if (!shifted_powers_vec) return std::nullopt;
    size_t start = 0;
    if (degree_bound) start = enforced_degree_bounds->back() - *degree_bound;
    Powers<E> p; p.ctx = ctx; p.powers_of_g = shifted_powers_vec->data() + start; p.n_powers = shifted_powers_vec->size() - start;
    p.powers_of_gamma_g = powers_of_gamma_g.data(); p.n_gamma = powers_of_gamma_g.size();
    p.srs_g = srs_shifted; p.srs_gamma = srs_gamma; p.g_offset = start; return p;
  }
  void release() { pc_hip_srs_free(srs_powers); pc_hip_srs_free(srs_shifted); pc_hip_srs_free(srs_gamma); srs_powers = srs_shifted = srs_gamma = nullptr; }
};

template <class E>
struct MarlinKZG10 {
  typedef FrT<E> Fr; typedef KZG10<E> K;

  // committer half of trim(): pp_powers_of_g has max_degree + 1 points, pp_gamma at least supported_hiding_bound + 2
  static Error trim(pc_ctx* ctx, const std::vector<G1Affine<E>>& pp_powers_of_g, const std::vector<G1Affine<E>>& pp_powers_of_gamma_g,
                    size_t supported_degree, size_t supported_hiding_bound, const std::vector<size_t>* enforced_degree_bounds,
                    CommitterKey<E>& ck) {
    ck = CommitterKey<E>(); ck.ctx = ctx;
    ck.max_degree = pp_powers_of_g.size() - 1;
    ck.powers.assign(pp_powers_of_g.begin(), pp_powers_of_g.begin() + supported_degree + 1);
    ck.powers_of_gamma_g.assign(pp_powers_of_gamma_g.begin(), pp_powers_of_gamma_g.begin() + supported_hiding_bound + 2);
    if (enforced_degree_bounds) {
      std::vector<size_t> v = *enforced_degree_bounds; std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
      ck.enforced_degree_bounds = v;
      if (!v.empty()) {
        size_t lowest_shifted_power = ck.max_degree - v.back();
        ck.shifted_powers_vec = std::vector<G1Affine<E>>(pp_powers_of_g.begin() + lowest_shifted_power, pp_powers_of_g.end());
      }
    }
    auto up = [&](const std::vector<G1Affine<E>>& v, pc_srs** out) {
      return pc_hip_srs_upload(ctx, E::ID, v.data(), v.size(), sizeof(G1Affine<E>), PC_MEM_HOST, out);
    };
    int rc = up(ck.powers, &ck.srs_powers);
    if (rc == PC_OK) rc = up(ck.powers_of_gamma_g, &ck.srs_gamma);
    if (rc == PC_OK && ck.shifted_powers_vec) rc = up(*ck.shifted_powers_vec, &ck.srs_shifted);
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    return Error();
  }

  // kzg10/mod.rs:424-449
  static Error check_degrees_and_bounds(const CommitterKey<E>& ck, const LabeledPolynomial<E>& p) {
    if (p.degree_bound) {
      size_t bound = *p.degree_bound;
      Error e; e.kind = Error::UnsupportedDegreeBound; e.a = bound;
      if (!ck.enforced_degree_bounds) return e;
      if (!std::binary_search(ck.enforced_degree_bounds->begin(), ck.enforced_degree_bounds->end(), bound)) return e;
      if (bound < p.polynomial.degree() || bound > ck.max_degree) { e.kind = Error::UnsupportedDegreeBound; e.msg = "IncorrectDegreeBound: " + p.label; return e; }
    }
    return Error();
  }

  static DensePolynomial<E> shift_polynomial(const CommitterKey<E>& ck, const DensePolynomial<E>& p, size_t degree_bound) {   // :34-53
    DensePolynomial<E> out;
    if (p.is_zero()) return out;
    out.coeffs.assign(ck.enforced_degree_bounds->back() - degree_bound, Fr::zero());
    out.coeffs.insert(out.coeffs.end(), p.coeffs.begin(), p.coeffs.end());
    return out;
  }

  // marlin_pc/mod.rs:172-242
  static Error commit(const CommitterKey<E>& ck, const std::vector<LabeledPolynomial<E>>& polynomials, RngCore<E>* rng,
                      std::vector<MarlinCommitment<E>>& commitments, std::vector<MarlinRandomness<E>>& states) {
    commitments.clear(); states.clear();
    for (const auto& p : polynomials) {
      if (Error e = check_degrees_and_bounds(ck, p)) return e;
      const size_t* hb = p.hiding_bound ? &*p.hiding_bound : nullptr;
      Commitment<E> comm; Randomness<E> rand;
      if (Error e = K::commit(ck.powers_view(), p.polynomial, hb, rng, comm, rand)) return e;
      MarlinCommitment<E> mc; MarlinRandomness<E> mr; mc.comm = comm.comm; mr.rand = rand;
      if (p.degree_bound) {
        auto sp = ck.shifted_powers(p.degree_bound);
        if (!sp) { Error e; e.kind = Error::UnsupportedDegreeBound; e.a = *p.degree_bound; return e; }
        Commitment<E> sc; Randomness<E> sr;
        if (Error e = K::commit(*sp, p.polynomial, hb, rng, sc, sr)) return e;      // a second full MSM (:219-225)
        mc.shifted_comm = sc.comm; mr.shifted_rand = sr;
      }
      commitments.push_back(mc); states.push_back(mr);
    }
    return Error();
  }

  static void axpy(DensePolynomial<E>& acc, const Fr& c, const DensePolynomial<E>& p) {     // p += (challenge, poly)
    if (acc.coeffs.size() < p.coeffs.size()) acc.coeffs.resize(p.coeffs.size(), Fr::zero());
    for (size_t i = 0; i < p.coeffs.size(); i++) acc.coeffs[i] = acc.coeffs[i] + c * p.coeffs[i];
  }

  // marlin_pc/mod.rs:245-336
  static Error open(const CommitterKey<E>& ck, const std::vector<LabeledPolynomial<E>>& labeled_polynomials, const Fr& point,
                    ChallengeSource<E>& sponge, const std::vector<MarlinRandomness<E>>& states, Proof<E>& out) {
    DensePolynomial<E> p, shifted_w, shifted_r_witness;
    Randomness<E> r = Randomness<E>::empty(), shifted_r = Randomness<E>::empty();
    bool enforce_degree_bound = false;
    for (size_t j = 0; j < labeled_polynomials.size(); j++) {
      const auto& polynomial = labeled_polynomials[j]; const auto& rand = states[j];
      if (Error e = check_degrees_and_bounds(ck, polynomial)) return e;
      Fr challenge_j = sponge.squeeze_challenge();                                              // :282
      axpy(p, challenge_j, polynomial.polynomial);                                               // :286
      axpy(r.blinding_polynomial, challenge_j, rand.rand.blinding_polynomial);                   // :287
      if (polynomial.degree_bound) {
        enforce_degree_bound = true;
        const Randomness<E>& shifted_rand = *rand.shifted_rand;
        DensePolynomial<E> witness, shifted_rand_witness; bool has_rw = false;
        if (Error e = K::compute_witness_polynomial(ck.ctx, polynomial.polynomial, point, shifted_rand, witness, &shifted_rand_witness, has_rw)) return e;
        Fr challenge_j_1 = sponge.squeeze_challenge();                                           // :299
        DensePolynomial<E> shifted_witness = shift_polynomial(ck, witness, *polynomial.degree_bound);
        axpy(shifted_w, challenge_j_1, shifted_witness);
        axpy(shifted_r.blinding_polynomial, challenge_j_1, shifted_rand.blinding_polynomial);
        if (has_rw) axpy(shifted_r_witness, challenge_j_1, shifted_rand_witness);
      }
    }
    Proof<E> proof;
    if (Error e = K::open(ck.powers_view(), p, point, r, proof)) return e;                       // :310
    G1Affine<E> w = proof.w; bool has_v = proof.has_random_v; Fr random_v = proof.random_v;
    if (enforce_degree_bound) {
      Proof<E> shifted_proof;
      auto sp = ck.shifted_powers(std::nullopt);
      const bool hiding = !shifted_r_witness.coeffs.empty();
      if (Error e = K::open_with_witness_polynomial(*sp, point, shifted_r, shifted_w, hiding ? &shifted_r_witness : nullptr, shifted_proof)) return e;
      w = w.add(shifted_proof.w);                                                               // :326
      if (shifted_proof.has_random_v && has_v) random_v = random_v + shifted_proof.random_v;     // :327-329
    }
    out.w = w; out.has_random_v = has_v; out.random_v = random_v;
    return Error();
  }
};

}  // namespace pc_host
