// C++ host mirror of MarlinKZG10's commit/open glue (poly-commit/src/marlin/marlin_pc/mod.rs)
// on top of pc_host::KZG10 (kzg10.hpp), i.e. on top of the C ABI.
//
//   CommitterKey::{powers, shifted_powers}       marlin_pc/data_structures.rs:48-83
//   trim (committer half)                          marlin_pc/mod.rs:80-169
//   shift_polynomial                               marlin_pc/mod.rs:34-53
//   check_degrees_and_bounds                       kzg10/mod.rs:424-449
//   MarlinKZG10::commit                            marlin_pc/mod.rs:172-242
//   MarlinKZG10::open                              marlin_pc/mod.rs:245-336
//   MarlinKZG10::batch_open                        marlin_pc/mod.rs:457-530
//   MarlinKZG10::open_combinations                 marlin_pc/mod.rs:407-430 -> Marlin::open_combinations, marlin/mod.rs:224-316
//     (combine_commitments :52-70, normalize_commitments :72-105)
// The Fiat-Shamir sponge is the caller's (`ChallengeSource`): the reference squeezes one
// 128-bit challenge per polynomial (and one more per degree-bounded polynomial) at
// marlin_pc/mod.rs:282,299.  setup/check stay with the reference (verifier side).
#pragma once
#include <algorithm>
#include <map>
#include <optional>
#include <set>
#include <tuple>
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

// LinearCombination<F> (data_structures.rs:248-365): (coefficient, term) pairs; a term without a label is LCTerm::One
template <class E>
struct LinearCombination {
  std::string label;
  std::vector<std::pair<FrT<E>, std::optional<std::string>>> terms;
};
// one element of a QuerySet<T> = BTreeSet<(String, (String, T))> (lib.rs:152): (polynomial / equation label, (point label, point))
template <class E>
struct Query { std::string label, point_label; FrT<E> point; };
template <class E> struct LabeledMarlinCommitment { std::string label; MarlinCommitment<E> commitment; std::optional<size_t> degree_bound; };

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

  // MarlinKZG10::batch_open (marlin_pc/mod.rs:457-530): the queries are grouped by POINT LABEL (BTreeMap order), the polynomials
  // queried at one point are opened together, in label order (BTreeSet), with one `open` above -- one proof per distinct point label.
  static Error batch_open(const CommitterKey<E>& ck, const std::vector<LabeledPolynomial<E>>& labeled_polynomials,
                          const std::vector<Query<E>>& query_set, ChallengeSource<E>& sponge, const std::vector<MarlinRandomness<E>>& states,
                          std::vector<Proof<E>>& proofs) {
    proofs.clear();
    std::map<std::string, size_t> by_label;
    for (size_t i = 0; i < labeled_polynomials.size(); i++) by_label[labeled_polynomials[i].label] = i;
    std::vector<Query<E>> qs = query_set;                     // BTreeSet iteration order: (label, (point label, point))
    std::sort(qs.begin(), qs.end(), [](const Query<E>& a, const Query<E>& b) { return std::tie(a.label, a.point_label) < std::tie(b.label, b.point_label); });
    std::map<std::string, std::pair<Fr, std::set<std::string>>> query_to_labels_map;
    for (const auto& q : qs) {
      auto it = query_to_labels_map.find(q.point_label);
      if (it == query_to_labels_map.end()) it = query_to_labels_map.emplace(q.point_label, std::make_pair(q.point, std::set<std::string>())).first;
      it->second.second.insert(q.label);
    }
    for (const auto& kv : query_to_labels_map) {
      std::vector<LabeledPolynomial<E>> query_polys; std::vector<MarlinRandomness<E>> query_states;
      for (const auto& label : kv.second.second) {
        auto f = by_label.find(label);
        if (f == by_label.end()) { Error e; e.kind = Error::MissingPolynomial; e.msg = label; return e; }
        query_polys.push_back(labeled_polynomials[f->second]); query_states.push_back(states[f->second]);
      }
      Proof<E> proof;
      if (Error e = open(ck, query_polys, kv.second.first, sponge, query_states, proof)) return e;
      proofs.push_back(proof);
    }
    return Error();
  }

  // Marlin::open_combinations (marlin/mod.rs:224-316): per equation the polynomial, the commitment state and the commitment are
  // combined (degree-bounded polynomials only alone and with coefficient one, :267-277; the constant term of an equation is not part
  // of the combined polynomial: the verifier subtracts it from the claimed value, :352-358), then ONE proof per query point is made
  // over the COMBINED polynomials.  `lc_commitments_out`: the combined commitments (combine_commitments + normalize_commitments) the
  // verifier rebuilds on its side -- returned so that a caller without pairings can check the proofs against them.
  // BatchLCProof{proof, evals: None}: `proofs` is the whole proof.
  static Error open_combinations(const CommitterKey<E>& ck, const std::vector<LinearCombination<E>>& lc_s,
                                 const std::vector<LabeledPolynomial<E>>& polynomials, const std::vector<MarlinCommitment<E>>& commitments,
                                 const std::vector<Query<E>>& query_set, ChallengeSource<E>& sponge, const std::vector<MarlinRandomness<E>>& states,
                                 std::vector<Proof<E>>& proofs, std::vector<LabeledMarlinCommitment<E>>* lc_commitments_out = nullptr) {
    std::map<std::string, size_t> label_map;
    for (size_t i = 0; i < polynomials.size(); i++) label_map[polynomials[i].label] = i;
    std::vector<LabeledPolynomial<E>> lc_polynomials; std::vector<MarlinRandomness<E>> lc_states;
    std::vector<LabeledMarlinCommitment<E>> lc_commitments;
    for (const auto& lc : lc_s) {
      LabeledPolynomial<E> lc_poly; lc_poly.label = lc.label;
      MarlinRandomness<E> randomness;                          // PCCommitmentState::empty()
      G1Affine<E> combined_comm = G1Affine<E>::zero(); std::optional<G1Affine<E>> combined_shifted_comm;
      const size_t num_polys = lc.terms.size();
      for (const auto& term : lc.terms) {
        if (!term.second) continue;                            // LCTerm::One
        const Fr& coeff = term.first;
        auto f = label_map.find(*term.second);
        if (f == label_map.end()) { Error e; e.kind = Error::MissingPolynomial; e.msg = *term.second; return e; }
        const LabeledPolynomial<E>& cur_poly = polynomials[f->second];
        const MarlinRandomness<E>& cur_state = states[f->second];
        const MarlinCommitment<E>& cur_comm = commitments[f->second];
        if (num_polys == 1 && cur_poly.degree_bound) {
          if (!(coeff == Fr::one())) { Error e; e.kind = Error::InvalidParameters; e.msg = "Coefficient must be one for degree-bounded equations"; return e; }   // an assert! in the reference
          lc_poly.degree_bound = cur_poly.degree_bound;
        } else if (cur_poly.degree_bound) { Error e; e.kind = Error::EquationHasDegreeBounds; e.msg = lc.label; return e; }
        if (cur_poly.hiding_bound && (!lc_poly.hiding_bound || *lc_poly.hiding_bound < *cur_poly.hiding_bound)) lc_poly.hiding_bound = cur_poly.hiding_bound;   // max, Some(_) > None
        axpy(lc_poly.polynomial, coeff, cur_poly.polynomial);
        // randomness += (coeff, cur_state)   (marlin_pc/data_structures.rs:322-343)
        axpy(randomness.rand.blinding_polynomial, coeff, cur_state.rand.blinding_polynomial);
        if (cur_state.shifted_rand) {
          if (!randomness.shifted_rand) randomness.shifted_rand = Randomness<E>::empty();
          axpy(randomness.shifted_rand->blinding_polynomial, coeff, cur_state.shifted_rand->blinding_polynomial);
        }
        // combine_commitments (marlin/mod.rs:52-70)
        combined_comm = combined_comm.add(coeff == Fr::one() ? cur_comm.comm : cur_comm.comm.mul(coeff));
        if (cur_comm.shifted_comm) {
          const G1Affine<E> cur = cur_comm.shifted_comm->mul(coeff);
          combined_shifted_comm = combined_shifted_comm ? combined_shifted_comm->add(cur) : cur;
        }
      }
      lc_polynomials.push_back(lc_poly); lc_states.push_back(randomness);
      LabeledMarlinCommitment<E> c; c.label = lc.label; c.degree_bound = lc_poly.degree_bound;
      c.commitment.comm = combined_comm; c.commitment.shifted_comm = combined_shifted_comm;      // (affine already: normalize_commitments is the identity here)
      lc_commitments.push_back(c);
    }
    if (lc_commitments_out) *lc_commitments_out = lc_commitments;
    return batch_open(ck, lc_polynomials, query_set, sponge, lc_states, proofs);
  }
};

}  // namespace pc_host
