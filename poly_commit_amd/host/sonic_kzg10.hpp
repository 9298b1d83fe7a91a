// C++ host mirror of SonicKZG10's commit/open glue (poly-commit/src/sonic_pc/mod.rs) on top of pc_host::KZG10 (kzg10.hpp), i.e. on
// top of the C ABI -- SURVEY.md 8(f) rank 4: Sonic reaches the MSM only through KZG10::commit / KZG10::open
// (sonic_pc/mod.rs:325, :378), so its whole committer side is the same two entry points over other slices of one resident SRS.
//
//   CommitterKey::{powers, shifted_powers}       sonic_pc/data_structures.rs:72-110
//   trim (committer half)                          sonic_pc/mod.rs:140-271
//   SonicKZG10::commit                             sonic_pc/mod.rs:273-338   one KZG10::commit per polynomial, over the SHIFTED powers
//                                                                           when it carries a degree bound (a commitment to
//                                                                           x^(D - d) p, not a second commitment as in Marlin)
//   SonicKZG10::open                               sonic_pc/mod.rs:340-383   one random combination, one KZG10::open over powers()
// What differs from marlin_kzg10.hpp: the shifted key keeps its own gamma powers per degree bound (the hiding polynomial of a
// bounded commitment is shifted too), and `open` never touches the shifted key (the verifier adjusts the witness instead,
// accumulate_elems :540-610).  The sponge is the caller's (ChallengeSource); setup / check stay with the reference.
#pragma once
#include <map>
#include "marlin_kzg10.hpp"

namespace pc_host {

template <class E>
struct SonicCommitterKey {          // sonic_pc/data_structures.rs:40-70
  std::vector<G1Affine<E>> powers_of_g, powers_of_gamma_g;
  std::optional<std::vector<G1Affine<E>>> shifted_powers_of_g;
  std::map<size_t, std::vector<G1Affine<E>>> shifted_powers_of_gamma_g;      // per degree bound
  std::optional<std::vector<size_t>> enforced_degree_bounds;
  size_t max_degree = 0;
  // HBM residents (uploaded once in trim)
  pc_ctx* ctx = nullptr; pc_srs* srs_powers = nullptr; pc_srs* srs_shifted = nullptr; pc_srs* srs_gamma = nullptr;
  std::map<size_t, pc_srs*> srs_shifted_gamma;

  size_t supported_degree() const { return powers_of_g.size() - 1; }
  Powers<E> powers() const {
    Powers<E> p; p.ctx = ctx; p.powers_of_g = powers_of_g.data(); p.n_powers = powers_of_g.size();
    p.powers_of_gamma_g = powers_of_gamma_g.data(); p.n_gamma = powers_of_gamma_g.size();
    p.srs_g = srs_powers; p.srs_gamma = srs_gamma; p.g_offset = 0; return p;
  }
  // shifted_powers(degree_bound): shifted_powers_of_g[(max_bound - degree_bound)..] with the gamma powers of that bound
  std::optional<Powers<E>> shifted_powers(std::optional<size_t> degree_bound) const {
    if (!shifted_powers_of_g || !enforced_degree_bounds || enforced_degree_bounds->empty()) return std::nullopt;
    const size_t max_bound = enforced_degree_bounds->back();
    const size_t bound = degree_bound ? *degree_bound : max_bound;
    auto gi = shifted_powers_of_gamma_g.find(bound);
    if (gi == shifted_powers_of_gamma_g.end()) return std::nullopt;
    const size_t start = max_bound - bound;
    Powers<E> p; p.ctx = ctx; p.powers_of_g = shifted_powers_of_g->data() + start; p.n_powers = shifted_powers_of_g->size() - start;
    p.powers_of_gamma_g = gi->second.data(); p.n_gamma = gi->second.size();
    p.srs_g = srs_shifted; p.g_offset = start;
    auto si = srs_shifted_gamma.find(bound); p.srs_gamma = si == srs_shifted_gamma.end() ? nullptr : si->second;
    return p;
  }
  void release() {
    pc_hip_srs_free(srs_powers); pc_hip_srs_free(srs_shifted); pc_hip_srs_free(srs_gamma); srs_powers = srs_shifted = srs_gamma = nullptr;
    for (auto& kv : srs_shifted_gamma) pc_hip_srs_free(kv.second);
    srs_shifted_gamma.clear();
  }
};

template <class E>
struct SonicKZG10 {
  typedef FrT<E> Fr; typedef KZG10<E> K;

  // committer half of trim(): pp_powers_of_g has max_degree + 1 points, pp_powers_of_gamma_g max_degree + 2 (index = degree)
  static Error trim(pc_ctx* ctx, const std::vector<G1Affine<E>>& pp_powers_of_g, const std::vector<G1Affine<E>>& pp_powers_of_gamma_g,
                    size_t supported_degree, size_t supported_hiding_bound, const std::vector<size_t>* enforced_degree_bounds,
                    SonicCommitterKey<E>& ck) {
    ck = SonicCommitterKey<E>(); ck.ctx = ctx;
    const size_t max_degree = pp_powers_of_g.size() - 1;
    ck.max_degree = max_degree;
    if (supported_degree > max_degree) { Error e; e.kind = Error::TrimmingDegreeTooLarge; return e; }          // :150-152
    if (enforced_degree_bounds) {
      std::vector<size_t> v = *enforced_degree_bounds; std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());   // :155-160
      ck.enforced_degree_bounds = v;
      if (!v.empty()) {
        const size_t highest = v.back();
        if (highest > supported_degree) { Error e; e.kind = Error::UnsupportedDegreeBound; e.a = highest; return e; }   // :186-188
        const size_t lowest_shift_degree = max_degree - highest;
        ck.shifted_powers_of_g = std::vector<G1Affine<E>>(pp_powers_of_g.begin() + lowest_shift_degree, pp_powers_of_g.end());
        for (size_t bound : v) {                                                                               // :199-211
          const size_t shift_degree = max_degree - bound;
          std::vector<G1Affine<E>> g;
          for (size_t i = 0; i <= supported_hiding_bound + 1; i++)
            if (shift_degree + i < max_degree + 2 && shift_degree + i < pp_powers_of_gamma_g.size()) g.push_back(pp_powers_of_gamma_g[shift_degree + i]);
          ck.shifted_powers_of_gamma_g[bound] = g;
        }
      }
    }
    ck.powers_of_g.assign(pp_powers_of_g.begin(), pp_powers_of_g.begin() + supported_degree + 1);                // :239
    ck.powers_of_gamma_g.assign(pp_powers_of_gamma_g.begin(), pp_powers_of_gamma_g.begin() + supported_hiding_bound + 2);
    auto up = [&](const std::vector<G1Affine<E>>& v, pc_srs** out) {
      return pc_hip_srs_upload(ctx, E::ID, v.data(), v.size(), sizeof(G1Affine<E>), PC_MEM_HOST, out);
    };
    int rc = up(ck.powers_of_g, &ck.srs_powers);
    if (rc == PC_OK) rc = up(ck.powers_of_gamma_g, &ck.srs_gamma);
    if (rc == PC_OK && ck.shifted_powers_of_g) rc = up(*ck.shifted_powers_of_g, &ck.srs_shifted);
    for (auto& kv : ck.shifted_powers_of_gamma_g) { pc_srs* s = nullptr; if (rc == PC_OK && !kv.second.empty()) { rc = up(kv.second, &s); ck.srs_shifted_gamma[kv.first] = s; } }
    if (rc != PC_OK) { ck.release(); Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    return Error();
  }

  // kzg10/mod.rs:424-449 with the key's own bounds
  static Error check_degrees_and_bounds(const SonicCommitterKey<E>& ck, const LabeledPolynomial<E>& p) {
    if (p.degree_bound) {
      const size_t bound = *p.degree_bound;
      Error e; e.kind = Error::UnsupportedDegreeBound; e.a = bound;
      if (!ck.enforced_degree_bounds) return e;
      if (!std::binary_search(ck.enforced_degree_bounds->begin(), ck.enforced_degree_bounds->end(), bound)) return e;
      if (bound < p.polynomial.degree() || bound > ck.max_degree) { e.msg = "IncorrectDegreeBound: " + p.label; return e; }
    }
    return Error();
  }

  // sonic_pc/mod.rs:273-338
  static Error commit(const SonicCommitterKey<E>& ck, const std::vector<LabeledPolynomial<E>>& polynomials, RngCore<E>* rng,
                      std::vector<Commitment<E>>& commitments, std::vector<Randomness<E>>& states) {
    commitments.clear(); states.clear();
    for (const auto& p : polynomials) {
      if (Error e = check_degrees_and_bounds(ck, p)) return e;
      const size_t* hb = p.hiding_bound ? &*p.hiding_bound : nullptr;
      Powers<E> powers = ck.powers();
      if (p.degree_bound) {                                                                    // :312-316
        auto sp = ck.shifted_powers(p.degree_bound);
        if (!sp) { Error e; e.kind = Error::UnsupportedDegreeBound; e.a = *p.degree_bound; return e; }
        powers = *sp;
      }
      Commitment<E> comm; Randomness<E> rand;
      if (Error e = K::commit(powers, p.polynomial, hb, rng, comm, rand)) return e;           // :318
      commitments.push_back(comm); states.push_back(rand);
    }
    return Error();
  }

  // sonic_pc/mod.rs:340-383
  static Error open(const SonicCommitterKey<E>& ck, const std::vector<LabeledPolynomial<E>>& labeled_polynomials, const Fr& point,
                    ChallengeSource<E>& sponge, const std::vector<Randomness<E>>& states, Proof<E>& out) {
    DensePolynomial<E> combined_polynomial;
    Randomness<E> combined_rand = Randomness<E>::empty();
    Fr curr_challenge = sponge.squeeze_challenge();                                           // :357
    for (size_t j = 0; j < labeled_polynomials.size(); j++) {
      if (Error e = check_degrees_and_bounds(ck, labeled_polynomials[j])) return e;
      MarlinKZG10<E>::axpy(combined_polynomial, curr_challenge, labeled_polynomials[j].polynomial);        // :372
      MarlinKZG10<E>::axpy(combined_rand.blinding_polynomial, curr_challenge, states[j].blinding_polynomial);   // :373
      curr_challenge = sponge.squeeze_challenge();                                             // :374
    }
    return K::open(ck.powers(), combined_polynomial, point, combined_rand, out);              // :378
  }
};

}  // namespace pc_host
