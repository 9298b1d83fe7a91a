// C++ host mirror of InnerProductArgPC's prover hot path (poly-commit/src/ipa_pc/mod.rs), above the C ABI:
//
//   cm_commit                  ipa_pc/mod.rs:54-72   (Pedersen commitment of a scalar vector, hiding term optional)
//   open's halving loop        ipa_pc/mod.rs:664-711 (l_vec, r_vec, final_comm_key, c)
//
//   open (no hiding, no bounds) ipa_pc/mod.rs:475-723 (combination with the sponge's challenges, the random-oracle
//                              challenges of :74-87 / :615-625 / :681-688, h_prime, the loop, Proof)
//   succinct_check / check     ipa_pc/mod.rs:91-203, 725-773 (no hiding, no bounds): the verifier, whose one large
//                              computation -- cm_commit(comm_key, check_poly.compute_coeffs()) -- runs on the device
//   commit_general / open_general / check_general: the same with hiding and degree bounds (:403-473, :475-723, :116-151);
//                              the rng's draws (row randomness, hiding polynomial) are arguments
//
// The coefficient vector, the powers of the evaluation point and the commitment key stay in HBM for the
// whole proof (pc_hip_malloc / the resident SRS); per round two points come down and one challenge goes up.
// The Fiat-Shamir hash producing the round challenge (compute_random_oracle_challenge: Blake2s over
// ark-serialize bytes of the transcript) is host work on two points: transcript.hpp.  open_rounds() takes
// the challenges through IpaChallengeSource (tests substitute a fixed stream); open() plugs in the
// reference's transcript.  The opening challenges of the combination come from the CALLER's sponge
// (sponge.squeeze_field_elements_with_sizes, :502/:525/:556) and are passed in.
#pragma once
#include "kzg10.hpp"
#include "transcript.hpp"
#include <algorithm>

namespace pc_host {

template <class E>
struct IpaProof {                                   // ipa_pc/data_structures.rs:161-185 (the fields this loop fills)
  std::vector<G1Affine<E>> l_vec, r_vec;
  G1Affine<E> final_comm_key = G1Affine<E>::zero();
  FrT<E> c = FrT<E>::zero();
};

template <class E>
struct IpaChallengeSource { virtual ~IpaChallengeSource() {} virtual FrT<E> next(const G1Affine<E>& l, const G1Affine<E>& r) = 0; };

// The reference's round challenges: round_challenge = RO(ser(round_challenge) || ser(l) || ser(r)), ipa_pc/mod.rs:681-688
template <class E>
struct IpaRandomOracle : IpaChallengeSource<E> {
  FrT<E> round_challenge;
  explicit IpaRandomOracle(const FrT<E>& first) : round_challenge(first) {}
  FrT<E> next(const G1Affine<E>& l, const G1Affine<E>& r) override {
    Transcript<E> t; t.append(round_challenge); t.append(l); t.append(r);
    round_challenge = t.challenge();
    return round_challenge;
  }
};

template <class E>
struct IpaCommitterKey {                            // ipa_pc/data_structures.rs:38-57
  std::vector<G1Affine<E>> comm_key;                // supported_degree + 1 generators, a power of two
  G1Affine<E> h = G1Affine<E>::zero(), s = G1Affine<E>::zero();
  size_t max_degree = 0;
  size_t supported_degree() const { return comm_key.size() - 1; }
};

template <class E>
struct InnerProductArgPC {
  typedef FrT<E> Fr;
  static Error backend_error(pc_ctx* ctx, int rc) {
    Error e; e.kind = Error::Backend; e.msg = std::string(pc_hip_strerror(rc)) + ": " + pc_hip_last_error(ctx); return e;
  }
  static G1Affine<E> from_out(const uint64_t* xy) {
    bool inf = true; for (int i = 0; i < 2 * E::NQ; i++) inf &= xy[i] == 0;
    return G1Affine<E>::from_xy(xy, inf);
  }

  // cm_commit(comm_key, scalars, hiding_generator, randomizer): MSM + optional h * r  (ipa_pc/mod.rs:54-72)
  static Error cm_commit(pc_ctx* ctx, const std::vector<G1Affine<E>>& comm_key, const std::vector<Fr>& scalars,
                         const G1Affine<E>* hiding_generator, const Fr* randomizer, G1Affine<E>& out) {
    pc_srs* srs = nullptr;
    int rc = pc_hip_srs_upload(ctx, E::ID, comm_key.data(), comm_key.size(), sizeof(G1Affine<E>), PC_MEM_HOST, &srs);
    if (rc != PC_OK) return backend_error(ctx, rc);
    uint64_t xy[2 * E::NQ]; int inf = 0;
    rc = pc_hip_msm(ctx, srs, 0, scalars.data(), PC_SCALARS_MONTGOMERY, PC_MEM_HOST, scalars.size(), xy, &inf);
    pc_hip_srs_free(srs);
    if (rc != PC_OK) return backend_error(ctx, rc);
    out = from_out(xy);
    if (randomizer) { if (!hiding_generator) { Error e; e.kind = Error::MissingRng; return e; } out = out.add(hiding_generator->mul(*randomizer)); }
    return Error();
  }

  // open() for polynomials without hiding and degree bounds (ipa_pc/mod.rs:475-723 with has_hiding == false):
  //   combined_polynomial = sum_j xi_j p_j, combined_commitment = sum_j xi_j C_j          :501-557
  //   combined_v = combined_polynomial(point)                                             :561
  //   round_challenge = RO(ser(combined_commitment) || ser(point) || ser(combined_v))     :615-623
  //   h_prime = h * round_challenge                                                       :625
  //   pad to d + 1 coefficients, the halving loop                                         :627-711
  // opening_challenges: what the caller's sponge squeezed, one per polynomial (:502, then :525/:556).
  static Error open(pc_ctx* ctx, const IpaCommitterKey<E>& ck, const std::vector<const DensePolynomial<E>*>& polynomials,
                    const std::vector<G1Affine<E>>& commitments, const Fr& point, const std::vector<Fr>& opening_challenges,
                    IpaProof<E>& proof, size_t fixed_key_below = (size_t)1 << 16, bool two_level_table = false, bool one_call = false) {
    const size_t d1 = ck.comm_key.size();
    if (polynomials.size() != commitments.size() || polynomials.size() != opening_challenges.size()) { Error e; e.kind = Error::Backend; e.msg = "ipa open: one commitment and one opening challenge per polynomial"; return e; }
    std::vector<Fr> combined(d1, Fr::zero());
    G1Affine<E> combined_commitment = G1Affine<E>::zero();
    std::vector<const void*> ptrs; std::vector<size_t> lens;
    for (size_t j = 0; j < polynomials.size(); j++) {
      const DensePolynomial<E>& p = *polynomials[j];
      if (Error e = KZG10<E>::check_degree_is_too_large(p.degree(), d1)) return e;           // check_degrees_and_bounds, :517
      ptrs.push_back(p.coeffs.data()); lens.push_back(std::min(p.coeffs.size(), d1));
      combined_commitment = combined_commitment.add(commitments[j].mul(opening_challenges[j]));
    }
    if (!polynomials.empty()) {
      int rc = pc_hip_fr_lincomb(ctx, E::ID, ptrs.data(), PC_MEM_HOST, lens.data(), ptrs.size(), opening_challenges.data(), combined.data(), PC_MEM_HOST, d1);
      if (rc != PC_OK) return backend_error(ctx, rc);
    }
    DensePolynomial<E> cp; cp.coeffs = combined;
    const Fr combined_v = cp.evaluate(point);
    Transcript<E> t; t.append(combined_commitment); t.append(point); t.append(combined_v);
    const Fr round_challenge = t.challenge();
    const G1Affine<E> h_prime = ck.h.mul(round_challenge);
    IpaRandomOracle<E> ro(round_challenge);
    if (one_call) return open_rounds_one_call(ctx, ck.comm_key, combined, point, h_prime, ro, proof, fixed_key_below);
    return open_rounds(ctx, ck.comm_key, combined, point, h_prime, ro, proof, fixed_key_below, two_level_table);
  }

  // The halving loop as ONE library call (pc_hip_ipa_open_rounds; what the Rust shim's open does): the transcript goes in as a callback,
  // the library picks fold table / ladder / fixed key -- and keeps the fixed key as a key object with its own window table, which the
  // loop over the single entry points below cannot express.  Same proof, bit for bit.
  static void challenge_trampoline(void* user, const void* l_xy, const void* r_xy, void* out_u_mont) {
    IpaChallengeSource<E>* src = (IpaChallengeSource<E>*)user;
    const Fr u = src->next(from_out((const uint64_t*)l_xy), from_out((const uint64_t*)r_xy));
    memcpy(out_u_mont, u.l, 32);
  }
  static Error open_rounds_one_call(pc_ctx* ctx, const std::vector<G1Affine<E>>& comm_key, const std::vector<Fr>& coeffs, const Fr& point,
                                    const G1Affine<E>& h_prime, IpaChallengeSource<E>& challenges, IpaProof<E>& proof,
                                    size_t fixed_key_below = 0 /* the library's default: 2^17 */) {
    const size_t n = coeffs.size();
    if (n == 0 || (n & (n - 1)) || comm_key.size() != n) { Error e; e.kind = Error::Backend; e.msg = "ipa: key / coefficient lengths must be one power of two"; return e; }
    proof = IpaProof<E>();
    size_t rounds = 0; while (((size_t)1 << rounds) < n) rounds++;
    pc_srs* srs = nullptr; void* cdev = nullptr;
    int rc = pc_hip_srs_upload(ctx, E::ID, comm_key.data(), n, sizeof(G1Affine<E>), PC_MEM_HOST, &srs);
    if (rc == PC_OK) rc = pc_hip_srs_precompute(ctx, srs, 0, 1);                               // per call here (the key is uploaded per call); once per key for a prover
    if (rc == PC_OK && n >= 2) { rc = pc_hip_srs_precompute_fold(ctx, srs); if (rc == PC_ERR_UNSUPPORTED) rc = PC_OK; }
    if (rc == PC_OK) rc = pc_hip_malloc(ctx, n * 32, &cdev);
    if (rc == PC_OK) rc = pc_hip_memcpy_h2d(ctx, cdev, coeffs.data(), n * 32);
    uint64_t hp[2 * E::NQ]; h_prime.to_xy(hp);
    std::vector<uint64_t> lxy((rounds ? rounds : 1) * 2 * E::NQ), rxy((rounds ? rounds : 1) * 2 * E::NQ);
    uint64_t kxy[2 * E::NQ];
    if (rc == PC_OK) rc = pc_hip_ipa_open_rounds(ctx, srs, cdev, n, point.l, hp, &challenge_trampoline, &challenges, fixed_key_below,
                                                 lxy.data(), rxy.data(), kxy, proof.c.l, nullptr, nullptr);
    if (rc == PC_OK) {
      for (size_t k = 0; k < rounds; k++) {
        proof.l_vec.push_back(from_out(lxy.data() + k * 2 * E::NQ));
        proof.r_vec.push_back(from_out(rxy.data() + k * 2 * E::NQ));
      }
      proof.final_comm_key = from_out(kxy);
    }
    pc_hip_free(ctx, cdev);
    pc_hip_srs_free(srs);
    return rc == PC_OK ? Error() : backend_error(ctx, rc);
  }

  // The halving loop of open(): n = comm_key.size() = coeffs.size() = 2^k.
  // Rounds with n <= fixed_key_below keep the resident key and fold per-base factors instead
  // (pc_hip_ipa_key_scalars: the same points, without a latency-bound scalar-multiplication pass per round).
  static Error open_rounds(pc_ctx* ctx, const std::vector<G1Affine<E>>& comm_key, const std::vector<Fr>& coeffs, const Fr& point,
                           const G1Affine<E>& h_prime, IpaChallengeSource<E>& challenges, IpaProof<E>& proof,
                           size_t fixed_key_below = (size_t)1 << 16, bool two_level_table = false) {
    size_t n = coeffs.size();
    if (n == 0 || (n & (n - 1)) || comm_key.size() != n) { Error e; e.kind = Error::Backend; e.msg = "ipa: key / coefficient lengths must be one power of two"; return e; }
    proof = IpaProof<E>();
    pc_srs* srs = nullptr; void* cdev = nullptr; void* zdev = nullptr;
    size_t n0 = 0; void* sdev = nullptr; void* alrdev = nullptr;      // alrdev: scalars of l | scalars of r (2 n0 elements)
    int rc = pc_hip_srs_upload(ctx, E::ID, comm_key.data(), n, sizeof(G1Affine<E>), PC_MEM_HOST, &srs);
    if (rc == PC_OK) rc = pc_hip_malloc(ctx, n * 32, &cdev);
    if (rc == PC_OK) rc = pc_hip_malloc(ctx, n * 32, &zdev);
    if (rc == PC_OK) rc = pc_hip_memcpy_h2d(ctx, cdev, coeffs.data(), n * 32);
    if (rc == PC_OK) rc = pc_hip_fr_powers(ctx, E::ID, point.l, n, zdev);                      // z = (1, point, point^2, ...)   :652-660
    uint64_t hp[2 * E::NQ]; h_prime.to_xy(hp);
    // the inner products of the first round; every later round gets its pair from the pass that folds the vectors
    Fr dots[2];
    if (rc == PC_OK) rc = pc_hip_ipa_fold_dots(ctx, E::ID, cdev, zdev, n, nullptr, nullptr, dots);
    Fr u_prev = Fr::zero(); bool have_u_prev = false;
    // two_level_table: what a prover holding a resident committer key does (the Rust shim's loop): the key's two-level fold table
    // (pc_hip_srs_precompute_fold_ex, once per key -- here per call, the key being uploaded per call) serves rounds 1 and 2: round 1
    // leaves the key alone, round 2's commitments run on the committer key by linearity (pc_hip_ipa_round2_msms), the key after both
    // folds comes out of the table in one step (pc_hip_ec_fold2_from).
    bool two_level = two_level_table && n >= 8 && n / 2 > fixed_key_below;
    if (rc == PC_OK && two_level) {
      rc = pc_hip_srs_precompute(ctx, srs, 0, 1);
      if (rc == PC_OK) rc = pc_hip_srs_precompute_fold_ex(ctx, srs, 2, 0);
      if (rc == PC_ERR_UNSUPPORTED) { two_level = false; rc = PC_OK; }                         // no memory for it: round by round
    }
    pc_srs* root = srs; Fr u_first = Fr::zero(); bool have_u_first = false;
    while (rc == PC_OK && n > 1) {
      const size_t h = n / 2;
      char* c = (char*)cdev; char* z = (char*)zdev;
      // l = cm_commit(key_l, coeffs_r) + h' <coeffs_r, z_l>;  r = cm_commit(key_r, coeffs_l) + h' <coeffs_l, z_r>   :666-675
      uint64_t lrxy[2][2 * E::NQ]; int lrinf[2] = {0, 0}; pc_job* jl = nullptr; pc_job* jr = nullptr;
      if (!n0 && n <= fixed_key_below) {                                                      // switch: key[0..n0) stays fixed
        n0 = n;
        const Fr one = Fr::one();
        rc = pc_hip_malloc(ctx, n0 * 32, &sdev);
        if (rc == PC_OK) rc = pc_hip_malloc(ctx, 2 * n0 * 32, &alrdev);
        if (rc == PC_OK) rc = pc_hip_fr_powers(ctx, E::ID, one.l, n0, sdev);                  // s = (1, 1, ...)
        if (rc != PC_OK) break;
        have_u_prev = false;                                                                  // the key itself carries every fold so far
      }
      if (n0) {
        // the fold of the factors by the previous challenge (size 2n) and this round's scalar vectors in one call, then the two
        // commitments on two pipelines of the fixed key (one two-row pc_hip_msm_many pass measured slower: 1.40 vs 1.20 ms per round)
        rc = pc_hip_ipa_key_scalars(ctx, E::ID, c, n, sdev, n0, have_u_prev ? u_prev.l : nullptr, have_u_prev ? 2 * n : 0, alrdev, (char*)alrdev + 32 * n0);
        if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, 0, alrdev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, lrxy[0], &lrinf[0], &jl);
        if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, 0, (char*)alrdev + 32 * n0, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, lrxy[1], &lrinf[1], &jr);
      } else if (have_u_first) {
        rc = pc_hip_ipa_round2_msms(ctx, srs, c, h, u_first.l, lrxy[0], &lrinf[0], lrxy[1], &lrinf[1]);   // round 2 on the committer key
      } else {
        rc = pc_hip_msm_async(ctx, srs, 0, c + 32 * h, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, lrxy[0], &lrinf[0], &jl);
        if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, h, c, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, lrxy[1], &lrinf[1], &jr);
      }
      int w1 = jl ? pc_hip_job_wait(ctx, jl) : PC_OK, w2 = jr ? pc_hip_job_wait(ctx, jr) : PC_OK;   // always reap queued jobs
      if (rc == PC_OK) rc = w1 != PC_OK ? w1 : w2;
      if (rc != PC_OK) break;
      G1Affine<E> l = from_out(lrxy[0]).add(h_prime.mul(dots[0])), r = from_out(lrxy[1]).add(h_prime.mul(dots[1]));
      proof.l_vec.push_back(l); proof.r_vec.push_back(r);
      const Fr u = challenges.next(l, r), u_inv = u.inverse();                                 // :681-689
      // coeffs_l += u^-1 coeffs_r, z_l += u z_r (:691-697) and the next round's two inner products in the same pass
      rc = pc_hip_ipa_fold_dots(ctx, E::ID, c, z, h, u.l, u_inv.l, dots);
      if (rc == PC_OK) {
        if (n0) { u_prev = u; have_u_prev = true; }                                            // applied to the factors at the top of the next round
        else if (two_level && !have_u_first && srs == root) { u_first = u; have_u_first = true; }   // round 1: the key stays
        else if (have_u_first) {                                                               // round 2: both folds out of the table
          pc_srs* work = nullptr;
          rc = pc_hip_ec_fold2_from(ctx, root, h, u_first.l, u.l, &work);
          if (rc == PC_OK) srs = work;
          have_u_first = false; two_level = false;
        }
        else rc = pc_hip_ec_fold(ctx, srs, h, u.l);                                            // key_l += u key_r, normalised :699-707
      }
      n = h;
    }
    if (rc == PC_OK && n0 && have_u_prev) rc = pc_hip_ipa_key_scalars(ctx, E::ID, nullptr, 0, sdev, n0, u_prev.l, 2, nullptr, nullptr);   // the last fold (size 2)
    if (rc == PC_OK) {
      uint64_t kxy[2 * E::NQ]; int kinf = 0;
      rc = n0 ? pc_hip_msm(ctx, srs, 0, sdev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, kxy, &kinf)      // sum_j s_j K0_j
              : pc_hip_srs_read(ctx, srs, 0, 1, kxy);
      if (rc == PC_OK) { proof.final_comm_key = from_out(kxy); rc = pc_hip_memcpy_d2h(ctx, proof.c.l, cdev, 32); }
    }
    pc_hip_free(ctx, cdev); pc_hip_free(ctx, zdev); pc_hip_free(ctx, sdev); pc_hip_free(ctx, alrdev);
    if (srs != root) pc_hip_srs_free(srs);                                                     // the working key of the two-level path
    pc_hip_srs_free(root);
    return rc == PC_OK ? Error() : backend_error(ctx, rc);
  }

  // ---- the general form: hiding and degree bounds -------------------------------------------------------------------
  // One labeled polynomial with its commitment and commitment state (LabeledPolynomial + LabeledCommitment<Commitment> +
  // Randomness, ipa_pc/data_structures.rs:94-160).  degree_bound < 0: none.  hiding == false: Randomness::empty().
  struct Labeled {
    const DensePolynomial<E>* polynomial = nullptr;
    long degree_bound = -1;
    bool hiding = false;
    Fr rand = Fr::zero(), shifted_rand = Fr::zero();
    G1Affine<E> comm = G1Affine<E>::zero(), shifted_comm = G1Affine<E>::zero();
  };
  struct GeneralProof {                              // Proof { l_vec, r_vec, final_comm_key, c, hiding_comm, rand }
    IpaProof<E> core;
    bool has_hiding = false;
    G1Affine<E> hiding_comm = G1Affine<E>::zero();
    Fr rand = Fr::zero();
  };
  static Error incorrect_degree_bound(size_t deg, long bound) { Error e; e.kind = Error::UnsupportedDegreeBound; e.a = deg; e.b = (size_t)bound; return e; }

  // commit for one polynomial (ipa_pc/mod.rs:403-473): fills comm / shifted_comm of `lp`
  static Error commit_general(pc_ctx* ctx, const IpaCommitterKey<E>& ck, Labeled& lp) {
    const DensePolynomial<E>& p = *lp.polynomial;
    const size_t d = ck.supported_degree();
    if (Error e = KZG10<E>::check_degree_is_too_large(p.degree(), d + 1)) return e;
    if (lp.degree_bound >= 0 && ((size_t)lp.degree_bound < p.degree() || (size_t)lp.degree_bound > d)) return incorrect_degree_bound(p.degree(), lp.degree_bound);
    const size_t m = p.degree() + 1;
    std::vector<Fr> co(p.coeffs.begin(), p.coeffs.begin() + std::min(m, p.coeffs.size()));
    std::vector<G1Affine<E>> key(ck.comm_key.begin(), ck.comm_key.begin() + co.size());
    if (Error e = cm_commit(ctx, key, co, &ck.s, lp.hiding ? &lp.rand : nullptr, lp.comm)) return e;
    if (lp.degree_bound >= 0) {
      std::vector<G1Affine<E>> skey(ck.comm_key.begin() + (d - (size_t)lp.degree_bound), ck.comm_key.begin() + (d - (size_t)lp.degree_bound) + co.size());
      if (Error e = cm_commit(ctx, skey, co, &ck.s, lp.hiding ? &lp.shifted_rand : nullptr, lp.shifted_comm)) return e;
    }
    return Error();
  }

  // open (ipa_pc/mod.rs:475-723).  challenges: the caller's sponge output in squeeze order (:502, then :525 and :556 per
  // polynomial); hiding_polynomial (d + 1 coefficients) / hiding_rand: what the reference draws at :577 / :579.
  static Error open_general(pc_ctx* ctx, const IpaCommitterKey<E>& ck, const std::vector<Labeled>& polys, const Fr& point,
                            const std::vector<Fr>& challenges, const std::vector<Fr>* hiding_polynomial, const Fr* hiding_rand,
                            GeneralProof& proof) {
    const size_t d1 = ck.comm_key.size(), d = d1 - 1;
    if (challenges.size() < 2 * polys.size() + 1) { Error e; e.kind = Error::Backend; e.msg = "ipa open: 2 k + 1 sponge challenges expected"; return e; }
    std::vector<std::vector<Fr>> shifted;                 shifted.reserve(polys.size());
    std::vector<const void*> ptrs; std::vector<size_t> lens; std::vector<Fr> xis;
    G1Affine<E> combined_commitment = G1Affine<E>::zero();
    Fr combined_rand = Fr::zero();
    bool has_hiding = false;
    size_t ci = 0;
    Fr cur = challenges[ci++];
    for (const Labeled& lp : polys) {
      const DensePolynomial<E>& p = *lp.polynomial;
      if (Error e = KZG10<E>::check_degree_is_too_large(p.degree(), d1)) return e;
      if (lp.degree_bound >= 0 && ((size_t)lp.degree_bound < p.degree() || (size_t)lp.degree_bound > d)) return incorrect_degree_bound(p.degree(), lp.degree_bound);
      ptrs.push_back(p.coeffs.data()); lens.push_back(std::min(p.coeffs.size(), d1)); xis.push_back(cur);
      combined_commitment = combined_commitment.add(lp.comm.mul(cur));
      if (lp.hiding) { has_hiding = true; combined_rand = combined_rand + cur * lp.rand; }
      cur = challenges[ci++];
      if (lp.degree_bound >= 0) {
        shifted.emplace_back(d - (size_t)lp.degree_bound, Fr::zero());                               // shift_polynomial, :230-239
        shifted.back().insert(shifted.back().end(), p.coeffs.begin(), p.coeffs.end());
        ptrs.push_back(shifted.back().data()); lens.push_back(std::min(shifted.back().size(), d1)); xis.push_back(cur);
        combined_commitment = combined_commitment.add(lp.shifted_comm.mul(cur));
        if (lp.hiding) combined_rand = combined_rand + cur * lp.shifted_rand;
      }
      cur = challenges[ci++];
    }
    std::vector<Fr> combined(d1, Fr::zero());
    if (!ptrs.empty()) {
      int rc = pc_hip_fr_lincomb(ctx, E::ID, ptrs.data(), PC_MEM_HOST, lens.data(), ptrs.size(), xis.data(), combined.data(), PC_MEM_HOST, d1);
      if (rc != PC_OK) return backend_error(ctx, rc);
    }
    DensePolynomial<E> cp; cp.coeffs = combined;
    const Fr combined_v = cp.evaluate(point);
    proof = GeneralProof();
    if (has_hiding) {
      if (!hiding_polynomial || !hiding_rand) { Error e; e.kind = Error::MissingRng; return e; }
      std::vector<Fr> hp(*hiding_polynomial); hp.resize(d1, Fr::zero());
      DensePolynomial<E> hpp; hpp.coeffs = hp;
      hp[0] = hp[0] - hpp.evaluate(point);                                                              // :578
      G1Affine<E> hiding_comm;
      if (Error e = cm_commit(ctx, ck.comm_key, hp, &ck.s, hiding_rand, hiding_comm)) return e;        // :580-585
      Transcript<E> t; t.append(combined_commitment); t.append(point); t.append(combined_v); t.append(hiding_comm);
      const Fr hc = t.challenge();                                                                       // :592-603
      for (size_t i = 0; i < d1; i++) combined[i] = combined[i] + hc * hp[i];
      combined_rand = combined_rand + hc * *hiding_rand;
      combined_commitment = combined_commitment.add(hiding_comm.mul(hc)).add(ck.s.mul(combined_rand).neg());   // :606-607
      proof.has_hiding = true; proof.hiding_comm = hiding_comm; proof.rand = combined_rand;
    }
    Transcript<E> t; t.append(combined_commitment); t.append(point); t.append(combined_v);
    const Fr round_challenge = t.challenge();
    const G1Affine<E> h_prime = ck.h.mul(round_challenge);
    IpaRandomOracle<E> ro(round_challenge);
    return open_rounds(ctx, ck.comm_key, combined, point, h_prime, ro, proof.core);
  }

  // check with hiding and degree bounds: the combination of succinct_check (:116-151), then the plain check of the one
  // combined commitment
  static Error check_general(pc_ctx* ctx, const IpaCommitterKey<E>& vk, const std::vector<Labeled>& comms, const Fr& point,
                             const std::vector<Fr>& values, const GeneralProof& proof, const std::vector<Fr>& challenges, bool& ok) {
    ok = false;
    const size_t d = vk.supported_degree();
    if (challenges.size() < 2 * comms.size() + 1 || values.size() != comms.size()) { Error e; e.kind = Error::Backend; e.msg = "ipa check: 2 k + 1 sponge challenges and k values expected"; return e; }
    G1Affine<E> combined_commitment = G1Affine<E>::zero();
    Fr combined_v = Fr::zero();
    size_t ci = 0;
    Fr cur = challenges[ci++];
    for (size_t j = 0; j < comms.size(); j++) {
      combined_v = combined_v + cur * values[j];
      combined_commitment = combined_commitment.add(comms[j].comm.mul(cur));
      cur = challenges[ci++];
      if (comms[j].degree_bound >= 0) {
        Fr shift = Fr::one(), b = point;                                                           // point^(d - bound), :128
        for (size_t e = d - (size_t)comms[j].degree_bound; e; e >>= 1) { if (e & 1) shift = shift * b; b = b * b; }
        combined_v = combined_v + cur * values[j] * shift;
        combined_commitment = combined_commitment.add(comms[j].shifted_comm.mul(cur));
      }
      cur = challenges[ci++];
    }
    if (proof.has_hiding) {                                                                        // :138-151
      Transcript<E> t; t.append(combined_commitment); t.append(point); t.append(combined_v); t.append(proof.hiding_comm);
      const Fr hc = t.challenge();
      combined_commitment = combined_commitment.add(proof.hiding_comm.mul(hc)).add(vk.s.mul(proof.rand).neg());
    }
    return check(ctx, vk, {combined_commitment}, point, {combined_v}, proof.core, {Fr::one()}, ok);
  }

  // SuccinctCheckPolynomial::evaluate (ipa_pc/data_structures.rs:223-236): prod_i (1 + u_i * point^(2^(log_d - i)))
  static Fr check_poly_evaluate(const std::vector<Fr>& challenges, const Fr& point) {
    const size_t log_d = challenges.size();
    std::vector<Fr> pw(log_d ? log_d : 1); pw[0] = point;                       // pw[k] = point^(2^k)
    for (size_t k = 1; k < log_d; k++) pw[k] = pw[k - 1] * pw[k - 1];
    Fr product = Fr::one();
    for (size_t i = 1; i <= log_d; i++) product = product * (Fr::one() + pw[log_d - i] * challenges[i - 1]);
    return product;
  }

  // succinct_check without hiding / degree bounds (ipa_pc/mod.rs:91-203): O(log d) group operations on the host.
  // Returns false when the equation fails; on success `round_challenges` holds the check polynomial.
  static bool succinct_check(const G1Affine<E>& h, const std::vector<G1Affine<E>>& commitments, const Fr& point,
                             const std::vector<Fr>& values, const IpaProof<E>& proof, const std::vector<Fr>& opening_challenges,
                             std::vector<Fr>& round_challenges) {
    G1Affine<E> combined_commitment = G1Affine<E>::zero();
    Fr combined_v = Fr::zero();
    for (size_t j = 0; j < commitments.size(); j++) {                                             // :116-131
      combined_v = combined_v + opening_challenges[j] * values[j];
      combined_commitment = combined_commitment.add(commitments[j].mul(opening_challenges[j]));
    }
    Transcript<E> t; t.append(combined_commitment); t.append(point); t.append(combined_v);       // :153-161
    Fr round_challenge = t.challenge();
    const G1Affine<E> h_prime = h.mul(round_challenge);                                           // :163
    G1Affine<E> round_commitment = combined_commitment.add(h_prime.mul(combined_v));              // :165
    round_challenges.clear();
    IpaRandomOracle<E> ro(round_challenge);
    for (size_t k = 0; k < proof.l_vec.size(); k++) {                                             // :170-184
      round_challenge = ro.next(proof.l_vec[k], proof.r_vec[k]);
      round_challenges.push_back(round_challenge);
      round_commitment = round_commitment.add(proof.l_vec[k].mul(round_challenge.inverse())).add(proof.r_vec[k].mul(round_challenge));
    }
    const Fr v_prime = check_poly_evaluate(round_challenges, point) * proof.c;                    // :187
    const G1Affine<E> check_commitment_elem = proof.final_comm_key.mul(proof.c).add(h_prime.mul(v_prime));   // :190-195
    return round_commitment == check_commitment_elem;
  }

  // check (ipa_pc/mod.rs:725-773): succinct_check, then final_comm_key == cm_commit(comm_key, check_poly.compute_coeffs()).
  // The 2^log_d coefficients (data_structures.rs:204-220) are built on the device: they are the factors the prover's key
  // folds apply to the bases (pc_hip_ipa_key_scalars folding a vector of ones by every challenge), so the verifier's one
  // large computation is a resident vector and one MSM.
  static Error check(pc_ctx* ctx, const IpaCommitterKey<E>& vk, const std::vector<G1Affine<E>>& commitments, const Fr& point,
                     const std::vector<Fr>& values, const IpaProof<E>& proof, const std::vector<Fr>& opening_challenges, bool& ok) {
    ok = false;
    const size_t n = vk.comm_key.size();
    size_t log_d = 0; while (((size_t)1 << log_d) < n) log_d++;                                   // ark_std::log2(d + 1)
    if (proof.l_vec.size() != proof.r_vec.size() || proof.l_vec.size() != log_d) {
      Error e; e.kind = Error::IncorrectInputLength; e.a = log_d; e.b = proof.l_vec.size();
      e.msg = "Expected proof vectors to be " + std::to_string(log_d) + ". Instead, l_vec size is " + std::to_string(proof.l_vec.size()) +
              " and r_vec size is " + std::to_string(proof.r_vec.size());
      return e;
    }
    if (commitments.size() != values.size() || commitments.size() != opening_challenges.size()) { Error e; e.kind = Error::Backend; e.msg = "ipa check: one value and one opening challenge per commitment"; return e; }
    std::vector<Fr> ch;
    if (!succinct_check(vk.h, commitments, point, values, proof, opening_challenges, ch)) return Error();
    pc_srs* srs = nullptr; void* sdev = nullptr;
    int rc = pc_hip_srs_upload(ctx, E::ID, vk.comm_key.data(), n, sizeof(G1Affine<E>), PC_MEM_HOST, &srs);
    if (rc == PC_OK) rc = pc_hip_malloc(ctx, n * 32, &sdev);
    const Fr one = Fr::one();
    if (rc == PC_OK) rc = pc_hip_fr_powers(ctx, E::ID, one.l, n, sdev);
    size_t m = n;
    for (size_t i = 0; rc == PC_OK && i < ch.size(); i++, m /= 2)
      rc = pc_hip_ipa_key_scalars(ctx, E::ID, nullptr, 0, sdev, n, ch[i].l, m, nullptr, nullptr);
    uint64_t kxy[2 * E::NQ]; int kinf = 0;
    if (rc == PC_OK) rc = pc_hip_msm(ctx, srs, 0, sdev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n, kxy, &kinf);
    pc_hip_free(ctx, sdev); pc_hip_srs_free(srs);
    if (rc != PC_OK) return backend_error(ctx, rc);
    ok = from_out(kxy) == proof.final_comm_key;
    return Error();
  }
};

}  // namespace pc_host
