// C++ host mirror of InnerProductArgPC's prover hot path (poly-commit/src/ipa_pc/mod.rs), above the C ABI:
//
//   cm_commit                  ipa_pc/mod.rs:54-72   (Pedersen commitment of a scalar vector, hiding term optional)
//   open's halving loop        ipa_pc/mod.rs:664-711 (l_vec, r_vec, final_comm_key, c)
//
//   open (no hiding, no bounds) ipa_pc/mod.rs:475-723 (combination with the sponge's challenges, the random-oracle
//                              challenges of :74-87 / :615-625 / :681-688, h_prime, the loop, Proof)
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
                    IpaProof<E>& proof) {
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
    return open_rounds(ctx, ck.comm_key, combined, point, h_prime, ro, proof);
  }

  // The halving loop of open(): n = comm_key.size() = coeffs.size() = 2^k.
  // Rounds with n <= fixed_key_below keep the resident key and fold per-base factors instead
  // (pc_hip_ipa_key_scalars: the same points, without a latency-bound scalar-multiplication pass per round).
  static Error open_rounds(pc_ctx* ctx, const std::vector<G1Affine<E>>& comm_key, const std::vector<Fr>& coeffs, const Fr& point,
                           const G1Affine<E>& h_prime, IpaChallengeSource<E>& challenges, IpaProof<E>& proof,
                           size_t fixed_key_below = (size_t)1 << 17) {
    size_t n = coeffs.size();
    if (n == 0 || (n & (n - 1)) || comm_key.size() != n) { Error e; e.kind = Error::Backend; e.msg = "ipa: key / coefficient lengths must be one power of two"; return e; }
    proof = IpaProof<E>();
    pc_srs* srs = nullptr; void* cdev = nullptr; void* zdev = nullptr;
    size_t n0 = 0; void* sdev = nullptr; void* aldev = nullptr; void* ardev = nullptr;
    int rc = pc_hip_srs_upload(ctx, E::ID, comm_key.data(), n, sizeof(G1Affine<E>), PC_MEM_HOST, &srs);
    if (rc == PC_OK) rc = pc_hip_malloc(ctx, n * 32, &cdev);
    if (rc == PC_OK) rc = pc_hip_malloc(ctx, n * 32, &zdev);
    if (rc == PC_OK) rc = pc_hip_memcpy_h2d(ctx, cdev, coeffs.data(), n * 32);
    if (rc == PC_OK) rc = pc_hip_fr_powers(ctx, E::ID, point.l, n, zdev);                      // z = (1, point, point^2, ...)   :652-660
    uint64_t hp[2 * E::NQ]; h_prime.to_xy(hp);
    while (rc == PC_OK && n > 1) {
      const size_t h = n / 2;
      char* c = (char*)cdev; char* z = (char*)zdev;
      // l = cm_commit(key_l, coeffs_r) + h' <coeffs_r, z_l>;  r = cm_commit(key_r, coeffs_l) + h' <coeffs_l, z_r>   :666-675
      uint64_t lxy[2 * E::NQ], rxy[2 * E::NQ]; int linf = 0, rinf = 0; pc_job* jl = nullptr; pc_job* jr = nullptr;
      if (!n0 && n <= fixed_key_below) {                                                      // switch: key[0..n0) stays fixed
        n0 = n;
        const Fr one = Fr::one();
        rc = pc_hip_malloc(ctx, n0 * 32, &sdev);
        if (rc == PC_OK) rc = pc_hip_malloc(ctx, n0 * 32, &aldev);
        if (rc == PC_OK) rc = pc_hip_malloc(ctx, n0 * 32, &ardev);
        if (rc == PC_OK) rc = pc_hip_fr_powers(ctx, E::ID, one.l, n0, sdev);                  // s = (1, 1, ...)
        if (rc != PC_OK) break;
      }
      if (n0) {
        rc = pc_hip_ipa_key_scalars(ctx, E::ID, c, n, sdev, n0, nullptr, 0, aldev, ardev);
        if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, 0, aldev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, lxy, &linf, &jl);
        if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, 0, ardev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, rxy, &rinf, &jr);
      } else {
        rc = pc_hip_msm_async(ctx, srs, 0, c + 32 * h, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, lxy, &linf, &jl);
        if (rc == PC_OK) rc = pc_hip_msm_async(ctx, srs, h, c, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, h, rxy, &rinf, &jr);
      }
      Fr ip_l, ip_r;
      if (rc == PC_OK) rc = pc_hip_fr_dot(ctx, E::ID, c + 32 * h, z, h, ip_l.l);
      if (rc == PC_OK) rc = pc_hip_fr_dot(ctx, E::ID, c, z + 32 * h, h, ip_r.l);
      int w1 = jl ? pc_hip_job_wait(ctx, jl) : PC_OK, w2 = jr ? pc_hip_job_wait(ctx, jr) : PC_OK;   // always reap queued jobs
      if (rc == PC_OK) rc = w1 != PC_OK ? w1 : w2;
      if (rc != PC_OK) break;
      G1Affine<E> l = from_out(lxy).add(h_prime.mul(ip_l)), r = from_out(rxy).add(h_prime.mul(ip_r));
      proof.l_vec.push_back(l); proof.r_vec.push_back(r);
      const Fr u = challenges.next(l, r), u_inv = u.inverse();                                 // :681-689
      rc = pc_hip_fr_fold(ctx, E::ID, c, c + 32 * h, h, u_inv.l);                               // coeffs_l += u^-1 coeffs_r   :691-693
      if (rc == PC_OK) rc = pc_hip_fr_fold(ctx, E::ID, z, z + 32 * h, h, u.l);                  // z_l += u z_r                :695-697
      if (rc == PC_OK) rc = n0 ? pc_hip_ipa_key_scalars(ctx, E::ID, nullptr, 0, sdev, n0, u.l, n, nullptr, nullptr)   // the same fold, on the factors
                               : pc_hip_ec_fold(ctx, srs, h, u.l);                              // key_l += u key_r, normalised :699-707
      n = h;
    }
    if (rc == PC_OK) {
      uint64_t kxy[2 * E::NQ]; int kinf = 0;
      rc = n0 ? pc_hip_msm(ctx, srs, 0, sdev, PC_SCALARS_MONTGOMERY, PC_MEM_DEVICE, n0, kxy, &kinf)      // sum_j s_j K0_j
              : pc_hip_srs_read(ctx, srs, 0, 1, kxy);
      if (rc == PC_OK) { proof.final_comm_key = from_out(kxy); rc = pc_hip_memcpy_d2h(ctx, proof.c.l, cdev, 32); }
    }
    pc_hip_free(ctx, cdev); pc_hip_free(ctx, zdev); pc_hip_free(ctx, sdev); pc_hip_free(ctx, aldev); pc_hip_free(ctx, ardev);
    pc_hip_srs_free(srs);
    return rc == PC_OK ? Error() : backend_error(ctx, rc);
  }
};

}  // namespace pc_host
