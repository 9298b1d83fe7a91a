// C++ host mirror of the reference's KZG10 commit/open glue, above the C ABI (include/pc_hip.h).
//
// The reference's host code is Rust; with no Rust toolchain in this image the same glue is
// restated here in C++ with the reference's names, argument meaning and error behaviour, so
// that tests/cpp/*.cpp read like the reference's own tests (kzg10/mod.rs:519-674).  The Rust
// shim a maintainer would ship is in INTEGRATION.md; both are thin: every MSM goes to
// pc_hip_msm, the division to pc_hip_witness_poly.
//
//   pc_host::KZG10<CurveTag>::commit                         poly-commit/src/kzg10/mod.rs:157-210
//   ...::compute_witness_polynomial                          :217-240
//   ...::open_with_witness_polynomial / open                 :243-310
//   check_degree_is_too_large / check_hiding_bound           :393-421
//   skip_leading_zeros_and_convert_to_bigints                :452-461 (conversion fused on the device)
//   Randomness::{empty, rand, is_hiding}                     kzg10/data_structures.rs:400-435
//   Powers / Commitment / Proof                              kzg10/data_structures.rs:124-136, 325-328, 489-495
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <utility>
#include <vector>
#include "../../include/pc_hip.h"
#include "../csrc/host_tail.hpp"

namespace pc_host {

struct Bls12_381 { typedef pc_curve_bls12_381 C; static constexpr pc_curve ID = PC_CURVE_BLS12_381; static constexpr int NQ = 6; };
struct Bn254 { typedef pc_curve_bn254 C; static constexpr pc_curve ID = PC_CURVE_BN254; static constexpr int NQ = 4; };
struct Pallas { typedef pc_curve_pallas C; static constexpr pc_curve ID = PC_CURVE_PALLAS; static constexpr int NQ = 4; };

// Error variants of poly-commit/src/error.rs that this path can raise.
struct Error {
  enum Kind { None, MissingRng, TooManyCoefficients, HidingBoundIsZero, HidingBoundToolarge, UnsupportedDegreeBound, InvalidParameters, IncorrectInputLength, InvalidNumberOfVariables, IncorrectCommitmentSize, InvalidCommitment, TrimmingDegreeTooLarge, MissingPolynomial, EquationHasDegreeBounds, Backend } kind = None;
  size_t a = 0, b = 0;      // (num_coefficients, num_powers) / (hiding_poly_degree, num_powers) / bound
  std::string msg;          // Backend: pc_hip_strerror / pc_hip_last_error
  explicit operator bool() const { return kind != None; }
};

// Scalar-field element as arkworks holds it: Montgomery, 4 x u64 LE.
template <class E>
struct FrT {
  typedef pc::host64::F64<typename E::C::FrP> F;
  uint64_t l[4];
  static FrT zero() { FrT r; memset(r.l, 0, 32); return r; }
  static FrT one() { FrT r; F o = F::one(); memcpy(r.l, o.l, 32); return r; }
  static FrT from_u64(uint64_t v) {   // canonical small integer -> Montgomery
    F t = F::zero(); t.l[0] = v; F r2; memcpy(r2.l, E::C::FrP::R2, 32); F m = t.mul(r2); FrT r; memcpy(r.l, m.l, 32); return r;
  }
  F f() const { F t; memcpy(t.l, l, 32); return t; }
  static FrT of(const F& t) { FrT r; memcpy(r.l, t.l, 32); return r; }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const FrT& o) const { return memcmp(l, o.l, 32) == 0; }
  FrT operator+(const FrT& o) const { return of(f().add(o.f())); }
  FrT operator-(const FrT& o) const { return of(f().sub(o.f())); }
  FrT operator*(const FrT& o) const { return of(f().mul(o.f())); }
  FrT neg() const { return zero() - *this; }
  FrT inverse() const { return of(f().inv()); }     // 0 -> 0
};

// Same memory layout as arkworks' short_weierstrass::Affine { x, y, infinity } (104 / 72 bytes).
template <class E>
struct G1Affine {
  uint64_t x[E::NQ], y[E::NQ];
  bool infinity = true;
  static G1Affine zero() { G1Affine a; memset(a.x, 0, sizeof(a.x)); memset(a.y, 0, sizeof(a.y)); a.infinity = true; return a; }
  bool is_zero() const { return infinity; }
  bool operator==(const G1Affine& o) const {
    if (infinity || o.infinity) return infinity == o.infinity;
    return memcmp(x, o.x, sizeof(x)) == 0 && memcmp(y, o.y, sizeof(y)) == 0;
  }
  void to_xy(uint64_t* out) const { if (infinity) memset(out, 0, 16 * E::NQ); else { memcpy(out, x, 8 * E::NQ); memcpy(out + E::NQ, y, 8 * E::NQ); } }
  static G1Affine from_xy(const uint64_t* in, bool inf) {
    G1Affine a = zero(); if (!inf) { memcpy(a.x, in, 8 * E::NQ); memcpy(a.y, in + E::NQ, 8 * E::NQ); a.infinity = false; } return a;
  }
  // group operations needed by the glue (a handful of points; host, like the reference)
  G1Affine add(const G1Affine& o) const {
    uint64_t pts[4 * E::NQ], out[2 * E::NQ]; to_xy(pts); o.to_xy(pts + 2 * E::NQ);
    pc_hip_points_sum(E::ID, pts, 2, out);
    bool inf = true; for (int i = 0; i < 2 * E::NQ; i++) inf &= out[i] == 0;
    return from_xy(out, inf);
  }
  G1Affine mul(const FrT<E>& k) const {
    uint64_t p[2 * E::NQ], out[2 * E::NQ]; to_xy(p);
    pc_hip_point_mul(E::ID, p, k.l, out);
    bool inf = true; for (int i = 0; i < 2 * E::NQ; i++) inf &= out[i] == 0;
    return from_xy(out, inf);
  }
  G1Affine neg() const {
    if (infinity) return *this;
    typedef pc::host64::F64<typename E::C::FqP> Fq;
    G1Affine r = *this; Fq yy; memcpy(yy.l, y, sizeof(y)); Fq n = Fq::zero().sub(yy); memcpy(r.y, n.l, sizeof(y)); return r;
  }
};

// DensePolynomial<Fr>: coefficient vector, low degree first.
template <class E>
struct DensePolynomial {
  std::vector<FrT<E>> coeffs;
  size_t degree() const { size_t n = coeffs.size(); while (n > 0 && coeffs[n - 1].is_zero()) n--; return n ? n - 1 : 0; }
  bool is_zero() const { for (auto& c : coeffs) if (!c.is_zero()) return false; return true; }
  FrT<E> evaluate(const FrT<E>& z) const { FrT<E> acc = FrT<E>::zero(); for (size_t i = coeffs.size(); i-- > 0;) acc = acc * z + coeffs[i]; return acc; }
};

// Source of uniformly random field elements (the reference takes `&mut dyn RngCore`; the
// byte-level stream of ark_std's rngs is not reproduced -- SURVEY.md Appendix A).
template <class E>
struct RngCore { virtual ~RngCore() {} virtual FrT<E> next_fr() = 0; };

template <class E>
struct Randomness {   // kzg10/data_structures.rs:400-435
  DensePolynomial<E> blinding_polynomial;
  static Randomness empty() { return Randomness(); }
  bool is_hiding() const { return !blinding_polynomial.is_zero(); }
  static size_t calculate_hiding_polynomial_degree(size_t hiding_bound) { return hiding_bound + 1; }
  static Randomness rand(size_t hiding_bound, RngCore<E>& rng) {
    Randomness r; size_t d = calculate_hiding_polynomial_degree(hiding_bound);
    for (size_t i = 0; i <= d; i++) r.blinding_polynomial.coeffs.push_back(rng.next_fr());   // P::rand(d): d+1 coefficients
    return r;
  }
};

template <class E> struct Commitment { G1Affine<E> comm = G1Affine<E>::zero(); };                     // Commitment(pub E::G1Affine)
template <class E> struct Proof { G1Affine<E> w = G1Affine<E>::zero(); bool has_random_v = false; FrT<E> random_v = FrT<E>::zero(); };

// Powers: borrowed slices of the SRS (kzg10/data_structures.rs:124-136) plus their HBM residents.
template <class E>
struct Powers {
  const G1Affine<E>* powers_of_g = nullptr; size_t n_powers = 0;
  const G1Affine<E>* powers_of_gamma_g = nullptr; size_t n_gamma = 0;
  pc_ctx* ctx = nullptr; pc_srs* srs_g = nullptr; pc_srs* srs_gamma = nullptr; size_t g_offset = 0;
  size_t size() const { return n_powers; }

  // `trim`-time hook: upload both slices once.  g_offset lets shifted_powers reuse the same resident.
  static Error upload(pc_ctx* ctx, const G1Affine<E>* g, size_t n, const G1Affine<E>* gamma, size_t n_gamma, Powers& out) {
    out = Powers(); out.ctx = ctx; out.powers_of_g = g; out.n_powers = n; out.powers_of_gamma_g = gamma; out.n_gamma = n_gamma;
    int rc = pc_hip_srs_upload(ctx, E::ID, g, n, sizeof(G1Affine<E>), PC_MEM_HOST, &out.srs_g);
    if (rc == PC_OK && n_gamma) rc = pc_hip_srs_upload(ctx, E::ID, gamma, n_gamma, sizeof(G1Affine<E>), PC_MEM_HOST, &out.srs_gamma);
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = std::string(pc_hip_strerror(rc)) + ": " + pc_hip_last_error(ctx); return e; }
    return Error();
  }
  void release() { pc_hip_srs_free(srs_g); pc_hip_srs_free(srs_gamma); srs_g = srs_gamma = nullptr; }
};

template <class E>
struct KZG10 {
  typedef FrT<E> Fr;

  static Error check_degree_is_too_large(size_t degree, size_t num_powers) {          // kzg10/mod.rs:393-403
    size_t num_coefficients = degree + 1;
    if (num_coefficients > num_powers) { Error e; e.kind = Error::TooManyCoefficients; e.a = num_coefficients; e.b = num_powers; return e; }
    return Error();
  }
  static Error check_hiding_bound(size_t hiding_poly_degree, size_t num_powers) {        // kzg10/mod.rs:405-421
    Error e;
    if (hiding_poly_degree == 0) { e.kind = Error::HidingBoundIsZero; return e; }
    if (hiding_poly_degree >= num_powers) { e.kind = Error::HidingBoundToolarge; e.a = hiding_poly_degree; e.b = num_powers; return e; }
    return e;
  }

  // msm_bigint(&bases[offset..], coeffs) with the Montgomery->bigint conversion fused on the device
  static Error msm(pc_ctx* ctx, const pc_srs* srs, size_t offset, const Fr* scalars, size_t n, G1Affine<E>& out) {
    uint64_t xy[2 * E::NQ]; int inf = 0;
    int rc = pc_hip_msm(ctx, srs, offset, scalars, PC_SCALARS_MONTGOMERY, PC_MEM_HOST, n, xy, &inf);
    if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = std::string(pc_hip_strerror(rc)) + ": " + pc_hip_last_error(ctx); return e; }
    out = G1Affine<E>::from_xy(xy, inf != 0);
    return Error();
  }

  // KZG10::commit, kzg10/mod.rs:157-210
  static Error commit(const Powers<E>& powers, const DensePolynomial<E>& polynomial, const size_t* hiding_bound, RngCore<E>* rng,
                      Commitment<E>& out_comm, Randomness<E>& out_rand) {
    if (Error e = check_degree_is_too_large(polynomial.degree(), powers.size())) return e;
    // skip_leading_zeros_and_convert_to_bigints (:452-461)
    size_t num_leading_zeros = 0;
    while (num_leading_zeros < polynomial.coeffs.size() && polynomial.coeffs[num_leading_zeros].is_zero()) num_leading_zeros++;
    G1Affine<E> commitment;
    if (Error e = msm(powers.ctx, powers.srs_g, powers.g_offset + num_leading_zeros, polynomial.coeffs.data() + num_leading_zeros,
                      polynomial.coeffs.size() - num_leading_zeros, commitment)) return e;                     // :175-178
    Randomness<E> randomness = Randomness<E>::empty();
    if (hiding_bound) {
      if (!rng) { Error e; e.kind = Error::MissingRng; return e; }                                            // :183
      randomness = Randomness<E>::rand(*hiding_bound, *rng);
      if (Error e = check_hiding_bound(randomness.blinding_polynomial.degree(), powers.n_gamma)) return e;
    }
    G1Affine<E> random_commitment = G1Affine<E>::zero();                                                       // :196-204
    if (!randomness.blinding_polynomial.coeffs.empty())
      if (Error e = msm(powers.ctx, powers.srs_gamma, 0, randomness.blinding_polynomial.coeffs.data(),
                        randomness.blinding_polynomial.coeffs.size(), random_commitment)) return e;
    out_comm.comm = commitment.add(random_commitment);                                                         // :206-209
    out_rand = randomness;
    return Error();
  }

  // compute_witness_polynomial, kzg10/mod.rs:217-240: quotient by (x - point); the blinding one on the host (tiny)
  static Error compute_witness_polynomial(pc_ctx* ctx, const DensePolynomial<E>& p, const Fr& point, const Randomness<E>& randomness,
                                          DensePolynomial<E>& witness, DensePolynomial<E>* random_witness, bool& has_random) {
    witness.coeffs.assign(p.coeffs.size() > 1 ? p.coeffs.size() - 1 : 0, Fr::zero());
    if (!witness.coeffs.empty()) {
      int rc = pc_hip_witness_poly(ctx, E::ID, p.coeffs.data(), PC_MEM_HOST, p.coeffs.size(), point.l, witness.coeffs.data(), PC_MEM_HOST);
      if (rc != PC_OK) { Error e; e.kind = Error::Backend; e.msg = pc_hip_strerror(rc); return e; }
    }
    has_random = randomness.is_hiding();
    if (has_random && random_witness) {
      const auto& c = randomness.blinding_polynomial.coeffs;
      random_witness->coeffs.assign(c.size() - 1, Fr::zero());
      Fr acc = Fr::zero();
      for (size_t i = c.size() - 1; i >= 1; i--) { acc = c[i] + point * acc; random_witness->coeffs[i - 1] = acc; }
    }
    return Error();
  }

  // open_with_witness_polynomial, kzg10/mod.rs:243-284
  static Error open_with_witness_polynomial(const Powers<E>& powers, const Fr& point, const Randomness<E>& randomness,
                                            const DensePolynomial<E>& witness_polynomial, const DensePolynomial<E>* hiding_witness_polynomial,
                                            Proof<E>& proof) {
    if (Error e = check_degree_is_too_large(witness_polynomial.degree(), powers.size())) return e;
    size_t lz = 0;
    while (lz < witness_polynomial.coeffs.size() && witness_polynomial.coeffs[lz].is_zero()) lz++;
    G1Affine<E> w;
    if (Error e = msm(powers.ctx, powers.srs_g, powers.g_offset + lz, witness_polynomial.coeffs.data() + lz,
                      witness_polynomial.coeffs.size() - lz, w)) return e;                                    // :255-258
    proof.has_random_v = false;
    if (hiding_witness_polynomial) {
      Fr blinding_evaluation = randomness.blinding_polynomial.evaluate(point);                                  // :264
      G1Affine<E> rw;
      if (Error e = msm(powers.ctx, powers.srs_gamma, 0, hiding_witness_polynomial->coeffs.data(), hiding_witness_polynomial->coeffs.size(), rw)) return e;
      w = w.add(rw);                                                                                          // :270-273
      proof.has_random_v = true; proof.random_v = blinding_evaluation;
    }
    proof.w = w;
    return Error();
  }

  // KZG10::open, kzg10/mod.rs:287-310
  static Error open(const Powers<E>& powers, const DensePolynomial<E>& p, const Fr& point, const Randomness<E>& rand, Proof<E>& proof) {
    if (Error e = check_degree_is_too_large(p.degree(), powers.size())) return e;
    DensePolynomial<E> witness_poly, hiding_witness_poly; bool has_random = false;
    if (Error e = compute_witness_polynomial(powers.ctx, p, point, rand, witness_poly, &hiding_witness_poly, has_random)) return e;
    return open_with_witness_polynomial(powers, point, rand, witness_poly, has_random ? &hiding_witness_poly : nullptr, proof);
  }

  // KZG10::setup restated for TEST-SIZED degrees only (kzg10/mod.rs:53-124): powers beta^i g and
  // gamma_g beta^i by host scalar multiplication.  (The device fixed-base generator is SURVEY 8f row 3.)
  static void setup_for_tests(size_t max_degree, const Fr& beta, const G1Affine<E>& g, const G1Affine<E>& gamma_g,
                              std::vector<G1Affine<E>>& powers_of_g, std::vector<G1Affine<E>>& powers_of_gamma_g) {
    Fr cur = Fr::one();
    for (size_t i = 0; i <= max_degree + 1; i++) {
      if (i <= max_degree) powers_of_g.push_back(g.mul(cur));
      powers_of_gamma_g.push_back(gamma_g.mul(cur));          // max_degree + 2 powers (:82-86)
      cur = cur * beta;
    }
  }
};

}  // namespace pc_host
