// Host-layer tests for pc_host::KZG10 (poly_commit_amd/host/kzg10.hpp), written after the
// reference's own KZG10 tests (poly-commit/src/kzg10/mod.rs:519-674).  The reference checks
// openings with a pairing; no pairing exists here, so `check` is replaced by the same equation
// evaluated in G1 with the test's known trapdoor beta:
//     C - v*g - v_bar*gamma_g  ==  (beta - z) * W          (kzg10/mod.rs:314-333)
#include <stdio.h>
#include <stdlib.h>
#include "../../poly_commit_amd/host/marlin_kzg10.hpp"
#include "../../poly_commit_amd/host/sonic_kzg10.hpp"
#include "../../poly_commit_amd/host/linear_codes.hpp"
#include "../../poly_commit_amd/host/ipa_pc.hpp"
#include "../../poly_commit_amd/host/hyrax.hpp"

using namespace pc_host;

template <class E>
struct TestRng : RngCore<E> {
  uint64_t s;
  explicit TestRng(uint64_t seed) : s(seed) {}
  uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  FrT<E> next_fr() override {   // uniform canonical value < r, then into Montgomery form
    typedef pc::host64::F64<typename E::C::FrP> F;
    for (;;) {
      F t; for (int i = 0; i < 4; i++) t.l[i] = next();
      t.l[3] &= (1ull << (E::C::FrP::BITS - 192)) - 1;
      if (F::geq(t.l)) continue;
      F r2; memcpy(r2.l, E::C::FrP::R2, 32);
      return FrT<E>::of(t.mul(r2));
    }
  }
};

#define CHECK(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)

template <class E>
static G1Affine<E> generator() {
  uint64_t xy[2 * E::NQ];
  memcpy(xy, E::C::GX, 8 * E::NQ); memcpy(xy + E::NQ, E::C::GY, 8 * E::NQ);
  return G1Affine<E>::from_xy(xy, false);
}

template <class E>
static DensePolynomial<E> rand_poly(size_t degree, TestRng<E>& rng) {
  DensePolynomial<E> p; for (size_t i = 0; i <= degree; i++) p.coeffs.push_back(rng.next_fr()); return p;
}

template <class E>
static void run(pc_ctx* ctx, const char* name) {
  typedef KZG10<E> K; typedef FrT<E> Fr;
  TestRng<E> rng(0x5EED);
  const size_t max_degree = 24;
  Fr beta = rng.next_fr();
  G1Affine<E> g = generator<E>(), gamma_g = g.mul(rng.next_fr());
  std::vector<G1Affine<E>> pg, pgg;
  K::setup_for_tests(max_degree, beta, g, gamma_g, pg, pgg);
  Powers<E> powers;
  Error e = Powers<E>::upload(ctx, pg.data(), pg.size(), pgg.data(), pgg.size(), powers);
  CHECK(!e);

  // add_commitments_test (kzg10/mod.rs:520-544): commit(f * p) == f * commit(p)
  {
    DensePolynomial<E> p = rand_poly<E>(10, rng);
    Fr f = rng.next_fr();
    DensePolynomial<E> f_p = p; for (auto& c : f_p.coeffs) c = c * f;
    Commitment<E> comm, f_comm; Randomness<E> r1, r2;
    CHECK(!K::commit(powers, p, nullptr, nullptr, comm, r1));
    CHECK(!K::commit(powers, f_p, nullptr, nullptr, f_comm, r2));
    CHECK(!comm.comm.is_zero() && !r1.is_hiding());
    CHECK(comm.comm.mul(f) == f_comm.comm);
  }
  // end_to_end_test (kzg10/mod.rs:546-575), hiding on; the pairing check in G1 via the trapdoor
  for (int iter = 0; iter < 12; iter++) {
    size_t degree = 1 + rng.next() % 19;
    DensePolynomial<E> p = rand_poly<E>(degree, rng);
    size_t hiding_bound = 1 + rng.next() % 3;
    const bool hide = iter % 3 != 0;
    Commitment<E> comm; Randomness<E> rand;
    CHECK(!K::commit(powers, p, hide ? &hiding_bound : nullptr, hide ? &rng : nullptr, comm, rand));
    CHECK(rand.is_hiding() == hide);
    Fr point = rng.next_fr();
    Fr value = p.evaluate(point);
    Proof<E> proof;
    CHECK(!K::open(powers, p, point, rand, proof));
    CHECK(proof.has_random_v == hide);
    G1Affine<E> lhs = comm.comm.add(g.mul(value).neg());
    if (hide) lhs = lhs.add(gamma_g.mul(proof.random_v).neg());
    G1Affine<E> rhs = proof.w.mul(beta - point);
    CHECK(lhs == rhs);
    // a wrong value must not satisfy the equation
    G1Affine<E> bad = comm.comm.add(g.mul(value + Fr::one()).neg());
    if (hide) bad = bad.add(gamma_g.mul(proof.random_v).neg());
    CHECK(!(bad == rhs));
  }
  // leading-zero coefficients: skip_leading_zeros_and_convert_to_bigints (:452-461)
  {
    DensePolynomial<E> p = rand_poly<E>(12, rng);
    p.coeffs[0] = p.coeffs[1] = Fr::zero();
    Commitment<E> c1; Randomness<E> r;
    CHECK(!K::commit(powers, p, nullptr, nullptr, c1, r));
    G1Affine<E> want = G1Affine<E>::zero();
    for (size_t i = 0; i < p.coeffs.size(); i++) want = want.add(pg[i].mul(p.coeffs[i]));
    CHECK(c1.comm == want);
  }
  // test_degree_is_too_large (kzg10/mod.rs:663-674) and the rng / hiding-bound errors
  {
    DensePolynomial<E> p = rand_poly<E>(max_degree + 1, rng);
    Commitment<E> c; Randomness<E> r;
    Error err = K::commit(powers, p, nullptr, nullptr, c, r);
    CHECK(err.kind == Error::TooManyCoefficients && err.a == max_degree + 2 && err.b == max_degree + 1);
    DensePolynomial<E> q = rand_poly<E>(5, rng);
    size_t hb = 2;
    CHECK(K::commit(powers, q, &hb, nullptr, c, r).kind == Error::MissingRng);
    size_t huge = max_degree + 5;
    CHECK(K::commit(powers, q, &huge, &rng, c, r).kind == Error::HidingBoundToolarge);
    Proof<E> pr;
    CHECK(K::open(powers, p, rng.next_fr(), Randomness<E>::empty(), pr).kind == Error::TooManyCoefficients);
  }
  powers.release();

  // ---- MarlinKZG10 (marlin_pc tests: single_poly / two_polys_degree_bound / full_end_to_end shapes, marlin_pc/mod.rs:570-814)
  // The verifier's equation (marlin/mod.rs:109-148 + kzg10 check) is evaluated in G1 with the trapdoor:
  //   sum_j xi_j (C_j - v_j g) + sum_{bounded j} xi'_j (shC_j - v_j beta^(max - d_j) g) - random_v gamma_g == (beta - z) w
  {
    typedef MarlinKZG10<E> M;
    struct Chal : ChallengeSource<E> { TestRng<E> r{0xC4A1}; std::vector<FrT<E>> log; FrT<E> squeeze_challenge() override { FrT<E> c = r.next_fr(); log.push_back(c); return c; } };
    for (int variant = 0; variant < 3; variant++) {
      const bool hiding = variant == 1;
      std::vector<size_t> bounds = {9, 14, 20};
      CommitterKey<E> ck;
      CHECK(!M::trim(ctx, pg, pgg, 22, 2, &bounds, ck));
      std::vector<LabeledPolynomial<E>> polys;
      size_t degs[4] = {7, 14, 22, 3};
      for (int j = 0; j < 4; j++) {
        LabeledPolynomial<E> lp; lp.label = "p" + std::to_string(j); lp.polynomial = rand_poly<E>(degs[j], rng);
        if (j == 0) lp.degree_bound = 9;
        if (j == 1 && variant != 2) lp.degree_bound = 14;
        if (hiding) lp.hiding_bound = 1;
        polys.push_back(lp);
      }
      std::vector<MarlinCommitment<E>> comms; std::vector<MarlinRandomness<E>> states;
      CHECK(!M::commit(ck, polys, hiding ? &rng : nullptr, comms, states));
      // degree-bound commitments: shifted_comm == beta^(max_degree - d) * comm when nothing is hidden
      if (!hiding) {
        Fr bp = Fr::one(); for (size_t i = 0; i < ck.max_degree - 9; i++) bp = bp * beta;
        CHECK(comms[0].shifted_comm.has_value() && *comms[0].shifted_comm == comms[0].comm.mul(bp));
      }
      Fr z = rng.next_fr();
      Chal sponge;
      Proof<E> proof;
      CHECK(!M::open(ck, polys, z, sponge, states, proof));
      CHECK(proof.has_random_v == hiding);
      G1Affine<E> lhs = G1Affine<E>::zero();
      size_t ci = 0;
      for (int j = 0; j < 4; j++) {
        Fr v = polys[j].polynomial.evaluate(z);
        Fr xi = sponge.log[ci++];
        lhs = lhs.add(comms[j].comm.add(g.mul(v).neg()).mul(xi));
        if (polys[j].degree_bound) {
          Fr xi1 = sponge.log[ci++];
          Fr bp = Fr::one(); for (size_t i = 0; i < ck.max_degree - *polys[j].degree_bound; i++) bp = bp * beta;
          lhs = lhs.add(comms[j].shifted_comm->add(g.mul(v * bp).neg()).mul(xi1));
        }
      }
      if (hiding) lhs = lhs.add(gamma_g.mul(proof.random_v).neg());
      CHECK(lhs == proof.w.mul(beta - z));
      // a degree bound the key does not enforce is rejected (Error::UnsupportedDegreeBound, kzg10/mod.rs:430-434)
      LabeledPolynomial<E> bad = polys[0]; bad.degree_bound = 11;
      CHECK(M::check_degrees_and_bounds(ck, bad).kind == Error::UnsupportedDegreeBound);

      // ---- open_combinations (marlin/mod.rs:224-316; the shapes of single_equation / two_equation / two_equation_degree_bound,
      // lib.rs:1302-1384): eq0 = 2 p2 + 3 p3 - 5 at z0;  eq1 = p2 - p3 at z0 and z1;  eq2 = p0 alone (degree-bounded, coefficient
      // one) at z1.  ONE proof per point label over the COMBINED polynomials; checked the way check_combinations does it
      // (marlin/mod.rs:318-409 -> batch_check -> accumulate_commitments_and_values :109-148), with the trapdoor in place of the
      // pairing: per point   sum_k xi_k (C_k - v_k g) + sum_{bounded k} xi'_k (shC_k - v_k beta^(max - d_k) g) - random_v gamma_g
      //                      == (beta - z) w,   C_k the combined commitments, v_k the equation's value MINUS its constant term.
      {
        auto L = [](const char* l) { return std::optional<std::string>(l); };
        std::vector<LinearCombination<E>> lcs(3);
        lcs[0].label = "eq0"; lcs[0].terms = {{Fr::from_u64(2), L("p2")}, {Fr::from_u64(3), L("p3")}, {Fr::zero() - Fr::from_u64(5), std::nullopt}};
        lcs[1].label = "eq1"; lcs[1].terms = {{Fr::one(), L("p2")}, {Fr::zero() - Fr::one(), L("p3")}};
        lcs[2].label = "eq2"; lcs[2].terms = {{Fr::one(), L("p0")}};
        const Fr z0 = rng.next_fr(), z1 = rng.next_fr();
        std::vector<Query<E>> qs = {{"eq2", "z1", z1}, {"eq1", "z1", z1}, {"eq0", "z0", z0}, {"eq1", "z0", z0}};     // (any order: a set)
        Chal sp2;
        std::vector<Proof<E>> proofs; std::vector<LabeledMarlinCommitment<E>> lcc;
        CHECK(!M::open_combinations(ck, lcs, polys, comms, qs, sp2, states, proofs, &lcc));
        CHECK(proofs.size() == 2 && lcc.size() == 3);                       // one proof per point label, BatchLCProof.evals = None
        // the combined commitments are the combinations of the commitments (what the verifier rebuilds)
        CHECK(lcc[0].commitment.comm == comms[2].comm.mul(Fr::from_u64(2)).add(comms[3].comm.mul(Fr::from_u64(3))) && !lcc[0].commitment.shifted_comm);
        CHECK(lcc[1].commitment.comm == comms[2].comm.add(comms[3].comm.neg()));
        CHECK(lcc[2].commitment.comm == comms[0].comm && lcc[2].commitment.shifted_comm && *lcc[2].commitment.shifted_comm == *comms[0].shifted_comm && lcc[2].degree_bound == polys[0].degree_bound);
        auto val = [&](int k, const Fr& z) {                                // the combined polynomial's value = equation value - constant
          const Fr a = polys[2].polynomial.evaluate(z), b = polys[3].polynomial.evaluate(z);
          return k == 0 ? Fr::from_u64(2) * a + Fr::from_u64(3) * b : k == 1 ? a - b : polys[0].polynomial.evaluate(z);
        };
        size_t ci = 0;
        const std::vector<std::vector<int>> at = {{0, 1}, {1, 2}};           // point "z0": eq0, eq1;  point "z1": eq1, eq2 (label order)
        for (int pt = 0; pt < 2; pt++) {
          const Fr z = pt == 0 ? z0 : z1;
          G1Affine<E> acc = G1Affine<E>::zero();
          for (int k : at[pt]) {
            const Fr v = val(k, z), xi = sp2.log[ci++];
            acc = acc.add(lcc[k].commitment.comm.add(g.mul(v).neg()).mul(xi));
            if (lcc[k].degree_bound) {
              const Fr xi1 = sp2.log[ci++];
              Fr bp = Fr::one(); for (size_t i = 0; i < ck.max_degree - *lcc[k].degree_bound; i++) bp = bp * beta;
              acc = acc.add(lcc[k].commitment.shifted_comm->add(g.mul(v * bp).neg()).mul(xi1));
            }
          }
          if (hiding) acc = acc.add(gamma_g.mul(proofs[pt].random_v).neg());
          CHECK(proofs[pt].has_random_v == hiding);
          CHECK(acc == proofs[pt].w.mul(beta - z));
        }
        CHECK(ci == sp2.log.size());
        // the trait's DEFAULT open_combinations (lib.rs:445-487) would have opened p0, p2, p3 individually: a different number of
        // challenges and different proofs -- the verifier above could not accept them.  The reference's error paths:
        std::vector<LinearCombination<E>> bad_lc(1);
        bad_lc[0].label = "bad"; bad_lc[0].terms = {{Fr::one(), L("p0")}, {Fr::one(), L("p2")}};             // a bounded polynomial inside a real combination
        Chal sp3;
        CHECK(M::open_combinations(ck, bad_lc, polys, comms, {{"bad", "z0", z0}}, sp3, states, proofs).kind == Error::EquationHasDegreeBounds);
        bad_lc[0].terms = {{Fr::one(), L("nope")}};
        CHECK(M::open_combinations(ck, bad_lc, polys, comms, {{"bad", "z0", z0}}, sp3, states, proofs).kind == Error::MissingPolynomial);
        // batch_open by itself: two polynomials at one point == one `open` over them in label order
        Chal sa, sb;
        std::vector<Proof<E>> bp1; Proof<E> single;
        CHECK(!M::batch_open(ck, polys, {{"p3", "pt", z0}, {"p2", "pt", z0}}, sa, states, bp1));
        CHECK(!M::open(ck, {polys[2], polys[3]}, z0, sb, {states[2], states[3]}, single));
        CHECK(bp1.size() == 1 && bp1[0].w == single.w);
      }
      ck.release();
    }
  }
  // ---- SonicKZG10 (sonic_pc tests: the same single_poly / degree_bound / hiding shapes, sonic_pc/mod.rs:729-928): a bounded
  // polynomial is committed ONCE, over the shifted powers (C_j = beta^(D - d_j) p_j(beta) g, its hiding term shifted alike); `open`
  // is one combination over the plain powers.  The verifier's equation (accumulate_elems / check_elems, :540-680) in G1 with the
  // trapdoor:   sum_j xi_j (beta^-(D - d_j) C_j - v_j g) - random_v gamma_g == (beta - z) w
  {
    typedef SonicKZG10<E> S;
    struct Chal : ChallengeSource<E> { TestRng<E> r{0x50A1C}; std::vector<FrT<E>> log; FrT<E> squeeze_challenge() override { FrT<E> c = r.next_fr(); log.push_back(c); return c; } };
    for (int variant = 0; variant < 3; variant++) {
      const bool hiding = variant == 1;
      std::vector<size_t> bounds = {9, 14, 20};
      SonicCommitterKey<E> ck;
      CHECK(!S::trim(ctx, pg, pgg, 22, 2, &bounds, ck));
      CHECK(ck.shifted_powers_of_g->size() == 20 + 1 && ck.shifted_powers_of_gamma_g.size() == 3);
      std::vector<LabeledPolynomial<E>> polys;
      size_t degs[4] = {7, 14, 22, 3};
      for (int j = 0; j < 4; j++) {
        LabeledPolynomial<E> lp; lp.label = "s" + std::to_string(j); lp.polynomial = rand_poly<E>(degs[j], rng);
        if (j == 0) lp.degree_bound = 9;
        if (j == 1 && variant != 2) lp.degree_bound = 14;
        if (hiding) lp.hiding_bound = 1;
        polys.push_back(lp);
      }
      std::vector<Commitment<E>> comms; std::vector<Randomness<E>> states;
      CHECK(!S::commit(ck, polys, hiding ? &rng : nullptr, comms, states));
      auto shift_of = [&](int j) { Fr bp = Fr::one(); if (polys[j].degree_bound) for (size_t i = 0; i < ck.max_degree - *polys[j].degree_bound; i++) bp = bp * beta; return bp; };
      if (!hiding) {      // the bounded commitment is the plain one times beta^(D - d)
        Commitment<E> plain; Randomness<E> r0;
        CHECK(!K::commit(ck.powers(), polys[0].polynomial, nullptr, nullptr, plain, r0));
        CHECK(comms[0].comm == plain.comm.mul(shift_of(0)));
      }
      Fr z = rng.next_fr();
      Chal sponge;
      Proof<E> proof;
      CHECK(!S::open(ck, polys, z, sponge, states, proof));
      CHECK(proof.has_random_v == hiding && sponge.log.size() == 5);       // one challenge up front, one after every polynomial
      G1Affine<E> lhs = G1Affine<E>::zero();
      for (int j = 0; j < 4; j++) {
        Fr v = polys[j].polynomial.evaluate(z);
        lhs = lhs.add(comms[j].comm.mul(shift_of(j).inverse()).add(g.mul(v).neg()).mul(sponge.log[j]));
      }
      if (hiding) lhs = lhs.add(gamma_g.mul(proof.random_v).neg());
      CHECK(lhs == proof.w.mul(beta - z));
      LabeledPolynomial<E> bad = polys[0]; bad.degree_bound = 11;
      CHECK(S::check_degrees_and_bounds(ck, bad).kind == Error::UnsupportedDegreeBound);
      std::vector<size_t> too_high = {23};
      SonicCommitterKey<E> ck2;
      CHECK(S::trim(ctx, pg, pgg, 22, 2, &too_high, ck2).kind == Error::UnsupportedDegreeBound);       // sonic_pc/mod.rs:186-188
      ck.release();
    }
  }
  // ---- Ligero encoder: test_reed_solomon (linear_codes/utils.rs:303-331) and the matrix shape ----
  {
    const size_t rho_inv = 3;
    for (int i = 1; i < 10; i++) {
      size_t m = (size_t)1 << i;
      DensePolynomial<E> pol = rand_poly<E>(m - 1, rng);
      std::vector<FrT<E>> encoded;
      CHECK(!LinearEncode<E>::reed_solomon(ctx, pol.coeffs, rho_inv, encoded));
      size_t size = 1; unsigned lg = 0; while (size < m * rho_inv) { size <<= 1; lg++; }
      CHECK(encoded.size() == size);
      FrT<E> w = domain_generator<E>(lg), x = FrT<E>::one();
      for (size_t j = 0; j < size; j++) { CHECK(pol.evaluate(x) == encoded[j]); x = x * w; }   // large_domain.element(j)
    }
    LigeroPCParams param;                                       // rho_inv 4, lambda 128
    std::pair<size_t, size_t> d24;
    CHECK(!param.compute_dimensions<E>((size_t)1 << 24, d24.first, d24.second));
    CHECK(d24.first == 512 && d24.second == 32768);
    DensePolynomial<E> pol = rand_poly<E>(999, rng);
    Matrix<E> mat, ext;
    CHECK(!LinearEncode<E>::compute_matrices(ctx, pol, param, mat, ext));
    CHECK(mat.n * mat.m >= 1000 && ext.n == mat.n && ext.m >= 4 * mat.m);
    FrT<E> w = domain_generator<E>(ark_log2(ext.m));
    for (size_t r = 0; r < mat.n; r++) {                        // every encoded row agrees with its row polynomial at omega^3
      DensePolynomial<E> row; row.coeffs.assign(mat.entries.begin() + r * mat.m, mat.entries.begin() + (r + 1) * mat.m);
      CHECK(row.evaluate(w * w * w) == ext.at(r, 3));
    }
  }
  // ---- LinearCodePCS::commit (linear_codes/mod.rs:234-297): fused device chain == the single steps ----
  {
    LigeroPCParams param;
    LinearCodePCS<E> pcs;
    DensePolynomial<E> pol = rand_poly<E>(4999, rng);
    LinCodePCCommitment com; LinCodePCCommitmentState<E> st;
    CHECK(!pcs.commit(ctx, pol, param, com, st));
    CHECK(com.metadata.n_rows == st.mat.n && com.metadata.n_cols == st.mat.m && com.metadata.n_ext_cols == st.ext_mat.m);
    Matrix<E> mat2, ext2;
    CHECK(!LinearEncode<E>::compute_matrices(ctx, pol, param, mat2, ext2));
    CHECK(ext2.entries.size() == st.ext_mat.entries.size());
    CHECK(memcmp(ext2.entries.data(), st.ext_mat.entries.data(), ext2.entries.size() * sizeof(FrT<E>)) == 0);
    std::vector<uint8_t> leaves(ext2.m * 32), nodes((ext2.m - 1) * 32);
    CHECK(pc_hip_column_hash(ctx, E::ID, ext2.entries.data(), PC_MEM_HOST, ext2.n, ext2.m, PC_HASH_BLAKE2S, leaves.data(), PC_MEM_HOST) == PC_OK);
    CHECK(pc_hip_merkle_tree(ctx, PC_HASH_SHA256, leaves.data(), PC_MEM_HOST, ext2.m, 1, nodes.data(), PC_MEM_HOST) == PC_OK);
    CHECK(leaves == st.leaves && nodes == st.nodes && memcmp(com.root, nodes.data(), 32) == 0);
    // the commitment without the encoded matrix coming back is the same commitment
    LinCodePCCommitment com2; LinCodePCCommitmentState<E> st2;
    CHECK(!pcs.commit(ctx, pol, param, com2, st2, false));
    CHECK(memcmp(com.root, com2.root, 32) == 0 && st2.ext_mat.entries.empty());
    // generate_proof step 1: v = b^T * mat, checked against the host field arithmetic
    std::vector<FrT<E>> b(st.mat.n), v;
    for (auto& x : b) x = rng.next_fr();
    CHECK(!LinearCodePCS<E>::row_mul(ctx, st.mat, b, v));
    for (size_t c = 0; c < st.mat.m; c += 7) {
      FrT<E> acc = FrT<E>::zero();
      for (size_t r = 0; r < st.mat.n; r++) acc = acc + b[r] * st.mat.at(r, c);
      CHECK(acc == v[c]);
    }
    b.pop_back();
    CHECK(LinearCodePCS<E>::row_mul(ctx, st.mat, b, v).kind == Error::Backend);
    // authentication path shape: log2(n_ext_cols) - 1 inner siblings
    uint8_t sib[32]; std::vector<uint8_t> path;
    LinearCodePCS<E>::merkle_path(st, 5, sib, path);
    CHECK(path.size() == (ark_log2(st.ext_mat.m) - 1) * 32 && memcmp(sib, st.leaves.data() + 4 * 32, 32) == 0);
    CHECK(memcmp(path.data() + path.size() - 32, st.nodes.data() + 2 * 32, 32) == 0);   // leaf 5 is in the left half: last sibling is node 2
    // open / check (linear_codes/mod.rs:300-503) with the sponge's outputs supplied: honest, wrong value, altered v / column / index
    {
      std::vector<size_t> idx; for (size_t j = 0; j < 24; j++) idx.push_back((j * 7919 + 13) % com.metadata.n_ext_cols);
      std::vector<FrT<E>> r(st.mat.n); for (auto& x : r) x = rng.next_fr();
      const FrT<E> z = rng.next_fr(), value = pol.evaluate(z);
      for (int wf = 0; wf < 2; wf++) {
        typename LinearCodePCS<E>::ProofSingle pr;
        CHECK(!pcs.open(ctx, com, st, z, idx, wf ? &r : nullptr, pr));
        bool ok = false;
        CHECK(!pcs.check(ctx, com, param, z, value, pr, idx, wf ? &r : nullptr, ok) && ok);
        CHECK(!pcs.check(ctx, com, param, z, value + FrT<E>::one(), pr, idx, wf ? &r : nullptr, ok) && !ok);
        { auto bad = pr; bad.v[3] = bad.v[3] + FrT<E>::one(); CHECK(pcs.check(ctx, com, param, z, value, bad, idx, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        { auto bad = pr; bad.columns[2][1] = bad.columns[2][1] + FrT<E>::one(); CHECK(pcs.check(ctx, com, param, z, value, bad, idx, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        { auto bad = idx; bad[0] = idx[1]; CHECK(pcs.check(ctx, com, param, z, value, pr, bad, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        CHECK(pcs.check(ctx, com, param, z, value, pr, idx, wf ? nullptr : &r, ok).kind == Error::InvalidCommitment);
        // malformed proofs from an untrusted prover end in InvalidCommitment, not in an out-of-bounds read (round-2 advisor finding)
        { auto bad = pr; bad.leaf_index.pop_back(); CHECK(pcs.check(ctx, com, param, z, value, bad, idx, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        { auto bad = pr; bad.leaf_sibling.clear(); CHECK(pcs.check(ctx, com, param, z, value, bad, idx, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        { auto bad = pr; bad.paths[1].resize(bad.paths[1].size() - 32); CHECK(pcs.check(ctx, com, param, z, value, bad, idx, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        { auto bad = pr; bad.columns.resize(1); bad.paths.resize(1); bad.leaf_index.resize(1); bad.leaf_sibling.resize(1);
          CHECK(pcs.check(ctx, com, param, z, value, bad, idx, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        { auto bad = idx; bad[0] = com.metadata.n_ext_cols; CHECK(pcs.check(ctx, com, param, z, value, pr, bad, wf ? &r : nullptr, ok).kind == Error::InvalidCommitment); }
        if (wf) { auto bad = pr; bad.well_formedness.pop_back(); CHECK(pcs.check(ctx, com, param, z, value, bad, idx, &r, ok).kind == Error::InvalidCommitment); }
      }
    }
  }
  // ---- InnerProductArgPC: cm_commit and the halving loop of open (ipa_pc/mod.rs:54-72, 664-711) against the
  //      same rounds replayed with host point / field arithmetic ----
  {
    struct Fixed : IpaChallengeSource<E> {
      std::vector<FrT<E>> u; size_t k = 0;
      FrT<E> next(const G1Affine<E>&, const G1Affine<E>&) override { return u[k++]; }
    } ch;
    const size_t n0 = 8;
    std::vector<G1Affine<E>> key; std::vector<FrT<E>> c, z;
    for (size_t i = 0; i < n0; i++) { key.push_back(g.mul(rng.next_fr())); c.push_back(rng.next_fr()); }
    for (int i = 0; i < 3; i++) ch.u.push_back(rng.next_fr());
    const FrT<E> point = rng.next_fr();
    const G1Affine<E> h_prime = g.mul(rng.next_fr());
    G1Affine<E> cm;
    CHECK(!InnerProductArgPC<E>::cm_commit(ctx, key, c, nullptr, nullptr, cm));
    { G1Affine<E> want = G1Affine<E>::zero(); for (size_t i = 0; i < n0; i++) want = want.add(key[i].mul(c[i])); CHECK(cm == want); }
    FrT<E> rnd = rng.next_fr();
    G1Affine<E> cmh;
    CHECK(!InnerProductArgPC<E>::cm_commit(ctx, key, c, &h_prime, &rnd, cmh));
    CHECK(cmh == cm.add(h_prime.mul(rnd)));
    IpaProof<E> proof;
    CHECK(!InnerProductArgPC<E>::open_rounds(ctx, key, c, point, h_prime, ch, proof));
    CHECK(proof.l_vec.size() == 3 && proof.r_vec.size() == 3 && ch.k == 3);
    { FrT<E> cur = FrT<E>::one(); for (size_t i = 0; i < n0; i++) { z.push_back(cur); cur = cur * point; } }
    size_t n = n0;
    for (size_t round = 0; n > 1; round++) {
      const size_t h = n / 2;
      G1Affine<E> l = G1Affine<E>::zero(), r = G1Affine<E>::zero();
      FrT<E> ipl = FrT<E>::zero(), ipr = FrT<E>::zero();
      for (size_t i = 0; i < h; i++) {
        l = l.add(key[i].mul(c[h + i])); r = r.add(key[h + i].mul(c[i]));
        ipl = ipl + c[h + i] * z[i]; ipr = ipr + c[i] * z[h + i];
      }
      l = l.add(h_prime.mul(ipl)); r = r.add(h_prime.mul(ipr));
      CHECK(l == proof.l_vec[round] && r == proof.r_vec[round]);
      const FrT<E> u = ch.u[round], ui = u.inverse();
      CHECK(u * ui == FrT<E>::one());
      for (size_t i = 0; i < h; i++) { c[i] = c[i] + ui * c[h + i]; z[i] = z[i] + u * z[h + i]; key[i] = key[i].add(key[h + i].mul(u)); }
      n = h;
    }
    CHECK(proof.final_comm_key == key[0] && proof.c == c[0]);
    std::vector<FrT<E>> odd(c.begin(), c.begin() + 3);
    std::vector<G1Affine<E>> k3(key.begin(), key.begin() + 3);
    CHECK(InnerProductArgPC<E>::open_rounds(ctx, k3, odd, point, h_prime, ch, proof).kind == Error::Backend);    // not a power of two
  }
  printf("%s: ipa cm_commit/open rounds, ligero reed_solomon/compute_matrices/commit/row_mul, marlin commit/open/batch_open/open_combinations with degree bounds (hiding on/off), add_commitments, end_to_end (hiding on/off), leading zeros, degree/rng/hiding errors OK\n", name);
}

// Host arithmetic that needs no device: the reference's own calculate_t tests
// (linear_codes/utils.rs:344-359, Fq of BLS12-377 = 377 bits) and the shape table of SURVEY.md 8d.
static void host_logic_checks() {
  long t;
  t = calculate_t(377, 128, 3, 4, (size_t)1 << 32); CHECK(t > 0 && t < 200);        // test_calculate_t_with_good_parameters
  t = calculate_t(377, 256, 3, 4, (size_t)1 << 32); CHECK(t > 0 && t < 400);
  CHECK(calculate_t(377, 377 - 60, 3, 4, (size_t)1 << 60) < 0);                     // test_calculate_t_with_bad_parameters
  CHECK(calculate_t(377, 400, 3, 4, (size_t)1 << 32) < 0);
  LigeroPCParams param;                                                              // rho_inv 4, lambda 128
  const size_t want[5][3] = {{12, 8, 512}, {16, 32, 2048}, {20, 128, 8192}, {22, 256, 16384}, {24, 512, 32768}};
  for (auto& w : want) {
    size_t n = 0, m = 0;
    CHECK(!param.compute_dimensions<Bls12_381>((size_t)1 << w[0], n, m)); CHECK(n == w[1] && m == w[2]);
    CHECK(!param.compute_dimensions<Bn254>((size_t)1 << w[0], n, m)); CHECK(n == w[1] && m == w[2]);
  }
  LigeroPCParams bad; bad.sec_param = 400;                                           // no 255-bit field gives 2^-400
  size_t n = 0, m = 0;
  CHECK(bad.compute_dimensions<Bls12_381>((size_t)1 << 20, n, m).kind == Error::InvalidParameters);
  // ---- device-free parts of the newer host layers ----
  { // SHA-256 (FIPS 180-4 "abc") and a Merkle path built and verified with the host digests (byte-digest Config)
    uint8_t d[32];
    Sha256Host::digest((const uint8_t*)"abc", 3, d);
    static const uint8_t want_abc[32] = {0xba, 0x78, 0x16, 0xbf, 0x8f, 0x01, 0xcf, 0xea, 0x41, 0x41, 0x40, 0xde, 0x5d, 0xae, 0x22, 0x23,
                                         0xb0, 0x03, 0x61, 0xa3, 0x96, 0x17, 0x7a, 0x9c, 0xb4, 0x10, 0xff, 0x61, 0xf2, 0x00, 0x15, 0xad};
    CHECK(memcmp(d, want_abc, 32) == 0);
    LinearCodePCS<Bn254> pcs;                                      // Blake2s columns, SHA-256 tree, length-prefixed leaves
    LinCodePCCommitmentState<Bn254> st;
    const size_t n_leaves = 8;
    st.leaves.resize(n_leaves * 32);
    for (size_t i = 0; i < st.leaves.size(); i++) st.leaves[i] = (uint8_t)(i * 37 + 11);
    st.nodes.assign((n_leaves - 1) * 32, 0);
    auto node = [&](size_t i) { return st.nodes.data() + i * 32; };
    for (size_t i = 0; i < n_leaves / 2; i++) {                      // bottom level: D(conv(l) || conv(r)), conv = u64 length || bytes
      uint8_t buf[80]; uint64_t len = 32;
      memcpy(buf, &len, 8); memcpy(buf + 8, &st.leaves[(2 * i) * 32], 32); memcpy(buf + 40, &len, 8); memcpy(buf + 48, &st.leaves[(2 * i + 1) * 32], 32);
      Sha256Host::digest(buf, 80, node(n_leaves / 2 - 1 + i));
    }
    for (size_t i = n_leaves / 2 - 1; i-- > 0;) { uint8_t buf[64]; memcpy(buf, node(2 * i + 1), 32); memcpy(buf + 32, node(2 * i + 2), 32); Sha256Host::digest(buf, 64, node(i)); }
    for (size_t idx = 0; idx < n_leaves; idx++) {
      uint8_t sib[32]; std::vector<uint8_t> path;
      LinearCodePCS<Bn254>::merkle_path(st, idx, sib, path);
      CHECK(path.size() == 2 * 32);
      CHECK(pcs.verify_path(node(0), &st.leaves[idx * 32], idx, sib, path));
      uint8_t bad[32]; memcpy(bad, &st.leaves[idx * 32], 32); bad[0] ^= 1;
      CHECK(!pcs.verify_path(node(0), bad, idx, sib, path));
      CHECK(!pcs.verify_path(node(0), &st.leaves[idx * 32], idx ^ 2, sib, path));
    }
  }
  { // tensors and check polynomials on small integers
    typedef FrT<Pallas> Fr;
    auto f = [](uint64_t v) { return Fr::from_u64(v); };
    std::vector<Fr> a, b;
    LinearCodePCS<Pallas>::tensor(f(3), 3, 2, a, b);                                  // ((1, z, z^2), (1, z^3))
    CHECK(a.size() == 3 && b.size() == 2 && a[0] == f(1) && a[1] == f(3) && a[2] == f(9) && b[0] == f(1) && b[1] == f(27));
    const Fr vals[2] = {f(3), f(5)};
    std::vector<Fr> tp = HyraxPC<Pallas>::tensor_prime(vals, 2);                      // first value in the top bit
    CHECK(tp.size() == 4 && tp[0] == (Fr::one() - f(3)) * (Fr::one() - f(5)) && tp[1] == (Fr::one() - f(3)) * f(5) &&
          tp[2] == f(3) * (Fr::one() - f(5)) && tp[3] == f(15));
    // SuccinctCheckPolynomial::evaluate against the product of its coefficient form: prod (1 + u_i x^(2^(log_d - i)))
    std::vector<Fr> u = {f(7), f(11), f(13)};
    const Fr x = f(2);
    Fr want = (Fr::one() + u[0] * f(16)) * (Fr::one() + u[1] * f(4)) * (Fr::one() + u[2] * f(2));
    CHECK(InnerProductArgPC<Pallas>::check_poly_evaluate(u, x) == want);
  }
  printf("host logic OK (calculate_t bounds of linear_codes/utils.rs:344-359, Ligero shape table, InvalidParameters, SHA-256 / Merkle paths, tensors)\n");
}

int main() {
  host_logic_checks();
  pc_ctx* ctx = nullptr;
  int rc = pc_hip_init(0, &ctx);
  if (rc != PC_OK) { printf("pc_hip_init failed: %s\n", pc_hip_strerror(rc)); return rc == PC_ERR_NO_DEVICE ? 77 : 1; }
  run<Bls12_381>(ctx, "bls12_381");
  run<Bn254>(ctx, "bn254");
  run<Pallas>(ctx, "pallas");
  pc_hip_shutdown(ctx);
  return 0;
}
