// Elementwise kernels of the InnerProductArgPC halving rounds
// (poly-commit/src/ipa_pc/mod.rs:664-711); the two MSMs per round go through msm.hpp.
//   fr_fold   c_l[i] += s * c_r[i]                    ipa_pc/mod.rs:691-697 (coeffs with u^-1, z with u)
//   fr_dot    <a, b>                                   utils.rs:150-155 (inner_product), used at :672,:675
//   ec_fold   k_l[i] = affine(k_l[i] + u * k_r[i])     ipa_pc/mod.rs:699-707 (key fold + normalize_batch)
//   fr_powers [1, z, z^2, ...]                         ipa_pc/mod.rs:641-649
// The round challenge u is one scalar shared by every lane, so the double-and-add ladder of
// ec_fold is branch-uniform across the wave.
#pragma once
#include "ec.hpp"

namespace pc {

template <class FrP>
struct FrFoldBody {
  typedef Fd<FrP> F;
  uint32_t* lo; const uint32_t* hi; F s;
  PC_HD void operator()(uint32_t i) const {
    F a = F::load(lo + (size_t)i * FrP::N), b = F::load(hi + (size_t)i * FrP::N);
    a.add(s.mul(b)).store(lo + (size_t)i * FrP::N);
  }
};

// partial[t] = sum over i = t, t + stride, ... of a[i]*b[i]; the host adds the partials.
template <class FrP>
struct FrDotBody {
  typedef Fd<FrP> F;
  const uint32_t* a; const uint32_t* b; uint32_t n; uint32_t stride; uint32_t* partial;
  PC_HD void operator()(uint32_t t) const {
    F acc = F::zero();
    for (uint32_t i = t; i < n; i += stride)
      acc = acc.add(F::load(a + (size_t)i * FrP::N).mul(F::load(b + (size_t)i * FrP::N)));
    acc.store(partial + (size_t)t * FrP::N);
  }
};

// partial[t] = sum over i = t, t + stride, ... of a[i]  (second stage of fr_dot)
template <class FrP>
struct FrSumBody {
  typedef Fd<FrP> F;
  const uint32_t* a; uint32_t n; uint32_t stride; uint32_t* partial;
  PC_HD void operator()(uint32_t t) const {
    F acc = F::zero();
    for (uint32_t i = t; i < n; i += stride) acc = acc.add(F::load(a + (size_t)i * FrP::N));
    acc.store(partial + (size_t)t * FrP::N);
  }
};

template <class FrP>
struct FrPowTable { uint32_t w[32][FrP::N]; };   // w[k] = z^(2^k)

template <class FrP>
struct FrPowersBody {
  typedef Fd<FrP> F;
  uint32_t* out; FrPowTable<FrP> pt;
  PC_HD void operator()(uint32_t i) const {
    F acc = F::one();
    for (uint32_t k = 0; (i >> k) != 0; k++)
      if ((i >> k) & 1) acc = acc.mul(F::load(pt.w[k]));
    acc.store(out + (size_t)i * FrP::N);
  }
};

// Late halving rounds without touching the key (ipa_pc/mod.rs:664-711 for n <= n0): the resident key K0 of the
// round where the prover switches stays fixed; the key the reference would hold after further folds is
//   key_i = sum over j = i (mod m) of s_j * K0_j,      s_j = product of the challenges whose fold put K0_j on
// the "times u" side, so the round's two commitments are MSMs over K0 with combined scalars,
//   L = sum_{j mod m <  h} (c[h + j mod m] * s_j) K0_j,     R = sum_{j mod m >= h} (c[j mod m - h] * s_j) K0_j,
// and final_comm_key = sum_j s_j K0_j.  A fold of n/2 full scalar multiplications is latency-bound at ~2.4 ms
// however small n gets; two more MSMs of n0 pairs are not.
template <class FrP>
struct IpaKeyScalarUpdateBody {      // the fold by u at size m: s_j *= u where (j mod m) >= m/2
  typedef Fd<FrP> F;
  uint32_t* s; uint32_t m; F u;
  PC_HD void operator()(uint32_t j) const {
    if ((j & (m - 1)) >= (m >> 1)) F::load(s + (size_t)j * FrP::N).mul(u).store(s + (size_t)j * FrP::N);
  }
};
template <class FrP>
struct IpaFixedKeyScalarsBody {      // scalar vectors of L and R for the round at size m
  typedef Fd<FrP> F;
  const uint32_t* c; const uint32_t* s; uint32_t m; uint32_t* out_l; uint32_t* out_r;
  PC_HD void operator()(uint32_t j) const {
    const uint32_t i = j & (m - 1), h = m >> 1;
    const F sj = F::load(s + (size_t)j * FrP::N);
    const F v = F::load(c + (size_t)(i < h ? h + i : i - h) * FrP::N).mul(sj);
    (i < h ? v : F::zero()).store(out_l + (size_t)j * FrP::N);
    (i < h ? F::zero() : v).store(out_r + (size_t)j * FrP::N);
  }
};

// One round's vector work in ONE pass (ipa_pc/mod.rs:672,675 and :691-697): the optional fold by the previous round's
// challenge at size 2m (c_l += u^-1 c_r, z_l += u z_r) and, on the folded values still in registers, the two inner products
// of the round at size m:  l = <c[m/2..m), z[0..m/2)>,  r = <c[0..m/2), z[m/2..m)>.  Lane t owns positions t and t + m/2 of
// the (folded) vectors; workgroup sums go through LDS, a second one-workgroup kernel adds them: 64 bytes come back.
// (Round 2: two fr_dot calls -- two launches, a blocking download of 256 partials and a host fold each -- plus two fr_fold
// launches per round: 18 + 1 ms of host time over the 22 rounds of an opening at 2^22.)
#if defined(__HIPCC__)
template <class FrP>
__global__ void __launch_bounds__(256) k_ipa_fold_dots(uint32_t* c, uint32_t* z, uint32_t m, uint32_t fold, Fd<FrP> u, Fd<FrP> ui,
                                                       uint32_t* partial) {
  typedef Fd<FrP> F;
  constexpr int N = FrP::N;
  __shared__ uint32_t red[2][256][N];
  const uint32_t q = m >> 1;
  F al = F::zero(), ar = F::zero();
  if (q == 0) {                       // m == 1: the last fold, no inner products left
    if (fold && blockIdx.x == 0 && threadIdx.x == 0) {
      F::load(c).add(ui.mul(F::load(c + N))).store(c);
      F::load(z).add(u.mul(F::load(z + N))).store(z);
    }
  } else {
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < q; t += gridDim.x * 256) {
      F c0 = F::load(c + (size_t)t * N), c1 = F::load(c + (size_t)(t + q) * N);
      F z0 = F::load(z + (size_t)t * N), z1 = F::load(z + (size_t)(t + q) * N);
      if (fold) {                     // reads at >= m, writes at < m: no lane reads what another writes
        c0 = c0.add(ui.mul(F::load(c + (size_t)(t + m) * N))); c1 = c1.add(ui.mul(F::load(c + (size_t)(t + q + m) * N)));
        z0 = z0.add(u.mul(F::load(z + (size_t)(t + m) * N))); z1 = z1.add(u.mul(F::load(z + (size_t)(t + q + m) * N)));
        c0.store(c + (size_t)t * N); c1.store(c + (size_t)(t + q) * N);
        z0.store(z + (size_t)t * N); z1.store(z + (size_t)(t + q) * N);
      }
      al = al.add(c1.mul(z0)); ar = ar.add(c0.mul(z1));
    }
  }
  al.store(red[0][threadIdx.x]); ar.store(red[1][threadIdx.x]);
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      F::load(red[0][threadIdx.x]).add(F::load(red[0][threadIdx.x + s])).store(red[0][threadIdx.x]);
      F::load(red[1][threadIdx.x]).add(F::load(red[1][threadIdx.x + s])).store(red[1][threadIdx.x]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 2 * N) partial[(size_t)blockIdx.x * 2 * N + threadIdx.x] = threadIdx.x < N ? red[0][0][threadIdx.x] : red[1][0][threadIdx.x - N];
}
template <class FrP>
__global__ void __launch_bounds__(256) k_ipa_dots_final(const uint32_t* partial, uint32_t nblocks, uint32_t* out) {
  typedef Fd<FrP> F;
  constexpr int N = FrP::N;
  __shared__ uint32_t red[2][256][N];
  F al = F::zero(), ar = F::zero();
  for (uint32_t b = threadIdx.x; b < nblocks; b += 256) {
    al = al.add(F::load(partial + (size_t)b * 2 * N)); ar = ar.add(F::load(partial + (size_t)b * 2 * N + N));
  }
  al.store(red[0][threadIdx.x]); ar.store(red[1][threadIdx.x]);
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      F::load(red[0][threadIdx.x]).add(F::load(red[0][threadIdx.x + s])).store(red[0][threadIdx.x]);
      F::load(red[1][threadIdx.x]).add(F::load(red[1][threadIdx.x + s])).store(red[1][threadIdx.x]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 2 * N) out[threadIdx.x] = threadIdx.x < N ? red[0][0][threadIdx.x] : red[1][0][threadIdx.x - N];
}
#endif

// scalar * affine point with a NAF-recoded scalar shared by all lanes (branch-uniform), Jacobian
template <class C, int NW>
PC_HD JacD<C> naf_mul(const NafMasks<NW>& naf, const AffD<C>& p) {
  JacD<C> acc = JacD<C>::infinity();
  const AffD<C> np = p.neg_if(true);
  for (int bit = 32 * (NW + 1) - 1; bit >= 0; bit--) {
    acc = acc.dbl();
    const uint32_t m = 1u << (bit & 31);
    if (naf.pos[bit >> 5] & m) acc.add_affine(p);
    else if (naf.neg[bit >> 5] & m) acc.add_affine(np);
  }
  return acc;
}

template <class C>
struct EcFoldBody {
  static constexpr int AW = 2 * Fd<typename C::FqP>::N;
  uint32_t* key;            // affine points; lane i updates key[i] from key[i] and key[half + i]
  uint32_t half;
  NafMasks<C::FrP::N> naf;  // the round challenge u, NAF-recoded on the host
  PC_HD void operator()(uint32_t i) const {
    AffD<C> kl = AffD<C>::load(key + (size_t)i * AW), kr = AffD<C>::load(key + (size_t)(half + i) * AW);
    JacD<C> acc = naf_mul<C, C::FrP::N>(naf, kr);
    acc.add_affine(kl);
    acc.to_affine().store(key + (size_t)i * AW);
  }
};

// out[i] = scalars[i] * g for one fixed base g: `g.batch_mul(powers_of_beta)`, the SRS generation
// of KZG10::setup (poly-commit/src/kzg10/mod.rs:76,83).  One lane per scalar (per-lane NAF);
// affine output.  Used to build TRUE structured reference strings for the trapdoor-checked
// end-to-end tests, not on the commit/open path.
template <class C>
struct FixedBaseMulBody {
  static constexpr int AW = 2 * Fd<typename C::FqP>::N;
  const uint32_t* scalars;   // n x Fr, Montgomery
  uint32_t g[AW];
  uint32_t* out;             // n affine points
  PC_HD void operator()(uint32_t i) const {
    typedef Fd<typename C::FrP> Fr;
    Fr k = Fr::load(scalars + (size_t)i * C::FrP::N).from_mont();
    NafMasks<C::FrP::N> naf; naf.from_scalar(k.l);
    naf_mul<C, C::FrP::N>(naf, AffD<C>::load(g)).to_affine().store(out + (size_t)i * AW);
  }
};

// The same batch_mul with a window table of the fixed base, what ark-ec's ScalarMul::batch_mul does
// (`g.batch_mul(&powers_of_beta)`, poly-commit/src/kzg10/mod.rs:76,83): T[w][d-1] = d * 2^(8 w) * g for d = 1..128 and
// the 32 (31 for 254 bits) byte windows of a scalar; a multiplication is then at most one mixed addition per window of the
// signed radix-256 recoding -- no doublings: ~320 field products instead of ~2900 for the per-lane NAF ladder.  The table
// (4096 affine points, 393 KB for BLS12-381) is built on the host per call and read through L2.  Results stay in XYZZ
// and are normalised by XyzzBatchAffineBody with one inversion per K points.
static constexpr uint32_t FIXED_BASE_C = 8;
template <class C>
struct FixedBaseTableMulBody {
  typedef Fd<typename C::FrP> Fr;
  static constexpr int AW = 2 * Fd<typename C::FqP>::N;
  const uint32_t* scalars;   // n x Fr, Montgomery
  const uint32_t* table;     // Wd x 128 affine points
  uint32_t Wd;
  uint32_t* out_xyzz;        // n x XyzzD::WORDS
  PC_HD void operator()(uint32_t i) const {
    ScalarDigits<typename C::FrP> sd; sd.load(scalars + (size_t)i * C::FrP::N, true);
    XyzzD<C> acc = XyzzD<C>::infinity();
    uint32_t carry = 0;
    const uint32_t half = 1u << (FIXED_BASE_C - 1);
    sd.for_each_window(FIXED_BASE_C, Wd, [&](uint32_t w, uint32_t bits) {
      uint32_t raw = bits + carry;
      carry = raw > half;
      const uint32_t mag = carry ? (2 * half - raw) : raw;
      if (mag) acc.add_affine(AffD<C>::load(table + ((size_t)w * half + (mag - 1)) * AW).neg_if(carry != 0));
    });
    acc.store(out_xyzz + (size_t)i * XyzzD<C>::WORDS);
  }
};

// XYZZ -> affine for n points, one inversion per K points (Montgomery's trick along a lane's run; x = X / ZZ, y = Y / ZZZ)
template <class C>
struct XyzzBatchAffineBody {
  typedef Fd<typename C::FqP> Fq;
  typedef XyzzD<C> Pt;
  static constexpr int FN = Fq::N, AW = 2 * FN;
  const uint32_t* in;        // n x Pt::WORDS
  uint32_t* scratch;         // n x Fq
  uint32_t* out;             // n affine points
  uint32_t n, K;
  PC_HD void operator()(uint32_t t) const {
    const uint32_t s = t * K, e = (n - s > K) ? s + K : n;
    Fq run = Fq::one();
    for (uint32_t j = s; j < e; j++) {
      run.store(scratch + (size_t)j * FN);
      const Pt p = Pt::load(in + (size_t)j * Pt::WORDS);
      if (!p.is_inf()) run = run.mul(p.ZZ.mul(p.ZZZ));
    }
    Fq inv = run.inv();
    for (uint32_t j = e; j-- > s;) {
      const Pt p = Pt::load(in + (size_t)j * Pt::WORDS);
      AffD<C> a = AffD<C>::infinity();
      if (!p.is_inf()) {
        const Fq t1 = inv.mul(Fq::load(scratch + (size_t)j * FN));     // 1 / (ZZ * ZZZ)
        inv = inv.mul(p.ZZ.mul(p.ZZZ));
        a.x = p.X.mul(t1.mul(p.ZZZ)); a.y = p.Y.mul(t1.mul(p.ZZ));
      }
      a.store(out + (size_t)j * AW);
    }
  }
};

}  // namespace pc
