// Short-Weierstrass a = 0 group arithmetic for the MSM buckets.
//
// Bases are affine (x, y) in Montgomery form -- the in-memory form of arkworks'
// short_weierstrass::Affine without the `infinity` flag; infinity is encoded as (0, 0),
// which is never on y^2 = x^3 + b with b != 0.  Buckets use XYZZ (extended Jacobian)
// coordinates: x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2; infinity <=> ZZ == 0.  Mixed addition
// costs 8M + 2S (madd-2008-s), full addition 12M + 2S (add-2008-s), doubling 6M + 4S... see
// the Explicit-Formulas Database, "XYZZ coordinates for short Weierstrass curves".
//
// Results are handed back as affine points, which are canonical: any correct schedule is
// bit-identical to ark-ec's msm_bigint followed by into_affine
// (poly-commit/src/kzg10/mod.rs:175-178, :209).
#pragma once
#include "fp32.hpp"

namespace pc {

template <class C>
struct AffD {
  typedef Fd<typename C::FqP> Fq;
  Fq x, y;
  static PC_HD AffD infinity() { AffD a; a.x = Fq::zero(); a.y = Fq::zero(); return a; }
  PC_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  static PC_HD AffD load(const uint32_t* p) { AffD a; a.x = Fq::load(p); a.y = Fq::load(p + Fq::N); return a; }
  PC_HD void store(uint32_t* p) const { x.store(p); y.store(p + Fq::N); }
  PC_HD AffD neg_if(bool s) const { AffD a; a.x = x; a.y = s ? y.neg() : y; return a; }
};

template <class C>
struct XyzzD {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int WORDS = 4 * Fq::N;
  Fq X, Y, ZZ, ZZZ;

  static PC_HD XyzzD infinity() { XyzzD r; r.X = Fq::zero(); r.Y = Fq::zero(); r.ZZ = Fq::zero(); r.ZZZ = Fq::zero(); return r; }
  PC_HD bool is_inf() const { return ZZ.is_zero(); }
  static PC_HD XyzzD from_affine(const AffD<C>& a) {
    XyzzD r;
    if (a.is_inf()) return infinity();
    r.X = a.x; r.Y = a.y; r.ZZ = Fq::one(); r.ZZZ = Fq::one(); return r;
  }
  static PC_HD XyzzD load(const uint32_t* p) {
    XyzzD r; r.X = Fq::load(p); r.Y = Fq::load(p + Fq::N); r.ZZ = Fq::load(p + 2 * Fq::N); r.ZZZ = Fq::load(p + 3 * Fq::N); return r;
  }
  PC_HD void store(uint32_t* p) const { X.store(p); Y.store(p + Fq::N); ZZ.store(p + 2 * Fq::N); ZZZ.store(p + 3 * Fq::N); }

  // 2 * (affine point), mdbl-2008-s-1 with a = 0.
  static PC_HD XyzzD dbl_affine(const AffD<C>& a) {
    if (a.is_inf() || a.y.is_zero()) return infinity();
    XyzzD r;
    Fq U = a.y.dbl(), V = U.sqr(), W = U.mul(V), S = a.x.mul(V);
    Fq xx = a.x.sqr(), M = xx.dbl().add(xx);
    r.X = M.sqr().sub(S.dbl());
    r.Y = M.mul_add_mul(S.sub(r.X), W.neg(), a.y);
    r.ZZ = V; r.ZZZ = W;
    return r;
  }
  // dbl-2008-s-1 with a = 0.
  PC_HD XyzzD dbl() const {
    if (is_inf() || Y.is_zero()) return infinity();
    XyzzD r;
    Fq U = Y.dbl(), V = U.sqr(), W = U.mul(V), S = X.mul(V);
    Fq xx = X.sqr(), M = xx.dbl().add(xx);
    r.X = M.sqr().sub(S.dbl());
    r.Y = M.mul_add_mul(S.sub(r.X), W.neg(), Y);
    r.ZZ = V.mul(ZZ); r.ZZZ = W.mul(ZZZ);
    return r;
  }
  // this += affine (madd-2008-s), all special cases handled.
  PC_HD void add_affine(const AffD<C>& a) {
    if (a.is_inf()) return;
    if (is_inf()) { X = a.x; Y = a.y; ZZ = Fq::one(); ZZZ = Fq::one(); return; }
    Fq U2 = a.x.mul(ZZ), S2 = a.y.mul(ZZZ);
    Fq Pp = U2.sub(X), R = S2.sub(Y);
    if (Pp.is_zero()) {
      if (R.is_zero()) *this = dbl_affine(a); else *this = infinity();
      return;
    }
    Fq PP = Pp.sqr(), PPP = Pp.mul(PP), Q = X.mul(PP);
    Fq X3 = R.sqr().sub(PPP).sub(Q.dbl());
    Y = R.mul_add_mul(Q.sub(X3), Y.neg(), PPP);        // R (Q - X3) - Y PPP, one reduction for the pair
    X = X3;
    ZZ = ZZ.mul(PP); ZZZ = ZZZ.mul(PPP);
  }
  // The same mixed addition with LAZY coordinates (fp32.hpp, LAZY_OK): X, Y, ZZ, ZZZ stay in [0, 2p) between additions and
  // none of the nine multiplier calls ends in a conditional subtraction (BN254: the fused pair keeps its one); the affine operand is canonical, its negation
  // (`negate`: the sign of the signed digit) is p - y without a zero special case.  Infinity stays the exact ZZ == 0: a
  // product ZZ * PP = 0 (mod p) needs P = 0 (mod p), which takes the branch below.  `canonical()` where the sum leaves the
  // chain (bucket store, partial list).
  PC_HD void add_affine_lz(const AffD<C>& a, bool negate) {
    if (a.y.is_zero()) { if (a.x.is_zero()) return; }        // infinity = (0, 0); y alone decides for every point of the curve (no 2-torsion)
    const Fq ay = negate ? a.y.neg_lz_canonical() : a.y;
    if (is_inf()) { X = a.x; Y = ay; ZZ = Fq::one(); ZZZ = Fq::one(); return; }
    Fq U2 = a.x.mul_lz(ZZ), S2 = ay.mul_lz(ZZZ);
    Fq Pp = U2.sub_lz(X), R = S2.sub_lz(Y);
    if (Pp.is_zero_lz()) {             // same x: doubling or P + (-P) -- the canonical code (rare)
      XyzzD t = canonical();
      AffD<C> b; b.x = a.x; b.y = ay.canon();
      t.add_affine(b);
      *this = t;
      return;
    }
    Fq PP = Pp.sqr_lz(), PPP = Pp.mul_lz(PP), Q = X.mul_lz(PP);
    Fq X3 = R.sqr_lz().sub_lz(PPP).sub_lz(Q.dbl_lz());
    Y = R.mul_add_mul_lz(Q.sub_lz(X3), Y.neg_lz(), PPP);      // R (Q - X3) + (2p - Y) PPP <= 8 p^2: one reduction, below 2p (R >= 8p), or
                                                              // below 3p and one conditional subtraction (BN254)
    X = X3;
    ZZ = ZZ.mul_lz(PP); ZZZ = ZZZ.mul_lz(PPP);
  }
  // (every coordinate of a lazy sum is a multiplier output, a sub_lz result or a canonical input: strictly below 2p)
  PC_HD XyzzD canonical() const { XyzzD r; r.X = X.canon1(); r.Y = Y.canon1(); r.ZZ = ZZ.canon1(); r.ZZZ = ZZZ.canon1(); return r; }
  // this += o (add-2008-s), all special cases handled.
  PC_HD void add(const XyzzD& o) {
    if (o.is_inf()) return;
    if (is_inf()) { *this = o; return; }
    Fq U1 = X.mul(o.ZZ), U2 = o.X.mul(ZZ), S1 = Y.mul(o.ZZZ), S2 = o.Y.mul(ZZZ);
    Fq Pp = U2.sub(U1), R = S2.sub(S1);
    if (Pp.is_zero()) {
      if (R.is_zero()) *this = dbl(); else *this = infinity();
      return;
    }
    Fq PP = Pp.sqr(), PPP = Pp.mul(PP), Q = U1.mul(PP);
    Fq X3 = R.sqr().sub(PPP).sub(Q.dbl());
    Y = R.mul_add_mul(Q.sub(X3), S1.neg(), PPP);       // one reduction for the pair
    X = X3;
    ZZ = ZZ.mul(o.ZZ).mul(PP); ZZZ = ZZZ.mul(o.ZZZ).mul(PPP);
  }
  PC_HD AffD<C> to_affine() const {
    if (is_inf()) return AffD<C>::infinity();
    // x = X / ZZ, y = Y / ZZZ ; one inversion of ZZ*ZZZ
    Fq t = ZZ.mul(ZZZ).inv();
    AffD<C> a; a.x = X.mul(t.mul(ZZZ)); a.y = Y.mul(t.mul(ZZ)); return a;
  }
};


// ---- one XYZZ addition spread over TWO lanes (msm_coop.hpp, k_bucket_level_coop2) ------------------------------------------
// The even lane holds (X, ZZ) of a point, the odd lane (Y, ZZZ).  XyzzD::add is laid out so that both lanes run the same
// instruction stream on their halves, in four phases with a neighbour exchange after the first three:
//   p1  A1*B2, A2*B1, D = difference          U1, U2, P                 | S1, S2, R            -> exchange "D == 0"
//   p2  D^2, B1*B2, D*DD                      PP, ZZ1*ZZ2, PPP          | RR, ZZZ1*ZZZ2, -     -> even sends PPP, odd sends RR
//   p3  one product, X3                       Q = U1*PP, X3             | ZZZ3 = ZZZ12*PPP     -> even sends Q - X3
//   p4  one fused pair                        ZZ3 = ZZ12*PP (+ 0*0)     | Y3 = R*(Q - X3) - S1*PPP
// 7.3 multiplication times instead of 13; the values are those of XyzzD::add, bit for bit.  The phases are plain PC_HD code
// (the device exchanges by DPP, tests/emu steps the two lanes on the host).
template <class C>
struct HalfPt {
  typedef Fd<typename C::FqP> Fq;
  Fq a, b;        // even lane: X, ZZ   odd lane: Y, ZZZ
  static PC_HD HalfPt of(const XyzzD<C>& p, bool odd) { HalfPt h; h.a = odd ? p.Y : p.X; h.b = odd ? p.ZZZ : p.ZZ; return h; }
};
template <class Fq>
PC_HD Fq fq_sel(bool c, const Fq& x, const Fq& y) {
  Fq r;
  PC_UNROLL for (int i = 0; i < Fq::N; i++) r.l[i] = c ? x.l[i] : y.l[i];
  return r;
}
template <class C>
struct HalfAdd {
  typedef Fd<typename C::FqP> Fq;
  Fq T1, D, DD, BB, T5, rc1, R6, X3;
  PC_HD bool p1(const HalfPt<C>& p, const HalfPt<C>& o) { T1 = p.a.mul(o.b); const Fq T2 = o.a.mul(p.b); D = T2.sub(T1); return D.is_zero(); }
  PC_HD Fq p2(const HalfPt<C>& p, const HalfPt<C>& o, bool odd) { DD = D.sqr(); BB = p.b.mul(o.b); T5 = D.mul(DD); return fq_sel(odd, DD, T5); }
  PC_HD Fq p3(bool odd, const Fq& recv) {
    rc1 = recv;                                                       // even: RR, odd: PPP
    R6 = fq_sel(odd, BB, T1).mul(fq_sel(odd, rc1, DD));               // even: Q, odd: ZZZ3
    X3 = rc1.sub(T5).sub(R6.dbl());                                   // even: RR - PPP - 2Q
    return R6.sub(X3);                                                // even: Q - X3
  }
  PC_HD void p4(HalfPt<C>& p, bool odd, const Fq& rc2) {
    const Fq R7 = fq_sel(odd, D, BB).mul_add_mul(fq_sel(odd, rc2, DD), fq_sel(odd, T1.neg(), Fq::zero()), fq_sel(odd, rc1, Fq::zero()));
    p.a = fq_sel(odd, R7, X3);                                        // even: X3, odd: Y3
    p.b = fq_sel(odd, R6, R7);                                        // even: ZZ3, odd: ZZZ3
  }
};

// Jacobian coordinates (x = X/Z^2, y = Y/Z^3), a = 0: cheaper doubling (2M + 5S, dbl-2009-l) than
// XYZZ (6M + 3S).  Used by the per-element scalar multiplications (IPA key fold, fixed-base SRS
// generation), where doublings outnumber additions 3:1 after NAF recoding.
template <class C>
struct JacD {
  typedef Fd<typename C::FqP> Fq;
  Fq X, Y, Z;
  static PC_HD JacD infinity() { JacD r; r.X = Fq::one(); r.Y = Fq::one(); r.Z = Fq::zero(); return r; }
  PC_HD bool is_inf() const { return Z.is_zero(); }
  PC_HD JacD dbl() const {
    if (is_inf() || Y.is_zero()) return infinity();
    Fq A = X.sqr(), B = Y.sqr(), Cc = B.sqr();
    Fq t = X.add(B).sqr().sub(A).sub(Cc);
    Fq D = t.dbl(), E = A.dbl().add(A), F = E.sqr();
    JacD r;
    r.X = F.sub(D.dbl());
    r.Z = Y.mul(Z).dbl();
    r.Y = E.mul(D.sub(r.X)).sub(Cc.dbl().dbl().dbl());
    return r;
  }
  // this += affine (madd-2007-bl), all special cases handled
  PC_HD void add_affine(const AffD<C>& a) {
    if (a.is_inf()) return;
    if (is_inf()) { X = a.x; Y = a.y; Z = Fq::one(); return; }
    Fq Z1Z1 = Z.sqr(), U2 = a.x.mul(Z1Z1), S2 = a.y.mul(Z).mul(Z1Z1);
    Fq H = U2.sub(X), rr = S2.sub(Y);
    if (H.is_zero()) {
      if (rr.is_zero()) *this = dbl(); else *this = infinity();
      return;
    }
    rr = rr.dbl();
    Fq HH = H.sqr(), I = HH.dbl().dbl(), J = H.mul(I), V = X.mul(I);
    Fq X3 = rr.sqr().sub(J).sub(V.dbl());
    Fq Y3 = rr.mul_add_mul(V.sub(X3), Y.dbl().neg(), J);   // r (V - X3) - 2 Y1 J, one reduction for the pair
    Z = Z.add(H).sqr().sub(Z1Z1).sub(HH);
    X = X3; Y = Y3;
  }
  PC_HD AffD<C> to_affine() const {
    if (is_inf()) return AffD<C>::infinity();
    Fq zi = Z.inv(), zi2 = zi.sqr();
    AffD<C> a; a.x = X.mul(zi2); a.y = Y.mul(zi2).mul(zi); return a;
  }
};

// Jacobian -> affine for n points with ONE field inversion per lane instead of one per point (Montgomery's
// trick, what CurveGroup::normalize_batch does in the reference, ipa_pc/mod.rs:706-708): lane t owns points
// [t K, (t+1) K), multiplies their Z into a running product (prefixes parked in `scratch`), inverts once and
// peels the inverses off backwards: 1 + 3 products per point for 1/Z plus 1S + 3M to normalise, and 1/K of a
// Fermat inversion (~1.5 N^2.. i.e. ~380 products) -- against ~390 per point when every lane inverts alone.
template <class C>
struct JacBatchAffineBody {
  typedef Fd<typename C::FqP> Fq;
  static constexpr int FN = Fq::N, AW = 2 * FN;
  const uint32_t* jac;      // n x (X, Y, Z); Z == 0: infinity
  uint32_t* scratch;        // n x Fq
  uint32_t* out;            // n affine points, (0,0) = infinity
  uint32_t n, K;
  uint32_t out_stride = AW; // words between consecutive output points (>= AW)
  PC_HD void operator()(uint32_t t) const {
    const uint32_t s = t * K, e = (n - s > K) ? s + K : n;
    Fq run = Fq::one();
    for (uint32_t j = s; j < e; j++) {
      run.store(scratch + (size_t)j * FN);
      const Fq z = Fq::load(jac + (size_t)j * 3 * FN + 2 * FN);
      if (!z.is_zero()) run = run.mul(z);
    }
    Fq inv = run.inv();
    for (uint32_t j = e; j-- > s;) {
      const uint32_t* p = jac + (size_t)j * 3 * FN;
      const Fq z = Fq::load(p + 2 * FN);
      AffD<C> a = AffD<C>::infinity();
      if (!z.is_zero()) {
        const Fq zi = inv.mul(Fq::load(scratch + (size_t)j * FN));
        inv = inv.mul(z);
        const Fq zi2 = zi.sqr();
        a.x = Fq::load(p).mul(zi2); a.y = Fq::load(p + FN).mul(zi2).mul(zi);
      }
      a.store(out + (size_t)j * out_stride);
    }
  }
};

// Non-adjacent form of a canonical scalar as two bit masks (digit +1 / -1 per position):
// on average a third of the digits are non-zero (binary: a half).  NW = limbs of the scalar.
template <int NW>
struct NafMasks {
  uint32_t pos[NW + 1], neg[NW + 1];
  PC_HD void from_scalar(const uint32_t* k) {
    uint32_t t[NW + 1];
    for (int i = 0; i < NW; i++) t[i] = k[i];
    t[NW] = 0;
    for (int i = 0; i <= NW; i++) { pos[i] = 0; neg[i] = 0; }
    for (int bit = 0; bit < 32 * (NW + 1); bit++) {
      bool any = false; for (int i = 0; i <= NW; i++) any |= t[i] != 0;
      if (!any) break;
      if (t[0] & 1) {
        if ((t[0] & 3) == 3) {          // digit -1: t += 1
          neg[bit >> 5] |= 1u << (bit & 31);
          uint64_t c = 1; for (int i = 0; i <= NW; i++) { c += t[i]; t[i] = (uint32_t)c; c >>= 32; }
        } else {                        // digit +1: t -= 1
          pos[bit >> 5] |= 1u << (bit & 31);
          t[0] -= 1;
        }
      }
      for (int i = 0; i < NW; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
      t[NW] >>= 1;
    }
  }
};

}  // namespace pc
