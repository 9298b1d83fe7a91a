// gfx950 instruction-rate and field-arithmetic micro-benchmarks (design input for the
// Montgomery multiplier; not part of the product library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "../poly_commit_amd/csrc/ec.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = seed * 2654435761u + t, b = a ^ 0x9e3779b9u;
  uint64_t acc[8];
  uint32_t x[8];
  for (int i = 0; i < 8; i++) { acc[i] = a + i; x[i] = b + i * 7; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) acc[i] = (uint64_t)x[i] * (uint32_t)acc[i] + acc[i];                 // v_mad_u64_u32
      if (OP == 1) x[i] = x[i] * (uint32_t)acc[i] + 1;                                   // v_mul_lo_u32 (+add)
      if (OP == 2) x[i] = __umulhi(x[i], (uint32_t)acc[i]) + x[i];                       // v_mul_hi_u32
      if (OP == 3) { uint64_t s = acc[i] + ((uint64_t)x[i] << 32 | x[i]); acc[i] = s; } // 64-bit add (add_co + addc)
      if (OP == 4) x[i] = __umul24(x[i], (uint32_t)acc[i]) + x[i];       // v_mad_u32_u24
      if (OP == 5) x[i] = x[i] + (uint32_t)acc[i];                                       // v_add_u32
      if (OP == 6) { double d = __longlong_as_double(acc[i] | 0x3ff0000000000000ull); d = __builtin_fma(d, 1.0000001, 0.5); acc[i] = __double_as_longlong(d); }  // v_fma_f64
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ x[i];
  out[t] = r;
}

template <class P>
__global__ void __launch_bounds__(256) k_fmul(uint32_t* out, int iters) {
  typedef pc::Fd<P> F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = F::one(), y = F::one();
  x.l[0] += t; y.l[1] ^= t * 77;
  for (int it = 0; it < iters; it++) { x = x.mul(y); y = y.mul(x); }
  uint32_t r = 0; for (int i = 0; i < P::N; i++) r ^= x.l[i] ^ y.l[i];
  out[t] = r;
}
template <class P>
__global__ void __launch_bounds__(256) k_fsqr(uint32_t* out, int iters) {
  typedef pc::Fd<P> F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = F::one(), y = F::one();
  x.l[0] += t; y.l[1] ^= t * 77;
  uint32_t bad = 0;
  for (int it = 0; it < iters; it++) { x = x.sqr(); y = y.sqr(); }
  // cross-check of the dedicated squaring against the general product on the last values
  F a = x.mul(y), b = a.sqr(), c = a.mul(a);
  for (int i = 0; i < P::N; i++) bad |= b.l[i] ^ c.l[i];
  uint32_t r = bad ? 0xdeadbeefu : 0; for (int i = 0; i < P::N; i++) r ^= x.l[i] ^ y.l[i];
  out[t] = bad ? 0xdeadbeefu : (r == 0xdeadbeefu ? 0 : r);
}
template <class P>
__global__ void __launch_bounds__(256) k_fdual(uint32_t* out, int iters) {
  typedef pc::Fd<P> F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = F::one(), y = F::one(), u = F::one(), v = F::one();
  x.l[0] += t; y.l[1] ^= t * 77; u.l[2] += t * 3; v.l[0] ^= t * 5;
  for (int it = 0; it < iters; it++) { x = x.mul_add_mul(y, u, v); u = u.mul_add_mul(x, y, v); }
  // cross-check of the fused pair against two products and an addition
  F a = x.mul_add_mul(y, u, v), b = x.mul(y).add(u.mul(v));
  uint32_t bad = 0;
  for (int i = 0; i < P::N; i++) bad |= a.l[i] ^ b.l[i];
  uint32_t r = 0; for (int i = 0; i < P::N; i++) r ^= x.l[i] ^ u.l[i];
  out[t] = bad ? 0xdeadbeefu : (r == 0xdeadbeefu ? 0 : r);
}
template <class P>
__global__ void __launch_bounds__(256) k_fadd(uint32_t* out, int iters) {
  typedef pc::Fd<P> F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = F::one(), y = F::one();
  x.l[0] += t; y.l[1] ^= t * 77;
  for (int it = 0; it < iters; it++) { x = x.add(y); y = y.sub(x); }
  uint32_t r = 0; for (int i = 0; i < P::N; i++) r ^= x.l[i] ^ y.l[i];
  out[t] = r;
}
// memory-free radix-2 butterfly loop of the NTT (csrc/ntt.hpp lds_ntt_stages): (x, y) -> (x + w y, x - w y), one twiddle product, one
// modular addition, one modular subtraction; the twiddle walks through a short register-resident cycle so that no product is constant
template <class P>
__global__ void __launch_bounds__(256) k_butterfly(uint32_t* out, int iters) {
  typedef pc::Fd<P> F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F x = F::one(), y = F::one(), w = F::one();
  x.l[0] += t; y.l[1] ^= t * 77; w.l[2] += t * 13 + 5;
  for (int it = 0; it < iters; it++) {
    const F v = y.mul(w);
    const F a = x.add(v), b = x.sub(v);
    x = a; y = b;
    w.l[0] ^= (uint32_t)it;          // (keeps the compiler from hoisting anything; still a valid limb pattern below 2^255 for every field here)
  }
  uint32_t r = 0; for (int i = 0; i < P::N; i++) r ^= x.l[i] ^ y.l[i];
  out[t] = r;
}
// the radix-4 group of two stages the kernel actually runs per LDS round trip: four elements, four twiddle products, eight additions
template <class P>
__global__ void __launch_bounds__(256) k_butterfly4(uint32_t* out, int iters) {
  typedef pc::Fd<P> F;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F x0 = F::one(), x1 = F::one(), x2 = F::one(), x3 = F::one(), w1 = F::one(), w2 = F::one(), w3 = F::one();
  x0.l[0] += t; x1.l[1] ^= t * 77; x2.l[2] += t * 3; x3.l[3] ^= t * 5; w1.l[2] += t * 13 + 5; w2.l[1] += t + 9; w3.l[0] += 2 * t + 1;
  for (int it = 0; it < iters; it++) {
    x1 = x1.mul(w1); x3 = x3.mul(w1);
    F a0 = x0.add(x1), a1 = x0.sub(x1), a2 = x2.add(x3), a3 = x2.sub(x3);
    a2 = a2.mul(w2); a3 = a3.mul(w3);
    x0 = a0.add(a2); x2 = a0.sub(a2); x1 = a1.add(a3); x3 = a1.sub(a3);
    w1.l[0] ^= (uint32_t)it;
  }
  uint32_t r = 0; for (int i = 0; i < P::N; i++) r ^= x0.l[i] ^ x1.l[i] ^ x2.l[i] ^ x3.l[i];
  out[t] = r;
}
// memory-free loops of the IPA key fold's ladder (csrc/glv.hpp EcFoldGlvBody): Jacobian doublings, Jacobian += affine
template <class C>
__global__ void __launch_bounds__(256) k_jac_dbl(uint32_t* out, const uint32_t* pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int AW = 2 * C::FqP::N;
  pc::JacD<C> acc = pc::JacD<C>::infinity();
  acc.add_affine(pc::AffD<C>::load(pts + (size_t)(t % npts) * AW));
  for (int it = 0; it < iters; it++) acc = acc.dbl();
  uint32_t r = 0; for (int i = 0; i < C::FqP::N; i++) r ^= acc.X.l[i] ^ acc.Z.l[i];
  out[t] = r;
}
template <class C>
__global__ void __launch_bounds__(256) k_jac_madd(uint32_t* out, const uint32_t* pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int AW = 2 * C::FqP::N;
  pc::JacD<C> acc = pc::JacD<C>::infinity();
  for (int it = 0; it < iters; it++) acc.add_affine(pc::AffD<C>::load(pts + (size_t)((t * 31 + it) % npts) * AW));
  uint32_t r = 0; for (int i = 0; i < C::FqP::N; i++) r ^= acc.X.l[i] ^ acc.Z.l[i];
  out[t] = r;
}

template <class C>
__global__ void __launch_bounds__(256) k_madd(uint32_t* out, const uint32_t* pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  pc::XyzzD<C> acc = pc::XyzzD<C>::infinity();
  constexpr int AW = 2 * C::FqP::N;
  for (int it = 0; it < iters; it++) {
    pc::AffD<C> p = pc::AffD<C>::load(pts + (size_t)((t * 31 + it) % npts) * AW);
    acc.add_affine(p);
  }
  uint32_t r = 0; for (int i = 0; i < C::FqP::N; i++) r ^= acc.X.l[i] ^ acc.ZZ.l[i];
  out[t] = r;
}

// the same loop held to two waves per SIMD (what the accumulate kernel runs at): how much of the 3-wave rate is left
template <class C>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_madd_2waves(uint32_t* out, const uint32_t* pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  pc::XyzzD<C> acc = pc::XyzzD<C>::infinity();
  constexpr int AW = 2 * C::FqP::N;
  for (int it = 0; it < iters; it++) {
    pc::AffD<C> p = pc::AffD<C>::load(pts + (size_t)((t * 31 + it) % npts) * AW);
    acc.add_affine(p);
  }
  uint32_t r = 0; for (int i = 0; i < C::FqP::N; i++) r ^= acc.X.l[i] ^ acc.ZZ.l[i];
  out[t] = r;
}

// the lazily reduced form of the same loop (ec.hpp add_affine_lz: no conditional subtraction behind the nine multiplier
// calls; what the accumulate kernel runs for BLS12-381); `check` = the canonical x-coordinate hash, to compare with k_madd
template <class C>
__global__ void __launch_bounds__(256) k_madd_lazy(uint32_t* out, const uint32_t* pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  pc::XyzzD<C> acc = pc::XyzzD<C>::infinity();
  constexpr int AW = 2 * C::FqP::N;
  for (int it = 0; it < iters; it++) {
    pc::AffD<C> p = pc::AffD<C>::load(pts + (size_t)((t * 31 + it) % npts) * AW);
    acc.add_affine_lz(p, false);
  }
  acc = acc.canonical();
  uint32_t r = 0; for (int i = 0; i < C::FqP::N; i++) r ^= acc.X.l[i] ^ acc.ZZ.l[i];
  out[t] = r;
}


template <class K>
static float timeit(K launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}

// memory-free loops of the bucket accumulation's mixed addition for one curve: canonical, the same at 2 and 4 waves per SIMD, and
// (where the field has the headroom, fp32.hpp LAZY_OK) lazily reduced -- checked lane by lane against the canonical loop
template <class C, int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) k_madd_waves(uint32_t* out, const uint32_t* pts, int npts, int iters) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  pc::XyzzD<C> acc = pc::XyzzD<C>::infinity();
  constexpr int AW = 2 * C::FqP::N;
  for (int it = 0; it < iters; it++) {
    pc::AffD<C> p = pc::AffD<C>::load(pts + (size_t)((t * 31 + it) % npts) * AW);
    if constexpr (pc::Fd<typename C::FqP>::LAZY_OK) acc.add_affine_lz(p, false); else acc.add_affine(p);
  }
  if constexpr (pc::Fd<typename C::FqP>::LAZY_OK) acc = acc.canonical();
  uint32_t r = 0; for (int i = 0; i < C::FqP::N; i++) r ^= acc.X.l[i] ^ acc.ZZ.l[i];
  out[t] = r;
}
template <class C>
static int bench_madd(const char* name, uint32_t* out, int blocks, int threads) {
  constexpr int FN = C::FqP::N, AW = 2 * FN;
  const size_t lanes = (size_t)blocks * threads;
  std::vector<uint32_t> h(AW * 64);
  pc::AffD<C> g; for (int i = 0; i < FN; i++) { g.x.l[i] = C::GX[i]; g.y.l[i] = C::GY[i]; }
  pc::XyzzD<C> acc = pc::XyzzD<C>::from_affine(g);
  for (int i = 0; i < 64; i++) { pc::AffD<C> a = acc.to_affine(); a.store(&h[i * AW]); acc.add_affine(g); }
  uint32_t* dp; CHECK(hipMalloc(&dp, h.size() * 4)); CHECK(hipMemcpy(dp, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const int it = FN > 8 ? 64 : 128;
  float ms = timeit([&]() { hipLaunchKernelGGL(k_madd<C>, dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
  printf("XYZZ madd %-12s         %8.3f ms  %8.2f M madd/s\n", name, ms, (double)lanes * it / ms * 1e-3);
  std::vector<uint32_t> ref(lanes), got(lanes);
  CHECK(hipMemcpy(ref.data(), out, lanes * 4, hipMemcpyDeviceToHost));
  ms = timeit([&]() { hipLaunchKernelGGL(k_madd_2waves<C>, dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
  printf("  same, 2 waves per SIMD     %8.3f ms  %8.2f M madd/s\n", ms, (double)lanes * it / ms * 1e-3);
  if constexpr (pc::Fd<typename C::FqP>::LAZY_OK) {
    ms = timeit([&]() { hipLaunchKernelGGL(k_madd_lazy<C>, dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
    CHECK(hipMemcpy(got.data(), out, lanes * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < lanes; i++) bad += got[i] != ref[i];
    printf("  lazily reduced %-12s %8.3f ms  %8.2f M madd/s   (differs from the canonical loop on %zu of %zu lanes)\n", name, ms,
           (double)lanes * it / ms * 1e-3, bad, (size_t)lanes);
  }
  // the kernel's own addition (lazy where allowed) by occupancy
  { float m2 = timeit([&]() { hipLaunchKernelGGL((k_madd_waves<C, 2>), dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
    float m3 = timeit([&]() { hipLaunchKernelGGL((k_madd_waves<C, 3>), dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
    float m4 = timeit([&]() { hipLaunchKernelGGL((k_madd_waves<C, 4>), dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
    printf("  kernel form %-12s at 2 / 3 / 4 waves per SIMD: %8.2f / %8.2f / %8.2f M madd/s\n", name,
           (double)lanes * it / m2 * 1e-3, (double)lanes * it / m3 * 1e-3, (double)lanes * it / m4 * 1e-3); }
  // the ladder of the IPA key fold: Jacobian doubling / Jacobian += affine (bench.py prices EcFoldGlvBody against these two)
  { float md = timeit([&]() { hipLaunchKernelGGL(k_jac_dbl<C>, dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
    float ma = timeit([&]() { hipLaunchKernelGGL(k_jac_madd<C>, dim3(blocks), dim3(threads), 0, 0, out, dp, 64, it); });
    printf("  jacobian %-12s dbl / madd: %8.2f / %8.2f M ops/s\n", name, (double)lanes * it / md * 1e-3, (double)lanes * it / ma * 1e-3); }
  CHECK(hipFree(dp));
  return 0;
}

int main() {
  int ndev = 0; CHECK(hipGetDeviceCount(&ndev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int blocks = 256 * 8, threads = 256;
  const size_t lanes = (size_t)blocks * threads;
  uint32_t* out; CHECK(hipMalloc(&out, lanes * 4));
  const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32(+add)", "v_mul_hi_u32(+add)", "add64 (add_co+addc)", "v_mad_u32_u24", "v_add_u32", "v_fma_f64"};
  const int iters = 2048;
#define RATE(OP) { float ms = timeit([&]() { hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); \
    printf("%-22s %8.3f ms  %8.2f Gops/s (lane-ops)\n", names[OP], ms, (double)lanes * iters * 8 / ms * 1e-6); }
  RATE(0) RATE(1) RATE(2) RATE(3) RATE(4) RATE(5) RATE(6)
  {
    const int it = 256;
    float ms = timeit([&]() { hipLaunchKernelGGL(k_fmul<pc_bls12_381_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); });
    printf("fmul bls12_381_fq (12 limbs) %8.3f ms  %8.2f G mulmod/s\n", ms, (double)lanes * it * 2 / ms * 1e-6);
    ms = timeit([&]() { hipLaunchKernelGGL(k_fmul<pc_bn254_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); });
    printf("fmul bn254_fq (8 limbs)      %8.3f ms  %8.2f G mulmod/s\n", ms, (double)lanes * it * 2 / ms * 1e-6);
    ms = timeit([&]() { hipLaunchKernelGGL(k_fmul<pc_pallas_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); });
    printf("fmul pallas_fq (8 limbs)     %8.3f ms  %8.2f G mulmod/s\n", ms, (double)lanes * it * 2 / ms * 1e-6);
    // the scalar fields (the NTT's and the IPA vector kernels' multiplier)
#define FR_LINE(P, NAME) { ms = timeit([&]() { hipLaunchKernelGGL(k_fmul<P>, dim3(blocks), dim3(threads), 0, 0, out, it); });                 \
    printf("fmul %-16s (8 limbs) %8.3f ms  %8.2f G mulmod/s\n", NAME, ms, (double)lanes * it * 2 / ms * 1e-6);                              \
    ms = timeit([&]() { hipLaunchKernelGGL(k_butterfly<P>, dim3(blocks), dim3(threads), 0, 0, out, 2 * it); });                              \
    printf("  ntt butterfly %-16s      %8.3f ms  %8.2f G butterflies/s   (x + w y, x - w y: 1 product, 1 add, 1 sub)\n", NAME, ms, (double)lanes * it * 2 / ms * 1e-6); \
    ms = timeit([&]() { hipLaunchKernelGGL(k_butterfly4<P>, dim3(blocks), dim3(threads), 0, 0, out, it / 2); });                             \
    printf("  ntt radix-4 group %-16s  %8.3f ms  %8.2f G butterflies/s   (two stages on four elements in registers)\n", NAME, ms, (double)lanes * (it / 2) * 4 / ms * 1e-6); }
    FR_LINE(pc_bls12_381_fr, "bls12_381_fr") FR_LINE(pc_bn254_fr, "bn254_fr") FR_LINE(pc_pallas_fr, "pallas_fr")
    for (int rep = 0; rep < 2; rep++) {
      ms = rep == 0 ? timeit([&]() { hipLaunchKernelGGL(k_fsqr<pc_bls12_381_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); })
                    : timeit([&]() { hipLaunchKernelGGL(k_fsqr<pc_bn254_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); });
      std::vector<uint32_t> h(lanes); CHECK(hipMemcpy(h.data(), out, lanes * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; for (uint32_t v : h) bad += v == 0xdeadbeefu;
      printf("fsqr %s             %8.3f ms  %8.2f G sqrmod/s   (sqr != mul on %zu of %zu lanes)\n", rep == 0 ? "bls12_381_fq" : "bn254_fq    ", ms, (double)lanes * it * 2 / ms * 1e-6, bad, lanes);
    }
    for (int rep = 0; rep < 2; rep++) {
      ms = rep == 0 ? timeit([&]() { hipLaunchKernelGGL(k_fdual<pc_bls12_381_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); })
                    : timeit([&]() { hipLaunchKernelGGL(k_fdual<pc_bn254_fq>, dim3(blocks), dim3(threads), 0, 0, out, it); });
      std::vector<uint32_t> h(lanes); CHECK(hipMemcpy(h.data(), out, lanes * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; for (uint32_t v : h) bad += v == 0xdeadbeefu;
      printf("a*b+c*d %s          %8.3f ms  %8.2f G pairs/s   (fused != mul,mul,add on %zu of %zu lanes)\n", rep == 0 ? "bls12_381_fq" : "bn254_fq    ", ms, (double)lanes * it * 2 / ms * 1e-6, bad, lanes);
    }
    ms = timeit([&]() { hipLaunchKernelGGL(k_fadd<pc_bls12_381_fq>, dim3(blocks), dim3(threads), 0, 0, out, it * 8); });
    printf("fadd+fsub bls12_381_fq       %8.3f ms  %8.2f G addsub/s\n", ms, (double)lanes * it * 8 * 2 / ms * 1e-6);
  }
  if (bench_madd<pc_curve_bls12_381>("bls12_381", out, blocks, threads)) return 1;
  if (bench_madd<pc_curve_bn254>("bn254", out, blocks, threads)) return 1;
  if (bench_madd<pc_curve_pallas>("pallas", out, blocks, threads)) return 1;
  return 0;
}
