// Internal interfaces between the translation units of libpc_hip.so.
//
// The library is built from several .hip files compiled in parallel (poly_commit_amd/build.py):
//   abi.hip            the extern "C" entry points (include/pc_hip.h): lifetime, staging, error translation
//   curve_<name>.hip   everything templated on one curve: MSM pipeline, window table, key fold, fixed-base mul
//   field_<name>.hip   everything templated on one scalar field: NTT, division scan, IPA vector kernels,
//                      column digests
// abi.hip reaches the templates through the two tables of plain function pointers below, one
// instance per curve / field.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/pc_hip.h"
#include "hip_backend.hpp"
#include "msm.hpp"

namespace pc {

struct MsmRunner {
  virtual ~MsmRunner() {}
  virtual void enqueue(const uint32_t* bases, uint32_t base_off, const void* scalars, pc_mem where, size_t n, bool from_mont) = 0;
  // many-MSM runners only: `count` scalar vectors of m elements each in separate device buffers (a batch of polynomials)
  virtual void enqueue_vectors(const uint32_t* bases, uint32_t base_off, const uint64_t* ptrs_host, size_t count, size_t m, bool from_mont) = 0;
  // ONE MSM of n_total pairs in parts (MsmPlan::begin_parts): part k covers scalars [first, first + n) of the call -- host memory: copied on
  // the pipeline's auxiliary queue; device memory: readable now -- against bases base_off + first ..; `last` closes the call
  virtual void begin_parts(size_t n_total) = 0;
  virtual void add_part(const uint32_t* bases, uint32_t base_off, size_t first, const void* scalars_part, pc_mem where, size_t n, bool from_mont, bool last) = 0;
  virtual void finish(uint32_t* out_host) = 0;
  // geometry of the last enqueue: {window bits c, signed digits per scalar, buckets, 1 if the window table was used}
  virtual void shape(uint32_t out[4]) const = 0;
  virtual void trim() = 0;          // give back what can be rebuilt on demand (idle pipeline only)
};

struct NttRunner {
  virtual ~NttRunner() {}
  virtual void run(const uint32_t* in, size_t rows, size_t in_cols, uint32_t* out) = 0;
};

struct CurveOps {
  int aw;                 // 32-bit words per affine point
  uint32_t scalar_bits;
  MsmRunner* (*make_runner)(HipBackend& be, size_t n_max, const MsmConfig& cfg, uint32_t subs);
  void (*window_table)(HipBackend& be, const uint32_t* bases, uint32_t n, uint32_t c, uint32_t Wd, uint32_t* table, uint32_t stride);
  void (*ec_fold)(HipBackend& be, uint32_t* key, size_t half, const uint32_t* u_mont);
  // out[i] = affine(in[i] + u * in[half + i]); table: the key's one-level fold table of width-w NAF digits (or null: GLV ladder).  out == in: in place.
  void (*ec_fold_to)(HipBackend& be, const uint32_t* in, uint32_t* out, size_t half, const uint32_t* u_mont, const uint32_t* table, uint32_t w);
  // out[i] = affine(key_lo[i] + sum_t u_t * P_t[i]) from the fold table (term t = table points [t count, (t + 1) count)); false: does not fit
  bool (*ec_fold_table)(HipBackend& be, const uint32_t* key_lo, uint32_t* out, size_t count, size_t row_pts, uint32_t terms,
                        const uint32_t* const* u_monts, uint32_t w, const uint32_t* table);
  // fold table of `count` key points for width-w NAF digits: 2^(w-2) * fold_rows rows of `count` points
  void (*fold_table_build)(HipBackend& be, const uint32_t* pts, size_t count, uint32_t w, uint32_t* table);
  uint32_t fold_rows;
  void (*fixed_base)(HipBackend& be, const uint32_t* g, const uint32_t* scalars, size_t n, uint32_t* out);
  // ark-serialize bytes of n points (device) -> n resident affine points; returns the number of invalid points
  uint32_t (*srs_decode)(HipBackend& be, const uint8_t* bytes_dev, size_t n, int compressed, uint32_t* out);
  // n resident affine points -> their ark-serialize bytes (device buffers)
  void (*srs_encode)(HipBackend& be, const uint32_t* pts_dev, size_t n, int compressed, uint8_t* out_dev);
  // host-side helpers (a handful of points, as the reference does on the host)
  void (*points_sum)(const uint32_t* pts, size_t count, uint32_t* out);
  void (*point_mul)(const uint32_t* pt, const uint32_t* k_mont, uint32_t* out);
  void (*fr_mul)(const uint32_t* a_mont, const uint32_t* b_mont, uint32_t* out_mont);      // one scalar-field product on the host
  void (*fr_inv)(const uint32_t* a_mont, uint32_t* out_mont);                             // a^-1 (0 -> 0), host
  void (*fr_one)(uint32_t* out_mont);
};

struct FieldOps {
  NttRunner* (*make_ntt)(HipBackend& be, unsigned log_n);
  void (*poly_eval)(HipBackend& be, const uint32_t* x, size_t n, const uint32_t* z_host, uint32_t* out_host, uint32_t fan);
  void (*div_scan)(HipBackend& be, const uint32_t* x, size_t n, const uint32_t* z_host, const uint32_t* carry_in_host, uint32_t* out,
                   uint32_t fan);
  void (*witness)(HipBackend& be, const uint32_t* p, size_t n, const uint32_t* z_host, uint32_t* q, uint32_t fan);
  void (*fr_fold)(HipBackend& be, uint32_t* lo, const uint32_t* hi, size_t n, const uint32_t* s);
  void (*fr_dot)(HipBackend& be, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out_host);
  void (*ipa_fold_dots)(HipBackend& be, uint32_t* c, uint32_t* z, size_t m, const uint32_t* u_host, const uint32_t* u_inv_host, uint32_t* out_host);
  void (*fr_powers)(HipBackend& be, const uint32_t* z, size_t n, uint32_t* out);
  void (*ipa_key_scalars)(HipBackend& be, const uint32_t* c, size_t m, uint32_t* s, size_t n0, const uint32_t* fold_u, size_t fold_m,
                          uint32_t* out_l, uint32_t* out_r);
  void (*fr_lincomb)(HipBackend& be, const void* addr, const void* lens, const void* xi, size_t k, void* out, size_t n_out);
  void (*column_hash)(HipBackend& be, int hash, const uint32_t* ext, uint32_t rows, uint32_t n_cols, uint32_t* out);
  // one slab of rows absorbed into the per-column chaining states (hash.hpp, ColumnHashPartBody); columns [col0, col0 + cols)
  void (*column_hash_part)(HipBackend& be, int hash, const uint32_t* ext, uint32_t rows, uint32_t n_cols, uint32_t rows_total, uint32_t col0,
                           uint32_t cols, int first, int last, uint32_t* state, uint32_t* out);
};

// (accessor functions rather than global tables: a namespace-scope constant would also be emitted into the
// device image, where the host function addresses do not exist)
const CurveOps& curve_ops_bls12_381(); const CurveOps& curve_ops_bn254(); const CurveOps& curve_ops_pallas();
const FieldOps& field_ops_bls12_381(); const FieldOps& field_ops_bn254(); const FieldOps& field_ops_pallas();

inline const CurveOps& curve_ops(pc_curve c) {
  return c == PC_CURVE_BLS12_381 ? curve_ops_bls12_381() : c == PC_CURVE_BN254 ? curve_ops_bn254() : curve_ops_pallas();
}
inline const FieldOps& field_ops(pc_curve c) {
  return c == PC_CURVE_BLS12_381 ? field_ops_bls12_381() : c == PC_CURVE_BN254 ? field_ops_bn254() : field_ops_pallas();
}

// hash-only kernels (hash_tu.hip)
void gather_columns(HipBackend& be, const uint32_t* mat, size_t rows, size_t n_cols, const uint32_t* idx_dev, size_t t, uint32_t* out);
void merkle_level(HipBackend& be, int hash, const uint32_t* child, uint32_t* parent, uint32_t n_leaves, uint32_t bottom,
                  uint32_t len_prefix, size_t cnt);

}  // namespace pc
