// Hash-only kernels: one level of the Merkle tree over the column digests (hash.hpp).
#include "pc_internal.hpp"
#include "hash.hpp"
namespace pc {
void merkle_level(HipBackend& be, int hash, const uint32_t* child, uint32_t* parent, uint32_t n_leaves, uint32_t bottom,
                  uint32_t len_prefix, size_t cnt) {
  if (hash == PC_HASH_SHA256) { MerkleLevelBody<Sha256> b{child, parent, n_leaves, bottom, len_prefix}; be.launch(b, cnt, 64); }
  else { MerkleLevelBody<Blake2s256> b{child, parent, n_leaves, bottom, len_prefix}; be.launch(b, cnt, 64); }
}

// Columns of a row-major matrix of 32-byte elements: out[j * rows + r] = mat[r * n_cols + idx[j]]  (field-agnostic)
struct GatherColumnsBody {
  const uint4* mat; const uint32_t* idx; uint4* out; uint32_t rows, n_cols, t;
  PC_HD void operator()(uint32_t lane) const {
    const uint32_t j = lane / rows, r = lane - j * rows;
    const uint4* src = mat + ((size_t)r * n_cols + idx[j]) * 2;
    uint4* dst = out + ((size_t)j * rows + r) * 2;
    dst[0] = src[0]; dst[1] = src[1];
  }
};
void gather_columns(HipBackend& be, const uint32_t* mat, size_t rows, size_t n_cols, const uint32_t* idx_dev, size_t t, uint32_t* out) {
  GatherColumnsBody b{(const uint4*)mat, idx_dev, (uint4*)out, (uint32_t)rows, (uint32_t)n_cols, (uint32_t)t};
  be.launch(b, rows * t, 256);
}
}  // namespace pc
