// Hash-only kernels: one level of the Merkle tree over the column digests (hash.hpp).
#include "pc_internal.hpp"
#include "hash.hpp"
namespace pc {
void merkle_level(HipBackend& be, int hash, const uint32_t* child, uint32_t* parent, uint32_t n_leaves, uint32_t bottom,
                  uint32_t len_prefix, size_t cnt) {
  if (hash == PC_HASH_SHA256) { MerkleLevelBody<Sha256> b{child, parent, n_leaves, bottom, len_prefix}; be.launch(b, cnt, 64); }
  else { MerkleLevelBody<Blake2s256> b{child, parent, n_leaves, bottom, len_prefix}; be.launch(b, cnt, 64); }
}
}  // namespace pc
