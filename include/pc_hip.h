/* pc_hip.h -- C ABI of the MI355X (gfx950) backend for the commit/open hot path of
 * arkworks-rs/poly-commit.
 *
 * Each entry point replaces one call the reference makes into its (external) arithmetic
 * crates; a Rust shim crate binds these with `extern "C"`.  Citations
 * are relative to the reference checkout.
 *
 * Conventions (identical to arkworks' in-memory representations, so buffers cross the FFI
 * without conversion):
 *   - field element: little-endian limbs, Montgomery form, 32 bytes (Fr of all three curves,
 *     Fq of BN254 / Pallas) or 48 bytes (Fq of BLS12-381)       [ark-ff Fp<MontBackend>]
 *   - "bigint" scalar: canonical residue, 32 bytes LE               [F::into_bigint()]
 *   - affine point: x || y (Montgomery).  With stride == 2*sizeof(Fq) the point at infinity
 *     is encoded as (0,0); with a larger stride (Rust `Affine{x,y,infinity:bool}`, 104 / 72
 *     bytes) the byte at offset 2*sizeof(Fq) is the infinity flag.
 * All functions return PC_OK (0) or a negative pc_status; they never abort or throw across
 * the boundary.  A pc_ctx is bound to one GPU; calls on one ctx are serialised by an
 * internal mutex (the reference is called from arbitrary rayon threads, e.g.
 * poly-commit/src/hyrax/mod.rs:233-242).
 */
#ifndef PC_HIP_H
#define PC_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pc_ctx pc_ctx;
typedef struct pc_srs pc_srs;
typedef struct pc_job pc_job;

typedef enum { PC_CURVE_BLS12_381 = 0, PC_CURVE_BN254 = 1, PC_CURVE_PALLAS = 2 } pc_curve;
typedef enum { PC_SCALARS_CANONICAL = 0, PC_SCALARS_MONTGOMERY = 1 } pc_scalar_form;
typedef enum { PC_MEM_HOST = 0, PC_MEM_DEVICE = 1 } pc_mem;

typedef enum { PC_HASH_SHA256 = 0, PC_HASH_BLAKE2S = 1 } pc_hash;

typedef enum {
  PC_OK = 0,
  PC_ERR_INVALID_ARG = -1,
  PC_ERR_OOM = -2,
  PC_ERR_HIP = -3,
  PC_ERR_NO_DEVICE = -4,
  PC_ERR_TOO_LARGE = -5,
  PC_ERR_UNSUPPORTED = -6
} pc_status;

/* Library / device lifetime. */
int pc_hip_device_count(void);
int pc_hip_init(int device_id, pc_ctx** out);
void pc_hip_shutdown(pc_ctx* ctx);
const char* pc_hip_strerror(int status);
/* Last HIP error string recorded on this ctx (for Error::InvalidParameters(String)). */
const char* pc_hip_last_error(const pc_ctx* ctx);

/* SRS residency.  Replaces nothing in the reference -- it is the hook MarlinKZG10::trim
 * (poly-commit/src/marlin/marlin_pc/mod.rs:80-169, powers copied at :96) and
 * InnerProductArgPC::trim (ipa_pc/mod.rs:359-401) call once per committer key so that
 * `powers_of_g` / `comm_key` stay in HBM across commit/open calls.
 * n_max_scalars bounds the MSM length later issued against this SRS (0 = n). */
int pc_hip_srs_upload(pc_ctx* ctx, pc_curve curve, const void* bases, size_t n, size_t stride_bytes,
                      pc_mem where, pc_srs** out);
void pc_hip_srs_free(pc_srs* srs);
/* The same residency straight from ark-serialize bytes: `bytes` starts with a serialized Vec<G1Affine> -- u64 LE length,
 * then the points, compressed or not -- which is how kzg10::UniversalParams begins (its CanonicalSerialize writes
 * powers_of_g first: poly-commit/src/kzg10/data_structures.rs:57-77; deserialisation :80-112) and what an IPA key is
 * (ipa_pc/data_structures.rs:17-36).  At most max_points points (0 = all) are decoded ON THE DEVICE (compressed points:
 * y = (x^3 + b)^((p+1)/4) for BLS12-381 and BN254) into a resident SRS; *out_bytes_consumed = 8 + len * point size, where
 * the next field of the structure starts.  Pallas (p = 1 mod 2^32) takes its square roots by Tonelli-Shanks.  Every decoded point is checked to be on the curve (PC_ERR_INVALID_ARG
 * otherwise); the subgroup check of Validate::Yes is the verifier-side `check()` and is not repeated here.
 * Point encodings: host/transcript.hpp / csrc/serialize.hpp (ark-ec's generic short-Weierstrass flags; the zcash
 * encoding for BLS12-381). */
int pc_hip_srs_load_serialized(pc_ctx* ctx, pc_curve curve, const void* bytes, size_t n_bytes, int compressed,
                               size_t max_points, pc_srs** out, size_t* out_points, size_t* out_bytes_consumed);
/* The inverse: `count` resident points from `offset` on as the ark-serialize image of a Vec<G1Affine> (u64 LE length, then the
 * points, compressed or not; encoded on the device) -- what CanonicalSerialize writes for `powers_of_g` (kzg10/data_structures.rs:
 * 57-63) or an IPA `comm_key`.  *out_written = 8 + count * point size, also when out_bytes_host is NULL (size query) . */
int pc_hip_srs_serialize(pc_ctx* ctx, const pc_srs* srs, size_t offset, size_t count, int compressed, void* out_bytes_host,
                         size_t capacity, size_t* out_written);
/* Byte layout of a serialized kzg10::UniversalParams (host only, nothing is decoded): CanonicalSerialize writes powers_of_g:
 * Vec<G1Affine>, powers_of_gamma_g: BTreeMap<usize, G1Affine>, h, beta_h: G2Affine, neg_powers_of_h: BTreeMap<usize, G2Affine>
 * in this order (kzg10/data_structures.rs:57-77; deserialisation :80-112).  out = {offset of powers_of_g, its length, offset of
 * powers_of_gamma_g, its length, offset of h, offset of beta_h, offset of neg_powers_of_h, its length, total bytes}; offsets
 * point at the u64 length prefixes (map entries: u64 LE key, then the point).  The committer's half goes to the device with
 * pc_hip_srs_load_serialized(bytes + out[0]); the G2 fields stay with the verifier-side Rust.  BLS12-381 and BN254. */
int pc_hip_universal_params_layout(pc_curve curve, const void* bytes, size_t n_bytes, int compressed, size_t out[9]);
/* Optional, once per committer key (same place as the upload, i.e. `trim`): build the window
 * table T[w][i] = 2^(c w) * bases[i] in HBM, (bits/c + 1) x the size of the SRS (BLS12-381: every 96-byte point in its
 * own 128-byte line, so 4/3 of that: 25.8 GB for 2^24 points at c = 22; PC_HIP_TBL_PAD=0 packs them).  MSMs of at least
 * min_pairs pairs (0 = a quarter of the SRS) against this SRS then run with one bucket set shared
 * by all windows: fewer, wider windows, no window fold -- same results, bit for bit.  window_bits
 * 0 = choose from the SRS length.  Shorter MSMs keep the table-free path.  pc_hip_ec_fold drops
 * the table (the key changes).  Nothing in the reference corresponds to it: ark-ec's
 * VariableBaseMSM has no fixed-base state. */
int pc_hip_srs_precompute(pc_ctx* ctx, pc_srs* srs, unsigned window_bits, size_t min_pairs);
/* The same with the form of the table chosen by the caller.  PC_HIP_TABLE_GLV: the table holds only the windows of the ~128-bit halves of
 * the GLV split k = k1 + k2*lambda (BLS12-381, BN254 and Pallas all have the j = 0 endomorphism phi(x, y) = (beta x, y)): HALF the
 * memory and build time (12.9 instead of 25.8 GB for a 2^24-point BLS12-381 key); every scalar is split on the device, the digits of
 * k2 go to a second bucket set with the SAME table points, and phi is applied once to that set's reduced sum -- the same number of
 * bucket additions, one more bucket set to reduce (more below 2^22, where the reduction weighs more).  The split runs once per
 * scalar in a kernel of its own (40-byte records the sort's two passes read).  PC_HIP_TABLE_GLV_IF_TIGHT: the GLV form only when the
 * full table would take more than half of the free device memory.  PC_HIP_TABLE_GLV_IF_LARGE: the GLV form when the full table would
 * exceed 4 GiB (PC_HIP_TABLE_GLV_LARGE_MB overrides), i.e. for keys of 2^22 BLS12-381 points and more -- where the table is what
 * decides how many keys fit the device -- and the full table for small keys.  pc_hip_srs_precompute = IF_TIGHT (the full table is
 * 7 % faster at 2^24: blocking MSM 38.4 vs 41.4 ms) unless the environment says PC_HIP_TABLE_GLV=0 / 1 / large.  Results are
 * bit-identical in every form. */
#define PC_HIP_TABLE_GLV 1
#define PC_HIP_TABLE_GLV_IF_TIGHT 2
#define PC_HIP_TABLE_GLV_IF_LARGE 4
int pc_hip_srs_precompute_ex(pc_ctx* ctx, pc_srs* srs, unsigned window_bits, size_t min_pairs, unsigned flags);
size_t pc_hip_srs_len(const pc_srs* srs);
/* Device pointer of the packed (x||y) resident bases, for callers that build on it. */
void* pc_hip_srs_device_ptr(const pc_srs* srs);

/* Variable-base MSM:  out = sum_{i<n} scalars[i] * bases[base_offset + i].
 * Replaces <E::G1 as VariableBaseMSM>::msm_bigint(&powers_of_g[lz..], &coeffs) at
 * poly-commit/src/kzg10/mod.rs:175-178 and :255-258, and ipa_pc/mod.rs:64.
 * form = PC_SCALARS_MONTGOMERY accepts the polynomial's coefficient slice as it lies in
 * memory and fuses convert_to_bigints (kzg10/mod.rs:463-470) into the digit kernel.
 * out_xy: 2*sizeof(Fq) bytes on the host, affine, Montgomery; *out_is_infinity set if the
 * sum is the identity (out_xy is then all zero).
 * HOST scalars of 2^21 pairs and more (PC_HIP_HOST_SPLIT_LOG2) run as ONE MSM in parts (PC_HIP_HOST_PARTS: a count, or relative
 * weights; default "1,2,5,8"; 0 = two half-size MSMs on two pipelines): the PCIe copy and the sort of a part on an auxiliary queue
 * beside the accumulation of the part before, one bucket reduction and one host tail -- the same point, bit for bit. */
int pc_hip_msm(pc_ctx* ctx, const pc_srs* srs, size_t base_offset, const void* scalars,
               pc_scalar_form form, pc_mem where, size_t n, void* out_xy, int* out_is_infinity);

/* Batched MSM over one SRS (MarlinKZG10::commit's sequential loop over polynomials,
 * marlin_pc/mod.rs:192-237): n_polys independent scalar vectors, out_xy holds n_polys points.
 * Equal-length vectors against a key with a window table run 8 per pass (one sort / accumulate / reduce pipeline with a bucket set
 * each, two such pipelines alternating); HOST vectors are copied pass by pass to the pipeline that will run them, beside the other
 * pipeline's pass (64 x 2^20 BN254 scalars from pageable memory: 100 ms, device-resident: 92 ms). */
int pc_hip_msm_batch(pc_ctx* ctx, const pc_srs* srs, const size_t* base_offsets,
                     const void* const* scalars, const size_t* n, size_t n_polys,
                     pc_scalar_form form, pc_mem where, void* out_xy, int* out_is_infinity);

/* Asynchronous form of pc_hip_msm: queues the MSM on one of the SRS's independent pipelines
 * (own HIP stream + workspace) and returns; pc_hip_job_wait blocks until the result has been
 * written to out_xy / out_is_infinity (which must stay valid until then) and frees the job.
 * Lets a prover keep several commitments in flight so that the latency-bound tail of one MSM
 * overlaps the bucket accumulation of the next (pc_hip_msm_batch does this internally).
 * Threads: every entry point takes the context's lock, but pc_hip_job_wait waits for the device OUTSIDE it, so other threads
 * keep queueing work on the same context while one waits.  An SRS must not be freed while a job on it is being waited for. */
int pc_hip_msm_async(pc_ctx* ctx, const pc_srs* srs, size_t base_offset, const void* scalars,
                     pc_scalar_form form, pc_mem where, size_t n, void* out_xy, int* out_is_infinity,
                     pc_job** out_job);
int pc_hip_job_wait(pc_ctx* ctx, pc_job* job);

/* Residency accounting (bytes of device memory), so that a caller -- the Rust shim's key / polynomial caches, which replace the
 * residency hook MarlinKZG10::trim cannot offer (marlin_pc/mod.rs:80-169 returns the key by value) -- can hold a budget:
 *   pc_hip_srs_bytes_resident  one key: out[0] bases, [1] window table(s), [2] fold table, [3] workspaces of its MSM pipelines
 *   pc_hip_ctx_bytes_resident  out[0] everything this library holds on the context's device (all contexts of the process on that
 *                              device), [1] bases of this context's keys, [2] their window tables, [3] their fold tables,
 *                              [4] the context's staging / scratch buffers, [5] number of key objects alive
 *   pc_hip_ctx_trim            give back what can be rebuilt on demand: staging and scratch buffers, NTT plans, the working keys
 *                              that pc_hip_ec_fold_from caches per committer key, idle pipelines' sort scratch. */
int pc_hip_srs_bytes_resident(const pc_srs* srs, size_t out[4]);
int pc_hip_ctx_bytes_resident(pc_ctx* ctx, size_t out[6]);
int pc_hip_ctx_trim(pc_ctx* ctx);

/* Tuning (optional): window bits c (0 = auto), level-0 chunk length T (0 = auto).  Applies
 * to SRS objects uploaded afterwards. */
int pc_hip_set_msm_tuning(pc_ctx* ctx, unsigned window_bits, unsigned chunk);

/* Kernel-only timing of the last MSM issued on this ctx, in milliseconds, by phase
 * (digits+hist, scan, scatter, accumulate, seg-reduce, bucket-reduce, tail).  For bench.py.
 * After pc_hip_msm_batch over a window table (many-MSM passes): [0..5] summed over the passes, [6] the union of the passes'
 * accumulate intervals (passes overlap on two pipelines), [7] the number of passes.
 * After a blocking call on host memory that ran in parts (pc_hip_msm / pc_hip_kzg_open from 2^21 pairs): [0..2] the sort of the FIRST
 * part, [3] from there to the end of the last part's accumulation (the later parts' sorts and the bucket merges run inside it),
 * [4] the last part's segmented reduction, [5] the one bucket reduction.  (PC_HIP_HOST_PARTS=0, two half-size MSMs: the SUM of the
 * two halves' phases; marks and shape then describe the second half.) */
int pc_hip_last_msm_phases_ms(const pc_ctx* ctx, float out[8]);
/* The same phase boundaries of the last completed MSM as offsets (ms) from the moment pc_hip_set_timing(ctx, 1) was last
 * called: out[0] = the call was queued, out[1..6] = end of digits+hist, scan, scatter, accumulate, seg-reduce,
 * bucket-reduce (-1: not recorded).  MSMs of different pipelines overlap on the device; with absolute marks a caller can
 * take the UNION of the accumulate intervals [out[3], out[4]] of a timed region (bench.py's roofline.kernel_ms). */
int pc_hip_last_msm_marks_ms(const pc_ctx* ctx, float out[8]);
/* Geometry the last completed MSM ran with: {window bits c, signed digits per scalar (= mixed additions per pair),
 * buckets, 1 if the window table was used}.  For bench.py's arithmetic roofline. */
int pc_hip_last_msm_shape(const pc_ctx* ctx, uint32_t out[4]);


/* Device buffers for callers that do not link a HIP runtime themselves (the Rust shim, the C++ host
 * mirror): the vectors of an IPA opening stay in HBM across all rounds (ipa_pc/mod.rs:664-711) and
 * are addressed by the raw device pointers the pc_hip_fr_* / PC_MEM_DEVICE entry points take.
 * Copies are synchronous.  Nothing in the reference corresponds to them. */
int pc_hip_malloc(pc_ctx* ctx, size_t bytes, void** out_dev);
int pc_hip_free(pc_ctx* ctx, void* dev);
int pc_hip_memcpy_h2d(pc_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int pc_hip_memcpy_d2h(pc_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* Many short MSMs over the SAME bases in one pass:
 *   out[k] = sum_{j < m} scalars[k][j] * bases[base_offset + j],   k < n_msms,
 * scalars = n_msms x m elements, contiguous.  This is HyraxPC::commit's "one multi-commitment per
 * row" (poly-commit/src/hyrax/mod.rs:233-242: pedersen_commit(&ck.com_key, row) per matrix row,
 * :86-93, inside a par_iter) -- sqrt(n) MSMs of sqrt(n) pairs, each far below the latency floor of
 * a stand-alone launch sequence (~1 ms).  The m bases get a small window table (built on the first
 * call, cached per (base_offset, m, n_msms)); MSM k owns bucket set k of one shared sort /
 * accumulate / reduce pipeline; results are folded per MSM on the device and normalised with one
 * inversion.  out_xy: n_msms affine points; out_is_infinity: n_msms flags or NULL. */
int pc_hip_msm_many(pc_ctx* ctx, pc_srs* srs, size_t base_offset, const void* scalars, pc_scalar_form form, pc_mem where,
                    size_t m, size_t n_msms, void* out_xy, int* out_is_infinity);

/* Enable/disable hipEvent phase timing on this ctx (off by default). */
int pc_hip_set_timing(pc_ctx* ctx, int on);

/* Batched forward NTT == reed_solomon(row, rho_inv) for every row of the coefficient matrix:
 * GeneralEvaluationDomain::<F>::new(m * rho_inv).fft(msg), poly-commit/src/linear_codes/
 * utils.rs:112-127, called per row from LinearEncode::compute_matrices,
 * linear_codes/mod.rs:131-135.  `field_of` selects the SCALAR field of that curve.
 * in: rows x in_cols elements (row-major, Montgomery), in_cols <= 2^log_n; each row is
 * zero-padded to 2^log_n and transformed; out: rows x 2^log_n, natural order,
 * out[r][j] = sum_i in[r][i] * omega^(i j) with arkworks' omega (pinned by
 * test_reed_solomon, utils.rs:303-331).
 * log_n <= PC_HIP_NTT_MAX_LOG_N (the transform runs as two LDS-staged passes of 2^ceil(log_n/2) and
 * 2^floor(log_n/2) points; Ligero's rows are 2^17 at 2^24 coefficients); larger sizes return
 * PC_ERR_UNSUPPORTED before anything is allocated.  With `in` and `out` both in host memory and more than one
 * slab of rows (32 MB of output each, PC_HIP_LIGERO_SLAB_MB) the call runs like pc_hip_ligero_commit's
 * host-to-host form: a slab goes in and is transformed while the slabs before it travel back
 * (pc_hip_last_ntt_phases_ms reports zeros then: the kernels run under the copies). */
#define PC_HIP_NTT_MAX_LOG_N 22
int pc_hip_ntt_batch(pc_ctx* ctx, pc_curve field_of, const void* in, pc_mem where_in, size_t rows,
                     size_t in_cols, unsigned log_n, void* out, pc_mem where_out);
/* Kernel-only milliseconds of the last pc_hip_ntt_batch: [pass A, pass B]. */
int pc_hip_last_ntt_phases_ms(const pc_ctx* ctx, float out[2]);

/* Column digests of the encoded matrix: step 2 of LinearCodePCS::commit
 * (poly-commit/src/linear_codes/mod.rs:256-263) for the byte-digest column hasher
 * FieldToBytesColHasher<F, D> (bench-templates/src/lib.rs:309-338):
 *   out[j] = D( to_bytes!(column j) ),  to_bytes! = u64 LE length || 32-byte LE canonical residues,
 * D = SHA-256 or BLAKE2s-256.  ext_mat: rows x n_cols (row-major, Montgomery) -- the output of
 * pc_hip_ntt_batch, which therefore never has to leave HBM; out_digests: n_cols x 32 bytes
 * (feed them to pc_hip_merkle_tree without leaving the device). */
int pc_hip_column_hash(pc_ctx* ctx, pc_curve field_of, const void* ext_mat, pc_mem where_in, size_t rows,
                       size_t n_cols, pc_hash hash, void* out_digests, pc_mem where_out);

/* The same digests when the rows of the encoded matrix are spread over several devices (the rows of
 * LinearEncode::compute_matrices are independent, linear_codes/mod.rs:131-135, but the column hash of :256-263 needs a whole
 * column): every device absorbs ITS slab of `rows` consecutive rows (ext_slab_dev: rows x n_cols, device memory) into the
 * digests' chaining states of the columns [col0, col0 + cols) and the states travel from device to device -- 48 bytes per
 * column instead of a transpose of the matrix.  state_dev: n_cols x 48 bytes, indexed by column (12 words: the digest's h,
 * its byte counter, and the 8 message bytes that straddle the slab edge: the u64 length prefix shifts the 32-byte elements by
 * 8 against the 64-byte blocks); read unless `first`, written unless `last`.  first != 0: this slab starts the columns (IV,
 * length prefix of rows_total elements); last != 0: this slab ends them, out_digests_dev (n_cols x 32 bytes, device) receives
 * the digests of the columns of the range -- bit-identical to pc_hip_column_hash of the whole matrix.  Slabs other than the
 * last hold an even number of rows (PC_ERR_UNSUPPORTED otherwise). */
int pc_hip_column_hash_part(pc_ctx* ctx, pc_curve field_of, pc_hash hash, const void* ext_slab_dev, size_t rows,
                            size_t n_cols, size_t rows_total, size_t col0, size_t cols, int first, int last,
                            void* state_dev, void* out_digests_dev);

/* Merkle tree over the column digests: create_merkle_tree, poly-commit/src/linear_codes/
 * mod.rs:506-521 (called at :270-274) -> ark_crypto_primitives MerkleTree::new, for the Config
 * the reference's tests and benches instantiate (linear_codes/univariate_ligero/tests.rs:21-37):
 * identity leaf hash, byte-digest two-to-one hash `hash`, ByteDigestConverter.  The n_leaves
 * 32-byte digests (the output of pc_hip_column_hash, still in HBM) are padded with empty leaves
 * to 2^h >= 2 leaves (mod.rs:517-518); the bottom level hashes conv(left) || conv(right) with
 * conv = raw digest bytes (len_prefix = 0) or the ark-serialize image of the Vec<u8> digest,
 * u64 LE length || bytes (len_prefix = 1, what ByteDigestConverter produces); upper levels hash
 * left || right.  out_nodes: (2^h - 1) x 32 bytes in heap order -- root first, children of node i
 * at 2i+1 and 2i+2 -- the layout of MerkleTree::non_leaf_nodes, from which the caller reads the
 * root (commitment, mod.rs:277) and authentication paths (col_tree.generate_proof, mod.rs:555-557). */
int pc_hip_merkle_tree(pc_ctx* ctx, pc_hash hash, const void* leaf_digests, pc_mem where_in, size_t n_leaves,
                       int len_prefix, void* out_nodes, pc_mem where_out);

/* The queried columns of a resident encoded matrix: generate_proof's `ext_mat.cols()[i]` for the t indices the sponge
 * produced (poly-commit/src/linear_codes/mod.rs:546-552), without moving the matrix:
 *   out[j * rows + r] = mat[r * n_cols + indices[j]],  32-byte elements, any of the scalar fields.
 * mat_dev: device pointer (rows x n_cols, row-major: the ext_out of pc_hip_ligero_commit / pc_hip_ntt_batch). */
int pc_hip_matrix_columns(pc_ctx* ctx, const void* mat_dev, size_t rows, size_t n_cols, const uint32_t* indices_host, size_t t,
                          void* out, pc_mem where_out);

/* LinearCodePCS::commit steps 1-3 for one polynomial in one call, nothing but the results leaving
 * HBM (poly-commit/src/linear_codes/mod.rs:248-277): encode the rows x in_cols coefficient matrix
 * (pc_hip_ntt_batch), digest the 2^log_n columns (pc_hip_column_hash, col_hash), build the Merkle
 * tree (pc_hip_merkle_tree, tree_hash / len_prefix).  nodes_out_host: (2^h - 1) x 32 bytes, h =
 * max(1, log_n), root first -- the commitment is nodes[0] plus the metadata the caller already has
 * (mod.rs:280-287).  leaves_out_host (2^log_n x 32 bytes) and ext_out (rows x 2^log_n Fr, host or
 * device per where_ext) are what LinCodePCCommitmentState keeps for open (mod.rs:264-268); either
 * may be NULL.  With mat AND ext_out on the host -- the trait's shape -- the call runs in slabs of rows (32 MB of encoded matrix
 * each, PC_HIP_LIGERO_SLAB_MB; 0 = off): slab s is copied in, encoded and absorbed into the column digests' chaining states while
 * the slabs before it travel back, copied by three helper threads on queues of their own (pageable copies block their caller while
 * the pages are pinned), so the call costs little more than the encoded matrix's way over PCIe (config 5: 58-60 -> 39-43 ms)
 * and holds four slabs instead of the encoded matrix in HBM; same bits either way. */
int pc_hip_ligero_commit(pc_ctx* ctx, pc_curve field_of, const void* mat, pc_mem where_in, size_t rows, size_t in_cols,
                         unsigned log_n, pc_hash col_hash, pc_hash tree_hash, int len_prefix, void* ext_out,
                         pc_mem where_ext, void* leaves_out_host, void* nodes_out_host);
/* Kernel-only milliseconds of the last pc_hip_ligero_commit (timing on): [NTT pass A, NTT pass B,
 * column digests, Merkle tree]; a call that ran in slabs reports [0, 0, 0, Merkle tree] (its kernels run under the copies). */
int pc_hip_last_ligero_phases_ms(const pc_ctx* ctx, float out[4]);

/* out[i] = sum_j xi[j] * polys[j][i] for i < n_out (coefficients past lens[j] are zero): the
 * random linear combination MarlinKZG10::open forms before the one witness division,
 * poly-commit/src/marlin/marlin_pc/mod.rs:281-287 (and :291-301 for the shifted polynomials).
 * polys: k pointers, all host or all device (where_in); xi_host: k Fr (Montgomery).  The same
 * call is Ligero's row combination v = b^T * mat in open (Matrix::row_mul, poly-commit/src/utils.rs:120-147,
 * as used by generate_proof, linear_codes/mod.rs:539) with the rows as the polynomials. */
int pc_hip_fr_lincomb(pc_ctx* ctx, pc_curve field_of, const void* const* polys, pc_mem where_in, const size_t* lens,
                      size_t k, const void* xi_host, void* out, pc_mem where_out, size_t n_out);

/* Witness polynomial q = p / (x - z): KZG10::compute_witness_polynomial,
 * poly-commit/src/kzg10/mod.rs:217-240.  coeffs: n Fr (Montgomery); z: one Fr (Montgomery,
 * host); out: n-1 Fr.  Keeps the quotient in HBM between commit and open. */
int pc_hip_witness_poly(pc_ctx* ctx, pc_curve field_of, const void* coeffs, pc_mem where_in, size_t n,
                        const void* z_host, void* out, pc_mem where_out);
/* KZG10::open without hiding as ONE call (poly-commit/src/kzg10/mod.rs:287-310 = compute_witness_polynomial :217-240 +
 * the MSM of open_with_witness_polynomial :255-258):  out = sum_j q[j] * bases[base_offset + j],  q = p / (x - z), n - 1 pairs.
 * coeffs: n Fr (Montgomery), host or device; the quotient never leaves the device.  Large HOST polynomials are processed in
 * parts, top part first (its quotient needs nothing from below; every further part takes the carry of the one above), so that the
 * copy + division of a part run under the accumulation of the part before (see pc_hip_msm: the same parts, one MSM).  The
 * reference's degree checks (kzg10/mod.rs:393-407) stay with the caller; n - 1 > srs length - base_offset is PC_ERR_INVALID_ARG. */
int pc_hip_kzg_open(pc_ctx* ctx, const pc_srs* srs, size_t base_offset, const void* coeffs, pc_mem where, size_t n,
                    const void* z_host, void* out_xy, int* out_is_infinity);
/* p(z) for n coefficients (Montgomery; z and the result on the host): Polynomial::evaluate, which
 * KZG10::open applies to the blinding polynomial (poly-commit/src/kzg10/mod.rs:276) and every
 * caller to the opened polynomial; for a polynomial sharded over GPUs it is the value each shard
 * contributes to the division carry of the shards below it (one up-sweep, no output polynomial). */
int pc_hip_poly_eval(pc_ctx* ctx, pc_curve field_of, const void* coeffs, pc_mem where_in, size_t n, const void* z_host,
                     void* out_host);
/* The same recurrence exposed with a carry, for a polynomial sharded over several GPUs:
 *   acc = carry_in (or 0);  for i = n-1 .. 0:  acc = coeffs[i] + z*acc;  out[i] = acc.
 * out has n elements; on the shard that holds coefficient 0, out[0] = p(z) and out[1..n) is
 * the witness polynomial.  carry_in_host may be NULL. */
int pc_hip_poly_div_scan(pc_ctx* ctx, pc_curve field_of, const void* coeffs, pc_mem where_in, size_t n,
                         const void* z_host, const void* carry_in_host, void* out, pc_mem where_out);

/* Host-side sum of `count` affine points (x||y Montgomery, (0,0) = infinity): the handful of
 * point additions the reference also performs on the host (e.g. `commitment +=
 * &random_commitment`, kzg10/mod.rs:206) and the fold of per-GPU partial results. */
int pc_hip_points_sum(pc_curve curve, const void* points_xy, size_t count, void* out_xy);


/* ---- InnerProductArgPC halving rounds (poly-commit/src/ipa_pc/mod.rs:664-711) -------------
 * All vectors are device-resident; `field_of`/the SRS select the curve.  Scalars passed by
 * value are one Fr on the host in Montgomery form. */
/* lo[i] += s * hi[i], i < n_half: coeffs_l += u^-1 * coeffs_r and z_l += u * z_r (:691-697). */
int pc_hip_fr_fold(pc_ctx* ctx, pc_curve field_of, void* lo_dev, const void* hi_dev, size_t n_half,
                   const void* s_host);
/* out = <a, b> over n elements (utils.rs:150-155, used at ipa_pc/mod.rs:672,675). */
int pc_hip_fr_dot(pc_ctx* ctx, pc_curve field_of, const void* a_dev, const void* b_dev, size_t n,
                  void* out_host);
/* A round's vector work in one pass, 64 bytes back: with u_host / u_inv_host != NULL first the folds by the PREVIOUS round's
 * challenge at size 2m -- coeffs[i] += u^-1 * coeffs[m + i], z[i] += u * z[m + i], i < m (:691-697) -- then, on the folded
 * vectors of size m, the two inner products of the round (:672,675):
 *   out[0] = <coeffs[m/2 .. m), z[0 .. m/2)>   (the h' term of l),   out[1] = <coeffs[0 .. m/2), z[m/2 .. m)>   (of r).
 * m = 1 (the last fold) returns zeros.  m must be a power of two; both NULL: inner products only (the first round). */
int pc_hip_ipa_fold_dots(pc_ctx* ctx, pc_curve field_of, void* coeffs_dev, void* z_dev, size_t m, const void* u_host,
                         const void* u_inv_host, void* out_dots_host);
/* out[i] = z^i, i < n (ipa_pc/mod.rs:641-649). */
int pc_hip_fr_powers(pc_ctx* ctx, pc_curve field_of, const void* z_host, size_t n, void* out_dev);
/* In-place key fold on the resident comm_key: key[i] = affine(key[i] + u * key[n_half + i]) for
 * i < n_half -- `k_l += k_r.mul(round_challenge)` followed by normalize_batch (:699-707).  The
 * following round's MSMs address the halves with base_offset 0 and n_half/2. */
int pc_hip_ec_fold(pc_ctx* ctx, pc_srs* srs, size_t n_half, const void* u_host);
/* The FIRST key fold of an opening, out of place: *out = a new resident key of n_half points,
 *   out[i] = affine(src[i] + u * src[n_half + i]),  i < n_half,
 * leaving src (the committer key, resident across openings) untouched -- the round's two MSMs run on src itself (with its
 * window table, if built) and no working copy of the key is needed.  If pc_hip_srs_precompute_fold was called on src and
 * n_half is half its length, the multiplication by u costs ~86 mixed additions per element out of the fold table instead of
 * a 130-doubling ladder (a ONE-level table; a key with a two-level table folds with pc_hip_ec_fold2_from). */
int pc_hip_ec_fold_from(pc_ctx* ctx, const pc_srs* src, size_t n_half, const void* u_host, pc_srs** out);
/* Once per committer key (like pc_hip_srs_precompute, at `trim`): the fold table T[b][j] = 2^b * key[n/2 + j], b < 131, of the
 * upper half of the key (131 x n/2 affine points: 17 GB for a 2^22-point Pallas key), used by pc_hip_ec_fold_from: every
 * opening's first fold multiplies THIS half by its round challenge.  Nothing in the reference corresponds to it.  n even. */
int pc_hip_srs_precompute_fold(pc_ctx* ctx, pc_srs* srs);   /* PC_ERR_UNSUPPORTED when the table would exceed half of the free device memory (PC_HIP_FOLD_TABLE_MAX_FRAC) */
/* The same table in its general form (pc_hip_srs_precompute_fold = the library's choice, or PC_HIP_FOLD_TABLE="levels,width"):
 *   levels     1: the upper half of the key (the first fold of an opening); 2: the upper three quarters -- the key after the first
 *              TWO folds (ipa_pc/mod.rs:699-707, rounds 1 and 2) then comes straight from the committer key,
 *                K''[i] = K[i] + u2 K[q + i] + u1 K[2q + i] + (u1 u2) K[3q + i],  q = n / 4          (pc_hip_ec_fold2_from),
 *              as table additions only, and round 2's commitments are MSMs on the committer key itself (by linearity, see
 *              pc_hip_ec_fold2_from); n divisible by 4.  0: two levels from 2^16 points on, if the memory share allows
 *   naf_width  2 .. 5: the table holds the odd multiples d < 2^(w-1) of every doubling, for width-w NAF digits of the challenges'
 *              GLV halves: 2 x 130 / (w + 1) additions per term and element (86 / 65 / 52 / 43) for 131 x 2^(w-2) rows.  0: the
 *              widest form (up to 4) that fits PC_HIP_FOLD_TABLE_MAX_FRAC of the free device memory
 * A 2^22-point Pallas key: 17.6 GB (1, 2), 26 / 53 / 106 GB (2, 2 / 3 / 4).  Built once per committer key, like the window table. */
int pc_hip_srs_precompute_fold_ex(pc_ctx* ctx, pc_srs* srs, unsigned levels, unsigned naf_width);
/* What pc_hip_srs_precompute_fold[_ex] built on this key: levels (0: no table) and the NAF width of its digits. */
int pc_hip_srs_fold_table_info(const pc_srs* srs, unsigned* out_levels, unsigned* out_naf_width);
/* The first TWO key folds of an opening in one step, out of place: *out = a new resident key of n_quarter points,
 *   out[i] = affine(src[i] + u2 * src[q + i] + u1 * src[2q + i] + u1 u2 * src[3q + i]),  q = n_quarter,
 * i.e. the key after `k_l += k_r * u1` and `k_l += k_r * u2` (ipa_pc/mod.rs:699-707 in rounds 1 and 2); src (the committer key) is
 * untouched.  The caller runs round 2's commitments on src by linearity of the MSM,
 *   MSM(K'[a .. a + q), s) = MSM(K[a .. a + q), s) + u1 * MSM(K[a + 2q .. a + 3q), s),   K' = the key after the first fold,
 * so the once-folded key never exists.  With a two-level fold table on src: 3 x ~52 table additions per element (width 4); without
 * one the two folds run one after the other (table / ladder): same result, no gain. */
int pc_hip_ec_fold2_from(pc_ctx* ctx, const pc_srs* src, size_t n_quarter, const void* u1_host, const void* u2_host, pc_srs** out);
/* Round 2's two commitments of such an opening, computed on the committer key `srs` itself (ipa_pc/mod.rs:671-675 for the key after the
 * first fold, K' = K_l + u1 K_r, which is never formed): with q = n_quarter and c = the coefficient vector after the first fold (2q
 * Montgomery scalars on the device),
 *   out_l = MSM(K'[0 .. q), c[q .. 2q)) = MSM(K[0 .. 3q), (c_r | 0 | u1 c_r)),   out_r = MSM(K'[q .. 2q), c[0 .. q)) = MSM(K[q .. 4q), (c_l | 0 | u1 c_l))
 * -- two MSMs of 3q pairs (a third of them zero scalars: no bucket entries) against the key's window table, on two pipelines.  The caller
 * adds h' * <c_r, z_l> and h' * <c_l, z_r> as in every round.  Blocking; affine results (x || y, Montgomery), infinity flags optional. */
int pc_hip_ipa_round2_msms(pc_ctx* ctx, const pc_srs* srs, const void* coeffs_dev, size_t n_quarter, const void* u1_host,
                           void* out_l_xy, int* out_l_is_infinity, void* out_r_xy, int* out_r_is_infinity);
/* The whole halving loop of InnerProductArgPC::open (ipa_pc/mod.rs:664-711) as ONE call, for a resident committer key:
 *   per round k (size m -> m / 2):  l_k, r_k = the two commitments + h' * inner products (:666-677);  u_k = next_challenge(l_k, r_k)
 *   (:681-689 -- the transcript is the caller's: Blake2s over ark-serialize bytes in the reference, host/transcript.hpp here);
 *   coeffs_l += u_k^-1 coeffs_r, z_l += u_k z_r (:691-697);  key_l += u_k key_r (:699-707)
 * with everything the round-by-round entry points above offer, chosen by the library: the committer key's fold table (two levels: rounds
 * 1 and 2 in one step, round 2 on the committer key by linearity), GLV ladder folds, the fixed key with per-base factors from
 * `fixed_key_below` points on (0: the library's default, 2^17; 1: never), captured launch graphs for its small MSMs.  The committer key
 * is not modified (the working keys are cached with it); coeffs_dev (n Montgomery scalars, n a power of two <= the key's length) is
 * consumed.  Outputs: l_vec / r_vec = log2(n) affine points each (x || y, Montgomery; all zero = infinity), final_comm_key, c -- the
 * Proof of ipa_pc/data_structures.rs:175-195 without hiding.  out_round_ms / out_fold_ms: optional, log2(n) floats each (wall time of
 * every round, and of its key fold).  What it buys: one call instead of ~150 for a binding (the Rust shim, a C++ prover), and the
 * fixed key as a key object of its own (cached with the committer key, its points copied and its window table refilled on the device
 * by every opening, its captured launch graphs kept) -- the late rounds' MSMs then run against a table: 0.57 ms on 2^17 points instead of 0.85 ms on 2^16 per
 * round.  Measured at 2^22 over Pallas: 55.5-56.6 ms against 61.9-63.3 ms for the same sequence driven through the single entry points
 * (PC_HIP_IPA_FIXED_TABLE=0: 62.6 ms here as well -- the host between the rounds is not what the difference is). */
typedef void (*pc_ipa_challenge_fn)(void* user, const void* l_xy, const void* r_xy, void* out_u_mont);
int pc_hip_ipa_open_rounds(pc_ctx* ctx, const pc_srs* comm_key, void* coeffs_dev, size_t n, const void* point_host, const void* h_prime_xy_host,
                           pc_ipa_challenge_fn next_challenge, void* user, size_t fixed_key_below,
                           void* out_l_vec_xy, void* out_r_vec_xy, void* out_final_key_xy, void* out_c_host, float* out_round_ms, float* out_fold_ms);
/* Late halving rounds without folding the key (same l_vec / r_vec / final_comm_key, bit for bit): once n has
 * shrunk to n0 the resident key K0 = key[0..n0) stays as it is and the per-base factors s_j that the remaining
 * folds `k_l += k_r * u` (ipa_pc/mod.rs:699-701) would have applied are kept as a device vector s (n0 Fr,
 * Montgomery, initialised to ones, e.g. by pc_hip_fr_powers with z = 1).  Then for the round at size m <= n0
 *   l = MSM(K0, out_l) + h' <c_r, z_l>,  r = MSM(K0, out_r) + h' <c_l, z_r>        (ipa_pc/mod.rs:671-675)
 * with out_l[j] = (j mod m <  m/2) ? coeffs[m/2 + j mod m] * s[j] : 0,
 *      out_r[j] = (j mod m >= m/2) ? coeffs[j mod m - m/2] * s[j] : 0,
 * and the fold by u at size fold_m is s[j] *= u where (j mod fold_m) >= fold_m/2; final_comm_key = MSM(K0, s).
 * A fold of n/2 full-width scalar multiplications costs the latency of one 255-bit ladder (~2.4 ms) however small
 * n gets; two more MSMs over n0 bases do not.  One call does (in this order) the optional fold of s
 * (fold_u_host != NULL) and the optional scalar vectors of the round at size m (out_l_dev/out_r_dev != NULL,
 * coeffs_dev = the m current coefficients). */
int pc_hip_ipa_key_scalars(pc_ctx* ctx, pc_curve field_of, const void* coeffs_dev, size_t m, void* s_dev, size_t n0,
                           const void* fold_u_host, size_t fold_m, void* out_l_dev, void* out_r_dev);
/* Host-side scalar multiplication of one affine point by one Fr (Montgomery):
 * `h_prime.mul(inner_product(..))`, ipa_pc/mod.rs:672,675 -- one point, stays on the host as in
 * the reference. */
int pc_hip_point_mul(pc_curve curve, const void* point_xy, const void* scalar_mont, void* out_xy);
/* out[i] = scalars[i] * g for one fixed base (scalars: n Fr, Montgomery, device; out: n affine
 * points, device): `g.batch_mul(&powers_of_beta)` of KZG10::setup (kzg10/mod.rs:76,83).  Builds a
 * true SRS for end-to-end tests; not on the commit/open path. */
int pc_hip_fixed_base_batch_mul(pc_ctx* ctx, pc_curve curve, const void* g_xy_host, const void* scalars_dev,
                                size_t n, void* out_points_dev);
/* Copy `count` resident affine points starting at `offset` back to the host (final_comm_key). */
int pc_hip_srs_read(pc_ctx* ctx, const pc_srs* srs, size_t offset, size_t count, void* out_xy);

/* ---- One committer key over several GPUs of a node, driven from one process ----------------
 * The reference has no multi-device path; this is the form a prover that holds ONE CommitterKey needs.  The key is
 * cut into N contiguous chunks, one per device (chunk d also keeps the one power below it, so that commit and open
 * address the same resident chunk).  Every call runs the complete single-device path on each chunk in parallel
 * (one host thread per device) -- each device reduces its buckets to ONE point -- and adds the N partial points on
 * the host (what pc_hip_points_sum does; N * 96 bytes, no device-to-device collective: raw bucket arrays never
 * move).  Results are bit-identical to the single-device calls.  A device id may be listed more than once (tests on
 * a one-GPU machine).  The one-process-per-GPU form of the same protocol over RCCL is poly_commit_amd/sharded.py. */
typedef struct pc_group pc_group;
typedef struct pc_group_srs pc_group_srs;
int pc_hip_group_create(const int* device_ids, int n_devices, pc_group** out);
void pc_hip_group_destroy(pc_group* g);
int pc_hip_group_size(const pc_group* g);
pc_ctx* pc_hip_group_ctx(pc_group* g, int i);
/* `trim` for the sharded key (marlin_pc/mod.rs:80-169): upload chunk d to device d; precompute != 0 also builds
 * each chunk's window table (pc_hip_srs_precompute). */
int pc_hip_group_srs_upload(pc_group* g, pc_curve curve, const void* bases_host, size_t n, size_t stride_bytes,
                            int precompute, pc_group_srs** out);
void pc_hip_group_srs_free(pc_group_srs* srs);
size_t pc_hip_group_srs_len(const pc_group_srs* srs);
/* msm_bigint over the sharded key (kzg10/mod.rs:175-178; KZG10::commit): scalars on the host. */
int pc_hip_group_msm(pc_group* g, const pc_group_srs* srs, size_t base_offset, const void* scalars_host,
                     pc_scalar_form form, size_t n, void* out_xy, int* out_is_infinity);
/* MarlinKZG10::commit's loop over k polynomials (marlin_pc/mod.rs:192-237; BASELINE configs[2]: 64 polynomials, SRS
 * sharded): out_xy holds k points. */
int pc_hip_group_msm_batch(pc_group* g, const pc_group_srs* srs, const void* const* scalars_host, const size_t* n,
                           size_t k, pc_scalar_form form, void* out_xy, int* out_is_infinity);
/* KZG10::open, hiding off (kzg10/mod.rs:287-310): witness polynomial + its MSM over the sharded key.  coeffs: n Fr
 * (Montgomery, host), z: one Fr.  Per device one evaluation of its shard, the division carries composed on the
 * host (N field elements), one division scan, one MSM.  out_value_host (optional): p(z). */
int pc_hip_group_kzg_open(pc_group* g, const pc_group_srs* srs, const void* coeffs_host, size_t n, const void* z_host,
                          void* out_proof_xy, int* out_is_infinity, void* out_value_host);
/* KZG commit + open of ONE polynomial over the sharded key as one ASYNCHRONOUS job: commit = MSM(powers, coeffs)
 * (kzg10/mod.rs:175-178), open = witness polynomial + its MSM (:287-310), hiding off.  where = PC_MEM_HOST: `coeffs` is the
 * polynomial's n coefficients in host memory (Montgomery); every device copies its shard once, for both MSMs, into a
 * persistent buffer.  where = PC_MEM_DEVICE: `coeffs` is an array of N device pointers, shard d resident on device d.
 * Returns once the job is queued on the devices' worker threads (one persistent thread per device); up to two jobs stay in
 * flight, so the copy and sort of one polynomial overlap the bucket accumulation of the previous one.  z_host is copied;
 * the out_* buffers (commitment, proof: 2 Fq each; value p(z): one Fr, optional) must stay valid until
 * pc_hip_group_job_wait, which blocks, folds the N partial points and frees the job. */
typedef struct pc_group_job pc_group_job;
int pc_hip_group_commit_open_async(pc_group* g, const pc_group_srs* srs, const void* coeffs, pc_mem where, size_t n,
                                   const void* z_host, void* out_commit_xy, void* out_proof_xy, void* out_value_host,
                                   pc_group_job** out_job);
int pc_hip_group_job_wait(pc_group* g, pc_group_job* job);
/* pc_hip_ligero_commit with the rows of the coefficient matrix split over the devices of the group (even slabs of consecutive
 * rows): every device encodes its rows, the column digests are chained through the devices (pc_hip_column_hash_part: 48 bytes
 * per column travel, the matrix does not), the last device with rows builds the tree.  mat_host: rows x in_cols Fr (Montgomery).
 * nodes_out_host / leaves_out_host as in pc_hip_ligero_commit.  out_ext_slabs: NULL, or one pointer per device that receives
 * the device's resident slab of the encoded matrix (rows [d * per, (d + 1) * per), per = ceil(rows / N) rounded up to even; NULL
 * for a device without rows) -- what LinCodePCCommitmentState keeps for `open`; the caller frees them with pc_hip_free. */
int pc_hip_group_ligero_commit(pc_group* g, pc_curve field_of, const void* mat_host, size_t rows, size_t in_cols, unsigned log_n,
                               pc_hash col_hash, pc_hash tree_hash, int len_prefix, void** out_ext_slabs, void* leaves_out_host,
                               void* nodes_out_host);
/* pc_hip_ntt_batch with the rows split over the devices (rows are independent: linear_codes/mod.rs:131-135). */
int pc_hip_group_ntt_batch(pc_group* g, pc_curve field_of, const void* in_host, size_t rows, size_t in_cols,
                           unsigned log_n, void* out_host);

#ifdef __cplusplus
}
#endif
#endif /* PC_HIP_H */
