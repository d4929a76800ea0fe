/* include/stark_mlwe.h — C-ABI of libstark_mlwe_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the proving hot path of saholmes/stark-mlwe.  The reference has no FFI
 * seam of its own (100 % safe Rust); these entry points sit BENEATH the Rust signatures listed in
 * SURVEY.md §8(b), which stay unchanged.  Each declaration cites the reference item it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the Rust `extern "C"` block and
 * the wrappers a maintainer adds.
 *
 * Conventions
 *   - A field element is 4 little-endian uint64_t limbs in Montgomery form (R = 2^256): exactly the
 *     in-memory layout of ark-ff `Fp<MontBackend<_,4>,4>`, so `&[F]` <-> `const uint64_t*` is zero-copy.
 *   - Plain-named functions take HOST pointers (what a Rust slice hands over) and return when the
 *     result is in the caller's buffer.  `*_dev` variants take DEVICE pointers obtained from
 *     stark_malloc (or any hipMalloc'd / torch CUDA memory) and are stream-ordered on the context's
 *     stream; call stark_ctx_sync before reading results on the host.
 *   - Every function returns a status: 0 = OK, negative = error class.  stark_last_error() gives text.
 *     The Rust wrappers turn non-zero into panic!, matching the reference's assert!/panic! behaviour
 *     (fri.rs:86-87, merkle/src/lib.rs:148,161).
 *   - One context may be used by one host thread at a time; distinct contexts are independent (every
 *     entry point makes its context's device current, so one process may hold contexts on several GPUs).
 *   - Stream rule: all work of a context is enqueued on ONE stream, chosen at stark_ctx_create.  A caller
 *     that produces inputs or consumes outputs with its own kernels (torch, hipMemcpyAsync, ...) must do
 *     so on that same stream, or on a stream that is ordered against it.  Passing NULL selects the
 *     device's legacy default stream, which HIP orders against every blocking stream — including the
 *     default stream torch uses — so the NULL context is safe next to default-stream callers without
 *     manual synchronisation.  STARK_STREAM_PRIVATE asks for a private non-blocking stream instead:
 *     fastest in isolation, but then the CALLER brackets its own device work with stark_ctx_sync.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     STARK_ERR_HIP.
 */
#ifndef STARK_MLWE_H
#define STARK_MLWE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STARK_OK               0
#define STARK_ERR_INVALID_ARG (-1)
#define STARK_ERR_HIP         (-2)
#define STARK_ERR_RCCL        (-3)
#define STARK_ERR_OOM         (-4)
#define STARK_ERR_UNSUPPORTED (-5)

#define STARK_FIELD_PALLAS_FR    0   /* crates/field/src/lib.rs:13  (the prover field)        */
#define STARK_FIELD_BLS12_381_FR 1   /* crates/fft/src/lib.rs:1     (the `fft` crate's field) */

typedef struct stark_ctx stark_ctx_t;
typedef struct stark_params stark_params_t;
typedef struct stark_tree stark_tree_t;
typedef struct stark_fri_state stark_fri_state_t;
typedef struct stark_proof stark_proof_t;
typedef struct stark_fri_plan stark_fri_plan_t;
typedef struct stark_transcript stark_transcript_t;

/* ---- context / memory ------------------------------------------------------------------------- */
int32_t stark_version(void);
/* device: HIP device ordinal.  stream: the hipStream_t to run on (e.g. torch's current stream);
 * NULL = the legacy default stream; STARK_STREAM_PRIVATE = a private non-blocking stream (see "Stream rule").
 * (SURVEY.md §8(b) sketched `stark_ctx_create(devices, ndev)`: this build runs one process per GPU, so a
 * context is one device; the communicator that spans the GPUs is stark_comm_* below.) */
#define STARK_STREAM_PRIVATE ((void*)(intptr_t)-1)
int32_t stark_ctx_create(int32_t device, void* stream, stark_ctx_t** out);
/* Lifetime: every handle made from a context (stark_tree_t, stark_fri_state_t, stark_fri_plan_t, stark_transcript_t, a stark_params_t returned
 * to the caller) keeps the context alive.  stark_ctx_destroy with such handles outstanding synchronises, marks the context and returns
 * STARK_OK; the handles stay fully usable and the last one freed releases the context's resources.  Device pointers obtained FROM a handle
 * (stark_merkle_level_dev, layers of a FRI state) are valid for work ordered on the context's stream until the handle is freed: freeing
 * returns the block to the context's pool without a device synchronisation, so a caller that read it on ANOTHER stream synchronises first. */
int32_t stark_ctx_destroy(stark_ctx_t* ctx);
int32_t stark_ctx_sync(stark_ctx_t* ctx);
/* The library keeps released temporaries / layers / tree levels in a per-context cache (no hipMalloc /
 * hipFree on the hot path).  trim hands the cached blocks back to the driver. */
int32_t stark_ctx_trim(stark_ctx_t* ctx);     /* also drops the NTT plans (direct twiddle tables) and the NTT scratch vector */
/* Tuning / diagnostic options — explicit state of the context, never read from the environment:
 *   "ntt_direct_max_log" (default 24: one-product twiddle tables up to 2^24 points; 0 = always the two-level lookup),
 *   "ntt_merged_coset" (default 1: a coset transform's pre-scale is folded into its first pass's twiddle table; 0 = separate tables),
 *   "ntt_log_tile" (8..12, default 11; -1 restores the default), "ntt_min_waves" (2 | 4), "poseidon_lane_only" (0 | 1).
 *   "sponge_one_wave" (0 | 1: long serial sponges, small Merkle levels / leaf layers and short transcript hashes on the one-wave / wave-pair kernels
 *   instead of the five-wave latency kernel; comparison), "sponge_debug" (timing experiments on the five-wave kernel; digests are WRONG when set).
 * Changing an option synchronises the stream and drops the cached NTT plans. */
int32_t stark_ctx_set_option(stark_ctx_t* ctx, const char* key, int64_t value);
size_t  stark_ctx_cached_bytes(stark_ctx_t* ctx);
const char* stark_last_error(stark_ctx_t* ctx);
int32_t stark_malloc(stark_ctx_t* ctx, size_t bytes, void** dptr);
int32_t stark_free(stark_ctx_t* ctx, void* dptr);
int32_t stark_memcpy_h2d(stark_ctx_t* ctx, void* dst_dev, const void* src_host, size_t bytes);
int32_t stark_memcpy_d2h(stark_ctx_t* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Diagnostic: lane-level v_mad_u64_u32 (32x32+64 multiply-accumulate) issue rate of this device, measured live with every
 * SIMD saturated — the roofline `peak` of the integer-VALU-bound Poseidon kernels (bench.py "poseidon.roofline"). */
int32_t stark_diag_mac_rate(stark_ctx_t* ctx, double* lane_macs_per_s);
/* HIP-event timing on the context's stream (bench.py measures kernels with these). */
int32_t stark_timer_start(stark_ctx_t* ctx);
int32_t stark_timer_stop_ms(stark_ctx_t* ctx, float* ms);

/* ---- Poseidon constants -------------------------------------------------------------------------
 * PoseidonParams / PoseidonParamsDynamic (poseidon/src/lib.rs:16-21, 104-114).  Constants are passed
 * in as the Rust side derived them (row-major mds[i][j], rc_full[r][i], rc_partial[r]); the library
 * turns them into kernel form (LU factors, sparse partial-round matrices).  t in {9,17,33,65,129}. */
int32_t stark_poseidon_params_upload(stark_ctx_t* ctx, int32_t t, int32_t rf, int32_t rp, const uint64_t* mds,
                                     const uint64_t* rc_full, const uint64_t* rc_partial, stark_params_t** out);
/* Same derivations done inside the library (BLAKE3, utils/src/lib.rs:16-22):
 *   for_width:  poseidon_params_for_width(t)                        poseidon/src/lib.rs:120-146
 *   t17_seed :  params::generate_params_t17_x5(seed)                poseidon/src/lib.rs:318-356
 *               (seed "POSEIDON-T17-X5-TRANSCRIPT" = transcript::default_params, transcript/src/lib.rs:44-46) */
int32_t stark_poseidon_params_for_width(stark_ctx_t* ctx, int32_t t, stark_params_t** out);
int32_t stark_poseidon_params_t17_seed(stark_ctx_t* ctx, const uint8_t* seed, size_t seed_len, stark_params_t** out);
int32_t stark_poseidon_params_export(stark_params_t* p, int32_t* t, int32_t* rf, int32_t* rp, uint64_t* mds, uint64_t* rc_full, uint64_t* rc_partial);
int32_t stark_poseidon_params_free(stark_params_t* p);

/* ---- Poseidon ------------------------------------------------------------------------------------ */
/* permute / permute_dynamic (poseidon/src/lib.rs:31, 219): nstates states of t elements, in place. */
int32_t stark_poseidon_permute_batch(stark_ctx_t* ctx, stark_params_t* p, uint64_t* states, size_t nstates);
int32_t stark_poseidon_permute_batch_dev(stark_ctx_t* ctx, stark_params_t* p, uint64_t* states, size_t nstates);
/* hash_with_ds_dynamic (poseidon/src/lib.rs:288-312) over a batch: hash k absorbs
 * ds_fields[k*nds .. +nds] then inputs[k*cnt .. +cnt], pad 1||0*, squeezes state[0]. */
int32_t stark_poseidon_hash_with_ds_dynamic(stark_ctx_t* ctx, stark_params_t* p, const uint64_t* ds_fields, size_t nds,
                                            const uint64_t* inputs, size_t cnt, size_t n, uint64_t* out);
/* hash_with_ds (legacy, poseidon/src/lib.rs:85-100): t = 17, ds_tag in the capacity lane, no padding. */
int32_t stark_poseidon_hash_with_ds(stark_ctx_t* ctx, stark_params_t* p, const uint64_t* inputs, size_t cnt, const uint64_t* ds_tag, uint64_t* out);
/* One Merkle level: out[k] = hash_with_ds_dynamic([arity, level, pos0+k, tree_label], in[k*arity ..])
 * (merkle/src/lib.rs:167-176); the last chunk may be short. */
int32_t stark_poseidon_hash_ds_batch(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint32_t level, uint64_t pos0, uint64_t tree_label,
                                     const uint64_t* in, size_t n_in, uint64_t* out);
int32_t stark_poseidon_hash_ds_batch_dev(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint32_t level, uint64_t pos0, uint64_t tree_label,
                                         const uint64_t* in, size_t n_in, uint64_t* out);
/* hash_leaf_pair over a layer (fri.rs:38-44 as used at fri.rs:283): h[i] = hash_leaf_pair(f[i], s_i),
 * s_i = f_next[i / m], or zero when f_next == NULL (fri.rs:266).  tparams = transcript params (t=17). */
int32_t stark_leaf_pair_hash(stark_ctx_t* ctx, stark_params_t* tparams, const uint64_t* f, const uint64_t* f_next, size_t n, size_t m, uint64_t* h);
int32_t stark_leaf_pair_hash_dev(stark_ctx_t* ctx, stark_params_t* tparams, const uint64_t* f, const uint64_t* f_next, size_t n, size_t m, uint64_t* h);
/* tr_hash_fields_tagged (fri.rs:28-35): n_hashes independent transcript hashes of k fields each. */
int32_t stark_tr_hash_fields_tagged(stark_ctx_t* ctx, stark_params_t* tparams, const char* tag, const uint64_t* fields, size_t k, size_t n_hashes, uint64_t* out);
int32_t stark_tr_hash_fields_tagged_dev(stark_ctx_t* ctx, stark_params_t* tparams, const char* tag, const uint64_t* fields, size_t k, size_t n_hashes, uint64_t* out);

/* ---- Merkle ---------------------------------------------------------------------------------------
 * MerkleTree::new / new_pairs (merkle/src/lib.rs:147-193, 392-445): level-by-level build, all levels
 * kept resident on the device (openings read them, :261-291).  pairs != 0 => leaves are (f, cp) pairs
 * hashed with the leaf DS level 2^32-1 (:380-388).  `first_pos` / `level0` let a shard build its part
 * of a larger tree (DS positions are global): pass 0 / 0 for a whole tree. */
int32_t stark_merkle_build(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint64_t tree_label, const uint64_t* leaves, size_t n,
                           int32_t pairs, const uint64_t* cp, stark_tree_t** out);
int32_t stark_merkle_build_dev(stark_ctx_t* ctx, stark_params_t* p, size_t arity, uint64_t tree_label, const uint64_t* leaves, size_t n,
                               int32_t pairs, const uint64_t* cp, uint64_t first_pos, uint32_t level0, int32_t stop_at_len, stark_tree_t** out);
int32_t stark_merkle_num_levels(stark_tree_t* t);
size_t  stark_merkle_level_len(stark_tree_t* t, int32_t lvl);
int32_t stark_merkle_root(stark_tree_t* t, uint64_t* out4);
int32_t stark_merkle_level(stark_tree_t* t, int32_t lvl, uint64_t* out);                         /* host copy of a level */
const uint64_t* stark_merkle_level_dev(stark_tree_t* t, int32_t lvl);                           /* device pointer     */
int32_t stark_merkle_gather(stark_tree_t* t, int32_t lvl, const size_t* idx, size_t k, uint64_t* out);
/* open_union_of_paths (merkle/src/lib.rs:246-315) → canonical MerkleProof encoding (DESIGN.md
 * "Proof encoding"): call with buf == NULL to get the length. */
int32_t stark_merkle_open(stark_tree_t* t, const size_t* idx, size_t k, uint8_t* buf, size_t cap, size_t* len);
int32_t stark_merkle_free(stark_tree_t* t);

/* ---- FRI ------------------------------------------------------------------------------------------ */
/* fri_sample_z_ell (fri.rs:59-82). */
int32_t stark_fri_sample_z(stark_ctx_t* ctx, stark_params_t* tparams, uint64_t seed_z, size_t level, size_t domain_size, uint64_t* z4);
/* fri_fold_layer (fri.rs:85-102): out[b] = sum_{t<m} f[b*m+t] z^t, n % m == 0, m >= 2. */
int32_t stark_fri_fold(stark_ctx_t* ctx, const uint64_t* f, size_t n, const uint64_t* z4, size_t m, uint64_t* out);
int32_t stark_fri_fold_dev(stark_ctx_t* ctx, const uint64_t* f, size_t n, const uint64_t* z4, size_t m, uint64_t* out);
/* fri_build_transcript (fri.rs:231-312): all folds, per-layer leaf hashes and the L+1 trees.
 * z_l are derived inside (they depend only on (seed_z, l, size), fri.rs:250). */
int32_t stark_fri_build(stark_ctx_t* ctx, const uint64_t* f0, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state_t** out);
int32_t stark_fri_build_dev(stark_ctx_t* ctx, const uint64_t* f0, size_t n0, const size_t* schedule, size_t L, uint64_t seed_z, stark_fri_state_t** out);
int32_t stark_fri_num_layers(stark_fri_state_t* s);                 /* L + 1 */
size_t  stark_fri_layer_len(stark_fri_state_t* s, int32_t layer);
int32_t stark_fri_layer_f(stark_fri_state_t* s, int32_t layer, uint64_t* out);
int32_t stark_fri_layer_root(stark_fri_state_t* s, int32_t layer, uint64_t* out4);
int32_t stark_fri_layer_z(stark_fri_state_t* s, int32_t layer, uint64_t* out4);
stark_tree_t* stark_fri_layer_tree(stark_fri_state_t* s, int32_t layer);
int32_t stark_fri_state_free(stark_fri_state_t* s);

/* ---- DEEP-ALI (next row N1) ---------------------------------------------------------------------- */
/* deep_ali_merge_evals(_blinded) (deep_ali/src/lib.rs:48-105).  r_opt / beta may be NULL.  c_star may be NULL. */
int32_t stark_ali_merge(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt,
                        const uint64_t* beta4, const uint64_t* omega4, const uint64_t* z4, size_t n, uint64_t* f0, uint64_t* c_star4);
int32_t stark_ali_merge_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt,
                            const uint64_t* beta4, const uint64_t* omega4, const uint64_t* z4, size_t n, uint64_t* f0, uint64_t* c_star4);
/* DeepAliRealBuilder::build_f0 (fri.rs:535-569): 4 column sponges, (z, beta) sampling, merge.
 * aux7 (optional, host): col digests A,S,E,T, seed_f, z, beta. */
int32_t stark_build_f0(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, uint64_t* f0, uint64_t* aux7);
int32_t stark_build_f0_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, size_t n0, uint64_t* f0, uint64_t* aux7_host);

/* ---- end-to-end prove (next row N2) -------------------------------------------------------------- */
/* deep_fri_prove (fri.rs:601-641) with DeepAliRealBuilder::default().  If f0 != NULL the builder is
 * skipped and a,s,e,t are ignored ("prove given f0").  The proof is returned in the canonical byte
 * encoding (DESIGN.md "Proof encoding"). */
int32_t stark_deep_fri_prove(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0,
                             size_t n0, const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out);
int32_t stark_deep_fri_prove_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* f0,
                                 size_t n0, const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out);
/* `batch` independent traces of n0 rows each, one proof per trace (the reference's bench proves one trace after another,
 * channel/benches/end_to_end.rs:229-309).  a, s, e, t: HOST arrays of `batch` DEVICE pointers; out: host array of `batch` proof handles
 * (all NULL on failure).  The serial column sponges of build_f0 (fri.rs:548-557) bound a single prove and keep four waves of the chip
 * busy; the 4 * batch chains of a batch are independent and run in ONE launch, so the stage costs what it costs for one trace.
 * Every proof is byte-identical to stark_deep_fri_prove_dev on that trace alone.  stage_ms(0) of each proof = the shared sponge stage
 * of the whole batch + that trace's merge. */
int32_t stark_deep_fri_prove_batch_dev(stark_ctx_t* ctx, size_t batch, const uint64_t* const* a, const uint64_t* const* s, const uint64_t* const* e, const uint64_t* const* t,
                                       size_t n0, const size_t* schedule, size_t L, size_t r, uint64_t seed_z, stark_proof_t** out);
size_t  stark_proof_len(stark_proof_t* p);
int32_t stark_proof_bytes(stark_proof_t* p, uint8_t* out);
size_t  stark_proof_size_estimate(stark_proof_t* p);                /* deep_fri_proof_size_bytes, fri.rs:764-805 */
double  stark_proof_stage_ms(stark_proof_t* p, int32_t stage);     /* 0 build_f0, 1 fri_build, 2 queries+encode */
int32_t stark_proof_free(stark_proof_t* p);

/* ---- verification (next row N3) --------------------------------------------------------------------
 * deep_fri_verify (fri.rs:643-762) over the canonical proof bytes stark_proof_bytes returns; *accepted = 1 / 0 (the reference
 * returns bool; bytes that do not decode are rejected, inputs on which the reference would panic are rejected).  seed_z is
 * DeepFriParams.seed_z, carried for signature parity (the reference's verifier does not read it).  Host index logic in the
 * library, every hash (leaf pairs, DS nodes) batched onto the GPU kernels of the prover. */
int32_t stark_deep_fri_verify(stark_ctx_t* ctx, const uint8_t* proof, size_t len, const size_t* schedule, size_t L, size_t r, uint64_t seed_z, int32_t* accepted);
/* MerkleProver::new(MerkleChannelCfg::new(cfg_arity).with_tree_label(tree_label)).verify_single / .verify_pairs
 * (merkle/src/lib.rs:800-812, 841-855 over verify_many_ds :587-722 and verify_pairs_ds :723-773); `proof` = the canonical
 * MerkleProof encoding stark_merkle_open returns. */
int32_t stark_merkle_verify_many_ds(stark_ctx_t* ctx, size_t cfg_arity, uint64_t tree_label, const uint64_t* root4, const size_t* indices, size_t k, const uint64_t* values,
                                    const uint8_t* proof, size_t len, int32_t* accepted);
int32_t stark_merkle_verify_pairs_ds(stark_ctx_t* ctx, size_t cfg_arity, uint64_t tree_label, const uint64_t* root4, const size_t* indices, size_t k, const uint64_t* f_vals, const uint64_t* cp_vals,
                                     const uint8_t* proof, size_t len, int32_t* accepted);

/* ---- sum-check consumer (next row N4) ----------------------------------------------------------------
 * prove_plain / verify_plain and the Merkle-folded prove_mf / verify_mf (channel/src/lib.rs:1045-1240) over a witness of 2^k field
 * elements; vk = (k, tree_label[, queries_per_round]) (build_vk_plain / build_vk_mf, :1025-1043).  The witness and every folded
 * layer are committed with MerkleCommitment (commitment/src/lib.rs:60-114: arity 16, parameters "POSEIDON-T17-X5-SEED") on the GPU;
 * the Fiat-Shamir channel (:7-117) is a device-resident transcript.  The proof comes back as a stark_proof_t whose bytes are the
 * bincode 1.x layout of the reference's serde structs ProofPlain / ProofMF (:925-979) — read them with stark_proof_len /
 * stark_proof_bytes.  verify_*: *accepted = 1 / 0; a failed round check (an assert_eq! panic in the reference) is a rejection. */
/* trait CommitmentScheme for MerkleCommitment (commitment/src/lib.rs:13-27, 80-114): commit (arity 16, tree_label = ds_tag, parameters
 * "POSEIDON-T17-X5-SEED") -> a tree handle (root: stark_merkle_root; open: stark_merkle_open); verify over the bytes of stark_merkle_open. */
int32_t stark_commitment_commit(stark_ctx_t* ctx, uint64_t ds_tag, const uint64_t* leaves, size_t n, stark_tree_t** out);
int32_t stark_commitment_verify(stark_ctx_t* ctx, uint64_t ds_tag, const uint64_t* root4, const size_t* indices, size_t k, const uint64_t* values,
                                const uint8_t* proof, size_t len, int32_t* accepted);
/* Mle::evaluate (channel/src/lib.rs:279-295): the multilinear extension of a 2^k table at r (k elements); host pointers. */
int32_t stark_mle_evaluate(stark_ctx_t* ctx, const uint64_t* table, size_t k, const uint64_t* r, uint64_t* out4);
int32_t stark_sumcheck_prove_plain(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, stark_proof_t** out);
int32_t stark_sumcheck_prove_plain_dev(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, stark_proof_t** out);
int32_t stark_sumcheck_verify_plain(stark_ctx_t* ctx, size_t k, uint64_t tree_label, const uint8_t* proof, size_t len, int32_t* accepted);
int32_t stark_sumcheck_prove_mf(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, size_t queries_per_round, stark_proof_t** out);
int32_t stark_sumcheck_prove_mf_dev(stark_ctx_t* ctx, const uint64_t* witness, size_t k, uint64_t tree_label, size_t queries_per_round, stark_proof_t** out);
int32_t stark_sumcheck_verify_mf(stark_ctx_t* ctx, size_t k, uint64_t tree_label, size_t queries_per_round, const uint8_t* proof, size_t len, int32_t* accepted);

/* ---- Transcript (transcript/src/lib.rs:48-117) -------------------------------------------------------
 * Transcript::new(label, transcript::default_params()) / absorb_bytes / absorb_field(s) / challenge / challenges as an object:
 * the 17-lane state and the rate cursor live on the device; absorbs are queued on the host and executed (lazy permute-on-full,
 * :79-88) by one launch when the next challenge is drawn. */
int32_t stark_transcript_new(stark_ctx_t* ctx, const uint8_t* label, size_t label_len, stark_transcript_t** out);
int32_t stark_transcript_absorb_bytes(stark_transcript_t* t, const uint8_t* bytes, size_t n);
int32_t stark_transcript_absorb_fields(stark_transcript_t* t, const uint64_t* fields, size_t n);
int32_t stark_transcript_challenge(stark_transcript_t* t, const uint8_t* label, size_t label_len, uint64_t* out4);
int32_t stark_transcript_challenges(stark_transcript_t* t, const uint8_t* label, size_t label_len, size_t n, uint64_t* out);
int32_t stark_transcript_free(stark_transcript_t* t);

/* ---- One trace sharded over several GPUs ------------------------------------------
 * The commit phase shards by contiguous blocks (folds, leaf hashes and lower Merkle levels are
 * block-local: stark_fri_fold_dev, stark_leaf_pair_hash_dev, stark_merkle_build_dev with first_pos /
 * level0 / stop_at_len).  The pieces below complete the path without callbacks across the ABI:
 *  - stark_ali_merge_shard_dev: deep_ali_merge_evals(_blinded) (deep_ali/src/lib.rs:48-105) on the local
 *    block [j0, j0+n_local) of an n_global-point domain (omega4 NULL => the radix-2 domain generator); *partial4 (host, may be NULL) receives the
 *    block's share of sum_j phi_j w^j/(z - w^j); stark_ali_cstar_from_partials combines the ranks' shares
 *    (c* = (1/n) * sum, lib.rs:44,94).
 *  - stark_ali_challenges: (seed, z, beta) of DeepAliRealBuilder::build_f0 from the four column digests
 *    (fri.rs:551-560, ali_sample_z_beta_fs :511-533); digests16 = H(a),H(s),H(e),H(t) (host), aux12 = seed,z,beta (host).
 *    The column digests themselves are stark_tr_hash_fields_tagged_dev(tag "ALI/A|S|E|T", k = n0, n_hashes = 1).
 *  - query phase of deep_fri_prove (fri.rs:355-466, 613-640) over values that live on other ranks:
 *    plan_create derives every query index from the L+1 roots and lists the values the proof needs
 *    (kind 0: element `index` of layer `which`; kind 1: node `index` at `level` of tree `which`);
 *    the caller collects them (each from its owner) and plan_assemble returns the canonical proof bytes. */
int32_t stark_ali_merge_shard_dev(stark_ctx_t* ctx, const uint64_t* a, const uint64_t* s, const uint64_t* e, const uint64_t* t, const uint64_t* r_opt,
                                  const uint64_t* beta4, const uint64_t* omega4, const uint64_t* z4, size_t n_local, uint64_t j0, size_t n_global,
                                  uint64_t* f0, uint64_t* partial4);
int32_t stark_ali_cstar_from_partials(stark_ctx_t* ctx, const uint64_t* partials, size_t k, size_t n_global, uint64_t* c_star4);
int32_t stark_ali_challenges(stark_ctx_t* ctx, const uint64_t* digests16, size_t n0, uint64_t* aux12);
int32_t stark_fri_plan_create(stark_ctx_t* ctx, const uint64_t* roots, size_t n0, const size_t* schedule, size_t L, size_t r, stark_fri_plan_t** out);
size_t  stark_fri_plan_num_requests(stark_fri_plan_t* p);
int32_t stark_fri_plan_requests(stark_fri_plan_t* p, uint32_t* kind, uint32_t* which, uint32_t* level, uint64_t* index);
int32_t stark_fri_plan_assemble(stark_fri_plan_t* p, const uint64_t* values, size_t n_values, stark_proof_t** out);
int32_t stark_fri_plan_free(stark_fri_plan_t* p);

/* ---- field helpers (crates/field/src/lib.rs) ------------------------------------------------------ */
/* F::get_root_of_unity(2^log_n) — Domain::new's omega (field/src/lib.rs:43-53), FriDomain::new_radix2 (fri.rs:53-56).  Host-only. */
int32_t stark_root_of_unity(int32_t field_id, size_t log_n, uint64_t* out4);
/* compute_powers(base, n) = [1, base, ..., base^(n-1)] (field/src/lib.rs:125-133; Domain::precompute_elements with base = omega). */
int32_t stark_compute_powers(stark_ctx_t* ctx, int32_t field_id, const uint64_t* base4, size_t n, uint64_t* out);
int32_t stark_compute_powers_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* base4, size_t n, uint64_t* out);

/* ---- NTT (crates/fft/src/lib.rs:6-32) ------------------------------------------------------------- */
/* fft_in_place / ifft_in_place: natural order in and out; inverse != 0 includes the n^-1 scaling.
 * coset4 (optional): evaluate on coset4 * <w> (forward) / interpolate from it (inverse). */
int32_t stark_ntt(stark_ctx_t* ctx, int32_t field_id, uint64_t* data, size_t log_n, int32_t inverse, const uint64_t* coset4);
int32_t stark_ntt_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* data, size_t log_n, int32_t inverse, const uint64_t* coset4);
/* LDE: 2^log_n evaluations on <w_n> -> 2^(log_n+log_blowup) evaluations on coset4 * <w_N> (coset4 NULL => 1). */
int32_t stark_lde(stark_ctx_t* ctx, int32_t field_id, const uint64_t* evals, size_t log_n, size_t log_blowup, const uint64_t* coset4, uint64_t* out);
int32_t stark_lde_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* evals, size_t log_n, size_t log_blowup, const uint64_t* coset4, uint64_t* out);
/* Building blocks of the multi-GPU six-step NTT (one process per GPU; the exchange between the two
 * is an all-to-all done by the caller, see stark_mlwe_amd/dist.py):
 *   phase A: `ncols` column NTTs of size 2^log_rows on a row-major [2^log_rows][ncols] slab, then the
 *            twiddle w_N^(col_global * k) (N = 2^log_n, col_global = col0 + local column).
 *   phase B: `nrows` contiguous NTTs of size 2^log_cols (plain stark_ntt batched over rows). */
int32_t stark_ntt_columns_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* slab, size_t log_rows, size_t ncols, size_t col0, size_t log_n, int32_t inverse);
int32_t stark_ntt_rows_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* slab, size_t nrows, size_t log_cols, int32_t inverse, const uint64_t* scale4);
/*   phase A of a COSET transform: as stark_ntt_columns_dev (forward), with x[j] *= shift4^j on load, j = the element's natural
 *            index in the whole vector (row * 2^(log_n-log_rows) + col0 + local column) — one of the 2^log_blowup cosets of an LDE.
 *   stark_permute3_dev: dst (contiguous) = the [d0][d1][d2] array src with its axes permuted to (p0, p1, p2) — the layout
 *            changes on either side of an all-to-all (32-byte elements).
 *   stark_interleave_dev: dst[k*stride + offset] = src[k], k < n — the coset transforms of an LDE into natural order. */
/*   stark_ntt_rows_coset_dev: first local phase of a forward COSET transform on the layout the inverse six-step transform leaves behind
 *            (rows k1 = row0 .. row0+nrows of the [R][C] view c[k1 + R k'], C = 2^log_cols contiguous, R = 2^(log_n-log_cols)):
 *            dst[i][m] = w_n^(k1 m) * sum_k' src[i][k'] shift^(k' R + k1) w_C^(k' m).  src is not modified (all cosets of an LDE start from it);
 *            the second phase is a plain size-R transform over k1 after ONE exchange — no exchange between inverse and forward. */
int32_t stark_ntt_rows_coset_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* src, uint64_t* dst, size_t nrows, size_t log_cols, size_t row0, size_t log_n, const uint64_t* shift4);
/*   stark_lde_sharded_dev: the whole LDE of ONE column block-sharded over the ranks of the context's communicator (stark_comm_init; one rank without a
 *            communicator is allowed): rank q passes its natural-order block of 2^log_n / W evaluations and receives its block of the 2^(log_n+log_blowup)
 *            evaluations on shift * <w_N> — the composition of the building blocks above with FOUR all-to-alls, inside the library, so that a host
 *            without Python (the reference's Rust process) drives a multi-GPU LDE with one call per column.  (dist.py's ShardedLde is the same
 *            composition in Python and stays the form the CPU gloo tests exercise.) */
int32_t stark_lde_sharded_dev(stark_ctx_t* ctx, int32_t field_id, const uint64_t* block, size_t log_n, size_t log_blowup, const uint64_t* shift4, uint64_t* out);
/* Diagnostic: the same phases for `nranks` VIRTUAL ranks on this one GPU, every exchange done as device copies — what checks the index arithmetic of
 * stark_lde_sharded_dev for W > 1 without a second GPU.  evals: the whole 2^log_n vector; out: the whole extended vector (= stark_lde_dev's). */
int32_t stark_diag_lde_sharded_emulated_dev(stark_ctx_t* ctx, int32_t field_id, int32_t nranks, const uint64_t* evals, size_t log_n, size_t log_blowup, const uint64_t* shift4, uint64_t* out);
int32_t stark_ntt_columns_coset_dev(stark_ctx_t* ctx, int32_t field_id, uint64_t* slab, size_t log_rows, size_t ncols, size_t col0, size_t log_n, const uint64_t* shift4);
int32_t stark_permute3_dev(stark_ctx_t* ctx, const uint64_t* src, uint64_t* dst, size_t d0, size_t d1, size_t d2, int32_t p0, int32_t p1, int32_t p2);
int32_t stark_interleave_dev(stark_ctx_t* ctx, const uint64_t* src, uint64_t* dst, size_t n, size_t stride, size_t offset);

/* ---- the communicator (RCCL over xGMI; one process per GPU) ------------------------------------------------------------
 * The exchanges of the path behind the boundary, so that a host without torch can drive several GPUs:
 * rank 0 calls stark_comm_unique_id and passes the 128 bytes to its peers by its own means; every rank then calls
 * stark_comm_init on its context (collective: returns when all ranks have joined).  The data-path calls below are enqueued on
 * the context's stream (no host synchronisation) and must be made by all ranks in the same order.  Without a usable RCCL the
 * calls fail with STARK_ERR_RCCL; nothing else in the library depends on it. */
#define STARK_COMM_ID_BYTES 128
/* Local probe (dlopen + ncclGetVersion, no communication): STARK_OK when RCCL can be bound and reports the major version whose ABI this
 * library was checked against, else STARK_ERR_RCCL; *version_code (may be NULL) = its NCCL_VERSION_CODE.  Ranks agree on the answer BEFORE any
 * of them enters the collective stark_comm_init (a rank that cannot load RCCL must not leave its peers blocked in ncclCommInitRank).
 * STATUS: the N > 1 paths of this section have not run on hardware yet (no multi-GPU node was available to the build). */
int32_t stark_comm_available(int32_t* version_code);
int32_t stark_comm_unique_id(uint8_t* id128);
int32_t stark_comm_init(stark_ctx_t* ctx, int32_t nranks, int32_t rank, const uint8_t* id128);
int32_t stark_comm_destroy(stark_ctx_t* ctx);
int32_t stark_comm_size(stark_ctx_t* ctx);
int32_t stark_comm_rank(stark_ctx_t* ctx);
/* all-to-all: chunk q (bytes_per_peer) of `send` goes to rank q; chunk p of `recv` is what rank p sent here — the row/column
 * transpose of the six-step NTT, one message per peer link.  send != recv. */
int32_t stark_comm_all_to_all_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t bytes_per_peer);
int32_t stark_comm_all_gather_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t bytes);
int32_t stark_comm_all_reduce_u64_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t count);   /* SUM of uint64 words (send == recv allowed) */
int32_t stark_comm_gather_dev(stark_ctx_t* ctx, const void* send, void* recv, size_t bytes, int32_t root);

/* ---- synthetic inputs for benchmarks (DESIGN.md "Synthetic inputs") ------------------------------- */
/* The reference's OWN bench inputs (channel/benches/end_to_end.rs:249-253): ncols vectors of n elements from one
 * StdRng::seed_from_u64(seed) through ark-ff's Fp::rand, written to HOST memory (host-only, no context): with the seed chain of
 * end_to_end.rs:214-253 the prover's size estimate must equal the `proof_bytes` the reference published (benchmarkdata.csv). */
int32_t stark_ref_bench_inputs(uint64_t seed, size_t n, size_t ncols, uint64_t* out);
int32_t stark_synth_column_dev(stark_ctx_t* ctx, uint64_t seed, uint64_t col, size_t i0, size_t n, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* STARK_MLWE_H */
