//! rust/stark-mlwe-hip/src/lib.rs — wrappers with the reference's signatures over the C-ABI (`ffi.rs`, generated from
//! include/stark_mlwe.h).  Each function names the reference item whose BODY it replaces (paths relative to the reference
//! checkout); the reference's public types stay as they are, so callers and benches compile unchanged:
//!
//! ```text
//! crates/poseidon/src/lib.rs   permute, permute_dynamic, hash_with_ds, hash_with_ds_dynamic      -> poseidon::*  below
//! crates/merkle/src/lib.rs     MerkleTree::new / new_pairs / open_many, verify_many_ds / verify_pairs_ds -> merkle::*
//! crates/deep_ali/src/fri.rs   fri_fold_layer, compute_s_layer, fri_build_transcript, build_f0, deep_fri_prove, deep_fri_verify -> fri::*
//! crates/fft/src/lib.rs        fft, ifft, fft_in_place, ifft_in_place                            -> fft::*
//! crates/transcript/src/lib.rs Transcript::{new, absorb_bytes, absorb_fields, challenge, challenges} -> transcript::*
//! crates/channel/src/lib.rs    prove_plain, verify_plain, prove_mf, verify_mf                     -> channel::*
//! ```
//! Field elements cross the boundary as they sit in memory: `ark_ff::Fp<MontBackend<_,4>,4>` is `Fp(BigInt<4>([u64;4]),
//! PhantomData)` — 4 little-endian limbs in Montgomery form — so `&[F]` is handed over as `*const u64` without copying.
//! A non-zero status becomes `panic!`, matching the reference's `assert!`/`panic!` behaviour (fri.rs:86-87,
//! merkle/src/lib.rs:148,161); verifiers return `bool`.
//!
//! This image has no Rust toolchain: the crate is source for a maintainer and has not been compiled here.
#![allow(clippy::missing_safety_doc, clippy::too_many_arguments)]

pub mod ffi;

use ark_pallas::Fr as F;
use ffi::*;
use std::cell::RefCell;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;

const _: () = assert!(core::mem::size_of::<F>() == 32 && core::mem::align_of::<F>() == 8);
const _: () = assert!(core::mem::size_of::<ark_bls12_381::Fr>() == 32);

#[inline] fn limbs(xs: &[F]) -> *const u64 { xs.as_ptr() as *const u64 }
#[inline] fn limbs_mut(xs: &mut [F]) -> *mut u64 { xs.as_mut_ptr() as *mut u64 }
#[inline] fn limb1(x: &F) -> *const u64 { x as *const F as *const u64 }

/// One context per host thread (the reference is single-threaded and stateless; distinct contexts are independent).
pub struct Ctx { raw: *mut stark_ctx_t }
impl Ctx {
    /// `device`: HIP ordinal.  Runs on the legacy default stream (ordered against every blocking stream).
    pub fn new(device: i32) -> Self {
        let mut raw = ptr::null_mut();
        let rc = unsafe { stark_ctx_create(device, ptr::null_mut(), &mut raw) };
        if rc != STARK_OK { panic!("stark_mlwe_hip: no usable HIP device (status {rc}); the accelerated path has no CPU fallback") }
        Ctx { raw }
    }
    #[inline] pub fn raw(&self) -> *mut stark_ctx_t { self.raw }
    #[inline] fn chk(&self, rc: i32) {
        if rc != STARK_OK { panic!("stark_mlwe_hip: {}", unsafe { CStr::from_ptr(stark_last_error(self.raw)) }.to_string_lossy()) }
    }
}
impl Drop for Ctx { fn drop(&mut self) { unsafe { stark_ctx_destroy(self.raw); } } }

thread_local! { static CTX: RefCell<Option<Ctx>> = const { RefCell::new(None) }; }
/// Runs `f` with this thread's context (device from `STARK_MLWE_DEVICE`, default 0).
pub fn with_ctx<R>(f: impl FnOnce(&Ctx) -> R) -> R {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if c.is_none() { *c = Some(Ctx::new(std::env::var("STARK_MLWE_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0))); }
        f(c.as_ref().unwrap())
    })
}

/// Uploaded Poseidon constants (`PoseidonParams` / `PoseidonParamsDynamic`, poseidon/src/lib.rs:16-21, 104-114).
pub struct Params { raw: *mut stark_params_t, pub t: usize }
impl Params {
    /// Row-major `mds[i][j]`, `rc_full[r][i]`, `rc_partial[r]`, exactly as the Rust side derived them (BLAKE3).
    pub fn upload(ctx: &Ctx, t: usize, rf: usize, rp: usize, mds: &[F], rc_full: &[F], rc_partial: &[F]) -> Self {
        assert_eq!(mds.len(), t * t); assert_eq!(rc_full.len(), rf * t); assert_eq!(rc_partial.len(), rp);
        let mut raw = ptr::null_mut();
        ctx.chk(unsafe { stark_poseidon_params_upload(ctx.raw, t as i32, rf as i32, rp as i32, limbs(mds), limbs(rc_full), limbs(rc_partial), &mut raw) });
        Params { raw, t }
    }
    /// `poseidon_params_for_width(t)` derived inside the library (same BLAKE3 derivation; compared with the Rust one in tests).
    pub fn for_width(ctx: &Ctx, t: usize) -> Self {
        let mut raw = ptr::null_mut();
        ctx.chk(unsafe { stark_poseidon_params_for_width(ctx.raw, t as i32, &mut raw) });
        Params { raw, t }
    }
    /// `params::generate_params_t17_x5(seed)` (poseidon/src/lib.rs:318-356).
    pub fn t17_seed(ctx: &Ctx, seed: &[u8]) -> Self {
        let mut raw = ptr::null_mut();
        ctx.chk(unsafe { stark_poseidon_params_t17_seed(ctx.raw, seed.as_ptr(), seed.len(), &mut raw) });
        Params { raw, t: 17 }
    }
}
impl Drop for Params { fn drop(&mut self) { unsafe { stark_poseidon_params_free(self.raw); } } }

pub mod poseidon {
    use super::*;
    /// Flattens `PoseidonParams { mds: [[F;17];17], rc_full: [[F;17];8], rc_partial: [F;64] }` and uploads it.
    pub fn upload_static(ctx: &Ctx, mds: &[[F; 17]; 17], rc_full: &[[F; 17]; 8], rc_partial: &[F; 64]) -> Params {
        let m: Vec<F> = mds.iter().flatten().copied().collect();
        let r: Vec<F> = rc_full.iter().flatten().copied().collect();
        Params::upload(ctx, 17, 8, 64, &m, &r, rc_partial)
    }
    /// Flattens `PoseidonParamsDynamic { t, rounds_full, rounds_partial, mds: Vec<Vec<F>>, .. }`.
    pub fn upload_dynamic(ctx: &Ctx, t: usize, rounds_full: usize, rounds_partial: usize, mds: &[Vec<F>], rc_full: &[Vec<F>], rc_partial: &[F]) -> Params {
        let m: Vec<F> = mds.iter().flatten().copied().collect();
        let r: Vec<F> = rc_full.iter().flatten().copied().collect();
        Params::upload(ctx, t, rounds_full, rounds_partial, &m, &r, rc_partial)
    }
    /// `pub fn permute(state: &mut [F; T], params: &PoseidonParams)` — poseidon/src/lib.rs:31-68.
    pub fn permute(ctx: &Ctx, state: &mut [F; 17], params: &Params) {
        ctx.chk(unsafe { stark_poseidon_permute_batch(ctx.raw, params.raw, limbs_mut(&mut state[..]), 1) });
    }
    /// `pub fn permute_dynamic(state: &mut [F], params: &PoseidonParamsDynamic)` — poseidon/src/lib.rs:219-258.
    pub fn permute_dynamic(ctx: &Ctx, state: &mut [F], params: &Params) {
        assert_eq!(state.len(), params.t, "state width mismatch");                       // lib.rs:220
        ctx.chk(unsafe { stark_poseidon_permute_batch(ctx.raw, params.raw, limbs_mut(state), 1) });
    }
    /// Batched form for callers that hold many states (`nstates * t` elements, in place).
    pub fn permute_batch(ctx: &Ctx, states: &mut [F], params: &Params) {
        assert_eq!(states.len() % params.t, 0);
        ctx.chk(unsafe { stark_poseidon_permute_batch(ctx.raw, params.raw, limbs_mut(states), states.len() / params.t) });
    }
    /// `pub fn hash_with_ds(inputs: &[F], ds_tag: F, params: &PoseidonParams) -> F` — poseidon/src/lib.rs:85-100.
    pub fn hash_with_ds(ctx: &Ctx, inputs: &[F], ds_tag: F, params: &Params) -> F {
        let mut out = F::from(0u64);
        ctx.chk(unsafe { stark_poseidon_hash_with_ds(ctx.raw, params.raw, limbs(inputs), inputs.len(), limb1(&ds_tag), &mut out as *mut F as *mut u64) });
        out
    }
    /// `pub fn hash_with_ds_dynamic(ds_fields: &[F], inputs: &[F], params: &PoseidonParamsDynamic) -> F` — poseidon/src/lib.rs:288-312.
    pub fn hash_with_ds_dynamic(ctx: &Ctx, ds_fields: &[F], inputs: &[F], params: &Params) -> F {
        let mut out = F::from(0u64);
        ctx.chk(unsafe { stark_poseidon_hash_with_ds_dynamic(ctx.raw, params.raw, limbs(ds_fields), ds_fields.len(), limbs(inputs), inputs.len(), 1, &mut out as *mut F as *mut u64) });
        out
    }
}

pub mod merkle {
    use super::*;
    /// Device-resident tree behind `MerkleTree { leaves, root, levels, cfg, .. }` (merkle/src/lib.rs:114-128).  The reference's
    /// struct keeps its `pub levels` / `pub root` fields: fill them from `levels()` / `root()` (eagerly in `MerkleTree::new`, or
    /// lazily behind an accessor).
    pub struct Tree { raw: *mut stark_tree_t, pub arity: usize, pub tree_label: u64 }
    impl Tree {
        /// `MerkleTree::new(leaves: Vec<F>, cfg: MerkleChannelCfg)` — merkle/src/lib.rs:147-193.
        pub fn new(ctx: &Ctx, leaves: &[F], arity: usize, tree_label: u64, params: &Params) -> Self {
            let mut raw = ptr::null_mut();
            ctx.chk(unsafe { stark_merkle_build(ctx.raw, params.raw, arity, tree_label, limbs(leaves), leaves.len(), 0, ptr::null(), &mut raw) });
            Tree { raw, arity, tree_label }
        }
        /// `MerkleTree::new_pairs(f_vals, cp_vals, cfg)` — merkle/src/lib.rs:392-445.
        pub fn new_pairs(ctx: &Ctx, f_vals: &[F], cp_vals: &[F], arity: usize, tree_label: u64, params: &Params) -> Self {
            assert_eq!(f_vals.len(), cp_vals.len(), "length mismatch");                   // :399
            let mut raw = ptr::null_mut();
            ctx.chk(unsafe { stark_merkle_build(ctx.raw, params.raw, arity, tree_label, limbs(f_vals), f_vals.len(), 1, limbs(cp_vals), &mut raw) });
            Tree { raw, arity, tree_label }
        }
        pub fn root(&self, ctx: &Ctx) -> F {
            let mut out = F::from(0u64);
            ctx.chk(unsafe { stark_merkle_root(self.raw, &mut out as *mut F as *mut u64) });
            out
        }
        /// `tree.levels` (level 0 = leaf digests).
        pub fn levels(&self, ctx: &Ctx) -> Vec<Vec<F>> {
            let n = unsafe { stark_merkle_num_levels(self.raw) };
            (0..n).map(|l| {
                let len = unsafe { stark_merkle_level_len(self.raw, l) };
                let mut v = vec![F::from(0u64); len];
                ctx.chk(unsafe { stark_merkle_level(self.raw, l, limbs_mut(&mut v)) });
                v
            }).collect()
        }
        /// `open_many` / `open_many_single` (open_union_of_paths, merkle/src/lib.rs:246-315) as the canonical `MerkleProof`
        /// encoding; `decode_merkle_proof` below turns it into the reference's struct fields.
        pub fn open_many(&self, ctx: &Ctx, indices: &[usize]) -> Vec<u8> {
            let mut len = 0usize;
            ctx.chk(unsafe { stark_merkle_open(self.raw, indices.as_ptr(), indices.len(), ptr::null_mut(), 0, &mut len) });
            let mut buf = vec![0u8; len];
            ctx.chk(unsafe { stark_merkle_open(self.raw, indices.as_ptr(), indices.len(), buf.as_mut_ptr(), len, &mut len) });
            buf
        }
    }
    impl Drop for Tree { fn drop(&mut self) { unsafe { stark_merkle_free(self.raw); } } }

    /// `MerkleProver::verify_single` → `verify_many_ds` (merkle/src/lib.rs:587-722, 800-812).
    pub fn verify_single(ctx: &Ctx, cfg_arity: usize, tree_label: u64, root: &F, indices: &[usize], leaves: &[F], proof_bytes: &[u8]) -> bool {
        let mut ok = 0i32;
        ctx.chk(unsafe { stark_merkle_verify_many_ds(ctx.raw, cfg_arity, tree_label, limb1(root), indices.as_ptr(), indices.len(), limbs(leaves), proof_bytes.as_ptr(), proof_bytes.len(), &mut ok) });
        ok == 1
    }
    /// `MerkleProver::verify_pairs` → `verify_pairs_ds` (merkle/src/lib.rs:723-773, 841-855).
    pub fn verify_pairs(ctx: &Ctx, cfg_arity: usize, tree_label: u64, root: &F, indices: &[usize], f_vals: &[F], cp_vals: &[F], proof_bytes: &[u8]) -> bool {
        let mut ok = 0i32;
        ctx.chk(unsafe { stark_merkle_verify_pairs_ds(ctx.raw, cfg_arity, tree_label, limb1(root), indices.as_ptr(), indices.len(), limbs(f_vals), limbs(cp_vals), proof_bytes.as_ptr(), proof_bytes.len(), &mut ok) });
        ok == 1
    }

    /// Fields of `MerkleProof { indices, siblings, group_sizes, arity }` (merkle/src/lib.rs:131-143) from the canonical bytes:
    /// idxs(u64 count + u64s) | u64 levels, per level u64 count + 32-byte canonical LE elements | u64 levels, per level u64 count + bytes | u64 arity.
    pub fn decode_merkle_proof(b: &[u8]) -> Option<(Vec<usize>, Vec<Vec<F>>, Vec<Vec<u8>>, usize)> {
        use ark_serialize::CanonicalDeserialize;
        let mut p = 0usize;
        let mut u64_ = |b: &[u8], p: &mut usize| -> Option<u64> { let s = b.get(*p..*p + 8)?; *p += 8; Some(u64::from_le_bytes(s.try_into().ok()?)) };
        let n = u64_(b, &mut p)? as usize;
        let mut indices = Vec::with_capacity(n.min(1 << 20));
        for _ in 0..n { indices.push(u64_(b, &mut p)? as usize); }
        let nl = u64_(b, &mut p)? as usize;
        let mut siblings = Vec::new();
        for _ in 0..nl {
            let k = u64_(b, &mut p)? as usize; let mut lv = Vec::with_capacity(k.min(1 << 20));
            for _ in 0..k { let s = b.get(p..p + 32)?; p += 32; lv.push(F::deserialize_compressed(s).ok()?); }
            siblings.push(lv);
        }
        let ng = u64_(b, &mut p)? as usize;
        let mut group_sizes = Vec::new();
        for _ in 0..ng { let k = u64_(b, &mut p)? as usize; let s = b.get(p..p + k)?; p += k; group_sizes.push(s.to_vec()); }
        let arity = u64_(b, &mut p)? as usize;
        if p != b.len() { return None; }
        Some((indices, siblings, group_sizes, arity))
    }
}

pub mod fri {
    use super::*;
    /// `pub fn fri_fold_layer(f_l: &[F], z_l: F, m: usize) -> Vec<F>` — fri.rs:85-102.
    pub fn fri_fold_layer(ctx: &Ctx, f_l: &[F], z_l: F, m: usize) -> Vec<F> {
        assert!(m >= 2, "m >= 2");                                                       // fri.rs:86
        assert!(f_l.len() % m == 0, "layer size must be divisible by m");                // fri.rs:87
        let mut out = vec![F::from(0u64); f_l.len() / m];
        ctx.chk(unsafe { stark_fri_fold(ctx.raw, limbs(f_l), f_l.len(), limb1(&z_l), m, limbs_mut(&mut out)) });
        out
    }
    /// `pub fn compute_s_layer(f_l: &[F], z_l: F, m: usize) -> Vec<F>` — fri.rs:123-143: the folded layer, each value repeated m times.
    pub fn compute_s_layer(ctx: &Ctx, f_l: &[F], z_l: F, m: usize) -> Vec<F> {
        let folded = fri_fold_layer(ctx, f_l, z_l, m);
        let mut s = Vec::with_capacity(f_l.len());
        for v in folded { for _ in 0..m { s.push(v); } }
        s
    }
    /// `fri_sample_z_ell(seed_z, level, domain_size)` — fri.rs:59-82.
    pub fn fri_sample_z_ell(ctx: &Ctx, seed_z: u64, level: usize, domain_size: usize) -> F {
        let mut z = F::from(0u64);
        ctx.chk(unsafe { stark_fri_sample_z(ctx.raw, ptr::null_mut(), seed_z, level, domain_size, &mut z as *mut F as *mut u64) });
        z
    }
    /// Device-resident `FriProverState` (fri.rs:210-216): layers, challenges, roots and trees stay on the GPU; the accessors
    /// materialise what the reference's struct fields expose.
    pub struct ProverState { raw: *mut stark_fri_state_t }
    impl ProverState {
        pub fn num_layers(&self) -> usize { unsafe { stark_fri_num_layers(self.raw) as usize } }
        pub fn f_layer(&self, ctx: &Ctx, l: usize) -> Vec<F> {
            let mut v = vec![F::from(0u64); unsafe { stark_fri_layer_len(self.raw, l as i32) }];
            ctx.chk(unsafe { stark_fri_layer_f(self.raw, l as i32, limbs_mut(&mut v)) });
            v
        }
        pub fn root(&self, ctx: &Ctx, l: usize) -> F { let mut r = F::from(0u64); ctx.chk(unsafe { stark_fri_layer_root(self.raw, l as i32, &mut r as *mut F as *mut u64) }); r }
        pub fn z(&self, ctx: &Ctx, l: usize) -> F { let mut r = F::from(0u64); ctx.chk(unsafe { stark_fri_layer_z(self.raw, l as i32, &mut r as *mut F as *mut u64) }); r }
    }
    impl Drop for ProverState { fn drop(&mut self) { unsafe { stark_fri_state_free(self.raw); } } }
    /// `pub fn fri_build_transcript(f0: Vec<F>, domain0: FriDomain, params: &FriProverParams) -> FriProverState` — fri.rs:231-312.
    pub fn fri_build_transcript(ctx: &Ctx, f0: &[F], schedule: &[usize], seed_z: u64) -> ProverState {
        let mut raw = ptr::null_mut();
        ctx.chk(unsafe { stark_fri_build(ctx.raw, limbs(f0), f0.len(), schedule.as_ptr(), schedule.len(), seed_z, &mut raw) });
        ProverState { raw }
    }
    /// `impl DeepAliBuilder for DeepAliRealBuilder { fn build_f0(..) -> Vec<F> }` — fri.rs:535-569 (default builder: no blinding).
    pub fn build_f0(ctx: &Ctx, a: &[F], s: &[F], e: &[F], t: &[F], n0: usize) -> Vec<F> {
        assert!(a.len() == n0 && s.len() == n0 && e.len() == n0 && t.len() == n0);
        let mut f0 = vec![F::from(0u64); n0];
        ctx.chk(unsafe { stark_build_f0(ctx.raw, limbs(a), limbs(s), limbs(e), limbs(t), n0, limbs_mut(&mut f0), ptr::null_mut()) });
        f0
    }
    /// `pub fn deep_fri_prove<B: DeepAliBuilder>(builder, a, s, e, t, n0, params) -> DeepFriProof` — fri.rs:601-641, for
    /// `DeepAliRealBuilder::default()`.  Returns the canonical proof bytes and `deep_fri_proof_size_bytes`
    /// (fri.rs:764-805); `decode_deep_fri_proof` (a mechanical walk of the encoding, fields in declaration order of `DeepFriProof`)
    /// rebuilds the struct for callers that read its fields.
    pub fn deep_fri_prove(ctx: &Ctx, a: &[F], s: &[F], e: &[F], t: &[F], n0: usize, schedule: &[usize], r: usize, seed_z: u64) -> (Vec<u8>, usize) {
        let mut raw: *mut stark_proof_t = ptr::null_mut();
        ctx.chk(unsafe { stark_deep_fri_prove(ctx.raw, limbs(a), limbs(s), limbs(e), limbs(t), ptr::null(), n0, schedule.as_ptr(), schedule.len(), r, seed_z, &mut raw) });
        let mut bytes = vec![0u8; unsafe { stark_proof_len(raw) }];
        ctx.chk(unsafe { stark_proof_bytes(raw, bytes.as_mut_ptr()) });
        let est = unsafe { stark_proof_size_estimate(raw) };
        unsafe { stark_proof_free(raw); }
        (bytes, est)
    }
    /// The bench loop of `channel/benches/end_to_end.rs:229-309` proves one trace after another; `traces[p] = (a, s, e, t)` as DEVICE
    /// pointers of independent n0-row traces are proven in one call: the 4 * B serial column sponges of `build_f0` (fri.rs:548-557) run
    /// concurrently, every proof is byte-identical to `deep_fri_prove` of that trace alone.
    pub unsafe fn deep_fri_prove_batch_dev(ctx: &Ctx, traces: &[[*const u64; 4]], n0: usize, schedule: &[usize], r: usize, seed_z: u64) -> Vec<(Vec<u8>, usize)> {
        let col = |c: usize| traces.iter().map(|t| t[c]).collect::<Vec<_>>();
        let (a, s, e, t) = (col(0), col(1), col(2), col(3));
        let mut raw: Vec<*mut stark_proof_t> = vec![ptr::null_mut(); traces.len()];
        ctx.chk(stark_deep_fri_prove_batch_dev(ctx.raw, traces.len(), a.as_ptr(), s.as_ptr(), e.as_ptr(), t.as_ptr(), n0, schedule.as_ptr(), schedule.len(), r, seed_z, raw.as_mut_ptr()));
        raw.into_iter().map(|h| {
            let mut bytes = vec![0u8; stark_proof_len(h)];
            ctx.chk(stark_proof_bytes(h, bytes.as_mut_ptr()));
            let est = stark_proof_size_estimate(h);
            stark_proof_free(h);
            (bytes, est)
        }).collect()
    }
    /// `pub fn deep_fri_verify(params: &DeepFriParams, proof: &DeepFriProof) -> bool` — fri.rs:643-762, over the canonical bytes.
    pub fn deep_fri_verify(ctx: &Ctx, schedule: &[usize], r: usize, seed_z: u64, proof_bytes: &[u8]) -> bool {
        let mut ok = 0i32;
        ctx.chk(unsafe { stark_deep_fri_verify(ctx.raw, proof_bytes.as_ptr(), proof_bytes.len(), schedule.as_ptr(), schedule.len(), r, seed_z, &mut ok) });
        ok == 1
    }
    /// `deep_ali_merge_evals(a, s, e, t, omega, z) -> (f0, z, c*)` — deep_ali/src/lib.rs:48-105.
    pub fn deep_ali_merge_evals(ctx: &Ctx, a: &[F], s: &[F], e: &[F], t: &[F], omega: F, z: F) -> (Vec<F>, F, F) {
        let n = a.len();
        let mut f0 = vec![F::from(0u64); n]; let mut c_star = F::from(0u64);
        ctx.chk(unsafe { stark_ali_merge(ctx.raw, limbs(a), limbs(s), limbs(e), limbs(t), ptr::null(), ptr::null(), limb1(&omega), limb1(&z), n, limbs_mut(&mut f0), &mut c_star as *mut F as *mut u64) });
        (f0, z, c_star)
    }
}

pub mod fft {
    //! crates/fft/src/lib.rs:6-32 — the `fft` crate's field is BLS12-381 Fr; the prover field (Pallas Fr) uses the same entry
    //! points with `STARK_FIELD_PALLAS_FR`.
    use super::*;
    use ark_bls12_381::Fr as FB;
    fn run(ctx: &Ctx, data: &mut [FB], inverse: bool) {
        assert!(data.len().is_power_of_two(), "radix-2 domain");
        ctx.chk(unsafe { stark_ntt(ctx.raw, STARK_FIELD_BLS12_381_FR, data.as_mut_ptr() as *mut u64, data.len().trailing_zeros() as usize, inverse as i32, ptr::null()) });
    }
    /// `pub fn fft_in_place(domain, coeffs: &mut Vec<F>)` — fft/src/lib.rs:14-19.
    pub fn fft_in_place(ctx: &Ctx, coeffs: &mut [FB]) { run(ctx, coeffs, false) }
    /// `pub fn ifft_in_place(domain, evals: &mut Vec<F>)` — fft/src/lib.rs:6-11 (includes the n^-1 scaling).
    pub fn ifft_in_place(ctx: &Ctx, evals: &mut [FB]) { run(ctx, evals, true) }
    /// `pub fn fft(domain, coeffs: &[F]) -> Vec<F>` — fft/src/lib.rs:22-26.
    pub fn fft(ctx: &Ctx, coeffs: &[FB]) -> Vec<FB> { let mut v = coeffs.to_vec(); run(ctx, &mut v, false); v }
    /// `pub fn ifft(domain, evals: &[F]) -> Vec<F>` — fft/src/lib.rs:28-32.
    pub fn ifft(ctx: &Ctx, evals: &[FB]) -> Vec<FB> { let mut v = evals.to_vec(); run(ctx, &mut v, true); v }
    /// LDE of `evals` (Pallas Fr) to a 2^log_blowup times larger coset domain `coset * <w_N>` (definition of this build).
    pub fn lde(ctx: &Ctx, evals: &[F], log_blowup: usize, coset: Option<F>) -> Vec<F> {
        assert!(evals.len().is_power_of_two());
        let mut out = vec![F::from(0u64); evals.len() << log_blowup];
        let c = coset.as_ref().map(limb1).unwrap_or(ptr::null());
        ctx.chk(unsafe { stark_lde(ctx.raw, STARK_FIELD_PALLAS_FR, limbs(evals), evals.len().trailing_zeros() as usize, log_blowup, c, limbs_mut(&mut out)) });
        out
    }
}

pub mod transcript {
    //! `Transcript` (transcript/src/lib.rs:48-117) with the state on the device: absorbs are queued and executed by one launch
    //! when the next challenge is drawn.  Parameters are `transcript::default_params()`.
    use super::*;
    pub struct Transcript { raw: *mut stark_transcript_t }
    impl Transcript {
        /// `Transcript::new(label, default_params())` — :55-65.
        pub fn new(ctx: &Ctx, label: &[u8]) -> Self {
            let mut raw = ptr::null_mut();
            ctx.chk(unsafe { stark_transcript_new(ctx.raw(), label.as_ptr(), label.len(), &mut raw) });
            Transcript { raw }
        }
        /// `absorb_bytes` — :67-73.
        pub fn absorb_bytes(&mut self, ctx: &Ctx, bytes: &[u8]) { ctx.chk(unsafe { stark_transcript_absorb_bytes(self.raw, bytes.as_ptr(), bytes.len()) }); }
        /// `absorb_field` / `absorb_fields` — :75-88.
        pub fn absorb_fields(&mut self, ctx: &Ctx, xs: &[F]) { ctx.chk(unsafe { stark_transcript_absorb_fields(self.raw, limbs(xs), xs.len()) }); }
        /// `challenge(label)` — :92-101.
        pub fn challenge(&mut self, ctx: &Ctx, label: &[u8]) -> F {
            let mut r = F::from(0u64);
            ctx.chk(unsafe { stark_transcript_challenge(self.raw, label.as_ptr(), label.len(), &mut r as *mut F as *mut u64) });
            r
        }
        /// `challenges(label, n)` — :103-112.
        pub fn challenges(&mut self, ctx: &Ctx, label: &[u8], n: usize) -> Vec<F> {
            let mut out = vec![F::from(0u64); n];
            ctx.chk(unsafe { stark_transcript_challenges(self.raw, label.as_ptr(), label.len(), n, limbs_mut(&mut out)) });
            out
        }
    }
    impl Drop for Transcript { fn drop(&mut self) { unsafe { stark_transcript_free(self.raw); } } }
}

pub mod channel {
    //! crates/channel/src/lib.rs:1045-1240 — the NIZK entry points of the sum-check consumer.  The returned bytes are the bincode 1.x
    //! layout of `ProofPlain` / `ProofMF`, so `bincode::deserialize::<ProofPlain>(&bytes)` yields the reference's struct.
    use super::*;
    fn take(ctx: &Ctx, raw: *mut stark_proof_t) -> Vec<u8> {
        let mut bytes = vec![0u8; unsafe { stark_proof_len(raw) }];
        ctx.chk(unsafe { stark_proof_bytes(raw, bytes.as_mut_ptr()) });
        unsafe { stark_proof_free(raw); }
        bytes
    }
    /// `prove_plain(vk, witness)` — :1045-1076 (vk = build_vk_plain(k, F::from(tree_label))).
    pub fn prove_plain(ctx: &Ctx, k: usize, tree_label: u64, witness: &[F]) -> Vec<u8> {
        assert!(witness.len() == 1 << k, "MLE length must be 2^k");                        // :259
        let mut raw = ptr::null_mut();
        ctx.chk(unsafe { stark_sumcheck_prove_plain(ctx.raw(), limbs(witness), k, tree_label, &mut raw) });
        take(ctx, raw)
    }
    /// `verify_plain(vk, proof)` — :1080-1128.
    pub fn verify_plain(ctx: &Ctx, k: usize, tree_label: u64, proof: &[u8]) -> bool {
        let mut ok = 0i32;
        ctx.chk(unsafe { stark_sumcheck_verify_plain(ctx.raw(), k, tree_label, proof.as_ptr(), proof.len(), &mut ok) });
        ok == 1
    }
    /// `prove_mf(vk, witness)` — :1130-1172 (vk = build_vk_mf(k, F::from(tree_label), queries_per_round)).
    pub fn prove_mf(ctx: &Ctx, k: usize, tree_label: u64, queries_per_round: usize, witness: &[F]) -> Vec<u8> {
        assert!(witness.len() == 1 << k, "MLE length must be 2^k");
        let mut raw = ptr::null_mut();
        ctx.chk(unsafe { stark_sumcheck_prove_mf(ctx.raw(), limbs(witness), k, tree_label, queries_per_round, &mut raw) });
        take(ctx, raw)
    }
    /// `verify_mf(vk, proof)` — :1176-1240.
    pub fn verify_mf(ctx: &Ctx, k: usize, tree_label: u64, queries_per_round: usize, proof: &[u8]) -> bool {
        let mut ok = 0i32;
        ctx.chk(unsafe { stark_sumcheck_verify_mf(ctx.raw(), k, tree_label, queries_per_round, proof.as_ptr(), proof.len(), &mut ok) });
        ok == 1
    }
}

pub mod comm {
    //! The communicator behind the boundary (RCCL over xGMI, one process per GPU): rank 0 makes the id, the host passes the
    //! 128 bytes to its peers (MPI, a socket, a file), every rank joins; the all-to-all of the six-step NTT and the small
    //! gathers of the sharded prove then run on the context's stream (include/stark_mlwe.h "the communicator").
    use super::*;
    pub fn unique_id() -> [u8; STARK_COMM_ID_BYTES] { let mut id = [0u8; STARK_COMM_ID_BYTES]; let rc = unsafe { stark_comm_unique_id(id.as_mut_ptr()) }; assert_eq!(rc, STARK_OK, "RCCL unavailable"); id }
    pub fn init(ctx: &Ctx, nranks: i32, rank: i32, id: &[u8; STARK_COMM_ID_BYTES]) { ctx.chk(unsafe { stark_comm_init(ctx.raw(), nranks, rank, id.as_ptr()) }); }
    /// The LDE of one column of a trace block-sharded over the ranks (this rank's natural-order block in, its block of the extended column out):
    /// four exchanges inside the library — what a multi-GPU `build_f0` front end calls once per column.
    pub unsafe fn lde_sharded_dev(ctx: &Ctx, field_id: i32, block: *const u64, log_n: usize, log_blowup: usize, shift: &[u64; 4], out: *mut u64) {
        ctx.chk(stark_lde_sharded_dev(ctx.raw(), field_id, block, log_n, log_blowup, shift.as_ptr(), out));
    }
    pub unsafe fn all_to_all_dev(ctx: &Ctx, send: *const c_void, recv: *mut c_void, bytes_per_peer: usize) { ctx.chk(stark_comm_all_to_all_dev(ctx.raw(), send, recv, bytes_per_peer)); }
    pub unsafe fn all_gather_dev(ctx: &Ctx, send: *const c_void, recv: *mut c_void, bytes: usize) { ctx.chk(stark_comm_all_gather_dev(ctx.raw(), send, recv, bytes)); }
}
