"""ctypes bindings of include/stark_mlwe.h (one-to-one; no logic)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class StarkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"stark_mlwe error {code}: {msg}")
        self.code = code


def lib_path():
    return os.path.join(_HERE, "libstark_mlwe_hip.so")


_LIB = None

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
szp = C.POINTER(C.c_size_t)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)
i32, u32, u64, sz = C.c_int32, C.c_uint32, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/stark_mlwe.h declaration by declaration
SIGNATURES = {
    "stark_version": (i32, []),
    "stark_ctx_create": (i32, [i32, vp, vpp]),
    "stark_ctx_destroy": (i32, [vp]),
    "stark_ctx_sync": (i32, [vp]),
    "stark_ctx_trim": (i32, [vp]),
    "stark_ctx_cached_bytes": (sz, [vp]),
    "stark_ctx_set_option": (i32, [vp, C.c_char_p, C.c_int64]),
    "stark_last_error": (C.c_char_p, [vp]),
    "stark_malloc": (i32, [vp, sz, vpp]),
    "stark_free": (i32, [vp, vp]),
    "stark_memcpy_h2d": (i32, [vp, vp, vp, sz]),
    "stark_memcpy_d2h": (i32, [vp, vp, vp, sz]),
    "stark_diag_mac_rate": (i32, [vp, C.POINTER(C.c_double)]),
    "stark_timer_start": (i32, [vp]),
    "stark_timer_stop_ms": (i32, [vp, C.POINTER(C.c_float)]),
    "stark_poseidon_params_upload": (i32, [vp, i32, i32, i32, vp, vp, vp, vpp]),
    "stark_poseidon_params_for_width": (i32, [vp, i32, vpp]),
    "stark_poseidon_params_t17_seed": (i32, [vp, C.c_char_p, sz, vpp]),
    "stark_poseidon_params_export": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), vp, vp, vp]),
    "stark_poseidon_params_free": (i32, [vp]),
    "stark_poseidon_permute_batch": (i32, [vp, vp, vp, sz]),
    "stark_poseidon_permute_batch_dev": (i32, [vp, vp, vp, sz]),
    "stark_poseidon_hash_with_ds_dynamic": (i32, [vp, vp, vp, sz, vp, sz, sz, vp]),
    "stark_poseidon_hash_with_ds": (i32, [vp, vp, vp, sz, vp, vp]),
    "stark_poseidon_hash_ds_batch": (i32, [vp, vp, sz, u32, u64, u64, vp, sz, vp]),
    "stark_poseidon_hash_ds_batch_dev": (i32, [vp, vp, sz, u32, u64, u64, vp, sz, vp]),
    "stark_leaf_pair_hash": (i32, [vp, vp, vp, vp, sz, sz, vp]),
    "stark_leaf_pair_hash_dev": (i32, [vp, vp, vp, vp, sz, sz, vp]),
    "stark_tr_hash_fields_tagged": (i32, [vp, vp, C.c_char_p, vp, sz, sz, vp]),
    "stark_tr_hash_fields_tagged_dev": (i32, [vp, vp, C.c_char_p, vp, sz, sz, vp]),
    "stark_merkle_build": (i32, [vp, vp, sz, u64, vp, sz, i32, vp, vpp]),
    "stark_merkle_build_dev": (i32, [vp, vp, sz, u64, vp, sz, i32, vp, u64, u32, i32, vpp]),
    "stark_merkle_num_levels": (i32, [vp]),
    "stark_merkle_level_len": (sz, [vp, i32]),
    "stark_merkle_root": (i32, [vp, vp]),
    "stark_merkle_level": (i32, [vp, i32, vp]),
    "stark_merkle_level_dev": (vp, [vp, i32]),
    "stark_merkle_gather": (i32, [vp, i32, vp, sz, vp]),
    "stark_merkle_open": (i32, [vp, vp, sz, vp, sz, szp]),
    "stark_merkle_free": (i32, [vp]),
    "stark_fri_sample_z": (i32, [vp, vp, u64, sz, sz, vp]),
    "stark_fri_fold": (i32, [vp, vp, sz, vp, sz, vp]),
    "stark_fri_fold_dev": (i32, [vp, vp, sz, vp, sz, vp]),
    "stark_fri_build": (i32, [vp, vp, sz, vp, sz, u64, vpp]),
    "stark_fri_build_dev": (i32, [vp, vp, sz, vp, sz, u64, vpp]),
    "stark_fri_num_layers": (i32, [vp]),
    "stark_fri_layer_len": (sz, [vp, i32]),
    "stark_fri_layer_f": (i32, [vp, i32, vp]),
    "stark_fri_layer_root": (i32, [vp, i32, vp]),
    "stark_fri_layer_z": (i32, [vp, i32, vp]),
    "stark_fri_layer_tree": (vp, [vp, i32]),
    "stark_fri_state_free": (i32, [vp]),
    "stark_ali_merge": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]),
    "stark_ali_merge_dev": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]),
    "stark_build_f0": (i32, [vp, vp, vp, vp, vp, sz, vp, vp]),
    "stark_build_f0_dev": (i32, [vp, vp, vp, vp, vp, sz, vp, vp]),
    "stark_deep_fri_prove": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, sz, sz, u64, vpp]),
    "stark_deep_fri_prove_dev": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, sz, sz, u64, vpp]),
    "stark_deep_fri_prove_batch_dev": (i32, [vp, sz, vp, vp, vp, vp, sz, vp, sz, sz, u64, vp]),
    "stark_proof_len": (sz, [vp]),
    "stark_proof_bytes": (i32, [vp, vp]),
    "stark_proof_size_estimate": (sz, [vp]),
    "stark_proof_stage_ms": (C.c_double, [vp, i32]),
    "stark_proof_free": (i32, [vp]),
    "stark_deep_fri_verify": (i32, [vp, vp, sz, vp, sz, sz, u64, C.POINTER(i32)]),
    "stark_merkle_verify_many_ds": (i32, [vp, sz, u64, vp, vp, sz, vp, vp, sz, C.POINTER(i32)]),
    "stark_merkle_verify_pairs_ds": (i32, [vp, sz, u64, vp, vp, sz, vp, vp, vp, sz, C.POINTER(i32)]),
    "stark_commitment_commit": (i32, [vp, u64, vp, sz, vpp]),
    "stark_commitment_verify": (i32, [vp, u64, vp, vp, sz, vp, vp, sz, C.POINTER(i32)]),
    "stark_mle_evaluate": (i32, [vp, vp, sz, vp, vp]),
    "stark_sumcheck_prove_plain": (i32, [vp, vp, sz, u64, vpp]),
    "stark_sumcheck_prove_plain_dev": (i32, [vp, vp, sz, u64, vpp]),
    "stark_sumcheck_verify_plain": (i32, [vp, sz, u64, vp, sz, C.POINTER(i32)]),
    "stark_sumcheck_prove_mf": (i32, [vp, vp, sz, u64, sz, vpp]),
    "stark_sumcheck_prove_mf_dev": (i32, [vp, vp, sz, u64, sz, vpp]),
    "stark_sumcheck_verify_mf": (i32, [vp, sz, u64, sz, vp, sz, C.POINTER(i32)]),
    "stark_ali_merge_shard_dev": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, u64, sz, vp, vp]),
    "stark_ali_cstar_from_partials": (i32, [vp, vp, sz, sz, vp]),
    "stark_ali_challenges": (i32, [vp, vp, sz, vp]),
    "stark_fri_plan_create": (i32, [vp, vp, sz, vp, sz, sz, vpp]),
    "stark_fri_plan_num_requests": (sz, [vp]),
    "stark_fri_plan_requests": (i32, [vp, vp, vp, vp, vp]),
    "stark_fri_plan_assemble": (i32, [vp, vp, sz, vpp]),
    "stark_fri_plan_free": (i32, [vp]),
    "stark_root_of_unity": (i32, [i32, sz, vp]),
    "stark_compute_powers": (i32, [vp, i32, vp, sz, vp]),
    "stark_compute_powers_dev": (i32, [vp, i32, vp, sz, vp]),
    "stark_ntt": (i32, [vp, i32, vp, sz, i32, vp]),
    "stark_ntt_dev": (i32, [vp, i32, vp, sz, i32, vp]),
    "stark_lde": (i32, [vp, i32, vp, sz, sz, vp, vp]),
    "stark_lde_dev": (i32, [vp, i32, vp, sz, sz, vp, vp]),
    "stark_ntt_columns_dev": (i32, [vp, i32, vp, sz, sz, sz, sz, i32]),
    "stark_ntt_rows_dev": (i32, [vp, i32, vp, sz, sz, i32, vp]),
    "stark_ntt_rows_coset_dev": (i32, [vp, i32, vp, vp, sz, sz, sz, sz, vp]),
    "stark_lde_sharded_dev": (i32, [vp, i32, vp, sz, sz, vp, vp]),
    "stark_diag_lde_sharded_emulated_dev": (i32, [vp, i32, i32, vp, sz, sz, vp, vp]),
    "stark_ntt_columns_coset_dev": (i32, [vp, i32, vp, sz, sz, sz, sz, vp]),
    "stark_permute3_dev": (i32, [vp, vp, vp, sz, sz, sz, i32, i32, i32]),
    "stark_interleave_dev": (i32, [vp, vp, vp, sz, sz, sz]),
    "stark_comm_available": (i32, [C.POINTER(i32)]),
    "stark_comm_unique_id": (i32, [vp]),
    "stark_comm_init": (i32, [vp, i32, i32, vp]),
    "stark_comm_destroy": (i32, [vp]),
    "stark_comm_size": (i32, [vp]),
    "stark_comm_rank": (i32, [vp]),
    "stark_comm_all_to_all_dev": (i32, [vp, vp, vp, sz]),
    "stark_comm_all_gather_dev": (i32, [vp, vp, vp, sz]),
    "stark_comm_all_reduce_u64_dev": (i32, [vp, vp, vp, sz]),
    "stark_comm_gather_dev": (i32, [vp, vp, vp, sz, i32]),
    "stark_transcript_new": (i32, [vp, C.c_char_p, sz, vpp]),
    "stark_transcript_absorb_bytes": (i32, [vp, C.c_char_p, sz]),
    "stark_transcript_absorb_fields": (i32, [vp, vp, sz]),
    "stark_transcript_challenge": (i32, [vp, C.c_char_p, sz, vp]),
    "stark_transcript_challenges": (i32, [vp, C.c_char_p, sz, sz, vp]),
    "stark_transcript_free": (i32, [vp]),
    "stark_ref_bench_inputs": (i32, [u64, sz, sz, vp]),
    "stark_synth_column_dev": (i32, [vp, u64, u64, sz, sz, vp]),
}


def load_library():
    """Load libstark_mlwe_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Two HIP runtimes in
    # one process cannot both own the GPU, so when torch is part of the process (bench.py, tests) it must
    # be loaded FIRST: the dynamic loader then binds this library's libamdhip64.so.7 dependency to the
    # copy torch already mapped (same SONAME) and both share device memory, streams and events.
    try:
        import torch  # noqa: F401
    except Exception:  # a host without torch uses the system ROCm runtime
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
