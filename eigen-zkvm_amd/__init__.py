"""eigen-zkvm_amd -- MI355X-native backend for eigen-zkvm's starky hot path.

This module is only a thin ctypes binding over the C ABI of ``libzkgpu.so`` (include/zkgpu.h);
the product is the HIP library.  Names mirror the reference's Rust seams: ``fft``/``ifft``/
``interpolate`` (starky/src/fft_p.rs:242-355), ``Poseidon.hash`` (poseidon_opt.rs:76),
``LinearHash.hash`` (linearhash.rs:79), ``MerkleTreeGL`` (merklehash.rs:293-457).

There is NO CPU fallback: importing works without a GPU (so symbols can be inspected), but every
compute call goes to the HIP kernels and raises ``ZkError`` if the library or a GPU is missing.
"""
import ctypes as C
import os
import pathlib
import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ["ZKGPU_LIB"]) if os.environ.get("ZKGPU_LIB") else _HERE / "libzkgpu.so"   # ZKGPU_LIB: an experimental build (tools/)
P = 0xFFFFFFFF00000001

EXPORTS = [
    "zk_init", "zk_last_error", "zk_device_count", "zk_gl_modulus", "zk_gl_root_of_unity",
    "zk_gl_poseidon_selfcheck", "zk_dev_alloc", "zk_dev_free", "zk_dev_upload", "zk_dev_download", "zk_dev_sync", "zk_dev_memset", "zk_dev_fill_splitmix", "zk_dev_trim",
    "zk_gl_ntt", "zk_gl_lde", "zk_gl_ntt_dev", "zk_gl_lde_dev", "zk_gl_ntt_passes",
    "zk_gl_poseidon", "zk_gl_linearhash", "zk_gl_linearhash_rows_dev",
    "zk_merkle_n_nodes", "zk_gl_merkelize", "zk_gl_merkelize_dev", "zk_merkle_root", "zk_merkle_nodes", "zk_merkle_elements",
    "zk_merkle_depth", "zk_merkle_group_proof", "zk_merkle_group_proofs", "zk_merkle_elements_dev", "zk_merkle_nodes_dev",
    "zk_merkle_free",
    "zk_transcript_new", "zk_transcript_put", "zk_transcript_put_dev", "zk_transcript_get_field",
    "zk_transcript_get_field_dev", "zk_transcript_get_fields1", "zk_transcript_get_permutations",
    "zk_transcript_free",
    "zk_fri_fold_dev", "zk_fri_transpose_dev", "zk_stark_x_table_dev", "zk_stark_zh_inv_dev",
    "zk_stark_xdivxsub_dev", "zk_stark_lev_dev", "zk_stark_evals_dev", "zk_stark_qsplit_dev",
    "zk_stream_new", "zk_stream_sync", "zk_stream_free", "zk_program_compile", "zk_program_source", "zk_jit_cache_stats", "zk_program_run_dev", "zk_program_run_rows_dev", "zk_program_free",
    "zk_stark_get_pol_dev", "zk_stark_set_pol_dev", "zk_stark_calculate_h1h2_dev", "zk_stark_calculate_z_dev",
    "zk_msm_g1_bn254", "zk_msm_g1_bn254_dev", "zk_g1_bn254_mul_generator_dev",
    "zk_msm_g1_bls12_381", "zk_msm_g1_bls12_381_dev", "zk_g1_bls12_381_mul_generator_dev",
    "zk_msm_g2_bn254", "zk_msm_g2_bn254_dev", "zk_g2_bn254_mul_generator_dev",
    "zk_msm_g2_bls12_381", "zk_msm_g2_bls12_381_dev", "zk_g2_bls12_381_mul_generator_dev",
    "zk_bn128_load_constants", "zk_bn128_poseidon_selfcheck", "zk_bn128_poseidon", "zk_bn128_poseidon_dev", "zk_bn128_linearhash",
    "zk_bn128_merkle_n_nodes", "zk_bn128_merkelize", "zk_bn128_merkelize_dev", "zk_bn128_merkle_root", "zk_bn128_merkle_nodes",
    "zk_bn128_merkle_depth", "zk_bn128_merkle_group_proof", "zk_bn128_merkle_group_proofs", "zk_bn128_merkle_free",
    "zk_bn128_transcript_new", "zk_bn128_transcript_put", "zk_bn128_transcript_get_fields1", "zk_bn128_transcript_get_field",
    "zk_bn128_transcript_get_permutations", "zk_bn128_transcript_free",
    "zk_bls12381_load_constants", "zk_bls12381_poseidon_selfcheck", "zk_bls12381_poseidon", "zk_bls12381_poseidon_dev", "zk_bls12381_linearhash",
    "zk_bls12381_merkle_n_nodes", "zk_bls12381_merkelize", "zk_bls12381_merkelize_dev", "zk_bls12381_merkle_root", "zk_bls12381_merkle_nodes",
    "zk_bls12381_merkle_depth", "zk_bls12381_merkle_group_proof", "zk_bls12381_merkle_group_proofs", "zk_bls12381_merkle_free",
    "zk_bls12381_transcript_new", "zk_bls12381_transcript_put", "zk_bls12381_transcript_get_fields1", "zk_bls12381_transcript_get_field",
    "zk_bls12381_transcript_get_permutations", "zk_bls12381_transcript_free",
    "zk_starkinfo_generate", "zk_stark_setup_new", "zk_stark_setup_const_root", "zk_stark_setup_set_prover_addr", "zk_stark_setup_set_self_check", "zk_stark_verify", "zk_stark_verify_with", "zk_stark_verify_set_reference_compat", "zk_stark_setup_timing", "zk_stark_last_timing", "zk_stark_gen", "zk_stark_gen_dev", "zk_stark_gen_dev_on", "zk_stark_new", "zk_stark_commit_stage", "zk_stark_challenge", "zk_stark_set_challenge", "zk_stark_eval", "zk_stark_calculate_h1h2", "zk_stark_calculate_z", "zk_stark_evals", "zk_stark_fri_prove", "zk_stark_finish", "zk_stark_fri_pol_dev", "zk_stark_tree", "zk_stark_free", "zk_fri_prove_dev", "zk_string_free", "zk_stark_setup_free",
    "zk_msm_g1_bn254_table_bytes", "zk_msm_g1_bn254_table_build_dev", "zk_msm_g1_bn254_table_dev", "zk_msm_g1_bls12_381_table_bytes", "zk_msm_g1_bls12_381_table_build_dev", "zk_msm_g1_bls12_381_table_dev", "zk_msm_g2_bn254_table_bytes", "zk_msm_g2_bn254_table_build_dev", "zk_msm_g2_bn254_table_dev", "zk_msm_g2_bls12_381_table_bytes", "zk_msm_g2_bls12_381_table_build_dev", "zk_msm_g2_bls12_381_table_dev",
    "zk_fr_bn254_ntt", "zk_fr_bn254_ntt_dev", "zk_fr_bls12_381_ntt", "zk_fr_bls12_381_ntt_dev", "zk_fr_bn254_quotient_dev", "zk_fr_bls12_381_quotient_dev",
    "zk_c12_exec_new", "zk_c12_exec_dev", "zk_c12_exec_depth", "zk_c12_exec_free",
    "zk_fq_bn254_convert_dev", "zk_fq_bls12_381_convert_dev", "zk_groth16_setup_new", "zk_groth16_setup_info", "zk_groth16_prove", "zk_groth16_prove_dev", "zk_groth16_wtns_payload", "zk_groth16_setup_free",
]

# include/zkgpu.h enums
OP_ADD, OP_SUB, OP_MUL, OP_COPY = 0, 1, 2, 3
(OPND_TMP, OPND_MEM, OPND_NUMBER, OPND_PUBLIC, OPND_CHALLENGE, OPND_EVAL, OPND_X, OPND_ZI,
 OPND_XDIVXSUBXI, OPND_XDIVXSUBWXI) = range(10)


class Operand(C.Structure):
    """zk_operand"""
    _fields_ = [("kind", C.c_uint8), ("dim", C.c_uint8), ("prime", C.c_uint8), ("buf", C.c_uint8),
                ("id", C.c_uint32), ("stride", C.c_uint32), ("_pad", C.c_uint32), ("value", C.c_uint64)]


class Instr(C.Structure):
    """zk_instr"""
    _fields_ = [("op", C.c_uint32), ("_pad", C.c_uint32), ("dest", Operand), ("src", Operand * 2)]


class EvalCtx(C.Structure):
    """zk_eval_ctx"""
    _fields_ = [("bufs", C.c_void_p * 16), ("publics", C.c_void_p), ("challenges", C.c_void_p),
                ("evals", C.c_void_p), ("x", C.c_void_p), ("zi", C.c_void_p), ("zi_mask", C.c_uint64),
                ("xdivxsubxi", C.c_void_p), ("xdivxsubwxi", C.c_void_p)]


class EvalDesc(C.Structure):
    """zk_eval_desc (include/zkgpu.h)"""
    _fields_ = [("d_buf", C.c_void_p), ("width", C.c_uint64), ("offset", C.c_uint64),
                ("dim", C.c_uint32), ("prime", C.c_uint32)]


class ZkError(RuntimeError):
    pass


def _load():
    if not LIB_PATH.exists():
        raise ZkError(f"{LIB_PATH} is missing: build it with `make -C eigen-zkvm_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    u64p, vp = C.POINTER(C.c_uint64), C.c_void_p
    sig = {
        "zk_init": (C.c_int, [C.c_int]),
        "zk_last_error": (C.c_char_p, []),
        "zk_device_count": (C.c_int, []),
        "zk_gl_modulus": (C.c_uint64, []),
        "zk_gl_root_of_unity": (C.c_uint64, [C.c_uint32]),
        "zk_dev_alloc": (vp, [C.c_size_t]),
        "zk_dev_free": (C.c_int, [vp]),
        "zk_dev_upload": (C.c_int, [vp, vp, C.c_size_t]),
        "zk_dev_fill_splitmix": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp]),
        "zk_dev_download": (C.c_int, [vp, vp, C.c_size_t]),
        "zk_dev_sync": (C.c_int, []),
        "zk_dev_memset": (C.c_int, [vp, C.c_int, C.c_size_t]),
        "zk_dev_trim": (C.c_int, []),
        "zk_gl_ntt": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_int]),
        "zk_gl_lde": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_uint32]),
        "zk_gl_ntt_dev": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp]),
        "zk_gl_lde_dev": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, vp]),
        "zk_gl_ntt_passes": (C.c_int, [C.c_uint32]),
        "zk_gl_poseidon": (C.c_int, [vp, vp, vp, C.c_uint32]),
        "zk_gl_poseidon_selfcheck": (C.c_int, []),
        "zk_gl_linearhash": (C.c_int, [vp, C.c_size_t, vp]),
        "zk_gl_linearhash_rows_dev": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp, vp]),
        "zk_merkle_n_nodes": (C.c_uint64, [C.c_uint64]),
        "zk_gl_merkelize": (vp, [vp, C.c_uint32, C.c_uint64]),
        "zk_gl_merkelize_dev": (vp, [vp, C.c_uint32, C.c_uint64, vp]),
        "zk_merkle_root": (C.c_int, [vp, vp]),
        "zk_merkle_nodes": (C.c_int, [vp, vp]),
        "zk_merkle_elements": (C.c_int, [vp, vp]),
        "zk_merkle_depth": (C.c_uint32, [vp]),
        "zk_merkle_group_proof": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_merkle_group_proofs": (C.c_int, [vp, vp, C.c_uint32, vp, vp]),
        "zk_merkle_elements_dev": (vp, [vp]),
        "zk_merkle_nodes_dev": (vp, [vp]),
        "zk_merkle_free": (C.c_int, [vp]),
        "zk_transcript_new": (vp, []),
        "zk_transcript_put": (C.c_int, [vp, vp, C.c_size_t]),
        "zk_transcript_put_dev": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "zk_transcript_get_field": (C.c_int, [vp, vp]),
        "zk_transcript_get_field_dev": (C.c_int, [vp, vp, vp]),
        "zk_transcript_get_fields1": (C.c_int, [vp, vp]),
        "zk_transcript_get_permutations": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "zk_transcript_free": (C.c_int, [vp]),
        "zk_fri_fold_dev": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp]),
        "zk_fri_transpose_dev": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, vp]),
        "zk_stark_x_table_dev": (C.c_int, [C.c_uint32, C.c_uint64, vp, vp]),
        "zk_stark_zh_inv_dev": (C.c_int, [C.c_uint32, C.c_uint32, vp, vp]),
        "zk_stark_xdivxsub_dev": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, vp]),
        "zk_stark_lev_dev": (C.c_int, [vp, C.c_uint32, C.c_int, vp, vp, vp, vp]),
        "zk_stark_evals_dev": (C.c_int, [C.POINTER(EvalDesc), C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp]),
        "zk_stark_qsplit_dev": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]),
        "zk_bn128_load_constants": (C.c_int, [C.c_char_p]),
        "zk_bn128_poseidon_selfcheck": (C.c_int, [C.c_char_p]),
        "zk_bn128_poseidon": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32, vp]),
        "zk_bn128_poseidon_dev": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, C.c_uint32, vp, vp]),
        "zk_bn128_linearhash": (C.c_int, [vp, C.c_size_t, vp]),
        "zk_bn128_merkle_n_nodes": (C.c_uint64, [C.c_uint64]),
        "zk_bn128_merkelize": (vp, [vp, C.c_uint32, C.c_uint64]),
        "zk_bn128_merkelize_dev": (vp, [vp, C.c_uint32, C.c_uint64, vp]),
        "zk_bn128_merkle_root": (C.c_int, [vp, vp]),
        "zk_bn128_merkle_nodes": (C.c_int, [vp, vp]),
        "zk_bn128_merkle_depth": (C.c_uint32, [vp]),
        "zk_bn128_merkle_group_proof": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_bn128_merkle_group_proofs": (C.c_int, [vp, vp, C.c_uint32, vp, vp]),
        "zk_bn128_merkle_free": (C.c_int, [vp]),
        "zk_bn128_transcript_new": (vp, []),
        "zk_bn128_transcript_put": (C.c_int, [vp, vp, C.c_size_t]),
        "zk_bn128_transcript_get_fields1": (C.c_int, [vp, vp]),
        "zk_bn128_transcript_get_field": (C.c_int, [vp, vp]),
        "zk_bn128_transcript_get_permutations": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "zk_bn128_transcript_free": (C.c_int, [vp]),
        "zk_bls12381_load_constants": (C.c_int, [C.c_char_p]),
        "zk_bls12381_poseidon_selfcheck": (C.c_int, [C.c_char_p]),
        "zk_bls12381_poseidon": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32, vp]),
        "zk_bls12381_poseidon_dev": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, C.c_uint32, vp, vp]),
        "zk_bls12381_linearhash": (C.c_int, [vp, C.c_size_t, vp]),
        "zk_bls12381_merkle_n_nodes": (C.c_uint64, [C.c_uint64]),
        "zk_bls12381_merkelize": (vp, [vp, C.c_uint32, C.c_uint64]),
        "zk_bls12381_merkelize_dev": (vp, [vp, C.c_uint32, C.c_uint64, vp]),
        "zk_bls12381_merkle_root": (C.c_int, [vp, vp]),
        "zk_bls12381_merkle_nodes": (C.c_int, [vp, vp]),
        "zk_bls12381_merkle_depth": (C.c_uint32, [vp]),
        "zk_bls12381_merkle_group_proof": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_bls12381_merkle_group_proofs": (C.c_int, [vp, vp, C.c_uint32, vp, vp]),
        "zk_bls12381_merkle_free": (C.c_int, [vp]),
        "zk_bls12381_transcript_new": (vp, []),
        "zk_bls12381_transcript_put": (C.c_int, [vp, vp, C.c_size_t]),
        "zk_bls12381_transcript_get_fields1": (C.c_int, [vp, vp]),
        "zk_bls12381_transcript_get_field": (C.c_int, [vp, vp]),
        "zk_bls12381_transcript_get_permutations": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "zk_bls12381_transcript_free": (C.c_int, [vp]),
        "zk_starkinfo_generate": (vp, [C.c_char_p, C.c_char_p]),
        "zk_stark_setup_new": (vp, [C.c_char_p, C.c_char_p, vp, C.c_uint64]),
        "zk_stark_setup_const_root": (C.c_int, [vp, vp]),
        "zk_stark_setup_set_prover_addr": (C.c_int, [vp, C.c_char_p]),
        "zk_stark_setup_set_self_check": (C.c_int, [vp, C.c_int]),
        "zk_stark_verify": (C.c_int, [vp, C.c_char_p]),
        "zk_stark_verify_with": (C.c_int, [C.c_char_p, C.c_char_p, vp, C.c_char_p]),
        "zk_stark_verify_set_reference_compat": (C.c_int, [C.c_int]),
        "zk_stark_setup_timing": (C.c_char_p, [vp]),
        "zk_stark_last_timing": (C.c_char_p, [vp]),
        "zk_stark_gen": (vp, [vp, vp, C.c_uint64]),
        "zk_stark_gen_dev": (vp, [vp, vp, C.c_uint64]),
        "zk_stark_gen_dev_on": (vp, [vp, vp, C.c_uint64, vp]),
        "zk_stark_new": (vp, [vp, vp, vp, C.c_uint64, vp]),
        "zk_stark_commit_stage": (C.c_int, [vp, C.c_int, vp]),
        "zk_stark_challenge": (C.c_int, [vp, C.c_int, vp]),
        "zk_stark_set_challenge": (C.c_int, [vp, C.c_int, vp]),
        "zk_stark_eval": (C.c_int, [vp, C.c_int]),
        "zk_stark_calculate_h1h2": (C.c_int, [vp]),
        "zk_stark_calculate_z": (C.c_int, [vp]),
        "zk_stark_evals": (C.c_int, [vp, vp, C.c_uint64]),
        "zk_stark_fri_prove": (C.c_int, [vp]),
        "zk_stark_finish": (vp, [vp]),
        "zk_stark_fri_pol_dev": (vp, [vp]),
        "zk_stark_tree": (vp, [vp, C.c_int]),
        "zk_stark_free": (C.c_int, [vp]),
        "zk_fri_prove_dev": (vp, [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp]),
        "zk_string_free": (None, [vp]),
        "zk_stark_setup_free": (C.c_int, [vp]),
        "zk_msm_g1_bn254_table_bytes": (C.c_size_t, [C.c_uint64]),
        "zk_msm_g1_bn254_table_build_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g1_bn254_table_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp]),
        "zk_msm_g1_bls12_381_table_bytes": (C.c_size_t, [C.c_uint64]),
        "zk_msm_g1_bls12_381_table_build_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g1_bls12_381_table_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp]),
        "zk_msm_g2_bn254_table_bytes": (C.c_size_t, [C.c_uint64]),
        "zk_msm_g2_bn254_table_build_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g2_bn254_table_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp]),
        "zk_msm_g2_bls12_381_table_bytes": (C.c_size_t, [C.c_uint64]),
        "zk_msm_g2_bls12_381_table_build_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g2_bls12_381_table_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp]),
        "zk_fr_bn254_ntt": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_int]),
        "zk_fr_bn254_ntt_dev": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_int, vp]),
        "zk_fr_bls12_381_ntt": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_int]),
        "zk_fr_bls12_381_ntt_dev": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_int, vp]),
        "zk_fr_bn254_quotient_dev": (C.c_int, [vp, vp, vp, C.c_uint32, vp]),
        "zk_fr_bls12_381_quotient_dev": (C.c_int, [vp, vp, vp, C.c_uint32, vp]),
        "zk_fq_bn254_convert_dev": (C.c_int, [vp, C.c_uint64, C.c_int, vp]),
        "zk_fq_bls12_381_convert_dev": (C.c_int, [vp, C.c_uint64, C.c_int, vp]),
        "zk_c12_exec_new": (vp, [C.c_char_p, C.c_size_t, C.c_uint64]),
        "zk_c12_exec_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp, vp]),
        "zk_c12_exec_depth": (C.c_uint64, [vp]),
        "zk_c12_exec_free": (C.c_int, [vp]),
        "zk_groth16_setup_new": (vp, [C.c_char_p, vp, C.c_size_t, vp, C.c_size_t]),
        "zk_groth16_setup_info": (C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "zk_groth16_prove": (vp, [vp, vp, C.c_uint64, vp, vp, vp]),
        "zk_groth16_prove_dev": (vp, [vp, vp, C.c_uint64, vp, vp, vp, vp]),
        "zk_groth16_wtns_payload": (C.c_int, [vp, C.c_size_t, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "zk_groth16_setup_free": (C.c_int, [vp]),
        "zk_msm_g1_bn254": (C.c_int, [vp, vp, C.c_uint64, vp, C.POINTER(C.c_int)]),
        "zk_msm_g1_bn254_dev": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
        "zk_g1_bn254_mul_generator_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g1_bls12_381": (C.c_int, [vp, vp, C.c_uint64, vp, C.POINTER(C.c_int)]),
        "zk_msm_g1_bls12_381_dev": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
        "zk_g1_bls12_381_mul_generator_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g2_bn254": (C.c_int, [vp, vp, C.c_uint64, vp, C.POINTER(C.c_int)]),
        "zk_msm_g2_bn254_dev": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
        "zk_g2_bn254_mul_generator_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_msm_g2_bls12_381": (C.c_int, [vp, vp, C.c_uint64, vp, C.POINTER(C.c_int)]),
        "zk_msm_g2_bls12_381_dev": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
        "zk_g2_bls12_381_mul_generator_dev": (C.c_int, [vp, C.c_uint64, vp, vp]),
        "zk_program_compile": (vp, [C.POINTER(Instr), C.c_uint32]),
        "zk_program_source": (C.c_char_p, [vp]),
        "zk_jit_cache_stats": (None, [vp]),
        "zk_stream_new": (vp, []), "zk_stream_sync": (C.c_int, [vp]), "zk_stream_free": (C.c_int, [vp]),
        "zk_program_run_dev": (C.c_int, [vp, C.POINTER(EvalCtx), C.c_uint32, C.c_uint64, vp]),
        "zk_program_run_rows_dev": (C.c_int, [vp, C.POINTER(EvalCtx), C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, vp]),
        "zk_program_free": (C.c_int, [vp]),
        "zk_stark_get_pol_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, vp, vp]),
        "zk_stark_set_pol_dev": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, vp, vp]),
        "zk_stark_calculate_h1h2_dev": (C.c_int, [vp, vp, C.c_uint64, vp, vp, vp]),
        "zk_stark_calculate_z_dev": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _check(rc):
    if rc != 0:
        raise ZkError(lib().zk_last_error().decode() or "libzkgpu call failed")


def _np(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64).reshape(-1))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def init(device=0):
    _check(lib().zk_init(device))


# ---- fft_p.rs seams (host buffers) -----------------------------------------------------------
def fft(buffsrc, n_pols, nbits):
    """fft_p::fft (fft_p.rs:242): natural-order batched NTT of a row-major [1<<nbits][n_pols] matrix."""
    src = _np(buffsrc); dst = np.empty_like(src)
    _check(lib().zk_gl_ntt(_ptr(src), _ptr(dst), n_pols, nbits, 0)); return dst


def ifft(buffsrc, n_pols, nbits):
    """fft_p::ifft (fft_p.rs:246)."""
    src = _np(buffsrc); dst = np.empty_like(src)
    _check(lib().zk_gl_ntt(_ptr(src), _ptr(dst), n_pols, nbits, 1)); return dst


def interpolate(buffsrc, n_pols, nbits, nbitsext):
    """fft_p::interpolate (fft_p.rs:255): coset LDE, shift 49."""
    src = _np(buffsrc); dst = np.zeros((1 << nbitsext) * n_pols, np.uint64)
    _check(lib().zk_gl_lde(_ptr(src), n_pols, nbits, _ptr(dst), nbitsext)); return dst


# ---- poseidon_opt.rs / linearhash.rs seams ---------------------------------------------------
def poseidon_hash(inp, init_state, out=4):
    """Poseidon::hash(inp, init_state, out) (poseidon_opt.rs:76); wrong lengths are errors (:81-96)."""
    inp, cap = _np(inp), _np(init_state)
    if inp.size != 8:
        raise ZkError(f"Wrong inputs length {inp.size} != 8")
    if cap.size != 4:
        raise ZkError(f"Capacity inputs length {cap.size} != 4")
    o = np.zeros(out, np.uint64)
    _check(lib().zk_gl_poseidon(_ptr(inp), _ptr(cap), _ptr(o), out)); return o


def linearhash(flatvals):
    """LinearHash::hash(flatvals, 0) (linearhash.rs:79)."""
    v = _np(flatvals); o = np.zeros(4, np.uint64)
    _check(lib().zk_gl_linearhash(_ptr(v), v.size, _ptr(o))); return o


# ---- merklehash.rs seam -----------------------------------------------------------------------
class MerkleTreeGL:
    """trait MerkleTree for the GL hash (traits.rs:24-55, merklehash.rs:293-457); device resident."""

    def __init__(self):
        self._h = None; self.width = 0; self.height = 0

    def merkelize(self, buff, width, height):
        buff = _np(buff)
        if buff.size != width * height:
            raise ZkError("merkelize: buffer size != width*height")
        self.free()
        h = lib().zk_gl_merkelize(_ptr(buff), width, height)
        if not h:
            raise ZkError(lib().zk_last_error().decode())
        self._h, self.width, self.height = h, width, height

    def merkelize_dev(self, d_ptr, width, height, stream=0):
        self.free()
        h = lib().zk_gl_merkelize_dev(C.c_void_p(d_ptr), width, height, C.c_void_p(stream))
        if not h:
            raise ZkError(lib().zk_last_error().decode())
        self._h, self.width, self.height = h, width, height

    def root(self):
        o = np.zeros(4, np.uint64); _check(lib().zk_merkle_root(self._h, _ptr(o))); return o

    def nodes(self):
        o = np.zeros(lib().zk_merkle_n_nodes(self.height) * 4, np.uint64)
        _check(lib().zk_merkle_nodes(self._h, _ptr(o))); return o

    def elements(self):
        """the committed rows (to_extend, merklehash.rs:260-265)"""
        o = np.zeros(self.height * self.width, np.uint64)
        _check(lib().zk_merkle_elements(self._h, _ptr(o))); return o

    def get_group_proof(self, idx):
        d = lib().zk_merkle_depth(self._h)
        row = np.zeros(max(1, self.width), np.uint64); path = np.zeros(max(1, d) * 4, np.uint64)
        _check(lib().zk_merkle_group_proof(self._h, idx, _ptr(row), _ptr(path)))
        return row[:self.width], path[:4 * d].reshape(d, 4)

    def get_group_proofs(self, idxs):
        """every opening of a query list in one round trip: [(row, path)] in the order of idxs"""
        d = lib().zk_merkle_depth(self._h); n = len(idxs)
        ix = np.asarray(idxs, dtype=np.uint64)
        rows = np.zeros(max(1, n * self.width), np.uint64); paths = np.zeros(max(1, n * d * 4), np.uint64)
        _check(lib().zk_merkle_group_proofs(self._h, _ptr(ix), n, _ptr(rows), _ptr(paths)))
        return [(rows[q * self.width:(q + 1) * self.width], paths[q * 4 * d:(q + 1) * 4 * d].reshape(d, 4)) for q in range(n)]

    def free(self):
        if self._h:
            lib().zk_merkle_free(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """a non-blocking HIP stream (zk_stream_new); `.handle` is what the `stream` arguments take"""

    def __init__(self):
        self.handle = lib().zk_stream_new()
        if not self.handle:
            raise ZkError(lib().zk_last_error().decode())

    def sync(self):
        _check(lib().zk_stream_sync(self.handle))

    def free(self):
        if self.handle:
            _check(lib().zk_stream_free(self.handle)); self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- device buffers (HBM-resident words) ---------------------------------------------------------
class DevArray:
    """n u64 words in HBM (zk_dev_alloc); `.ptr` is the raw device address."""

    def __init__(self, n_words, src=None, zero=False):
        self.n = int(n_words)
        self.ptr = lib().zk_dev_alloc(max(1, self.n) * 8)
        if not self.ptr:
            raise ZkError(lib().zk_last_error().decode())
        if src is not None:
            a = _np(src)
            assert a.size == self.n
            _check(lib().zk_dev_upload(self.ptr, _ptr(a), self.n * 8))
        elif zero:
            _check(lib().zk_dev_memset(self.ptr, 0, self.n * 8))

    @classmethod
    def from_host(cls, a):
        a = _np(a)
        return cls(a.size, a)

    def to_host(self):
        _check(lib().zk_dev_sync())
        o = np.empty(self.n, np.uint64)
        if self.n:
            _check(lib().zk_dev_download(_ptr(o), self.ptr, self.n * 8))
        return o

    def free(self):
        if self.ptr:
            lib().zk_dev_free(self.ptr); self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- transcript.rs seam ----------------------------------------------------------------------------
class TranscriptGL:
    """trait Transcript (traits.rs:57-63) / TranscriptGL (transcript.rs:8-103); state on the device."""

    def __init__(self):
        self._h = lib().zk_transcript_new()
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())

    def put(self, es):
        v = _np(es)
        _check(lib().zk_transcript_put(self._h, _ptr(v), v.size))

    def put_dev(self, dev, n=None, stream=0):
        _check(lib().zk_transcript_put_dev(self._h, dev.ptr, dev.n if n is None else n, stream))

    def get_field(self):
        o = np.zeros(3, np.uint64); _check(lib().zk_transcript_get_field(self._h, _ptr(o))); return o

    def get_field_dev(self, dev, stream=0):
        _check(lib().zk_transcript_get_field_dev(self._h, dev.ptr, stream))

    def get_fields1(self):
        o = np.zeros(1, np.uint64); _check(lib().zk_transcript_get_fields1(self._h, _ptr(o))); return int(o[0])

    def get_permutations(self, n, nbits):
        o = np.zeros(n, np.uint64)
        _check(lib().zk_transcript_get_permutations(self._h, n, nbits, _ptr(o))); return o

    def __del__(self):
        try:
            if self._h:
                lib().zk_transcript_free(self._h); self._h = None
        except Exception:
            pass


# ---- fri.rs / stark_gen.rs glue (device resident operands) ---------------------------------------
def fri_fold(d_pol, pol_bits, step_bits, d_special_x, shift_inv, stream=0):
    """one step of FRI::prove's fold loop (fri.rs:101-126)"""
    out = DevArray(3 << step_bits)
    _check(lib().zk_fri_fold_dev(d_pol.ptr, pol_bits, step_bits, d_special_x.ptr, shift_inv, out.ptr, stream))
    return out


def fri_transpose(d_pol, n, tbits, stream=0):
    """get_transposed_buffer (fri.rs:299-317)"""
    out = DevArray(3 * n)
    _check(lib().zk_fri_transpose_dev(d_pol.ptr, n, tbits, out.ptr, stream)); return out


def x_table(nbits, shift=1, stream=0):
    out = DevArray(1 << nbits)
    _check(lib().zk_stark_x_table_dev(nbits, shift, out.ptr, stream)); return out


def zh_inv(nbits, extend_bits, stream=0):
    out = DevArray(1 << extend_bits)
    _check(lib().zk_stark_zh_inv_dev(nbits, extend_bits, out.ptr, stream)); return out


def xdivxsub(d_xi, mulw, nbits_ext, stream=0):
    out = DevArray(3 << nbits_ext)
    _check(lib().zk_stark_xdivxsub_dev(d_xi.ptr, mulw, nbits_ext, out.ptr, stream)); return out


def lev(d_xi, nbits, prime, stream=0):
    out, t1, t2 = DevArray(3 << nbits), DevArray(3 << nbits), DevArray(3 << nbits)
    _check(lib().zk_stark_lev_dev(d_xi.ptr, nbits, int(prime), out.ptr, t1.ptr, t2.ptr, stream))
    _check(lib().zk_dev_sync())
    return out


def evals(descs, nbits, ext, d_lev, d_lpev, stream=0):
    """descs: list of (DevArray buf, width, offset, dim, prime)"""
    arr = (EvalDesc * len(descs))(*[EvalDesc(b.ptr, w, o, d, int(p)) for (b, w, o, d, p) in descs])
    out = DevArray(3 * len(descs))
    _check(lib().zk_stark_evals_dev(arr, len(descs), nbits, ext, d_lev.ptr, d_lpev.ptr, out.ptr, stream))
    _check(lib().zk_dev_sync())
    return out


# ---- interpreter.rs seam: compiled step programs ---------------------------------------------------
def opnd(kind, id=0, dim=1, prime=False, buf=0, stride=0, value=0):
    return Operand(kind, dim, int(prime), buf, id, stride, 0, value)


def instr(op, dest, src0, src1=None):
    i = Instr(); i.op = op; i.dest = dest; i.src[0] = src0
    i.src[1] = src1 if src1 is not None else Operand()
    return i


class Program:
    """compile_code + Block::eval (interpreter.rs:187-225, :91-175) as one run-time compiled kernel."""

    def __init__(self, instrs):
        arr = (Instr * len(instrs))(*instrs)
        self._h = lib().zk_program_compile(arr, len(instrs))
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())

    @property
    def source(self):
        return lib().zk_program_source(self._h).decode()

    def run(self, bufs, nbits_domain, next_, publics=None, challenges=None, evals=None, x=None, zi=None,
            xdiv=None, xdivw=None, stream=0, rows=None):
        """rows=(row0, count): evaluate only those rows (zk_program_run_rows_dev)"""
        c = EvalCtx()
        for k, b in bufs.items():
            c.bufs[k] = b.ptr
        p = lambda d: d.ptr if d is not None else None
        c.publics, c.challenges, c.evals, c.x = p(publics), p(challenges), p(evals), p(x)
        c.zi, c.zi_mask = p(zi), (zi.n - 1 if zi is not None else 0)
        c.xdivxsubxi, c.xdivxsubwxi = p(xdiv), p(xdivw)
        if rows is None: _check(lib().zk_program_run_dev(self._h, C.byref(c), nbits_domain, next_, stream))
        else: _check(lib().zk_program_run_rows_dev(self._h, C.byref(c), nbits_domain, next_, rows[0], rows[1], stream))
        _check(lib().zk_dev_sync())

    def __del__(self):
        try:
            if self._h:
                lib().zk_program_free(self._h); self._h = None
        except Exception:
            pass


# ---- BN128-field hashing (verificationHashType "BN128"): poseidon_bn128_opt.rs, linearhash_bn128.rs, ------
# ---- merklehash_bn128.rs, transcript_bn128.rs.  Digests / field elements = 4 u64 raw (Montgomery) limbs.
def _fr(name, field):
    """the C entry point zk_<field>_<name>; field in ("bn128", "bls12381")"""
    if field not in ("bn128", "bls12381"):
        raise ZkError("unknown scalar field " + str(field))
    return getattr(lib(), "zk_%s_%s" % (field, name))


def bn128_tables_selfcheck(path=None, field="bn128"):
    """host-only check of the matrix-pipe tables of `field`'s Poseidon (zk_<field>_poseidon_selfcheck); raises ZkError naming the difference"""
    import pathlib
    p = path or str(pathlib.Path(__file__).resolve().parent / "data" / ("poseidon_%s_constants.bin" % field))
    _check(_fr("poseidon_selfcheck", field)(str(p).encode()))


def bn128_init(path=None, field="bn128"):
    """loads the Poseidon parameter tables of `field` on the current device (once)"""
    import pathlib
    p = path or str(pathlib.Path(__file__).resolve().parent / "data" / ("poseidon_%s_constants.bin" % field))
    _check(_fr("load_constants", field)(str(p).encode()))


def bn128_poseidon(inp, init_state=None, n_out=1, field="bn128"):
    """Poseidon::hash_ex: inp [n_in][4] raw limbs -> [n_out][4]"""
    a = _np(inp).reshape(-1); n_in = a.size // 4
    init = _np(init_state) if init_state is not None else np.zeros(4, np.uint64)
    out = np.zeros(4 * max(1, n_out), np.uint64)
    _check(_fr("poseidon", field)(_ptr(a), n_in, _ptr(init), n_out, _ptr(out)))
    return out.reshape(-1, 4)[:n_out]


def bn128_linearhash(vals, field="bn128"):
    """LinearHashBN128::hash_element_array"""
    v = _np(vals); out = np.zeros(4, np.uint64)
    _check(_fr("linearhash", field)(_ptr(v), v.size, _ptr(out))); return out


class MerkleTreeBN128:
    """trait MerkleTree for MerkleTreeBN128 (merklehash_bn128.rs:139-278)"""

    def __init__(self, field="bn128"):
        self._h, self.width, self.height, self.field = None, 0, 0, field

    def merkelize(self, buff, width, height):
        b = _np(buff)
        self._h = _fr("merkelize", self.field)(_ptr(b), width, height)
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())
        self.width, self.height = width, height

    def merkelize_dev(self, d_ptr, width, height, stream=0):
        self._h = _fr("merkelize_dev", self.field)(d_ptr, width, height, stream)
        if not self._h:
            raise ZkError(lib().zk_last_error().decode())
        self.width, self.height = width, height

    def root(self):
        o = np.zeros(4, np.uint64); _check(_fr("merkle_root", self.field)(self._h, _ptr(o))); return o

    def nodes(self):
        o = np.zeros(4 * _fr("merkle_n_nodes", self.field)(self.height), np.uint64)
        _check(_fr("merkle_nodes", self.field)(self._h, _ptr(o))); return o.reshape(-1, 4)

    def get_group_proof(self, idx):
        depth = _fr("merkle_depth", self.field)(self._h)
        row, path = np.zeros(max(1, self.width), np.uint64), np.zeros(max(1, depth) * 64, np.uint64)
        _check(_fr("merkle_group_proof", self.field)(self._h, idx, _ptr(row), _ptr(path)))
        return row[:self.width], path[:depth * 64].reshape(depth, 16, 4)

    def get_group_proofs(self, idxs):
        depth = _fr("merkle_depth", self.field)(self._h); n = len(idxs)
        ix = np.asarray(idxs, dtype=np.uint64)
        rows, paths = np.zeros(max(1, n * self.width), np.uint64), np.zeros(max(1, n * depth * 64), np.uint64)
        _check(_fr("merkle_group_proofs", self.field)(self._h, _ptr(ix), n, _ptr(rows), _ptr(paths)))
        return [(rows[q * self.width:(q + 1) * self.width], paths[q * 64 * depth:(q + 1) * 64 * depth].reshape(depth, 16, 4)) for q in range(n)]

    def free(self):
        if self._h:
            _fr("merkle_free", self.field)(self._h); self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TranscriptBN128:
    """trait Transcript for TranscriptBN128 (transcript_bn128.rs:51-132)"""

    def __init__(self, field="bn128"):
        self.field = field
        self._h = _fr("transcript_new", field)()

    def put(self, es):
        """es: list of 1-word (Goldilocks value) or 4-word (digest) entries"""
        for e in es:
            v = _np(e)
            _check(_fr("transcript_put", self.field)(self._h, _ptr(v), v.size))

    def get_fields1(self):
        o = np.zeros(1, np.uint64); _check(_fr("transcript_get_fields1", self.field)(self._h, _ptr(o))); return int(o[0])

    def get_field(self):
        o = np.zeros(3, np.uint64); _check(_fr("transcript_get_field", self.field)(self._h, _ptr(o))); return [int(v) for v in o]

    def get_permutations(self, n, nbits):
        o = np.zeros(n, np.uint64)
        _check(_fr("transcript_get_permutations", self.field)(self._h, n, nbits, _ptr(o))); return o

    def __del__(self):
        try:
            if self._h:
                _fr("transcript_free", self.field)(self._h); self._h = None
        except Exception:
            pass


# ---- groth16 multiexp seam (groth16/src/groth16.rs:88-96 -> bellman_ce multiexp) -------------------
_CURVES = {"bn254": 4, "bls12_381": 6}     # 64-bit words per Fq element


def msm_g1(bases, scalars, curve="bn254", group="g1"):
    """sum_i scalars[i] * bases[i] on G1 (or group="g2": the twist, Fq2 coordinates c0 || c1) of `curve`.
    bases: n x pw u64 (affine x||y, Montgomery limbs; pw = 2*nl for G1, 4*nl for G2), scalars: n x 4 u64 canonical.
    Returns (point[pw] u64 Montgomery affine, is_infinity)."""
    pw = _CURVES[curve] * (4 if group == "g2" else 2)
    b, s = _np(bases).reshape(-1), _np(scalars).reshape(-1)
    n = s.size // 4
    if b.size != n * pw or s.size != n * 4:
        raise ZkError("msm: bases/scalars length mismatch")
    out, inf = np.zeros(pw, np.uint64), C.c_int(0)
    _check(getattr(lib(), "zk_msm_%s_%s" % (group, curve))(_ptr(b), _ptr(s), n, _ptr(out), C.byref(inf)))
    return out, bool(inf.value)


def msm_g1_bn254(bases, scalars):
    return msm_g1(bases, scalars, "bn254")


def g1_mul_generator(d_k, curve="bn254", stream=0, group="g1"):
    """bases[i] = [k_i]G for the n non-zero u64 in d_k; returns a DevArray of n * pw words."""
    out = DevArray(d_k.n * _CURVES[curve] * (4 if group == "g2" else 2))
    _check(getattr(lib(), "zk_%s_%s_mul_generator_dev" % (group, curve))(d_k.ptr, d_k.n, out.ptr, stream)); return out


def g1_bn254_mul_generator(d_k, stream=0):
    return g1_mul_generator(d_k, "bn254", stream)


def msm_g1_dev(d_bases, d_scalars, n, curve="bn254", stream=0, group="g1"):
    """device-resident variant; returns a DevArray of pw + 1 words (the point, flag in the low 32 bits of the last)."""
    out = DevArray(_CURVES[curve] * (4 if group == "g2" else 2) + 1, zero=True)
    _check(getattr(lib(), "zk_msm_%s_%s_dev" % (group, curve))(d_bases.ptr, d_scalars.ptr, n, out.ptr, stream)); return out


class MsmTable:
    """Window table of n fixed bases (zk_msm_*_table_*): build once, then sum any sub-range without doublings."""

    def __init__(self, d_bases, n, curve="bn254", group="g1", stream=0):
        self.n, self.curve, self.group = n, curve, group
        self._pfx = "zk_msm_%s_%s_" % (group, curve)
        nbytes = getattr(lib(), self._pfx + "table_bytes")(n)
        self.table = DevArray((nbytes + 7) // 8)
        _check(getattr(lib(), self._pfx + "table_build_dev")(d_bases.ptr, n, self.table.ptr, stream))

    def msm(self, d_scalars, n=None, offset=0, stream=0):
        n = self.n - offset if n is None else n
        out = DevArray(_CURVES[self.curve] * (4 if self.group == "g2" else 2) + 1, zero=True)
        _check(getattr(lib(), self._pfx + "table_dev")(self.table.ptr, self.n, offset, d_scalars.ptr, n, out.ptr, stream)); return out


def msm_g1_bn254_dev(d_bases, d_scalars, n, stream=0):
    return msm_g1_dev(d_bases, d_scalars, n, "bn254", stream)


def qsplit(d_qq1, nbits, nbits_ext, q_dim, q_deg, stream=0):
    out = DevArray((1 << nbits_ext) * q_dim * q_deg, zero=True)
    _check(lib().zk_stark_qsplit_dev(d_qq1.ptr, nbits, q_dim, q_deg, out.ptr, stream)); return out
