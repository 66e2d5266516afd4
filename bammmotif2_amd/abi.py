"""ctypes binding of include/bamm_em.h -- the only way Python reaches the HIP path.

There is no fallback: if ``libbamm_em.so`` is missing this module raises, and on a machine
without a gfx950 GPU ``bamm_ctx_create`` fails with BAMM_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbamm_em.so")

OK, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_STATE, ERR_COMM = 0, -1, -2, -3, -4, -5, -6
MAX_ORDER = 10


class BammError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bamm error {code}: {msg}")
        self.code = code


class Packed(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("n_words", C.c_uint64), ("n_exc", C.c_uint64),
                ("total_len", C.c_uint64), ("max_len", C.c_uint32), ("min_len", C.c_uint32),
                ("words", C.POINTER(C.c_uint32)), ("word_off", C.POINTER(C.c_uint64)),
                ("len", C.POINTER(C.c_uint32)), ("exc_off", C.POINTER(C.c_uint64)),
                ("exc_pos", C.POINTER(C.c_uint32)), ("exc_kmer", C.POINTER(C.c_uint32)),
                ("exc_clean", C.POINTER(C.c_uint32))]


class EmParams(C.Structure):
    _fields_ = [("K", C.c_uint32), ("W", C.c_uint32), ("bg_order", C.c_uint32), ("q", C.c_float),
                ("optimize_q", C.c_int32), ("epsilon", C.c_float), ("max_iterations", C.c_uint32),
                ("n_seqs_global", C.c_uint64), ("n_seqs_bound", C.c_uint64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)

# every symbol include/bamm_em.h declares (tests check the library exports all of them)
SYMBOLS = [
    "bamm_last_error", "bamm_version", "bamm_pack_kmers", "bamm_pack_kmer_ptrs", "bamm_pack_codes", "bamm_pack_codes_seeded",
    "bamm_unpack_y", "bamm_packed_free", "bamm_shard_range", "bamm_ctx_create", "bamm_ctx_destroy",
    "bamm_ctx_sync", "bamm_ctx_device_name", "bamm_ctx_set_launch", "bamm_ctx_set_tuning", "bamm_seqs_upload",
    "bamm_seqs_destroy", "bamm_seqs_info", "bamm_em_default_params", "bamm_em_create",
    "bamm_em_destroy", "bamm_em_estep", "bamm_em_mstep", "bamm_em_optimize_q", "bamm_em_iterate",
    "bamm_em_optimize", "bamm_em_mask", "bamm_em_accumulate", "bamm_em_reduce_buffer", "bamm_em_update", "bamm_em_set_reduce_buffer",
    "bamm_em_set_allreduce", "bamm_em_set_comm", "bamm_comm_init_all", "bamm_comm_unique_id", "bamm_comm_init_rank",
    "bamm_comm_info", "bamm_comm_destroy", "bamm_comm_time_allreduce", "bamm_em_comm_mode", "bamm_seqs_from_codes", "bamm_seqs_bg_model", "bamm_sample_negatives", "bamm_rand_stream_draws", "bamm_comm_init_local", "bamm_comm_init_shm", "bamm_comm_abort", "bamm_device_count", "bamm_device_pci_bus_id", "bamm_device_can_access_peer", "bamm_em_get_v", "bamm_em_get_counts", "bamm_em_get_s", "bamm_em_get_q",
    "bamm_em_get_llh", "bamm_em_get_vdiff", "bamm_em_get_iteration", "bamm_em_get_r",
    "bamm_em_get_trace", "bamm_em_kernel_time", "bamm_em_set_kernel_timing", "bamm_em_plan", "bamm_em_plan_mixed", "bamm_seed_from_pwm", "bamm_set_host_threads", "bamm_logodds", "bamm_logodds_subset", "bamm_bg_model", "bamm_calculate_p", "bamm_v_size",
    "bamm_v_offset", "bamm_bg_size",
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m bammmotif2_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, f, i = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_int
    P = C.POINTER
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.bamm_last_error.restype = C.c_char_p
    L.bamm_version.restype = C.c_char_p
    L.bamm_pack_kmers.argtypes = [u64p, u64p, u64, P(P(Packed))]
    L.bamm_pack_kmer_ptrs.argtypes = [P(C.c_void_p), u64p, u64, P(P(Packed))]
    L.bamm_pack_codes.argtypes = [u8p, u64p, u64, i, P(P(Packed))]
    L.bamm_pack_codes_seeded.argtypes = [u8p, u64p, u64, i, u32, P(P(Packed))]
    L.bamm_unpack_y.argtypes = [P(Packed), u32, u32p]
    L.bamm_packed_free.argtypes = [P(Packed)]
    L.bamm_packed_free.restype = None
    L.bamm_shard_range.argtypes = [u32p, u64, u32, u32, u32, P(u64), P(u64)]
    L.bamm_ctx_create.argtypes = [i, vp, P(vp)]
    L.bamm_ctx_destroy.argtypes = [vp]
    L.bamm_ctx_sync.argtypes = [vp]
    L.bamm_ctx_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.bamm_ctx_set_launch.argtypes = [vp, u32, u32]
    L.bamm_ctx_set_tuning.argtypes = [vp, C.c_char_p, i]
    L.bamm_seqs_upload.argtypes = [vp, P(Packed), u64, u64, P(vp)]
    L.bamm_seqs_destroy.argtypes = [vp]
    L.bamm_seqs_info.argtypes = [vp, P(u64), P(u64), P(u32), P(u64)]
    L.bamm_em_default_params.argtypes = [P(EmParams)]
    L.bamm_em_default_params.restype = None
    L.bamm_em_create.argtypes = [vp, vp, P(EmParams), f32p, f32p, f32p, vp, P(vp)]
    for name in ("bamm_em_destroy", "bamm_em_estep", "bamm_em_mstep", "bamm_em_optimize_q",
                 "bamm_em_accumulate", "bamm_em_update"):
        getattr(L, name).argtypes = [vp]
    L.bamm_em_iterate.argtypes = [vp, u32]
    L.bamm_em_optimize.argtypes = [vp, P(u32)]
    L.bamm_em_mask.argtypes = [vp, C.c_float, P(u32), P(C.c_float), P(u64)]
    L.bamm_em_reduce_buffer.argtypes = [vp, P(vp), P(u64)]
    L.bamm_em_set_reduce_buffer.argtypes = [vp, vp, u64]
    L.bamm_em_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.bamm_em_set_comm.argtypes = [vp, vp]
    L.bamm_comm_init_all.argtypes = [P(vp), u32, P(vp)]
    L.bamm_comm_unique_id.argtypes = [vp, C.c_size_t]
    L.bamm_comm_init_rank.argtypes = [vp, vp, u32, u32, P(vp)]
    L.bamm_comm_info.argtypes = [vp, P(u32), P(u32), P(i)]
    L.bamm_comm_destroy.argtypes = [vp]
    L.bamm_comm_abort.argtypes = [vp]
    L.bamm_sample_negatives.argtypes = [vp, vp, u32, u64, i, u64, P(P(Packed)), P(vp)]
    L.bamm_seqs_from_codes.argtypes = [vp, u8p, u64p, u64, i, u32, P(P(Packed)), P(vp)]
    L.bamm_seqs_bg_model.argtypes = [vp, vp, u32, f32p, f32p]
    L.bamm_em_comm_mode.argtypes = [vp, P(i), C.c_char_p, C.c_size_t]
    L.bamm_rand_stream_draws.argtypes = [u32, u64, i, u32, P(C.c_int32), P(i)]
    L.bamm_comm_time_allreduce.argtypes = [vp, u64, u32, P(C.c_float)]
    L.bamm_comm_init_local.argtypes = [P(vp), u32, u64, P(vp)]
    L.bamm_comm_init_shm.argtypes = [vp, C.c_char_p, u32, u32, u64, P(vp)]
    L.bamm_device_count.argtypes = [P(i)]
    L.bamm_device_pci_bus_id.argtypes = [i, C.c_char_p, C.c_size_t]
    L.bamm_device_can_access_peer.argtypes = [i, i, P(i)]
    for name in ("bamm_em_get_v", "bamm_em_get_counts", "bamm_em_get_s"):
        getattr(L, name).argtypes = [vp, f32p]
    for name in ("bamm_em_get_q", "bamm_em_get_llh", "bamm_em_get_vdiff"):
        getattr(L, name).argtypes = [vp, P(f)]
    L.bamm_em_get_iteration.argtypes = [vp, P(u32)]
    L.bamm_em_get_r.argtypes = [vp, u64, u64, f32p, u64]
    L.bamm_em_get_trace.argtypes = [vp, f32p, f32p, f32p, u32, P(u32)]
    L.bamm_em_kernel_time.argtypes = [vp, P(f), P(u32)]
    L.bamm_em_set_kernel_timing.argtypes = [vp, u32]
    L.bamm_em_plan.argtypes = [vp, P(u64), P(u64), P(u32)]
    L.bamm_em_plan_mixed.argtypes = [vp, P(u64)]
    L.bamm_set_host_threads.argtypes = [u32]
    L.bamm_set_host_threads.restype = None
    L.bamm_seed_from_pwm.argtypes = [vp, vp, u32, u32, f32p, f, vp, vp, vp]
    L.bamm_logodds.argtypes = [vp, vp, u32, u32, u32, f32p, f32p, vp, u64, f32p, u64p]
    L.bamm_logodds_subset.argtypes = [vp, vp, vp, u32, u32, u32, f32p, f32p, vp, u64, f32p, u64p]
    L.bamm_bg_model.argtypes = [P(Packed), u32, f32p, f32p]
    L.bamm_calculate_p.argtypes = [f32p, f32p, u32, u32, u32, f32p]
    for name in ("bamm_v_size", "bamm_v_offset"):
        getattr(L, name).argtypes = [u32, u32]
        getattr(L, name).restype = C.c_size_t
    L.bamm_bg_size.argtypes = [u32]
    L.bamm_bg_size.restype = C.c_size_t
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != OK:
        raise BammError(rc, load().bamm_last_error().decode(errors="replace"))
