"""ctypes binding of libmazu_amd.so (the C ABI in include/mazu_amd.h).

The library is mandatory: there is no Python or CPU fallback for the query path.  If the
shared object is missing, import fails loudly with the build command to run.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MZ_LIB") or os.path.join(_HERE, "libmazu_amd.so")  # MZ_LIB: A/B builds of the same ABI

MZ_OK, MZ_EINVAL, MZ_ENOMEM, MZ_EHIP, MZ_EIO, MZ_ENOTMONOTONE, MZ_EEMPTY = range(7)
MZ_NO_SKEW_INDEX = 2**64 - 1


class MazuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mazu_amd error {code}: {msg}")
        self.code = code


class K2UPos(C.Structure):
    _fields_ = [("unitig_id", C.c_uint32), ("pos", C.c_uint32), ("unitig_len", C.c_uint32), ("match", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [("n_hit", C.c_uint64), ("n_miss", C.c_uint64), ("n_skipped", C.c_uint64)]


class BuildOpts(C.Structure):  # mz_build_opts
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("n_threads", C.c_int32), ("minimizer_lines", C.c_int32),
                ("lines_max_load", C.c_uint32), ("direct_bytes_per_key", C.c_uint64)]


class QueryOpts(C.Structure):  # mz_query_opts
    _fields_ = [("struct_size", C.c_uint32), ("point_kernel", C.c_int32), ("reads_lines", C.c_int32), ("reads_resolve", C.c_int32),
                ("reads_kernel", C.c_int32), ("reads_verify", C.c_int32), ("reads_grid", C.c_int32)]


class LineStats(C.Structure):  # mz_line_stats
    _fields_ = [("n_lines", C.c_uint64), ("g", C.c_uint32), ("t", C.c_uint32), ("entries_total", C.c_uint64), ("entries_inline", C.c_uint64),
                ("buckets_spilled", C.c_uint64), ("entries_spilled", C.c_uint64), ("buckets_heavy", C.c_uint64), ("entries_heavy", C.c_uint64),
                ("slots_empty", C.c_uint64), ("lines_unusable", C.c_uint64), ("entries_unusable", C.c_uint64), ("buckets_walk", C.c_uint64)]


class ValidateReport(C.Structure):
    _fields_ = [("n_kmers", C.c_uint64), ("n_fail", C.c_uint64), ("n_duplicate", C.c_uint64), ("first_fail_pos", C.c_uint64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mazu_amd/csrc`.")
    try:  # share torch's HIP runtime when torch is present, so device pointers and streams interoperate
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    u64, u32, p, i = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int
    sig = {
        "mz_last_error": (C.c_char_p, []),
        "mz_abi_version": (u32, []),
        "mz_device_count": (i, []),
        "mz_host_threads": (i, []),
        "mz_unitig_set_from_seqs": (i, [p, u64, u32, p]),
        "mz_unitig_set_from_cf_prefix": (i, [C.c_char_p, p]),
        "mz_unitig_set_from_packed": (i, [p, u64, p, u64, u32, p]),
        "mz_unitig_set_free": (None, [p]),
        "mz_unitig_set_k": (u32, [p]),
        "mz_unitig_set_n_unitigs": (u64, [p]),
        "mz_unitig_set_total_len": (u64, [p]),
        "mz_unitig_set_n_kmers": (u64, [p]),
        "mz_unitig_set_unitig_len": (u64, [p, u64]),
        "mz_unitig_set_unitig_start_pos": (u64, [p, u64]),
        "mz_unitig_set_pos_to_id": (u64, [p, u64]),
        "mz_unitig_set_useq_words": (p, [p]),
        "mz_sshash_from_unitig_set": (i, [p, u32, u64, u64, i, p]),
        "mz_sshash_from_unitig_set_device": (i, [p, u32, u64, u64, i, p]),
        "mz_sshash_from_unitig_set_opts": (i, [p, u32, u64, u64, p, p]),
        "mz_sshash_set_query_opts": (i, [p, p]),
        "mz_sshash_get_query_opts": (i, [p, p]),
        "mz_sshash_free": (None, [p]),
        "mz_sshash_k": (u32, [p]),
        "mz_sshash_n_kmers": (u64, [p]),
        "mz_sshash_sum_unitigs_len": (u64, [p]),
        "mz_sshash_unitig_len": (u64, [p, u64]),
        "mz_sshash_n_unitigs": (u64, [p]),
        "mz_sshash_unitig_seq": (i, [p, u64, p, p, p]),
        "mz_sshash_as_unitig_set": (p, [p]),
        "mz_sshash_w": (u32, [p]),
        "mz_sshash_skew_param": (u64, [p]),
        "mz_sshash_n_minimizers": (u64, [p]),
        "mz_sshash_n_minimizer_occs": (u64, [p]),
        "mz_sshash_n_kmers_in_skew_index": (u64, [p]),
        "mz_sshash_num_bits": (u64, [p]),
        "mz_sshash_minimizer_mphf_kind": (u32, [p]),
        "mz_sshash_has_minimizer_lines": (u32, [p]),
        "mz_sshash_line_stats": (i, [p, p]),
        "mz_sshash_device_bytes": (u64, [p]),
        "mz_sshash_built_on_device": (C.c_uint32, [p]),
        "mz_sshash_to_device": (i, [p, i]),
        "mz_sshash_k2u": (i, [p, u64, u32, p]),
        "mz_sshash_k2u_batch": (i, [p, p, u64, p, p]),
        "mz_sshash_k2u_batch_device": (i, [p, p, u64, p, p, p]),
        "mz_sshash_k2u_batch_compact_device": (i, [p, p, u64, p, p, p, p]),
        "mz_reads_n_windows": (u64, [p, u64, u32]),
        "mz_sshash_k2u_reads": (i, [p, p, p, u64, i, p, p]),
        "mz_sshash_k2u_reads_device": (i, [p, p, u64, p, p, u64, i, p, p, p]),
        "mz_sshash_validate_self": (i, [p, p]),
        "mz_sshash_validate_self_duplicates": (i, [p, p, p, u64, p]),
        "mz_sshash_extract_kmers_device": (i, [p, p, p, u64, p, p]),
        "mz_sshash_save": (i, [p, C.c_char_p]),
        "mz_sshash_load": (i, [C.c_char_p, p]),
        "mz_u2pos_dense_new": (i, [p, u64, p, u64, p]),
        "mz_u2pos_piscem_new": (i, [p, u64, u32, u32, u64, p, u32, u64, p]),
        "mz_u2pos_free": (None, [p]),
        "mz_u2pos_n_unitig_occs": (u64, [p, u64]),
        "mz_u2pos_num_total_occs": (u64, [p]),
        "mz_u2pos_project_device": (i, [p, u32, p, u64, p, p, u64, p]),
        "mz_u2pos_to_device": (i, [p, i]),
        "mz_u2pos_project_device_noalloc": (i, [p, u32, p, u64, p, p, u64, p]),
        "mz_u2pos_noalloc_max_queries": (u64, []),
        "mz_u2pos_check_refs_device": (i, [p, u32, p, u64, u32, p, p]),
        "mz_get_ref_pos_batch": (i, [p, p, p, u64, p, p, u64, p]),
        "mz_streaming_new": (i, [p, p]),
        "mz_streaming_free": (None, [p]),
        "mz_streaming_k2u": (i, [p, u64, u32, p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    L._mz_symbols = tuple(sig)
    return L


lib = _load()


def check(rc):
    if rc != MZ_OK:
        raise MazuError(rc, (lib.mz_last_error() or b"").decode())
