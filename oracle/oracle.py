"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  Nothing under mazu_amd/ may (tests/test_boundary.py greps for it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

K2UPOS_DTYPE = np.dtype([("unitig_id", "<u4"), ("pos", "<u4"), ("unitig_len", "<u4"), ("match", "<u4")])
NO_SKEW = 2**64 - 1


def _host_has_v3():
    """AVX2 / BMI2 / FMA / LZCNT / MOVBE on this host (the x86-64-v3 level liboracle_v3.so is compiled for)"""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return {"avx2", "bmi1", "bmi2", "fma", "abm", "movbe", "f16c"} <= have
    except OSError:
        pass
    return False


BUILD_FLAGS = None  # compiler flags of the library that was loaded (reported beside the CPU baseline)


def build_lib(force=False, name="liboracle.so"):
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "sshash_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB, BUILD_FLAGS
    if _LIB is None:
        name = "liboracle_v3.so" if _host_has_v3() and os.environ.get("MZ_ORACLE_PORTABLE") is None else "liboracle.so"
        so = os.path.join(_HERE, name)
        if not os.path.exists(so):
            so = build_lib(name=name)
        BUILD_FLAGS = "gcc -O3 -fopenmp" + (" -march=x86-64-v3" if name.endswith("_v3.so") else "")
        L = C.CDLL(so)
        u64, u32, p = C.c_uint64, C.c_uint32, C.c_void_p
        L.orc_build.restype = p
        L.orc_build.argtypes = [p, u64, p, u64, u32, u32, u64, u64, C.c_int]
        L.orc_free.argtypes = [p]
        L.orc_k2u.restype = C.c_int
        L.orc_k2u.argtypes = [p, u64, p]
        L.orc_dict_k2u.restype = C.c_int
        L.orc_dict_k2u.argtypes = [p, u64, p]
        L.orc_dict_n_duplicates.restype = u64
        L.orc_dict_n_duplicates.argtypes = [p]
        L.orc_k2u_batch.argtypes = [p, p, u64, p, p, C.c_int, C.c_int]
        L.orc_k2u_reads.argtypes = [p, p, p, u64, p, C.c_int, p, p]
        L.orc_reads_cold_lookups.restype = u64
        L.orc_reads_cold_lookups.argtypes = []
        L.orc_validate_self.restype = u64
        L.orc_validate_self.argtypes = [p, C.c_int]
        for name in ("orc_n_unitigs", "orc_n_kmers", "orc_sum_unitigs_len", "orc_n_minimizers",
                     "orc_n_minimizer_occs", "orc_n_kmers_in_skew_index"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [p]
        L.orc_k.restype = u32
        L.orc_k.argtypes = [p]
        for name in ("orc_unitig_len", "orc_unitig_start", "orc_pos_to_id", "orc_get_kmer"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [p, u64]
        L.orc_kmer_rc.restype = u64
        L.orc_kmer_rc.argtypes = [u64, u32]
        L.orc_util_msb.restype = u64
        L.orc_util_msb.argtypes = [u64]
        L.orc_word_touches.restype = C.c_double
        L.orc_word_touches.argtypes = [p, p, u64, p]
        L.orc_ef_roundtrip.restype = C.c_int
        L.orc_ef_roundtrip.argtypes = [p, u64, u64, p]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleSSHash:
    """CPU restatement of SSHash (sshash.rs) + brute-force dictionary."""

    def __init__(self, useq_words, total_len, starts, k, w, skew_param=NO_SKEW, seed=0, with_dict=True):
        self.words = np.ascontiguousarray(useq_words, dtype=np.uint64)
        self.starts = np.ascontiguousarray(starts, dtype=np.uint64)
        self.h = lib().orc_build(_ptr(self.words), int(total_len), _ptr(self.starts), len(self.starts) - 1,
                                 k, w, int(skew_param), int(seed), 1 if with_dict else 0)
        if not self.h:
            raise ValueError("orc_build failed (bad k/w or empty input)")
        self.k_ = k

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.orc_free(self.h)
            self.h = None

    # K2U trait accessors (kphf/mod.rs:58-67)
    def k(self): return lib().orc_k(self.h)
    def n_unitigs(self): return lib().orc_n_unitigs(self.h)
    def n_kmers(self): return lib().orc_n_kmers(self.h)
    def sum_unitigs_len(self): return lib().orc_sum_unitigs_len(self.h)
    def unitig_len(self, i): return lib().orc_unitig_len(self.h, i)
    def unitig_start(self, i): return lib().orc_unitig_start(self.h, i)
    def pos_to_id(self, pos): return lib().orc_pos_to_id(self.h, pos)
    def n_minimizers(self): return lib().orc_n_minimizers(self.h)
    def n_minimizer_occs(self): return lib().orc_n_minimizer_occs(self.h)
    def n_kmers_in_skew_index(self): return lib().orc_n_kmers_in_skew_index(self.h)
    def n_duplicate_kmers(self): return lib().orc_dict_n_duplicates(self.h)
    def get_kmer(self, pos): return lib().orc_get_kmer(self.h, pos)

    def k2u(self, fw_word, use_dict=False):
        out = np.zeros(1, dtype=K2UPOS_DTYPE)
        f = lib().orc_dict_k2u if use_dict else lib().orc_k2u
        f(self.h, int(fw_word), _ptr(out))
        return out[0]

    def k2u_batch(self, fw_words, n_threads=1, use_dict=False):
        fw = np.ascontiguousarray(fw_words, dtype=np.uint64)
        out = np.zeros(len(fw), dtype=K2UPOS_DTYPE)
        nh = C.c_uint64(0)
        lib().orc_k2u_batch(self.h, _ptr(fw), len(fw), _ptr(out), C.byref(nh), n_threads, 1 if use_dict else 0)
        return out, nh.value

    def k2u_reads(self, bases, read_offsets, streaming=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        ro = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        k = self.k_
        lens = (ro[1:] - ro[:-1]).astype(np.int64)
        nw = np.maximum(lens - k + 1, 0)
        ko = np.zeros(len(ro), dtype=np.uint64)
        np.cumsum(nw, out=ko[1:].view(np.int64))
        out = np.zeros(int(ko[-1]), dtype=K2UPOS_DTYPE)
        counts = np.zeros(3, dtype=np.uint64)
        lib().orc_k2u_reads(self.h, _ptr(bases), _ptr(ro), len(ro) - 1, _ptr(ko), 1 if streaming else 0,
                            _ptr(out), _ptr(counts))
        self.last_reads_cold = int(lib().orc_reads_cold_lookups())  # windows that took the whole chain (k2u_cold)
        return out, ko, counts

    def word_touches(self, fw_words):
        """mean 8-byte words read per query in the reference layout (incl. 1 word in, 2 words out), and the hits"""
        fw = np.ascontiguousarray(fw_words, dtype=np.uint64)
        nh = C.c_uint64(0)
        mean = lib().orc_word_touches(self.h, _ptr(fw), len(fw), C.byref(nh))
        return mean, nh.value

    def validate_self(self, use_dict=False):
        return lib().orc_validate_self(self.h, 1 if use_dict else 0)


def kmer_rc(x, k):
    return lib().orc_kmer_rc(int(x), k)


def msb(n):
    return lib().orc_util_msb(int(n))


def ef_roundtrip(xs, u):
    xs = np.ascontiguousarray(xs, dtype=np.uint64)
    out = np.zeros(len(xs), dtype=np.uint64)
    rc = lib().orc_ef_roundtrip(_ptr(xs), len(xs), int(u), _ptr(out))
    return rc, out
