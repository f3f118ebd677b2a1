"""Host-side mirror of mazu's `kphf` module for the SSHash path (reference src/kphf/mod.rs,
src/kphf/sshash.rs, src/unitig_set.rs, src/index/caching.rs): same names, argument meaning
and error behaviour, over the C ABI of include/mazu_amd.h.  Test and bench harness glue --
the index layout is owned by the C++ host library, queries run in the HIP kernels.
"""
import ctypes as C
import enum

import numpy as np

from . import _capi
from ._capi import MZ_NO_SKEW_INDEX, MazuError, check, lib

K2UPOS_DTYPE = np.dtype([("unitig_id", "<u4"), ("pos", "<u4"), ("unitig_len", "<u4"), ("match", "<u4")])


class MatchType(enum.IntEnum):  # kmers crate MatchType
    NoMatch = 0
    IdentityMatch = 1
    TwinMatch = 2
    Skipped = 3  # read window holding a non-ACGT byte (CanonicalKmerIterator skips it)


_CODE = {c: i for i, c in enumerate("ACGT")}
_CODE.update({c.lower(): i for c, i in list(_CODE.items())})


def kmer_word(s):
    """forward 2-bit word of a k-mer string, first base in the LSBs (Kmer::from(&str))."""
    w = 0
    for i, c in enumerate(s):
        w |= _CODE[c] << (2 * i)
    return w


class CanonicalKmer:
    """kmers::CanonicalKmer as used on the path: fw word + k; swap() exchanges fw and rc."""

    def __init__(self, s=None, word=None, k=None):
        if s is not None:
            self.k, self.fw = len(s), kmer_word(s)
        else:
            self.k, self.fw = k, word
        self.rc = _rc(self.fw, self.k)

    @classmethod
    def from_u64(cls, word, k):
        return cls(word=int(word), k=k)

    def swap(self):
        self.fw, self.rc = self.rc, self.fw

    def get_fw_mer(self):
        return self.fw

    def get_canonical_word(self):
        return min(self.fw, self.rc)

    def __len__(self):
        return self.k


def _rc(w, k):
    r = 0
    for i in range(k):
        r |= (3 - ((w >> (2 * i)) & 3)) << (2 * (k - 1 - i))
    return r


class K2UPos:  # kphf/mod.rs:13-19
    __slots__ = ("unitig_id", "unitig_len", "pos", "o")

    def __init__(self, unitig_id, unitig_len, pos, o):
        self.unitig_id, self.unitig_len, self.pos, self.o = int(unitig_id), int(unitig_len), int(pos), MatchType(int(o))

    def __eq__(self, other):
        return (self.unitig_id, self.unitig_len, self.pos, self.o) == (other.unitig_id, other.unitig_len, other.pos, other.o)

    def __repr__(self):
        return f"K2UPos(unitig_id={self.unitig_id}, unitig_len={self.unitig_len}, pos={self.pos}, o={self.o.name})"

    def reverse_match_type(self):  # kphf/mod.rs:22-29
        self.o = {MatchType.IdentityMatch: MatchType.TwinMatch, MatchType.TwinMatch: MatchType.IdentityMatch}.get(self.o, self.o)

    @classmethod
    def from_record(cls, r):
        if int(r["match"]) in (0, 3):
            return None
        return cls(r["unitig_id"], r["unitig_len"], r["pos"], r["match"])


class UnitigSet:  # unitig_set.rs:30-36
    def __init__(self, handle, owned=True):
        self._h, self._owned = handle, owned

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib.mz_unitig_set_free(self._h)
            self._h = None

    @classmethod
    def from_seqs(cls, seqs, k):  # unitig_set.rs:74-106
        arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
        h = C.c_void_p()
        check(lib.mz_unitig_set_from_seqs(arr, len(seqs), k, C.byref(h)))
        return cls(h)

    @classmethod
    def from_cf_prefix(cls, prefix):  # unitig_set.rs:109-165
        h = C.c_void_p()
        check(lib.mz_unitig_set_from_cf_prefix(str(prefix).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_packed(cls, words, total_len, starts, k):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        h = C.c_void_p()
        check(lib.mz_unitig_set_from_packed(words.ctypes.data, int(total_len), starts.ctypes.data, len(starts) - 1, k, C.byref(h)))
        return cls(h)

    def k(self): return lib.mz_unitig_set_k(self._h)
    def n_unitigs(self): return lib.mz_unitig_set_n_unitigs(self._h)
    def total_len(self): return lib.mz_unitig_set_total_len(self._h)
    def n_kmers(self): return lib.mz_unitig_set_n_kmers(self._h)
    def unitig_len(self, i): return lib.mz_unitig_set_unitig_len(self._h, i)
    def unitig_start_pos(self, i): return lib.mz_unitig_set_unitig_start_pos(self._h, i)
    def unitig_end_pos(self, i): return lib.mz_unitig_set_unitig_start_pos(self._h, i + 1)
    def pos_to_id(self, pos): return lib.mz_unitig_set_pos_to_id(self._h, pos)

    def useq_words(self):
        n = (self.total_len() + 31) // 32 + 2
        addr = lib.mz_unitig_set_useq_words(self._h)
        return np.ctypeslib.as_array((C.c_uint64 * n).from_address(addr))

    def unitig_seq(self, i):
        """unitig i as an ACGT string (SeqVectorSlice::to_string)."""
        words = self.useq_words()
        s, n = self.unitig_start_pos(i), self.unitig_len(i)
        return "".join("ACGT"[(int(words[(s + j) >> 5]) >> (2 * ((s + j) & 31))) & 3] for j in range(n))


class SSHash:  # sshash.rs:39-51
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib.mz_sshash_free(self._h)
            self._h = None

    @classmethod
    def from_unitig_set(cls, unitigs, w, skew_param, seed=0, n_threads=0):  # sshash.rs:405-412
        h = C.c_void_p()
        check(lib.mz_sshash_from_unitig_set(unitigs._h, w, int(skew_param), int(seed), n_threads, C.byref(h)))
        return cls(h)

    @classmethod
    def from_unitig_set_on_device(cls, unitigs, w, skew_param, seed=0, device=0):
        """same index, built on the GPU where the configuration allows (bit-identical arrays)"""
        h = C.c_void_p()
        check(lib.mz_sshash_from_unitig_set_device(unitigs._h, w, int(skew_param), int(seed), device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_unitig_set_opts(cls, unitigs, w, skew_param, seed=0, device=-1, n_threads=0, minimizer_lines=0, lines_max_load=0, direct_bytes_per_key=0):
        """construction with the layout choices of mz_build_opts (0 = the size rules); device >= 0 builds on that GPU"""
        o = _capi.BuildOpts(C.sizeof(_capi.BuildOpts), device, n_threads, minimizer_lines, lines_max_load, direct_bytes_per_key)
        h = C.c_void_p()
        check(lib.mz_sshash_from_unitig_set_opts(unitigs._h, w, int(skew_param), int(seed), C.byref(o), C.byref(h)))
        return cls(h)

    POINT_KERNEL = {"wave": 1, "tile": 2, "lane": 3}
    READS_LINES = {"lines": 1, "records": -1}
    READS_RESOLVE = {"item": 1, "run": 2}
    READS_KERNEL = {"wave": 1, "block": 2}
    READS_VERIFY = {"each": 1, "group": 2}

    def set_query_opts(self, point_kernel=None, reads_lines=None, reads_resolve=None, reads_kernel=None, reads_verify=None, reads_grid=0):
        """kernel choices of this handle's query entry points (mz_query_opts); None = the rule.  Results never change."""
        o = _capi.QueryOpts(C.sizeof(_capi.QueryOpts), self.POINT_KERNEL.get(point_kernel, 0), self.READS_LINES.get(reads_lines, 0),
                            self.READS_RESOLVE.get(reads_resolve, 0), self.READS_KERNEL.get(reads_kernel, 0), self.READS_VERIFY.get(reads_verify, 0), reads_grid)
        for name, val, table in (("point_kernel", point_kernel, self.POINT_KERNEL), ("reads_lines", reads_lines, self.READS_LINES),
                                 ("reads_resolve", reads_resolve, self.READS_RESOLVE), ("reads_kernel", reads_kernel, self.READS_KERNEL),
                                 ("reads_verify", reads_verify, self.READS_VERIFY)):
            if val is not None and val not in table:
                raise ValueError(f"{name}={val!r}: one of {sorted(table)}")
        check(lib.mz_sshash_set_query_opts(self._h, C.byref(o)))

    def get_query_opts(self):
        o = _capi.QueryOpts()
        check(lib.mz_sshash_get_query_opts(self._h, C.byref(o)))
        return {f: getattr(o, f) for f, _ in _capi.QueryOpts._fields_ if f != "struct_size"}

    @classmethod
    def from_unitig_set_no_skew_index(cls, unitigs, w, seed=0, n_threads=0):  # sshash.rs:397-403
        return cls.from_unitig_set(unitigs, w, MZ_NO_SKEW_INDEX, seed, n_threads)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        check(lib.mz_sshash_load(str(path).encode(), C.byref(h)))
        return cls(h)

    def save(self, path): check(lib.mz_sshash_save(self._h, str(path).encode()))

    # trait K2U, kphf/mod.rs:58-67
    def k(self): return lib.mz_sshash_k(self._h)
    def n_kmers(self): return lib.mz_sshash_n_kmers(self._h)
    def sum_unitigs_len(self): return lib.mz_sshash_sum_unitigs_len(self._h)
    def unitig_len(self, i): return lib.mz_sshash_unitig_len(self._h, i)
    def n_unitigs(self): return lib.mz_sshash_n_unitigs(self._h)
    def unitig_seq(self, i): return self.as_ref().unitig_seq(i)
    def as_ref(self): return UnitigSet(C.c_void_p(lib.mz_sshash_as_unitig_set(self._h)), owned=False)  # AsRef<UnitigSet>
    def w(self): return lib.mz_sshash_w(self._h)
    def n_minimizers(self): return lib.mz_sshash_n_minimizers(self._h)
    def n_minimizer_occs(self): return lib.mz_sshash_n_minimizer_occs(self._h)
    def n_kmers_in_skew_index(self): return lib.mz_sshash_n_kmers_in_skew_index(self._h)
    def num_bits(self): return lib.mz_sshash_num_bits(self._h)
    def minimizer_mphf_kind(self): return lib.mz_sshash_minimizer_mphf_kind(self._h)
    def has_minimizer_lines(self): return bool(lib.mz_sshash_has_minimizer_lines(self._h))
    def device_bytes(self): return lib.mz_sshash_device_bytes(self._h)
    def built_on_device(self): return bool(lib.mz_sshash_built_on_device(self._h))

    def line_stats(self):
        """where the bucket entries are answered from (mz_line_stats), plus the fractions of the entry mass"""
        o = _capi.LineStats()
        check(lib.mz_sshash_line_stats(self._h, C.byref(o)))
        d = {f: getattr(o, f) for f, _ in _capi.LineStats._fields_}
        tot = max(1, d["entries_total"])
        d.update(frac_inline=d["entries_inline"] / tot, frac_spilled=d["entries_spilled"] / tot, frac_heavy=d["entries_heavy"] / tot,
                 entries_per_line=d["entries_total"] / max(1, d["n_lines"]))
        return d
    def to_device(self, device=0): check(lib.mz_sshash_to_device(self._h, device))

    def k2u(self, km):
        """K2U::k2u(&CanonicalKmer) -> Option<K2UPos> (sshash.rs:471-554); wrong k raises (the reference panics)."""
        out = _capi.K2UPos()
        check(lib.mz_sshash_k2u(self._h, km.get_fw_mer(), len(km), C.byref(out)))
        return None if out.match == 0 else K2UPos(out.unitig_id, out.unitig_len, out.pos, out.match)

    def k2u_batch(self, fw_words, out=None):
        """host arrays in, host records out; `out` lets a caller reuse its result buffer (a fresh one costs a page fault per 4 KiB)"""
        fw = np.ascontiguousarray(fw_words, dtype=np.uint64)
        if out is None:
            out = np.zeros(len(fw), dtype=K2UPOS_DTYPE)
        assert out.dtype == K2UPOS_DTYPE and len(out) >= len(fw) and out.flags.c_contiguous
        cnt = _capi.Counts()
        check(lib.mz_sshash_k2u_batch(self._h, fw.ctypes.data, len(fw), out.ctypes.data, C.byref(cnt)))
        return out, (cnt.n_hit, cnt.n_miss, cnt.n_skipped)

    def k2u_reads(self, bases, read_offsets, streaming=False, out=None):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        ro = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        n_reads = len(ro) - 1
        n_win = lib.mz_reads_n_windows(ro.ctypes.data, n_reads, self.k())
        if out is None:
            out = np.zeros(n_win, dtype=K2UPOS_DTYPE)
        assert out.dtype == K2UPOS_DTYPE and len(out) >= n_win and out.flags.c_contiguous
        cnt = _capi.Counts()
        check(lib.mz_sshash_k2u_reads(self._h, bases.ctypes.data, ro.ctypes.data, n_reads, 1 if streaming else 0,
                                      out.ctypes.data, C.byref(cnt)))
        return out, (cnt.n_hit, cnt.n_miss, cnt.n_skipped)

    # device-pointer forms (torch tensors on the index's device)
    def k2u_batch_device(self, d_fw, d_out, d_counts=None, stream=None):
        check(lib.mz_sshash_k2u_batch_device(self._h, d_fw.data_ptr(), d_fw.numel(), d_out.data_ptr(),
                                             d_counts.data_ptr() if d_counts is not None else None, stream))

    def k2u_batch_compact_device(self, d_fw, d_hit_idx, d_hits, d_n_hits, stream=None):
        check(lib.mz_sshash_k2u_batch_compact_device(self._h, d_fw.data_ptr(), d_fw.numel(), d_hit_idx.data_ptr(),
                                                     d_hits.data_ptr(), d_n_hits.data_ptr(), stream))

    def k2u_reads_device(self, d_bases, d_read_offsets, d_kmer_offsets, d_out, d_counts=None, streaming=False, stream=None):
        check(lib.mz_sshash_k2u_reads_device(self._h, d_bases.data_ptr(), d_bases.numel(), d_read_offsets.data_ptr(),
                                             d_kmer_offsets.data_ptr(), d_read_offsets.numel() - 1, 1 if streaming else 0,
                                             d_out.data_ptr(), d_counts.data_ptr() if d_counts is not None else None, stream))

    def extract_kmers_device(self, d_positions, d_flip, d_fw, stream=None):
        check(lib.mz_sshash_extract_kmers_device(self._h, d_positions.data_ptr(), d_flip.data_ptr() if d_flip is not None else None,
                                                 d_positions.numel(), d_fw.data_ptr(), stream))

    def validate_self_report(self):
        rep = _capi.ValidateReport()
        check(lib.mz_sshash_validate_self(self._h, C.byref(rep)))
        return {"n_kmers": rep.n_kmers, "n_fail": rep.n_fail, "n_duplicate": rep.n_duplicate, "first_fail_pos": rep.first_fail_pos}

    def validate_self_duplicates(self, cap=4096):
        """validate_self_report plus the (sorted) global positions of up to `cap` duplicated k-mers"""
        rep = _capi.ValidateReport()
        buf = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint64(0)
        check(lib.mz_sshash_validate_self_duplicates(self._h, C.byref(rep), buf.ctypes.data, cap, C.byref(n)))
        d = {"n_kmers": rep.n_kmers, "n_fail": rep.n_fail, "n_duplicate": rep.n_duplicate, "first_fail_pos": rep.first_fail_pos}
        return d, np.sort(buf[: n.value])

    def validate_self(self):
        """K2U::validate_self (kphf/mod.rs:69-103): raises where the reference's asserts would fire."""
        rep = self.validate_self_report()
        if rep["n_fail"] or rep["n_kmers"] != self.n_kmers():
            raise AssertionError(f"validate_self failed: {rep}")

    validate_self_parallel = validate_self  # kphf/mod.rs:105-139 (the device pass is parallel by construction)

    def as_streaming(self):  # sshash.rs:435-437
        return StreamingK2U(self)


class StreamingK2U:  # index/caching.rs:12-17
    def __init__(self, k2u):
        self.k2u = k2u
        h = C.c_void_p()
        check(lib.mz_streaming_new(k2u._h, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.mz_streaming_free(self._h)
            self._h = None

    def k(self): return self.k2u.k()

    def k2u_streaming(self, km):  # index/caching.rs:65-71
        out = _capi.K2UPos()
        check(lib.mz_streaming_k2u(self._h, km.get_fw_mer(), len(km), C.byref(out)))
        return None if out.match == 0 else K2UPos(out.unitig_id, out.unitig_len, out.pos, out.match)
