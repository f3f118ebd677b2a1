"""Host-side mirror of mazu's index facade around the K2U path (reference src/index.rs,
src/index/validate.rs, src/index/caching.rs, src/index/dense_unitig_table.rs): ModIndex.from_parts,
get_ref_pos, validate_self, as_streaming.  The k2u step goes to the GPU through the K2U object
it is given (mazu_amd.SSHash).  Occurrence projection exists twice: a numpy restatement of
project_onto_u_occ (the host mirror the tests compare against) and the device form of mz_u2pos.hip
behind get_ref_pos_batch / validate_self_on_device / validate_fasta_on_device."""
import numpy as np

import ctypes as C

from ._capi import check, lib
from .kphf import CanonicalKmer, MatchType
from . import pf1

FORWARD, BACKWARD = 1, 0  # Orientation::to_u8, src/lib.rs:50-64


class MappedRefPos:  # src/index.rs:26-47
    __slots__ = ("ref_id", "pos", "o")

    def __init__(self, ref_id, pos, o):
        self.ref_id, self.pos, self.o = int(ref_id), int(pos), int(o)

    @classmethod
    def new_fw(cls, ref_id, pos): return cls(ref_id, pos, FORWARD)

    @classmethod
    def new_rc(cls, ref_id, pos): return cls(ref_id, pos, BACKWARD)

    def __eq__(self, other): return (self.ref_id, self.pos, self.o) == (other.ref_id, other.pos, other.o)
    def __repr__(self): return f"MappedRefPos(ref_id={self.ref_id}, pos={self.pos}, o={'fw' if self.o else 'rc'})"


MAPPED_DTYPE = np.dtype([("ref_id", "<u4"), ("pos", "<u4"), ("o", "<u4"), ("query", "<u4")])


class DenseUnitigTable:  # src/index/dense_unitig_table.rs:12-76 (U2Pos)
    def __init__(self, ctable, contig_offsets, ref_names):
        self.ctable = np.ascontiguousarray(ctable, dtype=np.uint64)
        self.contig_offsets = np.ascontiguousarray(contig_offsets, dtype=np.uint64)
        self.ref_names = list(ref_names)
        self._h = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.mz_u2pos_free(self._h)
            self._h = None

    def handle(self):
        """device-backed twin (mz_u2pos) for the projection kernels; created on first use"""
        if self._h is None:
            h = C.c_void_p()
            check(lib.mz_u2pos_dense_new(self.ctable.ctypes.data, len(self.ctable), self.contig_offsets.ctypes.data,
                                         len(self.contig_offsets) - 1, C.byref(h)))
            self._h = h
        return self._h

    def n_unitig_occs(self, ui): return lib.mz_u2pos_n_unitig_occs(self.handle(), ui)
    def num_total_occs(self): return lib.mz_u2pos_num_total_occs(self.handle())

    def encoded_unitig_occs(self, ui):
        return self.ctable[int(self.contig_offsets[ui]): int(self.contig_offsets[ui + 1])]

    def decode_occs(self, idx):
        """(ref_id, pos, forward?) of the occurrences at table indices idx"""
        return self.decode_pf1(self.ctable[idx])

    def decode_unitig_occs(self, ui):  # dense_unitig_table.rs:65-67
        return self.decode_pf1(self.encoded_unitig_occs(ui))

    @staticmethod
    def decode_pf1(words):  # UnitigOcc::decode_pf1, src/index.rs:334-346
        words = np.asarray(words, dtype=np.uint64)
        ref_id = words & np.uint64(0xFFFFFFFF)
        pw = words >> np.uint64(32)
        return ref_id, pw & np.uint64(0x7FFFFFFF), (pw & np.uint64(0x80000000)) != 0


def _pack(values, width):
    """IntVector with `width` bits per element (element i at bit i * width, first element in the low bits) as uint64 words"""
    values = np.asarray(values, dtype=np.uint64)
    n = len(values)
    words = np.zeros((n * width + 63) // 64 + 1, dtype=np.uint64)
    bit = np.arange(n, dtype=np.uint64) * np.uint64(width)
    wi, sh = (bit >> np.uint64(6)).astype(np.int64), bit & np.uint64(63)
    np.bitwise_or.at(words, wi, values << sh)
    spill = (sh + np.uint64(width)) > np.uint64(64)
    np.bitwise_or.at(words, wi[spill] + 1, values[spill] >> (np.uint64(64) - sh[spill]))
    return words[: (n * width + 63) // 64] if n else words[:0]


def _unpack(words, idx, width):
    idx = np.asarray(idx, dtype=np.uint64)
    w = np.concatenate([np.asarray(words, dtype=np.uint64), np.zeros(2, dtype=np.uint64)])
    bit = idx * np.uint64(width)
    wi, sh = (bit >> np.uint64(6)).astype(np.int64), bit & np.uint64(63)
    x = (w[wi] >> sh) | np.where(sh > 0, w[wi + 1] << ((np.uint64(64) - sh) & np.uint64(63)), np.uint64(0))
    return x if width == 64 else x & np.uint64((1 << width) - 1)


def _bit_width(n):
    return max(1, int(n).bit_length())


class PiscemUnitigTable:  # src/index/dense_unitig_table.rs:110-153 (U2Pos over occurrences packed into the minimal number of bits)
    def __init__(self, ctable_words, n_occs, occ_width, ref_shift, pos_mask, contig_offsets, ref_names):
        self.ctable_words = np.ascontiguousarray(ctable_words, dtype=np.uint64)
        self.n_occs, self.occ_width, self.ref_shift, self.pos_mask = int(n_occs), int(occ_width), int(ref_shift), int(pos_mask)
        self.contig_offsets = np.ascontiguousarray(contig_offsets, dtype=np.uint64)  # plain copy for the host mirror
        self.offsets_width = _bit_width(int(self.contig_offsets.max()) if len(self.contig_offsets) else 0)  # IntVector::pack, piscem_index.rs:30-31
        self.contig_offsets_words = _pack(self.contig_offsets, self.offsets_width)
        self.ref_names = list(ref_names)
        self._h = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.mz_u2pos_free(self._h)
            self._h = None

    @classmethod
    def from_occs(cls, ref_id, pos, fwd, contig_offsets, ref_names, max_ref_len):
        """TileOccTable::from_occ_info + set_occ (src/spt_compact.rs:132-160): widths from the longest reference and the
        number of references (required_num_bits, :221-243), words by UnitigOcc::encode_piscem (:90-96)"""
        pos_bits = int(max_ref_len).bit_length()          # ilog2(longest_ref) + 1
        ref_bits = int(len(ref_names)).bit_length()        # ilog2(num_refs) + 1
        width, ref_shift = 1 + pos_bits + ref_bits, pos_bits + 1
        words = (np.asarray(ref_id, dtype=np.uint64) << np.uint64(ref_shift)) | (np.asarray(pos, dtype=np.uint64) << np.uint64(1)) | np.asarray(fwd, dtype=np.uint64)
        return cls(_pack(words, width), len(words), width, ref_shift, (1 << pos_bits) - 1, contig_offsets, ref_names)

    def handle(self):
        if self._h is None:
            h = C.c_void_p()
            check(lib.mz_u2pos_piscem_new(self.ctable_words.ctypes.data, self.n_occs, self.occ_width, self.ref_shift, self.pos_mask,
                                          self.contig_offsets_words.ctypes.data, self.offsets_width, len(self.contig_offsets) - 1, C.byref(h)))
            self._h = h
        return self._h

    def n_unitig_occs(self, ui): return lib.mz_u2pos_n_unitig_occs(self.handle(), ui)
    def num_total_occs(self): return lib.mz_u2pos_num_total_occs(self.handle())

    def encoded_unitig_occs(self, ui):  # OccsRange, dense_unitig_table.rs:130-135
        return range(int(self.contig_offsets[ui]), int(self.contig_offsets[ui + 1]))

    def decode_occs(self, idx):  # UnitigOcc::decode_piscem, src/spt_compact.rs:97-112
        e = _unpack(self.ctable_words, idx, self.occ_width)
        return e >> np.uint64(self.ref_shift), (e >> np.uint64(1)) & np.uint64(self.pos_mask), (e & np.uint64(1)) != 0

    def decode_unitig_occs(self, ui):  # dense_unitig_table.rs:137-144
        r = self.encoded_unitig_occs(ui)
        return self.decode_occs(np.arange(r.start, r.stop, dtype=np.uint64))


class RefSeqCollection:  # src/refseq.rs:17-20
    def __init__(self, seq_words, prefix_sum):
        self.seq_words = seq_words
        self.prefix_sum = np.asarray(prefix_sum, dtype=np.uint64)

    def has_seq(self): return self.seq_words is not None
    def n_refs(self): return len(self.prefix_sum) - 1
    def ref_len(self, r): return int(self.prefix_sum[r + 1] - self.prefix_sum[r])

    def kmer_words(self, r, k):
        """forward words of every k-mer of reference r (RefSeqSlice::iter_kmers)"""
        s, n = int(self.prefix_sum[r]), self.ref_len(r) - k + 1
        if n <= 0:
            return np.zeros(0, dtype=np.uint64)
        pos = np.arange(s, s + n, dtype=np.uint64)
        b = pos * np.uint64(2)
        wi = (b >> np.uint64(6)).astype(np.int64)
        sh = b & np.uint64(63)
        w = np.concatenate([self.seq_words, np.zeros(2, dtype=np.uint64)])
        x = (w[wi] >> sh) | np.where(sh > 0, w[wi + 1] << ((np.uint64(64) - sh) & np.uint64(63)), np.uint64(0))
        return x & np.uint64((1 << (2 * k)) - 1)


def project_onto_u_occs(k, rec, ref_id, cpos, fwd):
    """project_onto_u_occ (src/index.rs:184-216), vectorised over the occurrences of one unitig"""
    upos, ulen = int(rec["pos"]), int(rec["unitig_len"])
    pos = np.where(fwd, upos + cpos.astype(np.int64), cpos.astype(np.int64) + (ulen - upos) - k)
    o = FORWARD if int(rec["match"]) == MatchType.IdentityMatch else BACKWARD
    oo = np.where(fwd, o, 1 - o)
    return [MappedRefPos(r, p, x) for r, p, x in zip(ref_id.tolist(), pos.tolist(), oo.tolist())]


class ModIndex:  # src/index.rs:49-124
    def __init__(self, base, k2u, u2pos, refs):
        self.base, self.k2u, self.u2pos, self.refs = base, k2u, u2pos, refs

    @classmethod
    def from_parts(cls, base, k2u, u2pos, refs):  # src/index.rs:80-87
        return cls(base, k2u, u2pos, refs)

    def k(self): return self.k2u.k()
    def as_k2u(self): return self.k2u
    def as_u2pos(self): return self.u2pos
    def as_refseqs(self): return self.refs
    def as_ref(self): return self.k2u.as_ref()
    def n_unitigs(self): return self.k2u.n_unitigs()
    def n_kmers(self): return self.k2u.n_kmers()

    def get_ref_pos_eager(self, km):  # src/index.rs:139-142,156-179
        if len(km) != self.k():
            raise ValueError(f"Got query k-mer size k={len(km)}, expected k={self.k()}.")  # the reference panics
        recs, _ = self.k2u.k2u_batch(np.array([km.get_fw_mer()], dtype=np.uint64))
        rec = recs[0]
        if int(rec["match"]) == 0:
            return None
        ref_id, cpos, fwd = self.u2pos.decode_unitig_occs(int(rec["unitig_id"]))
        return project_onto_u_occs(self.k(), rec, ref_id, cpos, fwd)

    def _validate_batch(self, fw_words, ref_id, ref_pos, k2u_batch):
        """every queried k-mer must map back to (ref_id, ref_pos): the assertion loop of
        Validate::validate_self / validate_ckmers (src/index/validate.rs:24-81), vectorised"""
        k = self.k()
        recs, _ = k2u_batch(fw_words)
        if (recs["match"] == 0).any():
            i = int(np.nonzero(recs["match"] == 0)[0][0])
            raise AssertionError(f"No MRPs found for true +ve kmer in ref {ref_id} @ pos {int(ref_pos[i])}")
        off = self.u2pos.contig_offsets
        uid = recs["unitig_id"].astype(np.int64)
        s, e = off[uid].astype(np.int64), off[uid + 1].astype(np.int64)
        cnt = e - s
        owner = np.repeat(np.arange(len(recs)), cnt)
        idx = np.repeat(s, cnt) + (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        orid, cpos, fwd = self.u2pos.decode_occs(idx)
        upos, ulen = recs["pos"].astype(np.int64)[owner], recs["unitig_len"].astype(np.int64)[owner]
        pos = np.where(fwd, upos + cpos.astype(np.int64), cpos.astype(np.int64) + (ulen - upos) - k)
        ok = (pos == ref_pos.astype(np.int64)[owner]) & (orid.astype(np.int64) == ref_id)
        found = np.zeros(len(recs), dtype=bool)
        np.logical_or.at(found, owner, ok)
        if not found.all():
            i = int(np.nonzero(~found)[0][0])
            raise AssertionError(f"No matching MRP for true +ve kmer in ref {ref_id} @ pos {int(ref_pos[i])}")
        return len(recs)

    def validate_self(self):  # src/index/validate.rs:24-52
        assert self.refs.has_seq()
        n = 0
        for r in range(self.refs.n_refs()):
            fw = self.refs.kmer_words(r, self.k())
            n += self._validate_batch(fw, r, np.arange(len(fw)), self.k2u.k2u_batch)
        return n

    def as_streaming(self):  # src/index/caching.rs:221-232
        return StreamingIndex(self)

    # ---- device-resident forms: k2u AND the projection on the GPU
    def get_ref_pos_batch(self, fw_words, cap=None):
        """get_ref_pos_eager for a batch: returns (offsets[n+1], MappedRefPos records grouped by query)"""
        fw = np.ascontiguousarray(fw_words, dtype=np.uint64)
        offs = np.zeros(len(fw) + 1, dtype=np.uint64)
        cap = int(cap if cap is not None else max(16, 4 * len(fw)))
        while True:
            out = np.zeros(cap, dtype=MAPPED_DTYPE)
            n_out = C.c_uint64(0)
            check(lib.mz_get_ref_pos_batch(self.k2u._h, self.u2pos.handle(), fw.ctypes.data, len(fw), offs.ctypes.data,
                                           out.ctypes.data, cap, C.byref(n_out)))
            if n_out.value <= cap:
                return offs, out[: n_out.value]
            cap = int(n_out.value)

    def validate_self_on_device(self):
        """Validate::validate_self (src/index/validate.rs:24-52) with lookup, projection and the assertion all
        on the device; returns the number of k-mers checked, raises where the reference would panic"""
        import torch
        assert self.refs.has_seq()
        dev = torch.device("cuda", torch.cuda.current_device())  # the caller's GPU (one process per GPU: its LOCAL_RANK)
        report = torch.tensor([0, 0, -1], dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        for r in range(self.refs.n_refs()):
            fw = self.refs.kmer_words(r, self.k())
            if len(fw) == 0:
                continue
            dq = torch.from_numpy(fw.view(np.int64)).to(dev)
            drec = torch.empty(2 * len(fw), dtype=torch.int64, device=dev)
            self.k2u.k2u_batch_device(dq, drec, None, stream)
            check(lib.mz_u2pos_check_refs_device(self.u2pos.handle(), self.k(), drec.data_ptr(), len(fw), r, report.data_ptr(), stream))
        torch.cuda.synchronize(dev)
        n, n_fail, first = report.cpu().tolist()
        if n_fail:
            raise AssertionError(f"No matching MRP for {n_fail} true +ve k-mers (first at position {first})")
        return n


class StreamingIndex:  # src/index/caching.rs:37-41,141-219
    def __init__(self, index):
        self.index = index
        self.k2u = index.k2u

    def k(self): return self.index.k()

    def validate_ckmers(self, ref_id, seq):  # caching.rs:175-200: CanonicalKmerIterator skips non-ACGT windows
        bases = np.frombuffer(seq.encode() if isinstance(seq, str) else seq, dtype=np.uint8)
        ro = np.array([0, len(bases)], dtype=np.uint64)
        recs, _ = self.k2u.k2u_reads(bases, ro, streaming=True)
        keep = recs["match"] != MatchType.Skipped
        pos = np.nonzero(keep)[0]
        idx = self.index

        def batch(_fw):
            return recs[keep], None
        return idx._validate_batch(np.zeros(int(keep.sum()), dtype=np.uint64), ref_id, pos, batch)

    def validate_fasta(self, path):  # caching.rs:202-218
        from .fasta import read_fasta
        return sum(self.validate_ckmers(ri, seq) for ri, (_, seq) in enumerate(read_fasta(path)))


def dense_index_from_pf1(d, k2u_factory):
    """DenseIndex::deserialize_from_cpp (src/pf1/dense_index.rs:33-97) with the K2U built by
    `k2u_factory(useq_words, total_len, starts, k)` -- the hook where a GPU SSHash is dropped in."""
    f = pf1.DenseIndexFiles(d)
    k2u = k2u_factory(f.useq_words, f.total_len, f.starts, f.k)
    u2pos = DenseUnitigTable(f.ctable, f.contig_offsets, f.ref_names)
    refs = RefSeqCollection(f.refseq_words, f.ref_prefix_sum)
    return ModIndex.from_parts({"index_type": "PF1Dense", "info": f.info}, k2u, u2pos, refs), f


class SPT:
    """Spectrum-preserving tiling with pufferfish-packed occurrence lists (reference src/spt.rs:14-152):
    cuttlefish reduced GFA (`<prefix>.cf_seg` unitigs, `<prefix>.cf_seq` tilings `name\tN3 12+ 7- ...`) ->
    inverted lists of unitig occurrences.  Host-side ingestion (numpy); the lookups run on the GPU."""

    def __init__(self, k, seqs, ref_names, ctable, offsets, ref_lens):
        self.k_, self.seqs, self.ref_names = k, seqs, ref_names
        self.ctable, self.offsets, self.ref_lens = ctable, offsets, ref_lens

    def k(self): return self.k_
    def n_unitigs(self): return len(self.seqs)
    def num_refs(self): return len(self.ref_lens)
    def n_total_occs(self): return len(self.ctable)
    def ref_name(self, ri): return self.ref_names[ri]
    def ref_len(self, ri): return int(self.ref_lens[ri])

    def get_occ(self, uid, i):  # spt.rs:56-60 -> (ref_id, pos, forward?)
        ref_id, pos, fwd = DenseUnitigTable.decode_pf1(self.ctable[int(self.offsets[uid]) + i: int(self.offsets[uid]) + i + 1])
        return int(ref_id[0]), int(pos[0]), bool(fwd[0])

    @classmethod
    def from_cf_prefix(cls, prefix):  # spt.rs:62-146
        import json
        with open(prefix + ".json") as f:
            k = json.load(f)["parameters info"]["k"]
        cfid2uid, seqs = {}, []
        with open(prefix + ".cf_seg") as f:
            for line in f:
                line = line.rstrip("\n")
                if line:
                    cid, seq = line.split("\t", 1)
                    cfid2uid[int(cid)] = len(seqs)   # unitig_set.rs:133-141: sequential ids in line order
                    seqs.append(seq)
        ulen = np.array([len(s) for s in seqs], dtype=np.int64)
        ref_names, occ_uid, occ_word, ref_lens = [], [], [], []
        with open(prefix + ".cf_seq") as f:
            for ref_id, line in enumerate(f):
                name, toks = line.rstrip("\n").split("\t", 1)
                ref_names.append(name)
                pos, prev_unitig = 0, False
                for tok in toks.split(" "):
                    if not tok:
                        continue
                    if tok[0] == "N":            # CfSeqToken::PolyN, spt.rs:104-110
                        pos += int(tok[1:])
                        if prev_unitig:
                            pos += k - 1
                        prev_unitig = False
                    else:                        # CfSeqToken::Unitig, spt.rs:111-122
                        uid = cfid2uid[int(tok[:-1])]
                        fwd = tok[-1] == "+"
                        word = ((pos | (0x80000000 if fwd else 0)) << 32) | ref_id   # UnitigOcc::encode_pf1, index.rs:315-331
                        occ_uid.append(uid)
                        occ_word.append(word)
                        pos += int(ulen[uid]) - k + 1
                        prev_unitig = True
                ref_lens.append(pos + k - 1 if prev_unitig else pos)
        occ_uid = np.array(occ_uid, dtype=np.int64)
        occ_word = np.array(occ_word, dtype=np.uint64)
        order = np.argsort(occ_uid, kind="stable")       # inverted lists keep tiling order inside a unitig (ctable_ptrs, spt.rs:118-120)
        offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(np.bincount(occ_uid, minlength=len(seqs)), out=offsets[1:].view(np.int64))
        return cls(k, seqs, ref_names, occ_word[order], offsets, np.array(ref_lens, dtype=np.uint64))

    def get_ref_seq_collection(self):  # spt.rs:148-153: lengths only, no sequence
        ps = np.zeros(len(self.ref_lens) + 1, dtype=np.uint64)
        np.cumsum(self.ref_lens, out=ps[1:])
        return RefSeqCollection(None, ps)


def index_from_cf_prefix(prefix, k2u_factory, piscem=False):
    """PufferfishDenseIndexDefault::from_cf_prefix (src/index/defaults.rs:16-54) with the K2U slot filled by
    `k2u_factory(seqs, k)` -- e.g. a GPU SSHash; piscem=True gives PiscemIndex::from_cf_prefix instead
    (src/index/piscem_index.rs:17-56): same tiling, occurrences packed into the minimal number of bits
    (SPTCompact, src/spt_compact.rs:287-389) behind a PiscemUnitigTable."""
    spt = SPT.from_cf_prefix(prefix)
    k2u = k2u_factory(spt.seqs, spt.k())
    if piscem:
        ref_id, pos, fwd = DenseUnitigTable.decode_pf1(spt.ctable)
        u2pos = PiscemUnitigTable.from_occs(ref_id, pos, fwd, spt.offsets, spt.ref_names, int(max(spt.ref_lens)) if len(spt.ref_lens) else 0)
        base = {"index_type": "Piscem"}
    else:
        u2pos, base = DenseUnitigTable(spt.ctable, spt.offsets, spt.ref_names), {"index_type": "PufferfishDense"}
    return ModIndex.from_parts(base, k2u, u2pos, spt.get_ref_seq_collection()), spt


def validate_fasta_on_device(index, path):
    """Validate::validate_fasta (src/index/validate.rs:83-100) with the reads kernel, the projection and the
    assertion on the device; returns the number of k-mers checked."""
    import torch
    from .fasta import read_fasta
    dev = torch.device("cuda", torch.cuda.current_device())  # the caller's GPU
    report = torch.tensor([0, 0, -1], dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = index.k()
    n_valid = 0
    for ri, (_, seq) in enumerate(read_fasta(path)):
        if len(seq) < k:
            continue
        bases = torch.frombuffer(bytearray(seq.encode()), dtype=torch.uint8).to(dev)
        ro = torch.tensor([0, len(seq)], dtype=torch.int64, device=dev)
        ko = torch.tensor([0, len(seq) - k + 1], dtype=torch.int64, device=dev)
        recs = torch.empty(2 * (len(seq) - k + 1), dtype=torch.int64, device=dev)
        cnt = torch.zeros(3, dtype=torch.int64, device=dev)
        index.k2u.k2u_reads_device(bases, ro, ko, recs, cnt, streaming=True, stream=stream)
        check(lib.mz_u2pos_check_refs_device(index.u2pos.handle(), k, recs.data_ptr(), len(seq) - k + 1, ri, report.data_ptr(), stream))
        torch.cuda.synchronize(dev)
        n_valid += int(cnt[0].item() + cnt[1].item())
    n, n_fail, first = report.cpu().tolist()
    if n_fail:
        raise AssertionError(f"No MRPs found for {n_fail} true +ve k-mers (first at position {first})")
    return n_valid
