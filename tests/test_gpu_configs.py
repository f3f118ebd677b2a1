"""GPU tests (-m gpu) at the sizes BASELINE.json's configurations name, plus the failure side of the validator.

  * s-eighth: the headline kernel combination -- k2u_wave_kernel<31,15> x direct minimizer records x negative filter x
    skew table on a GPU-built index -- byte-identical to the oracle on a 50/50 NEG-U mix and on NEG-1 negatives
  * s-txome reads through mz_sshash_k2u_reads_device (.as_streaming() request) == the oracle's StreamingK2U walk
  * s-human and s-pan at full size: K2U::validate_self on the device, every sampled positive equal to the answer
    computed from the plain unitig offsets (an independent ground truth that needs no index), negatives miss,
    reverse-complement symmetry
  * a corrupted index must FAIL validate_self, at the position that was corrupted
"""
import os
import struct

import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from oracle import oracle as O

pytestmark = pytest.mark.gpu

CF_YEAST = os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7")


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    assert M.lib.mz_device_count() >= 1, "GPU tests need a visible HIP device (no CPU fallback exists)"


def _log(msg):
    print(msg, flush=True)  # long steps at full size keep the log moving


def _host_arrays(us):
    return us["words"].cpu().numpy().view(np.uint64), us["starts"].cpu().numpy().view(np.uint64)


def _k2u_device(ix, q):
    import torch
    out = torch.zeros(q.numel() * 2, dtype=torch.int64, device=q.device)
    cnt = torch.zeros(3, dtype=torch.int64, device=q.device)
    ix.k2u_batch_device(q, out, cnt, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out, cnt.cpu().tolist()


@pytest.mark.parametrize("front", [1, -1])
def test_s_eighth_headline_kernel_matches_oracle(front):
    """BASELINE configs[3] at 1/8 scale (452 Mbp, 4.5 M unitigs): same kernel specialisation and index form as the headline.
    front=1: minimizer lines (the form the full-size sets get by the size rule, forced here); front=0: the minimizer
    record -> entry array path with the records' negative filter (what spilled buckets and small sets use)"""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-eighth", device="cuda")
    assert us["total_len"] == 3_589_802_592 // 8
    ix, rep, _ = synth.build_index_unique(us, 15, 32, on_device=True, log=_log, minimizer_lines=front)  # mz_build_opts: lines forced / never built
    assert ix.minimizer_mphf_kind() == 1 and ix.n_kmers_in_skew_index() > 0  # direct minimizer records + a populated skew table
    assert ix.has_minimizer_lines() == (front == 1)
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and rep["n_kmers"] == ix.n_kmers()
    q = synth.make_query_mix(ix, us, 2_000_000, seed=11, device="cuda")                         # 50/50 NEG-U
    q1 = synth.make_query_mix(ix, us, 400_000, seed=12, device="cuda", neg_kind="sub1")        # 200 k positives + 200 k NEG-1
    qq = torch.cat([q, q1])
    out, (n_hit, n_miss, _) = _k2u_device(ix, qq)
    words, starts = _host_arrays(us)
    orc = O.OracleSSHash(words, us["total_len"], starts, 31, 15, 32, with_dict=False)
    want, o_hit = orc.k2u_batch(qq.cpu().numpy().view(np.uint64), n_threads=M.lib.mz_host_threads())
    got = out.cpu().numpy().view(M.K2UPOS_DTYPE)
    assert got.tobytes() == want.tobytes()
    assert (n_hit, n_miss) == (o_hit, qq.numel() - o_hit)
    assert abs(n_hit - 1_200_000) <= 8  # positives hit; NEG-U / NEG-1 negatives miss (up to accidental members)


def test_s_eighth_rep_realistic_skew_matches_oracle():
    """BASELINE configs[3] at 1/8 scale with the repeat model its description asks for: 8 % of the sequence are copies of 300-base
    repeat families at 5-15 % per-copy divergence (synth._plant_repeat_families), so bucket sizes form a continuum from one to
    hundreds of entries instead of the two modes of the planted 16-mers.  Layout by the size rules; 2 M NEG-U + 200 k NEG-1
    queries byte-identical to the oracle; the line statistics must show the continuum (inline, spilled AND heavy mass)."""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-eighth-rep", device="cuda")
    assert us["total_len"] == 3_589_802_592 // 8 and us["rep_substitutions"] > 100_000  # copies shared 31-mers before the de-duplication
    ix, rep, _ = synth.build_index_unique(us, 15, 32, on_device=True, log=_log)
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and rep["n_kmers"] == ix.n_kmers()
    assert ix.minimizer_mphf_kind() == 1 and ix.has_minimizer_lines() and ix.n_kmers_in_skew_index() > 0
    st = ix.line_stats()
    _log(f"s-eighth-rep lines: {st}")
    assert st["entries_inline"] + st["entries_spilled"] + st["entries_heavy"] == st["entries_total"] == ix.n_minimizer_occs()
    assert st["frac_inline"] > 0.9 and st["buckets_spilled"] > 1000 and st["buckets_heavy"] > 1000
    q = synth.make_query_mix(ix, us, 2_000_000, seed=11, device="cuda")                         # 50/50 NEG-U
    q1 = synth.make_query_mix(ix, us, 400_000, seed=12, device="cuda", neg_kind="sub1")        # 200 k positives + 200 k NEG-1
    qq = torch.cat([q, q1])
    out, (n_hit, n_miss, _) = _k2u_device(ix, qq)
    words, starts = _host_arrays(us)
    orc = O.OracleSSHash(words, us["total_len"], starts, 31, 15, 32, with_dict=False)
    want, o_hit = orc.k2u_batch(qq.cpu().numpy().view(np.uint64), n_threads=M.lib.mz_host_threads())
    got = out.cpu().numpy().view(M.K2UPOS_DTYPE)
    assert got.tobytes() == want.tobytes()
    # (a k-mer of a repeat copy with one base substituted is, now and then, the k-mer of another copy: a few NEG-1 queries hit, here and in the oracle)
    assert (n_hit, n_miss) == (o_hit, qq.numel() - o_hit) and 1_200_000 <= n_hit <= 1_202_000
    # reads over the repeats: windows inside copies meet the large buckets
    bases, ro = synth.make_reads(us, 20_000, 150, seed=2000, device="cuda")
    ko = synth.kmer_offsets(ro, 31)
    want_r, _, oc = orc.k2u_reads(bases.cpu().numpy(), ro.cpu().numpy().astype(np.uint64), streaming=True)
    for verify in ("each", "group"):
        ix.set_query_opts(reads_verify=verify)
        out_r = torch.zeros(int(ko[-1]) * 2, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
        ix.k2u_reads_device(bases, ro, ko, out_r, cnt, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert out_r.cpu().numpy().view(M.K2UPOS_DTYPE).tobytes() == want_r.tobytes() and cnt.cpu().tolist() == oc.tolist(), verify


@pytest.mark.parametrize("bytes_per_key,kind,reads_lines,resolve,verify", [(48, 1, None, None, None), (48, 1, "lines", "run", None), (48, 1, "lines", "item", None),
                                                                            (16, 0, None, None, None), (48, 1, None, None, "group"), (48, 1, "lines", "item", "group")])
def test_s_txome_reads_match_oracle_streaming_walk(bytes_per_key, kind, reads_lines, resolve, verify):
    """BASELINE configs[2] shape: 150-bp reads over a 145 Mbp, 1.2 M-unitig set, requested as .as_streaming();
    every window must equal the oracle's StreamingK2U walk (index/caching.rs:65-103, bin/kphf/main.rs:299-312).
    Once with the direct-address minimizer records the size rule picks for this set (the reads kernel then resolves one
    minimizer per run of windows through the records, although the index also has minimizer lines), once through the
    minimizer lines (what the kernel does on large indexes), once with the BBHash levels the rule picks for smaller sets; and
    with the grouped verify-and-extend finish instead of one sequence-record gather per candidate.  All choices go through
    mz_build_opts / mz_query_opts."""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-txome", device="cuda")
    ix, rep, _ = synth.build_index_unique(us, 15, 32, on_device=True, log=_log, direct_bytes_per_key=bytes_per_key)
    # through the lines: one lookup per run of windows, or every window scanning its line (the rule on large indexes)
    ix.set_query_opts(reads_lines=reads_lines, reads_resolve=resolve, reads_verify=verify)
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and ix.minimizer_mphf_kind() == kind
    assert ix.has_minimizer_lines() == (kind == 1)
    n_reads, read_len = 20_000, 150
    bases, ro = synth.make_reads(us, n_reads, read_len, seed=2000, device="cuda")
    ko = synth.kmer_offsets(ro, 31)
    out = torch.zeros(int(ko[-1]) * 2, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
    ix.k2u_reads_device(bases, ro, ko, out, cnt, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    words, starts = _host_arrays(us)
    orc = O.OracleSSHash(words, us["total_len"], starts, 31, 15, 32, with_dict=False)
    want, ko_h, oc = orc.k2u_reads(bases.cpu().numpy(), ro.cpu().numpy().astype(np.uint64), streaming=True)
    cold, _, _ = orc.k2u_reads(bases.cpu().numpy(), ro.cpu().numpy().astype(np.uint64), streaming=False)
    assert want.tobytes() == cold.tobytes()  # unique k-mers: the streaming walk equals independent lookups
    got = out.cpu().numpy().view(M.K2UPOS_DTYPE)
    assert got.tobytes() == want.tobytes() and cnt.cpu().tolist() == oc.tolist()
    assert 0.3 < oc[0] / (n_reads * 120) < 0.99  # junction windows and substituted reads miss, the rest hit


@pytest.mark.parametrize("w,reads_kernel", [(17, None), (19, None), (21, None), (19, "block"), (20, None)])
def test_s_txome_reads_with_minimizers_beyond_32_bits_match_oracle(w, reads_kernel):
    """150-bp reads at w = 17 / 19 / 21: the wave-scoped reads kernel with its 20-bit window keys (a window whose two smallest w-mer hashes
    agree in those 20 bits is flagged and takes the exact walk: ~200 of the 2.4 M windows here) and every window scanning its hash-addressed
    minimizer line; w = 20 and reads_kernel = block take the workgroup-staged kernel through the BBHash levels.  Every window must equal
    the oracle's streaming walk (index/caching.rs:65-103), on sampled reads and on the crafted ones (reverse strand, unitig hops, N, tile edges)."""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-txome", device="cuda")
    ix, rep, _ = synth.build_index_unique(us, w, 32, on_device=True, log=_log)
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and ix.minimizer_mphf_kind() == 0 and ix.has_minimizer_lines() and ix.built_on_device()
    ix.set_query_opts(reads_kernel=reads_kernel)
    words, starts = _host_arrays(us)
    orc = O.OracleSSHash(words, us["total_len"], starts, 31, w, 32, with_dict=False)
    bases, ro = synth.make_reads(us, 20_000, 150, seed=2000 + w, device="cuda")
    cb, cro = _crafted_reads(us, np.random.default_rng(w))
    for b, r in ((bases, ro), (torch.from_numpy(cb).cuda(), torch.from_numpy(cro.astype(np.int64)).cuda())):
        ko = synth.kmer_offsets(r, 31)
        out = torch.zeros(int(ko[-1]) * 2, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
        ix.k2u_reads_device(b, r, ko, out, cnt, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want, _, oc = orc.k2u_reads(b.cpu().numpy(), r.cpu().numpy().astype(np.uint64), streaming=True)
        assert out.cpu().numpy().view(M.K2UPOS_DTYPE).tobytes() == want.tobytes() and cnt.cpu().tolist() == oc.tolist()
        assert oc[0] > 0.3 * int(ko[-1])


def _crafted_reads(us, rng):
    """reads that exercise the verify-and-extend finish of the reads kernel: whole reads on the reverse strand, reads that hop from one
    unitig to a distant one (in both strands), non-ACGT bytes, and lengths that put read ends on every side of the 256-position
    tile edges; returns (bases uint8, read_offsets uint64)"""
    words = us["words"].cpu().numpy().view(np.uint64)
    starts = us["starts"].cpu().numpy().astype(np.int64)
    L = us["total_len"]

    def seq(a, n):  # codes of bases [a, a + n) of the unitig concatenation
        i = np.arange(a, a + n, dtype=np.int64)
        return ((words[i >> 5] >> ((i & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)

    def rc(c):
        return (3 - c)[::-1]

    long_u = np.nonzero(starts[1:] - starts[:-1] >= 200)[0]
    reads = []
    for j in range(300):  # single-unitig reads, forward and reverse, lengths around the tile edges
        u = int(long_u[rng.integers(len(long_u))])
        n = int(rng.choice([31, 32, 63, 64, 65, 127, 128, 129, 150, 199, 200]))
        a = int(starts[u] + rng.integers(0, starts[u + 1] - starts[u] - n + 1))
        c = seq(a, n)
        reads.append(rc(c) if j % 2 else c)
    for j in range(300):  # hops: the tail of one unitig, then the head of another one far away; half of them reverse-complemented
        u, v = (int(long_u[rng.integers(len(long_u))]) for _ in range(2))
        na, nb = int(rng.integers(31, 100)), int(rng.integers(31, 100))
        c = np.concatenate([seq(int(starts[u + 1]) - na, na), seq(int(starts[v]), nb)])
        reads.append(rc(c) if j % 2 else c)
    for j in range(100):  # reads across the junctions of the concatenation itself (the windows on a junction miss)
        u = int(rng.integers(1, len(starts) - 2))
        a = max(0, int(starts[u]) - int(rng.integers(20, 120)))
        reads.append(seq(a, min(150, L - a)))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = [letters[c] for c in reads]
    for j in range(0, len(out), 7):  # non-ACGT bytes: windows over them are skipped, the stretches on either side continue
        r = out[j].copy()
        r[int(rng.integers(len(r)))] = ord("N")
        if len(r) > 100:
            r[int(rng.integers(len(r)))] = ord("n")
        out[j] = r
    out += [np.frombuffer(b"ACGTACGTAC", dtype=np.uint8), np.zeros(0, dtype=np.uint8), letters[seq(int(starts[5]), 30)]]  # too short for a window, empty
    # lengths 256 / 512 so that reads end exactly on tile edges wherever they start
    out += [letters[seq(int(rng.integers(0, L - 600)), n)] for n in (256, 512, 256, 255, 257, 1, 256)]
    order = rng.permutation(len(out))
    out = [out[i] for i in order]
    ro = np.zeros(len(out) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in out], out=ro[1:].view(np.int64))
    return np.concatenate(out), ro


@pytest.mark.parametrize("reads_lines", ["records", "lines"])
def test_s_txome_crafted_reads_every_finish(reads_lines):
    """index/caching.rs:65-103 walks forward along one unitig; the device finish verifies stretches in either direction and across
    unitig hops.  Crafted reads -- reverse strand, unitig-to-unitig hops, N breaks, ends on tile edges, empty and short reads --
    through every combination of resolve and verify forms must equal the oracle's streaming walk record for record."""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-txome", device="cuda")
    ix, rep, _ = synth.build_index_unique(us, 15, 32, on_device=True, log=_log)
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0
    words, starts = _host_arrays(us)
    orc = O.OracleSSHash(words, us["total_len"], starts, 31, 15, 32, with_dict=False)
    for seed in (1, 2):
        bases, ro = _crafted_reads(us, np.random.default_rng(seed))
        want, ko_h, oc = orc.k2u_reads(bases, ro, streaming=True)
        assert oc[2] > 0 and 0.3 < oc[0] / max(1, len(want)) < 0.99  # skipped windows exist; hits and misses both
        d_b, d_ro, d_ko = (torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x).cuda() for x in (bases, ro, ko_h))
        for resolve in ("item", "run"):
            for verify in ("each", "group"):
                ix.set_query_opts(reads_lines=reads_lines, reads_resolve=resolve, reads_verify=verify)
                out = torch.zeros(len(want) * 2, dtype=torch.int64, device="cuda")
                cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
                ix.k2u_reads_device(d_b, d_ro, d_ko, out, cnt, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                got = out.cpu().numpy().view(M.K2UPOS_DTYPE)
                assert got.tobytes() == want.tobytes() and cnt.cpu().tolist() == oc.tolist(), (seed, resolve, verify)


def _full_size_properties(preset, n_q):
    """validate_self on the device + sampled positives against the answer derived from the plain offsets"""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set(preset, device="cuda")
    assert us["total_len"] == synth.PRESETS[preset]["total_len"] and us["n_unitigs"] == synth.PRESETS[preset]["n_unitigs"]
    ix, rep, _ = synth.build_index_unique(us, 15, 32, on_device=True, log=_log)
    assert ix.minimizer_mphf_kind() == 1 and ix.has_minimizer_lines()
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and rep["n_kmers"] == ix.n_kmers()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    starts = us["starts"]
    pos = synth.sample_positive_positions(starts, 31, n_q, g)
    flip = torch.randint(0, 2, (n_q,), generator=g, device=dev, dtype=torch.uint8)
    q = torch.empty(n_q, dtype=torch.int64, device=dev)
    ix.extract_kmers_device(pos, flip, q)
    torch.cuda.synchronize()
    out, (n_hit, n_miss, _) = _k2u_device(ix, q)
    rec = out.view(torch.int32).view(-1, 4).to(torch.int64) & 0xFFFFFFFF  # unitig_id, pos, unitig_len, match
    uid = torch.searchsorted(starts, pos, right=True) - 1
    assert n_hit == n_q and n_miss == 0
    assert torch.equal(rec[:, 0], uid)
    assert torch.equal(rec[:, 1], pos - starts[uid])
    assert torch.equal(rec[:, 2], starts[uid + 1] - starts[uid])
    assert torch.equal(rec[:, 3], 1 + flip.to(torch.int64))  # forward word -> Identity, reverse complement -> Twin
    # NEG-U negatives miss (accidental members: ~n_kmers / 4^31 each), and k2u(rc(q)) mirrors k2u(q)
    neg = torch.randint(0, 2 ** 62, (n_q,), generator=g, device=dev, dtype=torch.int64)
    _, (h2, m2, _) = _k2u_device(ix, neg)
    assert h2 <= 2 and h2 + m2 == n_q
    q_rc = torch.empty_like(q)
    ix.extract_kmers_device(pos, 1 - flip, q_rc)
    torch.cuda.synchronize()
    out_rc, _ = _k2u_device(ix, q_rc)
    rec_rc = out_rc.view(torch.int32).view(-1, 4).to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(rec_rc[:, :3], rec[:, :3]) and torch.equal(rec_rc[:, 3], 3 - rec[:, 3])
    # reads (at this size every window scans its minimizer line itself): each window must equal the independent lookup of
    # its k-mer -- the definition of the reads configuration -- through both forms of the reads kernel
    n_reads, read_len = 100_000, 150
    bases, ro = synth.make_reads(us, n_reads, read_len, seed=7, device=dev)
    ko = synth.kmer_offsets(ro, 31)
    n_win = int(ko[-1])
    assert n_win == n_reads * (read_len - 30)
    lut = torch.full((256,), -1, dtype=torch.int64, device=dev)
    for c, v in zip(b"ACGT", range(4)):
        lut[c] = v
    code = lut[bases.to(torch.int64)].view(n_reads, read_len)
    assert int(code.min()) >= 0  # the generator draws ACGT only
    fw = torch.zeros((n_reads, read_len - 30), dtype=torch.int64, device=dev)
    for i in range(31):
        fw |= code[:, i:i + read_len - 30] << (2 * i)
    want, (wh, wm, _) = _k2u_device(ix, fw.reshape(-1).contiguous())
    for resolve, verify in (("item", "each"), ("run", "each"), ("item", "group"), ("run", "group")):
        ix.set_query_opts(reads_resolve=resolve, reads_verify=verify)
        got = torch.empty(2 * n_win, dtype=torch.int64, device=dev)
        cnt = torch.zeros(3, dtype=torch.int64, device=dev)
        ix.k2u_reads_device(bases, ro, ko, got, cnt, streaming=True)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (resolve, verify)
        assert cnt.cpu().tolist() == [wh, wm, 0] and 0.3 < wh / n_win < 0.99
    ix.set_query_opts()
    return ix


def test_s_human_full_size_properties():
    """BASELINE configs[3] on one GPU's replica: 36 145 130 unitigs, 3 589 802 592 bp (reference src/cuttlefish.rs:25-44)"""
    _full_size_properties("s-human", 4_000_000)


@pytest.mark.parametrize("w", [15, 19])
def test_s_human_reads_match_oracle_streaming_walk(w):
    """100 000 reads (12 M windows) on the full-size GRCh38-shaped set against the ORACLE's StreamingK2U walk (its own index,
    built here from the same unitig set), through both verify forms: no self-comparison at the headline size.  w = 19: the index the
    GPU builder makes for minimizers beyond 32 bits (BBHash levels + hash-addressed lines), the reads kernel with 20-bit window keys (about
    a thousand of the 12 M windows are undecided by them and take the exact walk), and 2.4 M point queries against the oracle as well."""
    import torch
    from mazu_amd import synth
    us = synth.make_unitig_set("s-human", device="cuda")
    ix, rep, _ = synth.build_index_unique(us, w, 32, on_device=True, log=_log)
    assert rep["n_fail"] == 0 and rep["n_duplicate"] == 0 and ix.built_on_device() and ix.has_minimizer_lines()
    n_reads, read_len = 100_000, 150
    bases, ro = synth.make_reads(us, n_reads, read_len, seed=2000, device="cuda")
    ko = synth.kmer_offsets(ro, 31)
    n_win = int(ko[-1])
    got = {}
    point = None
    if w > 16:
        qq = torch.cat([synth.make_query_mix(ix, us, 2_000_000, seed=11, device="cuda"), synth.make_query_mix(ix, us, 400_000, seed=12, device="cuda", neg_kind="sub1")])
        out_p, cnt_p = _k2u_device(ix, qq)
        point = (qq.cpu().numpy().view(np.uint64), out_p.cpu().numpy().view(M.K2UPOS_DTYPE), cnt_p)
        del qq, out_p
    for verify in (("each", "group") if w == 15 else ("each",)):
        ix.set_query_opts(reads_verify=verify)
        out = torch.zeros(n_win * 2, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
        ix.k2u_reads_device(bases, ro, ko, out, cnt, streaming=True, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got[verify] = (out.cpu().numpy().view(M.K2UPOS_DTYPE), cnt.cpu().tolist())
    bases_h, ro_h = bases.cpu().numpy(), ro.cpu().numpy().astype(np.uint64)
    del ix, out, bases, ro, ko
    torch.cuda.empty_cache()
    words, starts = _host_arrays(us)
    total = us["total_len"]
    del us
    os.environ["OMP_NUM_THREADS"] = str(M.lib.mz_host_threads())
    _log("building the oracle's index of s-human on the host ...")
    orc = O.OracleSSHash(words, total, starts, 31, w, 32, with_dict=False)
    _log("oracle streaming walk over 12 M windows ...")
    want, _, oc = orc.k2u_reads(bases_h, ro_h, streaming=True)
    for verify, (g, c) in got.items():
        assert g.tobytes() == want.tobytes() and c == oc.tolist(), verify
    assert 0.3 < oc[0] / n_win < 0.99
    if point is not None:
        q_h, got_p, (n_hit, n_miss, _) = point
        want_p, o_hit = orc.k2u_batch(q_h, n_threads=M.lib.mz_host_threads())
        assert got_p.tobytes() == want_p.tobytes() and (n_hit, n_miss) == (o_hit, len(q_h) - o_hit)


def test_s_pan_full_size_properties():
    """BASELINE configs[4]: 120 M unitigs, 9 Gbp, one GPU's replica (positions beyond 2^33, ids beyond 2^26)"""
    ix = _full_size_properties("s-pan", 4_000_000)
    assert ix.sum_unitigs_len() == 9_000_000_000 and ix.n_unitigs() == 120_000_000


# ---------------------------------------------------------------- the validator must be able to fail
class FlatIndexFile:
    """offsets of the arrays inside the flat index file written by mz_sshash_save (length-prefixed little-endian vectors)"""

    def __init__(self, data):
        self.d = bytearray(data)
        assert bytes(self.d[:8]) == b"MZSSH013"
        o = 8
        self.k, self.w = struct.unpack_from("<II", self.d, o); o += 8
        self.seed, self.skew_param, self.total_len, self.n_unitigs, self.n_minimizers = struct.unpack_from("<5Q", self.d, o); o += 40
        (self.has_skew,) = struct.unpack_from("<I", self.d, o); o += 4
        self.vec = {}

        def vec(name, o, item=8):
            (n,) = struct.unpack_from("<Q", self.d, o)
            self.vec[name] = (o + 8, n)
            return o + 8 + n * item
        o = vec("useq", o); o = vec("starts", o); o = vec("bv_blocks", o)
        o += 8 * 3 + 4  # accum_lens: u, n, n_high_bits, l
        o = vec("ef_high", o); o = vec("ef_sel", o); o = vec("ef_low", o)
        o += 8 + 4      # minimizer mphf: n_keys, direct
        o = vec("mm_blocks", o); o = vec("mm_levels", o, 24)
        o += 16         # bucket offsets: n, n_overflow
        o = vec("bef_blocks", o); o = vec("bef_ovf", o)
        o += 8 + 4      # pos: n, width
        o = vec("pos", o)
        if self.has_skew:
            o += 8 + 8 + 4  # n_buckets, n_keys, fp_bits
            o = vec("skew_buckets", o); o = vec("skew_ovf_word", o); o = vec("skew_ovf_pos", o)
        (self.has_front,) = struct.unpack_from("<I", self.d, o); o += 4
        if self.has_front:
            (self.front_g, self.front_t) = struct.unpack_from("<II", self.d, o); o += 8
            o = vec("front_lines", o)
        assert o == len(self.d), (o, len(self.d))

    def get(self, name, i):
        off, n = self.vec[name]
        assert i < n
        return struct.unpack_from("<Q", self.d, off + 8 * i)[0]

    def put(self, name, i, v):
        off, n = self.vec[name]
        assert i < n
        struct.pack_into("<Q", self.d, off + 8 * i, v)

    def n(self, name):
        return self.vec[name][1]


@pytest.fixture(scope="module")
def yeast_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("flat")
    us = M.UnitigSet.from_cf_prefix(CF_YEAST)
    no_skew = M.SSHash.from_unitig_set_no_skew_index(us, 15)
    skew = M.SSHash.from_unitig_set(us, 15, 8)
    no_skew.validate_self()
    skew.validate_self()
    no_skew.save(d / "noskew.mz")
    skew.save(d / "skew.mz")
    return d


def _load_corrupted(f, path):
    with open(path, "wb") as fh:
        fh.write(bytes(f.d))
    return M.SSHash.load(path)


def test_validate_self_fails_on_a_corrupted_pos_entry(yeast_files):
    f = FlatIndexFile(open(yeast_files / "noskew.mz", "rb").read())
    k, w = f.k, f.w
    e = f.n("pos") // 2
    ent = f.get("pos", e)
    mm_pos = ent >> 17
    f.put("pos", e, ((mm_pos + 5) << 17) | (ent & 0x1FFFF))  # same flanks and strand class, wrong position
    ix = _load_corrupted(f, yeast_files / "bad_pos.mz")
    rep = ix.validate_self_report()
    assert 2 <= rep["n_fail"] <= 2 * (k - w + 1) and rep["n_kmers"] == ix.n_kmers()
    assert mm_pos - (k - w) <= rep["first_fail_pos"] <= mm_pos  # a k-mer whose minimizer is that occurrence
    with pytest.raises(AssertionError):
        ix.validate_self()  # kphf/mod.rs:69-103 would panic here


def test_validate_self_fails_on_a_cleared_boundary_bit(yeast_files):
    """the lookup locates unitigs through the boundary bit vector (bv.rank, unitig_set.rs:185-187), the validator walks the
    unitigs through their offsets: a cleared bit merges two unitigs for the lookup only"""
    f = FlatIndexFile(open(yeast_files / "noskew.mz", "rb").read())
    i = f.n_unitigs // 2
    b = f.get("starts", i) - 1  # last base of unitig i-1
    wi = (b >> 8) * 8 + 4 + ((b & 255) >> 6)
    word = f.get("bv_blocks", wi)
    assert (word >> (b & 63)) & 1
    f.put("bv_blocks", wi, word & ~(1 << (b & 63)))
    ix = _load_corrupted(f, yeast_files / "bad_boundary.mz")
    rep = ix.validate_self_report()
    assert rep["n_fail"] > 0 and rep["n_kmers"] == ix.n_kmers()
    assert b - 256 <= rep["first_fail_pos"] <= b  # a k-mer of a sequence record that holds the cleared bit
    with pytest.raises(AssertionError):
        ix.validate_self()


def test_validate_self_fails_on_a_substituted_base(yeast_files):
    f = FlatIndexFile(open(yeast_files / "noskew.mz", "rb").read())
    x = f.total_len // 3
    word = f.get("useq", x >> 5)
    f.put("useq", x >> 5, word ^ (1 << (2 * (x & 31))))
    ix = _load_corrupted(f, yeast_files / "bad_base.mz")
    rep = ix.validate_self_report()
    assert 2 <= rep["n_fail"] <= 2 * f.k and x - f.k + 1 <= rep["first_fail_pos"] <= x  # k-mers that cover the base
    with pytest.raises(AssertionError):
        ix.validate_self()


def test_validate_self_fails_on_corrupted_skew_slots(yeast_files):
    """the skew table also holds the neighbours of every heavy occurrence, which are looked up through their own light
    buckets; only k-mers whose OWN minimizer is heavy go through it.  Every slot gets a wrong position (same fingerprint):
    exactly those k-mers must fail, the others must not."""
    f = FlatIndexFile(open(yeast_files / "skew.mz", "rb").read())
    assert f.has_skew and f.n("skew_buckets") > 16
    off, n = f.vec["skew_buckets"]
    slots = np.frombuffer(bytes(f.d[off:off + 8 * n]), dtype=np.uint64).copy()
    used = slots != np.uint64(2 ** 64 - 1)
    pos = slots & np.uint64((1 << 36) - 1)
    flip = used & ((pos ^ np.uint64(1)) <= np.uint64(f.total_len - f.k))
    slots[flip] ^= np.uint64(1)
    f.d[off:off + 8 * n] = slots.tobytes()
    good = M.SSHash.load(yeast_files / "skew.mz")
    ix = _load_corrupted(f, yeast_files / "bad_skew.mz")
    rep = ix.validate_self_report()
    assert 0 < rep["n_fail"] <= 2 * int(used.sum()) and rep["n_kmers"] == ix.n_kmers()
    fw = H.kmer_words_at(good.as_ref().useq_words(), [rep["first_fail_pos"]], f.k)
    assert good.k2u_batch(fw)[0]["match"][0] == 1 and ix.k2u_batch(fw)[0]["match"][0] == 0
    with pytest.raises(AssertionError):
        ix.validate_self()


def test_skew_fingerprint_collision_with_an_overflowed_key(monkeypatch, tmp_path):
    """4-bit fingerprints and every k-mer in the skew index: full buckets spill keys into the overflow list, and for many
    of those keys exactly one slot of the full home bucket shares the fingerprint.  The staged kernel must then fall back
    to the exact walk (which searches the overflow list) instead of reporting a miss."""
    monkeypatch.setenv("MZ_TEST_SKEW_FP_BITS", "4")
    k, seqs = H.load_cf_seqs(CF_YEAST)
    words, total, starts = H.pack_seqs(seqs)
    for on_device in (False, True):
        us = M.UnitigSet.from_packed(words, total, starts, k)
        ix = M.SSHash.from_unitig_set_on_device(us, 13, 0) if on_device else M.SSHash.from_unitig_set(us, 13, 0)
        assert ix.n_kmers_in_skew_index() == ix.n_kmers()
        ix.save(tmp_path / "fp4.mz")
        f = FlatIndexFile(open(tmp_path / "fp4.mz", "rb").read())
        assert f.n("skew_ovf_word") > 100  # keys beyond the 16th of their bucket
        rep = ix.validate_self_report()
        assert rep["n_fail"] == 0 and rep["n_kmers"] == ix.n_kmers()
    monkeypatch.delenv("MZ_TEST_SKEW_FP_BITS")
    orc = O.OracleSSHash(words, total, starts, k, 13, 0)
    q = H.query_mix(words, starts, k, 200_000, seed=5)
    got, _ = ix.k2u_batch(q)
    want, _ = orc.k2u_batch(q, n_threads=8)
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("w,skew", [(13, 32), (9, 4), (15, 16)])
def test_gpu_build_with_implied_strand_class(monkeypatch, tmp_path, w, skew):
    """occurrence keys one bit short (pan-genome lengths with w = 15) leave the strand class out of the sort key and derive
    it from the sequence; forced here on a small set: the arrays must still be the host builder's"""
    us = M.UnitigSet.from_cf_prefix(CF_YEAST)
    host = M.SSHash.from_unitig_set(us, w, skew)
    monkeypatch.setenv("MZ_TEST_GPU_BUILD_IMPLIED_CLASS", "1")
    dev = M.SSHash.from_unitig_set_on_device(us, w, skew)
    host.save(tmp_path / "h.mz")
    dev.save(tmp_path / "d.mz")
    assert open(tmp_path / "h.mz", "rb").read() == open(tmp_path / "d.mz", "rb").read()
    dev.validate_self()
