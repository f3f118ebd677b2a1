"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mazu_amd.h declares, host logic (UnitigSet / builder / file format / error codes) is
right, the product never touches the oracle, and the query path refuses to run without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H
import mazu_amd as M
from oracle import oracle as O

ROOT = H.ROOT
CF_TINY = os.path.join(H.GOLDEN, "cf", "tiny", "tiny")
CF_YEAST = os.path.join(H.GOLDEN, "cf", "yeast_chr7", "yeast_chr7")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mazu_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", M._capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (mz_[a-z0-9_]+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert set(M.lib._mz_symbols) == set(syms)  # the ctypes binding covers the whole header
    assert M.lib.mz_abi_version() == 1


def test_product_never_touches_the_oracle():
    pat = re.compile(r"oracle|liboracle|orc_")
    for base in ("mazu_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                    text = open(os.path.join(dp, fn), errors="ignore").read()
                    assert not pat.search(text), f"{fn} mentions the oracle"
    # and the shared object has no CPU query entry point
    out = subprocess.check_output(["nm", "-D", "--defined-only", M._capi.LIB_PATH], text=True)
    assert "host_k2u" not in out and "cpu_k2u" not in out


def test_unitig_set_tiny():  # unitig_set.rs:355-381
    us = M.UnitigSet.from_cf_prefix(CF_TINY)
    assert us.k() == 7 and us.n_unitigs() == 2
    assert us.unitig_seq(0) == "CACACACCAC" and us.unitig_seq(1) == "CCTCAATACG"
    assert us.unitig_len(0) == 10 and us.unitig_len(1) == 10
    assert all(us.pos_to_id(i) == 0 for i in range(10)) and all(us.pos_to_id(i) == 1 for i in range(10, 20))
    assert us.total_len() == 20 and us.n_kmers() == 8


def test_unitig_set_from_seqs_and_packed_agree():
    k, seqs = H.load_cf_seqs(CF_YEAST)
    a = M.UnitigSet.from_seqs(seqs, k)
    words, total, starts = H.pack_seqs(seqs)
    b = M.UnitigSet.from_packed(words, total, starts, k)
    c = M.UnitigSet.from_cf_prefix(CF_YEAST)
    for us in (a, b, c):
        assert (us.k(), us.n_unitigs(), us.total_len(), us.n_kmers()) == (31, 471, 1085476, 1071346)
        assert np.array_equal(us.useq_words()[: len(words) - 2], words[:-2])
    rng = np.random.default_rng(1)
    orc = O.OracleSSHash(words, total, starts, k, 15, with_dict=False)
    for i in rng.integers(0, 471, 50):
        assert a.unitig_len(int(i)) == orc.unitig_len(int(i)) and a.unitig_start_pos(int(i)) == orc.unitig_start(int(i))
    for p in rng.integers(0, total, 2000):
        assert a.pos_to_id(int(p)) == orc.pos_to_id(int(p))
    for i in range(471):  # every boundary, both sides
        s = int(starts[i])
        assert a.pos_to_id(s) == i and a.pos_to_id(int(starts[i + 1]) - 1) == i


def test_error_codes():
    us = M.UnitigSet.from_cf_prefix(CF_TINY)
    with pytest.raises(M.MazuError) as e:  # assert!(w <= k), sshash.rs:92
        M.SSHash.from_unitig_set(us, 8, 0)
    assert e.value.code == M._capi.MZ_EINVAL
    with pytest.raises(M.MazuError) as e:
        M.UnitigSet.from_seqs(["ACGTN"], 3)
    assert e.value.code == M._capi.MZ_EINVAL
    with pytest.raises(M.MazuError) as e:
        M.UnitigSet.from_cf_prefix("/nonexistent/prefix")
    assert e.value.code == M._capi.MZ_EIO
    with pytest.raises(M.MazuError) as e:
        M.UnitigSet.from_seqs(["ACGT"], 32)
    assert e.value.code == M._capi.MZ_EINVAL
    with pytest.raises(M.MazuError) as e:
        M.SSHash.load("/nonexistent/file")
    assert e.value.code == M._capi.MZ_EIO


def test_queries_fail_loudly_without_a_gpu():
    if M.lib.mz_device_count() > 0:
        pytest.skip("a GPU is visible")
    ix = M.SSHash.from_unitig_set_no_skew_index(M.UnitigSet.from_cf_prefix(CF_TINY), 3)
    with pytest.raises(M.MazuError) as e:
        ix.k2u(M.CanonicalKmer("CACACAC"))
    assert e.value.code == M._capi.MZ_EHIP
    with pytest.raises(M.MazuError):
        ix.validate_self()
    with pytest.raises(M.MazuError):
        ix.k2u_batch(np.zeros(4, dtype=np.uint64))


def test_builder_stats_and_file_roundtrip(tmp_path):
    us = M.UnitigSet.from_cf_prefix(CF_YEAST)
    # caching.rs:238-253 uses skew_param 32 and relies on some minimizer occurring > 32 times under
    # the reference's WyHash order (test_utils.rs:7-8).  Which w-mers become heavy depends on the
    # hash order (unpinned); under ours the largest yeast_chr7 bucket holds 17..24 occurrences, so the
    # "skew index is populated" sanity check is made at skew_param 16.
    ix = M.SSHash.from_unitig_set(us, 15, 16)
    assert ix.k() == 31 and ix.w() == 15 and ix.n_unitigs() == 471 and ix.n_kmers() == 1071346
    assert ix.n_minimizers() < ix.n_kmers()  # caching.rs:247
    assert ix.n_kmers_in_skew_index() != 0   # caching.rs:248
    assert ix.as_ref().n_unitigs() == 471 and ix.unitig_seq(3) == us.unitig_seq(3)
    noskew = M.SSHash.from_unitig_set_no_skew_index(us, 15)
    assert noskew.n_kmers_in_skew_index() == 0 and ix.num_bits() > noskew.num_bits()  # sshash.rs:859
    p = tmp_path / "yeast.mzsshash"
    ix.save(p)
    jx = M.SSHash.load(p)
    assert (jx.k(), jx.w(), jx.n_minimizers(), jx.n_minimizer_occs(), jx.n_kmers_in_skew_index(), jx.num_bits()) == \
           (ix.k(), ix.w(), ix.n_minimizers(), ix.n_minimizer_occs(), ix.n_kmers_in_skew_index(), ix.num_bits())
    q = tmp_path / "again.mzsshash"
    jx.save(q)
    assert open(p, "rb").read() == open(q, "rb").read()


@pytest.fixture(scope="module")
def host_check():
    exe = os.path.join(ROOT, "tests", "cpp", "host_check")
    src = os.path.join(ROOT, "tests", "cpp", "host_check.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", src,
                           os.path.join(ROOT, "mazu_amd", "csrc", "mz_host.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("prefix,w,skew", [(CF_TINY, 3, "none"), (CF_TINY, 5, "0"), (CF_TINY, 7, "none"), (CF_TINY, 1, "2"),
                                           (CF_YEAST, 15, "none"), (CF_YEAST, 15, "32"), (CF_YEAST, 15, "16"), (CF_YEAST, 9, "4"), (CF_YEAST, 4, "none"),
                                           (CF_YEAST, 12, "16"), (CF_YEAST, 17, "32"), (CF_YEAST, 19, "16"), (CF_YEAST, 21, "none"), (CF_YEAST, 31, "4")])
def test_builder_layout_self_consistent_on_host(host_check, prefix, w, skew):
    """TEST-ONLY host evaluation of the shared lookup chain over the arrays the builder produced
    (the product exports no such entry point): every k-mer, both orientations.  Where the builder made minimizer lines -- addressed by
    the minimizer word, or by a hash of it at w > 16 -- a plain restatement of how the kernels read a line must give the chain's answer
    for every k-mer, its reverse complement and a one-base neighbour, and most answers must come from the line itself."""
    import json
    r = subprocess.run([host_check, "validate", prefix, str(w), skew, "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout)
    assert rep["n_fail"] == 0 and rep["line_fail"] == 0
    if w > 16 or (prefix, w) == (CF_YEAST, 12):
        assert rep["lines"] == 1 and rep["answered_from_line"] > 2.5 * rep["n_kmers"], rep  # of 3 queries per k-mer
        assert (rep["walk_markers"] > 0) == (w > 16) and rep["lines_unusable"] == 0


def test_builder_layout_matches_oracle_on_mixed_queries(host_check, tmp_path):
    k, seqs = H.load_cf_seqs(CF_YEAST)
    words, total, starts = H.pack_seqs(seqs)
    q = H.query_mix(words, starts, k, 100000, seed=11)
    # NEG-1 negatives too: positives with one substituted base (usually share the minimizer)
    rng = np.random.default_rng(3)
    neg1 = q[:20000] ^ (np.uint64(1) << rng.integers(0, 62, 20000).astype(np.uint64))
    q = np.concatenate([q, neg1])
    qf, of = tmp_path / "q.u64", tmp_path / "o.bin"
    q.tofile(qf)
    subprocess.check_call([host_check, "batch", CF_YEAST, "15", "16", "0", str(qf), str(of)])
    got = np.fromfile(of, dtype=M.K2UPOS_DTYPE)
    orc = O.OracleSSHash(words, total, starts, k, 15, 16)
    want, _ = orc.k2u_batch(q, n_threads=4)
    truth, _ = orc.k2u_batch(q, n_threads=4, use_dict=True)
    assert want.tobytes() == truth.tobytes()
    assert got.tobytes() == want.tobytes()
