"""Bucket-size statistics of a synthetic index and a packing simulation for 128-byte minimizer lines (design aid).

    python tools/front_stats.py s-human

Builds the index on the GPU, reads its entry array back from the flat file, derives (key, bucket size) and simulates, for
several (slots per line, keys per line) settings, how much of the entry mass a line holds inline."""
import struct
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import mazu_amd as M  # noqa: E402
from mazu_amd import synth  # noqa: E402
import helpers as H  # noqa: E402


def vec_offsets(path):
    """(offset, count) of every length-prefixed vector of the flat index file, in file order"""
    out = {}
    with open(path, "rb") as fh:
        assert fh.read(8) == b"MZSSH013"
        fh.seek(8 + 8 + 40)
        (has_skew,) = struct.unpack("<I", fh.read(4))

        def vec(name, item=8):
            (n,) = struct.unpack("<Q", fh.read(8))
            out[name] = (fh.tell(), n)
            fh.seek(n * item, 1)
        for nm in ("useq", "starts", "bv_blocks"):
            vec(nm)
        fh.seek(8 * 3 + 4, 1)
        for nm in ("ef_high", "ef_sel", "ef_low"):
            vec(nm)
        fh.seek(8 + 4, 1)
        vec("mm_blocks")
        vec("mm_levels", 24)
        fh.seek(16, 1)
        vec("bef_blocks")
        vec("bef_ovf")
        fh.seek(8 + 4, 1)
        vec("pos")
    return out


name = sys.argv[1] if len(sys.argv) > 1 else "s-human"
skew = 32
t = time.time()
us = synth.make_unitig_set(name, device="cuda")
words = us["words"].cpu().numpy().view(np.uint64)
starts = us["starts"].cpu().numpy().view(np.uint64)
ix = M.SSHash.from_unitig_set_on_device(M.UnitigSet.from_packed(words, us["total_len"], starts, 31), 15, skew)
print("built", round(time.time() - t, 1), "s;", ix.n_minimizers(), "minimizers,", ix.n_minimizer_occs(), "entries", flush=True)
ix.save("/dev/shm/fs.mz")
del ix
off, n = vec_offsets("/dev/shm/fs.mz")["pos"]
ent = np.fromfile("/dev/shm/fs.mz", dtype=np.uint64, count=n - 2, offset=off)
pos = ent >> np.uint64(17)
cls = ((ent >> np.uint64(16)) & np.uint64(1)).astype(bool)
del ent
wm = H.kmer_words_at(words, pos, 15)
key = np.where(cls, H.rc_words(wm, 15), wm).astype(np.uint32)
del wm, pos, cls
print("keys derived", round(time.time() - t, 1), flush=True)
chg = np.flatnonzero(key[1:] != key[:-1]) + 1
st = np.concatenate([[0], chg])
sizes = np.diff(np.concatenate([st, [len(key)]])).astype(np.int64)
keys = key[st].astype(np.int64)
n_ent = len(key)
del key
h = np.bincount(np.minimum(sizes, 40))
print("buckets:", len(keys), "entries:", n_ent, "mean", round(sizes.mean(), 3), "max", sizes.max())
print("bucket count by size (last = 40+):", h.tolist())
print("entry mass by size:", (h * np.arange(len(h)) / n_ent).round(4).tolist())
heavy = sizes > skew
print("heavy buckets:", int(heavy.sum()), "entry mass in heavy:", round(sizes[heavy].sum() / n_ent, 4))
for slots, g in ((16, 3), (16, 4), (8, 1), (8, 2), (8, 3), (8, 4)):
    line = keys >> g
    # greedy by ascending size inside a line, one slot reserved for the marker of every bucket not (yet) inline
    s_eff = np.where(heavy, 1 << 20, sizes)
    order = np.lexsort((s_eff, line))
    l_s, z_s = line[order], s_eff[order]
    first = np.concatenate([[True], l_s[1:] != l_s[:-1]])
    idx_first = np.flatnonzero(first)
    grp = np.cumsum(first) - 1
    m = np.diff(np.concatenate([idx_first, [len(l_s)]]))  # members per line
    rank = np.arange(len(l_s)) - idx_first[grp]
    cs = np.cumsum(z_s)
    cs_excl_line = cs - (cs[idx_first[grp]] - z_s[idx_first[grp]])  # inclusive cumsum inside the line
    inline = cs_excl_line + (m[grp] - rank - 1) <= slots
    light = z_s < (1 << 20)
    mass_inline = z_s[inline & light].sum() / n_ent
    n_lines = (1 << 30) >> g
    used = np.zeros(len(m), dtype=np.int64)
    np.add.at(used, grp, np.where(inline, z_s, 1))
    print(f"slots={slots} g={g}: {n_lines} lines = {n_lines * slots * 8 / 1e9:.2f} GB, occupied lines {len(m)}, mean slots used in occupied lines {used.mean():.2f}, "
          f"entry mass inline {mass_inline:.4f}, buckets inline {inline.mean():.4f}, "
          f"entry mass spilled {z_s[~inline & light].sum() / n_ent:.4f}", flush=True)
