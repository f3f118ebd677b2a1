"""Synthetic unitig sets and query mixes for bench.py and the large-size tests.

Everything is generated with torch ops on whatever device the generator lives on (the GPU in
bench.py, the CPU in the `not gpu` tests), from a counter-free seeded torch.Generator, so every
rank of a multi-GPU run derives the same unitig set from the same seed.

Sequence model.  Bases are uniform over ACGT: a 2-bit packed random sequence IS a stream of
uniform 64-bit words.  Unitig lengths are k + Exp(mean) (capped), matching the mean / max of
the cuttlefish GRCh38 k=31 statistics quoted in reference src/cuttlefish.rs:25-44.  Minimizer
skew is created WITHOUT creating duplicate 31-mers: `n_families` random 16-mers are each
planted many times, each in the low half of a distinct 64-bit word.  A planted 16-mer carries two 15-mers; where
one of them wins the window it becomes a heavy minimizer (bucket sizes range from a handful
to ~copies), while every 31-mer around a plant still has >= 15 random flanking bases, so
canonical 31-mers stay unique up to the birthday bound of the random background
(~n^2 / 4^31 pairs; reported by validate_self as n_duplicate, never as failures).
"""
import math

import torch

PRESETS = {
    # name: n_unitigs, mean extra length, cap, fraction of bases planted, families, exact total length (None = as drawn)
    "s-tiny": dict(n_unitigs=300, mean_extra=400.0, cap=5000, plant_frac=0.08, n_families=8, total_len=None),
    "s-yeast": dict(n_unitigs=6000, mean_extra=1985.0, cap=80000, plant_frac=0.08, n_families=64, total_len=None),
    "s-txome": dict(n_unitigs=1_200_000, mean_extra=90.0, cap=20000, plant_frac=0.04, n_families=512, total_len=None),
    "s-eighth": dict(n_unitigs=36145130 // 8, mean_extra=68.8, cap=33414 - 31, plant_frac=0.08, n_families=2048, total_len=3_589_802_592 // 8),
    # cuttlefish GRCh38 k=31 (reference src/cuttlefish.rs:25-44): 36 145 130 unitigs, 3 589 802 592 bp, longest 33 414
    "s-human": dict(n_unitigs=36145130, mean_extra=68.8, cap=33414 - 31, plant_frac=0.08, n_families=16384, total_len=3_589_802_592),
    # pan-genome scale (an assumption: no in-tree figure exists): 120 M unitigs, 9 Gbp
    "s-pan": dict(n_unitigs=120_000_000, mean_extra=44.5, cap=33414 - 31, plant_frac=0.08, n_families=32768, total_len=9_000_000_000),
    # the same shapes with REPEAT FAMILIES instead of planted 16-mers (see _plant_repeat_families): 8 % of the sequence are copies of
    # 300-base family consensus sequences, each copy with its own 5-15 % of substituted bases, ~480 copies per family.  The minimizers
    # inside the copies get buckets of every size from one to hundreds of entries -- a continuum, where the planted 16-mers give
    # buckets of ~3 or ~1000.
    "s-human-rep": dict(n_unitigs=36145130, mean_extra=68.8, cap=33414 - 31, plant_frac=0.0, n_families=0, total_len=3_589_802_592,
                        rep_frac=0.08, rep_families=2000, rep_len=300, rep_div=(0.05, 0.15)),
    "s-eighth-rep": dict(n_unitigs=36145130 // 8, mean_extra=68.8, cap=33414 - 31, plant_frac=0.0, n_families=0, total_len=3_589_802_592 // 8,
                         rep_frac=0.08, rep_families=250, rep_len=300, rep_div=(0.05, 0.15)),
    "s-tiny-rep": dict(n_unitigs=300, mean_extra=400.0, cap=5000, plant_frac=0.0, n_families=0, total_len=None,
                       rep_frac=0.08, rep_families=2, rep_len=300, rep_div=(0.05, 0.15)),
}


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def _fit_total(lens, k, cap, target):
    """nudge unitig lengths (every one stays in [k, k + cap]) so that they sum to `target` exactly"""
    n = lens.numel()
    delta = int(target) - int(lens.sum())
    while delta != 0:
        step = 1 if delta > 0 else -1
        room = (lens < k + cap) if step > 0 else (lens > k)
        idx = torch.nonzero(room).flatten()
        if idx.numel() == 0:
            raise ValueError("cannot reach the requested total length")
        take = min(abs(delta), idx.numel())
        # spread over the eligible unitigs at a fixed stride, so the adjustment is not piled on the first ones
        pick = idx[(torch.arange(take, device=lens.device) * (idx.numel() // take))]
        lens[pick] += step
        delta -= step * take
    return lens


def make_unitig_set(preset, k=31, seed=0x6D617A75, device="cpu", total_len=None):
    """returns dict(words int64[nw+2], starts int64[n+1], total_len, n_unitigs, k) on `device`.
    total_len (or the preset's) makes the sum of the unitig lengths exact."""
    p = dict(PRESETS[preset]) if isinstance(preset, str) else dict(preset)
    dev = torch.device(device)
    g = _gen(dev, seed)
    n = p["n_unitigs"]
    u = torch.rand(n, generator=g, device=dev, dtype=torch.float64).clamp_(1e-12, 1.0)
    extra = (-torch.log(u) * p["mean_extra"]).floor_().clamp_(max=float(p["cap"])).to(torch.int64)
    lens = extra + k
    del u, extra
    target = total_len if total_len is not None else p.get("total_len")
    if target is not None:
        lens = _fit_total(lens, k, int(p["cap"]), target)
    starts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=starts[1:])
    del lens
    L = int(starts[-1])
    nw = (L + 31) // 32
    words = torch.randint(-(2**63), 2**63 - 1, (nw + 2,), generator=g, device=dev, dtype=torch.int64)
    # planted 16-mers at 32-bit aligned half-word slots
    n_fam = p["n_families"]
    n_plants = int(p["plant_frac"] * L / 16)
    if n_fam > 0 and n_plants > 0 and nw > 2:
        fam_words = torch.randint(0, 2**32, (n_fam,), generator=g, device=dev, dtype=torch.int64)
        # low half of distinct words only: two plants are always separated by >= 16 other bases,
        # so no 31-mer is made of planted bases alone
        w_idx = torch.randperm(nw - 1, generator=g, device=dev)[:n_plants] + 1
        fam = torch.randint(0, n_fam, (w_idx.numel(),), generator=g, device=dev)
        # Two copies of one family would otherwise share 31-mers with probability 4^-15 per pair and offset.  The 8 bases
        # on either side of a plant are therefore a tag that is unique among the copies of its family (a bijection of the
        # copy's rank): a 31-mer that holds a whole plant also holds one whole tag, so copies never repeat a 31-mer.
        order = torch.argsort(fam, stable=True)
        counts = torch.bincount(fam, minlength=n_fam)
        if int(counts.max()) > 65536:
            raise ValueError("more than 65536 copies of one planted family")
        first = torch.cumsum(counts, 0) - counts
        rank = torch.empty_like(fam)
        rank[order] = torch.arange(fam.numel(), device=dev) - first[fam[order]]
        tag_r = (rank * 40503 + fam * 0x9E37) & 0xFFFF
        tag_l = (rank * 25173 + fam * 0x7F4B + 0x1234) & 0xFFFF
        del order, counts, first, rank
        words[w_idx] = (words[w_idx] & -(1 << 48)) | fam_words[fam] | (tag_r << 32)     # plant + the 8 bases after it
        words[w_idx - 1] = (words[w_idx - 1] & ((1 << 48) - 1)) | (tag_l << 48)         # the 8 bases before it
        del w_idx, fam, tag_r, tag_l
    # zero the slack bits and the two pad words
    words[nw:] = 0
    if L & 31:
        keep = (1 << (2 * (L & 31))) - 1
        words[nw - 1] &= keep
    n_rep_sub = _plant_repeat_families(words, L, k, p, g, dev) if p.get("rep_frac") else 0
    return dict(words=words, starts=starts, total_len=L, n_unitigs=n, k=k, preset=preset if isinstance(preset, str) else "custom",
                rep_substitutions=n_rep_sub)


_M62 = (1 << 62) - 1


def kmer_words_at(words, pos, k=31):
    """forward 2-bit words of the k-mers (k <= 31) starting at global positions `pos` (int64 tensor) of a packed sequence"""
    wi, sh = pos >> 5, (pos & 31) * 2
    lo, hi = words[wi], words[wi + 1]
    zero = sh == 0
    a = (lo >> sh) & torch.where(zero, torch.full_like(sh, -1), (torch.ones_like(sh) << (64 - sh).clamp(max=62)) - 1)
    b = torch.where(zero, torch.zeros_like(hi), hi << (64 - sh).clamp(max=63))
    return (a | b) & ((1 << (2 * k)) - 1)


def kmer_rc_words(x, k=31):
    """reverse complements of 2-bit k-mer words (int64 tensor, k <= 31)"""
    y = ~x
    for s_, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF), (32, 0x00000000FFFFFFFF)):
        y = ((y >> s_) & m) | ((y & m) << s_)
    return (y >> (64 - 2 * k)) & ((1 << (2 * k)) - 1)


def _add_codes(words, pos, new_codes):
    """set the 2-bit codes at the (distinct) global positions `pos`: every field changes by (new - old) << shift, summed exactly"""
    wi, sh = pos >> 5, (pos & 31) * 2
    old = (words[wi] >> sh) & 3
    words.index_add_(0, wi, (new_codes - old) << sh)


def _plant_repeat_families(words, L, k, p, g, dev):
    """Repeat families as in a genome's interspersed repeats: `rep_families` random consensus sequences of `rep_len` bases; copies
    fill `rep_frac` of the sequence at distinct slots, every copy with its own substitution rate drawn from `rep_div` and its own
    random substitutions.  Copies of one family share many 15-mers (the minimizers inside them own buckets of every size) and, at
    first, also 31-mers -- which a cdBG cannot contain twice.  Those are removed here on the device: the canonical 31-mers of all
    windows that touch a copy are sorted, and in every stretch shared by two copies bases are substituted, one per 16 positions
    (enough to make each 31-base window differ), until no window of a copy repeats.  Returns the number of bases substituted."""
    rep_len, n_fam = int(p["rep_len"]), int(p["rep_families"])
    n_slots = (L - 2 * k) // rep_len
    n_copies = min(int(p["rep_frac"] * L / rep_len), n_slots)
    if n_copies <= 0 or n_fam <= 0:
        return 0
    cons = torch.randint(0, 4, (n_fam, rep_len), generator=g, device=dev, dtype=torch.int64)
    slot = torch.randperm(n_slots, generator=g, device=dev)[:n_copies]
    start = slot * rep_len + k  # (the first k bases of the sequence stay background)
    fam = torch.randint(0, n_fam, (n_copies,), generator=g, device=dev)
    lo_d, hi_d = p["rep_div"]
    n_sub_total = 0
    chunk = max(1, (64 << 20) // rep_len)  # copies per pass: bounded temporaries
    for a in range(0, n_copies, chunk):
        st, fm = start[a:a + chunk], fam[a:a + chunk]
        div = lo_d + (hi_d - lo_d) * torch.rand(st.numel(), generator=g, device=dev)
        sub = torch.rand(st.numel(), rep_len, generator=g, device=dev) < div[:, None]
        shift = torch.randint(1, 4, (st.numel(), rep_len), generator=g, device=dev, dtype=torch.int64)
        codes = torch.where(sub, (cons[fm] + shift) & 3, cons[fm])
        pos = st[:, None] + torch.arange(rep_len, device=dev)[None, :]
        _add_codes(words, pos.reshape(-1), codes.reshape(-1))
        del div, sub, shift, codes, pos
    # ---- no 31-mer of a copy may occur twice
    win = torch.arange(-(k - 1), rep_len, device=dev)  # windows that touch a copy
    for rnd in range(16):
        # all windows at once would be 330 x n_copies words; the sort needs them together (duplicates pair ACROSS copies)
        pos = torch.unique((start[:, None] + win[None, :]).reshape(-1))  # (the windows of neighbouring slots overlap)
        pos = pos[pos <= L - k]
        fw = kmer_words_at(words, pos, k)
        canon = torch.minimum(fw, kmer_rc_words(fw, k))
        del fw
        srt, order = torch.sort(canon)
        del canon
        eq = srt[1:] == srt[:-1]
        del srt
        later = pos[order[1:][eq]]  # every occurrence but the first of its sorted run
        del order, eq, pos
        n_dup = int(later.numel())
        if n_dup == 0:
            break
        # one substitution per 16 positions of a shared stretch breaks every 31-base window of it; after a few rounds, whatever is left
        pick = later if rnd >= 6 else later[(later & 15) == ((5 * rnd) & 15)]
        if pick.numel() == 0:
            continue
        # (late rounds: at a random base of the window -- by then the copies of a family may carry all four bases at a window's middle)
        tgt = torch.unique(pick + (k // 2 if rnd < 6 else torch.randint(0, k, (pick.numel(),), generator=g, device=dev)))
        old = (words[tgt >> 5] >> ((tgt & 31) * 2)) & 3
        _add_codes(words, tgt, (old + torch.randint(1, 4, (tgt.numel(),), generator=g, device=dev, dtype=torch.int64)) & 3)  # (drawn per copy: copies must part ways)
        n_sub_total += int(tgt.numel())
        del later, pick, tgt, old
    else:
        raise AssertionError("repeat copies still share 31-mers after 16 rounds")
    return n_sub_total


def mutate_positions(us, positions):
    """substitute the base at each global position (code + 1 mod 4) in place: used to break up the handful of duplicate
    31-mers that a random background of billions of bases holds by the birthday bound"""
    words = us["words"]
    for p in sorted(set(int(x) for x in positions)):
        wi, sh = p >> 5, (p & 31) * 2
        v = int(words[wi]) & (2**64 - 1)
        old = (v >> sh) & 3
        v ^= (old ^ ((old + 1) & 3)) << sh
        words[wi] = v - 2**64 if v >= 2**63 else v
    return us


def build_index_unique(us, w, skew_param, seed=0, device=0, on_device=True, max_rounds=4, log=None, check=True, **build_opts):
    """build the SSHash of a synthetic unitig set and make its canonical k-mers unique: duplicates reported by the device
    validate_self (a few at GRCh38 scale: the birthday bound of random sequence) are broken up by substituting one base
    in the middle of the later copy, and the index is rebuilt.  Returns (index, report, rounds).  check=False builds once and
    validates nothing (stage-cost experiments with kernels whose results are meaningless).  build_opts: the layout choices of
    SSHash.from_unitig_set_opts (minimizer_lines, lines_max_load, direct_bytes_per_key)."""
    from . import kphf
    import numpy as np
    k = us["k"]
    for rnd in range(max_rounds):
        words_h = us["words"].cpu().numpy().view(np.uint64)
        starts_h = us["starts"].cpu().numpy().view(np.uint64)
        uset = kphf.UnitigSet.from_packed(words_h, us["total_len"], starts_h, k)
        ix = kphf.SSHash.from_unitig_set_opts(uset, w, skew_param, seed, device=device if on_device else -1, **build_opts)
        del uset
        ix.to_device(device)
        if not check:
            return ix, None, 0
        rep, dups = ix.validate_self_duplicates()
        if log:
            log(f"validate_self round {rnd}: {rep}")
        if rep["n_fail"] != 0:
            raise AssertionError(f"validate_self failed: {rep}")
        if rep["n_duplicate"] == 0:
            return ix, rep, rnd
        if len(dups) < min(rep["n_duplicate"], 1):
            raise AssertionError("duplicates reported but not listed")
        mutate_positions(us, [int(p) + k // 2 for p in dups])
        del ix
    raise AssertionError("duplicate k-mers remain after %d rounds" % max_rounds)


def sample_positive_positions(starts, k, n, generator):
    """n global positions of valid k-mers, uniform over all k-mers of the unitig set."""
    dev = starts.device
    n_unitigs = starts.numel() - 1
    # k-mer index space: unitig u owns [kcum[u], kcum[u+1]) with kcum[u] = starts[u] - u*(k-1)
    kcum = starts - torch.arange(n_unitigs + 1, device=dev, dtype=torch.int64) * (k - 1)
    n_kmers = int(kcum[-1])
    r = torch.randint(0, n_kmers, (n,), generator=generator, device=dev, dtype=torch.int64)
    u = torch.searchsorted(kcum, r, right=True) - 1
    return r + u * (k - 1)


def make_query_mix(sshash, us, n, seed, device, neg_frac=0.5, neg_kind="uniform"):
    """n forward words on `device`: (1-neg_frac) positives drawn through the index's own packed
    sequence (random orientation), neg_frac negatives, shuffled.  neg_kind "uniform" = NEG-U (uniform
    2k-bit words), "sub1" = NEG-1 (a positive with one substituted base: usually shares the minimizer
    and the flanks, so it reaches the k-mer verification).  Needs a GPU (positives are extracted by the HIP kernel)."""
    dev = torch.device(device)
    g = _gen(dev, seed)
    k = us["k"]
    n_neg = int(n * neg_frac)
    n_pos = n - n_neg
    starts = us["starts"].to(dev)
    q = torch.empty(n, dtype=torch.int64, device=dev)
    n_draw = n if neg_kind == "sub1" else n_pos
    pos = sample_positive_positions(starts, k, n_draw, g)
    flip = torch.randint(0, 2, (n_draw,), generator=g, device=dev, dtype=torch.uint8)
    sshash.extract_kmers_device(pos, flip, q[:n_draw])
    if neg_kind == "sub1":
        base = torch.randint(0, k, (n_neg,), generator=g, device=dev, dtype=torch.int64)
        delta = torch.randint(1, 4, (n_neg,), generator=g, device=dev, dtype=torch.int64)
        q[n_pos:] ^= delta << (2 * base)
    elif neg_kind == "uniform":
        q[n_pos:] = torch.randint(0, 2 ** (2 * k), (n_neg,), generator=g, device=dev, dtype=torch.int64)
    else:
        raise ValueError(neg_kind)
    torch.cuda.synchronize(dev)
    perm = torch.randperm(n, generator=g, device=dev)
    return q[perm].contiguous()


def make_reads(us, n_reads, read_len, seed, device="cpu", err_frac=0.5, err_rate=0.01):
    """reads drawn from the concatenated unitig sequence (so some span unitig junctions: those
    windows are true negatives), random strand, `err_frac` of reads get `err_rate` substitutions.
    Returns (bases uint8[n_reads*read_len] ASCII, read_offsets int64[n_reads+1]) on `device`."""
    dev = torch.device(device)
    g = _gen(dev, seed)
    L = us["total_len"]
    words = us["words"].to(dev)
    start = torch.randint(0, L - read_len, (n_reads,), generator=g, device=dev, dtype=torch.int64)
    idx = start[:, None] + torch.arange(read_len, device=dev)[None, :]
    codes = (words[idx >> 5] >> ((idx & 31) * 2)) & 3
    # substitutions
    has_err = torch.rand(n_reads, generator=g, device=dev) < err_frac
    sub = (torch.rand(n_reads, read_len, generator=g, device=dev) < err_rate) & has_err[:, None]
    shift = torch.randint(1, 4, (n_reads, read_len), generator=g, device=dev, dtype=torch.int64)
    codes = torch.where(sub, (codes + shift) & 3, codes)
    # random strand
    rcflag = torch.rand(n_reads, generator=g, device=dev) < 0.5
    codes = torch.where(rcflag[:, None], (3 - codes).flip(1), codes)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    bases = lut[codes].reshape(-1).contiguous()
    ro = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * read_len
    return bases, ro


def kmer_offsets(read_offsets, k):
    lens = read_offsets[1:] - read_offsets[:-1]
    nw = (lens - k + 1).clamp_(min=0)
    ko = torch.zeros_like(read_offsets)
    torch.cumsum(nw, 0, out=ko[1:])
    return ko


def expected_sizes(preset, k=31):
    p = PRESETS[preset]
    L = p["n_unitigs"] * (k + p["mean_extra"])
    return dict(total_len=L, n_kmers=L - p["n_unitigs"] * (k - 1), useq_gb=L / 4 / 1e9, note=math.nan)
