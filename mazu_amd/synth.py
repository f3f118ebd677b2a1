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
    # name: n_unitigs, mean extra length, cap, fraction of bases planted, families
    "s-tiny": dict(n_unitigs=300, mean_extra=400.0, cap=5000, plant_frac=0.08, n_families=8),
    "s-yeast": dict(n_unitigs=6000, mean_extra=1985.0, cap=80000, plant_frac=0.08, n_families=64),
    "s-txome": dict(n_unitigs=1_200_000, mean_extra=90.0, cap=20000, plant_frac=0.04, n_families=512),
    "s-eighth": dict(n_unitigs=36145130 // 8, mean_extra=68.3, cap=33414 - 31, plant_frac=0.08, n_families=2048),
    "s-human": dict(n_unitigs=36145130, mean_extra=68.3, cap=33414 - 31, plant_frac=0.08, n_families=16384),
}


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def make_unitig_set(preset, k=31, seed=0x6D617A75, device="cpu", total_len=None):
    """returns dict(words int64[nw+2], starts int64[n+1], total_len, n_unitigs, k) on `device`."""
    p = dict(PRESETS[preset]) if isinstance(preset, str) else dict(preset)
    dev = torch.device(device)
    g = _gen(dev, seed)
    n = p["n_unitigs"]
    u = torch.rand(n, generator=g, device=dev, dtype=torch.float64).clamp_(1e-12, 1.0)
    extra = (-torch.log(u) * p["mean_extra"]).floor_().clamp_(max=float(p["cap"])).to(torch.int64)
    lens = extra + k
    starts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=starts[1:])
    if total_len is not None:  # stretch / trim the last unitig so that the sum is exact
        delta = int(total_len) - int(starts[-1])
        if int(lens[-1]) + delta < k:
            raise ValueError("cannot reach requested total_len")
        starts[-1] += delta
    L = int(starts[-1])
    nw = (L + 31) // 32
    words = torch.randint(-(2**63), 2**63 - 1, (nw + 2,), generator=g, device=dev, dtype=torch.int64)
    # planted 16-mers at 32-bit aligned half-word slots
    n_fam = p["n_families"]
    n_plants = int(p["plant_frac"] * L / 16)
    if n_fam > 0 and n_plants > 0:
        fam_words = torch.randint(0, 2**32, (n_fam,), generator=g, device=dev, dtype=torch.int64)
        # low half of distinct words only: two plants are always separated by >= 16 random bases,
        # so no 31-mer is made of planted bases alone
        w_idx = torch.randperm(nw, generator=g, device=dev)[:n_plants]
        fam = torch.randint(0, n_fam, (w_idx.numel(),), generator=g, device=dev)
        lo_mask = torch.tensor(0xFFFFFFFF, dtype=torch.int64, device=dev)
        words[w_idx] = (words[w_idx] & ~lo_mask) | fam_words[fam]
    # zero the slack bits and the two pad words
    words[nw:] = 0
    if L & 31:
        keep = (1 << (2 * (L & 31))) - 1
        words[nw - 1] &= keep
    return dict(words=words, starts=starts, total_len=L, n_unitigs=n, k=k, preset=preset if isinstance(preset, str) else "custom")


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
