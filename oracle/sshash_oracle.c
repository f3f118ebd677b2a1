/*
 * sshash_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of mazu's SSHash k-mer -> unitig dictionary
 * (reference: /root/reference/src/kphf/sshash.rs and the files it leans on).
 * It is the CHECKER for the HIP path in mazu_amd/csrc: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (libmazu_amd.so) never links, loads or calls this file.
 *
 * Parity status
 *   - K2UPos RESULTS (unitig_id, unitig_len, pos, match) are pinned by the
 *     reference's own known-answer tests (sshash.rs:645-763, :792-824,
 *     :841-884; kphf/mod.rs:147-161) -- see tests/golden/ and
 *     tests/test_oracle_golden.py.
 *   - INTERNALS (hash values, minimizer word/offset, MPHF slot numbering,
 *     Elias-Fano bit patterns) are "parity unpinned": the arithmetic lives in
 *     the un-vendored crates kmers (git thejasonfan/kmers@main, no rev),
 *     boomphf ^0.5.9, wyhash ^0.5.0 and simple-sds (git fork, no rev), and no
 *     reference test asserts any of those values.  The published algorithms
 *     are restated here (BBHash levels, Vigna's Elias-Fano, a wyhash-style
 *     mum hash); for unitig sets with unique canonical k-mers the results do
 *     not depend on them: a unique k-mer has one position, whatever the
 *     minimizer order and MPHF slots used to find it.
 *
 * Also contains a brute-force dictionary (sorted canonical k-mer -> global
 * position) that is independent of every succinct structure below; it is the
 * ground truth both this restatement and the HIP path are tested against.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* result record: same 16-byte shape as mz_k2upos in include/mazu_amd.h */
typedef struct {
    uint32_t unitig_id;
    uint32_t pos;
    uint32_t unitig_len;
    uint32_t match; /* 0 = None, 1 = IdentityMatch, 2 = TwinMatch, 3 = window skipped (non-ACGT) */
} orc_k2upos;

#define ORC_NONE 0u
#define ORC_IDENTITY 1u
#define ORC_TWIN 2u
#define ORC_SKIPPED 3u

/* ------------------------------------------------------------------ */
/* Word-touch counter: when enabled, every 8-byte word a QUERY reads from the index arrays of
 * this restatement (reference layout: BBHash levels + rank samples, Elias-Fano high/low/select samples, packed pos,
 * 2-bit sequence, boundary bit vector + rank samples) is tallied; orc_word_touches() reports the total.  It counts
 * words per access site (a word read twice by two sites counts twice; the sites of one query rarely overlap). */
static _Thread_local uint64_t g_touch = 0;
static int g_touch_on = 0;
#define TOUCH(n) do { if (g_touch_on) g_touch += (n); } while (0)

/* ------------------------------------------------------------------ */
/* util.rs:48-55  msb(n): index of most significant bit, msb(0)=0      */
static uint64_t orc_msb(uint64_t n) { return n == 0 ? 0 : 63u - (uint64_t)__builtin_clzll(n); }

/* number of bits needed to store v (IntVector::pack width; >=1) */
static unsigned bit_len(uint64_t v) { return v == 0 ? 1u : 64u - (unsigned)__builtin_clzll(v); }

/* ------------------------------------------------------------------ */
/* packed int vector (simple-sds IntVector restated): element i at bit i*width, LSB first */
typedef struct {
    uint64_t *w;
    uint64_t n;
    unsigned width;
} intvec;

static void iv_init(intvec *v, uint64_t n, unsigned width) {
    v->n = n;
    v->width = width;
    uint64_t bits = n * (uint64_t)width;
    v->w = (uint64_t *)calloc(bits / 64 + 2, 8);
}
static inline void iv_set(intvec *v, uint64_t i, uint64_t x) {
    uint64_t b = i * v->width;
    uint64_t wi = b >> 6;
    unsigned sh = (unsigned)(b & 63);
    uint64_t mask = v->width == 64 ? ~0ull : ((1ull << v->width) - 1);
    v->w[wi] = (v->w[wi] & ~(mask << sh)) | ((x & mask) << sh);
    if (sh + v->width > 64) {
        unsigned done = 64 - sh;
        v->w[wi + 1] = (v->w[wi + 1] & ~(mask >> done)) | ((x & mask) >> done);
    }
}
static inline uint64_t iv_get(const intvec *v, uint64_t i) {
    uint64_t b = i * v->width;
    uint64_t wi = b >> 6;
    unsigned sh = (unsigned)(b & 63);
    TOUCH(sh + v->width > 64 ? 2 : 1);
    uint64_t mask = v->width == 64 ? ~0ull : ((1ull << v->width) - 1);
    uint64_t x = v->w[wi] >> sh;
    if (sh + v->width > 64) x |= v->w[wi + 1] << (64 - sh);
    return x & mask;
}
static void iv_free(intvec *v) { free(v->w); v->w = NULL; }

/* ------------------------------------------------------------------ */
/* bit vector with rank (ones in [0,i)) and select (i-th one, 0-based);
 * conventions as simple-sds BitVector (pinned by reference src/wm.rs:520-537) */
typedef struct {
    uint64_t *w;
    uint64_t n_bits;
    uint64_t n_words;
    uint64_t *cum;   /* cum[j] = ones in words [0,j) ; n_words+1 entries */
    uint64_t *sel;   /* sel[j] = word index holding the (64*j)-th one */
    uint64_t n_ones;
} bitvec;

static void bv_init(bitvec *b, uint64_t n_bits) {
    memset(b, 0, sizeof *b);
    b->n_bits = n_bits;
    b->n_words = n_bits / 64 + 1;
    b->w = (uint64_t *)calloc(b->n_words + 1, 8);
}
static inline void bv_set(bitvec *b, uint64_t i) { b->w[i >> 6] |= 1ull << (i & 63); }
static inline int bv_get(const bitvec *b, uint64_t i) { TOUCH(1); return (int)((b->w[i >> 6] >> (i & 63)) & 1); }
static void bv_enable_rank_select(bitvec *b) {
    b->cum = (uint64_t *)malloc((b->n_words + 1) * 8);
    uint64_t c = 0;
    for (uint64_t j = 0; j < b->n_words; ++j) {
        b->cum[j] = c;
        c += (uint64_t)__builtin_popcountll(b->w[j]);
    }
    b->cum[b->n_words] = c;
    b->n_ones = c;
    b->sel = (uint64_t *)malloc((c / 64 + 2) * 8);
    uint64_t next = 0, js = 0;
    for (uint64_t j = 0; j < b->n_words; ++j) {
        while (next < b->cum[j + 1]) { /* the next-th one lives in word j */
            b->sel[js++] = j;
            next += 64;
        }
    }
}
static inline uint64_t bv_rank(const bitvec *b, uint64_t i) {
    TOUCH(2); /* cumulative count + data word */
    uint64_t wi = i >> 6;
    unsigned sh = (unsigned)(i & 63);
    uint64_t m = sh ? (b->w[wi] & ((1ull << sh) - 1)) : 0;
    return b->cum[wi] + (uint64_t)__builtin_popcountll(m);
}
static inline uint64_t bv_select(const bitvec *b, uint64_t i) {
    uint64_t j = b->sel[i >> 6];
    TOUCH(3); /* select sample + cumulative count + data word */
    while (b->cum[j + 1] <= i) { ++j; TOUCH(1); }
    uint64_t r = i - b->cum[j];
    uint64_t x = b->w[j];
    while (r--) x &= x - 1;
    return (j << 6) + (uint64_t)__builtin_ctzll(x);
}
static void bv_free(bitvec *b) { free(b->w); free(b->cum); free(b->sel); memset(b, 0, sizeof *b); }

/* ------------------------------------------------------------------ */
/* Elias-Fano, elias_fano.rs:21-122                                    */
typedef struct {
    uint64_t u, l, n;
    bitvec high;
    intvec low;
} efvec;

/* elias_fano.rs:55-114 from_iter: l = msb(u/n) forced >= 1; high length n + (u>>l) */
static int ef_build(efvec *e, const uint64_t *xs, uint64_t n, uint64_t u) {
    memset(e, 0, sizeof *e);
    if (n == 0) return -2; /* Error::EFEmpty */
    uint64_t l = orc_msb(u / n);
    if (l == 0) l = 1;
    e->u = u; e->l = l; e->n = n;
    uint64_t low_mask = (1ull << l) - 1;
    iv_init(&e->low, n, (unsigned)l);
    uint64_t hb_len = n + (u >> l);
    bv_init(&e->high, hb_len);
    uint64_t last = 0, h0s = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t x = xs[i];
        if (x < last) return -1; /* Error::EFNotMonotone */
        iv_set(&e->low, i, x & low_mask);
        uint64_t gap = (x >> l) - (last >> l);
        bv_set(&e->high, h0s + i + gap);
        last = x;
        h0s += gap;
    }
    bv_enable_rank_select(&e->high);
    return 0;
}
/* elias_fano.rs:116-122 */
static inline uint64_t ef_get(const efvec *e, uint64_t i) {
    uint64_t high = bv_select(&e->high, i) - i;
    return (high << e->l) | iv_get(&e->low, i);
}
static void ef_free(efvec *e) { bv_free(&e->high); iv_free(&e->low); }

/* ------------------------------------------------------------------ */
/* 2-bit sequence: base i at bits [2i,2i+1] of word i/32, first base in LSBs,
 * A=0 C=1 G=2 T=3 (pinned by the pufferfish fixtures under tests/golden)  */
static inline int base_code(uint8_t c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}
static inline uint64_t seq_get_kmer(const uint64_t *w, uint64_t pos, unsigned k) {
    uint64_t b = pos * 2, wi = b >> 6;
    unsigned sh = (unsigned)(b & 63);
    TOUCH(sh + 2 * k > 64 ? 2 : 1);
    uint64_t x = w[wi] >> sh;
    if (sh) x |= w[wi + 1] << (64 - sh);
    return k == 32 ? x : (x & ((1ull << (2 * k)) - 1));
}
static inline uint64_t kmer_rc(uint64_t x, unsigned k) {
    x = ~x; /* complement: A<->T (0<->3), C<->G (1<->2) */
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
}

/* ------------------------------------------------------------------ */
/* hash of a u64 key: wyhash-style mum (stands in for WyHashState, kphf/mod.rs:32-52;
 * bit-compat with the wyhash crate is unpinned and not needed, note A) */
static inline uint64_t mum(uint64_t a, uint64_t b) {
    __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
static inline uint64_t orc_hash64(uint64_t x, uint64_t seed) {
    const uint64_t P0 = 0xa0761d6478bd642full, P1 = 0xe7037ed1a0b428dbull, P2 = 0x8ebc6af09c88c6e3ull;
    return mum(mum(x ^ P0, seed ^ P1) ^ P2, 8 ^ P1);
}

/* canonical minimizer of a k-mer (call site sshash.rs:476; definition per the
 * comment sshash.rs:32-37: mini(g*) with g* = min(g, g')): the w-mer of the
 * CANONICAL k-mer with the smallest hash, leftmost on ties; returns the w-mer
 * word and its offset inside g*.  *is_fw tells whether g* == fw.          */
static inline void canonical_minimizer(uint64_t fw, unsigned k, unsigned w, uint64_t seed,
                                       uint64_t *mm_word, unsigned *offset, int *is_fw) {
    uint64_t rc = kmer_rc(fw, k);
    uint64_t g = fw <= rc ? fw : rc;
    *is_fw = fw <= rc;
    uint64_t wmask = w == 32 ? ~0ull : ((1ull << (2 * w)) - 1);
    uint64_t best_h = 0, best_w = 0;
    unsigned best_o = 0;
    for (unsigned o = 0; o + w <= k; ++o) {
        uint64_t m = (g >> (2 * o)) & wmask;
        uint64_t h = orc_hash64(m, seed);
        if (o == 0 || h < best_h) { best_h = h; best_w = m; best_o = o; }
    }
    *mm_word = best_w;
    *offset = best_o;
}

/* ------------------------------------------------------------------ */
/* MPHF: BBHash / boomphf algorithm restated (levels of bit arrays, gamma,
 * collision-free bits kept, rank = index).  try_hash semantics as boomphf:
 * None when no level holds the key's bit.  Slot numbering is NOT boomphf's. */
typedef struct {
    unsigned n_levels;
    bitvec *levels;
    uint64_t *level_rank0; /* ones in all previous levels */
    uint64_t n_keys;
} mphf;

static inline uint64_t mphf_level_pos(uint64_t key, unsigned level, uint64_t size) {
    uint64_t h = orc_hash64(key, 0x9E3779B97F4A7C15ull * (level + 1));
    return (uint64_t)(((__uint128_t)h * size) >> 64);
}

static void mphf_build(mphf *m, const uint64_t *keys_in, uint64_t n, double gamma) {
    memset(m, 0, sizeof *m);
    m->n_keys = n;
    unsigned cap = 128;
    m->levels = (bitvec *)calloc(cap, sizeof(bitvec));
    m->level_rank0 = (uint64_t *)calloc(cap + 1, 8);
    uint64_t *keys = (uint64_t *)malloc((n ? n : 1) * 8);
    memcpy(keys, keys_in, n * 8);
    uint64_t remaining = n, placed = 0;
    while (remaining > 0) {
        if (m->n_levels == cap) { fprintf(stderr, "oracle mphf: too many levels\n"); abort(); }
        uint64_t size = (uint64_t)(gamma * (double)remaining) + 1;
        size = (size + 63) & ~63ull;
        bitvec occ, col;
        bv_init(&occ, size);
        bv_init(&col, size);
        {
            unsigned lvl = m->n_levels;
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)remaining; ++i) {
                uint64_t p = mphf_level_pos(keys[i], lvl, size), bit = 1ull << (p & 63);
                uint64_t old = __atomic_fetch_or(&occ.w[p >> 6], bit, __ATOMIC_RELAXED);
                if (old & bit) __atomic_fetch_or(&col.w[p >> 6], bit, __ATOMIC_RELAXED);
            }
        }
#pragma omp parallel for schedule(static)
        for (int64_t j = 0; j < (int64_t)occ.n_words; ++j) occ.w[j] &= ~col.w[j];
        /* keys whose bit did not survive go on to the next level, in their old order (chunks compacted in parallel) */
        uint64_t nr = 0;
        {
            int nt = 1;
#ifdef _OPENMP
            nt = omp_get_max_threads();
#endif
            uint64_t *cnt = (uint64_t *)calloc((size_t)nt + 1, 8);
            uint64_t *next = (uint64_t *)malloc((remaining ? remaining : 1) * 8);
            unsigned lvl = m->n_levels;
            uint64_t per = (remaining + (uint64_t)nt - 1) / (uint64_t)nt;
#pragma omp parallel for schedule(static, 1)
            for (int t = 0; t < nt; ++t) {
                uint64_t b = per * (uint64_t)t, e = b + per < remaining ? b + per : remaining, c = 0;
                for (uint64_t i = b; i < e; ++i) {
                    uint64_t p = mphf_level_pos(keys[i], lvl, size);
                    c += !((occ.w[p >> 6] >> (p & 63)) & 1);
                }
                cnt[t + 1] = c;
            }
            for (int t = 0; t < nt; ++t) cnt[t + 1] += cnt[t];
#pragma omp parallel for schedule(static, 1)
            for (int t = 0; t < nt; ++t) {
                uint64_t b = per * (uint64_t)t, e = b + per < remaining ? b + per : remaining, o = cnt[t];
                for (uint64_t i = b; i < e; ++i) {
                    uint64_t p = mphf_level_pos(keys[i], lvl, size);
                    if (!((occ.w[p >> 6] >> (p & 63)) & 1)) next[o++] = keys[i];
                }
            }
            nr = cnt[nt];
            free(cnt); free(keys); keys = next;
        }
        bv_free(&col);
        bv_enable_rank_select(&occ);
        m->levels[m->n_levels] = occ;
        m->level_rank0[m->n_levels] = placed;
        placed += occ.n_ones;
        m->n_levels++;
        remaining = nr;
    }
    m->level_rank0[m->n_levels] = placed;
    free(keys);
}
/* returns 1 and *out on Some, 0 on None (boomphf::Mphf::try_hash) */
static inline int mphf_try_hash(const mphf *m, uint64_t key, uint64_t *out) {
    for (unsigned lv = 0; lv < m->n_levels; ++lv) {
        const bitvec *b = &m->levels[lv];
        uint64_t p = mphf_level_pos(key, lv, b->n_bits);
        if (bv_get(b, p)) {
            *out = m->level_rank0[lv] + bv_rank(b, p);
            return 1;
        }
    }
    return 0;
}
static void mphf_free(mphf *m) {
    for (unsigned i = 0; i < m->n_levels; ++i) bv_free(&m->levels[i]);
    free(m->levels); free(m->level_rank0);
    memset(m, 0, sizeof *m);
}

/* ------------------------------------------------------------------ */
/* UnitigSet, unitig_set.rs:30-36 */
typedef struct {
    unsigned k;
    uint64_t *useq;      /* 2-bit packed concatenation */
    uint64_t total_len;
    uint64_t n_unitigs;
    efvec accum_lens;    /* n_unitigs+1 prefix sums */
    bitvec bv;           /* bit set at the LAST base of every unitig (unitig_set.rs:90-96) */
} unitig_set;

static inline uint64_t us_unitig_start(const unitig_set *u, uint64_t i) { return ef_get(&u->accum_lens, i); }       /* :196-199 */
static inline uint64_t us_unitig_end(const unitig_set *u, uint64_t i) { return ef_get(&u->accum_lens, i + 1); }     /* :201-204 */
static inline uint64_t us_unitig_len(const unitig_set *u, uint64_t i) { return us_unitig_end(u, i) - us_unitig_start(u, i); } /* :178-180 */
static inline uint64_t us_pos_to_id(const unitig_set *u, uint64_t pos) { return bv_rank(&u->bv, pos); }              /* :185-187 */
static inline uint64_t us_n_kmers(const unitig_set *u) { return u->total_len - (uint64_t)u->k * u->n_unitigs + u->n_unitigs; } /* :212-214 */
/* unitig_set.rs:235-245 */
static inline int us_is_valid_useq_pos(const unitig_set *u, uint64_t pos) {
    if (u->total_len < u->k) return 0;
    if (pos > u->total_len - u->k) return 0;
    for (unsigned j = 0; j + 1 < u->k; ++j)
        if (bv_get(&u->bv, pos + j)) return 0;
    return 1;
}

/* ------------------------------------------------------------------ */
/* SSHash, sshash.rs:39-57 */
typedef struct {
    uint64_t seed;
    unsigned w;
    unitig_set us;
    mphf mm_mphf;
    uint64_t n_minimizers;
    efvec occs_prefix_sum;
    intvec pos;
    uint64_t skew_param;   /* UINT64_MAX = no skew index */
    int has_skew;
    mphf skew_mphf;
    intvec skew_pos;
    /* brute-force ground truth */
    uint64_t *dict_words;
    uint64_t *dict_pos;
    uint64_t dict_n;
} orc_index;

typedef struct { uint64_t word, pos, seq; } mm_occ;

static int cmp_mm_occ(const void *a, const void *b) {
    const mm_occ *x = (const mm_occ *)a, *y = (const mm_occ *)b;
    if (x->word != y->word) return x->word < y->word ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable: par_sort_by_key keeps insertion order */
}
typedef struct { uint64_t word, pos; } wp_pair;
static int cmp_wp(const void *a, const void *b) {
    const wp_pair *x = (const wp_pair *)a, *y = (const wp_pair *)b;
    if (x->word != y->word) return x->word < y->word ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

void orc_free(orc_index *ix);

/* sshash.rs:86-329 (SSHashBuilder::from_unitig_set + finish) over unitig_set.rs:74-106 (from_seqs).
 * useq_words: 2-bit packed concatenation; starts: n_unitigs+1 prefix sums. */
orc_index *orc_build(const uint64_t *useq_words, uint64_t total_len, const uint64_t *starts,
                     uint64_t n_unitigs, uint32_t k, uint32_t w, uint64_t skew_param, uint64_t seed,
                     int with_dict) {
    if (k == 0 || k > 31 || w == 0 || w > k || n_unitigs == 0) return NULL; /* assert!(w <= k), sshash.rs:92 */
    orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
    ix->seed = seed; ix->w = w; ix->skew_param = skew_param;
    unitig_set *us = &ix->us;
    us->k = k; us->total_len = total_len; us->n_unitigs = n_unitigs;
    uint64_t nw = total_len / 32 + 2;
    us->useq = (uint64_t *)calloc(nw + 1, 8);
    memcpy(us->useq, useq_words, ((total_len + 31) / 32) * 8);
    if (ef_build(&us->accum_lens, starts, n_unitigs + 1, starts[n_unitigs]) != 0) { orc_free(ix); return NULL; }
    bv_init(&us->bv, total_len);
    for (uint64_t i = 1; i <= n_unitigs; ++i) bv_set(&us->bv, starts[i] - 1);
    bv_enable_rank_select(&us->bv);

    /* 1. collect minimizer occurrences, sshash.rs:97-143: per unitig, first the
     *    fw-canonical k-mers then the rc-canonical k-mers, each with its own
     *    consecutive-duplicate filter; pos = minimizer start on the fw strand.
     *    Unitigs are split into contiguous chunks over the OpenMP threads and the
     *    per-chunk lists concatenated in chunk order, so the global insertion
     *    order is the reference's (unitig order, fw pass before rc pass). */
    int n_chunks = 1;
#ifdef _OPENMP
    n_chunks = omp_get_max_threads();
#endif
    if ((uint64_t)n_chunks > n_unitigs) n_chunks = (int)n_unitigs;
    mm_occ **part = (mm_occ **)calloc((size_t)n_chunks, sizeof *part);
    uint64_t *part_n = (uint64_t *)calloc((size_t)n_chunks, 8);
#pragma omp parallel for schedule(static, 1)
    for (int c = 0; c < n_chunks; ++c) {
        /* chunk c = unitigs whose start lies in [total_len*c/n, total_len*(c+1)/n) */
        uint64_t lo_pos = total_len / (uint64_t)n_chunks * (uint64_t)c;
        uint64_t hi_pos = c + 1 == n_chunks ? total_len : total_len / (uint64_t)n_chunks * (uint64_t)(c + 1);
        uint64_t a = 0, b = n_unitigs; /* first unitig with start >= lo_pos */
        while (a < b) { uint64_t m = (a + b) / 2; if (starts[m] < lo_pos) a = m + 1; else b = m; }
        uint64_t u_lo = a;
        a = 0; b = n_unitigs;
        while (a < b) { uint64_t m = (a + b) / 2; if (starts[m] < hi_pos) a = m + 1; else b = m; }
        uint64_t u_hi = a;
        uint64_t cap = 1024, n = 0, tcap = 0;
        mm_occ *occ = (mm_occ *)malloc(cap * sizeof *occ);
        uint64_t *t_w = NULL, *t_p = NULL; uint8_t *t_f = NULL; /* per-unitig minimizer of every k-mer */
        uint64_t *h_f = NULL, *h_r = NULL;                      /* hash of every w-mer of the unitig, forward strand / reverse complement */
        const uint64_t wmask = w == 32 ? ~0ull : ((1ull << (2 * w)) - 1);
        for (uint64_t ui = u_lo; ui < u_hi; ++ui) {
            uint64_t s = starts[ui], e = starts[ui + 1];
            if (e - s < k) continue;
            uint64_t nk = e - s - k + 1, nwm = e - s - w + 1;
            if (nk > tcap) {
                tcap = nk * 2;
                t_w = (uint64_t *)realloc(t_w, tcap * 8); t_p = (uint64_t *)realloc(t_p, tcap * 8); t_f = (uint8_t *)realloc(t_f, tcap);
                h_f = (uint64_t *)realloc(h_f, (tcap + k) * 8); h_r = (uint64_t *)realloc(h_r, (tcap + k) * 8);
            }
            /* the sliding form of canonical_minimizer() (iter_canonical_minimizers, call site sshash.rs:104-134): every
             * w-mer is hashed once per strand; the k-mer at j then takes the leftmost minimum over the k-w+1 w-mers of its
             * canonical strand -- the same (word, offset) canonical_minimizer() returns for that k-mer */
            for (uint64_t j = 0; j < nwm; ++j) {
                uint64_t m = seq_get_kmer(us->useq, s + j, w);
                h_f[j] = orc_hash64(m, seed);
                h_r[j] = orc_hash64(kmer_rc(m, w), seed);
            }
            for (uint64_t j = 0; j < nk; ++j) {
                uint64_t fw = seq_get_kmer(us->useq, s + j, k), rc = kmer_rc(fw, k);
                int is_fw = fw <= rc;
                uint64_t g = is_fw ? fw : rc, best_h = 0;
                unsigned o_best = 0;
                for (unsigned o = 0; o + w <= k; ++o) { /* offset o inside g*; on the reverse strand it is forward w-mer j + (k-w-o) */
                    uint64_t h = is_fw ? h_f[j + o] : h_r[j + (k - w - o)];
                    if (o == 0 || h < best_h) { best_h = h; o_best = o; }
                }
                t_w[j] = (g >> (2 * o_best)) & wmask; t_f[j] = (uint8_t)is_fw;
                t_p[j] = is_fw ? s + j + o_best : s + j + (k - o_best - w);
            }
            for (int pass = 0; pass < 2; ++pass) {
                int have_prev = 0; uint64_t prev_w = 0, prev_p = 0;
                for (uint64_t j = 0; j < nk; ++j) {
                    if ((pass == 0) != (t_f[j] != 0)) continue;
                    if (!have_prev || prev_w != t_w[j] || prev_p != t_p[j]) {
                        if (n == cap) { cap *= 2; occ = (mm_occ *)realloc(occ, cap * sizeof *occ); }
                        occ[n].word = t_w[j]; occ[n].pos = t_p[j]; occ[n].seq = 0; ++n;
                    }
                    have_prev = 1; prev_w = t_w[j]; prev_p = t_p[j];
                }
            }
        }
        free(h_f); free(h_r);
        free(t_w); free(t_p); free(t_f);
        part[c] = occ; part_n[c] = n;
    }
    uint64_t n_occ = 0;
    for (int c = 0; c < n_chunks; ++c) n_occ += part_n[c];
    if (n_occ == 0) { for (int c = 0; c < n_chunks; ++c) free(part[c]); free(part); free(part_n); orc_free(ix); return NULL; }
    mm_occ *occs = (mm_occ *)malloc(n_occ * sizeof *occs);
    {
        uint64_t off = 0;
        for (int c = 0; c < n_chunks; ++c) {
            memcpy(occs + off, part[c], part_n[c] * sizeof *occs);
            off += part_n[c];
            free(part[c]);
        }
        free(part); free(part_n);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n_occ; ++i) occs[i].seq = (uint64_t)i;
    }
    /* 2. stable sort by w-mer word (par_sort_by_key, sshash.rs:150): partition on the top bits of
     *    the word, then sort every partition by (word, insertion index) */
    {
        unsigned wb = 2 * w, B = wb < 10 ? wb : 10, shift = wb - B;
        uint64_t nbk = 1ull << B;
        uint64_t *bstart = (uint64_t *)calloc(nbk + 1, 8);
        int nt = 1;
#ifdef _OPENMP
        nt = omp_get_max_threads();
#endif
        /* counting sort on the top bits, chunks of the input handled in parallel and laid down in chunk order (stable) */
        uint64_t *hist = (uint64_t *)calloc((size_t)nt * nbk, 8);
        uint64_t per = (n_occ + (uint64_t)nt - 1) / (uint64_t)nt;
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nt; ++t) {
            uint64_t b = per * (uint64_t)t, e = b + per < n_occ ? b + per : n_occ;
            for (uint64_t i = b; i < e; ++i) hist[(uint64_t)t * nbk + (occs[i].word >> shift)]++;
        }
        {
            uint64_t acc = 0;
            for (uint64_t b = 0; b < nbk; ++b) {
                bstart[b] = acc;
                for (int t = 0; t < nt; ++t) { uint64_t c = hist[(uint64_t)t * nbk + b]; hist[(uint64_t)t * nbk + b] = acc; acc += c; }
            }
            bstart[nbk] = acc;
        }
        mm_occ *tmp = (mm_occ *)malloc(n_occ * sizeof *tmp);
#pragma omp parallel for schedule(static, 1)
        for (int t = 0; t < nt; ++t) {
            uint64_t b = per * (uint64_t)t, e = b + per < n_occ ? b + per : n_occ;
            uint64_t *cur = hist + (uint64_t)t * nbk;
            for (uint64_t i = b; i < e; ++i) tmp[cur[occs[i].word >> shift]++] = occs[i];
        }
        free(hist); free(occs); occs = tmp;
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)nbk; ++b)
            qsort(occs + bstart[b], bstart[b + 1] - bstart[b], sizeof *occs, cmp_mm_occ);
        free(bstart);
    }
    /* run-length -> mm_set / mm_occs, sshash.rs:153-172 */
    uint64_t n_mm = 0;
    for (uint64_t i = 0; i < n_occ; ++i) if (i == 0 || occs[i].word != occs[i - 1].word) ++n_mm;
    uint64_t *mm_set = (uint64_t *)malloc(n_mm * 8), *mm_cnt = (uint64_t *)calloc(n_mm, 8);
    uint64_t *ranges = (uint64_t *)malloc((n_mm + 1) * 8);
    {
        uint64_t j = 0;
        for (uint64_t i = 0; i < n_occ; ++i) {
            if (i == 0 || occs[i].word != occs[i - 1].word) { mm_set[j] = occs[i].word; ranges[j] = i; ++j; }
            mm_cnt[j - 1]++;
        }
        ranges[n_mm] = n_occ;
    }
    ix->n_minimizers = n_mm;
    /* 3. MPHF over the minimizer set, gamma 1.7, sshash.rs:177 */
    mphf_build(&ix->mm_mphf, mm_set, n_mm, 1.7);
    /* 4. bucket sizes by hash value -> prefix sums, sshash.rs:181-189 (util.rs:26-38) */
    uint64_t *n_by_h = (uint64_t *)calloc(n_mm, 8), *psum = (uint64_t *)malloc((n_mm + 1) * 8);
    uint64_t *slot = (uint64_t *)malloc(n_mm * 8);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n_mm; ++i) {
        uint64_t h = 0; mphf_try_hash(&ix->mm_mphf, mm_set[i], &h);
        slot[i] = h;
        n_by_h[h] = mm_cnt[i];
    }
    { uint64_t acc = 0; for (uint64_t i = 0; i < n_mm; ++i) { psum[i] = acc; acc += n_by_h[i]; } psum[n_mm] = acc; }
    /* 5. scatter positions, sshash.rs:196-219 */
    uint64_t *pos = (uint64_t *)malloc(n_occ * 8), max_pos = 0;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n_mm; ++i) {
        uint64_t sh = psum[slot[i]];
        for (uint64_t j = ranges[i]; j < ranges[i + 1]; ++j) pos[sh + (j - ranges[i])] = occs[j].pos;
    }
    for (uint64_t i = 0; i < n_occ; ++i) if (pos[i] > max_pos) max_pos = pos[i];
    free(slot);
    /* 6. skew index, sshash.rs:222-296 */
    if (skew_param != UINT64_MAX) {
        int nt = 1;
#ifdef _OPENMP
        nt = omp_get_max_threads();
#endif
        wp_pair **sp = (wp_pair **)calloc((size_t)nt, sizeof *sp);
        uint64_t *sp_n = (uint64_t *)calloc((size_t)nt, 8), *sp_cap = (uint64_t *)calloc((size_t)nt, 8);
#pragma omp parallel
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n_mm; ++i) {
                if (mm_cnt[i] <= skew_param) continue;
                for (uint64_t j = ranges[i]; j < ranges[i + 1]; ++j) {
                    uint64_t mp = occs[j].pos;
                    uint64_t start_pos = mp < (uint64_t)(k - w) ? 0 : mp - (k - w);
                    for (uint64_t off = 0; off < (uint64_t)(k - w + 1); ++off) {
                        uint64_t p = start_pos + off;
                        if (us_is_valid_useq_pos(us, p)) {
                            uint64_t x = seq_get_kmer(us->useq, p, k), r = kmer_rc(x, k);
                            if (sp_n[t] == sp_cap[t]) { sp_cap[t] = sp_cap[t] ? sp_cap[t] * 2 : 1024; sp[t] = (wp_pair *)realloc(sp[t], sp_cap[t] * sizeof(wp_pair)); }
                            sp[t][sp_n[t]].word = x < r ? x : r; sp[t][sp_n[t]].pos = p; ++sp_n[t];
                        }
                    }
                }
            }
        }
        uint64_t sn = 0;
        for (int t = 0; t < nt; ++t) sn += sp_n[t];
        wp_pair *tup = (wp_pair *)malloc((sn ? sn : 1) * sizeof *tup);
        { uint64_t off = 0; for (int t = 0; t < nt; ++t) { if (sp_n[t]) memcpy(tup + off, sp[t], sp_n[t] * sizeof *tup); off += sp_n[t]; free(sp[t]); } }
        free(sp); free(sp_n); free(sp_cap);
        { /* sort by (word, pos): partition on the top bits of the 2k-bit word, sort partitions */
            unsigned wb = 2 * k, B = wb < 10 ? wb : 10, shift = wb - B;
            uint64_t nbk = 1ull << B;
            uint64_t *bstart = (uint64_t *)calloc(nbk + 1, 8);
            for (uint64_t i = 0; i < sn; ++i) bstart[(tup[i].word >> shift) + 1]++;
            for (uint64_t b = 0; b < nbk; ++b) bstart[b + 1] += bstart[b];
            uint64_t *cur = (uint64_t *)malloc(nbk * 8);
            memcpy(cur, bstart, nbk * 8);
            wp_pair *tmp = (wp_pair *)malloc((sn ? sn : 1) * sizeof *tmp);
            for (uint64_t i = 0; i < sn; ++i) tmp[cur[tup[i].word >> shift]++] = tup[i];
            free(cur); free(tup); tup = tmp;
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t b = 0; b < (int64_t)nbk; ++b)
                qsort(tup + bstart[b], bstart[b + 1] - bstart[b], sizeof *tup, cmp_wp);
            free(bstart);
        }
        uint64_t un = 0;
        for (uint64_t i = 0; i < sn; ++i) if (i == 0 || tup[i].word != tup[un - 1].word) tup[un++] = tup[i];
        uint64_t *km_set = (uint64_t *)malloc((un ? un : 1) * 8), smax = 0;
        for (uint64_t i = 0; i < un; ++i) { km_set[i] = tup[i].word; if (tup[i].pos > smax) smax = tup[i].pos; }
        mphf_build(&ix->skew_mphf, km_set, un, 1.7);
        iv_init(&ix->skew_pos, un, bit_len(smax));
        {
            uint64_t *hs = (uint64_t *)malloc((un ? un : 1) * 8);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)un; ++i) { uint64_t h = 0; mphf_try_hash(&ix->skew_mphf, km_set[i], &h); hs[i] = h; }
            for (uint64_t i = 0; i < un; ++i) iv_set(&ix->skew_pos, hs[i], tup[i].pos);
            free(hs);
        }
        ix->has_skew = 1;
        free(km_set); free(tup);
    }
    /* 7. finish, sshash.rs:310-329: EF-encode prefix sums, bit-pack pos */
    ef_build(&ix->occs_prefix_sum, psum, n_mm + 1, psum[n_mm]);
    iv_init(&ix->pos, n_occ, bit_len(max_pos));
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < (int64_t)((n_occ + 63) / 64); ++g) /* 64 elements of any width end on a word boundary: groups never share a word */
        for (uint64_t i = (uint64_t)g * 64; i < n_occ && i < ((uint64_t)g + 1) * 64; ++i) iv_set(&ix->pos, i, pos[i]);
    free(pos); free(psum); free(n_by_h); free(ranges); free(mm_cnt); free(mm_set); free(occs);

    /* brute-force dictionary: every valid k-mer position, keyed by canonical word */
    if (with_dict) {
        /* counted, not us_n_kmers(): the reference formula (unitig_set.rs:212-214) assumes every unitig holds >= k bases */
        uint64_t nk = 0, j = 0;
        for (uint64_t ui = 0; ui < n_unitigs; ++ui) if (starts[ui + 1] - starts[ui] >= k) nk += starts[ui + 1] - starts[ui] - k + 1;
        wp_pair *d = (wp_pair *)malloc((nk ? nk : 1) * sizeof *d);
        for (uint64_t ui = 0; ui < n_unitigs; ++ui)
            for (uint64_t p = starts[ui]; p + k <= starts[ui + 1]; ++p) {
                uint64_t x = seq_get_kmer(us->useq, p, k), r = kmer_rc(x, k);
                d[j].word = x < r ? x : r; d[j].pos = p; ++j;
            }
        qsort(d, j, sizeof *d, cmp_wp);
        ix->dict_n = j;
        ix->dict_words = (uint64_t *)malloc((j ? j : 1) * 8);
        ix->dict_pos = (uint64_t *)malloc((j ? j : 1) * 8);
        for (uint64_t i = 0; i < j; ++i) { ix->dict_words[i] = d[i].word; ix->dict_pos[i] = d[i].pos; }
        free(d);
    }
    return ix;
}

void orc_free(orc_index *ix) {
    if (!ix) return;
    free(ix->us.useq); ef_free(&ix->us.accum_lens); bv_free(&ix->us.bv);
    mphf_free(&ix->mm_mphf); ef_free(&ix->occs_prefix_sum); iv_free(&ix->pos);
    if (ix->has_skew) { mphf_free(&ix->skew_mphf); iv_free(&ix->skew_pos); }
    free(ix->dict_words); free(ix->dict_pos);
    free(ix);
}

/* CanonicalKmer::get_word_equivalency (call sites sshash.rs:503,532): Identity tested first */
static inline uint32_t word_equivalency(uint64_t fw, uint64_t rc, uint64_t kw) {
    return kw == fw ? ORC_IDENTITY : (kw == rc ? ORC_TWIN : ORC_NONE);
}

static inline int fill_hit(const orc_index *ix, uint64_t km_pos, uint32_t mt, int check_end, orc_k2upos *out) {
    const unitig_set *us = &ix->us;
    uint64_t uid = us_pos_to_id(us, km_pos);
    uint64_t ulen = us_unitig_len(us, uid);
    uint64_t p = km_pos - us_unitig_start(us, uid);
    if (check_end && km_pos + us->k > us_unitig_end(us, uid)) return 0; /* sshash.rs:509-514 */
    out->unitig_id = (uint32_t)uid; out->unitig_len = (uint32_t)ulen; out->pos = (uint32_t)p; out->match = mt;
    return 1;
}

/* sshash.rs:415-433 + :59-66 */
static int k2u_skew_index(const orc_index *ix, uint64_t fw, uint64_t rc, orc_k2upos *out) {
    if (!ix->has_skew) return 0;
    uint64_t word = fw < rc ? fw : rc, h;
    if (!mphf_try_hash(&ix->skew_mphf, word, &h)) return 0;
    uint64_t pos = iv_get(&ix->skew_pos, h);
    uint64_t kw = seq_get_kmer(ix->us.useq, pos, ix->us.k);
    uint32_t mt = word_equivalency(fw, rc, kw);
    if (mt == ORC_NONE) return 0;
    return fill_hit(ix, pos, mt, 0, out);
}

/* sshash.rs:471-554  K2U::k2u.  fw = forward word of the queried CanonicalKmer. */
int orc_k2u(const orc_index *ix, uint64_t fw, orc_k2upos *out) {
    const unitig_set *us = &ix->us;
    unsigned k = us->k, w = ix->w;
    out->unitig_id = out->pos = out->unitig_len = 0; out->match = ORC_NONE;
    uint64_t rc = kmer_rc(fw, k);
    uint64_t mmer; unsigned offset; int is_fw;
    canonical_minimizer(fw, k, w, ix->seed, &mmer, &offset, &is_fw);   /* :476 */
    uint64_t h;
    if (!mphf_try_hash(&ix->mm_mphf, mmer, &h)) return 0;              /* :478-479 */
    uint64_t pos_start = ef_get(&ix->occs_prefix_sum, h);              /* :482 */
    uint64_t pos_end = ef_get(&ix->occs_prefix_sum, h + 1);            /* :483 */
    uint64_t n_occs = pos_end - pos_start;
    if (n_occs > ix->skew_param) return k2u_skew_index(ix, fw, rc, out); /* :486-490 */
    uint64_t last_km_start_pos = us->total_len - k;                    /* :492 */
    uint64_t rc_offset = k - offset - w;                               /* :493 */
    for (uint64_t pi = pos_start; pi < pos_end; ++pi) {                /* :494 */
        uint64_t mm_pos = iv_get(&ix->pos, pi);
        if (mm_pos >= offset && mm_pos - offset <= last_km_start_pos) { /* :498 */
            uint64_t km_pos = mm_pos - offset;
            uint32_t mt = word_equivalency(fw, rc, seq_get_kmer(us->useq, km_pos, k));
            if (mt != ORC_NONE && fill_hit(ix, km_pos, mt, 1, out)) return 1;
        }
        if (mm_pos >= rc_offset && mm_pos - rc_offset <= last_km_start_pos) { /* :527 */
            uint64_t km_pos = mm_pos - rc_offset;
            uint32_t mt = word_equivalency(fw, rc, seq_get_kmer(us->useq, km_pos, k));
            if (mt != ORC_NONE && fill_hit(ix, km_pos, mt, 1, out)) return 1;
        }
    }
    out->unitig_id = out->pos = out->unitig_len = 0; out->match = ORC_NONE;
    return 0;
}

/* brute-force ground truth: first dictionary entry for the canonical word */
int orc_dict_k2u(const orc_index *ix, uint64_t fw, orc_k2upos *out) {
    unsigned k = ix->us.k;
    out->unitig_id = out->pos = out->unitig_len = 0; out->match = ORC_NONE;
    uint64_t rc = kmer_rc(fw, k), word = fw < rc ? fw : rc;
    uint64_t lo = 0, hi = ix->dict_n;
    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (ix->dict_words[mid] < word) lo = mid + 1; else hi = mid; }
    if (lo == ix->dict_n || ix->dict_words[lo] != word) return 0;
    uint64_t p = ix->dict_pos[lo];
    uint64_t kw = seq_get_kmer(ix->us.useq, p, k);
    /* unitig id by linear-free route: binary search over EF starts (independent of bv rank) */
    uint64_t a = 0, b = ix->us.n_unitigs;
    while (b - a > 1) { uint64_t m = (a + b) / 2; if (us_unitig_start(&ix->us, m) <= p) a = m; else b = m; }
    out->unitig_id = (uint32_t)a;
    out->unitig_len = (uint32_t)us_unitig_len(&ix->us, a);
    out->pos = (uint32_t)(p - us_unitig_start(&ix->us, a));
    out->match = kw == fw ? ORC_IDENTITY : ORC_TWIN;
    return 1;
}
/* number of canonical words that occur more than once (0 for a cdBG unitig set) */
uint64_t orc_dict_n_duplicates(const orc_index *ix) {
    uint64_t d = 0;
    for (uint64_t i = 1; i < ix->dict_n; ++i) d += ix->dict_words[i] == ix->dict_words[i - 1];
    return d;
}

/* batch of independent point queries (validate_self_parallel-style sharding, kphf/mod.rs:105-139) */
void orc_k2u_batch(const orc_index *ix, const uint64_t *fw_words, uint64_t n, orc_k2upos *out,
                   uint64_t *n_hit, int n_threads, int use_dict) {
    uint64_t hits = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(static) reduction(+ : hits)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
        hits += (uint64_t)(use_dict ? orc_dict_k2u(ix, fw_words[i], &out[i]) : orc_k2u(ix, fw_words[i], &out[i]));
    if (n_hit) *n_hit = hits;
}

/* index/caching.rs:12-103 StreamingK2U */
typedef struct { int is_warm; orc_k2upos prev; } orc_stream;
static uint64_t g_reads_cold; /* cold lookups of the last orc_k2u_reads call (single-threaded walk) */

static int k2u_cold(const orc_index *ix, orc_stream *st, uint64_t fw, orc_k2upos *out) { /* :99-103 */
    orc_k2upos r;
    ++g_reads_cold;
    if (!orc_k2u(ix, fw, &r)) { *out = r; return 0; } /* `?` : state untouched on a miss */
    st->prev = r; st->is_warm = 1; *out = r;
    return 1;
}
static int k2u_streaming(const orc_index *ix, orc_stream *st, uint64_t fw, orc_k2upos *out) { /* :65-97 */
    unsigned k = ix->us.k;
    if (!st->is_warm) return k2u_cold(ix, st, fw, out);
    uint64_t next_pos = (uint64_t)st->prev.pos + 1;
    uint64_t last_km_pos = (uint64_t)st->prev.unitig_len - k;
    if (next_pos > last_km_pos) return k2u_cold(ix, st, fw, out);
    uint64_t ustart = us_unitig_start(&ix->us, st->prev.unitig_id);
    uint64_t kw = seq_get_kmer(ix->us.useq, ustart + next_pos, k);
    uint32_t mt = word_equivalency(fw, kmer_rc(fw, k), kw);
    if (mt == ORC_NONE) return k2u_cold(ix, st, fw, out);
    st->prev.pos = (uint32_t)next_pos; st->prev.match = mt; *out = st->prev;
    return 1;
}

/* reads: the bench loop of bin/kphf/main.rs:273-339 (CanonicalKmerIterator over each
 * record; windows holding a non-ACGT byte are skipped).  Window j of read r is written
 * to out[kmer_offsets[r] + j]; skipped windows get match = ORC_SKIPPED.  kmer_offsets[r]
 * must be sum over earlier reads of max(0, len - k + 1).  streaming != 0 uses
 * k2u_streaming with one state per read batch exactly like the reference (state is
 * carried ACROSS reads, main.rs:301-312). */
void orc_k2u_reads(const orc_index *ix, const uint8_t *bases, const uint64_t *read_offsets, uint64_t n_reads,
                   const uint64_t *kmer_offsets, int streaming, orc_k2upos *out, uint64_t *counts /* hit, miss, skipped */) {
    unsigned k = ix->us.k;
    uint64_t mask = (1ull << (2 * k)) - 1, n_hit = 0, n_miss = 0, n_skip = 0;
    orc_stream st; memset(&st, 0, sizeof st);
    g_reads_cold = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        uint64_t s = read_offsets[r], e = read_offsets[r + 1];
        uint64_t fw = 0; unsigned valid = 0;
        for (uint64_t i = s; i < e; ++i) {
            int c = base_code(bases[i]);
            if (c < 0) { valid = 0; fw = 0; } else { fw = (fw >> 2) | ((uint64_t)c << (2 * (k - 1))); fw &= mask; ++valid; }
            if (i + 1 >= s + k) {
                orc_k2upos *o = &out[kmer_offsets[r] + (i + 1 - k - s)];
                if (valid >= k) {
                    int hit = streaming ? k2u_streaming(ix, &st, fw, o) : orc_k2u(ix, fw, o);
                    if (!streaming) ++g_reads_cold;
                    if (hit) ++n_hit; else ++n_miss;
                } else { o->unitig_id = o->pos = o->unitig_len = 0; o->match = ORC_SKIPPED; ++n_skip; }
            }
        }
    }
    if (counts) { counts[0] = n_hit; counts[1] = n_miss; counts[2] = n_skip; }
}

/* windows of the last orc_k2u_reads call that went through the whole k2u chain (k2u_cold, caching.rs:99-103; every window
 * when streaming == 0); the others were answered by the warm step from the previous position */
uint64_t orc_reads_cold_lookups(void) { return g_reads_cold; }

/* kphf/mod.rs:69-103 K2U::validate_self: every k-mer of every unitig, both orientations.
 * Returns the number of failed assertions (0 == the reference's test passes). */
uint64_t orc_validate_self(const orc_index *ix, int use_dict) {
    const unitig_set *us = &ix->us;
    unsigned k = us->k;
    uint64_t fails = 0;
    for (uint64_t ui = 0; ui < us->n_unitigs; ++ui) {
        uint64_t s = us_unitig_start(us, ui), ulen = us_unitig_len(us, ui);
        for (uint64_t p = 0; p + k <= ulen; ++p) {
            uint64_t fw = seq_get_kmer(us->useq, s + p, k);
            orc_k2upos r;
            int ok = use_dict ? orc_dict_k2u(ix, fw, &r) : orc_k2u(ix, fw, &r);
            if (!ok || r.unitig_id != ui || r.unitig_len != ulen || r.pos != p || r.match != ORC_IDENTITY) ++fails;
            ok = use_dict ? orc_dict_k2u(ix, kmer_rc(fw, k), &r) : orc_k2u(ix, kmer_rc(fw, k), &r); /* km.swap() */
            if (!ok || r.unitig_id != ui || r.unitig_len != ulen || r.pos != p || r.match != ORC_TWIN) ++fails;
        }
    }
    return fails;
}

/* mean 8-byte words touched per query by orc_k2u over a batch (single thread; +1 word in, +2 words out per query) */
double orc_word_touches(const orc_index *ix, const uint64_t *fw_words, uint64_t n, uint64_t *n_hit) {
    orc_k2upos r;
    uint64_t hits = 0;
    g_touch = 0;
    g_touch_on = 1;
    for (uint64_t i = 0; i < n; ++i) hits += (uint64_t)orc_k2u(ix, fw_words[i], &r);
    g_touch_on = 0;
    if (n_hit) *n_hit = hits;
    return n ? (double)(g_touch + 3 * n) / (double)n : 0.0;
}

/* ---- accessors for the tests ------------------------------------- */
uint32_t orc_k(const orc_index *ix) { return ix->us.k; }
uint64_t orc_n_unitigs(const orc_index *ix) { return ix->us.n_unitigs; }
uint64_t orc_n_kmers(const orc_index *ix) { return us_n_kmers(&ix->us); }
uint64_t orc_sum_unitigs_len(const orc_index *ix) { return ix->us.total_len; }
uint64_t orc_unitig_len(const orc_index *ix, uint64_t i) { return us_unitig_len(&ix->us, i); }
uint64_t orc_unitig_start(const orc_index *ix, uint64_t i) { return us_unitig_start(&ix->us, i); }
uint64_t orc_pos_to_id(const orc_index *ix, uint64_t pos) { return us_pos_to_id(&ix->us, pos); }
uint64_t orc_n_minimizers(const orc_index *ix) { return ix->n_minimizers; }
uint64_t orc_n_minimizer_occs(const orc_index *ix) { return ix->pos.n; }
uint64_t orc_n_kmers_in_skew_index(const orc_index *ix) { return ix->has_skew ? ix->skew_pos.n : 0; }
uint64_t orc_get_kmer(const orc_index *ix, uint64_t pos) { return seq_get_kmer(ix->us.useq, pos, ix->us.k); }
uint64_t orc_kmer_rc(uint64_t x, uint32_t k) { return kmer_rc(x, k); }
uint64_t orc_util_msb(uint64_t n) { return orc_msb(n); }

/* Elias-Fano known-answer hook (elias_fano.rs:148-164): encode xs, decode into out; returns 0, -1 (not monotone), -2 (empty) */
int orc_ef_roundtrip(const uint64_t *xs, uint64_t n, uint64_t u, uint64_t *out) {
    efvec e;
    int rc = ef_build(&e, xs, n, u);
    if (rc == 0) for (uint64_t i = 0; i < n; ++i) out[i] = ef_get(&e, i);
    if (rc != -2) ef_free(&e);
    return rc;
}
/* (the average number of 8-byte words a query touches is measured by the word-touch counter above, not modelled here) */
