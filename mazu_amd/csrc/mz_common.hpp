// mz_common.hpp -- arithmetic shared by the host builder and the gfx950 kernels.
// Everything here is pure 64-bit integer work; the same inline functions are compiled
// for the host (index construction) and for the device (queries) so both sides agree
// on every hash and bit position by construction.
#pragma once
#include <cstdint>

#include "../../include/mazu_amd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MZ_HD __host__ __device__ __forceinline__
#else
#define MZ_HD inline
#endif

namespace mz {

// ---- layout constants -------------------------------------------------------------
// Rank-interleaved MPHF block (one 64-byte line):  w0 = ones before this block (over the
// whole MPHF), w1 = six 9-bit cumulative counts (ones before data word j), w2..w7 = data.
constexpr uint32_t kMphfDataWords = 6;
constexpr uint32_t kMphfBitsPerBlock = 64 * kMphfDataWords; // 384
constexpr uint32_t kMphfMaxLevels = 64;
// Boundary block (one 64-byte line) covering 256 bases of the concatenated unitigs:
// w0 = id of the unitig holding the block's first base, w1 = start of that unitig,
// w2 = end (exclusive) of the unitig holding the block's last base, w3 = 0,
// w4..w7 = bits, one per base, set at the LAST base of every unitig (unitig_set.rs:90-96).
constexpr uint32_t kBvBasesPerBlockLog2 = 8;
constexpr uint32_t kBvBasesPerBlock = 1u << kBvBasesPerBlockLog2;
// Elias-Fano select sample rate: position of every 2^kEfSelLog2-th one is stored.
constexpr uint32_t kEfSelLog2 = 4;

// ---- 64x64 -> 128 multiply fold ("mum"), the wyhash primitive -----------------------
MZ_HD uint64_t mum(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (a * b) ^ __umul64hi(a, b);
#else
    __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
#endif
}

constexpr uint64_t kP0 = 0xa0761d6478bd642full;
constexpr uint64_t kP1 = 0xe7037ed1a0b428dbull;
constexpr uint64_t kP2 = 0x8ebc6af09c88c6e3ull;
constexpr uint64_t kP3 = 0x589965cc75374cc3ull;

// Minimizer order: WyHashState(seed) applied to the w-mer word (kphf/mod.rs:32-52).  A
// wyhash-style two-round mum of the 8-byte key; bit-compat with the wyhash crate is not
// needed for result parity (a unique k-mer has one position whatever the minimizer order) and is unpinned.
// For w <= 16 the word fits 32 bits and the 32-bit member of the wyhash family (two 32x32->64
// multiply-folds, cf. wyhash32) is used, cut to its top kMmerHash32Bits bits: the order is "smallest hash, leftmost
// w-mer on ties", so a window minimum can be taken over single 32-bit keys (hash << position bits | position) with one
// v_min_u32 per element -- (hash, offset in the k-mer) for a point query, (hash, position in the tile) for the sliding
// minimum over a staged read tile.
constexpr uint32_t kMmerHash32Bits = 21;
MZ_HD uint64_t hash_mmer64(uint64_t x, uint64_t seed) { return mum(mum(x ^ kP0, seed ^ kP1) ^ kP2, kP3); }
MZ_HD uint32_t hash_mmer32(uint32_t x, uint64_t seed) {
    const uint32_t c0 = 0x53c5ca59u, c1 = 0x74743c1bu;
    uint64_t c = (uint64_t)(x ^ c0 ^ (uint32_t)seed) * (uint64_t)((c1 ^ (uint32_t)(seed >> 32)) | 1u);
    uint32_t a = (uint32_t)c ^ c0, b = (uint32_t)(c >> 32) ^ c1;
    c = (uint64_t)a * b;
    return ((uint32_t)c ^ (uint32_t)(c >> 32)) >> (32 - kMmerHash32Bits);
}
MZ_HD uint64_t hash_mmer(uint64_t x, uint64_t seed, uint32_t w) { return w <= 16 ? hash_mmer32((uint32_t)x, seed) : hash_mmer64(x, seed); }

// per-level MPHF hash: one multiply-fold of the seeded key
MZ_HD uint64_t hash_level(uint64_t key, uint64_t level_seed) {
    uint64_t h = mum(key ^ level_seed, kP1);
    return h ^ (h >> 31);
}

MZ_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// position of a key inside one MPHF level of n_blocks blocks: block chosen by the high half of
// the hash, bit inside the block by the low half (both by multiply-shift, no division)
MZ_HD void mphf_level_slot(uint64_t key, uint64_t level_seed, uint32_t n_blocks, uint32_t &blk, uint32_t &bit) {
    uint64_t h = hash_level(key, level_seed);
    blk = mulhi32((uint32_t)(h >> 32), n_blocks);
    bit = mulhi32((uint32_t)h, kMphfBitsPerBlock);
}

// ---- 2-bit k-mer words (A=0 C=1 G=2 T=3, first base in the LSBs) ---------------------
MZ_HD uint64_t kmer_mask(uint32_t k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }

MZ_HD uint64_t kmer_rc(uint64_t x, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    // reversing all 64 bits reverses the bases and swaps the two bits of every base: swap them back, complement, align
    uint32_t lo = __brev((uint32_t)(x >> 32)), hi = __brev((uint32_t)x);
    lo = ~(((lo >> 1) & 0x55555555u) | ((lo << 1) & 0xAAAAAAAAu));
    hi = ~(((hi >> 1) & 0x55555555u) | ((hi << 1) & 0xAAAAAAAAu));
    return (((uint64_t)hi << 32) | lo) >> (64 - 2 * k);
#else
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
#endif
}

// k-mer starting at base `pos` of a packed sequence (two-word read; the array is padded)
MZ_HD uint64_t read_kmer(const uint64_t *words, uint64_t pos, uint32_t k) {
    uint64_t b = pos * 2, wi = b >> 6;
    uint32_t sh = (uint32_t)(b & 63);
    uint64_t lo = words[wi], hi = words[wi + 1];
    uint64_t x = (lo >> sh) | ((hi << 1) << (63 - sh)); // branch-free funnel shift (sh == 0: second term vanishes): both loads stay unconditional
    return x & kmer_mask(k);
}

// fixed-width packed integer (IntVector::get): element i at bit i*width, array padded
MZ_HD uint64_t read_packed(const uint64_t *words, uint64_t i, uint32_t width) {
    uint64_t b = i * width, wi = b >> 6;
    uint32_t sh = (uint32_t)(b & 63);
    uint64_t lo = words[wi], hi = words[wi + 1];
    uint64_t x = (lo >> sh) | ((hi << 1) << (63 - sh)); // branch-free funnel shift (sh == 0: second term vanishes): both loads stay unconditional
    return width >= 64 ? x : (x & ((1ull << width) - 1));
}

MZ_HD uint32_t popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}
MZ_HD uint32_t ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffsll((long long)x) - 1);
#else
    return (uint32_t)__builtin_ctzll(x);
#endif
}
MZ_HD uint32_t clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clzll((long long)x);
#else
    return (uint32_t)__builtin_clzll(x);
#endif
}

// position of the r-th (0-based) set bit of x; requires r < popcount(x)
MZ_HD uint32_t select_in_word(uint64_t x, uint32_t r) {
    uint32_t pos = 0;
    uint32_t lo = (uint32_t)x;
    uint32_t c = (uint32_t)__builtin_popcount(lo);
    if (r >= c) { r -= c; lo = (uint32_t)(x >> 32); pos = 32; }
    c = (uint32_t)__builtin_popcount(lo & 0xFFFFu);
    if (r >= c) { r -= c; lo >>= 16; pos += 16; }
    c = (uint32_t)__builtin_popcount(lo & 0xFFu);
    if (r >= c) { r -= c; lo >>= 8; pos += 8; }
    c = (uint32_t)__builtin_popcount(lo & 0xFu);
    if (r >= c) { r -= c; lo >>= 4; pos += 4; }
    c = (uint32_t)__builtin_popcount(lo & 0x3u);
    if (r >= c) { r -= c; lo >>= 2; pos += 2; }
    if (r >= (lo & 1u)) pos += 1;
    return pos;
}

// util.rs:48-55
MZ_HD uint64_t msb(uint64_t n) { return n == 0 ? 0 : 63 - clz64(n); }
MZ_HD uint32_t bit_len(uint64_t v) { return v == 0 ? 1u : 64u - clz64(v); }

// Canonical minimizer of a k-mer (call site sshash.rs:476; definition sshash.rs:32-37):
// the w-mer of g* = min(fw, rc) with the smallest hash, leftmost on ties.
struct Minimizer {
    uint64_t word;
    uint32_t offset; // inside g*
};
MZ_HD Minimizer canonical_minimizer(uint64_t fw, uint64_t rc, uint32_t k, uint32_t w, uint64_t seed) {
    uint64_t g = fw <= rc ? fw : rc;
    uint64_t wmask = kmer_mask(w);
    if (w <= 16) { // 21-bit hashes: the minimum of (hash << 5 | offset) is the smallest hash at its leftmost offset (k - w <= 30)
        uint32_t best = ~0u;
        for (uint32_t o = 0; o + w <= k; ++o) {
            const uint32_t key = (hash_mmer32((uint32_t)((g >> (2 * o)) & wmask), seed) << 5) | o;
            best = key < best ? key : best;
        }
        const uint32_t o = best & 31u;
        return Minimizer{(g >> (2 * o)) & wmask, o};
    }
    Minimizer best{g & wmask, 0};
    uint64_t best_h = hash_mmer64(best.word, seed);
    for (uint32_t o = 1; o + w <= k; ++o) {
        uint64_t m = (g >> (2 * o)) & wmask;
        uint64_t h = hash_mmer64(m, seed);
        if (h < best_h) { best_h = h; best.word = m; best.offset = o; }
    }
    return best;
}

// ---- device-visible index description (plain pointers; filled by the host) -----------
struct MphfLevel {
    uint64_t block_off; // first block of this level
    uint64_t seed;
    uint32_t n_blocks;
    uint32_t pad;
};

struct MphfView {
    const uint64_t *blocks; // n_total_blocks * 8 words
    const MphfLevel *levels;
    uint32_t n_levels;
    uint32_t direct;        // != 0: `blocks` is a direct-address rank bitmap over the key universe (see direct_lookup)
    uint64_t n_keys;
};

struct EfView {
    const uint64_t *high; // unary high parts, padded with zero words
    const uint64_t *sel;  // bit position of every 2^kEfSelLog2-th one
    const uint64_t *low;  // l-bit low parts, packed
    uint64_t n;           // number of elements
    uint64_t n_high_words;
    uint32_t l;
    uint32_t pad;
};


// Partitioned Elias-Fano for the bucket prefix sums (the reference's occs_prefix_sum EFVector,
// sshash.rs:46, elias_fano.rs) re-blocked so that get(i) and get(i+1) cost ONE 64-byte line:
// block j holds elements [64j, 64j+64] (65 values, the last shared with block j+1).
//   w0      : bits 0..39 base = x[64j]; bits 40..45 l (low width); bit 63 = overflow
//   w1..    : high parts of elements 1..64 in unary (element t sets bit (r_t >> l) + t-1),
//             r_t = x[64j+t] - base
//   w7-p    : bit plane p (p < l) of the low parts, bit t-1 = bit p of r_t's low part
// A block whose encoding does not fit 448 bits is flagged and its 65 values are stored plain in
// `ovf` (w0 bits 0..39 = first word of the record).
constexpr uint32_t kBefShift = 6, kBefSpan = 64, kBefMaxL = 6;
struct BefView {
    const uint64_t *blocks; // 8 words per block
    const uint64_t *ovf;    // 65 words per overflowed block
    uint64_t n;             // number of elements
};
// decode elements t and t+1 of one 8-word block; returns false when the block is flagged as
// overflowed and no overflow table is given (the caller then falls back to another structure)
MZ_HD bool bef_decode_pair(const uint64_t (&L)[8], uint32_t t, const uint64_t *ovf, uint64_t &v0, uint64_t &v1);
MZ_HD bool bef_block_pair(const uint64_t *b, uint32_t t, const uint64_t *ovf, uint64_t &v0, uint64_t &v1) {
    uint64_t L[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 8; ++j) L[j] = b[j];
    return bef_decode_pair(L, t, ovf, v0, v1);
}
// the same from a block already held in registers
MZ_HD bool bef_decode_pair(const uint64_t (&L)[8], uint32_t t, const uint64_t *ovf, uint64_t &v0, uint64_t &v1) {
    const uint64_t hdr = L[0];
    const uint64_t base = hdr & ((1ull << 40) - 1);
    if (hdr >> 63) { // overflow block: plain values
        if (!ovf) return false;
        const uint64_t *o = ovf + base;
        v0 = o[t];
        v1 = o[t + 1];
        return true;
    }
    const uint32_t l = (uint32_t)((hdr >> 40) & 63);
    // ones number t-1 (element t) and t (element t+1) of the unary high part; element 0 is the base
    uint32_t need = t == 0 ? 0 : t - 1; // index of the first one we need
    uint32_t p_first = 0, p_second = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // straight-line form for the wavefront: running popcounts pick the word holding one number `need` (one in-word select),
    // the following one is the next set bit of that word or the first set bit of the next non-empty word
    {
        uint32_t cum[8];
        cum[0] = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) cum[j] = cum[j - 1] + popc64(L[j]);
        uint64_t word = L[1];
        uint32_t prev = 0, wi = 0;
        uint32_t nxt = 0;                           // first set bit after word wi, as a position in the 448-bit vector
        uint32_t after[8];                          // after[j]: first set bit in words j+1..7
        after[7] = 0;
#pragma unroll
        for (int j = 6; j >= 1; --j) after[j] = L[j + 1] ? (uint32_t)j * 64 + ctz64(L[j + 1]) : after[j + 1];
        nxt = after[1];
#pragma unroll
        for (int j = 2; j < 8; ++j) {
            const bool adv = cum[j - 1] <= need;
            word = adv ? L[j] : word;
            prev = adv ? cum[j - 1] : prev;
            wi = adv ? (uint32_t)(j - 1) : wi;
            nxt = adv ? after[j] : nxt;
        }
        const uint32_t bit = select_in_word(word, need - prev);
        p_first = wi * 64 + bit;
        const uint64_t rest = (word >> bit) >> 1;   // bits above `bit`
        p_second = rest ? p_first + 1 + ctz64(rest) : nxt;
    }
#else
    uint32_t acc = 0;
    bool got_first = false, got_second = false;
    for (int j = 1; j < 8; ++j) {
        uint64_t word = L[j];
        uint32_t c = popc64(word);
        if (!got_first && need < acc + c) {
            uint32_t bit = select_in_word(word, need - acc);
            p_first = (uint32_t)(j - 1) * 64 + bit;
            got_first = true;
            uint64_t rest = bit == 63 ? 0 : (word & (~0ull << (bit + 1)));
            if (rest) { p_second = (uint32_t)(j - 1) * 64 + ctz64(rest); got_second = true; }
        } else if (got_first && !got_second && word) {
            p_second = (uint32_t)(j - 1) * 64 + ctz64(word);
            got_second = true;
        }
        acc += c;
    }
#endif
    // low parts from the bit planes
    uint32_t lo_a = 0, lo_b = 0; // low part of element `need+1` and of the following element
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t pl = 0; pl < kBefMaxL; ++pl) {
        if (pl < l) {
            uint64_t plane = L[7 - pl];
            lo_a |= (uint32_t)((plane >> need) & 1) << pl;
            lo_b |= (uint32_t)((plane >> ((need + 1) & 63)) & 1) << pl;
        }
    }
    uint64_t r_a = ((uint64_t)(p_first - need) << l) | lo_a;            // element need+1
    uint64_t r_b = ((uint64_t)(p_second - (need + 1)) << l) | lo_b;     // element need+2
    if (t == 0) { v0 = base; v1 = base + r_a; }
    else { v0 = base + r_a; v1 = base + r_b; }
    return true;
}
// Encode xs[0..cnt] (cnt <= 64 elements after the base xs[0]) into one 8-word partitioned-EF block
// (layout: BefView in mz_common.hpp).  Missing elements are padded with copies of the last value so
// the unary part always holds 64 ones.  Returns false (W untouched) when it does not fit.
MZ_HD bool bef_encode_block(const uint64_t *xs, uint32_t cnt, uint64_t *W) {
    const uint64_t base = xs[0], span = xs[cnt] - base;
    uint32_t l = 0;
    bool fits = false;
    for (l = span >= kBefSpan ? (uint32_t)msb(span / kBefSpan) : 0; l <= kBefMaxL; ++l) {
        uint64_t high_bits = (span >> l) + kBefSpan + 1;
        if (high_bits + 64ull * l <= 448) { fits = true; break; }
    }
    if (!fits || base >= (1ull << 40)) return false;
    for (int q = 0; q < 8; ++q) W[q] = 0;
    W[0] = base | ((uint64_t)l << 40);
    for (uint32_t t = 1; t <= kBefSpan; ++t) {
        uint64_t r = xs[t <= cnt ? t : cnt] - base;
        uint64_t hb = (r >> l) + (t - 1);
        W[1 + (hb >> 6)] |= 1ull << (hb & 63);
        for (uint32_t pl = 0; pl < l; ++pl) W[7 - pl] |= ((r >> pl) & 1) << (t - 1);
    }
    return true;
}

MZ_HD void bef_get_pair(const BefView &e, uint64_t i, uint64_t &v0, uint64_t &v1) {
    bef_block_pair(e.blocks + (i >> kBefShift) * 8, (uint32_t)(i & (kBefSpan - 1)), e.ovf, v0, v1);
}

// ---- skew index (reference SkewIndex: boomphf over the canonical k-mers of the heavy buckets + a position per slot,
// sshash.rs:53-66) as ONE gather: a hash table of 128-byte buckets, 16 slots of (28-bit fingerprint << 36 | global k-mer
// position), empty = all ones.  A k-mer lives in its home bucket; the keys of a bucket are kept in increasing word order
// and the few beyond the 16th (load is 8 per bucket: 0.4 % of the buckets) go to a small list sorted by (bucket, word).
// The k-mer found through a fingerprint is verified against the sequence like every other candidate.
constexpr uint32_t kSkewSlots = 16, kSkewFpBits = 28, kSkewPosBits = 36, kSkewLoad = 8;
constexpr uint64_t kSkewSeed = 0x6d617a755f736b77ull, kSkewEmpty = ~0ull;
struct SkewView {
    const uint64_t *buckets;   // n_buckets * 16 slots
    const uint64_t *ovf_word;  // overflowed keys, sorted by (home bucket, word)
    const uint64_t *ovf_pos;
    uint64_t n_buckets, n_ovf, n_keys;
    uint32_t fp_mask, pad;     // (1 << fingerprint bits) - 1; kSkewFpBits unless a test narrowed it to force collisions
};
MZ_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((__uint128_t)a * b) >> 64);
#endif
}

// ---- minimizer lines ("front table") ------------------------------------------------------------------------------
// For w-mers with 2w <= 32 the bucket entries of a group of minimizer words live in ONE 128-byte line of 16 slots that is
// addressed directly by bits of the word -- no hash, no probing: a query learns membership, its bucket's entries and their
// flank fingerprints from a single gather, where the reference probes an MPHF, two Elias-Fano offsets and then the
// position array (sshash.rs:478-495).  A word is cut into  tag (t bits) | line index (L bits) | low (g bits),  2w = t + L + g:
// line = (word >> g) mod 2^L, and the slot carries low | tag << g so that the 2^(g+t) words that share a line are told
// apart.  g + t <= 8, g <= 4; t = 0 is "2^g consecutive words per line"; t > 0 folds the key space so that the table of a
// small set has as many lines as its entries need, not 4^w / 16.  Slot (64 bits):
//   bits 0..1   kind: 0 = bucket entry, 1 = the key's bucket is heavy (answered by the skew index, sshash.rs:486-490),
//               2 = the key's bucket did not fit the line (payload: its range in the entry array), 3 = empty slot
//   bits 2..10  low | tag << g (9-bit field, at most 8 bits in use: the all-ones field of an empty slot matches no word)
//   bits 11..26 kind 0: flank16 of the occurrence in the orientation of the canonical k-mer; kind 2: bucket size (saturated)
//   bit  27     kind 0: strand class of the occurrence
//   bits 28..63 kind 0: minimizer position; kind 2: first entry of the bucket in the entry array
// The entries of a key are in bucket order.  Every member key owns at least one slot, so a key with no slot in its line
// is not a minimizer of the set; a table in which some line has more than 16 member keys is not used (the builders then
// halve t and build again; t = 0 cannot overflow).  Which buckets go inline is decided per line by front_build_line().
constexpr uint32_t kFrontSlots = 16, kFrontMaxG = 4, kFrontMaxKeyBits = 8, kFrontKindBits = 2, kFrontKeyShift = 2, kFrontKeyBits = 9, kFrontFlankShift = 11,
                   kFrontClsShift = 27, kFrontPosShift = 28;
constexpr uint32_t kFrontEntry = 0, kFrontHeavy = 1, kFrontSpill = 2, kFrontEmptyKind = 3;
constexpr uint64_t kFrontEmpty = ~0ull;
constexpr uint32_t kFrontKeyMask = ((1u << kFrontKeyBits) - 1) << kFrontKeyShift;
struct FrontView {
    const uint64_t *lines; // n_lines * 16 slots; nullptr: the index has no minimizer lines
    uint64_t n_lines;      // 2^L
    uint32_t g, t;         // (2w > 32, lines addressed by front_hash of the word: g = 0, t = L)
};
MZ_HD uint64_t front_line_of(uint32_t key, uint32_t g, uint64_t n_lines) { return (uint64_t)(key >> g) & (n_lines - 1); }
// low | tag << g of a word; gl = g + L (<= 32)
MZ_HD uint32_t front_keyfield(uint32_t key, uint32_t g, uint32_t gl) { return (key & ((1u << g) - 1)) | ((uint32_t)((uint64_t)key >> gl) << g); }
struct IndexView {
    uint32_t k, w;
    uint64_t seed;
    uint64_t total_len;
    uint64_t n_unitigs;
    uint64_t skew_param;
    const uint64_t *seqbv;     // 64-byte records: words 0..3 = 128 bases of the 2-bit packed unitig concatenation,
                               // words 4..7 = the unitig boundaries of the same 128 bases (verify + locate share ONE access)
    MphfView mm;               // minimizer -> bucket id
    BefView buckets;           // bucket id -> [start,end) in pos  (occs_prefix_sum, partitioned Elias-Fano)
    const uint64_t *pos;       // packed entries: (minimizer position << 17) | (strand class << 16) | flank16
    uint32_t pos_width;
    uint32_t has_skew;
    SkewView skew;             // canonical k-mer of a heavy bucket -> global k-mer position
    FrontView front;           // minimizer lines: membership + inline bucket entries in one gather (optional)
    const uint64_t *starts;    // plain unitig starts (n_unitigs+1); used by validate_self only
};

// ---- structure probes (used by both sides) -------------------------------------------
// Direct-address monotone MPHF + bucket offsets in ONE 128-byte line ("minimizer record"), used when
// the key universe is small (w-mers with 2w <= 32 bits).  Record r covers keys [256r, 256r+256):
//   first 64 bytes : four pairs {header, 64 keys of the membership bitmap}; header = members before this record
//                    (== bucket id of its first member) | members in the earlier pairs << 40
//   second 64 bytes: a partitioned-Elias-Fano block (bef_block_pair) with the bucket starts of this
//                    record's members, psum[w0 .. w0+64]; flagged "overflow" if the record holds more
//                    than 64 members or does not fit -- then the global BlockedEF is consulted.
// bucket id = rank(key): a minimal perfect hash (order preserving) that also answers membership
// exactly, so a foreign w-mer stops after this one line, where boomphf's try_hash (sshash.rs:478)
// would usually hand back a false-positive slot; and a member needs no second gather for its
// bucket range (sshash.rs:482-483).  HBM fetches whole 128-byte lines, so the second half is free.
constexpr uint32_t kMinRecKeysLog2 = 8, kMinRecWords = 16;
// First half as four 16-byte pairs {header_j, bitmap word j} (j = key bits 6..7), so a lookup is ONE 16-byte load:
// header_j = members before the record (40 bits) | members in bitmap words < j, inside the record << 40.
constexpr uint64_t kMinRecRankMask = (1ull << 40) - 1;
MZ_HD uint32_t minrec_bitmap_word(uint32_t j) { return 2 * j + 1; }
// writes the four headers of a record whose bitmap words are in place
MZ_HD void minrec_set_headers(uint64_t *W, uint64_t rank0) {
    uint64_t c = 0;
    for (uint32_t j = 0; j < 4; ++j) {
        W[2 * j] = rank0 | (c << 40);
        c += popc64(W[2 * j + 1]);
    }
}
MZ_HD uint32_t minrec_members(const uint64_t *W) { return popc64(W[1]) + popc64(W[3]) + popc64(W[5]) + popc64(W[7]); }
// member test and rank inside the record from the pair of the key
MZ_HD bool minrec_probe(uint64_t hdr, uint64_t wv, uint32_t key, uint32_t &local) {
    const uint32_t bi = key & 63;
    local = popc64(wv & ~(~0ull << bi)) + (uint32_t)(hdr >> 40);
    return (wv >> bi) & 1;
}
MZ_HD bool direct_lookup(const uint64_t *blocks, uint64_t key, uint64_t &idx, uint32_t *local = nullptr) {
    const uint32_t x = (uint32_t)key;
    const uint64_t *b = blocks + (uint64_t)(x >> kMinRecKeysLog2) * kMinRecWords + 2 * ((x & ((1u << kMinRecKeysLog2) - 1)) >> 6);
    const uint64_t hdr = b[0], wv = b[1];
    uint32_t r;
    const bool member = minrec_probe(hdr, wv, x, r);
    idx = (hdr & kMinRecRankMask) + r;
    if (local) *local = r;
    return member;
}
// flank16 of an entry emitted on the reverse strand, brought into the orientation of the canonical word g*: the 8 bases
// reversed and complemented
MZ_HD uint32_t flank_revcomp(uint32_t f) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t x = __brev(f) >> 16;                    // bits reversed ...
    x = ((x >> 1) & 0x5555u) | ((x & 0x5555u) << 1); // ... and swapped back inside each base
#else
    uint32_t x = f & 0xFFFFu;
    x = ((x >> 8) & 0x00FFu) | ((x & 0x00FFu) << 8);
    x = ((x >> 4) & 0x0F0Fu) | ((x & 0x0F0Fu) << 4);
    x = ((x >> 2) & 0x3333u) | ((x & 0x3333u) << 2); // 2-bit groups reversed
#endif
    return x ^ 0xFFFFu;
}
// ---- minimizer lines: construction (layout at FrontView) -----------------------------------------------------------
MZ_HD uint64_t front_entry_slot(uint64_t pos, uint32_t cls, uint32_t flank_canon, uint32_t keylow) {
    return (pos << kFrontPosShift) | ((uint64_t)cls << kFrontClsShift) | ((uint64_t)flank_canon << kFrontFlankShift) | ((uint64_t)keylow << kFrontKeyShift) | kFrontEntry;
}
MZ_HD uint64_t front_marker_slot(uint32_t kind, uint32_t keylow, uint64_t first, uint64_t count) {
    const uint64_t c = count < 0xFFFFull ? count : 0xFFFFull;
    return (first << kFrontPosShift) | (c << kFrontFlankShift) | ((uint64_t)keylow << kFrontKeyShift) | kind;
}
// the 2^g membership bits of the words (group << g) .. (group << g) + 2^g - 1 from the minimizer records, and the bucket id
// of the first member among them (bucket ids are ranks, so the members' buckets are bucket0, bucket0 + 1, ...)
MZ_HD uint32_t front_line_members(const uint64_t *minrec, uint32_t g, uint64_t group, uint64_t &bucket0) {
    const uint64_t key0 = group << g;
    const uint64_t *W = minrec + (key0 >> kMinRecKeysLog2) * kMinRecWords;
    const uint32_t in_rec = (uint32_t)(key0 & ((1u << kMinRecKeysLog2) - 1)), j = in_rec >> 6, bi = in_rec & 63;
    const uint64_t hdr = W[2 * j], wv = W[2 * j + 1];
    bucket0 = (hdr & kMinRecRankMask) + (hdr >> 40) + popc64(wv & ~(~0ull << bi));
    return (uint32_t)((wv >> bi) & ((1ull << (1u << g)) - 1)); // 2^g <= 16 keys never straddle a bitmap word
}
// Minimizer lines for w-mers of more than 32 bits (2w > 32).  The words cannot address the table, so a HASH of the word does:
// line = low bits of front_hash(word), and the slot's key field carries 9 more bits of the hash as a fingerprint.  The fingerprint is
// not the key, so the builder keeps the table exact another way: member words of a line that share a fingerprint get ONE slot between
// them, a marker that says "take the exact walk through the MPHF" (a spill marker whose size field is saturated, kFrontWalkSize), and a
// line that more than 16 member words hash to is not used at all (its first slot is kFrontUnusable; its queries walk too).  Every other
// member word owns slots under a fingerprint of its own, so "no slot with my fingerprint" means "not a minimizer of the set", and a
// foreign word that happens to share a member's fingerprint is caught like every false candidate: by the flanks, then by the sequence.
// At a mean of 2.4 member words per line 0.5 % of the words sit behind a walk marker.
constexpr uint64_t kFrontHashSeed = 0x6d617a755f66726full;
MZ_HD uint64_t front_hash(uint64_t word) { const uint64_t h = mum(word ^ kFrontHashSeed, kP3); return h ^ (h >> 29); }
MZ_HD bool front_is_hashed(uint32_t w) { return 2 * w > 32; }
MZ_HD uint64_t front_hashed_line(uint64_t h, uint64_t n_lines) { return h & (n_lines - 1); }
MZ_HD uint32_t front_hashed_keyfield(uint64_t h) { // 9 bits; 0x100 (the pattern of an idle query, the key of kFrontUnusable) and 0x1FF are left out
    const uint32_t f = (uint32_t)(h >> 55);
    return f == 0x100u || f == 0x1FFu ? f ^ 0x80u : f;
}
constexpr uint64_t kFrontWalkSize = ~0ull; // bucket size of a member that only gets a walk marker (front_merge_clashes)
// member words of one line that share a fingerprint are merged into one member of size kFrontWalkSize (member order kept); returns the new count
MZ_HD uint32_t front_merge_clashes(uint32_t m, uint32_t *keyfield, uint64_t *first, uint64_t *size, uint32_t *n_walk) {
    uint32_t o = 0;
    for (uint32_t a = 0; a < m; ++a) {
        bool clash = false;
        for (uint32_t c = 0; c < o && !clash; ++c)
            if (keyfield[c] == keyfield[a]) {
                if (size[c] != kFrontWalkSize) ++*n_walk;
                size[c] = kFrontWalkSize;
                first[c] = 0;
                clash = true;
            }
        if (!clash) { keyfield[o] = keyfield[a]; first[o] = first[a]; size[o] = size[a]; ++o; }
    }
    return o;
}
constexpr uint64_t kFrontUnusable = ((uint64_t)0x100u << kFrontKeyShift) | kFrontSpill; // key field 0x100: no query shows it
// Packs the m <= 16 member keys of one line (key field, first entry and size of each bucket) into its 16 slots.  Light buckets
// are taken inline in order of increasing size (ties: member order) while every bucket left out still keeps one slot for
// its marker.
MZ_HD void front_pack_line(uint32_t m, const uint32_t *keyfield, const uint64_t *first, const uint64_t *size, const uint64_t *entries, uint64_t skew_param, uint64_t *out) {
    uint32_t size_c[kFrontSlots];
    bool inl[kFrontSlots];
    for (uint32_t a = 0; a < m; ++a) {
        size_c[a] = size[a] > skew_param || size[a] > kFrontSlots ? 0xFFFFFFFFu : (uint32_t)size[a]; // heavy or hopeless: never inline
        inl[a] = false;
    }
    uint32_t used = m; // one slot per member to begin with
    for (;;) {         // repeatedly take the smallest bucket not yet inline (ties: first member)
        uint32_t best = kFrontSlots;
        for (uint32_t a = 0; a < m; ++a)
            if (!inl[a] && size_c[a] != 0xFFFFFFFFu && (best == kFrontSlots || size_c[a] < size_c[best])) best = a;
        if (best == kFrontSlots || used - 1 + size_c[best] > kFrontSlots) break;
        inl[best] = true;
        used += size_c[best] - 1;
    }
    uint32_t o = 0;
    for (uint32_t a = 0; a < m; ++a) {
        if (inl[a]) {
            for (uint64_t e = 0; e < size[a]; ++e) {
                const uint64_t ent = entries[first[a] + e];
                const uint32_t cls = (uint32_t)(ent >> 16) & 1u, fl = (uint32_t)(ent & 0xFFFFu);
                out[o++] = front_entry_slot(ent >> 17, cls, cls ? flank_revcomp(fl) : fl, keyfield[a]);
            }
        } else {
            out[o++] = front_marker_slot(size[a] != kFrontWalkSize && size[a] > skew_param ? kFrontHeavy : kFrontSpill, keyfield[a], first[a], size[a]);
        }
    }
    for (; o < kFrontSlots; ++o) out[o] = kFrontEmpty;
}
// Packs line `line` of a table with 2^line_bits lines.  minrec: the minimizer records (membership and bucket ids of the
// 2^t groups of 2^g words that share the line); psum: bucket starts; entries: the entry array (`pos` layout).  Light buckets
// are taken inline in order of increasing size (ties: member order) while every bucket left out still keeps one slot for
// its marker.  Returns false, with the line left empty, when more than 16 member keys share the line.
MZ_HD bool front_build_line(uint32_t g, uint32_t t, uint32_t line_bits, uint64_t line, const uint64_t *minrec, const uint64_t *psum,
                            const uint64_t *entries, uint64_t skew_param, uint64_t *out) {
    uint32_t keyfield[kFrontSlots];
    uint64_t first[kFrontSlots], size[kFrontSlots];
    uint32_t m = 0;
    bool overfull = false;
    for (uint32_t tag = 0; tag < (1u << t) && !overfull; ++tag) {
        uint64_t bucket0;
        const uint32_t member_bits = front_line_members(minrec, g, line | ((uint64_t)tag << line_bits), bucket0);
        uint32_t seen = 0;
        for (uint32_t j = 0; j < (1u << g); ++j)
            if ((member_bits >> j) & 1) {
                if (m == kFrontSlots) { overfull = true; break; }
                keyfield[m] = j | (tag << g);
                first[m] = psum[bucket0 + seen];
                size[m] = psum[bucket0 + seen + 1] - first[m];
                ++seen;
                ++m;
            }
    }
    if (overfull) {
        for (uint32_t o = 0; o < kFrontSlots; ++o) out[o] = kFrontEmpty;
        return false;
    }
    front_pack_line(m, keyfield, first, size, entries, skew_param, out);
    return true;
}
// Shape of the table for a key universe of 2^(2w) words and n_occ entries: the fewest lines (a power of two, at least
// 4^w / 256) that keep the mean line load at or under max_load entries; g = as many of the shared-line bits as fit the low
// part (<= 4), the rest is tag.  Returns false when 2w > 32.
MZ_HD bool front_shape_for_load(uint32_t w, uint64_t n_occ, uint64_t max_load, uint32_t &g, uint32_t &t) {
    if (2 * w > 32) return false;
    const uint32_t u = 2 * w, min_bits = u > kFrontMaxKeyBits ? u - kFrontMaxKeyBits : 0;
    uint32_t line_bits = min_bits;
    while (line_bits < u && n_occ > max_load * (1ull << line_bits)) ++line_bits;
    g = u - line_bits < kFrontMaxG ? u - line_bits : kFrontMaxG;
    t = u - line_bits - g;
    return true;
}
// The table must stay within max(64 MiB, 64 bytes per entry, min(9 GiB, 1 KiB per entry)): one MI355X holds 288 GB, and the
// lines buy a fifth more point-query throughput than the minimizer records alone.  Preferred: a mean line load of at most 3
// of the 16 slots (S-human: 2.0 entries per line, 96 % of the entry mass inline, 1.3 % spilled -- a spilled bucket costs a
// query an extra dependent access, and 6 % of the throughput separate this shape from the next); if that table is too
// large, a load of at most 5 (94 % inline, 3 % spilled), then at most 10 (88 % inline, 9 % spilled).
constexpr uint64_t kFrontFreeBytes = 9ull << 30, kFrontFreeBytesPerEntry = 1024, kFrontBytesPerEntry = 64;
MZ_HD uint64_t front_table_cap(uint64_t n_occ) {
    uint64_t cap = 64ull << 20;
    if (kFrontBytesPerEntry * n_occ > cap) cap = kFrontBytesPerEntry * n_occ;
    const uint64_t free_cap = kFrontFreeBytesPerEntry * n_occ < kFrontFreeBytes ? kFrontFreeBytesPerEntry * n_occ : kFrontFreeBytes;
    return free_cap > cap ? free_cap : cap;
}
MZ_HD bool front_choose_shape(uint32_t w, uint64_t n_occ, uint32_t &g, uint32_t &t) {
    const uint64_t loads[3] = {3, 5, 10};
    for (int a = 0; a < 3; ++a)
        if (front_shape_for_load(w, n_occ, loads[a], g, t) && (1ull << (2 * w - g - t)) * kFrontSlots * 8 <= front_table_cap(n_occ)) return true;
    return false;
}

// ---- negative filter in the slack of a record's offset block --------------------------------------------
// The words of the partitioned-EF block between the end of its unary part and its bit planes are unused (typically 3-4 of
// the 7: ~45 members, span ~130).  They hold a one-probe Bloom filter over (member index, inner flank) of every entry of
// the record's LIGHT buckets; inner flank = the 2 bases left and the 2 right of the minimizer occurrence in the
// orientation of the canonical k-mer (bits 4..11 of flank16).  A query whose k-mer covers those 4 bases and whose pattern
// is absent cannot match any entry of its bucket: a foreign k-mer that merely shares a member minimizer (97 % of random
// negatives at S-human) then stops after the record instead of fetching and scanning a bucket.  The decoder never looks
// past the 64th one of the unary part, so the filter words are invisible to it.  Header bits 48..50 = number of filter
// words, 51..53 = index of the first one.
constexpr uint32_t kFilterFlankBases = 2;
MZ_HD uint64_t minrec_filter_header(uint64_t bef_hdr, uint64_t span) {
    const uint32_t l = (uint32_t)((bef_hdr >> 40) & 63);
    const uint32_t unary_words = (uint32_t)(((span >> l) + kBefSpan + 1 + 63) >> 6);
    const uint32_t first = 1 + unary_words, last = 7 - l; // words first..last are free
    const uint32_t nf = last >= first ? last - first + 1 : 0;
    return nf ? (((uint64_t)nf << 48) | ((uint64_t)first << 51)) : 0;
}
// filter slot of (member index inside the record, flank16 in the orientation of g*) as (word inside the block, bit);
// returns false when the record has no filter
MZ_HD bool minrec_filter_slot(uint64_t bef_hdr, uint32_t local, uint32_t flank16_g, uint32_t &word, uint32_t &bit) {
    const uint32_t nf = (uint32_t)(bef_hdr >> 48) & 7u, first = (uint32_t)(bef_hdr >> 51) & 7u;
    const uint32_t x = ((local << 8) | ((flank16_g >> 4) & 0xFFu)) * 0x9E3779B1u;
    const uint32_t slot = ((x >> 16) * (nf * 64u)) >> 16;
    word = first + (slot >> 6);
    bit = slot & 63u;
    return nf != 0 && (bef_hdr >> 63) == 0;
}

// member test + bucket id + bucket range from one record; `global` is the fallback for overflowed records
MZ_HD bool minrec_lookup(const uint64_t *blocks, const BefView &global, uint64_t key, uint64_t &h, uint64_t &ps, uint64_t &pe) {
    uint32_t local;
    if (!direct_lookup(blocks, key, h, &local)) return false;
    const uint64_t *b = blocks + (uint64_t)((uint32_t)key >> kMinRecKeysLog2) * kMinRecWords + 8;
    if (local >= kBefSpan || !bef_block_pair(b, local, nullptr, ps, pe)) bef_get_pair(global, h, ps, pe);
    return true;
}

// MPHF probe == boomphf try_hash: first level whose bit for the key is set gives the index.
MZ_HD bool mphf_lookup(const MphfView &m, uint64_t key, uint64_t &idx) {
    if (m.direct) return direct_lookup(m.blocks, key, idx);
    for (uint32_t lv = 0; lv < m.n_levels; ++lv) {
        const MphfLevel L = m.levels[lv];
        uint32_t blk, bit;
        mphf_level_slot(key, L.seed, L.n_blocks, blk, bit);
        const uint64_t *b = m.blocks + (L.block_off + blk) * 8;
        uint32_t dw = bit >> 6, bi = bit & 63;
        uint64_t word = b[2 + dw];
        uint64_t h0 = b[0], h1 = b[1];
        if ((word >> bi) & 1) {
            uint64_t below = bi ? (word & ((1ull << bi) - 1)) : 0;
            idx = h0 + ((h1 >> (9 * dw)) & 511u) + popc64(below);
            return true;
        }
    }
    return false;
}

// Elias-Fano get(i) and get(i+1) in one go (elias_fano.rs:116-122): select1 on the high bits via
// the sampled hint, then the packed low bits.  Requires i+1 < n.
MZ_HD void ef_get_pair(const EfView &e, uint64_t i, uint64_t &v0, uint64_t &v1) {
    uint64_t s = e.sel[i >> kEfSelLog2];
    uint32_t r = (uint32_t)(i & ((1u << kEfSelLog2) - 1));
    uint64_t wi = s >> 6;
    uint64_t word = e.high[wi] & (~0ull << (s & 63));
    uint32_t c = popc64(word);
    while (r >= c) {
        r -= c;
        ++wi;
        word = e.high[wi];
        c = popc64(word);
    }
    uint32_t b0 = select_in_word(word, r);
    uint64_t p0 = (wi << 6) + b0;
    // next one
    word = b0 == 63 ? 0 : (word & (~0ull << (b0 + 1)));
    while (word == 0) {
        ++wi;
        word = e.high[wi];
    }
    uint64_t p1 = (wi << 6) + ctz64(word);
    uint64_t lowpair_bits = i * e.l;
    uint64_t lw = lowpair_bits >> 6;
    uint32_t sh = (uint32_t)(lowpair_bits & 63);
    uint64_t a = e.low[lw], b = e.low[lw + 1];
    uint64_t x = sh ? ((a >> sh) | (b << (64 - sh))) : a; // 64 bits from bit i*l; 2*l <= 64 asserted at build
    uint64_t lmask = (1ull << e.l) - 1;
    v0 = ((p0 - i) << e.l) | (x & lmask);
    v1 = ((p1 - (i + 1)) << e.l) | ((x >> e.l) & lmask);
}

MZ_HD uint64_t ef_get(const EfView &e, uint64_t i) {
    uint64_t s = e.sel[i >> kEfSelLog2];
    uint32_t r = (uint32_t)(i & ((1u << kEfSelLog2) - 1));
    uint64_t wi = s >> 6;
    uint64_t word = e.high[wi] & (~0ull << (s & 63));
    uint32_t c = popc64(word);
    while (r >= c) {
        r -= c;
        ++wi;
        word = e.high[wi];
        c = popc64(word);
    }
    uint64_t p0 = (wi << 6) + select_in_word(word, r);
    return ((p0 - i) << e.l) | read_packed(e.low, i, e.l);
}

// Boundary block probe: unitig id, start and end of the unitig that holds global base `pos`
// (== bv.rank(pos) + accum_lens.get(id) + accum_lens.get(id+1), unitig_set.rs:178-204) from ONE line.
struct UnitigSpan {
    uint64_t id, start, end;
};
MZ_HD UnitigSpan locate_unitig(const uint64_t *b, uint32_t off, uint64_t base);
MZ_HD UnitigSpan locate_unitig(const uint64_t *bv_blocks, uint64_t pos) {
    return locate_unitig(bv_blocks + (pos >> kBvBasesPerBlockLog2) * 8, (uint32_t)(pos & (kBvBasesPerBlock - 1)),
                         pos & ~(uint64_t)(kBvBasesPerBlock - 1));
}
// boundary block given by value: header (first unitig id, its start, end of the last unitig) + the four words of boundary bits
MZ_HD UnitigSpan locate_unitig_w(uint64_t id0, uint64_t start0, uint64_t end_last, uint64_t d0, uint64_t d1, uint64_t d2, uint64_t d3,
                                 uint32_t off, uint64_t base) {
    uint64_t d[4] = {d0, d1, d2, d3};
    uint32_t wsel = off >> 6, bi = off & 63;
    uint64_t below_mask = bi ? ((1ull << bi) - 1) : 0;
    UnitigSpan s;
    uint32_t rank = 0;
    int last_below = -1;  // highest set bit strictly below off
    int first_at = -1;    // lowest set bit at or above off
    for (uint32_t j = 0; j < 4; ++j) {
        uint64_t w = d[j];
        uint64_t lo = j < wsel ? w : (j == wsel ? (w & below_mask) : 0);
        uint64_t hi = j > wsel ? w : (j == wsel ? (w & ~below_mask) : 0);
        rank += popc64(lo);
        if (lo) last_below = (int)(j * 64 + 63 - clz64(lo));
        if (hi && first_at < 0) first_at = (int)(j * 64 + ctz64(hi));
    }
    s.id = id0 + rank;
    s.start = last_below >= 0 ? base + (uint64_t)last_below + 1 : start0;
    s.end = first_at >= 0 ? base + (uint64_t)first_at + 1 : end_last;
    return s;
}
MZ_HD UnitigSpan locate_unitig(const uint64_t *b, uint32_t off, uint64_t base) {
    return locate_unitig_w(b[0], b[1], b[2], b[4], b[5], b[6], b[7], off, base);
}


// ---- interleaved sequence / boundary records (IndexView::seqbv) ----------------------------------------
// One 64-byte record per 96 bases, holding 128: record r = bases [96 r, 96 r + 128) (words 0..3, the 2-bit packed bases:
// 96 bases = 3 words, so these are words 3r .. 3r+3 of the packed sequence) + the unitig boundaries of the same 128 bases:
//   w4 = id of the unitig holding the record's first base | low 32 bits of that unitig's start << 32
//   w5 = end (exclusive) of the unitig holding the record's last base (36 bits) | bits 32..35 of the start << 36
//   w6, w7 = one bit per base, set at the LAST base of every unitig (unitig_set.rs:90-96)
// Consecutive records overlap by 32 bases, so a k-mer (k <= 31) starting at any position lies inside ONE record
// (r = pos / 96) and its verification plus the unitig lookup (== bv.rank(pos) + accum_lens.get(id) + accum_lens.get(id+1),
// unitig_set.rs:178-204) cost one random 64-byte access -- half a 128-byte line, which the memory system serves at a
// higher rate than accesses that touch both halves.
constexpr uint32_t kSeqRecWords = 8;
constexpr uint32_t kSeqRecStride = 96, kSeqRecStrideWords = kSeqRecStride / 32, kSeqRecBases = 128;
MZ_HD uint64_t seq_rec_of(uint64_t pos, uint32_t &off) { // positions are < 2^36 (kPosBits)
    const uint32_t r = (uint32_t)(pos >> 5) / kSeqRecStrideWords;
    off = (uint32_t)(pos - (uint64_t)r * kSeqRecStride);
    return r;
}
MZ_HD uint64_t read_kmer_i(const uint64_t *rec, uint64_t pos, uint32_t k) {
    uint32_t off;
    const uint64_t *R = rec + seq_rec_of(pos, off) * kSeqRecWords;
    const uint32_t wi = off >> 5, sh = 2 * (off & 31); // off <= 95: wi + 1 <= 3
    const uint64_t lo = R[wi], hi = R[wi + 1];
    const uint64_t x = (lo >> sh) | ((hi << 1) << (63 - sh)); // branch-free funnel shift (sh == 0: second term vanishes): both loads stay unconditional
    return x & kmer_mask(k);
}
MZ_HD void seq_rec_header(uint64_t id0, uint64_t start0, uint64_t end_last, uint64_t &h0, uint64_t &h1) {
    h0 = id0 | (start0 << 32);
    h1 = end_last | ((start0 >> 32) << 36);
}
// boundary half of a record given by value: unitig id, start and end of the unitig that holds base `base + off` (off < 128)
MZ_HD UnitigSpan locate_unitig_r(uint64_t h0, uint64_t h1, uint64_t d0, uint64_t d1, uint32_t off, uint64_t base) {
    const bool up = off >= 64;
    const uint64_t below = ~(~0ull << (off & 63));                    // bits under `off` inside its word
    const uint64_t lo0 = up ? d0 : d0 & below, lo1 = up ? d1 & below : 0; // boundary bits strictly below off
    const uint64_t hi0 = up ? 0 : d0 & ~below, hi1 = up ? d1 & ~below : d1; // ... at or above off
    UnitigSpan s;
    s.id = (h0 & 0xFFFFFFFFull) + popc64(lo0) + popc64(lo1);
    const uint64_t start0 = (h0 >> 32) | ((h1 >> 36) << 32), end_last = h1 & ((1ull << 36) - 1);
    s.start = lo1 ? base + 128 - clz64(lo1) : lo0 ? base + 64 - clz64(lo0) : start0;   // after the highest boundary below off
    s.end = hi0 ? base + ctz64(hi0) + 1 : hi1 ? base + 65 + ctz64(hi1) : end_last;      // through the lowest boundary at or above off
    return s;
}
MZ_HD UnitigSpan locate_unitig_i(const uint64_t *rec, uint64_t pos) {
    uint32_t off;
    const uint64_t r = seq_rec_of(pos, off);
    const uint64_t *R = rec + r * kSeqRecWords;
    return locate_unitig_r(R[4], R[5], R[6], R[7], off, r * kSeqRecStride);
}

// ---- `pos` entry: (minimizer position << 17) | (strand class << 16) | flank16 ---------------------
// flank16 = the 4 bases left of the minimizer occurrence (low byte; base pos-4 in the low bits) and
// the 4 bases right of it (high byte) on the forward strand of the unitig concatenation.  A query
// whose k-mer really sits on this occurrence must agree with those bases wherever its k-mer covers
// them, so almost every foreign entry of a bucket is rejected WITHOUT touching the packed sequence
// (at least 4 flank bases are always covered: min(4,off) + min(4,k-w-off) >= 4 when k-w >= 4).
// Entries that pass are verified against the 2-bit sequence exactly as in sshash.rs:500-504.
constexpr uint32_t kFlankBases = 4, kEntryFlankBits = 4 * kFlankBases, kEntryShift = kEntryFlankBits + 1;
constexpr uint32_t kPosEntryBits = 64; // stored width of an entry: whole words, so a gather is one aligned 8-byte load that never crosses a line
MZ_HD uint64_t make_entry(uint64_t pos, uint32_t cls, uint32_t flank16) { return (pos << kEntryShift) | ((uint64_t)cls << kEntryFlankBits) | flank16; }
MZ_HD uint32_t flank_of(const uint64_t *useq, uint64_t total_len, uint64_t pos, uint32_t w) {
    uint32_t f = 0;
    for (uint32_t j = 0; j < kFlankBases; ++j) {
        uint64_t pl = pos + j - kFlankBases, pr = pos + w + j; // pl wraps for pos < 4: treated as absent
        uint32_t bl = (pos + j >= kFlankBases && pl < total_len) ? (uint32_t)((useq[pl >> 5] >> (2 * (pl & 31))) & 3) : 0;
        uint32_t br = pr < total_len ? (uint32_t)((useq[pr >> 5] >> (2 * (pr & 31))) & 3) : 0;
        f |= bl << (2 * j);
        f |= br << (2 * (kFlankBases + j));
    }
    return f;
}
// does the expected strand word X (minimizer at offset `off` inside it) agree with the entry's flanks?
MZ_HD bool flank_ok(uint64_t X, uint32_t off, uint32_t k, uint32_t w, uint32_t flank16) {
    uint32_t nl = off < kFlankBases ? off : kFlankBases;
    uint32_t room = k - w - off, nr = room < kFlankBases ? room : kFlankBases;
    uint32_t ql = (uint32_t)(X >> (2 * (off - nl))) & ((1u << (2 * nl)) - 1);
    uint32_t el = ((flank16 & 0xFFu) >> (2 * (kFlankBases - nl))) & ((1u << (2 * nl)) - 1);
    uint32_t qr = nr ? (uint32_t)(X >> (2 * (off + w))) & ((1u << (2 * nr)) - 1) : 0;
    uint32_t er = (flank16 >> (2 * kFlankBases)) & ((1u << (2 * nr)) - 1);
    return ql == el && qr == er;
}

// ---- the k2u lookup chain (reference src/kphf/sshash.rs:471-554) ------------------------
// Compiled for the device by mz_device.hip (the product path).  tests/cpp/host_check.cpp also
// compiles it for the host to debug the builder without a GPU; libmazu_amd.so exports no host
// query entry point.
MZ_HD mz_k2upos make_none() { return mz_k2upos{0u, 0u, 0u, MZ_NO_MATCH}; }

// CanonicalKmer::get_word_equivalency (sshash.rs:503,532): Identity is tested first
MZ_HD uint32_t word_equivalency(uint64_t fw, uint64_t rc, uint64_t kw) {
    return kw == fw ? MZ_IDENTITY_MATCH : (kw == rc ? MZ_TWIN_MATCH : MZ_NO_MATCH);
}

MZ_HD mz_k2upos make_hit(const UnitigSpan &s, uint64_t km_pos, uint32_t mt) {
    return mz_k2upos{(uint32_t)s.id, (uint32_t)(km_pos - s.start), (uint32_t)(s.end - s.start), mt};
}

// sshash.rs:415-433 + SkewIndex::lookup :59-66
// home bucket and fingerprint of a canonical k-mer word
MZ_HD void skew_home(uint64_t word, uint64_t n_buckets, uint32_t fp_mask, uint64_t &bucket, uint32_t &fp) {
    const uint64_t h = mum(word ^ kSkewSeed, kP1);
    bucket = mulhi64(h ^ (h >> 32), n_buckets);
    fp = (uint32_t)h & fp_mask;
}
MZ_HD uint64_t skew_slot(uint32_t fp, uint64_t pos) { return ((uint64_t)fp << kSkewPosBits) | pos; }
// exact lookup (every slot with the k-mer's fingerprint is verified, then the overflow list): sshash.rs:415-433 + :59-66
MZ_HD mz_k2upos skew_lookup(const IndexView &ix, uint64_t fw, uint64_t rc) {
    if (!ix.has_skew || ix.skew.n_buckets == 0) return make_none();
    const uint64_t word = fw < rc ? fw : rc;
    uint64_t b;
    uint32_t fp;
    skew_home(word, ix.skew.n_buckets, ix.skew.fp_mask, b, fp);
    const uint64_t *B = ix.skew.buckets + b * kSkewSlots;
    uint32_t used = 0;
    for (; used < kSkewSlots; ++used) {
        const uint64_t e = B[used];
        if (e == kSkewEmpty) break;
        if ((uint32_t)(e >> kSkewPosBits) != fp) continue;
        const uint64_t p = e & ((1ull << kSkewPosBits) - 1);
        const uint32_t mt = word_equivalency(fw, rc, read_kmer_i(ix.seqbv, p, ix.k));
        if (mt != MZ_NO_MATCH) return make_hit(locate_unitig_i(ix.seqbv, p), p, mt);
    }
    if (used < kSkewSlots) return make_none();
    uint64_t lo = 0, hi = ix.skew.n_ovf; // first overflow key >= (b, word)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1, mw = ix.skew.ovf_word[mid];
        uint64_t mb;
        uint32_t mf;
        skew_home(mw, ix.skew.n_buckets, ix.skew.fp_mask, mb, mf);
        if (mb < b || (mb == b && mw < word)) lo = mid + 1; else hi = mid;
    }
    if (lo < ix.skew.n_ovf && ix.skew.ovf_word[lo] == word) {
        const uint64_t p = ix.skew.ovf_pos[lo];
        const uint32_t mt = word_equivalency(fw, rc, read_kmer_i(ix.seqbv, p, ix.k));
        if (mt != MZ_NO_MATCH) return make_hit(locate_unitig_i(ix.seqbv, p), p, mt);
    }
    return make_none();
}

// sshash.rs:478-554 once the canonical minimizer is known
MZ_HD mz_k2upos lookup_with_minimizer(const IndexView &ix, uint64_t fw, uint64_t rc, Minimizer mn) {
    const uint32_t k = ix.k, w = ix.w;
    uint64_t h, ps, pe;
    if (ix.mm.direct) {                                            // :478-483 from one minimizer record
        if (!minrec_lookup(ix.mm.blocks, ix.buckets, mn.word, h, ps, pe)) return make_none();
    } else {
        if (!mphf_lookup(ix.mm, mn.word, h)) return make_none();   // :478-479
        bef_get_pair(ix.buckets, h, ps, pe);                       // :482-483
    }
    if (pe - ps > ix.skew_param) return skew_lookup(ix, fw, rc);   // :486-490
    const uint64_t last_km_start = ix.total_len - k;               // :492
    const uint64_t off_a = mn.offset, off_b = k - mn.offset - w;   // :493
    for (uint64_t pi = ps; pi < pe; ++pi) {                        // :494
        // Entry = (minimizer position << 1) | class.  Class 0: the occurrence was emitted by a
        // forward-canonical k-mer, so only candidate A (:498) can be a true occurrence; class 1:
        // reverse-canonical, only candidate B (:527).  The other candidate of the reference loop
        // can never verify: had B verified on a class-0 entry, the verified k-mer would be reverse-canonical and would itself have
        // emitted this (word, position) with class 1, which is its own entry.  Skipping the other candidate keeps results identical.
        uint64_t ent = ix.pos[pi];
        const bool cls = (ent >> kEntryFlankBits) & 1;
        uint64_t mm_pos = ent >> kEntryShift, off = cls ? off_b : off_a;
        // expected word on the unitig strand: g* for class 0, its reverse complement for class 1
        const uint64_t g = fw <= rc ? fw : rc, X = cls ? (fw <= rc ? rc : fw) : g;
        if (!flank_ok(X, (uint32_t)off, k, w, (uint32_t)(ent & 0xFFFFu))) continue;
        if (mm_pos >= off && mm_pos - off <= last_km_start) {
            uint64_t p = mm_pos - off;
            uint32_t mt = word_equivalency(fw, rc, read_kmer_i(ix.seqbv, p, k));
            if (mt != MZ_NO_MATCH) {
                UnitigSpan s = locate_unitig_i(ix.seqbv, p);
                if (p + k <= s.end) return make_hit(s, p, mt);     // :509-514, :538-541
            }
        }
    }
    return make_none();                                            // :553
}

MZ_HD mz_k2upos lookup(const IndexView &ix, uint64_t fw) {
    fw &= kmer_mask(ix.k);
    uint64_t rc = kmer_rc(fw, ix.k);
    Minimizer mn = canonical_minimizer(fw, rc, ix.k, ix.w, ix.seed); // :476
    return lookup_with_minimizer(ix, fw, rc, mn);
}


} // namespace mz
