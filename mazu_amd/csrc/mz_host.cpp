// mz_host.cpp -- index construction on the host (multi-threaded C++17).
// Steps mirror reference src/kphf/sshash.rs:86-329; the produced arrays are the HBM layout
// described next to the views in mz_common.hpp.  No query code lives here: queries run on the GPU only.
#include "mz_host.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <thread>

#include "../../include/mazu_amd.h"

namespace mz {

// ------------------------------------------------------------------ errors / threads
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// usable host threads: hardware concurrency capped by the cgroup CPU quota (a container may see
// 256 CPUs but own 16 of them; oversubscribing the quota only adds throttling)
int hw_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    double quota = 0;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
        char q[64];
        long long period = 0;
        if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) quota = std::atof(q) / (double)period;
        std::fclose(f);
    } else {
        long long q = -1, period = 0;
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &q) != 1) q = -1; std::fclose(g); }
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
        if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota >= 1.0 && quota < (double)n) n = (unsigned)(quota + 0.5);
    if (const char *e = std::getenv("MZ_NUM_THREADS")) { int v = std::atoi(e); if (v > 0) n = (unsigned)v; }
    return (int)n;
}

// run fn(tid, begin, end) over [0, n) split into contiguous shards
template <class F>
static void parallel_for(int n_threads, uint64_t n, F fn) {
    if (n_threads <= 1 || n < 4096) {
        fn(0, (uint64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    uint64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        uint64_t b = std::min<uint64_t>(n, per * t), e = std::min<uint64_t>(n, b + per);
        if (b >= e) break;
        th.emplace_back([=] { fn(t, b, e); });
    }
    for (auto &x : th) x.join();
}

// run fn(tid, task) for task in [0, n_tasks) with dynamic scheduling
template <class F>
static void parallel_tasks(int n_threads, uint64_t n_tasks, F fn) {
    if (n_threads <= 1 || n_tasks <= 1) {
        for (uint64_t i = 0; i < n_tasks; ++i) fn(0, i);
        return;
    }
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t] {
            for (;;) {
                uint64_t i = next.fetch_add(1);
                if (i >= n_tasks) break;
                fn(t, i);
            }
        });
    for (auto &x : th) x.join();
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ PackedVec
void PackedVec::assign(const uint64_t *xs, uint64_t n_, uint64_t max_value, int n_threads) {
    n = n_;
    width = bit_len(max_value);
    uint64_t n_words = (n * width + 63) / 64 + 2;
    w.assign(n_words, 0);
    // shard on word-aligned element boundaries so no two threads touch one word:
    // element e starts a new word when e*width % 64 == 0, i.e. e multiple of 64/gcd(width,64) <= 64
    uint64_t n_groups = (n + 63) / 64;
    uint64_t *W = w.data();
    uint32_t wd = width;
    parallel_for(n_threads, n_groups, [=](int, uint64_t gb, uint64_t ge) {
        uint64_t eb = gb * 64, ee = std::min(n, ge * 64);
        uint64_t mask = wd >= 64 ? ~0ull : ((1ull << wd) - 1);
        for (uint64_t i = eb; i < ee; ++i) {
            uint64_t b = i * wd, wi = b >> 6;
            uint32_t sh = (uint32_t)(b & 63);
            uint64_t x = xs[i] & mask;
            W[wi] |= x << sh;
            if (sh + wd > 64) W[wi + 1] |= x >> (64 - sh);
        }
    });
}

// ------------------------------------------------------------------ Elias-Fano
// reference src/elias_fano.rs:55-114 (from_iter)
int EFVector::build(const uint64_t *xs, uint64_t n_, uint64_t u_) {
    if (n_ == 0) {
        set_error("Elias-Fano: empty sequence (Error::EFEmpty)");
        return MZ_EEMPTY;
    }
    n = n_;
    u = u_;
    uint64_t lb = msb(u / n);
    if (lb == 0) lb = 1; // "hack to avoid l == 0", elias_fano.rs:65-75
    l = (uint32_t)lb;
    if (l > 32) {
        set_error("Elias-Fano: low width %u > 32 unsupported", l);
        return MZ_EINVAL;
    }
    n_high_bits = n + (u >> l);
    high.assign(n_high_bits / 64 + 4, 0);
    low.assign((n * l + 63) / 64 + 2, 0);
    uint64_t low_mask = (1ull << l) - 1, last = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t x = xs[i];
        if (x < last) {
            set_error("Elias-Fano: sequence not monotone at %llu (Error::EFNotMonotone)", (unsigned long long)i);
            return MZ_ENOTMONOTONE;
        }
        last = x;
        uint64_t hb = (x >> l) + i; // == h0s + i + gap of the reference loop
        high[hb >> 6] |= 1ull << (hb & 63);
        uint64_t b = i * l, wi = b >> 6;
        uint32_t sh = (uint32_t)(b & 63);
        uint64_t lo = x & low_mask;
        low[wi] |= lo << sh;
        if (sh + l > 64) low[wi + 1] |= lo >> (64 - sh);
    }
    // select samples: bit position of one number j*2^kEfSelLog2
    sel.assign((n >> kEfSelLog2) + 2, 0);
    uint64_t ones = 0;
    for (uint64_t wi = 0; wi < high.size(); ++wi) {
        uint64_t word = high[wi];
        while (word) {
            if ((ones & ((1u << kEfSelLog2) - 1)) == 0) sel[ones >> kEfSelLog2] = (wi << 6) + ctz64(word);
            ++ones;
            word &= word - 1;
        }
    }
    return MZ_OK;
}


// ------------------------------------------------------------------ partitioned Elias-Fano
int BlockedEF::build(const uint64_t *xs, uint64_t n_, int n_threads) {
    if (n_ == 0) {
        set_error("Elias-Fano: empty sequence (Error::EFEmpty)");
        return MZ_EEMPTY;
    }
    for (uint64_t i = 1; i < n_; ++i)
        if (xs[i] < xs[i - 1]) {
            set_error("Elias-Fano: sequence not monotone at %llu (Error::EFNotMonotone)", (unsigned long long)i);
            return MZ_ENOTMONOTONE;
        }
    if (xs[n_ - 1] >= (1ull << 40)) {
        set_error("Elias-Fano: values above 2^40 unsupported");
        return MZ_EINVAL;
    }
    n = n_;
    uint64_t nb = ((n - 1) >> kBefShift) + 1;
    blocks.assign(nb * 8, 0);
    std::vector<uint8_t> over(nb, 0);
    uint64_t *B = blocks.data();
    uint8_t *OV = over.data();
    parallel_for(n_threads, nb, [=](int, uint64_t bb, uint64_t be) {
        for (uint64_t j = bb; j < be; ++j) {
            uint64_t first = j << kBefShift;
            uint64_t last = std::min<uint64_t>(first + kBefSpan, n - 1); // elements first..last (<= 65)
            if (!bef_encode_block(xs + first, (uint32_t)(last - first), B + j * 8)) OV[j] = 1;
        }
    });
    ovf.clear();
    n_overflow = 0;
    for (uint64_t j = 0; j < nb; ++j) {
        if (!over[j]) continue;
        uint64_t first = j << kBefShift, off = ovf.size();
        for (uint32_t t = 0; t <= kBefSpan; ++t) ovf.push_back(xs[std::min<uint64_t>(first + t, n - 1)]);
        blocks[j * 8] = off | (1ull << 63);
        ++n_overflow;
    }
    if (ovf.empty()) ovf.assign(2, 0);
    return MZ_OK;
}

// ------------------------------------------------------------------ MPHF
int Mphf::build(const uint64_t *keys_in, uint64_t n, double gamma, int n_threads) {
    n_keys = n;
    direct = 0;
    blocks.clear();
    levels.clear();
    std::vector<uint64_t> keys(keys_in, keys_in + n), next_keys;
    std::vector<uint64_t> occ, col;
    uint64_t total_blocks = 0;
    uint64_t lvl_seed = 0x9E3779B97F4A7C15ull;
    while (!keys.empty()) {
        if (levels.size() >= kMphfMaxLevels) {
            set_error("MPHF: more than %u levels (duplicate keys?)", kMphfMaxLevels);
            return MZ_EINVAL;
        }
        uint64_t rem = keys.size();
        uint64_t nb = (uint64_t)((gamma * (double)rem) / kMphfBitsPerBlock) + 1;
        if (nb > 0xFFFFFFFFull) {
            set_error("MPHF: level too large");
            return MZ_EINVAL;
        }
        lvl_seed = lvl_seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;
        MphfLevel L{total_blocks, lvl_seed, (uint32_t)nb, 0};
        occ.assign(nb * kMphfDataWords, 0);
        col.assign(nb * kMphfDataWords, 0);
        uint64_t *O = occ.data(), *Cc = col.data();
        const uint64_t *K = keys.data();
        parallel_for(n_threads, rem, [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                uint32_t blk, bit;
                mphf_level_slot(K[i], L.seed, L.n_blocks, blk, bit);
                uint64_t wi = (uint64_t)blk * kMphfDataWords + (bit >> 6), m = 1ull << (bit & 63);
                uint64_t old = __atomic_fetch_or(&O[wi], m, __ATOMIC_RELAXED);
                if (old & m) __atomic_fetch_or(&Cc[wi], m, __ATOMIC_RELAXED);
            }
        });
        parallel_for(n_threads, occ.size(), [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) O[i] &= ~Cc[i];
        });
        // keys whose bit did not survive go to the next level
        int nt = std::max(1, n_threads);
        std::vector<std::vector<uint64_t>> parts(nt);
        parallel_for(n_threads, rem, [&, O, K](int t, uint64_t b, uint64_t e) {
            auto &out = parts[t];
            for (uint64_t i = b; i < e; ++i) {
                uint32_t blk, bit;
                mphf_level_slot(K[i], L.seed, L.n_blocks, blk, bit);
                uint64_t wi = (uint64_t)blk * kMphfDataWords + (bit >> 6);
                if (!((O[wi] >> (bit & 63)) & 1)) out.push_back(K[i]);
            }
        });
        next_keys.clear();
        for (auto &p : parts) next_keys.insert(next_keys.end(), p.begin(), p.end());
        // append interleaved blocks (headers filled below)
        uint64_t base = blocks.size();
        blocks.resize(base + nb * 8, 0);
        uint64_t *B = blocks.data() + base;
        parallel_for(n_threads, nb, [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                uint64_t sub = 0, c = 0;
                for (uint32_t j = 0; j < kMphfDataWords; ++j) {
                    uint64_t wv = O[i * kMphfDataWords + j];
                    B[i * 8 + 2 + j] = wv;
                    sub |= c << (9 * j);
                    c += popc64(wv);
                }
                B[i * 8 + 1] = sub;
                B[i * 8 + 0] = c; // temporarily: ones in this block
            }
        });
        levels.push_back(L);
        total_blocks += nb;
        keys.swap(next_keys);
    }
    // absolute ranks
    uint64_t run = 0;
    for (uint64_t i = 0; i < total_blocks; ++i) {
        uint64_t c = blocks[i * 8];
        blocks[i * 8] = run;
        run += c;
    }
    if (run != n) {
        set_error("MPHF: placed %llu of %llu keys", (unsigned long long)run, (unsigned long long)n);
        return MZ_EINVAL;
    }
    if (blocks.empty()) blocks.assign(8, 0); // never hand out a null pointer
    return MZ_OK;
}

// negative filters of the minimizer records (mz_common.hpp): one bit per entry of every light bucket
void Mphf::fill_filters(const uint64_t *keys, uint64_t n, const uint64_t *psum, const uint64_t *entries, uint64_t skew_param, int n_threads) {
    uint64_t *B = blocks.data();
    parallel_for(n_threads, n, [=](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) { // bucket id of keys[i] is i (monotone)
            const uint64_t ps = psum[i], pe = psum[i + 1];
            if (pe - ps > skew_param) continue; // heavy: answered by the skew index, never scanned
            uint64_t *W = B + (uint64_t)((uint32_t)keys[i] >> kMinRecKeysLog2) * kMinRecWords;
            const uint32_t local = (uint32_t)(i - (W[0] & kMinRecRankMask));
            for (uint64_t j = ps; j < pe; ++j) {
                const uint64_t ent = entries[j];
                const uint32_t fl = (uint32_t)(ent & 0xFFFFu), g = ((ent >> kEntryFlankBits) & 1) ? flank_revcomp(fl) : fl;
                uint32_t word, bit;
                if (minrec_filter_slot(W[8], local, g, word, bit)) __atomic_fetch_or(&W[8 + word], 1ull << bit, __ATOMIC_RELAXED);
            }
        }
    });
}

int Mphf::build_direct(const uint64_t *keys, uint64_t n, uint32_t key_bits, int n_threads) {
    if (key_bits > 32) {
        set_error("direct MPHF: key universe of 2^%u too large", key_bits);
        return MZ_EINVAL;
    }
    n_keys = n;
    direct = 1;
    levels.clear();
    uint64_t universe = 1ull << key_bits;
    uint64_t nr = (universe >> kMinRecKeysLog2) + 1;
    blocks.assign(nr * kMinRecWords, 0);
    uint64_t *B = blocks.data();
    parallel_for(n_threads, n, [=](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            uint32_t x = (uint32_t)keys[i], bit = x & ((1u << kMinRecKeysLog2) - 1);
            __atomic_fetch_or(&B[(uint64_t)(x >> kMinRecKeysLog2) * kMinRecWords + minrec_bitmap_word(bit >> 6)], 1ull << (bit & 63), __ATOMIC_RELAXED);
        }
    });
    uint64_t run = 0;
    for (uint64_t j = 0; j < nr; ++j) {
        uint64_t *W = B + j * kMinRecWords;
        minrec_set_headers(W, run);
        run += minrec_members(W);
        B[j * kMinRecWords + 8] = 1ull << 63; // bucket-offset half: "consult the global structure" until fill_bucket_offsets()
    }
    if (run != n) {
        set_error("direct MPHF: %llu distinct keys of %llu", (unsigned long long)run, (unsigned long long)n);
        return MZ_EINVAL;
    }
    return MZ_OK;
}

// second half of every minimizer record: the bucket starts psum[rank0 .. rank0+members] of its members
uint64_t Mphf::fill_bucket_offsets(const uint64_t *psum, uint64_t n_psum, int n_threads) {
    uint64_t nr = blocks.size() / kMinRecWords;
    uint64_t *B = blocks.data();
    std::vector<uint64_t> over(std::max(1, n_threads), 0);
    uint64_t *OV = over.data();
    parallel_for(n_threads, nr, [=](int t, uint64_t b, uint64_t e) {
        for (uint64_t j = b; j < e; ++j) {
            uint64_t *W = B + j * kMinRecWords;
            uint32_t c = minrec_members(W);
            if (c == 0) continue;
            uint64_t rank0 = W[0] & kMinRecRankMask;
            bool ok = c <= kBefSpan && rank0 + c < n_psum && bef_encode_block(psum + rank0, c, W + 8);
            if (!ok) { W[8] = 1ull << 63; OV[t]++; }
            else W[8] |= minrec_filter_header(W[8], psum[rank0 + c] - psum[rank0]);
        }
    });
    uint64_t tot = 0;
    for (auto v : over) tot += v;
    return tot;
}

// ------------------------------------------------------------------ UnitigSet
static inline int base_code(unsigned char c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

int UnitigSet::finish() {
    if (k == 0 || k > 31) {
        set_error("UnitigSet: k=%u unsupported (1..31)", k);
        return MZ_EINVAL;
    }
    if (n_unitigs == 0 || total_len == 0) {
        set_error("UnitigSet: empty");
        return MZ_EEMPTY;
    }
    for (uint64_t i = 0; i < n_unitigs; ++i)
        if (starts[i + 1] <= starts[i]) {
            set_error("UnitigSet: unitig %llu is empty or offsets not increasing", (unsigned long long)i);
            return MZ_EINVAL;
        }
    // mz_k2upos carries unitig id, offset and length as 32-bit fields; positions take 36 bits in the device tiles
    if (n_unitigs > 0xFFFFFFFFull || total_len >= (1ull << 36)) {
        set_error("UnitigSet: %llu unitigs / %llu bases exceed the 2^32 unitigs / 2^36 bases this build addresses",
                  (unsigned long long)n_unitigs, (unsigned long long)total_len);
        return MZ_EINVAL;
    }
    for (uint64_t i = 0; i < n_unitigs; ++i)
        if (starts[i + 1] - starts[i] > 0xFFFFFFFFull) {
            set_error("UnitigSet: unitig %llu is longer than 2^32-1 bases", (unsigned long long)i);
            return MZ_EINVAL;
        }
    // accum_lens as Elias-Fano (unitig_set.rs:98), plus the blocked boundary vector
    int rc = accum_lens.build(starts.data(), n_unitigs + 1, starts[n_unitigs]);
    if (rc != MZ_OK) return rc;
    uint64_t nb = (total_len >> kBvBasesPerBlockLog2) + 1;
    bv_blocks.assign(nb * 8, 0);
    for (uint64_t i = 1; i <= n_unitigs; ++i) { // bit at the last base of each unitig, unitig_set.rs:90-93
        uint64_t p = starts[i] - 1;
        uint64_t b = p >> kBvBasesPerBlockLog2, off = p & (kBvBasesPerBlock - 1);
        bv_blocks[b * 8 + 4 + (off >> 6)] |= 1ull << (off & 63);
    }
    uint64_t uid = 0; // unitig holding the first base of the current block
    for (uint64_t b = 0; b < nb; ++b) {
        uint64_t first = b << kBvBasesPerBlockLog2;
        uint64_t lastb = std::min<uint64_t>(first + kBvBasesPerBlock - 1, total_len - 1);
        if (first >= total_len) { // tail block past the end: never probed by a valid position
            bv_blocks[b * 8 + 0] = n_unitigs;
            bv_blocks[b * 8 + 1] = total_len;
            bv_blocks[b * 8 + 2] = total_len;
            continue;
        }
        while (starts[uid + 1] <= first) ++uid;
        uint64_t ul = uid;
        while (starts[ul + 1] <= lastb) ++ul;
        bv_blocks[b * 8 + 0] = uid;
        bv_blocks[b * 8 + 1] = starts[uid];
        bv_blocks[b * 8 + 2] = starts[ul + 1];
    }
    build_seqbv();
    return MZ_OK;
}

// interleave the packed sequence and the unitig boundaries into 64-byte records (IndexView::seqbv, layout at seq_rec_of).
// The boundary bits of a record are cut out of the boundary bit vector (bv_blocks, the reference's `bv`: a bit at the
// last base of every unitig, unitig_set.rs:90-96); the header words come from the offsets.
void UnitigSet::build_seqbv() {
    const uint64_t nr = (total_len + kSeqRecStride - 1) / kSeqRecStride + 1; // + a tail record: never probed by a valid position
    seqbv.assign(nr * kSeqRecWords, 0);
    const uint64_t nw = useq.size(), n_bvb = bv_blocks.size() / 8;
    auto bv_word = [&](uint64_t wi) -> uint64_t { // 64 boundary bits starting at base 64 * wi
        const uint64_t blk = wi >> 2;
        return blk < n_bvb ? bv_blocks[blk * 8 + 4 + (wi & 3)] : 0;
    };
    uint64_t uid = 0; // unitig holding the first base of the current record
    for (uint64_t r = 0; r < nr; ++r) {
        uint64_t *W = seqbv.data() + r * kSeqRecWords;
        for (uint32_t j = 0; j < kSeqRecBases / 32; ++j) {
            const uint64_t wi = r * kSeqRecStrideWords + j;
            W[j] = wi < nw ? useq[wi] : 0;
        }
        const uint64_t first = r * kSeqRecStride;
        if (first >= total_len) {
            seq_rec_header(n_unitigs, total_len, total_len, W[4], W[5]);
            continue;
        }
        const uint64_t lastb = std::min<uint64_t>(first + kSeqRecBases - 1, total_len - 1);
        for (uint32_t j = 0; j < kSeqRecBases / 64; ++j) { // bits [first + 64 j, first + 64 j + 64): first is a multiple of 32
            const uint64_t b = first + 64ull * j, wi = b >> 6;
            const uint32_t sh = (uint32_t)(b & 63);
            W[6 + j] = sh ? (bv_word(wi) >> sh) | (bv_word(wi + 1) << (64 - sh)) : bv_word(wi);
        }
        while (starts[uid + 1] <= first) ++uid;
        uint64_t ul = uid; // -> the unitig holding the last base of the record
        while (starts[ul + 1] <= lastb) ++ul;
        seq_rec_header(uid, starts[uid], starts[ul + 1], W[4], W[5]);
    }
}

int UnitigSet::from_packed(const uint64_t *words, uint64_t total_len_, const uint64_t *starts_, uint64_t n, uint32_t k_) {
    if (!words || !starts_ || n == 0) {
        set_error("UnitigSet::from_packed: null or empty input");
        return MZ_EINVAL;
    }
    if (starts_[0] != 0 || starts_[n] != total_len_) {
        set_error("UnitigSet::from_packed: starts[0] must be 0 and starts[n] == total_len");
        return MZ_EINVAL;
    }
    k = k_;
    total_len = total_len_;
    n_unitigs = n;
    uint64_t nw = (total_len + 31) / 32;
    useq.assign(nw + 2, 0);
    std::memcpy(useq.data(), words, nw * 8);
    if (total_len & 31) useq[nw - 1] &= (1ull << (2 * (total_len & 31))) - 1; // zero the slack bits
    starts.assign(starts_, starts_ + n + 1);
    return finish();
}

// reference src/unitig_set.rs:74-106
int UnitigSet::from_seqs(const char *const *seqs, uint64_t n, uint32_t k_) {
    if (!seqs || n == 0) {
        set_error("UnitigSet::from_seqs: no sequences");
        return MZ_EEMPTY;
    }
    k = k_;
    n_unitigs = n;
    starts.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) starts[i + 1] = starts[i] + std::strlen(seqs[i]);
    total_len = starts[n];
    useq.assign((total_len + 31) / 32 + 2, 0);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t p = starts[i];
        for (const char *s = seqs[i]; *s; ++s, ++p) {
            int c = base_code((unsigned char)*s);
            if (c < 0) {
                set_error("UnitigSet::from_seqs: non-ACGT byte 0x%02x in sequence %llu", (unsigned char)*s, (unsigned long long)i);
                return MZ_EINVAL;
            }
            useq[p >> 5] |= (uint64_t)c << (2 * (p & 31));
        }
    }
    return finish();
}

// reference src/unitig_set.rs:119-165 (.cf_seg lines `id\tSEQ`; k from the cuttlefish json,
// cuttlefish.rs:45-108, key "parameters info" / "k")
int UnitigSet::from_cf_prefix(const std::string &prefix) {
    std::ifstream js(prefix + ".json");
    if (!js) {
        set_error("cannot open %s.json", prefix.c_str());
        return MZ_EIO;
    }
    std::string text((std::istreambuf_iterator<char>(js)), std::istreambuf_iterator<char>());
    size_t p = text.find("\"parameters info\"");
    size_t q = p == std::string::npos ? p : text.find("\"k\"", p);
    if (q == std::string::npos) {
        set_error("%s.json: no \"parameters info\".k", prefix.c_str());
        return MZ_EIO;
    }
    q = text.find(':', q);
    uint32_t kk = (uint32_t)std::strtoul(text.c_str() + q + 1, nullptr, 10);
    std::ifstream seg(prefix + ".cf_seg");
    if (!seg) {
        set_error("cannot open %s.cf_seg", prefix.c_str());
        return MZ_EIO;
    }
    std::vector<std::string> seqs;
    std::string line;
    while (std::getline(seg, line)) {
        if (line.empty()) continue;
        size_t tab = line.find('\t');
        if (tab == std::string::npos) {
            set_error("%s.cf_seg: cannot split line %zu", prefix.c_str(), seqs.size());
            return MZ_EIO;
        }
        seqs.emplace_back(line.substr(tab + 1));
    }
    std::vector<const char *> ptrs;
    for (auto &s : seqs) ptrs.push_back(s.c_str());
    return from_seqs(ptrs.data(), ptrs.size(), kk);
}

// ------------------------------------------------------------------ SSHash builder
struct Occ {
    uint64_t word, pos; // pos = (position << 1) | strand class for minimizer occurrences; plain position for skew tuples
};
static inline bool occ_less(const Occ &a, const Occ &b) { return a.word != b.word ? a.word < b.word : a.pos < b.pos; }

// Parallel sort of (word,pos) tuples: partition by the top bits of `word`, then sort partitions.
static void sort_occs(std::vector<Occ> &v, uint32_t word_bits, int n_threads) {
    uint64_t n = v.size();
    if (n_threads <= 1 || n < (1u << 16)) {
        std::sort(v.begin(), v.end(), occ_less);
        return;
    }
    uint32_t B = std::min<uint32_t>(word_bits, 12);
    uint32_t shift = word_bits - B;
    uint64_t nbk = 1ull << B;
    int nt = n_threads;
    std::vector<std::vector<uint64_t>> cnt(nt, std::vector<uint64_t>(nbk, 0));
    const Occ *src = v.data();
    parallel_for(nt, n, [&, src](int t, uint64_t b, uint64_t e) {
        auto &c = cnt[t];
        for (uint64_t i = b; i < e; ++i) c[src[i].word >> shift]++;
    });
    std::vector<uint64_t> bstart(nbk + 1, 0);
    {
        uint64_t acc = 0;
        for (uint64_t b = 0; b < nbk; ++b) {
            bstart[b] = acc;
            for (int t = 0; t < nt; ++t) {
                uint64_t c = cnt[t][b];
                cnt[t][b] = acc;
                acc += c;
            }
        }
        bstart[nbk] = acc;
    }
    std::vector<Occ> tmp(n);
    Occ *dst = tmp.data();
    parallel_for(nt, n, [&, src, dst](int t, uint64_t b, uint64_t e) {
        auto &c = cnt[t];
        for (uint64_t i = b; i < e; ++i) dst[c[src[i].word >> shift]++] = src[i];
    });
    parallel_tasks(nt, nbk, [&, dst](int, uint64_t b) { std::sort(dst + bstart[b], dst + bstart[b + 1], occ_less); });
    v.swap(tmp);
}

// Minimizer occurrences of one unitig (reference sshash.rs:100-143 via iter_canonical_minimizers):
// for every k-mer p the canonical strand g* decides which w-mer hashes compete; the occurrence is
// (w-mer word of g*, start of that w-mer on the FORWARD strand).  Consecutive duplicates are
// dropped per strand class, as the reference's two filtered passes do.
static void scan_unitig(const uint64_t *useq, uint64_t s, uint64_t e, uint32_t k, uint32_t w, uint64_t seed,
                        std::vector<Occ> &out) {
    if (e - s < k) return;
    const uint32_t n_win = k - w + 1;
    const uint64_t kmask = kmer_mask(k), wmask = kmer_mask(w);
    uint64_t hf[64], hr[64], wf[64], wr[64];
    uint64_t fwk = 0, rck = 0, fwm = 0, rcm = 0;
    uint64_t fmin = 0, rmin = 0; // absolute w-mer start of the current argmins
    bool have_f = false, have_r = false;
    Occ prev_f{~0ull, ~0ull}, prev_r{~0ull, ~0ull};
    for (uint64_t i = s; i < e; ++i) {
        uint64_t c = (useq[i >> 5] >> (2 * (i & 31))) & 3;
        fwk = ((fwk >> 2) | (c << (2 * (k - 1)))) & kmask;
        rck = ((rck << 2) | (3 - c)) & kmask;
        fwm = ((fwm >> 2) | (c << (2 * (w - 1)))) & wmask;
        rcm = ((rcm << 2) | (3 - c)) & wmask;
        if (i + 1 < s + w) continue;
        uint64_t j = i + 1 - w; // w-mer start
        uint32_t slot = (uint32_t)(j & 63);
        wf[slot] = fwm;
        wr[slot] = rcm;
        hf[slot] = hash_mmer(fwm, seed, w);
        hr[slot] = hash_mmer(rcm, seed, w);
        if (i + 1 < s + k) { // window not full yet: keep running argmins
            if (!have_f || hf[slot] < hf[fmin & 63]) { fmin = j; have_f = true; }
            if (!have_r || hr[slot] <= hr[rmin & 63]) { rmin = j; have_r = true; }
            continue;
        }
        uint64_t p = i + 1 - k; // k-mer start == first w-mer of the window
        if (fmin < p) { // expired: rescan, leftmost minimum
            fmin = p;
            for (uint64_t q = p + 1; q <= j; ++q)
                if (hf[q & 63] < hf[fmin & 63]) fmin = q;
        } else if (hf[slot] < hf[fmin & 63]) fmin = j;
        if (rmin < p) { // expired: rescan, rightmost minimum
            rmin = p;
            for (uint64_t q = p + 1; q <= j; ++q)
                if (hr[q & 63] <= hr[rmin & 63]) rmin = q;
        } else if (hr[slot] <= hr[rmin & 63]) rmin = j;
        (void)n_win;
        if (fwk <= rck) {
            Occ o{wf[fmin & 63], fmin << 1};
            if (o.word != prev_f.word || o.pos != prev_f.pos) out.push_back(o);
            prev_f = o;
        } else {
            Occ o{wr[rmin & 63], (rmin << 1) | 1};
            if (o.word != prev_r.word || o.pos != prev_r.pos) out.push_back(o);
            prev_r = o;
        }
    }
}

// Direct-address minimizer records (one 128-byte record per 256 possible w-mers) instead of BBHash levels: when the record
// table stays within max(64 MiB, 48 bytes per key, min(1 GiB, 1 KiB per key)).  The last term spends up to 1 GiB of the
// 288 GB on sets of half a million minimizers and more (a yeast-sized set at w = 15: 0.54 GB of records, 25.0 against
// 18.4 G k-mers/s through the BBHash levels); MZ_DIRECT_BYTES_PER_KEY replaces the rule by "at most that many bytes per key".
bool use_direct_records(uint32_t w, uint64_t n_mm, const BuildOpts &opts) {
    if (2 * w > 32) return false;
    const uint64_t rec_bytes = (((1ull << (2 * w)) >> kMinRecKeysLog2) + 1) * kMinRecWords * 8;
    if (opts.direct_bytes_per_key) return rec_bytes <= std::max<uint64_t>(64ull << 20, opts.direct_bytes_per_key * n_mm);
    const uint64_t free_cap = std::min<uint64_t>(1ull << 30, 1024 * n_mm);
    return rec_bytes <= std::max<uint64_t>(std::max<uint64_t>(64ull << 20, 48 * n_mm), free_cap);
}
BuildOpts BuildOpts::with_env() const {
    BuildOpts o = *this;
    if (o.minimizer_lines == 0)
        if (const char *e = std::getenv("MZ_FRONT_TABLE")) o.minimizer_lines = e[0] == '1' ? 1 : e[0] == '0' ? -1 : 0;
    if (o.lines_max_load == 0)
        if (const char *e = std::getenv("MZ_FRONT_MAX_LOAD")) { const double v = std::atof(e); if (v >= 1) o.lines_max_load = (uint32_t)v; }
    if (o.direct_bytes_per_key == 0)
        if (const char *e = std::getenv("MZ_DIRECT_BYTES_PER_KEY")) o.direct_bytes_per_key = strtoull(e, nullptr, 10);
    return o;
}
QueryOpts QueryOpts::from_env() {
    QueryOpts o;
    if (const char *e = std::getenv("MZ_K2U_KERNEL")) o.point_kernel = e[0] == 't' ? 2 : (e[0] == 'v' && e[1] == '1') ? 3 : 0;
    if (const char *e = std::getenv("MZ_READS_LINES")) o.reads_lines = e[0] == '0' ? -1 : 1;
    if (const char *e = std::getenv("MZ_READS_RESOLVE")) o.reads_resolve = e[0] == 'r' ? 2 : e[0] == 'i' ? 1 : 0;
    if (const char *e = std::getenv("MZ_READS_KERNEL")) o.reads_kernel = e[0] == 'b' ? 2 : e[0] == 'w' ? 1 : 0;
    if (const char *e = std::getenv("MZ_READS_VERIFY")) o.reads_verify = e[0] == 'g' ? 2 : e[0] == 'e' ? 1 : 0;
    if (const char *e = std::getenv("MZ_READS_GRID")) { const int v = std::atoi(e); if (v > 0) o.reads_grid = v; }
    return o;
}
uint32_t skew_fp_bits_for_build() {
    if (const char *e = std::getenv("MZ_TEST_SKEW_FP_BITS")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= (int)kSkewFpBits) return (uint32_t)v;
    }
    return kSkewFpBits;
}

void SkewTable::build(const uint64_t *words, const uint64_t *pos, uint64_t n, int n_threads) {
    n_keys = n;
    fp_bits = skew_fp_bits_for_build();
    n_buckets = n ? (n + kSkewLoad - 1) / kSkewLoad : 0;
    buckets.assign(std::max<uint64_t>(n_buckets * kSkewSlots, 2), kSkewEmpty);
    ovf_word.clear();
    ovf_pos.clear();
    n_ovf = 0;
    if (n == 0) return;
    // (home bucket, index) pairs in increasing order: inside a bucket the keys keep their word order
    std::vector<Occ> by(n);
    {
        Occ *BY = by.data();
        const uint64_t nb = n_buckets;
        const uint32_t fm = fp_mask();
        parallel_for(n_threads, n, [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                uint64_t hb;
                uint32_t fp;
                skew_home(words[i], nb, fm, hb, fp);
                BY[i] = Occ{hb, i};
            }
        });
    }
    sort_occs(by, bit_len(n_buckets), n_threads);
    uint64_t *B = buckets.data();
    uint64_t run_start = 0;
    for (uint64_t j = 0; j < n; ++j) { // serial: one pass, ranks inside the runs of equal buckets
        if (j && by[j].word != by[j - 1].word) run_start = j;
        const uint64_t rank = j - run_start, i = by[j].pos;
        uint64_t hb;
        uint32_t fp;
        skew_home(words[i], n_buckets, fp_mask(), hb, fp);
        if (rank < kSkewSlots) B[hb * kSkewSlots + rank] = skew_slot(fp, pos[i]);
        else { ovf_word.push_back(words[i]); ovf_pos.push_back(pos[i]); }
    }
    n_ovf = ovf_word.size();
}

bool FrontTable::choose(uint32_t w, uint64_t n_occ, bool direct, const BuildOpts &opts, uint32_t &g, uint32_t &t) {
    if (!direct || 2 * w > 32) return false; // the lines are addressed by key and built from the minimizer records
    if (opts.minimizer_lines < 0) return false;
    if (opts.lines_max_load >= 1) return front_shape_for_load(w, n_occ, opts.lines_max_load, g, t);
    if (opts.minimizer_lines > 0) return front_shape_for_load(w, n_occ, 10, g, t); // whatever the table size
    return front_choose_shape(w, n_occ, g, t);
}

bool FrontTable::next_shape(uint32_t w, uint64_t n_occ, const BuildOpts &opts, uint32_t &g, uint32_t &t) {
    if (t == 0) return false; // cannot happen: at most 2^g <= 16 keys share a line
    --t;
    return opts.minimizer_lines > 0 || (1ull << (2 * w - g - t)) * kFrontSlots * 8 <= front_table_cap(n_occ);
}

void FrontTable::build(const uint64_t *minrec, uint32_t w, uint64_t n_occ, bool direct, const uint64_t *psum, const uint64_t *entries, uint64_t skew_param,
                       int n_threads, const BuildOpts &opts) {
    lines.clear();
    present = choose(w, n_occ, direct, opts, g, t);
    while (present) {
        const uint32_t line_bits = 2 * w - g - t;
        const uint64_t n_lines = 1ull << line_bits;
        lines.assign(n_lines * kFrontSlots, 0);
        uint64_t *L = lines.data();
        const uint32_t gg = g, tt = t;
        std::vector<uint8_t> bad((size_t)std::max(n_threads, 1), 0);
        uint8_t *B = bad.data();
        parallel_for(n_threads, n_lines, [=](int th, uint64_t b, uint64_t e) {
            for (uint64_t line = b; line < e; ++line)
                if (!front_build_line(gg, tt, line_bits, line, minrec, psum, entries, skew_param, L + line * kFrontSlots)) B[th] = 1;
        });
        bool overfull = false;
        for (uint8_t x : bad) overfull = overfull || x;
        if (!overfull) return;
        present = next_shape(w, n_occ, opts, g, t); // some line has more than 16 member keys: fold the key space less
    }
    lines.clear();
    g = t = 0;
}

// size of the hash-addressed table (2w > 32): a power of two of lines at a mean load of at most 3 entries (else 5, else 10) within the same
// budget as the direct tables; mz_build_opts decide otherwise
bool FrontTable::choose_hashed(uint32_t w, uint64_t n_occ, uint64_t n_mm, const BuildOpts &opts, uint32_t &bits) {
    bits = 0;
    if (!front_is_hashed(w) || opts.minimizer_lines < 0 || n_mm == 0) return false;
    const bool forced = opts.minimizer_lines > 0 || opts.lines_max_load >= 1;
    const uint64_t loads[3] = {3, 5, 10};
    for (int a = 0; a < 3; ++a) {
        const uint64_t load = opts.lines_max_load >= 1 ? opts.lines_max_load : opts.minimizer_lines > 0 ? loads[2] : loads[a];
        bits = 6;
        while (bits < 32 && n_occ > load * (1ull << bits)) ++bits;
        if (forced || (kFrontSlots * 8ull << bits) <= front_table_cap(n_occ)) return true;
    }
    bits = 0;
    return false;
}

void FrontTable::build_hashed(const uint64_t *mm_set, const uint64_t *bucket_of, uint64_t n_mm, uint32_t w, uint64_t n_occ, const uint64_t *psum,
                              const uint64_t *entries, uint64_t skew_param, int n_threads, const BuildOpts &opts) {
    lines.clear();
    present = false;
    g = t = 0;
    n_unusable = n_walk = 0;
    uint32_t bits = 0;
    present = choose_hashed(w, n_occ, n_mm, opts, bits);
    if (!present) return;
    t = bits;
    const uint64_t n_lines = 1ull << bits;
    lines.assign(n_lines * kFrontSlots, kFrontEmpty);
    // the member words grouped by line, inside a line in word order
    std::vector<Occ> by(n_mm);
    {
        Occ *BY = by.data();
        parallel_for(n_threads, n_mm, [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) BY[i] = Occ{front_hashed_line(front_hash(mm_set[i]), n_lines), i};
        });
    }
    sort_occs(by, bits, n_threads);
    // shards of whole lines
    const uint64_t n_shards = std::min<uint64_t>((uint64_t)std::max(n_threads, 1) * 64, std::max<uint64_t>(1, n_mm / 4096));
    std::vector<uint64_t> bounds(n_shards + 1);
    for (uint64_t sh = 0; sh <= n_shards; ++sh) {
        uint64_t c = n_mm / n_shards * sh + std::min(sh, n_mm % n_shards);
        while (c > 0 && c < n_mm && by[c].word == by[c - 1].word) ++c; // a shard starts at the first word of a line
        bounds[sh] = c;
    }
    std::vector<uint64_t> bad((size_t)std::max(n_threads, 1), 0), walk((size_t)std::max(n_threads, 1), 0);
    uint64_t *L = lines.data(), *BAD = bad.data(), *WALK = walk.data();
    const Occ *BY = by.data();
    const uint64_t *BO = bounds.data();
    parallel_tasks(n_threads, n_shards, [=](int th, uint64_t sh) {
        uint64_t i = BO[sh];
        const uint64_t end = BO[sh + 1];
        while (i < end) {
            uint64_t j = i;
            while (j < end && BY[j].word == BY[i].word) ++j;
            uint64_t *out = L + BY[i].word * kFrontSlots;
            uint32_t keyfield[kFrontSlots];
            uint64_t first[kFrontSlots], size[kFrontSlots];
            const bool ok = j - i <= kFrontSlots;
            for (uint64_t a = i; a < j && ok; ++a) {
                const uint64_t mi = BY[a].pos, b = bucket_of[mi];
                keyfield[a - i] = front_hashed_keyfield(front_hash(mm_set[mi]));
                first[a - i] = psum[b];
                size[a - i] = psum[b + 1] - psum[b];
            }
            if (ok) {
                uint32_t nw = 0;
                const uint32_t m = front_merge_clashes((uint32_t)(j - i), keyfield, first, size, &nw); // member words under one fingerprint: one walk marker between them
                WALK[th] += nw;
                front_pack_line(m, keyfield, first, size, entries, skew_param, out);
            } else { out[0] = kFrontUnusable; ++BAD[th]; }
            i = j;
        }
    });
    for (uint64_t x : bad) n_unusable += x;
    for (uint64_t x : walk) n_walk += x;
}

int SSHashHost::build(const UnitigSet &unitigs, uint32_t w_, uint64_t skew_param_, uint64_t seed_, int n_threads, const BuildOpts &opts_) {
    const BuildOpts opts = opts_.with_env();
    double t0 = now_s();
    const bool timing = std::getenv("MZ_BUILD_TIMING") != nullptr;
    double tp = t0;
    auto phase = [&](const char *name) {
        if (timing) { double t = now_s(); std::fprintf(stderr, "[mz build] %-28s %7.2f s\n", name, t - tp); tp = t; }
    };
    if (n_threads <= 0) n_threads = hw_threads();
    if (w_ == 0 || w_ > unitigs.k) { // assert!(w <= unitigs.k()), sshash.rs:92
        set_error("SSHash: minimizer size w=%u must be in 1..=k (k=%u)", w_, unitigs.k);
        return MZ_EINVAL;
    }
    us = unitigs;
    w = w_;
    seed = seed_;
    skew_param = skew_param_;
    has_skew = false;
    const uint32_t k = us.k;
    const uint64_t *useq = us.useq.data();

    phase("copy unitig set");
    // 1. collect minimizer occurrences (sshash.rs:94-143), unitigs sharded over threads by bases
    int nt = n_threads;
    std::vector<std::vector<Occ>> parts(nt);
    {
        std::vector<uint64_t> cut(nt + 1, us.n_unitigs);
        cut[0] = 0;
        for (int t = 1; t < nt; ++t) {
            uint64_t target = us.total_len / nt * t;
            cut[t] = std::lower_bound(us.starts.begin(), us.starts.end() - 1, target) - us.starts.begin();
        }
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                auto &out = parts[t];
                for (uint64_t ui = cut[t]; ui < cut[t + 1]; ++ui)
                    scan_unitig(useq, us.starts[ui], us.starts[ui + 1], k, w, seed, out);
            });
        for (auto &x : th) x.join();
    }
    std::vector<Occ> occs;
    {
        uint64_t tot = 0;
        for (auto &p : parts) tot += p.size();
        occs.reserve(tot);
        for (auto &p : parts) {
            occs.insert(occs.end(), p.begin(), p.end());
            std::vector<Occ>().swap(p);
        }
    }
    if (occs.empty()) {
        set_error("SSHash: no k-mers (every unitig shorter than k=%u)", k);
        return MZ_EEMPTY;
    }
    phase("scan minimizers");
    // 2. sort by (word,pos), drop exact duplicates, run-length -> minimizer set (sshash.rs:145-172)
    sort_occs(occs, 2 * w, nt);
    occs.erase(std::unique(occs.begin(), occs.end(), [](const Occ &a, const Occ &b) { return a.word == b.word && a.pos == b.pos; }),
               occs.end());
    uint64_t n_occ = occs.size();
    std::vector<uint64_t> mm_set, range; // range[i] = first occurrence of minimizer i
    for (uint64_t i = 0; i < n_occ; ++i)
        if (i == 0 || occs[i].word != occs[i - 1].word) {
            mm_set.push_back(occs[i].word);
            range.push_back(i);
        }
    uint64_t n_mm = mm_set.size();
    range.push_back(n_occ);
    n_minimizers = n_mm;
    phase("sort + dedup + run-length");
    // 3. MPHF over the minimizer set, gamma = 1.7 (sshash.rs:177)
    // 2w <= 32: direct-address monotone MPHF (one line per probe, exact membership); else BBHash levels
    //   (only while the records stay small next to the key set: use_direct_records())
    const bool use_direct = use_direct_records(w, n_mm, opts);
    int rc = use_direct ? mm.build_direct(mm_set.data(), n_mm, 2 * w, nt) : mm.build(mm_set.data(), n_mm, 1.7, nt);
    if (rc != MZ_OK) return rc;
    phase("minimizer mphf");
    // 4. bucket sizes in hash order -> prefix sums (sshash.rs:179-189)
    std::vector<uint64_t> slot(n_mm), psum(n_mm + 1, 0);
    {
        MphfView mv = mm.view();
        uint64_t *S = slot.data(), *P = psum.data();
        const uint64_t *M = mm_set.data(), *R = range.data();
        const bool monotone = mm.direct != 0; // direct mode: bucket id == rank of the word == index in the sorted set
        parallel_for(nt, n_mm, [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                uint64_t h = i;
                if (!monotone) mphf_lookup(mv, M[i], h);
                S[i] = h;
                P[h + 1] = R[i + 1] - R[i];
            }
        });
        for (uint64_t i = 0; i < n_mm; ++i) psum[i + 1] += psum[i];
    }
    phase("bucket sizes + prefix");
    // 5. scatter positions into their buckets (sshash.rs:191-219)
    std::vector<uint64_t> posv(n_occ);
    uint64_t max_pos = 0;
    {
        uint64_t *PV = posv.data();
        const uint64_t *S = slot.data(), *P = psum.data(), *R = range.data();
        const Occ *O = occs.data();
        const uint64_t total_len_ = us.total_len;
        parallel_for(nt, n_mm, [=](int, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                uint64_t d = P[S[i]];
                for (uint64_t j = R[i]; j < R[i + 1]; ++j)
                    PV[d++] = make_entry(O[j].pos >> 1, (uint32_t)(O[j].pos & 1), flank_of(useq, total_len_, O[j].pos >> 1, w));
            }
        });
        max_pos = ~0ull; // entries are stored as whole 64-bit words: one aligned 8-byte load each (kPosEntryBits)
    }
    phase("scatter + flanks");
    // 6. skew index (sshash.rs:221-296): k-mers around every occurrence of a heavy minimizer
    if (skew_param != ~0ull) {
        std::vector<std::vector<Occ>> sparts(nt);
        const uint64_t *R = range.data();
        const Occ *O = occs.data();
        const uint64_t *bvb = us.bv_blocks.data();
        uint64_t total_len = us.total_len, sp = skew_param;
        parallel_for(nt, n_mm, [&, R, O, bvb, total_len, sp](int t, uint64_t b, uint64_t e) {
            auto &out = sparts[t];
            for (uint64_t i = b; i < e; ++i) {
                if (R[i + 1] - R[i] <= sp) continue;
                for (uint64_t j = R[i]; j < R[i + 1]; ++j) {
                    uint64_t mp = O[j].pos >> 1;
                    uint64_t start_pos = mp < (uint64_t)(k - w) ? 0 : mp - (k - w);
                    for (uint64_t off = 0; off < (uint64_t)(k - w + 1); ++off) {
                        uint64_t p = start_pos + off;
                        if (p + k > total_len) break;
                        UnitigSpan s = locate_unitig(bvb, p); // is_valid_useq_pos, unitig_set.rs:235-245
                        if (p + k > s.end) continue;
                        uint64_t x = read_kmer(useq, p, k), r = kmer_rc(x, k);
                        out.push_back(Occ{x < r ? x : r, p});
                    }
                }
            }
        });
        std::vector<Occ> tup;
        for (auto &p : sparts) {
            tup.insert(tup.end(), p.begin(), p.end());
            std::vector<Occ>().swap(p);
        }
        sort_occs(tup, 2 * k, nt);
        tup.erase(std::unique(tup.begin(), tup.end(), [](const Occ &a, const Occ &b) { return a.word == b.word; }), tup.end());
        uint64_t sn = tup.size();
        std::vector<uint64_t> km_set(sn), spos(sn);
        for (uint64_t i = 0; i < sn; ++i) { km_set[i] = tup[i].word; spos[i] = tup[i].pos; }
        std::vector<Occ>().swap(tup);
        skew.build(km_set.data(), spos.data(), sn, nt);
        has_skew = true;
    }
    phase("skew index");
    // 7. finish (sshash.rs:310-329): Elias-Fano the prefix sums, bit-pack the positions
    rc = buckets.build(psum.data(), n_mm + 1, nt);
    if (rc != MZ_OK) return rc;
    if (mm.direct) {
        n_minrec_overflow = mm.fill_bucket_offsets(psum.data(), n_mm + 1, nt);
        mm.fill_filters(mm_set.data(), n_mm, psum.data(), posv.data(), skew_param, nt);
    }
    pos.assign(posv.data(), n_occ, max_pos, nt);
    phase("EF + minrec + pack");
    if (front_is_hashed(w)) front.build_hashed(mm_set.data(), slot.data(), n_mm, w, n_occ, psum.data(), posv.data(), skew_param, nt, opts);
    else front.build(mm.blocks.data(), w, n_occ, mm.direct != 0, psum.data(), posv.data(), skew_param, nt, opts);
    phase("minimizer lines");
    build_seconds = now_s() - t0;
    return MZ_OK;
}

IndexView SSHashHost::view() const {
    IndexView v{};
    v.k = us.k;
    v.w = w;
    v.seed = seed;
    v.total_len = us.total_len;
    v.n_unitigs = us.n_unitigs;
    v.skew_param = skew_param;
    v.seqbv = us.seqbv.data();
    v.mm = mm.view();
    v.buckets = buckets.view();
    v.pos = pos.w.data();
    v.pos_width = pos.width;
    v.has_skew = has_skew ? 1 : 0;
    if (has_skew) {
        v.skew = skew.view();
    }
    v.front = front.view();
    v.starts = us.starts.data();
    return v;
}

// sshash.rs:341-358 (the MPHF term is the exact size of ours, not a bincode estimate)
uint64_t SSHashHost::num_bits() const { // from the scalars: the arrays may still be in HBM only
    uint64_t b = 64 + us.num_bits() + 64 * (n_bucket_block_words() + n_bucket_ovf_words()) + pos.num_bits() +
                 64 * n_mm_block_words() + mm.levels.size() * sizeof(MphfLevel) * 8;
    if (has_skew) b += skew.num_bits();
    return b + 64 * front.n_words(w);
}

// ------------------------------------------------------------------ flat file
static const char kMagic[8] = {'M', 'Z', 'S', 'S', 'H', '0', '1', '3'};
template <class T>
static void put(std::ofstream &f, const T &x) { f.write((const char *)&x, sizeof x); }
template <class T>
static void putv(std::ofstream &f, const std::vector<T> &v) {
    uint64_t n = v.size();
    put(f, n);
    f.write((const char *)v.data(), n * sizeof(T));
}
template <class T>
static bool get(std::ifstream &f, T &x) { return (bool)f.read((char *)&x, sizeof x); }
template <class T>
static bool getv(std::ifstream &f, std::vector<T> &v) { // the length is checked against what is left of the file before anything is allocated
    uint64_t n;
    if (!get(f, n)) return false;
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    if (here < 0 || end < here || n > (uint64_t)(end - here) / sizeof(T)) return false;
    v.resize(n);
    return (bool)f.read((char *)v.data(), n * sizeof(T));
}
static void put_mphf(std::ofstream &f, const Mphf &m) { put(f, m.n_keys); put(f, m.direct); putv(f, m.blocks); putv(f, m.levels); }
static bool get_mphf(std::ifstream &f, Mphf &m) { return get(f, m.n_keys) && get(f, m.direct) && getv(f, m.blocks) && getv(f, m.levels); }
static void put_ef(std::ofstream &f, const EFVector &e) { put(f, e.u); put(f, e.n); put(f, e.n_high_bits); put(f, e.l); putv(f, e.high); putv(f, e.sel); putv(f, e.low); }
static bool get_ef(std::ifstream &f, EFVector &e) { return get(f, e.u) && get(f, e.n) && get(f, e.n_high_bits) && get(f, e.l) && getv(f, e.high) && getv(f, e.sel) && getv(f, e.low); }
static void put_pv(std::ofstream &f, const PackedVec &p) { put(f, p.n); put(f, p.width); putv(f, p.w); }
static bool get_pv(std::ifstream &f, PackedVec &p) { return get(f, p.n) && get(f, p.width) && getv(f, p.w); }

int SSHashHost::save(const std::string &path) const {
    if (!arrays_on_host) {
        set_error("index arrays are resident in HBM only; bring them to the host first");
        return MZ_EINVAL;
    }
    std::ofstream f(path, std::ios::binary);
    if (!f) {
        set_error("cannot create %s", path.c_str());
        return MZ_EIO;
    }
    f.write(kMagic, 8);
    put(f, us.k); put(f, w); put(f, seed); put(f, skew_param); put(f, us.total_len); put(f, us.n_unitigs); put(f, n_minimizers);
    uint32_t hs = has_skew ? 1 : 0;
    put(f, hs);
    putv(f, us.useq); putv(f, us.starts); putv(f, us.bv_blocks); put_ef(f, us.accum_lens);
    put_mphf(f, mm); put(f, buckets.n); put(f, buckets.n_overflow); putv(f, buckets.blocks); putv(f, buckets.ovf); put_pv(f, pos);
    if (has_skew) { put(f, skew.n_buckets); put(f, skew.n_keys); put(f, skew.fp_bits); putv(f, skew.buckets); putv(f, skew.ovf_word); putv(f, skew.ovf_pos); }
    const uint32_t hf = front.present ? 1 : 0;
    put(f, hf);
    if (front.present) { put(f, front.g); put(f, front.t); putv(f, front.lines); }
    return f ? MZ_OK : (set_error("short write to %s", path.c_str()), MZ_EIO);
}

// every size and offset the kernels rely on is checked here, so that a truncated, stale or damaged file is refused
// instead of sending a kernel out of bounds; returns a description of the first inconsistency or nullptr
static const char *check_loaded(const SSHashHost &h) {
    const UnitigSet &us = h.us;
    if (us.k == 0 || us.k > 31) return "k out of range";
    if (h.w == 0 || h.w > us.k) return "w out of range";
    if (us.n_unitigs == 0 || us.n_unitigs > 0xFFFFFFFFull || us.total_len == 0 || us.total_len >= (1ull << 36)) return "unitig set out of range";
    if (us.useq.size() != (us.total_len + 31) / 32 + 2) return "packed sequence length";
    if (us.starts.size() != us.n_unitigs + 1 || us.starts[0] != 0 || us.starts[us.n_unitigs] != us.total_len) return "unitig offsets";
    for (uint64_t i = 0; i < us.n_unitigs; ++i)
        if (us.starts[i + 1] <= us.starts[i] || us.starts[i + 1] - us.starts[i] > 0xFFFFFFFFull) return "unitig offsets not increasing";
    if (us.bv_blocks.size() != ((us.total_len >> kBvBasesPerBlockLog2) + 1) * 8) return "boundary blocks";
    if (us.accum_lens.n != us.n_unitigs + 1 || us.accum_lens.l == 0 || us.accum_lens.l > 32 || us.accum_lens.sel.size() != (us.accum_lens.n >> kEfSelLog2) + 2 ||
        us.accum_lens.high.size() != us.accum_lens.n_high_bits / 64 + 4 || us.accum_lens.low.size() != (us.accum_lens.n * us.accum_lens.l + 63) / 64 + 2)
        return "unitig offset Elias-Fano";
    if (h.pos.width != kPosEntryBits || h.pos.w.size() != h.pos.n + 2 || h.pos.n == 0) return "entry array";
    if (h.n_minimizers == 0 || h.buckets.n != h.n_minimizers + 1 || h.buckets.blocks.size() != (((h.buckets.n - 1) >> kBefShift) + 1) * 8) return "bucket offsets";
    for (uint64_t j = 0; j < h.buckets.blocks.size(); j += 8) { // overflowed blocks point into the plain table
        const uint64_t hdr = h.buckets.blocks[j];
        if ((hdr >> 63) && (hdr & ((1ull << 40) - 1)) + kBefSpan + 1 > h.buckets.ovf.size()) return "bucket offset overflow table";
    }
    {
        uint64_t a, b;
        h.buckets.get_pair(h.n_minimizers - 1, a, b);
        if (b != h.pos.n || a > b) return "bucket offsets do not end at the entry count";
    }
    if (h.mm.n_keys != h.n_minimizers) return "minimizer count";
    if (h.mm.direct) {
        if (2 * h.w > 32 || !h.mm.levels.empty() || h.mm.blocks.size() != (((1ull << (2 * h.w)) >> kMinRecKeysLog2) + 1) * kMinRecWords) return "minimizer records";
        uint64_t run = 0;
        for (uint64_t j = 0; j < h.mm.blocks.size(); j += kMinRecWords) {
            const uint64_t *W = h.mm.blocks.data() + j;
            if ((W[0] & kMinRecRankMask) != run) return "minimizer record ranks";
            const uint32_t c = minrec_members(W);
            run += c;
            if (c && !(W[8] >> 63)) { // bucket starts of the record's members: the last one bounds them all
                uint64_t a = 0, b = 0;
                if (c > kBefSpan || !bef_block_pair(W + 8, c - 1, nullptr, a, b) || b > h.pos.n || a > b) return "minimizer record offsets";
            }
        }
        if (run != h.n_minimizers) return "minimizer record ranks";
    } else {
        if (h.mm.blocks.size() % 8 != 0 || h.mm.levels.size() > kMphfMaxLevels) return "minimizer MPHF";
        for (const MphfLevel &L : h.mm.levels)
            if (L.n_blocks == 0 || L.block_off + L.n_blocks > h.mm.blocks.size() / 8) return "minimizer MPHF levels";
    }
    if (h.has_skew) {
        const SkewTable &t = h.skew;
        if (t.fp_bits < 1 || t.fp_bits > kSkewFpBits || t.buckets.size() != std::max<uint64_t>(t.n_buckets * kSkewSlots, 2) || t.ovf_word.size() != t.ovf_pos.size()) return "skew table";
        const uint64_t last = us.total_len >= us.k ? us.total_len - us.k : 0, pmask = (1ull << kSkewPosBits) - 1;
        for (uint64_t e : t.buckets)
            if (e != kSkewEmpty && (e & pmask) > last) return "skew table position";
        for (uint64_t p : t.ovf_pos)
            if (p > last) return "skew overflow position";
    }
    if (h.front.present) { // every slot must point inside the entry array / the sequence
        const FrontTable &t = h.front;
        if (front_is_hashed(h.w)) {
            if (h.mm.direct || t.g != 0 || t.t > 32 || t.lines.size() != ((uint64_t)kFrontSlots << t.t)) return "minimizer lines (hashed)";
        } else if (!h.mm.direct || t.g > kFrontMaxG || t.g + t.t > kFrontMaxKeyBits || t.g + t.t > 2 * h.w || t.lines.size() != (1ull << (2 * h.w - t.g - t.t)) * kFrontSlots) return "minimizer lines";
        for (uint64_t e : t.lines) {
            const uint32_t kind = (uint32_t)e & 3u;
            const uint64_t a = e >> kFrontPosShift;
            if (kind == kFrontEntry && a >= us.total_len) return "minimizer line position";
            if (kind == kFrontSpill && a + ((e >> kFrontFlankShift) & 0xFFFF) > h.pos.n && ((e >> kFrontFlankShift) & 0xFFFF) != 0xFFFF) return "minimizer line bucket range";
            if (kind == kFrontSpill && a >= h.pos.n) return "minimizer line bucket range";
        }
    }
    return nullptr;
}

int SSHashHost::load(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    char magic[8];
    if (!f || !f.read(magic, 8) || std::memcmp(magic, kMagic, 8) != 0) {
        set_error("%s: not a mazu_amd SSHash file (or one of another format version)", path.c_str());
        return MZ_EIO;
    }
    uint32_t hs = 0;
    bool ok = get(f, us.k) && get(f, w) && get(f, seed) && get(f, skew_param) && get(f, us.total_len) && get(f, us.n_unitigs) &&
              get(f, n_minimizers) && get(f, hs) && getv(f, us.useq) && getv(f, us.starts) && getv(f, us.bv_blocks) &&
              get_ef(f, us.accum_lens) && get_mphf(f, mm) && get(f, buckets.n) && get(f, buckets.n_overflow) && getv(f, buckets.blocks) && getv(f, buckets.ovf) && get_pv(f, pos);
    has_skew = hs != 0;
    if (ok && has_skew) ok = get(f, skew.n_buckets) && get(f, skew.n_keys) && get(f, skew.fp_bits) && getv(f, skew.buckets) && getv(f, skew.ovf_word) && getv(f, skew.ovf_pos);
    uint32_t hf = 0;
    ok = ok && get(f, hf);
    front.present = hf != 0;
    front.lines.clear();
    if (ok && front.present) ok = get(f, front.g) && get(f, front.t) && getv(f, front.lines);
    if (!ok) {
        set_error("%s: truncated", path.c_str());
        return MZ_EIO;
    }
    skew.n_ovf = has_skew ? skew.ovf_word.size() : 0;
    arrays_on_host = true;
    if (const char *what = check_loaded(*this)) {
        set_error("%s: inconsistent index file (%s)", path.c_str(), what);
        return MZ_EIO;
    }
    us.build_seqbv();
    return MZ_OK;
}

} // namespace mz
