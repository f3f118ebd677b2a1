// mz_build_gpu.hip -- index construction on the device.
//
// Produces, array for array and bit for bit, what SSHashHost::build (mz_host.cpp, the restatement of
// reference src/kphf/sshash.rs:86-329 in this repository's layout) produces on the host:
//   G1  minimizer occurrences of every k-mer (sliding-window minimizer, w-mer hashes staged in LDS), one
//       64-bit key (word | position | strand class) per run of equal keys, wave-ballot compacted
//   G2  radix sort + unique (rocPRIM device primitives), run-length -> minimizer set, bucket sizes,
//       exclusive scan -> bucket starts
//   G3  minimizer records (membership bitmap, ranks, per-record Elias-Fano halves), global partitioned
//       Elias-Fano, 64-bit `pos` entries with strand class and flank fingerprint, negative filters of the records
//   G4  skew index: k-mers around heavy minimizers, sort/unique, hash table of 128-byte buckets (stable sort by home
//       bucket, rank inside the bucket, overflow list)
// 2w <= 32 (direct-address minimizer records): an occurrence (word, position, strand class) must fit one 64-bit sort key:
// 2w + bit_len(total_len) + 1 <= 64, or one bit more for odd w, where the class is implied (no w-mer of odd length equals its reverse
// complement, so the class is 1 exactly when the forward w-mer at the position differs from the word) and is left out of the key;
// otherwise the caller falls back to the host builder.
// 2w > 32 ("wide"): the occurrence is kept as two words (word; position << 1 | class) and sorted by two stable radix passes; the
// minimizer MPHF is the BBHash of Mphf::build, built level by level on the device (the bit arrays of a level do not depend on the order
// of the keys, so they equal the host's); buckets are numbered by the MPHF, so sizes and entries are scattered through the slot of every
// minimizer; the minimizer lines are the hash-addressed ones of FrontTable::build_hashed.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mz_device.hpp"

namespace mz {

// hipErrorOutOfMemory is reported as MZ_ENOMEM (and cleared), so the caller can fall back to the host builder
#define BG_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
            if (_e == hipErrorOutOfMemory) { (void)hipGetLastError(); return MZ_ENOMEM; }             \
            return MZ_EHIP;                                                                           \
        }                                                                                             \
    } while (0)

namespace {

constexpr int kB = 256, kQ = 4, kT = kB * kQ; // scan tile: 1024 k-mer start positions

struct DevBuf { // frees everything it allocated when it goes out of scope
    std::vector<void *> ptrs;
    ~DevBuf() { for (void *p : ptrs) (void)hipFree(p); }
    template <class T>
    hipError_t alloc(T **p, uint64_t count) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<uint64_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(q);
        *p = (T *)q;
        return e;
    }
    void release(void *p) {
        for (auto &q : ptrs) if (q == p) { (void)hipFree(p); q = nullptr; }
    }
    // hand a buffer over to `d` (it is no longer freed here)
    void give(void *p, DeviceIndex *d, uint64_t bytes) {
        for (auto &q : ptrs) if (q == p) { q = nullptr; d->allocs.push_back(p); d->bytes += bytes; return; }
    }
};

struct ScanParams {
    const uint64_t *useq;   // plain packed sequence (+2 pad words)
    const uint64_t *seqbv;  // interleaved records (unitig boundaries)
    uint64_t total_len, seed;
    uint32_t k, w, key_shift; // key = (word << key_shift) | (pos << cls_bits) | (cls if cls_bits)
    uint32_t cls_bits;        // 1: strand class kept in the key's lowest bit; 0: implied (odd w only), see key_pos_cls()
};
// position and strand class of an occurrence key
__device__ __forceinline__ void key_pos_cls(const ScanParams &P, uint64_t key, uint64_t &pos, uint32_t &cls) {
    const uint64_t pc = key & ((1ull << P.key_shift) - 1);
    pos = pc >> P.cls_bits;
    cls = P.cls_bits ? (uint32_t)(pc & 1) : (uint32_t)(read_kmer(P.useq, pos, P.w) != (key >> P.key_shift));
}

// G1: one key per k-mer start position; a key is kept when it differs from its left neighbour's (the
// reference drops consecutive duplicates per strand class, sshash.rs:109-134; the unique pass after the sort
// removes what is left).  STORE == false only counts.
// WIDE: out_keys receives the minimizer word and out_pc (position << 1) | strand class
template <bool STORE, bool WIDE>
__global__ __launch_bounds__(kB) void scan_minimizers_kernel(ScanParams P, uint64_t n_tiles, uint64_t *__restrict__ out_keys, uint64_t *__restrict__ out_pc,
                                                             unsigned long long *counter) {
    __shared__ uint64_t hf[kT + 32], hr[kT + 32];
    __shared__ uint64_t keys[kT];
    __shared__ uint64_t pcs[WIDE ? kT : 1];
    __shared__ uint32_t s_wtot[kB / 64];
    __shared__ unsigned long long s_base;
    const uint32_t t = threadIdx.x, k = P.k, w = P.w, n_win = k - w + 1;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t P0 = tile * kT;
        for (uint32_t j = t; j < kT + n_win - 1; j += kB) { // w-mer hashes of the tile, both strands
            uint64_t p = P0 + j;
            uint64_t m = p + w <= P.total_len ? read_kmer(P.useq, p, w) : 0;
            hf[j] = hash_mmer(m, P.seed, w);
            hr[j] = hash_mmer(kmer_rc(m, w), P.seed, w);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const uint32_t i = q * kB + t;
            const uint64_t p = P0 + i;
            uint64_t key = ~0ull, pc = 0;
            if (p + k <= P.total_len) {
                UnitigSpan s = locate_unitig_i(P.seqbv, p);
                if (p + k <= s.end) {
                    const uint64_t fw = read_kmer(P.useq, p, k), rc = kmer_rc(fw, k);
                    if (fw <= rc) { // forward-canonical: leftmost minimum of the forward hashes
                        uint32_t bo = 0;
                        uint64_t bh = hf[i];
                        for (uint32_t o = 1; o < n_win; ++o) { uint64_t h = hf[i + o]; if (h < bh) { bh = h; bo = o; } }
                        if (WIDE) { key = (fw >> (2 * bo)) & kmer_mask(w); pc = (p + bo) << 1; }
                        else key = (((fw >> (2 * bo)) & kmer_mask(w)) << P.key_shift) | ((p + bo) << P.cls_bits);
                    } else { // reverse-canonical: rightmost minimum of the reverse hashes; word of g* = rc of the forward w-mer
                        uint32_t bj = 0;
                        uint64_t bh = hr[i];
                        for (uint32_t j = 1; j < n_win; ++j) { uint64_t h = hr[i + j]; if (h <= bh) { bh = h; bj = j; } }
                        const uint32_t og = (k - w) - bj; // offset inside g* = rc
                        if (WIDE) { key = (rc >> (2 * og)) & kmer_mask(w); pc = ((p + bj) << 1) | 1; }
                        else key = (((rc >> (2 * og)) & kmer_mask(w)) << P.key_shift) | ((p + bj) << P.cls_bits) | P.cls_bits;
                    }
                }
            }
            keys[i] = key;
            if (WIDE) pcs[i] = pc;
        }
        __syncthreads();
        // compaction: ONE global atomic per tile (a same-address atomic costs ~12 ns; one per wave and slice was 0.47 s at
        // S-human), waves take consecutive ranges of the tile's reservation
        unsigned long long bal[kQ];
        uint32_t wave_total = 0;
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const uint32_t i = q * kB + t;
            const uint64_t key = keys[i];
            const bool emit = key != ~0ull && (i == 0 || keys[i - 1] != key || (WIDE && pcs[i - 1] != pcs[i]));
            bal[q] = __ballot(emit);
            wave_total += (uint32_t)__popcll(bal[q]);
        }
        const unsigned lane = __lane_id(), wave = t >> 6;
        if (lane == 0) s_wtot[wave] = wave_total;
        __syncthreads();
        if (t == 0) {
            unsigned long long tot = 0;
            for (uint32_t wv = 0; wv < kB / 64; ++wv) tot += s_wtot[wv];
            s_base = tot ? atomicAdd(counter, tot) : 0ull;
        }
        __syncthreads();
        if (STORE) {
            unsigned long long base = s_base;
            for (uint32_t wv = 0; wv < wave; ++wv) base += s_wtot[wv];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                if ((bal[q] >> lane) & 1) {
                    const unsigned long long o = base + __popcll(bal[q] & ((1ull << lane) - 1));
                    out_keys[o] = keys[q * kB + t];
                    if (WIDE) out_pc[o] = pcs[q * kB + t];
                }
                base += (unsigned long long)__popcll(bal[q]);
            }
        }
        __syncthreads();
    }
}

struct KeyToWord {
    uint32_t shift;
    __host__ __device__ uint64_t operator()(uint64_t key) const { return key >> shift; }
};

__global__ __launch_bounds__(kB) void set_member_bits_kernel(const uint64_t *__restrict__ words, uint64_t n, uint64_t *minrec) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
        uint32_t x = (uint32_t)words[i], bit = x & ((1u << kMinRecKeysLog2) - 1);
        atomicOr((unsigned long long *)&minrec[(uint64_t)(x >> kMinRecKeysLog2) * kMinRecWords + minrec_bitmap_word(bit >> 6)], 1ull << (bit & 63));
    }
}
__global__ __launch_bounds__(kB) void record_counts_kernel(const uint64_t *__restrict__ minrec, uint64_t n_rec, uint64_t *__restrict__ counts) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n_rec; j += stride) {
        const uint64_t *W = minrec + j * kMinRecWords;
        counts[j] = minrec_members(W);
    }
}
// ranks + Elias-Fano half of every minimizer record (Mphf::build_direct + fill_bucket_offsets of the host)
__global__ __launch_bounds__(kB) void finish_records_kernel(uint64_t *minrec, uint64_t n_rec, const uint64_t *__restrict__ rank0, const uint64_t *__restrict__ counts,
                                                            const uint64_t *__restrict__ psum, uint64_t n_psum, unsigned long long *n_overflow) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n_rec; j += stride) {
        uint64_t *W = minrec + j * kMinRecWords;
        const uint64_t r0 = rank0[j];
        const uint32_t c = (uint32_t)counts[j];
        minrec_set_headers(W, r0);
        W[8] = 1ull << 63;
        if (c == 0) continue;
        uint64_t xs[kBefSpan + 1], enc[8];
        bool ok = c <= kBefSpan && r0 + c < n_psum;
        if (ok) {
            for (uint32_t q = 0; q <= c; ++q) xs[q] = psum[r0 + q];
            ok = bef_encode_block(xs, c, enc);
        }
        if (ok) {
            enc[0] |= minrec_filter_header(enc[0], xs[c] - xs[0]);
            for (int q = 0; q < 8; ++q) W[8 + q] = enc[q];
        } else {
            atomicAdd(n_overflow, 1ull);
        }
    }
}
// negative filters of the records (Mphf::fill_filters of the host): one lane per minimizer walks its light bucket
__global__ __launch_bounds__(kB) void record_filters_kernel(uint64_t *minrec, const uint64_t *__restrict__ words, uint64_t n_mm, const uint64_t *__restrict__ psum,
                                                            const uint64_t *__restrict__ entries, uint64_t skew_param) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_mm; i += stride) {
        const uint64_t ps = psum[i], pe = psum[i + 1];
        if (pe - ps > skew_param) continue;
        uint64_t *W = minrec + (uint64_t)((uint32_t)words[i] >> kMinRecKeysLog2) * kMinRecWords;
        const uint32_t local = (uint32_t)(i - (W[0] & kMinRecRankMask));
        const uint64_t hdr = W[8];
        for (uint64_t j = ps; j < pe; ++j) {
            const uint64_t ent = entries[j];
            const uint32_t fl = (uint32_t)(ent & 0xFFFFu), g = ((ent >> kEntryFlankBits) & 1) ? flank_revcomp(fl) : fl;
            uint32_t word, bit;
            if (minrec_filter_slot(hdr, local, g, word, bit)) atomicOr((unsigned long long *)&W[8 + word], 1ull << bit);
        }
    }
}
// global partitioned Elias-Fano over psum (BlockedEF::build): block encode + overflow flag
__global__ __launch_bounds__(kB) void bef_blocks_kernel(const uint64_t *__restrict__ psum, uint64_t n, uint64_t n_blocks, uint64_t *__restrict__ blocks,
                                                        uint64_t *__restrict__ ovflag) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n_blocks; j += stride) {
        uint64_t first = j << kBefShift, last = first + kBefSpan < n - 1 ? first + kBefSpan : n - 1;
        uint64_t xs[kBefSpan + 1], enc[8];
        uint32_t cnt = (uint32_t)(last - first);
        for (uint32_t q = 0; q <= cnt; ++q) xs[q] = psum[first + q];
        bool ok = bef_encode_block(xs, cnt, enc);
        for (int q = 0; q < 8; ++q) blocks[j * 8 + q] = ok ? enc[q] : 0;
        ovflag[j] = ok ? 0 : 1;
    }
}
__global__ __launch_bounds__(kB) void bef_overflow_kernel(const uint64_t *__restrict__ psum, uint64_t n, uint64_t n_blocks, const uint64_t *__restrict__ ovflag,
                                                          const uint64_t *__restrict__ ovrank, uint64_t *blocks, uint64_t *__restrict__ ovf) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n_blocks; j += stride) {
        if (!ovflag[j]) continue;
        uint64_t first = j << kBefShift, off = ovrank[j] * (kBefSpan + 1);
        for (uint32_t q = 0; q <= kBefSpan; ++q) ovf[off + q] = psum[first + q < n - 1 ? first + q : n - 1];
        blocks[j * 8] = off | (1ull << 63);
    }
}
// packed `pos` entries (PackedVec::assign of make_entry(pos, cls, flank)): one lane per OUTPUT word, which gathers the
// (at most three) entries overlapping it -- plain stores, no atomics on words shared by neighbouring lanes
__global__ __launch_bounds__(kB) void pos_entries_kernel(const uint64_t *__restrict__ keys, uint64_t n, ScanParams P, uint32_t width, uint64_t n_words,
                                                         uint64_t *__restrict__ packed) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t wi = (uint64_t)blockIdx.x * kB + threadIdx.x; wi < n_words; wi += stride) {
        const uint64_t bit0 = wi * 64;
        uint64_t word = 0;
        for (uint64_t i = bit0 / width; i < n && i * width < bit0 + 64; ++i) {
            uint64_t pos;
            uint32_t cls;
            key_pos_cls(P, keys[i], pos, cls);
            const uint64_t ent = make_entry(pos, cls, flank_of(P.useq, P.total_len, pos, P.w));
            const uint64_t b = i * width;
            word |= b >= bit0 ? ent << (b - bit0) : ent >> (bit0 - b);
        }
        packed[wi] = word;
    }
}

// minimizer lines (FrontTable::build of the host): one lane per line; *overfull is raised when a line has more than 16 member keys
__global__ __launch_bounds__(kB) void front_lines_kernel(const uint64_t *__restrict__ minrec, uint32_t g, uint32_t t, uint32_t line_bits,
                                                         const uint64_t *__restrict__ psum, const uint64_t *__restrict__ entries, uint64_t skew_param,
                                                         uint64_t *__restrict__ lines, unsigned long long *overfull) {
    const uint64_t n_lines = 1ull << line_bits, stride = (uint64_t)gridDim.x * kB;
    for (uint64_t line = (uint64_t)blockIdx.x * kB + threadIdx.x; line < n_lines; line += stride) {
        uint64_t out[kFrontSlots];
        if (!front_build_line(g, t, line_bits, line, minrec, psum, entries, skew_param, out)) *overfull = 1ull; // (any number of lanes may raise it)
#pragma unroll
        for (uint32_t q = 0; q < kFrontSlots; ++q) lines[line * kFrontSlots + q] = out[q];
    }
}

// ---- skew index ----
// tuples around every occurrence of a heavy minimizer (sshash.rs:231-262): one lane per (occurrence, offset)
template <bool STORE>
__global__ __launch_bounds__(kB) void skew_tuples_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ run_start, const uint64_t *__restrict__ run_len,
                                                         uint64_t n_mm, ScanParams P, uint64_t skew_param, uint64_t *__restrict__ out_word,
                                                         uint64_t *__restrict__ out_pos, unsigned long long *counter) {
    // grid-stride over minimizers; heavy ones are walked by the whole workgroup
    const uint64_t pmask = (1ull << P.key_shift) - 1;
    const uint32_t k = P.k, w = P.w, span = k - w + 1;
    for (uint64_t i = blockIdx.x; i < n_mm; i += gridDim.x) {
        const uint64_t len = run_len[i];
        if (len <= skew_param) continue;
        const uint64_t s = run_start[i], items = len * span;
        for (uint64_t x0 = 0; x0 < items; x0 += kB) {
            const uint64_t x = x0 + threadIdx.x;
            bool emit = false;
            uint64_t word = 0, p = 0;
            if (x < items) {
                const uint64_t occ = x / span, off = x - occ * span;
                const uint64_t mp = (keys[s + occ] & pmask) >> P.cls_bits;
                const uint64_t start_pos = mp < (uint64_t)(k - w) ? 0 : mp - (k - w);
                p = start_pos + off;
                if (p + k <= P.total_len) {
                    UnitigSpan sp = locate_unitig_i(P.seqbv, p);
                    if (p + k <= sp.end) {
                        const uint64_t fw = read_kmer(P.useq, p, k), rc = kmer_rc(fw, k);
                        word = fw < rc ? fw : rc;
                        emit = true;
                    }
                }
            }
            unsigned long long b = __ballot(emit);
            if (b) {
                unsigned lane = __lane_id(), leader = (unsigned)(__ffsll((long long)b) - 1);
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(b));
                base = __shfl(base, (int)leader);
                if (STORE && emit) {
                    unsigned long long slot = base + __popcll(b & ((1ull << lane) - 1));
                    out_word[slot] = word;
                    out_pos[slot] = p;
                }
            }
        }
    }
}

// ---- skew table (SkewTable::build of the host) ----
__global__ __launch_bounds__(kB) void skew_home_kernel(const uint64_t *__restrict__ words, uint64_t n, uint64_t n_buckets, uint32_t fp_mask,
                                                       uint64_t *__restrict__ hb, uint64_t *__restrict__ idx) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
        uint64_t b;
        uint32_t fp;
        skew_home(words[i], n_buckets, fp_mask, b, fp);
        hb[i] = b;
        idx[i] = i;
    }
}
__global__ __launch_bounds__(kB) void skew_heads_kernel(const uint64_t *__restrict__ hb, uint64_t n, uint64_t *__restrict__ head) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n; j += stride) head[j] = (j == 0 || hb[j] != hb[j - 1]) ? j : 0;
}
__global__ __launch_bounds__(kB) void skew_place_kernel(const uint64_t *__restrict__ hb, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ first,
                                                        const uint64_t *__restrict__ words, const uint64_t *__restrict__ pos, uint64_t n, uint64_t n_buckets,
                                                        uint32_t fp_mask, uint64_t *__restrict__ table, uint8_t *__restrict__ ovflag) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n; j += stride) {
        const uint64_t rank = j - first[j], i = idx[j];
        uint64_t b;
        uint32_t fp;
        skew_home(words[i], n_buckets, fp_mask, b, fp);
        if (rank < kSkewSlots) table[hb[j] * kSkewSlots + rank] = skew_slot(fp, pos[i]);
        ovflag[j] = rank >= kSkewSlots;
    }
}
__global__ __launch_bounds__(kB) void gather_kernel(const uint64_t *__restrict__ src, const uint64_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ dst) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n; j += stride) dst[j] = src[idx[j]];
}

__global__ __launch_bounds__(kB) void first_of_run_kernel(const uint64_t *__restrict__ words, uint64_t n, uint8_t *__restrict__ flag) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) flag[i] = i == 0 || words[i] != words[i - 1];
}

// ---- wide path (2w > 32) ----
__global__ __launch_bounds__(kB) void first_of_pair_run_kernel(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint64_t n, uint8_t *__restrict__ flag) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) flag[i] = i == 0 || a[i] != a[i - 1] || b[i] != b[i - 1];
}
// one BBHash level (Mphf::build): occupied / collided bits of every key's slot
__global__ __launch_bounds__(kB) void mphf_mark_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t seed, uint32_t n_blocks, unsigned long long *occ,
                                                       unsigned long long *col) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
        uint32_t blk, bit;
        mphf_level_slot(keys[i], seed, n_blocks, blk, bit);
        const uint64_t wi = (uint64_t)blk * kMphfDataWords + (bit >> 6);
        const unsigned long long m = 1ull << (bit & 63);
        const unsigned long long old = atomicOr(&occ[wi], m);
        if (old & m) atomicOr(&col[wi], m);
    }
}
// interleaved blocks of the level (word 0: ones in the block, replaced by the absolute rank at the end; word 1: six 9-bit sub-counts)
__global__ __launch_bounds__(kB) void mphf_blocks_kernel(uint64_t *occ, const uint64_t *__restrict__ col, uint64_t n_blocks, uint64_t *__restrict__ B) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_blocks; i += stride) {
        uint64_t sub = 0, c = 0;
        for (uint32_t j = 0; j < kMphfDataWords; ++j) {
            const uint64_t wv = occ[i * kMphfDataWords + j] & ~col[i * kMphfDataWords + j];
            occ[i * kMphfDataWords + j] = wv;
            B[i * 8 + 2 + j] = wv;
            sub |= c << (9 * j);
            c += popc64(wv);
        }
        B[i * 8 + 1] = sub;
        B[i * 8 + 0] = c;
    }
}
// keys whose bit did not survive go to the next level
__global__ __launch_bounds__(kB) void mphf_unplaced_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t seed, uint32_t n_blocks, const uint64_t *__restrict__ occ,
                                                           uint8_t *__restrict__ flag) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
        uint32_t blk, bit;
        mphf_level_slot(keys[i], seed, n_blocks, blk, bit);
        flag[i] = !((occ[(uint64_t)blk * kMphfDataWords + (bit >> 6)] >> (bit & 63)) & 1);
    }
}
__global__ __launch_bounds__(kB) void mphf_counts_kernel(const uint64_t *__restrict__ B, uint64_t n_blocks, uint64_t *__restrict__ counts) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_blocks; i += stride) counts[i] = B[i * 8];
}
__global__ __launch_bounds__(kB) void mphf_ranks_kernel(uint64_t *B, uint64_t n_blocks, const uint64_t *__restrict__ ranks) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_blocks; i += stride) B[i * 8] = ranks[i];
}
// bucket id of every minimizer (its MPHF slot), the bucket's size stored at that id, and a mark at the first occurrence of the minimizer
__global__ __launch_bounds__(kB) void bucket_ids_kernel(MphfView mv, const uint64_t *__restrict__ mm, const uint64_t *__restrict__ cnt, const uint64_t *__restrict__ range,
                                                        uint64_t n_mm, uint64_t *__restrict__ slot, uint64_t *__restrict__ bsize, uint64_t *__restrict__ head,
                                                        unsigned long long *n_bad) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_mm; i += stride) {
        uint64_t h = 0;
        if (!mphf_lookup(mv, mm[i], h) || h >= n_mm) { atomicAdd(n_bad, 1ull); h = 0; }
        slot[i] = h;
        bsize[h] = cnt[i];
        head[range[i]] = i;
    }
}
// entries of the sorted occurrences, scattered into their buckets (step 5 of SSHashHost::build): owner[j] = minimizer of occurrence j
__global__ __launch_bounds__(kB) void wide_entries_kernel(const uint64_t *__restrict__ pc, const uint64_t *__restrict__ owner, const uint64_t *__restrict__ range,
                                                          const uint64_t *__restrict__ slot, const uint64_t *__restrict__ psum, uint64_t n_occ, ScanParams P,
                                                          uint64_t *__restrict__ entries) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < n_occ; j += stride) {
        const uint64_t i = owner[j], pos = pc[j] >> 1;
        entries[psum[slot[i]] + (j - range[i])] = make_entry(pos, (uint32_t)(pc[j] & 1), flank_of(P.useq, P.total_len, pos, P.w));
    }
}
__global__ __launch_bounds__(kB) void hashed_line_of_kernel(const uint64_t *__restrict__ mm, uint64_t n_mm, uint64_t n_lines, uint32_t *__restrict__ line, uint32_t *__restrict__ idx) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_mm; i += stride) { // (lines and minimizers both number fewer than 2^32)
        line[i] = (uint32_t)front_hashed_line(front_hash(mm[i]), n_lines);
        idx[i] = (uint32_t)i;
    }
}
// hash-addressed minimizer lines (FrontTable::build_hashed): the lane at the first member of a line packs it; `lines` starts all empty
__global__ __launch_bounds__(kB) void hashed_lines_kernel(const uint32_t *__restrict__ line, const uint32_t *__restrict__ idx, uint64_t n_mm,
                                                          const uint64_t *__restrict__ mm, const uint64_t *__restrict__ slot, const uint64_t *__restrict__ psum,
                                                          const uint64_t *__restrict__ entries, uint64_t skew_param, uint64_t *__restrict__ lines,
                                                          unsigned long long *n_unusable) {
    uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n_mm; i += stride) {
        if (i && line[i] == line[i - 1]) continue;
        uint64_t j = i;
        while (j < n_mm && j - i <= kFrontSlots && line[j] == line[i]) ++j; // (one past 16 members is enough to know the line is given up)
        uint64_t out[kFrontSlots];
        uint32_t keyfield[kFrontSlots];
        uint64_t first[kFrontSlots], size[kFrontSlots];
        const bool ok = j - i <= kFrontSlots;
        for (uint64_t a = i; a < j && ok; ++a) {
            const uint64_t mi = idx[a], b = slot[mi];
            keyfield[a - i] = front_hashed_keyfield(front_hash(mm[mi]));
            first[a - i] = psum[b];
            size[a - i] = psum[b + 1] - psum[b];
        }
        uint64_t *dst = lines + (uint64_t)line[i] * kFrontSlots;
        if (ok) {
            uint32_t nw = 0;
            const uint32_t m = front_merge_clashes((uint32_t)(j - i), keyfield, first, size, &nw);
            if (nw) atomicAdd(n_unusable + 1, (unsigned long long)nw);
            front_pack_line(m, keyfield, first, size, entries, skew_param, out);
#pragma unroll
            for (uint32_t q = 0; q < kFrontSlots; ++q) dst[q] = out[q];
        } else {
            dst[0] = kFrontUnusable;
            atomicAdd(n_unusable, 1ull);
        }
    }
}

unsigned grid_of(uint64_t n, unsigned cap = 256 * 32) { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kB - 1) / kB, cap)); }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// rocPRIM calls: size query, allocate, run
#define ROCPRIM_RUN(buf, call_with_tmp)                                   \
    do {                                                                  \
        void *tmp = nullptr;                                              \
        size_t tmp_bytes = 0;                                             \
        BG_HIP(call_with_tmp);                                            \
        BG_HIP((buf).alloc((char **)&tmp, tmp_bytes));                    \
        BG_HIP(call_with_tmp);                                            \
        BG_HIP(hipDeviceSynchronize());                                   \
        (buf).release(tmp);                                               \
    } while (0)

template <class T>
int download(std::vector<T> &dst, const T *src, uint64_t n) {
    dst.resize(n);
    if (n) BG_HIP(hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    return MZ_OK;
}

} // namespace

bool gpu_build_supported(const UnitigSet &us, uint32_t w) {
    if (w == 0 || w > us.k) return false;
    if (2 * w > 32) return w <= 31 && bit_len(us.total_len) + 1 <= 63; // wide path: (word; position << 1 | class) as two sort words
    if (2 * w + bit_len(us.total_len) + ((w & 1) ? 0 : 1) > 64) return false;
    uint64_t rec_bytes = (((1ull << (2 * w)) >> kMinRecKeysLog2) + 1) * kMinRecWords * 8;
    return rec_bytes <= (8ull << 30); // the host rule additionally needs >= rec_bytes/16 minimizers; checked after G2
}

// returns MZ_OK and fills `h` like SSHashHost::build, or MZ_EINVAL with last_error "unsupported" (caller falls back)
int build_on_device(SSHashHost &h, const UnitigSet &unitigs, uint32_t w, uint64_t skew_param, uint64_t seed, int device, DeviceIndex **keep, const BuildOpts &opts_) {
    const BuildOpts opts = opts_.with_env();
    // keep != nullptr: the finished arrays stay in HBM as the replica *keep and are NOT copied to the host (h.arrays_on_host
    // == false; device_download() brings them over when somebody needs them there)
    if (keep) *keep = nullptr;
    const bool resident = keep != nullptr;
    const double t_begin = now_s();
    const bool timing = std::getenv("MZ_BUILD_TIMING") != nullptr;
    double tp = t_begin;
    auto phase = [&](const char *name) {
        if (timing) { (void)hipDeviceSynchronize(); double t = now_s(); std::fprintf(stderr, "[mz gpu build] %-28s %7.3f s\n", name, t - tp); tp = t; }
    };
    if (!gpu_build_supported(unitigs, w)) { set_error("GPU build unsupported for this (w, length)"); return MZ_EINVAL; }
    if (device_count() <= 0) { set_error("no HIP device visible"); return MZ_EHIP; }
    BG_HIP(hipSetDevice(device));
    h.us = unitigs;
    h.w = w; h.seed = seed; h.skew_param = skew_param; h.has_skew = false;
    const UnitigSet &us = h.us;
    const uint32_t k = us.k;
    DevBuf buf;
    ScanParams P{};
    uint64_t *d_useq = nullptr, *d_seqbv = nullptr;
    BG_HIP(buf.alloc(&d_useq, us.useq.size()));
    BG_HIP(buf.alloc(&d_seqbv, us.seqbv.size()));
    BG_HIP(hipMemcpy(d_useq, us.useq.data(), us.useq.size() * 8, hipMemcpyHostToDevice));
    BG_HIP(hipMemcpy(d_seqbv, us.seqbv.data(), us.seqbv.size() * 8, hipMemcpyHostToDevice));
    P.useq = d_useq; P.seqbv = d_seqbv; P.total_len = us.total_len; P.seed = seed; P.k = k; P.w = w;
    const bool wide = 2 * w > 32;
    P.cls_bits = wide || 2 * w + bit_len(us.total_len) + 1 <= 64 ? 1u : 0u;
    if (!wide && (w & 1) && std::getenv("MZ_TEST_GPU_BUILD_IMPLIED_CLASS")) P.cls_bits = 0; // tests walk the implied-class form on small sets
    P.key_shift = wide ? 63u : bit_len(us.total_len) + P.cls_bits; // (wide: the occurrence word the later kernels see is position << 1 | class)
    phase("upload sequence");

    // ---- G1: count, then store
    unsigned long long *d_counter = nullptr;
    BG_HIP(buf.alloc(&d_counter, 4));
    BG_HIP(hipMemset(d_counter, 0, 32));
    const uint64_t n_tiles = us.total_len >= k ? (us.total_len - k + 1 + kT - 1) / kT : 0;
    if (n_tiles == 0) { set_error("SSHash: no k-mers (every unitig shorter than k=%u)", k); return MZ_EEMPTY; }
    const unsigned sgrid = (unsigned)std::min<uint64_t>(n_tiles, 256 * 64);
    if (wide) hipLaunchKernelGGL((scan_minimizers_kernel<false, true>), dim3(sgrid), dim3(kB), 0, 0, P, n_tiles, (uint64_t *)nullptr, (uint64_t *)nullptr, d_counter);
    else hipLaunchKernelGGL((scan_minimizers_kernel<false, false>), dim3(sgrid), dim3(kB), 0, 0, P, n_tiles, (uint64_t *)nullptr, (uint64_t *)nullptr, d_counter);
    uint64_t n_raw = 0;
    BG_HIP(hipMemcpy(&n_raw, d_counter, 8, hipMemcpyDeviceToHost));
    if (n_raw == 0) { set_error("SSHash: no k-mers (every unitig shorter than k=%u)", k); return MZ_EEMPTY; }
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr, *d_pc = nullptr, *d_pc2 = nullptr; // (d_pc: wide only)
    BG_HIP(buf.alloc(&d_keys, n_raw));
    BG_HIP(buf.alloc(&d_keys2, n_raw));
    if (wide) { BG_HIP(buf.alloc(&d_pc, n_raw)); BG_HIP(buf.alloc(&d_pc2, n_raw)); }
    BG_HIP(hipMemset(d_counter, 0, 8));
    if (wide) hipLaunchKernelGGL((scan_minimizers_kernel<true, true>), dim3(sgrid), dim3(kB), 0, 0, P, n_tiles, d_keys, d_pc, d_counter);
    else hipLaunchKernelGGL((scan_minimizers_kernel<true, false>), dim3(sgrid), dim3(kB), 0, 0, P, n_tiles, d_keys, (uint64_t *)nullptr, d_counter);
    BG_HIP(hipGetLastError());
    phase("G1 minimizer scan");

    // ---- G2: sort, unique, run-length
    uint64_t *d_nsel = nullptr;
    BG_HIP(buf.alloc(&d_nsel, 2));
    uint64_t n_occ = 0;
    uint64_t *d_mm = nullptr, *d_cnt = nullptr;
    uint64_t n_mm = 0;
    if (wide) {
        // by (word, position, class): stable sort by the second word, then by the first; then the first of every run of equal pairs
        ROCPRIM_RUN(buf, rocprim::radix_sort_pairs(tmp, tmp_bytes, d_pc, d_pc2, d_keys, d_keys2, n_raw, 0, bit_len(us.total_len) + 1));
        ROCPRIM_RUN(buf, rocprim::radix_sort_pairs(tmp, tmp_bytes, d_keys2, d_keys, d_pc2, d_pc, n_raw, 0, 2 * w));
        uint8_t *d_first = nullptr;
        BG_HIP(buf.alloc(&d_first, n_raw));
        hipLaunchKernelGGL(first_of_pair_run_kernel, dim3(grid_of(n_raw)), dim3(kB), 0, 0, d_keys, d_pc, n_raw, d_first);
        ROCPRIM_RUN(buf, rocprim::select(tmp, tmp_bytes, d_keys, d_first, d_keys2, d_nsel, n_raw));
        ROCPRIM_RUN(buf, rocprim::select(tmp, tmp_bytes, d_pc, d_first, d_pc2, d_nsel, n_raw));
        BG_HIP(hipMemcpy(&n_occ, d_nsel, 8, hipMemcpyDeviceToHost));
        buf.release(d_first);
        std::swap(d_keys, d_keys2); // d_keys / d_pc: the sorted distinct occurrences; d_keys2 / d_pc2: free from here on
        std::swap(d_pc, d_pc2);
        if (n_occ >= (1ull << 32)) { set_error("GPU build unsupported: more than 2^32 minimizer occurrences"); return MZ_EINVAL; } // rocPRIM run_length_encode size
        BG_HIP(buf.alloc(&d_mm, n_occ));
        BG_HIP(buf.alloc(&d_cnt, n_occ + 1));
        ROCPRIM_RUN(buf, rocprim::run_length_encode(tmp, tmp_bytes, d_keys, n_occ, d_mm, d_cnt, d_nsel));
        buf.release(d_keys); // (the word of every occurrence: the runs have it from here on)
        d_keys = nullptr;
    } else {
        const unsigned key_bits = 2 * w + P.key_shift;
        ROCPRIM_RUN(buf, rocprim::radix_sort_keys(tmp, tmp_bytes, d_keys, d_keys2, n_raw, 0, key_bits));
        ROCPRIM_RUN(buf, rocprim::unique(tmp, tmp_bytes, d_keys2, d_keys, d_nsel, n_raw));
        BG_HIP(hipMemcpy(&n_occ, d_nsel, 8, hipMemcpyDeviceToHost));
        buf.release(d_keys2);
        if (n_occ >= (1ull << 32)) { set_error("GPU build unsupported: more than 2^32 minimizer occurrences"); return MZ_EINVAL; } // rocPRIM run_length_encode size
        auto words_it = rocprim::make_transform_iterator(d_keys, KeyToWord{P.key_shift});
        BG_HIP(buf.alloc(&d_mm, n_occ));
        BG_HIP(buf.alloc(&d_cnt, n_occ + 1));
        ROCPRIM_RUN(buf, rocprim::run_length_encode(tmp, tmp_bytes, words_it, n_occ, d_mm, d_cnt, d_nsel));
    }
    BG_HIP(hipMemcpy(&n_mm, d_nsel, 8, hipMemcpyDeviceToHost));
    h.n_minimizers = n_mm;
    if (!wide && !use_direct_records(w, n_mm, opts)) { set_error("GPU build unsupported: host rule selects the BBHash minimizer MPHF"); return MZ_EINVAL; }
    // bucket starts: psum[i] = occurrences before minimizer i (n_mm + 1 values); run_start == psum in monotone mode
    uint64_t *d_psum = nullptr;
    BG_HIP(buf.alloc(&d_psum, n_mm + 1));
    BG_HIP(hipMemset(d_cnt + n_mm, 0, 8));
    ROCPRIM_RUN(buf, rocprim::exclusive_scan(tmp, tmp_bytes, d_cnt, d_psum, (uint64_t)0, n_mm + 1, rocprim::plus<uint64_t>()));
    phase("G2 sort/unique/run-length");

    // ---- G3 (wide): BBHash levels over the minimizer set, bucket ids, bucket starts in MPHF order
    uint64_t *d_range = d_psum;          // first occurrence of every minimizer in the sorted occurrences (== bucket starts when the MPHF is monotone)
    uint64_t *d_slot = nullptr, *d_owner = nullptr, *d_mphf = nullptr;
    MphfLevel *d_levels = nullptr;
    if (wide) {
        h.mm.n_keys = n_mm; h.mm.direct = 0; h.mm.levels.clear(); h.mm.blocks.clear();
        std::vector<uint64_t *> lvl_blocks;
        uint64_t *d_cur = nullptr, *d_next = nullptr;
        uint8_t *d_flag = nullptr;
        BG_HIP(buf.alloc(&d_cur, n_mm)); BG_HIP(buf.alloc(&d_next, n_mm)); BG_HIP(buf.alloc(&d_flag, n_mm));
        BG_HIP(hipMemcpy(d_cur, d_mm, n_mm * 8, hipMemcpyDeviceToDevice));
        uint64_t rem = n_mm, total_blocks = 0, lvl_seed = 0x9E3779B97F4A7C15ull;
        while (rem) { // Mphf::build, level by level
            if (h.mm.levels.size() >= kMphfMaxLevels) { set_error("MPHF: more than %u levels (duplicate keys?)", kMphfMaxLevels); return MZ_EINVAL; }
            const uint64_t nb = (uint64_t)((1.7 * (double)rem) / kMphfBitsPerBlock) + 1;
            if (nb > 0xFFFFFFFFull) { set_error("MPHF: level too large"); return MZ_EINVAL; }
            lvl_seed = lvl_seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;
            uint64_t *d_occb = nullptr, *d_colb = nullptr, *d_B = nullptr;
            BG_HIP(buf.alloc(&d_occb, nb * kMphfDataWords)); BG_HIP(buf.alloc(&d_colb, nb * kMphfDataWords)); BG_HIP(buf.alloc(&d_B, nb * 8));
            BG_HIP(hipMemset(d_occb, 0, nb * kMphfDataWords * 8)); BG_HIP(hipMemset(d_colb, 0, nb * kMphfDataWords * 8));
            hipLaunchKernelGGL(mphf_mark_kernel, dim3(grid_of(rem)), dim3(kB), 0, 0, d_cur, rem, lvl_seed, (uint32_t)nb, (unsigned long long *)d_occb, (unsigned long long *)d_colb);
            hipLaunchKernelGGL(mphf_blocks_kernel, dim3(grid_of(nb)), dim3(kB), 0, 0, d_occb, d_colb, nb, d_B);
            hipLaunchKernelGGL(mphf_unplaced_kernel, dim3(grid_of(rem)), dim3(kB), 0, 0, d_cur, rem, lvl_seed, (uint32_t)nb, d_occb, d_flag);
            BG_HIP(hipGetLastError());
            ROCPRIM_RUN(buf, rocprim::select(tmp, tmp_bytes, d_cur, d_flag, d_next, d_nsel, rem));
            uint64_t left = 0;
            BG_HIP(hipMemcpy(&left, d_nsel, 8, hipMemcpyDeviceToHost));
            buf.release(d_occb); buf.release(d_colb);
            h.mm.levels.push_back(MphfLevel{total_blocks, lvl_seed, (uint32_t)nb, 0});
            lvl_blocks.push_back(d_B);
            total_blocks += nb;
            std::swap(d_cur, d_next);
            rem = left;
        }
        buf.release(d_cur); buf.release(d_next); buf.release(d_flag);
        BG_HIP(buf.alloc(&d_mphf, std::max<uint64_t>(total_blocks, 1) * 8));
        BG_HIP(hipMemset(d_mphf, 0, std::max<uint64_t>(total_blocks, 1) * 64));
        for (size_t l = 0; l < lvl_blocks.size(); ++l) {
            BG_HIP(hipMemcpy(d_mphf + h.mm.levels[l].block_off * 8, lvl_blocks[l], (uint64_t)h.mm.levels[l].n_blocks * 64, hipMemcpyDeviceToDevice));
            buf.release(lvl_blocks[l]);
        }
        {   // absolute ranks: word 0 of every block = ones before it
            uint64_t *d_c = nullptr, *d_r = nullptr;
            BG_HIP(buf.alloc(&d_c, total_blocks + 1)); BG_HIP(buf.alloc(&d_r, total_blocks + 1));
            hipLaunchKernelGGL(mphf_counts_kernel, dim3(grid_of(total_blocks)), dim3(kB), 0, 0, d_mphf, total_blocks, d_c);
            BG_HIP(hipMemset(d_c + total_blocks, 0, 8));
            ROCPRIM_RUN(buf, rocprim::exclusive_scan(tmp, tmp_bytes, d_c, d_r, (uint64_t)0, total_blocks + 1, rocprim::plus<uint64_t>()));
            uint64_t placed = 0;
            BG_HIP(hipMemcpy(&placed, d_r + total_blocks, 8, hipMemcpyDeviceToHost));
            if (placed != n_mm) { set_error("MPHF: placed %llu of %llu keys", (unsigned long long)placed, (unsigned long long)n_mm); return MZ_EINVAL; }
            hipLaunchKernelGGL(mphf_ranks_kernel, dim3(grid_of(total_blocks)), dim3(kB), 0, 0, d_mphf, total_blocks, d_r);
            BG_HIP(hipGetLastError());
            buf.release(d_c); buf.release(d_r);
        }
        BG_HIP(buf.alloc(&d_levels, h.mm.levels.size()));
        BG_HIP(hipMemcpy(d_levels, h.mm.levels.data(), h.mm.levels.size() * sizeof(MphfLevel), hipMemcpyHostToDevice));
        phase("G3 minimizer MPHF (BBHash levels)");
        // bucket ids, sizes by bucket id -> bucket starts; owner of every occurrence
        uint64_t *d_bsize = nullptr, *d_head = d_keys2; // (d_keys2 / d_pc2 are free: n_raw >= n_occ words each)
        d_owner = d_pc2;
        BG_HIP(buf.alloc(&d_slot, n_mm)); BG_HIP(buf.alloc(&d_bsize, n_mm + 1));
        BG_HIP(buf.alloc(&d_psum, n_mm + 1));
        BG_HIP(hipMemset(d_head, 0, n_occ * 8));
        BG_HIP(hipMemset(d_counter, 0, 8));
        const MphfView mv{d_mphf, d_levels, (uint32_t)h.mm.levels.size(), 0u, n_mm};
        hipLaunchKernelGGL(bucket_ids_kernel, dim3(grid_of(n_mm)), dim3(kB), 0, 0, mv, d_mm, d_cnt, d_range, n_mm, d_slot, d_bsize, d_head, d_counter);
        BG_HIP(hipGetLastError());
        unsigned long long n_bad = 0;
        BG_HIP(hipMemcpy(&n_bad, d_counter, 8, hipMemcpyDeviceToHost));
        if (n_bad) { set_error("MPHF: %llu minimizers without a slot", n_bad); return MZ_EINVAL; }
        BG_HIP(hipMemset(d_bsize + n_mm, 0, 8));
        ROCPRIM_RUN(buf, rocprim::exclusive_scan(tmp, tmp_bytes, d_bsize, d_psum, (uint64_t)0, n_mm + 1, rocprim::plus<uint64_t>()));
        ROCPRIM_RUN(buf, rocprim::inclusive_scan(tmp, tmp_bytes, d_head, d_owner, n_occ, rocprim::maximum<uint64_t>()));
        buf.release(d_bsize);
        phase("G3 bucket ids + starts");
    }

    // ---- G3: minimizer records
    const uint64_t n_rec = wide ? 0 : ((1ull << (2 * w)) >> kMinRecKeysLog2) + 1;
    uint64_t *d_minrec = nullptr, *d_rcnt = nullptr, *d_rrank = nullptr;
    if (!wide) {
    BG_HIP(buf.alloc(&d_minrec, n_rec * kMinRecWords));
    BG_HIP(buf.alloc(&d_rcnt, n_rec));
    BG_HIP(buf.alloc(&d_rrank, n_rec));
    BG_HIP(hipMemset(d_minrec, 0, n_rec * kMinRecWords * 8));
    hipLaunchKernelGGL(set_member_bits_kernel, dim3(grid_of(n_mm)), dim3(kB), 0, 0, d_mm, n_mm, d_minrec);
    hipLaunchKernelGGL(record_counts_kernel, dim3(grid_of(n_rec)), dim3(kB), 0, 0, d_minrec, n_rec, d_rcnt);
    ROCPRIM_RUN(buf, rocprim::exclusive_scan(tmp, tmp_bytes, d_rcnt, d_rrank, (uint64_t)0, n_rec, rocprim::plus<uint64_t>()));
    BG_HIP(hipMemset(d_counter, 0, 8));
    hipLaunchKernelGGL(finish_records_kernel, dim3(grid_of(n_rec)), dim3(kB), 0, 0, d_minrec, n_rec, d_rrank, d_rcnt, d_psum, n_mm + 1, d_counter);
    BG_HIP(hipMemcpy(&h.n_minrec_overflow, d_counter, 8, hipMemcpyDeviceToHost));
    h.mm.n_keys = n_mm; h.mm.direct = 1; h.mm.levels.clear();
    buf.release(d_rcnt); buf.release(d_rrank); // d_minrec stays until its filters are filled from the entries
    phase("G3 minimizer records");
    if (timing) std::fprintf(stderr, "[mz gpu build] %llu of %llu minimizer records overflow to the global offsets\n", (unsigned long long)h.n_minrec_overflow, (unsigned long long)n_rec);
    } else {
        h.n_minrec_overflow = 0;
    }

    // global partitioned Elias-Fano over psum
    uint64_t *d_bef_blocks = nullptr, *d_bef_ovf = nullptr, *d_packed_keep = nullptr, *d_lines_keep = nullptr, *d_minrec_keep = nullptr;
    {
        const uint64_t n = n_mm + 1, nb = ((n - 1) >> kBefShift) + 1;
        uint64_t *d_blocks = nullptr, *d_flag = nullptr, *d_rank = nullptr;
        BG_HIP(buf.alloc(&d_blocks, nb * 8));
        BG_HIP(buf.alloc(&d_flag, nb + 1));
        BG_HIP(buf.alloc(&d_rank, nb + 1));
        hipLaunchKernelGGL(bef_blocks_kernel, dim3(grid_of(nb)), dim3(kB), 0, 0, d_psum, n, nb, d_blocks, d_flag);
        BG_HIP(hipMemset(d_flag + nb, 0, 8));
        ROCPRIM_RUN(buf, rocprim::exclusive_scan(tmp, tmp_bytes, d_flag, d_rank, (uint64_t)0, nb + 1, rocprim::plus<uint64_t>()));
        uint64_t n_ov = 0;
        BG_HIP(hipMemcpy(&n_ov, d_rank + nb, 8, hipMemcpyDeviceToHost));
        h.buckets.n = n; h.buckets.n_overflow = n_ov;
        BG_HIP(buf.alloc(&d_bef_ovf, n_ov ? n_ov * (kBefSpan + 1) : 2));
        if (n_ov) {
            hipLaunchKernelGGL(bef_overflow_kernel, dim3(grid_of(nb)), dim3(kB), 0, 0, d_psum, n, nb, d_flag, d_rank, d_blocks, d_bef_ovf);
        } else {
            BG_HIP(hipMemset(d_bef_ovf, 0, 16));
        }
        d_bef_blocks = d_blocks;
        if (!resident) {
            if (int rc = download(h.buckets.ovf, d_bef_ovf, n_ov ? n_ov * (kBefSpan + 1) : 2)) return rc;
            if (int rc = download(h.buckets.blocks, d_blocks, nb * 8)) return rc;
            buf.release(d_blocks); buf.release(d_bef_ovf);
        }
        buf.release(d_flag); buf.release(d_rank);
    }
    // packed pos entries
    {
        h.pos.n = n_occ;
        h.pos.width = kPosEntryBits;
        const uint64_t n_words = (n_occ * h.pos.width + 63) / 64 + 2;
        uint64_t *d_packed = nullptr;
        BG_HIP(buf.alloc(&d_packed, n_words));
        if (wide) {
            BG_HIP(hipMemset(d_packed + n_occ, 0, (n_words - n_occ) * 8)); // (the two pad words)
            hipLaunchKernelGGL(wide_entries_kernel, dim3(grid_of(n_occ)), dim3(kB), 0, 0, d_pc, d_owner, d_range, d_slot, d_psum, n_occ, P, d_packed);
            BG_HIP(hipDeviceSynchronize());
            buf.release(d_keys2); buf.release(d_pc2); // (head marks and owners)
            d_keys2 = d_pc2 = d_owner = nullptr;
        } else {
            hipLaunchKernelGGL(pos_entries_kernel, dim3(grid_of(n_words)), dim3(kB), 0, 0, d_keys, n_occ, P, h.pos.width, n_words, d_packed);
            hipLaunchKernelGGL(record_filters_kernel, dim3(grid_of(n_mm)), dim3(kB), 0, 0, d_minrec, d_mm, n_mm, d_psum, d_packed, skew_param);
        }
        BG_HIP(hipGetLastError());
        if (!resident) {
            if (int rc = download(h.pos.w, d_packed, n_words)) return rc;
            if (int rc = download(h.mm.blocks, wide ? d_mphf : d_minrec, wide ? h.n_mm_block_words() : n_rec * kMinRecWords)) return rc;
        }
        // minimizer lines from the finished records, bucket starts and entries
        h.front.lines.clear();
        h.front.n_unusable = h.front.n_walk = 0;
        uint32_t hashed_bits = 0;
        h.front.present = wide ? FrontTable::choose_hashed(w, n_occ, n_mm, opts, hashed_bits) : FrontTable::choose(w, n_occ, true, opts, h.front.g, h.front.t);
        if (wide && h.front.present) { // hash-addressed lines (FrontTable::build_hashed): members grouped by line with a stable sort, one lane per line
            h.front.g = 0; h.front.t = hashed_bits;
            const uint64_t n_lines = 1ull << hashed_bits;
            uint64_t *d_lines = nullptr;
            uint32_t *d_line = nullptr, *d_idx = nullptr, *d_line2 = nullptr, *d_idx2 = nullptr;
            if (hashed_bits > 32 || n_mm >= (1ull << 32)) { set_error("GPU build unsupported: more than 2^32 minimizers or lines"); return MZ_EINVAL; }
            BG_HIP(buf.alloc(&d_lines, n_lines * kFrontSlots));
            BG_HIP(buf.alloc(&d_line, n_mm)); BG_HIP(buf.alloc(&d_idx, n_mm)); BG_HIP(buf.alloc(&d_line2, n_mm)); BG_HIP(buf.alloc(&d_idx2, n_mm));
            BG_HIP(hipMemset(d_lines, 0xFF, n_lines * kFrontSlots * 8)); // kFrontEmpty
            hipLaunchKernelGGL(hashed_line_of_kernel, dim3(grid_of(n_mm)), dim3(kB), 0, 0, d_mm, n_mm, n_lines, d_line, d_idx);
            ROCPRIM_RUN(buf, rocprim::radix_sort_pairs(tmp, tmp_bytes, d_line, d_line2, d_idx, d_idx2, n_mm, 0, std::max(hashed_bits, 1u)));
            BG_HIP(hipMemset(d_counter, 0, 16));
            hipLaunchKernelGGL(hashed_lines_kernel, dim3(grid_of(n_mm)), dim3(kB), 0, 0, d_line2, d_idx2, n_mm, d_mm, d_slot, d_psum, d_packed, skew_param, d_lines, d_counter);
            BG_HIP(hipGetLastError());
            unsigned long long n_un[2] = {0, 0}; // lines given up, walk markers
            BG_HIP(hipMemcpy(n_un, d_counter, 16, hipMemcpyDeviceToHost));
            h.front.n_unusable = n_un[0];
            h.front.n_walk = n_un[1];
            buf.release(d_line); buf.release(d_idx); buf.release(d_line2); buf.release(d_idx2);
            if (!resident) {
                if (int rc = download(h.front.lines, d_lines, n_lines * kFrontSlots)) return rc;
                buf.release(d_lines);
            }
            d_lines_keep = d_lines;
        }
        while (!wide && h.front.present) {
            const uint32_t line_bits = 2 * w - h.front.g - h.front.t;
            const uint64_t n_lines = 1ull << line_bits;
            uint64_t *d_lines = nullptr;
            BG_HIP(buf.alloc(&d_lines, n_lines * kFrontSlots));
            BG_HIP(hipMemset(d_counter, 0, 8));
            hipLaunchKernelGGL(front_lines_kernel, dim3(grid_of(n_lines)), dim3(kB), 0, 0, d_minrec, h.front.g, h.front.t, line_bits, d_psum, d_packed, skew_param,
                               d_lines, d_counter);
            BG_HIP(hipGetLastError());
            unsigned long long overfull = 0;
            BG_HIP(hipMemcpy(&overfull, d_counter, 8, hipMemcpyDeviceToHost));
            if (overfull) { // some line has more than 16 member keys: fold the key space less (as the host builder does)
                buf.release(d_lines);
                h.front.present = FrontTable::next_shape(w, n_occ, opts, h.front.g, h.front.t);
                continue;
            }
            if (!resident) {
                if (int rc = download(h.front.lines, d_lines, n_lines * kFrontSlots)) return rc;
                buf.release(d_lines);
            }
            d_lines_keep = d_lines;
            break;
        }
        if (!h.front.present) h.front.g = h.front.t = 0;
        d_packed_keep = d_packed;
        d_minrec_keep = wide ? d_mphf : d_minrec;
        if (!resident) {
            buf.release(d_packed);
            buf.release(wide ? d_mphf : d_minrec);
        }
    }
    phase("G3 offsets + entries + lines");

    // ---- G4: skew index
    uint64_t *d_skew_tab = nullptr, *d_skew_ow = nullptr, *d_skew_op = nullptr;
    if (skew_param != ~0ull) {
        BG_HIP(hipMemset(d_counter, 0, 8));
        const unsigned kgrid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(n_mm, 1), 256 * 64);
        const uint64_t *d_occ_keys = wide ? d_pc : d_keys; // the word of an occurrence that carries its position
        hipLaunchKernelGGL(skew_tuples_kernel<false>, dim3(kgrid), dim3(kB), 0, 0, d_occ_keys, d_range, d_cnt, n_mm, P, skew_param, (uint64_t *)nullptr,
                           (uint64_t *)nullptr, d_counter);
        uint64_t n_tup = 0;
        BG_HIP(hipMemcpy(&n_tup, d_counter, 8, hipMemcpyDeviceToHost));
        uint64_t *d_tw = nullptr, *d_tp = nullptr, *d_tw2 = nullptr, *d_tp2 = nullptr;
        BG_HIP(buf.alloc(&d_tw, n_tup)); BG_HIP(buf.alloc(&d_tp, n_tup));
        BG_HIP(buf.alloc(&d_tw2, n_tup)); BG_HIP(buf.alloc(&d_tp2, n_tup));
        uint64_t sn = 0;
        if (n_tup) {
            BG_HIP(hipMemset(d_counter, 0, 8));
            hipLaunchKernelGGL(skew_tuples_kernel<true>, dim3(kgrid), dim3(kB), 0, 0, d_occ_keys, d_range, d_cnt, n_mm, P, skew_param, d_tw, d_tp, d_counter);
            // sort by (word, pos): stable sort by pos, then by word; keep the first tuple of every word
            ROCPRIM_RUN(buf, rocprim::radix_sort_pairs(tmp, tmp_bytes, d_tp, d_tp2, d_tw, d_tw2, n_tup, 0, bit_len(us.total_len)));
            ROCPRIM_RUN(buf, rocprim::radix_sort_pairs(tmp, tmp_bytes, d_tw2, d_tw, d_tp2, d_tp, n_tup, 0, 2 * k));
            uint8_t *d_first = nullptr;
            BG_HIP(buf.alloc(&d_first, n_tup));
            hipLaunchKernelGGL(first_of_run_kernel, dim3(grid_of(n_tup)), dim3(kB), 0, 0, d_tw, n_tup, d_first);
            ROCPRIM_RUN(buf, rocprim::select(tmp, tmp_bytes, d_tw, d_first, d_tw2, d_nsel, n_tup));
            ROCPRIM_RUN(buf, rocprim::select(tmp, tmp_bytes, d_tp, d_first, d_tp2, d_nsel, n_tup));
            BG_HIP(hipMemcpy(&sn, d_nsel, 8, hipMemcpyDeviceToHost));
            buf.release(d_first);
        }
        phase("G4 skew tuples + sort");
        // skew table over d_tw2[0..sn) (word order) / d_tp2 (SkewTable::build of the host)
        SkewTable &st = h.skew;
        st.n_keys = sn;
        st.fp_bits = skew_fp_bits_for_build();
        st.n_buckets = sn ? (sn + kSkewLoad - 1) / kSkewLoad : 0;
        st.ovf_word.clear();
        st.ovf_pos.clear();
        const uint64_t n_slots = std::max<uint64_t>(st.n_buckets * kSkewSlots, 2);
        if (sn) {
            uint64_t *d_tab = nullptr, *d_hb = d_tw, *d_ix = d_tp, *d_hb2 = nullptr, *d_ix2 = nullptr, *d_first = nullptr;
            uint8_t *d_ovflag = nullptr;
            BG_HIP(buf.alloc(&d_tab, n_slots));
            BG_HIP(buf.alloc(&d_hb2, sn)); BG_HIP(buf.alloc(&d_ix2, sn)); BG_HIP(buf.alloc(&d_first, sn)); BG_HIP(buf.alloc(&d_ovflag, sn));
            BG_HIP(hipMemset(d_tab, 0xFF, n_slots * 8));
            hipLaunchKernelGGL(skew_home_kernel, dim3(grid_of(sn)), dim3(kB), 0, 0, d_tw2, sn, st.n_buckets, st.fp_mask(), d_hb, d_ix);
            // stable sort by home bucket: the keys of a bucket stay in word order
            ROCPRIM_RUN(buf, rocprim::radix_sort_pairs(tmp, tmp_bytes, d_hb, d_hb2, d_ix, d_ix2, sn, 0, bit_len(st.n_buckets)));
            hipLaunchKernelGGL(skew_heads_kernel, dim3(grid_of(sn)), dim3(kB), 0, 0, d_hb2, sn, d_hb); // d_hb reused: head positions
            ROCPRIM_RUN(buf, rocprim::inclusive_scan(tmp, tmp_bytes, d_hb, d_first, sn, rocprim::maximum<uint64_t>()));
            hipLaunchKernelGGL(skew_place_kernel, dim3(grid_of(sn)), dim3(kB), 0, 0, d_hb2, d_ix2, d_first, d_tw2, d_tp2, sn, st.n_buckets, st.fp_mask(), d_tab, d_ovflag);
            ROCPRIM_RUN(buf, rocprim::select(tmp, tmp_bytes, d_ix2, d_ovflag, d_ix, d_nsel, sn)); // d_ix reused: overflowed key indices, in (bucket, word) order
            uint64_t n_ovf = 0;
            BG_HIP(hipMemcpy(&n_ovf, d_nsel, 8, hipMemcpyDeviceToHost));
            st.n_ovf = n_ovf;
            BG_HIP(buf.alloc(&d_skew_ow, n_ovf));
            BG_HIP(buf.alloc(&d_skew_op, n_ovf));
            if (n_ovf) {
                hipLaunchKernelGGL(gather_kernel, dim3(grid_of(n_ovf)), dim3(kB), 0, 0, d_tw2, d_ix, n_ovf, d_skew_ow);
                hipLaunchKernelGGL(gather_kernel, dim3(grid_of(n_ovf)), dim3(kB), 0, 0, d_tp2, d_ix, n_ovf, d_skew_op);
                BG_HIP(hipGetLastError());
            }
            d_skew_tab = d_tab;
            if (!resident) {
                if (n_ovf) {
                    if (int rc = download(st.ovf_word, d_skew_ow, n_ovf)) return rc;
                    if (int rc = download(st.ovf_pos, d_skew_op, n_ovf)) return rc;
                }
                if (int rc = download(st.buckets, d_tab, n_slots)) return rc;
            }
        } else {
            st.n_ovf = 0;
            BG_HIP(buf.alloc(&d_skew_tab, n_slots));
            BG_HIP(hipMemset(d_skew_tab, 0xFF, n_slots * 8));
            BG_HIP(buf.alloc(&d_skew_ow, 0));
            BG_HIP(buf.alloc(&d_skew_op, 0));
            if (!resident) st.buckets.assign(n_slots, kSkewEmpty);
        }
        h.has_skew = true;
        phase("G4 skew table");
    }
    BG_HIP(hipDeviceSynchronize());
    BG_HIP(hipGetLastError());
    h.arrays_on_host = !resident;
    if (resident) { // the finished arrays become the replica of this device
        h.mm.blocks.clear(); h.buckets.blocks.clear(); h.buckets.ovf.clear(); h.pos.w.clear(); h.front.lines.clear();
        h.skew.buckets.clear(); h.skew.ovf_word.clear(); h.skew.ovf_pos.clear();
        uint64_t *d_starts = nullptr;
        BG_HIP(buf.alloc(&d_starts, us.starts.size()));
        BG_HIP(hipMemcpy(d_starts, us.starts.data(), us.starts.size() * 8, hipMemcpyHostToDevice));
        DeviceIndex *d = new DeviceIndex();
        d->device = device;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) d->n_cu = (uint32_t)prop.multiProcessorCount;
        IndexView v = h.view();
        v.seqbv = d_seqbv;                 buf.give(d_seqbv, d, us.seqbv.size() * 8);
        v.starts = d_starts;               buf.give(d_starts, d, us.starts.size() * 8);
        v.mm.blocks = d_minrec_keep;       buf.give(d_minrec_keep, d, h.n_mm_block_words() * 8);
        v.mm.levels = nullptr;
        if (wide) { v.mm.levels = d_levels; buf.give(d_levels, d, h.mm.levels.size() * sizeof(MphfLevel)); }
        v.buckets.blocks = d_bef_blocks;   buf.give(d_bef_blocks, d, h.n_bucket_block_words() * 8);
        v.buckets.ovf = d_bef_ovf;         buf.give(d_bef_ovf, d, h.n_bucket_ovf_words() * 8);
        v.pos = d_packed_keep;             buf.give(d_packed_keep, d, (h.pos.n + 2) * 8);
        if (h.has_skew) {
            v.skew = h.skew.view();
            v.skew.buckets = d_skew_tab;   buf.give(d_skew_tab, d, h.skew.n_slots() * 8);
            v.skew.ovf_word = d_skew_ow;   buf.give(d_skew_ow, d, h.skew.n_ovf * 8);
            v.skew.ovf_pos = d_skew_op;    buf.give(d_skew_op, d, h.skew.n_ovf * 8);
        }
        v.front = h.front.view();
        if (h.front.present) {
            v.front = FrontView{d_lines_keep, wide ? (uint64_t)1 << h.front.t : (uint64_t)1 << (2 * w - h.front.g - h.front.t), h.front.g, h.front.t};
            buf.give(d_lines_keep, d, h.front.n_words(w) * 8);
        }
        d->view = v;
        d->n_entries = h.pos.n;
        *keep = d;
    }
    h.build_seconds = now_s() - t_begin;
    return MZ_OK;
}

} // namespace mz
