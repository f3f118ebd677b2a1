// mz_device.hip -- gfx950 (MI355X / CDNA4) kernels for the SSHash k-mer -> unitig path.
//
// Every kernel is pure 64-bit integer work on a latency/bandwidth-bound gather chain
// (no MFMA): one query per lane, 64-lane wavefronts, results written as coalesced 16-byte
// records, hit tallies reduced with wavefront ballots so each wave issues one atomic.
// Reference for the algorithm: src/kphf/sshash.rs:471-554 (k2u), :415-433 (skew),
// src/index/caching.rs:65-103 (streaming), src/kphf/mod.rs:69-139 (validate_self).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mz_device.hpp"

namespace mz {

#define MZ_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
            return MZ_EHIP;                                                                 \
        }                                                                                   \
    } while (0)

// static instruction counts between points of a kernel: -DMZ_MARKS puts comments into the assembly (tools/count_marks.py)
#ifdef MZ_MARKS
#define MZ_MARK(name) asm volatile("; MZMARK " name)
#else
#define MZ_MARK(name) ((void)0)
#endif
constexpr int kBlock = 256;

// Hit / miss / skipped tallies.  Lanes ballot and every wave keeps wave-uniform popcounts in
// registers for the whole kernel; at the end the waves of a workgroup combine them in LDS and ONE
// lane issues the global atomics.  (A device-scope atomic on one address costs ~12 ns, so a
// per-wave-per-iteration tally would cap the whole kernel at ~5 G queries/s on its own.)
struct Tally {
    uint32_t hit = 0, miss = 0, skipped = 0;
    __device__ __forceinline__ void add(bool h, bool m, bool s) {
        hit += (uint32_t)__popcll(__ballot(h));
        miss += (uint32_t)__popcll(__ballot(m));
        skipped += (uint32_t)__popcll(__ballot(s));
    }
    // call once per thread at the end of the kernel, from uniform control flow
    __device__ __forceinline__ void flush(mz_counts *counts) {
        __shared__ uint32_t s_tally[3];
        if (threadIdx.x == 0) { s_tally[0] = 0; s_tally[1] = 0; s_tally[2] = 0; }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            if (hit) atomicAdd(&s_tally[0], hit);
            if (miss) atomicAdd(&s_tally[1], miss);
            if (skipped) atomicAdd(&s_tally[2], skipped);
        }
        __syncthreads();
        if (threadIdx.x == 0 && counts) {
            if (s_tally[0]) atomicAdd((unsigned long long *)&counts->n_hit, (unsigned long long)s_tally[0]);
            if (s_tally[1]) atomicAdd((unsigned long long *)&counts->n_miss, (unsigned long long)s_tally[1]);
            if (s_tally[2]) atomicAdd((unsigned long long *)&counts->n_skipped, (unsigned long long)s_tally[2]);
        }
    }
};

// ------------------------------------------------------------------------------------
// K: batched point queries (mz_sshash_k2u_batch_device)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k2u_batch_kernel(IndexView ix, const uint64_t *__restrict__ fw_words, uint64_t n,
                                                           mz_k2upos *__restrict__ out, mz_counts *counts) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    Tally tally;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        mz_k2upos r = lookup(ix, fw_words[i]);
        out[i] = r;
        tally.add(r.match != MZ_NO_MATCH, r.match == MZ_NO_MATCH, false);
    }
    tally.flush(counts);
}


// ------------------------------------------------------------------------------------
// K: LDS-staged tile kernel -- the production k2u path.
//
// A workgroup owns a tile of kTileQ queries whose state lives in LDS; the dependent chain of
// sshash.rs:471-554 is cut into stages separated by barriers, and between stages the work is
// re-compacted with wavefront ballots so that every vector memory instruction is issued by
// (nearly) full waves, with several independent gathers in flight per lane:
//   S0  canonical encode + minimizer + the flank pattern an entry must show (ALU only)        dense over queries
//   S1  minimizer record: {header, bitmap word} probe + bucket-offset block + negative filter, all loads of the
//       128-byte line in one batch (BBHash path: one level set per round, unresolved keys ballot-compacted into
//       an LDS list; then the partitioned Elias-Fano block)                                  dense over queries
//   S3  bucket entries of the tile flattened by an LDS prefix sum; owners from a max-scan over bucket-head marks;
//       entry -> flank filter; the first survivor in bucket order is the item's candidate (LDS atomicMin)
//   S4  heavy buckets: one gather of the skew-table bucket, fingerprint match -> candidate position
//   S5  candidates compacted; one gather of the sequence+boundary record verifies the k-mer and gives
//       unitig id / offset / length (exact in-order walk only if a flagged item's first candidate fails);
//       results stored as coalesced 16-byte records
// ------------------------------------------------------------------------------------
constexpr int kQpt = 4;                  // queries per thread
constexpr int kTileQ = kBlock * kQpt;    // queries per tile
constexpr unsigned long long kBestNone = ~0ull;
constexpr uint32_t kPosBits = 36;
// best[i] of a candidate: global k-mer position | kBestTwin when the k-mer, if it verifies, is the reverse complement of the query
// (strand class of the entry xor strand of the query's canonical k-mer) | kBestUnk when the source of the candidate does not say
// (skew-table slots) | bucket order << kBestOrderShift (stage S3: the smallest value is the first survivor)
constexpr unsigned long long kBestTwin = 1ull << kPosBits, kBestUnk = 1ull << (kPosBits + 1);
constexpr unsigned long long kBestDefer = ~0ull - 1; // after S5: the item waits in the wave's list of deferred exact walks
constexpr uint32_t MZ_DEFER = 0xFFu;                 // match code of such an item (never leaves a kernel)
constexpr uint32_t kBestOrderShift = kPosBits + 2;
#ifndef MZ_SPECULATE
#define MZ_SPECULATE 1
#endif
constexpr bool kSpeculate = MZ_SPECULATE != 0;
#ifndef MZ_STOP_AFTER
#define MZ_STOP_AFTER 0 // stage-cost experiments only (tools/ab_build.sh): 1 = S0, 2 = S1/S2, 3 = S3, 4 = S4; results are then meaningless
#endif
constexpr int kStopAfter = MZ_STOP_AFTER;
#ifndef MZ_REC_BATCH
#define MZ_REC_BATCH 1
#endif
#ifndef MZ_POINT_TOUCH
#define MZ_POINT_TOUCH 0 // experiment (measured: no gain): the point kernel touches the lines of its next tile's first row before S5
#endif
#ifndef MZ_ROW_S0
#define MZ_ROW_S0 1 // wave-scoped point kernel: stage S0 row by row inside SF (see tile_resolve)
#endif
#ifndef MZ_WAVES_PER_EU
#define MZ_WAVES_PER_EU 4
#endif
#ifndef MZ_READS_SF_LANES
#define MZ_READS_SF_LANES 4 // lanes per minimizer line in stage SF of the reads kernel (tile_resolve)
#endif
#ifndef MZ_READS_WAVES
#define MZ_READS_WAVES 3 // minimum waves per SIMD the reads kernel is compiled for (<= 168 VGPRs; at 2 the allocator takes 179 and the kernel runs 20 % slower)
#endif
constexpr int kRecBatch = MZ_REC_BATCH; // minimizer records in flight per lane (S1)
// S.n[i]: bucket size (< kNSkew), or one of the markers below (kNIdle: no query in this slot)
constexpr uint32_t kNMiss = 0xFFFFFFFFu, kNIdle = 0xFFFFFFFEu, kNSkew = 0xFFFFFFFDu, kNPending = 0xFFFFFFFCu, kNExact = 0xFFFFFFFBu; // kNPending: SF left a marker slot in slot[i]; kNExact (runs only): the bucket's size is unknown, exact walk
constexpr uint64_t kSlotInLine = 1ull << 63; // slot[i] (bucket start) points into the minimizer lines (slot layout) instead of the entry array
// S.qinfo[i]: the flank16 an entry must show for item i (mz_common.hpp `pos` entry), in the orientation of the canonical
// word g*: bits 0..7 = the 4 bases left of the minimizer in g*, bits 8..15 = the 4 bases right of it | offset << 16 |
// kQMulti once a second bucket entry survived the flank filter
constexpr uint32_t kQOffShift = 16, kQOffMask = 0x1Fu, kQMulti = 0x80000000u, kQIdle = 0x40000000u, kQClass = 1u << 21, kQLine = 1u << 22;
// kQClass: the canonical k-mer is the reverse strand; kQIdle: no query in this slot; kQLine: the survivors behind kQMulti are entries of the query's
// minimizer line (front_walk serves)
__device__ __forceinline__ uint32_t query_flank(uint64_t g, uint32_t off, uint32_t w) {
    const uint32_t ql = (uint32_t)(off >= kFlankBases ? g >> (2 * (off - kFlankBases)) : g << (2 * (kFlankBases - off))) & 0xFFu;
    const uint32_t qr = (uint32_t)(g >> (2 * (off + w))) & 0xFFu;
    return ql | (qr << 8);
}

// boundary half of a sequence+boundary record, fetched as two 16-byte loads (records are 64-byte aligned)
struct SeqBoundary {
    ulonglong2 h, d;
};
__device__ __forceinline__ SeqBoundary load_boundary(const uint64_t *seqbv, uint64_t pos) {
    uint32_t off;
    const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(seqbv + seq_rec_of(pos, off) * kSeqRecWords + 4);
    return SeqBoundary{v[0], v[1]};
}
__device__ __forceinline__ UnitigSpan locate_boundary(const SeqBoundary &b, uint64_t pos) {
    uint32_t off;
    const uint64_t r = seq_rec_of(pos, off);
    return locate_unitig_r(b.h.x, b.h.y, b.d.x, b.d.y, off, r * kSeqRecStride);
}

// Execution scope of a tile: a whole 256-thread workgroup (barriers) or ONE wavefront (no barriers: the four
// waves of a workgroup then run their own tiles independently and never wait for each other).
struct BlockScope {
    static constexpr int NT = kBlock;
    __device__ static __forceinline__ uint32_t tid() { return threadIdx.x; }
    __device__ static __forceinline__ void sync() { __syncthreads(); }
};
struct WaveScope {
    static constexpr int NT = 64;
    __device__ static __forceinline__ uint32_t tid() { return threadIdx.x & 63; }
    // lanes of one wave execute in lockstep and LDS operations of a wave complete in order: only the compiler has to be
    // told not to move LDS accesses across this point
    __device__ static __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};

template <int NT>
struct TileLdsT {
    using ListT = typename std::conditional<(NT * kQpt <= 256), uint8_t, uint16_t>::type; // item indices of one tile
    uint64_t fw[NT * kQpt];               // forward word of the query
    uint64_t slot[NT * kQpt];             // minimizer word -> bucket id h -> bucket start (skew: canonical word -> slot); after S5: unitig id << 32 | length
    unsigned long long best[NT * kQpt];   // candidate: (bucket order << 37) | global k-mer position; after S5: 0 identity / 1 twin / kBestNone
    uint32_t n[NT * kQpt];                // bucket size, kNSkew, kNIdle or kNMiss; after S5: offset of the k-mer in its unitig
    uint32_t eoff[NT * kQpt + 1];         // exclusive prefix sum of bucket sizes
    uint32_t qinfo[NT * kQpt];            // expected flank16 of the query | minimizer offset inside g* << 16 | kQMulti
    ListT list_a[NT * kQpt], list_b[NT * kQpt];
    uint32_t cnt_a, cnt_b, cnt_skew, wave_tot[NT / 64];
#ifdef MZ_DEBUG_COUNT
    uint32_t dbg[6]; // scratch builds only (tools/ab_build.sh -DMZ_DEBUG_COUNT=n): S3 entries, S5 candidates (grouped: records gathered), exact walks, runs of the tile, exact walks that are not walks of a minimizer line
#endif
#ifdef MZ_LDS_PAD
    uint64_t pad[MZ_LDS_PAD * NT / 64];
#endif
};
using TileLds = TileLdsT<kBlock>;

// append `idx` of the lanes with pred to an LDS list (one LDS atomic per wave)
template <class T>
__device__ __forceinline__ void wave_push(T *list, uint32_t *cnt, bool pred, uint32_t idx) {
    unsigned long long b = __ballot(pred);
    if (b == 0) return;
    unsigned lane = __lane_id();
    unsigned leader = (unsigned)(__ffsll((long long)b) - 1);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(cnt, (uint32_t)__popcll(b));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader); // leader is wave-uniform: a scalar read, not an LDS-crossbar shuffle
    if (pred) list[base + __popcll(b & ((1ull << lane) - 1))] = (T)idx;
}

// wave64 inclusive scans on the VALU (DPP row shifts + row broadcasts, the classic 7-step sequence) instead of six
// dependent ds_bpermute round trips through the LDS crossbar.  Values are unsigned; identity 0 serves both add and max.
template <bool kMax>
__device__ __forceinline__ uint32_t wave_scan(uint32_t x) {
    auto op = [](uint32_t a, uint32_t b) { return kMax ? (a > b ? a : b) : a + b; };
#define MZ_DPP(src, ctrl, rmask, bmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(src), ctrl, rmask, bmask, false)
    uint32_t v = op(x, MZ_DPP(x, 0x111, 0xf, 0xf)); // row_shr:1
    v = op(v, MZ_DPP(x, 0x112, 0xf, 0xf));          // row_shr:2
    v = op(v, MZ_DPP(x, 0x113, 0xf, 0xf));          // row_shr:3
    v = op(v, MZ_DPP(v, 0x114, 0xf, 0xe));          // row_shr:4, banks 1..3
    v = op(v, MZ_DPP(v, 0x118, 0xf, 0xc));          // row_shr:8, banks 2..3
    v = op(v, MZ_DPP(v, 0x142, 0xa, 0xf));          // row_bcast:15 into rows 1 and 3
    v = op(v, MZ_DPP(v, 0x143, 0xc, 0xf));          // row_bcast:31 into rows 2 and 3
#undef MZ_DPP
    return v;
}
// value of the previous lane (0 for lane 0)
__device__ __forceinline__ uint32_t wave_prev(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false); } // wave_shr:1

struct MphfProbe { // one level: issue the loads first, evaluate later
    uint64_t word, h0, h1;
    uint32_t dw, bi;
};
__device__ __forceinline__ MphfProbe mphf_issue(const MphfView &m, const MphfLevel &L, uint64_t key) {
    uint32_t blk, bit;
    mphf_level_slot(key, L.seed, L.n_blocks, blk, bit);
    const uint64_t *b = m.blocks + (L.block_off + blk) * 8;
    MphfProbe p;
    p.dw = bit >> 6;
    p.bi = bit & 63;
    p.word = b[2 + p.dw];
    p.h0 = b[0];
    p.h1 = b[1];
    return p;
}
__device__ __forceinline__ bool mphf_eval(const MphfProbe &p, uint64_t &idx) {
    if (!((p.word >> p.bi) & 1)) return false;
    uint64_t below = p.bi ? (p.word & ((1ull << p.bi) - 1)) : 0;
    idx = p.h0 + ((p.h1 >> (9 * p.dw)) & 511u) + popc64(below);
    return true;
}

// S1 / S4: resolve slot[i] (key) -> MPHF index for the items of list_a[0..cnt_a); items that fall
// through every level get n[i] = kNMiss.  `dense_first` == all (SC::NT * kQpt) items (level 0, no list).
template <class SC>
__device__ __forceinline__ void tile_mphf(TileLdsT<SC::NT> &S, const MphfView &m, bool dense_first) {
    const uint32_t t = SC::tid();
    uint32_t lv = 0;
    if (dense_first) {
        if (m.n_levels == 0) {
            for (int q = 0; q < kQpt; ++q) { uint32_t i = q * SC::NT + t; if (S.n[i] < kNIdle) S.n[i] = kNMiss; }
            return;
        }
        const MphfLevel L = m.levels[0];
        MphfProbe pr[kQpt];
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * SC::NT + t;
            pr[q] = mphf_issue(m, L, S.slot[i]);
        }
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * SC::NT + t;
            uint64_t idx;
            bool hit = mphf_eval(pr[q], idx), act = S.n[i] < kNIdle;
            if (hit && act) S.slot[i] = idx;
            wave_push(S.list_a, &S.cnt_a, !hit && act, i);
        }
        lv = 1;
    }
    typename TileLdsT<SC::NT>::ListT *cur = S.list_a, *nxt = S.list_b;
    uint32_t *ccur = &S.cnt_a, *cnxt = &S.cnt_b;
    uint32_t round = 0;
    for (;;) {
        SC::sync();
        const uint32_t count = *ccur;
        if (count == 0) break;
        if (lv >= m.n_levels) { // fell through every level: boomphf try_hash -> None
            for (uint32_t j = t; j < count; j += SC::NT) S.n[cur[j]] = kNMiss;
            SC::sync();
            if (t == 0) *ccur = 0;
            break;
        }
        // levels probed per round grow 2, 4, 4, ...: the later (sparser) the list, the more levels are
        // issued speculatively at once, so the tail of the geometric distribution costs few rounds
        const uint32_t want = !kSpeculate ? 1u : (round == 0 ? 2u : 4u);
        const uint32_t nl = m.n_levels - lv < want ? m.n_levels - lv : want;
        MphfLevel L[4];
#pragma unroll
        for (uint32_t a = 0; a < 4; ++a) L[a] = m.levels[lv + (a < nl ? a : 0)];
        for (uint32_t j = t; j < ((count + 63) & ~63u); j += SC::NT) {
            bool act = j < count;
            uint32_t i = act ? cur[j] : 0;
            uint64_t key = S.slot[i];
            MphfProbe pr[4];
#pragma unroll
            for (uint32_t a = 0; a < 4; ++a)
                if (a < nl) pr[a] = mphf_issue(m, L[a], key);
            uint64_t idx = 0;
            bool hit = false;
#pragma unroll
            for (uint32_t a = 0; a < 4; ++a)
                if (a < nl && !hit) hit = mphf_eval(pr[a], idx);
            if (act && hit) S.slot[i] = idx;
            wave_push(nxt, cnxt, act && !hit, i);
        }
        ++round;
        lv += nl;
        SC::sync();
        if (t == 0) *ccur = 0;
        auto *tl = cur; cur = nxt; nxt = tl;
        uint32_t *tc = ccur; ccur = cnxt; cnxt = tc;
    }
    SC::sync();
}

// ---- tile producer / consumer API ------------------------------------------------------
// A kernel fills the tile with tile_set_query() (S0), calls tile_resolve() (S1-S4) from uniform
// control flow, then reads each item's answer with tile_result() (S5).
// the exact in-order walk of an item's bucket (sshash.rs:494-552), for the few items whose recorded candidates all failed although
// more entries had agreeing flanks.  MZ_WALK_CALL: compiled as a real function, so that its ~1500 instructions and its register
// needs stay out of the kernels' main path.
#ifndef MZ_WALK_CALL
#define MZ_WALK_CALL 0
#endif
__device__ mz_k2upos front_walk(const IndexView &ix, uint64_t fw, uint64_t rc, uint32_t off_a);
#if MZ_WALK_CALL
__device__ __attribute__((noinline))
#else
__device__ __forceinline__
#endif
mz_k2upos exact_walk(const IndexView &ix, uint64_t fw, uint32_t qi);
template <class SC>
__device__ __forceinline__ void tile_begin(TileLdsT<SC::NT> &S) {
    if (SC::tid() == 0) { S.cnt_a = 0; S.cnt_b = 0; S.cnt_skew = 0; }
}
// flank bits of a query that its k-mer actually covers, as a mask over flank16 (minimizer at offset `off` of the canonical word)
__device__ __forceinline__ uint32_t flank_cover_mask(uint32_t off, uint32_t room) {
    const uint32_t nl = off < kFlankBases ? off : kFlankBases, nr = room < kFlankBases ? room : kFlankBases;
    return ((0xFFu << (2 * (kFlankBases - nl))) & 0xFFu) | (((1u << (2 * nr)) - 1) << 8);
}
constexpr uint32_t kFrontIdleKey = 0x100u << kFrontKeyShift; // key field no slot can show (real keys use at most 8 bits, empty slots all nine)

// `lines`: the index has minimizer lines (keys fit 32 bits): the line of the query, its slot pattern (kind 0, the key's low and
// tag bits, the flanks) and the pattern's mask are prepared here, once per query, for stage SF (line in the low half of slot[],
// pattern in the high half, mask in eoff[], which S3 only needs later)
template <class SC>
__device__ __forceinline__ void tile_set_query(TileLdsT<SC::NT> &S, uint32_t i, uint64_t fw, uint64_t rc, Minimizer mn, const IndexView &ix, bool valid,
                                               bool lines) {
    const uint32_t k = ix.k, w = ix.w;
    const uint32_t fl = query_flank(fw <= rc ? fw : rc, mn.offset, w);
    S.fw[i] = fw;
    S.qinfo[i] = fl | (mn.offset << kQOffShift) | (valid ? 0u : kQIdle) | (fw <= rc ? 0u : kQClass);
    S.best[i] = kBestNone;
    S.n[i] = valid ? 0u : kNIdle;
    if (lines) {
        uint32_t line, kf;
        if (front_is_hashed(w)) { // 2w > 32: line and key field from a hash of the word
            const uint64_t h = front_hash(mn.word);
            line = (uint32_t)front_hashed_line(h, ix.front.n_lines);
            kf = front_hashed_keyfield(h);
        } else {
            const uint32_t key = (uint32_t)mn.word, g = ix.front.g, gl = 2 * w - ix.front.t; // word = tag | line index | low g bits: tag starts at bit gl
            line = (key >> g) & (uint32_t)(ix.front.n_lines - 1);
            kf = front_keyfield(key, g, gl);
        }
        const uint32_t pat = (fl << kFrontFlankShift) | (kf << kFrontKeyShift) | (valid ? 0u : kFrontIdleKey); // (an idle slot has key 0: line 0 is always there)
        S.slot[i] = (uint64_t)line | ((uint64_t)pat << 32);
        S.eoff[i] = 3u | kFrontKeyMask | (flank_cover_mask(mn.offset, k - w - mn.offset) << kFrontFlankShift);
    } else {
        S.slot[i] = mn.word;
    }
}

// first entry of the key's bucket, in bucket order, that verifies -- lookup() for a query whose bucket lies inline in its
// minimizer line (`off_a`: offset of the minimizer in the canonical k-mer).  The slow path of a window whose line held more
// than one entry with matching flanks and whose first one failed: the line is scanned again and every surviving entry is
// verified in turn, exactly as sshash.rs:494-552 walks the bucket.
__device__ __forceinline__ mz_k2upos front_walk(const IndexView &ix, uint64_t fw, uint64_t rc, uint32_t off_a) {
    const uint32_t k = ix.k, w = ix.w, g = ix.front.g, gl = 2 * w - ix.front.t;
    const uint64_t gs = fw <= rc ? fw : rc;
    const uint64_t word = (gs >> (2 * off_a)) & kmer_mask(w);
    const uint64_t wh = front_hash(word);
    const uint32_t key = (uint32_t)word;
    const uint64_t *line = ix.front.lines + (front_is_hashed(w) ? front_hashed_line(wh, ix.front.n_lines) : (uint64_t)((key >> g) & (uint32_t)(ix.front.n_lines - 1))) * kFrontSlots;
    const uint32_t pat = (query_flank(gs, off_a, w) << kFrontFlankShift) | ((front_is_hashed(w) ? front_hashed_keyfield(wh) : front_keyfield(key, g, gl)) << kFrontKeyShift);
    const uint32_t msk = 3u | kFrontKeyMask | (flank_cover_mask(off_a, k - w - off_a) << kFrontFlankShift);
    uint32_t alive = 0;
#pragma unroll 1
    for (uint32_t j0 = 0; j0 < kFrontSlots / 2; j0 += 2) { // (a rare path: few registers rather than all loads at once)
        const ulonglong2 a = reinterpret_cast<const ulonglong2 *>(line)[j0], b = reinterpret_cast<const ulonglong2 *>(line)[j0 + 1];
        const uint32_t m = ((((uint32_t)a.x ^ pat) & msk) == 0 ? 1u : 0u) | ((((uint32_t)a.y ^ pat) & msk) == 0 ? 2u : 0u) |
                           ((((uint32_t)b.x ^ pat) & msk) == 0 ? 4u : 0u) | ((((uint32_t)b.y ^ pat) & msk) == 0 ? 8u : 0u);
        alive |= m << (2 * j0);
    }
    while (alive) {
        const uint64_t e = line[__ffs((int)alive) - 1];
        alive &= alive - 1;
        const uint64_t p = (e >> kFrontPosShift) - (((e >> kFrontClsShift) & 1) ? k - w - off_a : off_a);
        if (p > ix.total_len - k) continue;
        const uint32_t mt = word_equivalency(fw, rc, read_kmer_i(ix.seqbv, p, k));
        if (mt == MZ_NO_MATCH) continue;
        const UnitigSpan sp = locate_unitig_i(ix.seqbv, p);
        if (p + k <= sp.end) return make_hit(sp, p, mt);
    }
    return make_none();
}
#if MZ_WALK_CALL
__device__ __attribute__((noinline))
#else
__device__ __forceinline__
#endif
mz_k2upos exact_walk(const IndexView &ix, uint64_t fw, uint32_t qi) {
#ifdef MZ_WALK_NEVER
    if (ix.seed != 0x0123456789abcdefull) return make_none(); // experiments only: the code is there, and never runs
#endif
    return (qi & kQLine) ? front_walk(ix, fw, kmer_rc(fw, ix.k), (qi >> kQOffShift) & kQOffMask) : lookup(ix, fw);
}
// kRuns: consecutive items of the tile tend to carry the same minimizer (the windows of a read): the direct-address stage then
// reads one record per RUN of equal keys instead of one per item
// kRowS0: a callable s0(q) that computes stage S0 of the items q * 64 + lane (wave scope only).  With minimizer lines, batch b
// of stage SF needs only row b of the tile, so the rows are prepared one batch AHEAD: the S0 arithmetic of row b + 1 runs while
// the lines of row b are in flight.  Callers without it fill the whole tile before they call tile_resolve.
struct NoRowS0 { __device__ __forceinline__ void operator()(int) const {} };
// kFinish = false: the candidates are left in best[] (stages S0-S4 only) for a caller that verifies them itself
// (tile_finish_grouped)
// kSfLanes: lanes that fetch and scan the minimizer line of one query in stage SF: 8 (one 16-byte load each, the line as ONE request
// of the wave) or 4 (two loads each, one per 64-byte half of the line: half as many rounds for the same lines -- fewer
// instructions per query, for kernels that are bound by instruction issue and find most of their lines in the caches)
// kDefer: an item that needs the exact walk is not walked here: its best[] becomes kBestDefer, tile_result() reports MZ_DEFER and the
// kernel queues it (defer_push)
// PreS5: a callable run right before the sequence-record gathers of S5 go out (experiments with work of the NEXT tile placed there)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <class SC, bool kRuns = false, class RowS0 = NoRowS0, bool kFinish = true, int kSfLanes = 8, bool kDefer = false, class PreS5 = NoHook>
__device__ __forceinline__ void tile_resolve(TileLdsT<SC::NT> &S, const IndexView &ix, RowS0 s0 = RowS0{}, PreS5 pre_s5 = PreS5{}) {
    const uint32_t t = SC::tid(), k = ix.k, w = ix.w;
    constexpr bool kHasS0 = !std::is_same<RowS0, NoRowS0>::value;
    static_assert(!kHasS0 || (SC::NT == 64 && !kRuns), "row-wise S0 is for the wave-scoped point kernel");
    const bool sf_rows = kHasS0 && ix.front.lines != nullptr; // uniform: rows are prepared inside SF
    if constexpr (kHasS0) {
        if (!sf_rows || kStopAfter == 1) {
#pragma unroll
            for (int q = 0; q < kQpt; ++q) s0(q);
        }
    }
    SC::sync();
    if constexpr (kStopAfter == 1) return;
    // ---- SF: minimizer lines.  ONE gather answers membership and, for most keys, hands over the bucket's entries with their
    //          flank fingerprints: eight lanes fetch the 128-byte line of a query with one coalesced 16-byte load each and
    //          test two slots each against the query's slot pattern (one xor/and/compare per slot).  The lane that holds the
    //          first surviving entry in bucket order, or the key's marker, only parks that slot in the query's LDS words; a
    //          dense pass afterwards (one lane per query, all lanes busy) turns parked slots into the candidate position for
    //          S5, a skew-index request for S4, or -- for a bucket that did not fit its line -- its range in the entry array
    //          for S3.  A query none of whose line's slots carries its key is a miss.
    constexpr bool kRunScope = kRuns && SC::NT == 64;
    if (ix.front.lines != nullptr && !kRunScope) {
#ifndef MZ_SF_BATCH
#define MZ_SF_BATCH 8
#endif
        static_assert(kSfLanes == 8 || kSfLanes == 4, "lanes per line");
        constexpr uint32_t kLpq = kSfLanes, kQpr = 64 / kLpq;  // lanes per query, queries per round
        constexpr uint32_t kHalves = kFrontSlots / 2 / kLpq;    // 16-byte loads per lane: load h covers slots [8 h / kHalves ...) two per lane, i.e. lane `sub` holds slots (16 / kHalves) h + 2 sub, + 1
        constexpr int kFR = MZ_SF_BATCH / (int)kHalves;          // rounds whose lines are in flight together (64 lines)
        constexpr uint32_t kRounds = 64 * kQpt / kQpr;          // per wavefront
        const uint32_t lane = t & 63, sub = lane % kLpq, grp = lane / kLpq;
        const uint32_t wave_q0 = (t >> 6) * (64 * kQpt);      // first item of this wave's share of the tile
        const unsigned long long grp_mask = ((1ull << kLpq) - 1) << (lane & ~(kLpq - 1)), low_mask = grp_mask & ((1ull << lane) - 1); // my query's lanes / those before me
        static_assert(!kHasS0 || kFR * kQpr == 64, "one SF batch per row of the tile");
        if constexpr (kHasS0) { s0(0); SC::sync(); }
#pragma unroll
        for (uint32_t r0 = 0; r0 < kRounds; r0 += kFR) { // (a rolling window of requests instead of batches measured the same: the gathers are bandwidth-bound)
            ulonglong2 sl[kFR][kHalves];
            uint32_t pat[kFR];
#pragma unroll
            for (int a = 0; a < kFR; ++a) { // all loads of the batch first
                const uint64_t lp = S.slot[wave_q0 + (r0 + a) * kQpr + grp];     // line | pattern << 32
                pat[a] = (uint32_t)(lp >> 32);
                const uint64_t *line = ix.front.lines + (uint64_t)(uint32_t)lp * kFrontSlots + 2 * sub;
#pragma unroll
                for (uint32_t h = 0; h < kHalves; ++h) sl[a][h] = *reinterpret_cast<const ulonglong2 *>(line + h * (kFrontSlots / kHalves));
            }
            if constexpr (kHasS0) {
                if (r0 / kFR + 1 < (uint32_t)kQpt) { s0((int)(r0 / kFR) + 1); SC::sync(); } // the next row, while this row's lines are in flight
            }
#pragma unroll
            for (int a = 0; a < kFR; ++a) {
                const uint32_t q = wave_q0 + (r0 + a) * kQpr + grp, msk = S.eoff[q];
                bool first = false, marker = false, seen = false; // seen: an earlier half of the line holds a survivor of my query
                uint64_t first_e = 0, marker_e = 0;
                uint32_t n_alive = 0;
#pragma unroll
                for (uint32_t h = 0; h < kHalves; ++h) {
                    const uint64_t e0 = sl[a][h].x, e1 = sl[a][h].y;
                    const uint32_t x0 = (uint32_t)e0 ^ pat[a], x1 = (uint32_t)e1 ^ pat[a];
                    const bool s0 = (x0 & msk) == 0, s1 = (x1 & msk) == 0; // an entry (kind 0) of this key whose flanks agree
                    const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1);
                    if ((s0 || s1) && !seen && ((b0 | b1) & low_mask) == 0) { // slots run in line order, two per lane and half: this lane holds the first survivor
                        first = true;
                        first_e = s0 ? e0 : e1;
                    }
                    n_alive += (uint32_t)(__popcll(b0 & grp_mask) + __popcll(b1 & grp_mask));
                    if constexpr (kHalves > 1) seen = seen || ((b0 | b1) & grp_mask) != 0;
                    // heavy or spill marker of this key: same key bits, kind 1 or 2 (a key has entries or one marker, never both)
                    const bool m0 = (x0 & (kFrontKeyMask | 3u)) - 1u < 2u, m1 = (x1 & (kFrontKeyMask | 3u)) - 1u < 2u;
                    if (m0 || m1) { marker = true; marker_e = m0 ? e0 : e1; }
                    if (front_is_hashed(w) && h == 0 && sub == 0 && e0 == kFrontUnusable) { // hash-addressed line the builder gave up: exact walk through the MPHF
                        S.best[q] = 0;
                        atomicOr(&S.qinfo[q], kQMulti);
                    }
                }
                if (first) {
                    S.best[q] = first_e;
                    if (n_alive > 1) atomicOr(&S.qinfo[q], kQMulti | kQLine);
                }
                if (marker) {
                    S.slot[q] = marker_e;
                    S.n[q] = kNPending;
                }
            }
        }
        SC::sync();
#pragma unroll
        for (int qq = 0; qq < kQpt; ++qq) { // parked slots -> what the later stages expect
            const uint32_t i = qq * SC::NT + t;
            const unsigned long long b = S.best[i];
            const uint32_t nn = S.n[i];
            if (b != kBestNone) {
                const uint32_t qi = S.qinfo[i], off_a = (qi >> kQOffShift) & kQOffMask;
                const uint32_t cls = (uint32_t)(b >> kFrontClsShift) & 1u;
                const uint64_t p = (b >> kFrontPosShift) - (cls ? k - w - off_a : off_a);
                const bool in_range = p <= ix.total_len - k;          // p wraps around when the k-mer would start before base 0
                S.best[i] = in_range ? p | ((cls != 0) != ((qi & kQClass) != 0) ? kBestTwin : 0ull) : 0;  // out of range (first or last bases of the sequence only): exact walk
                if (!in_range) S.qinfo[i] = qi | kQMulti;
            }
            bool heavy = false;
            if (nn == kNPending) {
                const uint64_t e = S.slot[i];
                const uint32_t cnt = (uint32_t)(e >> kFrontFlankShift) & 0xFFFFu;
                heavy = ((uint32_t)e & 3u) == kFrontHeavy;
                if (heavy) {
                    S.n[i] = kNSkew;
                } else if (cnt == 0xFFFFu) { // size does not fit the marker (an index without a skew index only): exact walk
                    S.best[i] = 0;
                    S.qinfo[i] |= kQMulti;
                    S.n[i] = 0;
                } else {
                    S.slot[i] = e >> kFrontPosShift;
                    S.n[i] = cnt;
                }
            }
            wave_push(S.list_b, &S.cnt_skew, heavy, i);
        }
    } else
    // ---- S1: minimizer -> bucket id.  Direct-address rank bitmap (2w <= 32): one dense gather, no rounds;
    //          otherwise the BBHash level loop
    if (kRunScope) {
        // S1+S2 once per run of equal keys.  Items are numbered along the read (item q * 64 + lane).  The windows whose canonical
        // k-mer is the forward strand keep their minimizer for a stretch of positions, and so do those whose canonical k-mer is
        // the reverse strand, but the two strands interleave at random: an item belongs to the run of the nearest earlier item of
        // ITS strand class when the keys agree, else it leads a new run.  Leaders go into a dense list and are resolved by full
        // waves (record decode or MPHF levels + bucket offsets); the other items copy the bucket range of their leader.
        // The negative filter of a minimizer record is per query and is skipped here: S3 / S5 reject the same items, later.
        const uint32_t lane = t;
        const unsigned long long le_mask = ~0ull >> (63 - lane), lt_mask = le_mask >> 1;
        uint32_t leader_of[kQpt];                                   // item that leads my run (myself for a leader)
        uint32_t n_runs = 0, follows = 0;
        uint32_t last_key0 = 0, last_key1 = 0, last_ldr0 = ~0u, last_ldr1 = ~0u; // per strand class: key of its latest item and that item's leader (uniform)
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint32_t i = q * 64 + lane, qi = S.qinfo[i];
            const bool act = !(qi & kQIdle), cls = (qi & kQClass) != 0;
            const uint64_t key = S.slot[i];
            const unsigned long long act_b = __ballot(act), cls_b = __ballot(cls);
            const unsigned long long mine_b = (cls ? cls_b : ~cls_b) & act_b;       // active items of my class in this row
            const unsigned long long before = mine_b & lt_mask;
            const uint32_t src = before ? 63u - (uint32_t)__clzll(before) : lane;   // nearest earlier one
            const uint64_t pk = (uint64_t)__shfl((unsigned long long)key, (int)src);
            const uint32_t carried = cls ? last_ldr1 : last_ldr0;
            const bool first_of_class = before == 0;
            const bool lead = act && (first_of_class ? carried == ~0u || (uint32_t)key != (cls ? last_key1 : last_key0) || (key >> 32) != 0 : key != pk);
            const unsigned long long lead_b = __ballot(lead);
            const unsigned long long lead_mine = lead_b & mine_b & le_mask;         // leaders of my class at or before me in this row
            leader_of[q] = lead_mine ? q * 64 + 63u - (uint32_t)__clzll(lead_mine) : carried;
            if (lead) S.eoff[n_runs + (uint32_t)__popcll(lead_b & lt_mask)] = i;
            if (act && !lead) follows |= 1u << q;
            if (const unsigned long long cb = ~cls_b & act_b) { // carry each class's latest item over to the next row
                const int lastl = 63 - __clzll(cb);
                last_key0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, lastl);
                last_ldr0 = (uint32_t)__builtin_amdgcn_readlane((int)leader_of[q], lastl);
            }
            if (const unsigned long long cb = cls_b & act_b) {
                const int lastl = 63 - __clzll(cb);
                last_key1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, lastl);
                last_ldr1 = (uint32_t)__builtin_amdgcn_readlane((int)leader_of[q], lastl);
            }
            n_runs += (uint32_t)__popcll(lead_b);
        }
        SC::sync();
#ifdef MZ_DEBUG_COUNT
        if (t == 0) S.dbg[3] = n_runs;
#endif
        if (ix.front.lines != nullptr) {
            // leaders through the minimizer lines: eight lanes fetch the leader's line (one coalesced 16-byte load each) and find
            // the slots of its key -- they are contiguous, in bucket order.  The run's "bucket" is then that stretch of the line
            // itself (slot[] flagged kSlotInLine: S3 reads and decodes slots instead of entries), or what the key's marker says.
            constexpr uint32_t kLpq = kFrontSlots / 2, kQpr = 64 / kLpq;
            const uint32_t sub = lane % kLpq, grp = lane / kLpq, g = ix.front.g, gl = 2 * w - ix.front.t, line_mask = (uint32_t)(ix.front.n_lines - 1);
            const unsigned long long grp_mask = ((1ull << kLpq) - 1) << (lane & ~(kLpq - 1));
            for (uint32_t base = 0; base < n_runs; base += kQpr * 4) {
                ulonglong2 sl[4];
                uint32_t pat[4], item[4];
                uint64_t line0[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { // the loads of four rounds first
                    const uint32_t li = base + a * kQpr + grp;
                    item[a] = li < n_runs ? S.eoff[li] : ~0u;
                    const uint32_t key = item[a] != ~0u ? (uint32_t)S.slot[item[a]] : 0u; // (no leader: line 0 is always there)
                    pat[a] = front_keyfield(key, g, gl) << kFrontKeyShift;
                    line0[a] = (uint64_t)((key >> g) & line_mask) * kFrontSlots;
                    sl[a] = *reinterpret_cast<const ulonglong2 *>(ix.front.lines + line0[a] + 2 * sub);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const uint32_t x0 = (uint32_t)sl[a].x ^ pat[a], x1 = (uint32_t)sl[a].y ^ pat[a];
                    const bool e0 = (x0 & (kFrontKeyMask | 3u)) == 0, e1 = (x1 & (kFrontKeyMask | 3u)) == 0;      // entries of the key
                    const bool m0 = (x0 & (kFrontKeyMask | 3u)) - 1u < 2u, m1 = (x1 & (kFrontKeyMask | 3u)) - 1u < 2u; // its heavy / spill marker
                    const unsigned long long b0 = __ballot(e0) & grp_mask, b1 = __ballot(e1) & grp_mask, mb = __ballot(m0 || m1) & grp_mask;
                    const uint32_t i = item[a];
                    if (i == ~0u) continue;
                    if (m0 || m1) {
                        const uint64_t e = m0 ? sl[a].x : sl[a].y;
                        const uint32_t cnt = (uint32_t)(e >> kFrontFlankShift) & 0xFFFFu;
                        S.slot[i] = e >> kFrontPosShift;
                        S.n[i] = ((uint32_t)e & 3u) == kFrontHeavy ? kNSkew : cnt == 0xFFFFu ? kNExact : cnt; // 0xFFFF: size not in the marker (no skew index)
                    } else if (sub == 0 && mb == 0) {
                        const unsigned long long any = b0 | b1;
                        if (any == 0) {
                            S.n[i] = kNMiss; // no slot of this key in its line: not a minimizer of the set
                        } else {
                            const uint32_t fl = (uint32_t)__ffsll((long long)any) - 1;                    // first lane (of the wave) holding one
                            const uint32_t f = 2 * (fl % kLpq) + (((b0 >> fl) & 1) ? 0u : 1u);          // first slot of the key
                            S.slot[i] = kSlotInLine | (line0[a] + f);
                            S.n[i] = (uint32_t)(__popcll(b0) + __popcll(b1));
                        }
                    }
                }
            }
        } else if (ix.mm.direct) {
            for (uint32_t base = 0; base < n_runs; base += 64) {
                const bool have = base + lane < n_runs;
                const uint32_t i = have ? S.eoff[base + lane] : 0u;
                const uint32_t key = have ? (uint32_t)S.slot[i] : 0u;
                const uint64_t *rec = ix.mm.blocks + (uint64_t)(key >> kMinRecKeysLog2) * kMinRecWords;
                const ulonglong2 pr = *reinterpret_cast<const ulonglong2 *>(rec + 2 * ((key & ((1u << kMinRecKeysLog2) - 1)) >> 6));
                uint64_t L[8];
                const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(rec + 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const ulonglong2 x = v[j]; L[2 * j] = x.x; L[2 * j + 1] = x.y; }
                uint32_t local;
                const bool member = minrec_probe(pr.x, pr.y, key, local);
                uint64_t ps = 0, pe = 0;
                const bool ok = bef_decode_pair(L, local < kBefSpan ? local : 0u, nullptr, ps, pe) && local < kBefSpan;
                if (have && member && !ok) bef_get_pair(ix.buckets, (pr.x & kMinRecRankMask) + local, ps, pe); // the record defers to the global offsets (rare)
                const uint64_t cnt = pe - ps;
                if (have) {
                    S.slot[i] = ps;
                    S.n[i] = !member ? kNMiss : cnt > ix.skew_param ? kNSkew : (uint32_t)(cnt < 0x7FFFFFFFull ? cnt : 0x7FFFFFFFull);
                }
            }
        } else {
            for (uint32_t j = lane; j < n_runs; j += 64) S.list_a[j] = (typename TileLdsT<SC::NT>::ListT)S.eoff[j];
            if (lane == 0) S.cnt_a = n_runs;
            tile_mphf<SC>(S, ix.mm, false);
            for (uint32_t j = lane; j < n_runs; j += 64) {
                const uint32_t i = S.eoff[j];
                if (S.n[i] < kNIdle) {
                    uint64_t ps, pe;
                    bef_get_pair(ix.buckets, S.slot[i], ps, pe);
                    const uint64_t cnt = pe - ps;
                    S.slot[i] = ps;
                    S.n[i] = cnt > ix.skew_param ? kNSkew : (uint32_t)(cnt < 0x7FFFFFFFull ? cnt : 0x7FFFFFFFull);
                }
            }
        }
        SC::sync();
        uint64_t cps[kQpt];
        uint32_t cn[kQpt];
#pragma unroll
        for (int q = 0; q < kQpt; ++q) { // (reads first: a leader's words are another item's source)
            const uint32_t li = (follows >> q) & 1u ? leader_of[q] : q * 64 + lane;
            cps[q] = S.slot[li];
            cn[q] = S.n[li];
        }
        SC::sync();
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint32_t i = q * 64 + lane;
            if ((follows >> q) & 1u) { S.slot[i] = cps[q]; S.n[i] = cn[q]; }
            if (cn[q] == kNExact) { S.best[i] = 0; S.qinfo[i] |= kQMulti; S.n[i] = 0; } // position 0 fails its verification (or is the answer): exact walk
            wave_push(S.list_b, &S.cnt_skew, cn[q] == kNSkew, i);
        }
    } else if (ix.mm.direct) {
        // S1+S2 fused: member test, bucket id and bucket range from ONE 128-byte minimizer record.  Straight-line and in
        // two batches so that the gathers of all kQpt items of a lane are in flight together: (a) the {header, bitmap word}
        // pair of the key, ONE 16-byte load, (b) the bucket-offset half of the same line.  Idle slots probe record 0 (always there).
        uint32_t key[kQpt], local[kQpt];
        uint64_t w0[kQpt];
        uint32_t slow = 0; // items whose record defers to the global offsets (0.15 % of the records at S-human)
#pragma unroll
        for (int q0 = 0; q0 < kQpt; q0 += kRecBatch) { // kRecBatch whole records in flight per lane
            uint64_t wv[kRecBatch], L[kRecBatch][8];
#pragma unroll
            for (int qq = 0; qq < kRecBatch; ++qq) {
                const int q = q0 + qq;
                const uint32_t i = q * SC::NT + t;
                key[q] = S.n[i] < kNIdle ? (uint32_t)S.slot[i] : 0u;
                const uint64_t *rec = ix.mm.blocks + (uint64_t)(key[q] >> kMinRecKeysLog2) * kMinRecWords;
                const ulonglong2 pr = *reinterpret_cast<const ulonglong2 *>(rec + 2 * ((key[q] & ((1u << kMinRecKeysLog2) - 1)) >> 6));
                w0[q] = pr.x;
                wv[qq] = pr.y;
                const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(rec + 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const ulonglong2 x = v[j]; L[qq][2 * j] = x.x; L[qq][2 * j + 1] = x.y; }
            }
#pragma unroll
            for (int qq = 0; qq < kRecBatch; ++qq) {
                const int q = q0 + qq;
                const uint32_t i = q * SC::NT + t;
                const bool act = S.n[i] < kNIdle;
                const bool member = minrec_probe(w0[q], wv[qq], key[q], local[q]);
                uint64_t ps = 0, pe = 0;
                const bool ok = bef_decode_pair(L[qq], local[q] < kBefSpan ? local[q] : 0u, nullptr, ps, pe) && local[q] < kBefSpan;
                // negative filter in the slack of the offset block: the query's inner flank must be among the entries of its bucket
                const uint32_t qi = S.qinfo[i], off_a = (qi >> kQOffShift) & kQOffMask;
                uint32_t fword, fbit;
                const bool filtered = minrec_filter_slot(L[qq][0], local[q], qi, fword, fbit) && off_a >= kFilterFlankBases &&
                                      k - w - off_a >= kFilterFlankBases; // the k-mer spans the inner flanks
                uint32_t fx = 0;
#pragma unroll
                for (uint32_t j = 2; j < 8; ++j) fx |= fword == j ? (uint32_t)(L[qq][j] >> fbit) : 0u; // masked OR: no dynamic register index
                const bool absent = filtered && !(fx & 1u);
                bool heavy = false;
                if (act) {
                    if (!member) {
                        S.n[i] = kNMiss;
                    } else if (ok) {
                        const uint64_t cnt = pe - ps;
                        heavy = cnt > ix.skew_param;                        // sshash.rs:486-490
                        S.slot[i] = ps;
                        S.n[i] = heavy ? kNSkew : absent ? kNMiss : (uint32_t)(cnt < 0x7FFFFFFFull ? cnt : 0x7FFFFFFFull);
                    } else {
                        slow |= 1u << q;
                    }
                }
                wave_push(S.list_b, &S.cnt_skew, heavy, i);
            }
        }
        if (__ballot(slow != 0)) {
#pragma unroll
            for (int q = 0; q < kQpt; ++q) {
                const uint32_t i = q * SC::NT + t;
                bool heavy = false;
                if (slow & (1u << q)) {
                    uint64_t ps, pe;
                    bef_get_pair(ix.buckets, (w0[q] & kMinRecRankMask) + local[q], ps, pe);
                    const uint64_t cnt = pe - ps;
                    heavy = cnt > ix.skew_param;
                    S.slot[i] = ps;
                    S.n[i] = heavy ? kNSkew : (uint32_t)(cnt < 0x7FFFFFFFull ? cnt : 0x7FFFFFFFull);
                }
                wave_push(S.list_b, &S.cnt_skew, heavy, i);
            }
        }
    } else {
        tile_mphf<SC>(S, ix.mm, true);
        // ---- S2: bucket range from the partitioned Elias-Fano prefix sums
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * SC::NT + t;
            bool act = S.n[i] < kNIdle;
            bool heavy = false;
            if (act) {
                uint64_t ps, pe;
                bef_get_pair(ix.buckets, S.slot[i], ps, pe);
                uint64_t cnt = pe - ps;
                heavy = cnt > ix.skew_param;                        // sshash.rs:486-490
                S.slot[i] = ps;
                S.n[i] = heavy ? kNSkew : (uint32_t)(cnt < 0x7FFFFFFFull ? cnt : 0x7FFFFFFFull);
            }
            wave_push(S.list_b, &S.cnt_skew, heavy, i);
        }
    }
    SC::sync();
    if constexpr (kStopAfter == 2) return;
    // ---- S4 (heavy buckets, sshash.rs:415-433), first half: eight lanes fetch the k-mer's 128-byte skew-table bucket with one
    //          coalesced 16-byte load each.  The requests of the first eight heavy items of the wave go out HERE, ahead of S3, so
    //          that the few items S3 and S4 serve in a tile (about three each at S-human) cost the tile ONE dependent round
    //          trip to HBM instead of two.
    struct SkewReq { ulonglong2 sl; uint32_t fp, item; };
    const uint32_t n_heavy = (ix.has_skew && ix.skew.n_buckets != 0) ? S.cnt_skew : 0u; // uniform
    const uint32_t s4_lane = t & 63, s4_sub = s4_lane & 7, s4_grp = s4_lane >> 3;
    auto skew_issue = [&](uint32_t j) {
        SkewReq rq;
        rq.item = j < n_heavy ? (uint32_t)S.list_b[j] : ~0u;
        const uint64_t fw = rq.item != ~0u ? S.fw[rq.item] : 0ull, rc = kmer_rc(fw, k);
        uint64_t hb;
        skew_home(fw < rc ? fw : rc, ix.skew.n_buckets, ix.skew.fp_mask, hb, rq.fp);
        rq.sl = *reinterpret_cast<const ulonglong2 *>(ix.skew.buckets + (rq.item != ~0u ? hb : 0ull) * kSkewSlots + 2 * s4_sub);
        return rq;
    };
    SkewReq s4_first{};
    if (n_heavy != 0) s4_first = skew_issue((t >> 6) * 8 + s4_grp);
    // ---- S3: bucket entries of the whole tile as ONE flat sequence (exclusive scan of the bucket sizes, items blocked by
    //          thread), so lanes are busy whatever the bucket-size distribution.  Chunks of kEpt consecutive entries per lane;
    //          the owner of an entry comes from a max-scan over "bucket head" marks (the item index, dropped at the slot of
    //          its first entry): no per-entry search.
    {
        uint32_t nq[kQpt], eo[kQpt], sum = 0;
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint32_t nn = S.n[t * kQpt + q];
            nq[q] = (nn >= kNSkew) ? 0u : nn;
            eo[q] = sum;
            sum += nq[q];
        }
        const uint32_t incl = wave_scan<false>(sum);
        if constexpr (SC::NT > 64) {
            if ((t & 63) == 63) S.wave_tot[t >> 6] = incl;
            SC::sync();
        }
        uint32_t wbase = 0;
        if constexpr (SC::NT > 64)
            for (uint32_t wv = 0; wv < (t >> 6); ++wv) wbase += S.wave_tot[wv];
        const uint32_t excl = wbase + incl - sum;
#pragma unroll
        for (int q = 0; q < kQpt; ++q) { eo[q] += excl; S.eoff[t * kQpt + q] = eo[q]; }
        if (t == SC::NT - 1) S.eoff[(SC::NT * kQpt)] = excl + sum;
        SC::sync();
        const uint32_t E = S.eoff[(SC::NT * kQpt)];
#ifdef MZ_DEBUG_COUNT
        if (t == 0) S.dbg[0] = E;
#endif
        const uint64_t last_km_start = ix.total_len - k;
        constexpr int kEpt = 4;                    // entries per lane and chunk, their `pos` gathers in flight together
        constexpr uint32_t kChunk = SC::NT * kEpt; // == items per tile, so list_a can hold the marks
        static_assert(kEpt == kQpt, "marks live in list_a");
        uint32_t carry = 0;                        // owner of the entries before the first head of a chunk
        for (uint32_t base = 0; base < E; base += kChunk) {
#pragma unroll
            for (int a = 0; a < kEpt; ++a) S.list_a[t * kEpt + a] = 0;
            SC::sync();
#pragma unroll
            for (int q = 0; q < kQpt; ++q)
                if (nq[q] != 0 && eo[q] >= base && eo[q] - base < kChunk) S.list_a[eo[q] - base] = (typename TileLdsT<SC::NT>::ListT)(t * kQpt + q);
            SC::sync();
            uint32_t own[kEpt];
#pragma unroll
            for (int a = 0; a < kEpt; ++a) {
                own[a] = S.list_a[t * kEpt + a];
                if (a) own[a] = own[a] > own[a - 1] ? own[a] : own[a - 1];
            }
            const uint32_t run = wave_scan<true>(own[kEpt - 1]); // inclusive max-scan over lanes (item indices grow with the entry index)
            uint32_t before = wave_prev(run);
            uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)run, 63);
            if constexpr (SC::NT > 64) {
                if ((t & 63) == 63) S.wave_tot[t >> 6] = run;
                SC::sync();
                total = 0;
                for (uint32_t wv = 0; wv < SC::NT / 64; ++wv) {
                    const uint32_t wt = S.wave_tot[wv];
                    if (wv < (t >> 6) && wt > before) before = wt;
                    if (wt > total) total = wt;
                }
            }
            if (carry > before) before = carry;
            if (carry < total) carry = total;
            // owners back into the mark array, then entries are taken strided (lane t: base + a * NT + t) so that one gather
            // instruction covers consecutive entries: entries of one bucket share their lines within ONE instruction
#pragma unroll
            for (int a = 0; a < kEpt; ++a) S.list_a[t * kEpt + a] = (typename TileLdsT<SC::NT>::ListT)(own[a] < before ? before : own[a]);
            SC::sync();
            uint32_t jdx[kEpt];
            uint64_t ent[kEpt];
            bool canon[kEpt] = {}; // the entry's flanks are already in the orientation of the canonical k-mer (it came from a minimizer line)
#pragma unroll
            for (int a = 0; a < kEpt; ++a) own[a] = S.list_a[a * SC::NT + t];
            SC::sync(); // the mark array is rewritten by the next chunk
#pragma unroll
            for (int a = 0; a < kEpt; ++a) { // no branches: the kEpt gathers of a lane are in flight together (slots past E re-read entry E-1)
                const uint32_t e = base + a * SC::NT + t;
                const uint32_t ec = e < E ? e : E - 1;
                if (e >= E) own[a] = carry;
                jdx[a] = ec - S.eoff[own[a]];
                const uint64_t sb = S.slot[own[a]];
                if constexpr (kRunScope) { // a run resolved through the minimizer lines reads its entries from the line itself
                    const uint64_t *src = (sb & kSlotInLine) ? ix.front.lines : ix.pos;
                    const uint64_t e = src[(sb & ~kSlotInLine) + jdx[a]];
                    // slot -> entry layout (the slot's flanks are already in the orientation of the canonical k-mer: bit 63 of jdx marks that)
                    ent[a] = (sb & kSlotInLine) ? make_entry(e >> kFrontPosShift, (uint32_t)(e >> kFrontClsShift) & 1u, (uint32_t)(e >> kFrontFlankShift) & 0xFFFFu) : e;
                    canon[a] = (sb & kSlotInLine) != 0;
                } else {
                    ent[a] = ix.pos[sb + jdx[a]];                                                   // sshash.rs:495
                }
            }
#pragma unroll
            for (int a = 0; a < kEpt; ++a) {
                if (base + a * SC::NT + t >= E) continue;
                const uint32_t i = own[a];
                const bool cls = (ent[a] >> kEntryFlankBits) & 1;
                const uint64_t mm_pos = ent[a] >> kEntryShift;
                const uint32_t qi = S.qinfo[i];
                const uint32_t off_a = (qi >> kQOffShift) & kQOffMask, room = k - w - off_a;
                const uint32_t off = cls ? room : off_a;                            // class 1: candidate B (:527), class 0: A (:498)
                // flank filter: foreign entries of the bucket are dropped without a sequence gather
                const uint32_t nl = off_a < kFlankBases ? off_a : kFlankBases, nr = room < kFlankBases ? room : kFlankBases;
                const uint32_t mask = ((0xFFu << (2 * (kFlankBases - nl))) & 0xFFu) | (((1u << (2 * nr)) - 1) << 8);
                const uint32_t fl = (uint32_t)(ent[a] & 0xFFFFu);
                if ((((cls && !canon[a] ? flank_revcomp(fl) : fl) ^ qi) & mask) != 0 || mm_pos < off || mm_pos - off > last_km_start) continue;
                // the first survivor in bucket order is the item's candidate (verified in S5); a second one is only flagged
                const uint64_t jj = jdx[a] < 0x3FFFFFFu ? jdx[a] : 0x3FFFFFFu;
                const unsigned long long old = atomicMin(&S.best[i], (jj << kBestOrderShift) | (cls != ((qi & kQClass) != 0) ? kBestTwin : 0ull) | (mm_pos - off));
                if (old != kBestNone && !(qi & kQMulti)) S.qinfo[i] = qi | kQMulti;
            }
        }
    }
    SC::sync();
    if constexpr (kStopAfter == 3) return;
    // ---- S4, second half: the slot with the k-mer's fingerprint gives the candidate position for S5 (slots run in lane order,
    //          two per lane: the first matching one counts).  More than one such slot, or a full bucket (its keys may have
    //          overflowed, and the one matching slot may then be a mere fingerprint collision), raises the item's flag: if S5
    //          cannot verify the candidate it falls back to the exact walk, which also searches the overflow list.
    if (n_heavy != 0) { // uniform
        const unsigned long long grp_mask = 0xFFull << (s4_lane & ~7u);
        auto skew_finish = [&](const SkewReq &rq) {
            const uint64_t e0 = rq.sl.x, e1 = rq.sl.y;
            const bool emp0 = e0 == kSkewEmpty, emp1 = e1 == kSkewEmpty;
            const bool m0 = !emp0 && (uint32_t)(e0 >> kSkewPosBits) == rq.fp, m1 = !emp1 && (uint32_t)(e1 >> kSkewPosBits) == rq.fp;
            const unsigned long long b0 = __ballot(m0) & grp_mask, b1 = __ballot(m1) & grp_mask, be = __ballot(emp0 || emp1) & grp_mask;
            if (rq.item == ~0u) return;
            const uint32_t n_match = (uint32_t)(__popcll(b0) + __popcll(b1));
            if (const unsigned long long any = b0 | b1) {
                const uint32_t fl = (uint32_t)__ffsll((long long)any) - 1;
                if (s4_lane == fl) S.best[rq.item] = ((((b0 >> fl) & 1) ? e0 : e1) & ((1ull << kSkewPosBits) - 1)) | kBestUnk;
            }
            if (s4_sub == 0 && (be == 0 || n_match > 1)) {
                S.qinfo[rq.item] |= kQMulti;
                if (n_match == 0) S.best[rq.item] = 0; // any valid position: its verification fails and the flag sends the item to the exact walk
            }
        };
        skew_finish(s4_first);
        for (uint32_t base = (t >> 6) * 8 + (SC::NT / 64) * 8; base < n_heavy; base += (SC::NT / 64) * 8) skew_finish(skew_issue(base + s4_grp));
        SC::sync();
    }
    if constexpr (kStopAfter == 4) return;
    if constexpr (!kFinish) return;
    // ---- S5: every item has at most one recorded candidate position.  ONE gather of its 64-byte sequence+boundary record
    //          verifies the k-mer (sshash.rs:500-504) and yields unitig id / offset / length (:509-514); the answer is parked
    //          in the item's own slot[] / n[] words, which are dead by now.  Candidates are compacted first, so the gathers
    //          are issued by full waves, two records in flight per lane.
#pragma unroll
    for (int q = 0; q < kQpt; ++q) {
        uint32_t i = q * SC::NT + t;
        wave_push(S.list_a, &S.cnt_a, S.best[i] != kBestNone, i);
    }
    SC::sync();
    pre_s5();
    {
        const uint32_t nc = S.cnt_a;
#ifdef MZ_DEBUG_COUNT
        if (t == 0) { S.dbg[1] = nc; S.dbg[2] = 0; S.dbg[3] = 0; S.dbg[4] = 0; S.dbg[5] = 0; }
        SC::sync();
#endif
#ifndef MZ_CPT
#define MZ_CPT 2
#endif
        constexpr int kCpt = MZ_CPT; // candidate records in flight per lane
        for (uint32_t j0 = t; j0 < ((nc + SC::NT - 1) / SC::NT) * SC::NT; j0 += SC::NT * kCpt) {
            uint32_t ci[kCpt];
            uint64_t cp[kCpt], kw[kCpt];
            SeqBoundary bd[kCpt];
            bool act[kCpt];
#pragma unroll
            for (int a = 0; a < kCpt; ++a) { // no branches: idle slots read the record of position 0
                const uint32_t j = j0 + a * SC::NT;
                act[a] = j < nc;
                ci[a] = S.list_a[act[a] ? j : 0u];
                cp[a] = act[a] ? S.best[ci[a]] & ((1ull << kPosBits) - 1) : 0ull;
                kw[a] = read_kmer_i(ix.seqbv, cp[a], k);
                bd[a] = load_boundary(ix.seqbv, cp[a]);
            }
#pragma unroll
            for (int a = 0; a < kCpt; ++a) {
                if (!act[a]) continue;
                const uint32_t i = ci[a];
                const uint64_t fw = S.fw[i], rc = kmer_rc(fw, k);
                const uint32_t mt = word_equivalency(fw, rc, kw[a]);
                const UnitigSpan sp = locate_boundary(bd[a], cp[a]);
                mz_k2upos r = make_none();
                bool defer = false;
                if (mt != MZ_NO_MATCH && cp[a] + k <= sp.end) r = make_hit(sp, cp[a], mt);
                else if (const uint32_t qi = S.qinfo[i]; qi & kQMulti) { // more than one survivor and the first one failed: exact in-order walk
                    if constexpr (kDefer) {
                        defer = true;
                    } else {
#ifndef MZ_NO_WALK
                        r = exact_walk(ix, fw, qi);
#endif
#ifdef MZ_DEBUG_COUNT
                        atomicAdd(&S.dbg[2], 1u);
#endif
                    }
                }
                if (r.match != MZ_NO_MATCH) {
                    S.slot[i] = ((uint64_t)r.unitig_id << 32) | r.unitig_len;
                    S.n[i] = r.pos;
                    S.best[i] = r.match == MZ_TWIN_MATCH ? 1ull : 0ull;
                } else {
                    S.best[i] = defer ? kBestDefer : kBestNone;
                }
            }
        }
    }
    SC::sync();
    if (t == 0) S.cnt_a = 0;
}

// answer of item i (any thread may ask for any item after tile_resolve)
template <class SC>
__device__ __forceinline__ mz_k2upos tile_result(const TileLdsT<SC::NT> &S, const IndexView &ix, uint32_t i) {
    (void)ix;
    const unsigned long long b = S.best[i];
    if constexpr (kStopAfter != 0) return mz_k2upos{(uint32_t)S.slot[i], S.n[i], (uint32_t)b, MZ_NO_MATCH};
    if (b == kBestNone) return make_none();
    if (b == kBestDefer) return mz_k2upos{0u, 0u, 0u, MZ_DEFER};
    const uint64_t s = S.slot[i];
    return mz_k2upos{(uint32_t)(s >> 32), S.n[i], (uint32_t)s, b ? MZ_TWIN_MATCH : MZ_IDENTITY_MATCH};
}

// ---- S5 for tiles whose items are consecutive base positions of reads (wave scope): verify and extend ------------------------
// The windows of a read that hit lie at consecutive positions of a unitig -- ascending when the read runs along the stored
// strand, descending when it runs against it, and on another unitig after a junction.  index/caching.rs:73-97 uses this one
// window at a time and in one direction only (try pos + 1 on the same unitig, else a full lookup).  Here a whole stretch is
// verified at once, in either direction: the candidates of the tile are grouped by (orientation, diagonal, sequence record) --
// diagonal = candidate position minus the window's position in the tile for a read along the stored strand, plus it for
// a read against it; a window whose candidate continues the group of the nearest earlier candidate joins it.  The first window
// of a group (its leader) fetches the 64-byte sequence+boundary record ONCE and XORs its 128 bases with the tile's own bases
// laid along the group's diagonal (the reverse-complemented tile for a read against the strand): the XOR is zero exactly where
// read and unitig agree, so it holds the matching stretch of the whole group, however far it extends to either side of the
// leader.  Every window of the group then tests the 2k bits of its own k-mer in the stashed XOR (sshash.rs:500-504) and
// locates its unitig from the stashed boundary half (:509-514).  One dependent round trip to HBM per tile and a handful of
// gathers (one per ~27 candidates on the GRCh38-shaped reads), where the per-candidate form (stage S5 of tile_resolve) makes
// two round trips for ~180 gathers; and no window computes a reverse complement or compares k-mer words.  Every answer is
// still that of the window's own lookup: a window joins a group only through the candidate its own lookup produced.
// Candidates whose orientation is unknown (skew table) lead a group of their own with the raw record stashed.
// The stash reuses words of the tile that are dead after stage S4: slot[] (32 records) and n[] + eoff[] (32 more).
// `aligned`: words 0..14 = three pad words + the tile's 2-bit packed bases (12 words: 256 positions, the k-1 tail, junk), words
// 15..30 = the reverse complement of those 12 words (position x' there = complement of tile position 383 - x') + 4 junk words.
// Results come back in registers: res[q] answers item q * 64 + lane.
constexpr uint32_t kAlignPad = 3, kAlignTileWords = kQpt * 2 + 4, kAlignRcAt = kAlignPad + kAlignTileWords, kAlignWords = (kAlignRcAt + kAlignTileWords + 4 + 1) & ~1u;
// Stash slot of a group (12 words): w0..w3 = record bases xor the tile's bases along the diagonal (the raw record for a candidate of
// unknown orientation); w4, w5 = boundary bits of the record's 128 bases; w6, w7 = per 32-base quarter j = 0..2 of the record a
// 32-bit table { boundaries below the quarter | (position + 1 of the highest boundary below the quarter, 0 = none) << 8 |
// (position + 1 of the lowest boundary above the quarter, 0 = none) << 16 } so that a window locates its unitig from ONE 32-bit
// piece of the boundary bits (rank, start and end as locate_unitig_r gives them), then the id of the unitig holding the record's
// first base; w8 = start of that unitig, w9 = end of the unitig holding the record's last base.
// The stash lives in words that are dead by now: n[] + eoff[] of the tile (21 slots) and `stash_b`, kStashBWords words lent by
// the caller (the window keys of the reads kernel: 24 slots).
constexpr uint32_t kStashWords = 12, kStashA = 2048 / (8 * kStashWords), kStashBWords = 2 * (64 * kQpt + 32) / 2, kStashB = kStashBWords / kStashWords, kStashSlots = kStashA + kStashB;
// emit(q, r): called once for every item q * 64 + lane with its answer r -- except the items of the returned mask (bit q: item
// q * 64 + lane), which need the exact in-order walk (the caller queues them: defer_push)
template <class SC, class Emit>
__device__ __forceinline__ uint32_t tile_finish_grouped(TileLdsT<SC::NT> &S, const IndexView &ix, const uint64_t *aligned, uint64_t *stash_b, Emit emit) {
    static_assert(SC::NT == 64, "items are numbered along the wave's tile");
    using Lds = TileLdsT<SC::NT>;
    static_assert(offsetof(Lds, eoff) == offsetof(Lds, n) + sizeof(uint32_t) * 64 * kQpt && offsetof(Lds, n) % 16 == 0 && sizeof(uint32_t) * (2 * 64 * kQpt + 1) >= 2048 &&
                      (kStashWords * 8) % 16 == 0, "stash slots in n[] + eoff[]");
    const uint32_t lane = SC::tid(), k = ix.k;
    constexpr uint64_t kPosMask = (1ull << kPosBits) - 1;
    constexpr uint32_t kTileBases = 32 * kAlignTileWords, kNoSlot = ~0u;
    MZ_MARK("FIN_A");
    uint32_t sidx[kQpt];            // stash slot (number of the group) of my window's candidate; kNoSlot: none
    uint32_t n_lead = 0, carry_a = ~0u, carry_b = 0; // uniform; groups so far, key of the last window of the row before
#pragma unroll
    for (int q = 0; q < kQpt; ++q) {
        const uint32_t i = q * 64 + lane;
        const unsigned long long b = S.best[i];
        const bool has = b != kBestNone, twin = (b & kBestTwin) != 0, unk = has && (b & kBestUnk) != 0;
        const uint64_t p = b & kPosMask;
        uint32_t off;
        // key: record | orientation << 30 | unknown << 31 (records are < 2^30), and the low half of the diagonal (two candidates in
        // one record lie fewer than 128 positions apart, so equal low halves are equal diagonals).  A window joins the group of the
        // window before it when the keys agree (a window without a candidate has a key of its own and ends a group).
        const uint32_t ka = has ? (uint32_t)seq_rec_of(p, off) | (twin ? 1u << 30 : 0u) | (unk ? 1u << 31 : 0u) : ~0u;
        const uint32_t kb = twin ? (uint32_t)p + i : (uint32_t)p - i;
        uint32_t pa = wave_prev(ka), pb = wave_prev(kb);
        if (lane == 0) { pa = carry_a; pb = carry_b; }
        const bool lead = has && (pa != ka || pb != kb || unk);
        const unsigned long long lead_b = __ballot(lead);
        // leaders at or before me in this row; with none, the group that came over from the row before is the latest one
        const uint32_t slot = n_lead + __builtin_amdgcn_mbcnt_hi((uint32_t)(lead_b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lead_b, 0u)) + (lead ? 1u : 0u) - 1u;
        sidx[q] = has ? slot : kNoSlot;
        if (lead) S.list_a[slot] = (typename Lds::ListT)i;
        carry_a = (uint32_t)__builtin_amdgcn_readlane((int)ka, 63);
        carry_b = (uint32_t)__builtin_amdgcn_readlane((int)kb, 63);
        n_lead += (uint32_t)__popcll(lead_b);
    }
    MZ_MARK("FIN_A_END");
    SC::sync();
#ifdef MZ_DEBUG_COUNT
    if (lane == 0) { S.dbg[1] = n_lead; S.dbg[2] = 0; }
#endif
    auto stash = [&](uint32_t j) -> uint64_t * { return j < kStashA ? reinterpret_cast<uint64_t *>(S.n) + j * kStashWords : stash_b + (j - kStashA) * kStashWords; };
    uint32_t walk = 0; // bit q: item q * 64 + lane needs the exact in-order walk
    for (uint32_t base = 0; base < n_lead; base += kStashSlots) { // uniform; one round unless more than kStashSlots groups start in the tile
        {
            MZ_MARK("FIN_LEAD");
            const bool act = lane < kStashSlots && base + lane < n_lead;
            const uint32_t item = S.list_a[act ? base + lane : 0u];
            const unsigned long long b = S.best[item];
            const uint64_t p = (b & kPosMask) <= ix.total_len - k ? b & kPosMask : 0ull; // (a candidate is always a k-mer position of the sequence; never trust a gather address blindly)
            uint32_t off;
            const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(ix.seqbv + seq_rec_of(p, off) * kSeqRecWords); // idle lanes re-read the first leader's record
            // the tile's bases along the diagonal: record base y <-> aligned position a0 + y (taken from LDS before the record is requested:
            // fewer registers are live while the loads are in flight)
            const bool twin = (b & kBestTwin) != 0;
            const uint32_t a0 = twin ? 32 * kAlignRcAt + (kTileBases - item - k) - off : 32 * kAlignPad + item - off; // in bases from word 0 of `aligned`
            const uint32_t W = a0 >> 5, sh = 2 * (a0 & 31);
            uint64_t X[4];
            {
                uint64_t A[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) A[j] = aligned[W + j];
                const uint64_t raw = (b & kBestUnk) ? 0ull : ~0ull;
#pragma unroll
                for (int j = 0; j < 4; ++j) X[j] = ((A[j] >> sh) | ((A[j + 1] << 1) << (63 - sh))) & raw;
            }
            const ulonglong2 r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
            // boundary tables of the quarters 0..2 (a window starts in the first 96 bases of its record)
            const uint32_t q0 = (uint32_t)r3.x, q1 = (uint32_t)(r3.x >> 32), q2 = (uint32_t)r3.y, q3 = (uint32_t)(r3.y >> 32);
            const uint32_t c1 = (uint32_t)__popc(q0), c2 = c1 + (uint32_t)__popc(q1);
            const uint32_t hi1 = q0 ? 32u - (uint32_t)__clz((int)q0) : 0u, hi2 = q1 ? 64u - (uint32_t)__clz((int)q1) : hi1; // highest boundary below quarter 1 / 2
            const uint32_t lo3 = q3 ? 96u + (uint32_t)__ffs((int)q3) : 0u, lo2 = q2 ? 64u + (uint32_t)__ffs((int)q2) : lo3, lo1 = q1 ? 32u + (uint32_t)__ffs((int)q1) : lo2; // lowest boundary from quarter 3 / 2 / 1 on
            const uint32_t t0 = lo1 << 16, t1 = c1 | (hi1 << 8) | (lo2 << 16), t2 = c2 | (hi2 << 8) | (lo3 << 16);
            if (act) {
                ulonglong2 *st = reinterpret_cast<ulonglong2 *>(stash(lane));
                st[0] = ulonglong2{r0.x ^ X[0], r0.y ^ X[1]};
                st[1] = ulonglong2{r1.x ^ X[2], r1.y ^ X[3]};
                st[2] = r3;
                st[3] = ulonglong2{t0 | ((uint64_t)t1 << 32), t2 | ((r2.x & 0xFFFFFFFFull) << 32)};
                st[4] = ulonglong2{(r2.x >> 32) | ((r2.y >> 36) << 32), r2.y & ((1ull << 36) - 1)};
            }
        }
        MZ_MARK("FIN_LEAD_END");
        SC::sync();
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            MZ_MARK("FIN_WIN");
            if (sidx[q] - base >= kStashSlots) continue;        // no candidate, or a group of another round
            const uint32_t i = q * 64 + lane;
            const uint64_t *st = stash(sidx[q] - base);
            const unsigned long long b = S.best[i];
            const uint64_t p = b & kPosMask;
            uint32_t off;
            const uint64_t rbase = seq_rec_of(p, off) * kSeqRecStride;
            const uint32_t wi = off >> 5, sh = 2 * (off & 31);
            const uint64_t lo = st[wi], hi = st[wi + 1];
            const uint64_t x = ((lo >> sh) | ((hi << 1) << (63 - sh))) & kmer_mask(k);
            uint32_t mt = x == 0 ? ((b & kBestTwin) ? MZ_TWIN_MATCH : MZ_IDENTITY_MATCH) : MZ_NO_MATCH;
            if (__ballot((b & kBestUnk) != 0)) { // (uniform, rows with heavy windows only) a skew-table candidate says nothing about its orientation: compare words
                const uint32_t ti = 32 * kAlignPad + i, tw = ti >> 5, ts = 2 * (ti & 31);
                const uint64_t fw = ((aligned[tw] >> ts) | ((aligned[tw + 1] << 1) << (63 - ts))) & kmer_mask(k);
                if (b & kBestUnk) mt = word_equivalency(fw, kmer_rc(fw, k), x);
            }
            // unitig of the window from the quarter of the boundary bits that holds its first base
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(st);
            const uint32_t dq = s32[8 + wi], tab = s32[12 + wi], id0 = s32[15], bi = off & 31;
            const uint32_t below = dq & ~(~0u << bi), above = dq >> bi;
            const uint32_t rank = (tab & 0xFFu) + (uint32_t)__popc(below);
            const uint32_t srel = below ? 32 * wi + 32 - (uint32_t)__clz((int)below) : (tab >> 8) & 0xFFu; // position + 1 of the last boundary before the window, 0: none in the record
            const uint32_t erel = above ? off + (uint32_t)__ffs((int)above) : tab >> 16;                    // position + 1 of the first boundary at or after it, 0: none
            const ulonglong2 se = *reinterpret_cast<const ulonglong2 *>(st + 8);
            UnitigSpan sp;
            sp.id = id0 + rank;
            sp.start = srel ? rbase + srel : se.x;
            sp.end = erel ? rbase + erel : se.y;
            MZ_MARK("FIN_WIN_LOC");
            if (mt != MZ_NO_MATCH && p + k <= sp.end) { emit(q, make_hit(sp, p, mt)); sidx[q] = kNoSlot - 1; } // (answered)
            else if (S.qinfo[i] & kQMulti) { walk |= 1u << q; sidx[q] = kNoSlot - 1; }
        }
        MZ_MARK("FIN_WIN_END");
        SC::sync(); // the next round overwrites the stash
    }
#pragma unroll
    for (int q = 0; q < kQpt; ++q)
        if (sidx[q] != kNoSlot - 1) emit(q, make_none()); // no candidate, or one that failed
    MZ_MARK("FIN_END");
    return walk;
}

// ---- deferred exact walks ----------------------------------------------------------------------------------------------
// An item whose recorded candidate failed although more entries of its bucket had agreeing flanks needs the exact in-order
// walk of the bucket (sshash.rs:494-552): a chain of dependent loads that ONE lane follows while the other 63 of its wave
// wait -- about 15 microseconds a time, for 0.3 % of the windows of the GRCh38-shaped reads, which used to be a fifth of
// that kernel's time.  The wave-scoped kernels therefore do not walk on the spot: the item (where its answer goes, its forward
// word, what the walk needs of its flags) is queued in a small per-wave list, and when kDeferCap have gathered -- and at the
// end of the kernel -- one pass walks them all, a lane each, their loads in flight together.
constexpr uint32_t kDeferCap = 16;
struct DeferLds {
    uint64_t where[kDeferCap]; // index of the answer in the output array | minimizer offset << 58 | kQLine << 63
    uint64_t fw[kDeferCap];
};
// walks the queued items: lane j takes entry j; store(where, r) writes an answer
template <class Store>
__device__ __forceinline__ void defer_flush(const DeferLds &D, uint32_t cnt, const IndexView &ix, Tally &tally, Store store) {
    const uint32_t lane = __lane_id();
    const bool act = lane < cnt;
    mz_k2upos r = make_none();
    if (act) {
        const uint64_t wh = D.where[lane];
        r = exact_walk(ix, D.fw[lane], (uint32_t)((wh >> 58) & kQOffMask) << kQOffShift | ((wh >> 63) ? kQLine : 0u));
        store(wh & ((1ull << 58) - 1), r);
    }
    tally.add(act && r.match != MZ_NO_MATCH, act && r.match == MZ_NO_MATCH, false);
    WaveScope::sync();
}
// queues the items of the lanes with `defer` (cnt: entries in the list, wave-uniform); a full list is flushed on the way
template <class Flush>
__device__ __forceinline__ void defer_push(DeferLds &D, uint32_t &cnt, bool defer, uint64_t where, uint64_t fw, uint32_t qi, Flush flush) {
    unsigned long long db = __ballot(defer);
    if (db == 0) return;
    const uint32_t lane = __lane_id();
    const uint64_t wh = where | ((uint64_t)((qi >> kQOffShift) & kQOffMask) << 58) | ((qi & kQLine) ? 1ull << 63 : 0ull);
    while (db) {
        const uint32_t room = kDeferCap - cnt, rank = (uint32_t)__popcll(db & ((1ull << lane) - 1));
        if (((db >> lane) & 1) && rank < room) { D.where[cnt + rank] = wh; D.fw[cnt + rank] = fw; }
        const uint32_t n = (uint32_t)__popcll(db), take = n < room ? n : room;
        cnt += take;
        for (uint32_t a = 0; a < take; ++a) db &= db - 1; // the lanes served
        WaveScope::sync();
        if (cnt == kDeferCap) { flush(); cnt = 0; }
    }
}

// KW: compile-time (k << 8 | w) for the common configuration (constant shifts, masks and trip counts;
// fully unrolled minimizer loop), 0 = read k and w from the index
template <uint32_t KW>
__device__ __forceinline__ void fix_kw(IndexView &ix) {
    if (KW != 0) { ix.k = KW >> 8; ix.w = KW & 255; }
}
constexpr uint32_t kKW3115 = (31u << 8) | 15u;

template <uint32_t KW>
__global__ __launch_bounds__(kBlock) void k2u_tile_kernel(IndexView ix, const uint64_t *__restrict__ fw_words, uint64_t n,
                                                          mz_k2upos *__restrict__ out, mz_counts *counts) {
    fix_kw<KW>(ix);
    __shared__ TileLds S;
    const uint32_t t = threadIdx.x, k = ix.k, w = ix.w;
    const uint64_t kmask = kmer_mask(k);
    const uint64_t n_tiles = (n + kTileQ - 1) / kTileQ;
    Tally tally;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * kTileQ;
        const uint32_t n_valid = (uint32_t)(n - base < (uint64_t)kTileQ ? n - base : (uint64_t)kTileQ);
        tile_begin<BlockScope>(S);
        // ---- S0: canonical encode + minimizer
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            uint64_t fw = i < n_valid ? (fw_words[base + i] & kmask) : 0;
            uint64_t rc = kmer_rc(fw, k);
            tile_set_query<BlockScope>(S, i, fw, rc, canonical_minimizer(fw, rc, k, w, ix.seed), ix, i < n_valid, ix.front.lines != nullptr);
        }
        tile_resolve<BlockScope>(S, ix);
        // ---- S5: coalesced 16-byte stores
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            bool act = i < n_valid;
            mz_k2upos r = act ? tile_result<BlockScope>(S, ix, i) : make_none();
            if (act) out[base + i] = r;
            tally.add(act && r.match != MZ_NO_MATCH, act && r.match == MZ_NO_MATCH, false);
        }
        __syncthreads();
    }
    tally.flush(counts);
}

// Wave-scoped form of the same kernel: every wavefront owns a 256-query tile in its own LDS slice and walks the
// stages on its own -- no workgroup barriers, so a wave waiting for its gathers never holds three others back.
template <uint32_t KW>
__global__ __launch_bounds__(kBlock, MZ_WAVES_PER_EU) void k2u_wave_kernel(IndexView ix, const uint64_t *__restrict__ fw_words, uint64_t n,
                                                          mz_k2upos *__restrict__ out, mz_counts *counts) {
    fix_kw<KW>(ix);
    using SC = WaveScope;
    constexpr int kWT = SC::NT * kQpt; // queries per wave tile
    __shared__ TileLdsT<SC::NT> SS[kBlock / 64];
    __shared__ DeferLds DD[kBlock / 64];
    TileLdsT<SC::NT> &S = SS[threadIdx.x >> 6];
    DeferLds &D = DD[threadIdx.x >> 6];
    const uint32_t t = SC::tid(), k = ix.k, w = ix.w;
    const uint64_t kmask = kmer_mask(k);
    const uint64_t n_tiles = (n + kWT - 1) / kWT;
    const uint64_t wave_id = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (kBlock / 64);
    Tally tally;
    uint32_t n_defer = 0; // queued exact walks (wave-uniform)
    auto flush = [&]() { defer_flush(D, n_defer, ix, tally, [&](uint64_t where, const mz_k2upos &r) { out[where] = r; }); };
    for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const uint64_t base = tile * kWT;
        const uint32_t n_valid = (uint32_t)(n - base < (uint64_t)kWT ? n - base : (uint64_t)kWT);
        tile_begin<SC>(S);
        uint64_t fwq[kQpt];
#pragma unroll
        for (int q = 0; q < kQpt; ++q) { // all query words of the lane in flight together
            const uint32_t i = q * SC::NT + t;
            fwq[q] = fw_words[base + (i < n_valid ? i : 0u)];
        }
        auto s0_row = [&](int q) {
            const uint32_t i = q * SC::NT + t;
            const uint64_t fw = i < n_valid ? (fwq[q] & kmask) : 0;
            const uint64_t rc = kmer_rc(fw, k);
            tile_set_query<SC>(S, i, fw, rc, canonical_minimizer(fw, rc, k, w, ix.seed), ix, i < n_valid, ix.front.lines != nullptr);
        };
#if MZ_POINT_TOUCH
        // experiment: the lines of the NEXT tile's first row are touched (one 4-byte load per lane) right before this tile's record gathers go
        // out, so that stage SF of the next tile finds them in the L2
        uint32_t touched = 0;
        auto touch_next = [&]() {
            const uint64_t nt = tile + n_waves;
            if (nt < n_tiles && ix.front.lines != nullptr) {
                const uint64_t nb = nt * kWT, left = n - nb;
                const uint64_t fw = fw_words[nb + (t < left ? t : 0u)] & kmask, rc = kmer_rc(fw, k);
                const Minimizer mn = canonical_minimizer(fw, rc, k, w, ix.seed);
                const uint64_t line = front_is_hashed(w) ? front_hashed_line(front_hash(mn.word), ix.front.n_lines)
                                                         : (uint64_t)(((uint32_t)mn.word >> ix.front.g) & (uint32_t)(ix.front.n_lines - 1));
                touched = *reinterpret_cast<const uint32_t *>(ix.front.lines + line * kFrontSlots);
            }
        };
        tile_resolve<SC, false, decltype(s0_row), true, 8, true, decltype(touch_next)>(S, ix, s0_row, touch_next);
        asm volatile("" ::"v"(touched));
#elif MZ_ROW_S0
        tile_resolve<SC, false, decltype(s0_row), true, 8, true>(S, ix, s0_row);
#else
#pragma unroll
        for (int q = 0; q < kQpt; ++q) s0_row(q);
        tile_resolve<SC, false, NoRowS0, true, 8, true>(S, ix);
#endif
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * SC::NT + t;
            bool act = i < n_valid;
            mz_k2upos r = act ? tile_result<SC>(S, ix, i) : make_none();
            const bool defer = act && r.match == MZ_DEFER; // answered later, by the pass over the wave's deferred exact walks
            if (act && !defer) out[base + i] = r;
            tally.add(act && !defer && r.match != MZ_NO_MATCH, act && !defer && r.match == MZ_NO_MATCH, false);
            defer_push(D, n_defer, defer, base + i, S.fw[i], S.qinfo[i], flush);
        }
        SC::sync();
    }
    if (n_defer) flush();
    tally.flush(counts);
}

// hits only, compacted per wavefront with a ballot + prefix popcount (K7); same tile machinery
__global__ __launch_bounds__(kBlock) void k2u_compact_kernel(IndexView ix, const uint64_t *__restrict__ fw_words, uint64_t n,
                                                             uint64_t *__restrict__ hit_idx, mz_k2upos *__restrict__ hits,
                                                             unsigned long long *n_hits) {
    __shared__ TileLds S;
    const uint32_t t = threadIdx.x, k = ix.k, w = ix.w;
    const uint64_t kmask = kmer_mask(k);
    const uint64_t n_tiles = (n + kTileQ - 1) / kTileQ;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * kTileQ;
        const uint32_t n_valid = (uint32_t)(n - base < (uint64_t)kTileQ ? n - base : (uint64_t)kTileQ);
        tile_begin<BlockScope>(S);
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            uint64_t fw = i < n_valid ? (fw_words[base + i] & kmask) : 0;
            uint64_t rc = kmer_rc(fw, k);
            tile_set_query<BlockScope>(S, i, fw, rc, canonical_minimizer(fw, rc, k, w, ix.seed), ix, i < n_valid, ix.front.lines != nullptr);
        }
        tile_resolve<BlockScope>(S, ix);
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            mz_k2upos r = i < n_valid ? tile_result<BlockScope>(S, ix, i) : make_none();
            bool hit = r.match != MZ_NO_MATCH;
            unsigned long long b = __ballot(hit);
            unsigned lane = __lane_id();
            unsigned long long slot0 = 0;
            if (lane == 0 && b) slot0 = atomicAdd(n_hits, (unsigned long long)__popcll(b));
            slot0 = __shfl(slot0, 0);
            if (hit) {
                unsigned long long slot = slot0 + __popcll(b & ((1ull << lane) - 1));
                hit_idx[slot] = base + i;
                hits[slot] = r;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// K: reads (mz_sshash_k2u_reads_device).  One tile = kTileQ consecutive base positions of the
// concatenated read buffer.  The window is staged in LDS as three bit planes (low code bit, high
// code bit, invalid) built with wavefront ballots; every w-mer hash of the tile is computed ONCE
// into LDS (forward and reverse-complement strand) and each position takes the minimum over its
// k-w+1 window instead of re-hashing -- the sliding-window minimizer.  The lookups then run
// through the same staged tile machinery as the batch kernel.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread_bits(uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__device__ __forceinline__ uint32_t plane_window(const uint64_t *plane, uint32_t t, uint32_t nbits) {
    uint32_t wi = t >> 6, sh = t & 63;
    uint64_t x = plane[wi] >> sh;
    if (sh) x |= plane[wi + 1] << (64 - sh);
    return (uint32_t)(x & ((1ull << nbits) - 1));
}
__device__ __forceinline__ uint32_t base_code_dev(uint8_t c) {
    c &= 0xDF; // fold case
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

constexpr int kPlaneWords = kTileQ / 64 + 2; // plane words incl. the k-1 tail and a pad word

constexpr int kReadsStaged = 64; // reads whose offsets are staged per tile (a tile that more reads touch takes the slow walk)
template <typename H> // H = uint32_t when the w-mer hash has 32 significant bits (w <= 16), else uint64_t
struct alignas(16) ReadsLds {
    uint64_t plane_lo[kPlaneWords], plane_hi[kPlaneWords], plane_inv[kPlaneWords];
    uint64_t packed[kTileQ / 32 + 4];            // the tile's bases 2-bit packed (base i at bits [2i, 2i+1] of word i / 32), built from the planes
    H hf[kTileQ + 32], hr[kTileQ + 32];          // w-mer hashes of the tile, forward / reverse-complement strand
    uint64_t roff[kReadsStaged + 1];             // read_offsets[read0 + j]
    uint64_t koff[kReadsStaged + 1];             // kmer_offsets[read0 + j]
    uint64_t start_plane[kTileQ / 64];           // bit i: one of the staged reads 1.. starts at tile position i
    uint32_t start_cum[kTileQ / 64];             // read starts in the plane words before this one
    uint64_t read0;                              // read holding the first base of the tile
    uint32_t slow;                               // != 0: more reads touch the tile than are staged (or a read is empty): walk read_offsets in global memory
};
// w- or k-mer starting at tile position i from the packed tile
template <typename H>
__device__ __forceinline__ uint64_t tile_mer(const ReadsLds<H> &R, uint32_t i, uint32_t len) {
    const uint32_t wi = i >> 5, sh = 2 * (i & 31);
    const uint64_t lo = R.packed[wi], hi = R.packed[wi + 1];
    return ((lo >> sh) | ((hi << 1) << (63 - sh))) & kmer_mask(len);
}

// canonical minimizer of the window at tile position i from the staged w-mer hashes (== canonical_minimizer())
template <typename H>
__device__ __forceinline__ Minimizer window_minimizer(const ReadsLds<H> &R, uint32_t i, uint64_t fw, uint64_t rc, uint32_t k, uint32_t w) {
    const uint32_t n_win = k - w + 1;
    Minimizer mn;
    if (fw <= rc) { // g* = fw: leftmost minimum of the forward hashes
        uint32_t bo = 0;
        H bh = R.hf[i];
        for (uint32_t o = 1; o < n_win; ++o) {
            H h = R.hf[i + o];
            if (h < bh) { bh = h; bo = o; }
        }
        mn.offset = bo;
        mn.word = (fw >> (2 * bo)) & kmer_mask(w);
    } else { // g* = rc: offset o of g* is forward position i + (k-w-o); leftmost o == rightmost j
        uint32_t bj = 0;
        H bh = R.hr[i];
        for (uint32_t j = 1; j < n_win; ++j) {
            H h = R.hr[i + j];
            if (h <= bh) { bh = h; bj = j; }
        }
        mn.offset = (k - w) - bj;
        mn.word = (rc >> (2 * mn.offset)) & kmer_mask(w);
    }
    return mn;
}

// Sliding minimum for the compile-time (k, w) = (31, 15) configuration.  The staged values are single 32-bit keys,
// hash << 11 | tile position on the forward strand and hash << 11 | (2047 - tile position) on the reverse strand, so that
// an unsigned minimum is the smallest hash at its leftmost position in the orientation of the canonical k-mer
// (canonical_minimizer()'s tie rule).  Thread t takes the four windows that start at positions 4t .. 4t+3: their 17-key
// spans share keys 4t+3 .. 4t+16, whose minimum is computed once; 20 keys come in as five 16-byte LDS reads.
constexpr uint32_t kTilePosBits = 11;
static_assert(kTileQ + 32 <= (1u << kTilePosBits) && kMmerHash32Bits + kTilePosBits <= 32, "tile positions must fit the key");
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); } // v_min3_u32
__device__ __forceinline__ void window_minima_17(const uint32_t *keys, uint32_t t, uint32_t (&out)[4]) {
    uint32_t x[20];
    const uint4 *v = reinterpret_cast<const uint4 *>(keys + 4 * t);
#pragma unroll
    for (int a = 0; a < 5; ++a) { const uint4 q = v[a]; x[4 * a] = q.x; x[4 * a + 1] = q.y; x[4 * a + 2] = q.z; x[4 * a + 3] = q.w; }
    uint32_t mid = min3u(x[3], x[4], x[5]); // keys 3 .. 16, shared by the four windows
    mid = min3u(mid, x[6], x[7]);
    mid = min3u(mid, x[8], x[9]);
    mid = min3u(mid, x[10], x[11]);
    mid = min3u(mid, x[12], x[13]);
    mid = min3u(mid, x[14], x[15]);
    mid = min(mid, x[16]);
    out[0] = min(min3u(x[0], x[1], x[2]), mid);
    out[1] = min(min3u(x[1], x[2], x[17]), mid);
    out[2] = min(min3u(x[2], x[17], x[18]), mid);
    out[3] = min(min3u(x[17], x[18], x[19]), mid);
}

// The same for windows of N keys (N <= 17): thread t takes the windows 4t .. 4t+3, keys 3 .. N-1 of its span are shared.
template <int N>
__device__ __forceinline__ void window_minima_n(const uint32_t *keys, uint32_t t, uint32_t (&out)[4]) {
    static_assert(N >= 4 && N <= 17, "window length");
    constexpr int kVec = (N + 3 + 3) / 4;
    uint32_t x[4 * kVec];
    const uint4 *v = reinterpret_cast<const uint4 *>(keys + 4 * t);
#pragma unroll
    for (int a = 0; a < kVec; ++a) { const uint4 q = v[a]; x[4 * a] = q.x; x[4 * a + 1] = q.y; x[4 * a + 2] = q.z; x[4 * a + 3] = q.w; }
    uint32_t mid = x[3];
#pragma unroll
    for (int a = 4; a < N; ++a) mid = min(mid, x[a]);
    out[0] = min(min3u(x[0], x[1], x[2]), mid);
    out[1] = min(min3u(x[1], x[2], x[N]), mid);
    out[2] = min(min3u(x[2], x[N], x[N + 1]), mid);
    out[3] = min(min3u(x[N], x[N + 1], x[N + 2]), mid);
}
// Minimizers of more than 16 bases order their w-mers by a 64-bit hash, which no 32-bit key can carry.  The keys then hold the TOP 20 bits
// of the hash (hash20 << 12 | position, bit 11 clear): the window minimum over them is the true minimizer unless a second w-mer of the
// window shows the same 20 bits as the smallest one -- then the lower 44 bits decide, and the window is flagged (bit 11 of the result)
// so that its caller recomputes the minimizer exactly.  A tie shows when the minimum taken with the positions mirrored (the same
// smallest hash20 at its OTHER end) lands on a different position.
constexpr uint32_t kWideHashBits = 20, kWideTieBit = 1u << kTilePosBits;
template <int N>
__device__ __forceinline__ void window_minima_tie(const uint32_t *keys, uint32_t t, uint32_t (&out)[4]) {
    static_assert(N >= 4 && N <= 17, "window length");
    constexpr int kVec = (N + 3 + 3) / 4;
    constexpr uint32_t kPosMask = (1u << kTilePosBits) - 1;
    uint32_t x[4 * kVec];
    const uint4 *v = reinterpret_cast<const uint4 *>(keys + 4 * t);
#pragma unroll
    for (int a = 0; a < kVec; ++a) { const uint4 q = v[a]; x[4 * a] = q.x; x[4 * a + 1] = q.y; x[4 * a + 2] = q.z; x[4 * a + 3] = q.w; }
    uint32_t mid = x[3], mid_o = x[3] ^ kPosMask;
#pragma unroll
    for (int a = 4; a < N; ++a) { mid = min(mid, x[a]); mid_o = min(mid_o, x[a] ^ kPosMask); }
    auto one = [&](uint32_t a, uint32_t b, uint32_t c) {
        const uint32_t lo = min(min3u(a, b, c), mid);
        const uint32_t op = min(min3u(a ^ kPosMask, b ^ kPosMask, c ^ kPosMask), mid_o);
        return lo | ((op ^ kPosMask) != lo ? kWideTieBit : 0u);
    };
    out[0] = one(x[0], x[1], x[2]);
    out[1] = one(x[1], x[2], x[N]);
    out[2] = one(x[2], x[N], x[N + 1]);
    out[3] = one(x[N], x[N + 1], x[N + 2]);
}

// Every window is looked up through the tile machinery (the loop of bin/kphf/main.rs:314-324); what consecutive windows
// share -- their minimizer, for ~9 positions -- is resolved once per run (tile_resolve<SC, true>).
// The `.as_streaming()` shortcut of the reference (index/caching.rs:65-103: try prev.pos+1 on the same
// unitig first) was also built here, twice, as "look up one anchor window per 32, extend the rest along the
// anchor's unitig in both directions, fall back to the lookup": it returned identical results but ran
// SLOWER than looking every window up (11.0 vs 12.3 G k-mers/s on clean reads over long unitigs, 6.2 vs
// 8.9 on the junction-heavy s-txome set at the time): every window's k-mer has to be compared with the sequence
// either way, and a window that fails the extension (an error, a junction: a third of the windows of the
// benchmark's reads) needs the full lookup in a second staged pass.  Streaming requests therefore take this
// same path; by the reference's own contract (and for every unique-k-mer set) the results are the same.
template <uint32_t KW>
__global__ __launch_bounds__(kBlock, MZ_READS_WAVES) void k2u_reads_kernel(IndexView ix, const uint8_t *__restrict__ bases, uint64_t n_bases,
                                                           const uint64_t *__restrict__ read_offsets,
                                                           const uint64_t *__restrict__ kmer_offsets, uint64_t n_reads,
                                                           mz_k2upos *__restrict__ out, mz_counts *counts) {
    fix_kw<KW>(ix);
    using H = typename std::conditional<(KW != 0 && (KW & 255u) <= 16u), uint32_t, uint64_t>::type;
    // bases and w-mer hashes are staged by the workgroup (1024 positions + the k-1 tail); the lookups of each 256-position
    // quarter then run wave-scoped, like k2u_wave_kernel: no workgroup barrier inside tile_resolve
    using SC = WaveScope;
    __shared__ TileLdsT<SC::NT> SS[kBlock / 64];
    __shared__ ReadsLds<H> R;
    const uint32_t k = ix.k, w = ix.w, n_win = k - w + 1;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    TileLdsT<SC::NT> &S = SS[wave];
    const uint64_t n_tiles = (n_bases + kTileQ - 1) / kTileQ;
    // every workgroup takes a contiguous run of tiles, so the read that holds a tile's first base follows from the previous tile
    const uint64_t tiles_per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const uint64_t tile_begin_wg = (uint64_t)blockIdx.x * tiles_per_wg, tile_end_wg = tile_begin_wg + tiles_per_wg < n_tiles ? tile_begin_wg + tiles_per_wg : n_tiles;
    uint64_t next_read0 = ~0ull; // wave 0: read0 of the next tile when it is known, else ~0 (search)
    Tally tally;
    // the bases of a tile: kQpt x 256 positions + the k-1 tail, read by wave 1.  (Requesting them one tile ahead, so that the
    // loads are in flight while the previous tile is resolved, holds five more registers across the lookups: the kernel then
    // spills, and runs 2.6 % slower than with the loads at the top of the tile.)
    uint32_t code[kQpt], tail_code = (uint32_t)'N';
    auto request_bases = [&](uint64_t tile_) {
        const uint64_t B = tile_ * kTileQ;
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint64_t b = B + (uint64_t)q * kBlock + t;
            code[q] = b < n_bases ? bases[b] : (uint32_t)'N';
        }
        const uint64_t bt = B + kTileQ + lane;
        tail_code = (wave == 1 && lane < k - 1 && bt < n_bases) ? bases[bt] : (uint32_t)'N';
    };
    for (uint64_t tile = tile_begin_wg; tile < tile_end_wg; ++tile) {
        const uint64_t B0 = tile * kTileQ;
        tile_begin<SC>(S);
        request_bases(tile);
        // stage the bit planes: one ballot triple per wave and slice of 256 positions (wave 0 looks the reads up first, while
        // the bases are in flight)
        if (wave == 0) {
            // read holding the first base of the tile: largest r with read_offsets[r] <= B0.  Carried over from the previous tile,
            // else a 64-ary search (every lane probes one offset per round)
            uint64_t r0 = next_read0;
            if (r0 == ~0ull) {
                uint64_t lo = 0, hi = n_reads; // invariant: read_offsets[lo] <= B0 < read_offsets[hi] (offsets[n_reads] = n_bases > B0)
                while (hi - lo > 1) {
                    const uint64_t step = (hi - lo + 63) >> 6, idx = lo + (uint64_t)lane * step;
                    const bool le = idx < hi && read_offsets[idx] <= B0;
                    const uint32_t c = (uint32_t)__popcll(__ballot(le)); // probes are ascending: the first c of them are <= B0 (c >= 1: lane 0 probes lo)
                    lo = lo + (uint64_t)(c - 1) * step;
                    hi = lo + step < hi ? lo + step : hi;
                }
                r0 = lo;
            }
            // offsets of the reads r0 .. r0 + 64 (beyond the last read: +infinity) and their first output slots
            const uint64_t j0 = r0 + lane, j1 = r0 + kReadsStaged;
            const uint64_t o0 = j0 <= n_reads ? read_offsets[j0] : ~0ull, o1 = j1 <= n_reads ? read_offsets[j1] : ~0ull;
            const uint64_t k0 = j0 < n_reads ? kmer_offsets[j0] : 0ull;
            R.roff[lane] = o0;
            R.koff[lane] = k0;
            if (lane == 0) { R.roff[kReadsStaged] = o1; R.koff[kReadsStaged] = 0; R.read0 = r0; }
            if (lane < kTileQ / 64) R.start_plane[lane] = 0;
            const uint64_t o_next = (uint64_t)__shfl((unsigned long long)o0, (int)((lane + 1) & 63)); // offset of read r0 + lane + 1
            const uint64_t nxt = lane == 63 ? o1 : o_next;
            // slow walk when a 65th read could start inside the tile (+ its k-1 tail) or a staged read that matters is empty
            const bool empty_read = o0 != ~0ull && nxt == o0 && o0 < B0 + kTileQ + k;
            const bool too_many = o1 != ~0ull && o1 < B0 + kTileQ + k;
            const bool slow = __ballot(empty_read) != 0 || too_many;
            if (lane == 0) R.slow = slow ? 1u : 0u;
            const bool starts_here = lane >= 1 && o0 > B0 && o0 < B0 + kTileQ; // read r0 + lane starts at tile position o0 - B0 (> 0)
            if (starts_here) atomicOr((unsigned long long *)&R.start_plane[(o0 - B0) >> 6], 1ull << ((o0 - B0) & 63));
            // next tile: reads that start at or before its first base
            const uint32_t adv = (uint32_t)__popcll(__ballot(lane >= 1 && o0 <= B0 + kTileQ));
            next_read0 = slow || adv == 63 ? ~0ull : r0 + adv;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < kTileQ / 64) { // exclusive prefix counts of the start plane
                const uint32_t c = (uint32_t)__popcll(R.start_plane[lane]);
                R.start_cum[lane] = wave_scan<false>(c) - c;
            } else {
                (void)wave_scan<false>(0u);
            }
        }
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint32_t c = base_code_dev((uint8_t)code[q]);
            unsigned long long blo = __ballot(c & 1u), bhi = __ballot((c >> 1) & 1u), binv = __ballot(c >> 2);
            if (lane == 0) { uint32_t wi = q * (kBlock / 64) + wave; R.plane_lo[wi] = blo; R.plane_hi[wi] = bhi; R.plane_inv[wi] = binv; }
        }
        if (wave == 1) {
            const uint32_t ct = base_code_dev((uint8_t)tail_code);
            unsigned long long tlo = __ballot(ct & 1u), thi = __ballot((ct >> 1) & 1u), tinv = __ballot(ct >> 2);
            if (lane == 0) {
                R.plane_lo[kTileQ / 64] = tlo; R.plane_hi[kTileQ / 64] = thi; R.plane_inv[kTileQ / 64] = tinv;
                R.plane_lo[kTileQ / 64 + 1] = 0; R.plane_hi[kTileQ / 64 + 1] = 0; R.plane_inv[kTileQ / 64 + 1] = ~0ull;
            }
        }
        __syncthreads();
        // the tile 2-bit packed: word j = bases [32 j, 32 j + 32) from the low / high code bit planes
        if (t < kTileQ / 32 + 4) {
            const uint32_t half = 32 * (t & 1);
            const uint64_t pl = t / 2 < kPlaneWords ? R.plane_lo[t / 2] : 0, ph = t / 2 < kPlaneWords ? R.plane_hi[t / 2] : 0;
            R.packed[t] = spread_bits((uint32_t)(pl >> half)) | (spread_bits((uint32_t)(ph >> half)) << 1);
        }
        __syncthreads();
        // w-mer hashes of the tile, both strands
        constexpr bool kPackedKeys = KW == kKW3115; // single 32-bit keys and the shared-span sliding minimum (window_minima_17)
        for (uint32_t j = t; j < kTileQ + n_win - 1; j += kBlock) {
            const uint64_t m = tile_mer(R, j, w);
            if constexpr (kPackedKeys) {
                R.hf[j] = (H)((hash_mmer32((uint32_t)m, ix.seed) << kTilePosBits) | j);
                R.hr[j] = (H)((hash_mmer32((uint32_t)kmer_rc(m, w), ix.seed) << kTilePosBits) | ((1u << kTilePosBits) - 1 - j));
            } else {
                R.hf[j] = (H)hash_mmer(m, ix.seed, w);
                R.hr[j] = (H)hash_mmer(kmer_rc(m, w), ix.seed, w);
            }
        }
        __syncthreads();
        if constexpr (kPackedKeys) { // window minima in place: every key is read before any is overwritten
            uint32_t wf[4], wr[4];
            window_minima_17(reinterpret_cast<const uint32_t *>(R.hf), t, wf);
            window_minima_17(reinterpret_cast<const uint32_t *>(R.hr), t, wr);
            __syncthreads();
            *reinterpret_cast<uint4 *>(&R.hf[4 * t]) = uint4{wf[0], wf[1], wf[2], wf[3]};
            *reinterpret_cast<uint4 *>(&R.hr[4 * t]) = uint4{wr[0], wr[1], wr[2], wr[3]};
            __syncthreads();
        }
        // S0: one item per base position
        uint64_t out_idx[kQpt];
        uint32_t state[kQpt]; // 0 = no window here, 1 = window with a non-ACGT byte, 2 = valid k-mer (answer pending), 3 = valid k-mer whose minimizer the 20-bit keys could not decide: straight to the exact walk
        const bool slow = R.slow != 0;
        uint64_t r = R.read0;
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint32_t it = q * SC::NT + lane;               // item inside the wave's tile
            const uint32_t i = wave * (SC::NT * kQpt) + it;      // position inside the workgroup's 1024
            const uint64_t b = B0 + i;
            state[q] = 0;
            out_idx[q] = 0;
            uint64_t fw = 0, rc = 0;
            Minimizer mn{0, 0};
            if (b < n_bases) {
                uint64_t rs, re, ko;
                if (!slow) { // the read of position i: staged reads that start at or before it (from the start plane)
                    const uint32_t wi = i >> 6;
                    const uint32_t rl = R.start_cum[wi] + (uint32_t)__popcll(R.start_plane[wi] & (~0ull >> (63 - (i & 63))));
                    rs = R.roff[rl]; re = R.roff[rl + 1]; ko = R.koff[rl];
                } else {
                    while (r + 1 < n_reads && read_offsets[r + 1] <= b) ++r;
                    rs = read_offsets[r]; re = read_offsets[r + 1]; ko = kmer_offsets[r];
                }
                if (b >= rs && b + k <= re) {
                    out_idx[q] = ko + (b - rs);
                    state[q] = plane_window(R.plane_inv, i, k) == 0 ? 2u : 1u;
                }
            }
            bool query = state[q] == 2;
            if (query) {
                fw = tile_mer(R, i, k);
                rc = kmer_rc(fw, k);
                if constexpr (kPackedKeys) { // minimum key of the window -> offset of the minimizer inside g*
                    const uint32_t key = fw <= rc ? (uint32_t)R.hf[i] : (uint32_t)R.hr[i], low = key & ((1u << kTilePosBits) - 1);
                    mn.offset = fw <= rc ? low - i : (k - w) - ((1u << kTilePosBits) - 1 - low - i);
                    mn.word = ((fw <= rc ? fw : rc) >> (2 * mn.offset)) & kmer_mask(w);
                } else {
                    mn = window_minimizer(R, i, fw, rc, k, w);
                }
            }
            tile_set_query<SC>(S, it, fw, rc, mn, ix, query, false); // (runs are resolved per leader: no per-item line pattern)
            if (state[q] == 1) { // non-ACGT window: reported, never queried
                mz_k2upos sk = make_none();
                sk.match = MZ_SKIPPED;
                out[out_idx[q]] = sk;
            }
            tally.add(false, false, state[q] == 1);
        }
        tile_resolve<SC, true>(S, ix);
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const bool pend = state[q] == 2;
            mz_k2upos res = pend ? tile_result<SC>(S, ix, q * SC::NT + lane) : make_none();
            if (pend) out[out_idx[q]] = res;
            tally.add(pend && res.match != MZ_NO_MATCH, pend && res.match == MZ_NO_MATCH, false);
        }
        __syncthreads();
    }
    tally.flush(counts);
}

// ------------------------------------------------------------------------------------
// K: reads, wave-scoped (the (31, 15) configuration).  Same stages as k2u_reads_kernel, but every wavefront stages its
// OWN 256 positions (+ the 30-base tail its last windows need) in its own LDS slice and walks front end and lookups
// alone: no workgroup barrier anywhere, so a wave never waits for the slowest of four, and the read bookkeeping that one
// wave did for the whole workgroup is done by each wave for its tile.  The tail is staged twice (by this wave and as the
// head of the next tile): 18 % more front-end work for no synchronisation.
// ------------------------------------------------------------------------------------
constexpr int kWaveTile = 64 * kQpt;   // positions per wave tile
constexpr int kWaveReadsStaged = 32;   // reads whose offsets are staged per wave tile
struct alignas(16) ReadsWaveLds {
    uint64_t plane_lo[kQpt + 2], plane_hi[kQpt + 2], plane_inv[kQpt + 2]; // kQpt rows of 64 positions, the tail row, a pad word
    uint64_t aligned[kAlignWords];                                          // pad words, the 2-bit packed bases of the tile and its tail (packed()), their reverse complement (tile_finish_grouped)
    __device__ __forceinline__ uint64_t *packed() { return aligned + kAlignPad; }
    uint32_t keys_f[kWaveTile + 32], keys_r[kWaveTile + 32];                // window keys (see window_minima_17), then the window minima
    uint64_t roff[kWaveReadsStaged + 1], koff[kWaveReadsStaged + 1];
    uint64_t start_plane[kQpt];
    uint32_t start_cum[kQpt];
    uint64_t read0;
    uint32_t slow;
};
__device__ __forceinline__ uint64_t packed_mer(const uint64_t *packed, uint32_t i, uint32_t len) {
    const uint32_t wi = i >> 5, sh = 2 * (i & 31);
    const uint64_t lo = packed[wi], hi = packed[wi + 1];
    return ((lo >> sh) | ((hi << 1) << (63 - sh))) & kmer_mask(len);
}
// kPerItemLines: true: every window scans its minimizer line itself (stage SF), as a point query does; false: one lookup per run of windows
// kGrouped: candidates are verified per group of windows that share a sequence record (tile_finish_grouped) instead of one by one
// KW: the compile-time (k, w): (31, 15), or (31, 17 / 19 / 21) with minimizer lines addressed by a hash of the word (per-item form only)
template <uint32_t KW, bool kPerItemLines, bool kGrouped>
__global__ __launch_bounds__(kBlock, MZ_READS_WAVES) void k2u_reads_wave_kernel(IndexView ix, const uint8_t *__restrict__ bases, uint64_t n_bases,
                                                                              const uint64_t *__restrict__ read_offsets,
                                                                              const uint64_t *__restrict__ kmer_offsets, uint64_t n_reads,
                                                                              mz_k2upos *__restrict__ out, mz_counts *counts) {
    fix_kw<KW>(ix);
    using SC = WaveScope;
    __shared__ TileLdsT<SC::NT> SS[kBlock / 64];
    __shared__ ReadsWaveLds RR[kBlock / 64];
    __shared__ DeferLds DD[kBlock / 64];
    constexpr uint32_t k = KW >> 8, w = KW & 255u, n_win = k - w + 1, T = kWaveTile;
    constexpr bool kWide = w > 16; // 64-bit w-mer hashes: 20-bit keys with a tie flag (window_minima_tie)
    static_assert(k == 31 && w <= k && n_win >= 4 && n_win <= 17, "tail of k - 1 bases, windows of at most 17 keys");
    static_assert(!kGrouped || KW == kKW3115, "grouped finish: (31, 15) only");
    static_assert(!kWide || kPerItemLines, "w > 16: through the hash-addressed lines only");
    static_assert(T + n_win - 1 <= kWaveTile + 32 - 12 && 4 * 63 + n_win + 2 < kWaveTile + 32, "keys of the tile and its tail");
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TileLdsT<SC::NT> &S = SS[wave];
    ReadsWaveLds &R = RR[wave];
    const uint64_t n_tiles = (n_bases + T - 1) / T, n_waves = (uint64_t)gridDim.x * (kBlock / 64), wave_id = (uint64_t)blockIdx.x * (kBlock / 64) + wave;
    // every wave takes a contiguous run of tiles, so the read that holds a tile's first base follows from the previous tile
    const uint64_t tiles_per_wave = (n_tiles + n_waves - 1) / n_waves;
    const uint64_t tile_first = wave_id * tiles_per_wave, tile_end = tile_first + tiles_per_wave < n_tiles ? tile_first + tiles_per_wave : n_tiles;
    uint64_t next_read0 = ~0ull; // read0 of the next tile when it is known, else ~0 (search)
    Tally tally;
    DeferLds &D = DD[wave];
    uint32_t n_defer = 0; // queued exact walks (wave-uniform)
    auto flush = [&]() { defer_flush(D, n_defer, ix, tally, [&](uint64_t where, const mz_k2upos &r) { out[where] = r; }); };
#ifndef MZ_READS_WAVE_PREFETCH
#define MZ_READS_WAVE_PREFETCH 1 // the bases of tile + 1 are requested while tile is resolved (five registers held across the lookups: the kernel then sits exactly at its 168-register limit, without scratch; +6.5 %)
#endif
    uint32_t code[kQpt], tail_code = (uint32_t)'N';
    auto request_bases = [&](uint64_t tile_) {
        const uint64_t B = tile_ * T;
#pragma unroll
        for (int q = 0; q < kQpt; ++q) { // the loads first: the reads are looked up while they are in flight
            const uint64_t b = B + (uint64_t)q * 64 + lane;
            code[q] = b < n_bases ? bases[b] : (uint32_t)'N';
        }
        const uint64_t bt = B + T + lane;
        tail_code = (lane < k - 1 && bt < n_bases) ? bases[bt] : (uint32_t)'N';
    };
    for (uint64_t tile = tile_first; tile < tile_end; ++tile) {
        const uint64_t B0 = tile * T;
        tile_begin<SC>(S);
#if MZ_READS_WAVE_PREFETCH
        if (tile == tile_first) request_bases(tile);
#else
        request_bases(tile);
#endif
        {
            // read holding the first base of the tile: largest r with read_offsets[r] <= B0.  Carried over from the previous tile,
            // else a 64-ary search (every lane probes one offset per round)
            uint64_t r0 = next_read0;
            if (r0 == ~0ull) {
                uint64_t lo = 0, hi = n_reads; // invariant: read_offsets[lo] <= B0 < read_offsets[hi] (offsets[n_reads] = n_bases > B0)
                while (hi - lo > 1) {
                    const uint64_t step = (hi - lo + 63) >> 6, idx = lo + (uint64_t)lane * step;
                    const bool le = idx < hi && read_offsets[idx] <= B0;
                    const uint32_t c = (uint32_t)__popcll(__ballot(le)); // probes are ascending: the first c of them are <= B0 (c >= 1: lane 0 probes lo)
                    lo = lo + (uint64_t)(c - 1) * step;
                    hi = lo + step < hi ? lo + step : hi;
                }
                r0 = lo;
            }
            // offsets of the reads r0 .. r0 + 32 (beyond the last read: +infinity) and their first output slots, one per lane
            const bool in = lane < kWaveReadsStaged;
            const uint64_t j0 = r0 + lane, j1 = r0 + kWaveReadsStaged;
            const uint64_t o0 = in && j0 <= n_reads ? read_offsets[j0] : ~0ull, o1 = j1 <= n_reads ? read_offsets[j1] : ~0ull;
            const uint64_t k0 = in && j0 < n_reads ? kmer_offsets[j0] : 0ull;
            if (in) { R.roff[lane] = o0; R.koff[lane] = k0; }
            if (lane == 0) { R.roff[kWaveReadsStaged] = o1; R.koff[kWaveReadsStaged] = 0; R.read0 = r0; }
            if (lane < kQpt) R.start_plane[lane] = 0;
            const uint64_t o_next = (uint64_t)__shfl((unsigned long long)o0, (int)((lane + 1) & 63)); // offset of read r0 + lane + 1
            const uint64_t nxt = lane == kWaveReadsStaged - 1 ? o1 : o_next;
            // slow walk when a 33rd read could start inside the tile (+ its k-1 tail) or a staged read that matters is empty
            const bool empty_read = in && o0 != ~0ull && nxt == o0 && o0 < B0 + T + k;
            const bool too_many = o1 != ~0ull && o1 < B0 + T + k;
            const bool slow = __ballot(empty_read) != 0 || too_many;
            if (lane == 0) R.slow = slow ? 1u : 0u;
            SC::sync(); // start_plane is cleared before bits are set
            const bool starts_here = in && lane >= 1 && o0 > B0 && o0 < B0 + T; // read r0 + lane starts at tile position o0 - B0 (> 0)
            if (starts_here) atomicOr((unsigned long long *)&R.start_plane[(o0 - B0) >> 6], 1ull << ((o0 - B0) & 63));
            // next tile: reads that start at or before its first base
            const uint32_t adv = (uint32_t)__popcll(__ballot(in && lane >= 1 && o0 <= B0 + T));
            next_read0 = slow || adv == kWaveReadsStaged - 1 ? ~0ull : r0 + adv;
            SC::sync();
            const uint32_t c = lane < kQpt ? (uint32_t)__popcll(R.start_plane[lane]) : 0u; // exclusive prefix counts of the start plane
            const uint32_t cum = wave_scan<false>(c) - c;
            if (lane < kQpt) R.start_cum[lane] = cum;
        }
        // bit planes: one ballot triple per row of 64 positions, the tail row, a pad word
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            const uint32_t c = base_code_dev((uint8_t)code[q]);
            const unsigned long long blo = __ballot(c & 1u), bhi = __ballot((c >> 1) & 1u), binv = __ballot(c >> 2);
            if (lane == 0) { R.plane_lo[q] = blo; R.plane_hi[q] = bhi; R.plane_inv[q] = binv; }
        }
        {
            const uint32_t c = base_code_dev((uint8_t)tail_code);
            const unsigned long long blo = __ballot(c & 1u), bhi = __ballot((c >> 1) & 1u), binv = __ballot(c >> 2);
            if (lane == 0) {
                R.plane_lo[kQpt] = blo; R.plane_hi[kQpt] = bhi; R.plane_inv[kQpt] = binv;
                R.plane_lo[kQpt + 1] = 0; R.plane_hi[kQpt + 1] = 0; R.plane_inv[kQpt + 1] = ~0ull;
            }
        }
        SC::sync();
#if MZ_READS_WAVE_PREFETCH
        if (tile + 1 < tile_end) request_bases(tile + 1);
#endif
        if (lane < T / 32 + 4) { // the tile 2-bit packed: word j = bases [32 j, 32 j + 32)
            const uint32_t half = 32 * (lane & 1);
            const uint64_t pl = lane / 2 < kQpt + 2 ? R.plane_lo[lane / 2] : 0, ph = lane / 2 < kQpt + 2 ? R.plane_hi[lane / 2] : 0;
            R.packed()[lane] = spread_bits((uint32_t)(pl >> half)) | (spread_bits((uint32_t)(ph >> half)) << 1);
        }
        SC::sync();
        if constexpr (kGrouped) { // the tile reverse-complemented, word by word in reverse order (for the windows of reads that run against the stored strand)
            if (lane < kAlignTileWords) R.aligned[kAlignRcAt + lane] = kmer_rc(R.packed()[kAlignTileWords - 1 - lane], 32);
        }
        for (uint32_t j = lane; j < T + n_win - 1; j += 64) { // window keys of both strands (layout: window_minima_17)
            if constexpr (kWide) {
                const uint64_t m = packed_mer(R.packed(), j, w);
                R.keys_f[j] = ((uint32_t)(hash_mmer64(m, ix.seed) >> (64 - kWideHashBits)) << (kTilePosBits + 1)) | j;
                R.keys_r[j] = ((uint32_t)(hash_mmer64(kmer_rc(m, w), ix.seed) >> (64 - kWideHashBits)) << (kTilePosBits + 1)) | ((1u << kTilePosBits) - 1 - j);
            } else {
                const uint32_t m = (uint32_t)packed_mer(R.packed(), j, w);
                R.keys_f[j] = (hash_mmer32(m, ix.seed) << kTilePosBits) | j;
                R.keys_r[j] = (hash_mmer32((uint32_t)kmer_rc(m, w), ix.seed) << kTilePosBits) | ((1u << kTilePosBits) - 1 - j);
            }
        }
        SC::sync();
        {
            uint32_t wf[4], wr[4]; // window minima in place: every key is read before any is overwritten
            if constexpr (kWide) {
                window_minima_tie<(int)n_win>(R.keys_f, lane, wf);
                window_minima_tie<(int)n_win>(R.keys_r, lane, wr);
            } else if constexpr (n_win == 17) {
                window_minima_17(R.keys_f, lane, wf);
                window_minima_17(R.keys_r, lane, wr);
            } else {
                window_minima_n<(int)n_win>(R.keys_f, lane, wf);
                window_minima_n<(int)n_win>(R.keys_r, lane, wr);
            }
            SC::sync();
            *reinterpret_cast<uint4 *>(&R.keys_f[4 * lane]) = uint4{wf[0], wf[1], wf[2], wf[3]};
            *reinterpret_cast<uint4 *>(&R.keys_r[4 * lane]) = uint4{wr[0], wr[1], wr[2], wr[3]};
            SC::sync();
        }
        // S0: one item per base position
        uint64_t out_idx[kQpt];
        uint32_t state[kQpt]; // 0 = no window here, 1 = window with a non-ACGT byte, 2 = valid k-mer (answer pending)
        const bool slow = R.slow != 0;
        uint64_t r = R.read0;
        auto s0_row = [&](int q) {
            const uint32_t i = q * 64 + lane; // item == position inside the tile
            const uint64_t b = B0 + i;
            state[q] = 0;
            out_idx[q] = 0;
            uint64_t fw = 0, rc = 0;
            Minimizer mn{0, 0};
            if (b < n_bases) {
                uint64_t rs, re, ko;
                if (!slow) { // the read of position i: staged reads that start at or before it (from the start plane)
                    const uint32_t rl = R.start_cum[q] + (uint32_t)__popcll(R.start_plane[q] & (~0ull >> (63 - lane)));
                    rs = R.roff[rl]; re = R.roff[rl + 1]; ko = R.koff[rl];
                } else {
                    while (r + 1 < n_reads && read_offsets[r + 1] <= b) ++r;
                    rs = read_offsets[r]; re = read_offsets[r + 1]; ko = kmer_offsets[r];
                }
                if (b >= rs && b + k <= re) {
                    out_idx[q] = ko + (b - rs);
                    const uint64_t inv = (R.plane_inv[q] >> lane) | (lane ? R.plane_inv[q + 1] << (64 - lane) : 0ull); // invalid flags of positions i, i+1, ...
                    state[q] = (inv & ((1ull << k) - 1)) == 0 ? 2u : 1u;
                }
            }
            if (state[q] == 2) {
                fw = packed_mer(R.packed(), i, k);
                rc = kmer_rc(fw, k);
                const uint32_t key = fw <= rc ? R.keys_f[i] : R.keys_r[i], low = key & ((1u << kTilePosBits) - 1); // minimum key of the window
                mn.offset = fw <= rc ? low - i : (k - w) - ((1u << kTilePosBits) - 1 - low - i);
                mn.word = ((fw <= rc ? fw : rc) >> (2 * mn.offset)) & kmer_mask(w);
                if (kWide && (key & kWideTieBit)) state[q] = 3; // (lookup() of the deferred walk computes the minimizer from the full hashes)
            }
            const bool query = state[q] == 2;
            tile_set_query<SC>(S, i, fw, rc, mn, ix, query, kPerItemLines); // (runs are resolved per leader: no per-item line pattern)
            if (state[q] == 1) { // non-ACGT window: reported, never queried
                mz_k2upos sk = make_none();
                sk.match = MZ_SKIPPED;
                out[out_idx[q]] = sk;
            }
            tally.add(false, false, state[q] == 1);
        };
#ifndef MZ_READS_ROW_S0
#define MZ_READS_ROW_S0 0 // 1: per-item form with the rows of the tile prepared inside stage SF, one batch of lines ahead, as the point kernel does (44 bytes of scratch per lane here: 21.4 against 23.5 G windows/s on S-human)
#endif
        if constexpr (kGrouped && kStopAfter == 0) {
#pragma unroll
            for (int q = 0; q < kQpt; ++q) s0_row(q);
            tile_resolve<SC, !kPerItemLines, NoRowS0, false, MZ_READS_SF_LANES>(S, ix);
            uint32_t hits = 0; // bit q: the window of row q hit
            auto emit = [&](int q, const mz_k2upos &r) {
                if (state[q] == 2) {
                    out[out_idx[q]] = r;
                    if (r.match != MZ_NO_MATCH) hits |= 1u << q;
                }
            };
            const uint32_t walk = tile_finish_grouped<SC>(S, ix, R.aligned, reinterpret_cast<uint64_t *>(R.keys_f), emit);
#pragma unroll
            for (int q = 0; q < kQpt; ++q) {
                const bool pend = state[q] == 2, defer = pend && ((walk >> q) & 1u); // (answered later, by the pass over the wave's deferred exact walks)
                tally.add(pend && !defer && ((hits >> q) & 1u), pend && !defer && !((hits >> q) & 1u), false);
                defer_push(D, n_defer, defer, out_idx[q], S.fw[q * 64 + lane], S.qinfo[q * 64 + lane], flush);
            }
        } else {
            if constexpr (kPerItemLines && MZ_READS_ROW_S0) {
                tile_resolve<SC, false, decltype(s0_row), true, 8, true>(S, ix, s0_row);
            } else {
#pragma unroll
                for (int q = 0; q < kQpt; ++q) s0_row(q);
                tile_resolve<SC, !kPerItemLines, NoRowS0, true, MZ_READS_SF_LANES, true>(S, ix);
            }
#pragma unroll
            for (int q = 0; q < kQpt; ++q) {
                const bool tie = kWide && state[q] == 3, pend = state[q] == 2 || tie;
                const mz_k2upos res = pend && !tie ? tile_result<SC>(S, ix, q * 64 + lane) : make_none();
                const bool defer = pend && (tie || res.match == MZ_DEFER);
                if (pend && !defer) out[out_idx[q]] = res;
                tally.add(pend && !defer && res.match != MZ_NO_MATCH, pend && !defer && res.match == MZ_NO_MATCH, false);
                defer_push(D, n_defer, defer, out_idx[q], S.fw[q * 64 + lane], S.qinfo[q * 64 + lane], flush);
            }
        }
#ifdef MZ_DEBUG_COUNT
        tally.skipped += S.dbg[MZ_DEBUG_COUNT - 1];
#endif
        SC::sync();
    }
    if (n_defer) flush();
    tally.flush(counts);
}

// ------------------------------------------------------------------------------------
// K: validate_self on the device (kphf/mod.rs:69-139).  One wave tile = 128 consecutive global base
// positions, each queried as km (items 0..127) and km.swap() (items 128..255) through the tile
// machinery; the expected (unitig, offset, length) comes from the plain `starts` array, which the
// lookup path never reads.
// ------------------------------------------------------------------------------------
template <uint32_t KW>
__global__ __launch_bounds__(kBlock, 4) void validate_self_kernel(IndexView ix, unsigned long long *report, unsigned long long *dup_list,
                                                                    uint64_t dup_cap) {
    fix_kw<KW>(ix);
    using SC = WaveScope; // every wavefront validates its own 128 positions (256 items), like k2u_wave_kernel
    __shared__ TileLdsT<SC::NT> SS[kBlock / 64];
    TileLdsT<SC::NT> &S = SS[threadIdx.x >> 6];
    constexpr int kPos = SC::NT * kQpt / 2, kPpt = kQpt / 2;
    const uint32_t t = SC::tid(), k = ix.k, w = ix.w;
    const uint64_t n_tiles = (ix.total_len + kPos - 1) / kPos;
    const uint64_t wave_id = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (kBlock / 64);
    unsigned n_k = 0, n_fail = 0, n_dup = 0;
    for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const uint64_t P0 = tile * kPos;
        tile_begin<SC>(S);
        uint64_t u = 0;
        if (t == 0) { // unitig holding the first position of the tile
            uint64_t lo = 0, hi = ix.n_unitigs;
            while (hi - lo > 1) {
                uint64_t mid = (lo + hi) >> 1;
                if (ix.starts[mid] <= P0) lo = mid; else hi = mid;
            }
            u = lo;
        }
        u = (uint64_t)__shfl((unsigned long long)u, 0);
        uint64_t ui[kPpt], us[kPpt], ue[kPpt];
        bool valid[kPpt];
#pragma unroll
        for (int q = 0; q < kPpt; ++q) {
            const uint32_t i = q * SC::NT + t;
            const uint64_t p = P0 + i;
            valid[q] = false;
            ui[q] = us[q] = ue[q] = 0;
            uint64_t fw = 0, rc = 0;
            if (p + k <= ix.total_len) {
                while (ix.starts[u + 1] <= p) ++u;
                ui[q] = u; us[q] = ix.starts[u]; ue[q] = ix.starts[u + 1];
                valid[q] = p + k <= ue[q];
                if (valid[q]) { fw = read_kmer_i(ix.seqbv, p, k); rc = kmer_rc(fw, k); }
            }
            Minimizer mn = canonical_minimizer(fw, rc, k, w, ix.seed);
            tile_set_query<SC>(S, i, fw, rc, mn, ix, valid[q], ix.front.lines != nullptr);          // km
            tile_set_query<SC>(S, kPos + i, rc, fw, mn, ix, valid[q], ix.front.lines != nullptr);   // km.swap(): same canonical k-mer, same minimizer
        }
        tile_resolve<SC>(S, ix);
#pragma unroll
        for (int q = 0; q < kPpt; ++q) {
            if (!valid[q]) continue;
            const uint32_t i = q * SC::NT + t;
            const uint64_t p = P0 + i;
            ++n_k;
            for (int o = 0; o < 2; ++o) {
                mz_k2upos r = tile_result<SC>(S, ix, o ? kPos + i : i);
                uint32_t want = o ? MZ_TWIN_MATCH : MZ_IDENTITY_MATCH;
                bool ok = r.match == want && r.unitig_id == (uint32_t)ui[q] && r.pos == (uint32_t)(p - us[q]) &&
                          r.unitig_len == (uint32_t)(ue[q] - us[q]);
                if (!ok) {
                    bool dup = false;
                    if (r.match != MZ_NO_MATCH && r.unitig_id < ix.n_unitigs) {
                        uint64_t at = ix.starts[r.unitig_id] + r.pos;
                        if (at + k <= ix.total_len && at != p) {
                            uint64_t kw = read_kmer_i(ix.seqbv, at, k), fw = S.fw[i];
                            dup = kw == fw || kw == kmer_rc(fw, k);
                        }
                    }
                    if (dup) {
                        ++n_dup;
                        if (dup_list && o == 0) { // listed once per k-mer: the position that is NOT the one the lookup answers with
                            const unsigned long long slot = atomicAdd(&report[4], 1ull);
                            if (slot < dup_cap) dup_list[slot] = p;
                        }
                    } else {
                        ++n_fail;
                        atomicMin(&report[3], (unsigned long long)p);
                    }
                }
            }
        }
        SC::sync();
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_k += __shfl_down(n_k, off);
        n_fail += __shfl_down(n_fail, off);
        n_dup += __shfl_down(n_dup, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_k) atomicAdd(&report[0], (unsigned long long)n_k);
        if (n_fail) atomicAdd(&report[1], (unsigned long long)n_fail);
        if (n_dup) atomicAdd(&report[2], (unsigned long long)n_dup);
    }
}

// what the minimizer lines hold, by slot kind (the layout's own print_stats, cf. sshash.rs:360-395): out[0] entries inline,
// [1] buckets behind a spill marker, [2] their entries, [3] heavy buckets, [4] their entries (a marker's count saturates at
// 65535), [5] empty slots, [6] hash-addressed lines the builder gave up (kFrontUnusable), [7] markers without a size (exact walk: a bucket
// too large for the field, or member words that share a fingerprint in a hash-addressed line)
constexpr int kLineStatWords = 8;
__global__ __launch_bounds__(kBlock) void line_stats_kernel(const uint64_t *__restrict__ lines, uint64_t n_slots, unsigned long long *out) {
    unsigned long long c[kLineStatWords] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_slots; i += stride) {
        const uint64_t e = lines[i];
        const uint32_t kind = (uint32_t)e & 3u, cnt = (uint32_t)(e >> kFrontFlankShift) & 0xFFFFu;
        c[0] += kind == kFrontEntry;
        const bool sized = cnt != 0xFFFFu;
        c[1] += kind == kFrontSpill && e != kFrontUnusable && sized; c[2] += kind == kFrontSpill && sized ? cnt : 0u;
        c[6] += e == kFrontUnusable;
        c[7] += kind == kFrontSpill && !sized;
        c[3] += kind == kFrontHeavy; c[4] += kind == kFrontHeavy ? cnt : 0u;
        c[5] += kind == kFrontEmptyKind;
    }
#pragma unroll
    for (int j = 0; j < kLineStatWords; ++j) {
        for (int off = 32; off > 0; off >>= 1) c[j] += __shfl_down(c[j], off);
        if ((threadIdx.x & 63) == 0 && c[j]) atomicAdd(&out[j], c[j]);
    }
}

// forward words at given global positions (positive-query synthesis, stays in HBM)
__global__ __launch_bounds__(kBlock) void extract_kmers_kernel(IndexView ix, const uint64_t *__restrict__ positions,
                                                               const uint8_t *__restrict__ flip, uint64_t n,
                                                               uint64_t *__restrict__ fw_words) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t p = positions[i];
        uint64_t x = p + ix.k <= ix.total_len ? read_kmer_i(ix.seqbv, p, ix.k) : 0;
        fw_words[i] = (flip && flip[i]) ? kmer_rc(x, ix.k) : x;
    }
}

// ------------------------------------------------------------------------------------
// host side: device replica + launchers
// ------------------------------------------------------------------------------------
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <class T>
static int upload(DeviceIndex *d, const T *src, uint64_t count, const T **dst) {
    uint64_t bytes = std::max<uint64_t>(count, 1) * sizeof(T);
    void *p = nullptr;
    MZ_HIP_CHECK(hipMalloc(&p, bytes));
    d->allocs.push_back(p);
    d->bytes += bytes;
    if (count) MZ_HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T *)p;
    return MZ_OK;
}

static int upload_mphf(DeviceIndex *d, const Mphf &m, MphfView *v) {
    int rc;
    *v = m.view();
    if ((rc = upload(d, m.blocks.data(), m.blocks.size(), &v->blocks)) != MZ_OK) return rc;
    if ((rc = upload(d, m.levels.data(), m.levels.size(), &v->levels)) != MZ_OK) return rc;
    return MZ_OK;
}

int device_upload(const SSHashHost &h, int device, DeviceIndex **out) {
    int n = device_count();
    if (n <= 0) {
        set_error("no HIP device visible: the k2u path runs on the GPU only (no CPU fallback)");
        return MZ_EHIP;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (have %d)", device, n);
        return MZ_EINVAL;
    }
    MZ_HIP_CHECK(hipSetDevice(device));
    DeviceIndex *d = new DeviceIndex();
    d->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) d->n_cu = (uint32_t)prop.multiProcessorCount;
    IndexView v = h.view();
    int rc = MZ_OK;
    auto fail = [&](int code) { device_free(d); return code; };
    if ((rc = upload(d, h.us.seqbv.data(), h.us.seqbv.size(), &v.seqbv)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.us.starts.data(), h.us.starts.size(), &v.starts)) != MZ_OK) return fail(rc);
    if ((rc = upload_mphf(d, h.mm, &v.mm)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.buckets.blocks.data(), h.buckets.blocks.size(), &v.buckets.blocks)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.buckets.ovf.data(), h.buckets.ovf.size(), &v.buckets.ovf)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.pos.w.data(), h.pos.w.size(), &v.pos)) != MZ_OK) return fail(rc);
    if (h.has_skew) {
        v.skew = h.skew.view();
        if ((rc = upload(d, h.skew.buckets.data(), h.skew.buckets.size(), &v.skew.buckets)) != MZ_OK) return fail(rc);
        if ((rc = upload(d, h.skew.ovf_word.data(), h.skew.ovf_word.size(), &v.skew.ovf_word)) != MZ_OK) return fail(rc);
        if ((rc = upload(d, h.skew.ovf_pos.data(), h.skew.ovf_pos.size(), &v.skew.ovf_pos)) != MZ_OK) return fail(rc);
    }
    if (h.front.present) {
        v.front = h.front.view();
        if ((rc = upload(d, h.front.lines.data(), h.front.lines.size(), &v.front.lines)) != MZ_OK) return fail(rc);
    }
    d->view = v;
    d->n_entries = h.pos.n;
    *out = d;
    return MZ_OK;
}

int device_download(const DeviceIndex &d, SSHashHost &h) {
    if (h.arrays_on_host) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    auto get = [&](std::vector<uint64_t> &dst, const uint64_t *src, uint64_t n) -> hipError_t {
        dst.resize(n);
        return n ? hipMemcpy(dst.data(), src, n * 8, hipMemcpyDeviceToHost) : hipSuccess;
    };
    const IndexView &v = d.view;
    MZ_HIP_CHECK(get(h.mm.blocks, v.mm.blocks, h.n_mm_block_words()));
    MZ_HIP_CHECK(get(h.buckets.blocks, v.buckets.blocks, h.n_bucket_block_words()));
    MZ_HIP_CHECK(get(h.buckets.ovf, v.buckets.ovf, h.n_bucket_ovf_words()));
    MZ_HIP_CHECK(get(h.pos.w, v.pos, h.pos.n + 2));
    if (h.has_skew) {
        MZ_HIP_CHECK(get(h.skew.buckets, v.skew.buckets, h.skew.n_slots()));
        MZ_HIP_CHECK(get(h.skew.ovf_word, v.skew.ovf_word, h.skew.n_ovf));
        MZ_HIP_CHECK(get(h.skew.ovf_pos, v.skew.ovf_pos, h.skew.n_ovf));
    }
    if (h.front.present) MZ_HIP_CHECK(get(h.front.lines, v.front.lines, h.front.n_words(h.w)));
    h.arrays_on_host = true;
    return MZ_OK;
}

void device_free(DeviceIndex *d) {
    if (!d) return;
    if (d->device >= 0) (void)hipSetDevice(d->device);
    for (void *p : d->allocs) (void)hipFree(p);
    delete d;
}

static unsigned grid_for(const DeviceIndex &d, uint64_t n_items, unsigned blocks_per_cu) {
    uint64_t need = (n_items + kBlock - 1) / kBlock;
    uint64_t cap = (uint64_t)d.n_cu * blocks_per_cu;
    return (unsigned)std::max<uint64_t>(1, std::min(need, cap));
}

int launch_k2u_batch(const DeviceIndex &d, const QueryOpts &opts, const uint64_t *d_fw, uint64_t n, mz_k2upos *d_out, mz_counts *d_counts,
                     hipStream_t stream) {
    if (n == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    if (d.view.total_len >= (1ull << kPosBits)) {
        set_error("unitig set longer than 2^%u bases is not supported by the tile kernel", kPosBits);
        return MZ_EINVAL;
    }
    if (opts.point_kernel == 3) { // lane-per-query kernel (kept for A/B measurements)
        hipLaunchKernelGGL(k2u_batch_kernel, dim3(grid_for(d, n, 32)), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
    } else if (opts.point_kernel != 2) {
        uint64_t n_blocks = (n + kTileQ - 1) / kTileQ; // four 256-query wave tiles per workgroup
        unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_blocks, (uint64_t)d.n_cu * 4096));
        if (d.view.k == 31 && d.view.w == 15)
            hipLaunchKernelGGL(k2u_wave_kernel<kKW3115>, dim3(grid), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
        else
            hipLaunchKernelGGL(k2u_wave_kernel<0>, dim3(grid), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
    } else { // workgroup-scoped tiles, one tile per workgroup (4 .. 1000 tiles per workgroup measured 10.5 .. 11.0 G/s when this was the default)
        uint64_t n_tiles = (n + kTileQ - 1) / kTileQ;
        unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)d.n_cu * 4096));
        if (d.view.k == 31 && d.view.w == 15)
            hipLaunchKernelGGL(k2u_tile_kernel<kKW3115>, dim3(grid), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
        else
            hipLaunchKernelGGL(k2u_tile_kernel<0>, dim3(grid), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
    }
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

static unsigned tile_grid(const DeviceIndex &d, uint64_t n_tiles, unsigned blocks_per_cu) {
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)d.n_cu * blocks_per_cu));
}

int launch_k2u_compact(const DeviceIndex &d, const uint64_t *d_fw, uint64_t n, uint64_t *d_hit_idx, mz_k2upos *d_hits,
                       uint64_t *d_n_hits, hipStream_t stream) {
    if (n == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    hipLaunchKernelGGL(k2u_compact_kernel, dim3(tile_grid(d, (n + kTileQ - 1) / kTileQ, 16)), dim3(kBlock), 0, stream, d.view, d_fw, n,
                       d_hit_idx, d_hits, (unsigned long long *)d_n_hits);
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_k2u_reads(const DeviceIndex &d, const QueryOpts &opts, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_read_offsets,
                     const uint64_t *d_kmer_offsets, uint64_t n_reads, int streaming, mz_k2upos *d_out,
                     mz_counts *d_counts, hipStream_t stream) {
    if (n_bases == 0 || n_reads == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    const dim3 grid(tile_grid(d, (n_bases + kTileQ - 1) / kTileQ, 8));
    (void)streaming; // same kernels: a streaming request changes nothing (see the note above k2u_reads_kernel)
    // The windows of a read share their minimizer for ~9 positions, and the windows of one strand class form runs of ~4.6.
    // Small indexes: one lookup per run through the minimizer records, the run's bucket then scanned by each of its windows
    // (stage S3).  Large indexes (2^28 bucket entries and more): through the minimizer lines, and there every window scans
    // its line itself as a point query does (stage SF: the flank test of all 16 slots costs a window less than S3 costs it
    // when buckets hold several entries, 4.5 per window on the GRCh38-shaped set; the line comes from the caches for all
    // but the first window of a run).  Measured in G windows/s, run / item: s-txome (60 M entries, lines forced) 31.2 / 28.5,
    // records 32.1; S-human (547 M) 21.2 / 23.4; s-pan (1.2 G) 15.9 / 19.3.  QueryOpts::reads_lines / reads_resolve override.
    IndexView v = d.view;
    const bool use_lines = opts.reads_lines ? opts.reads_lines > 0 : d.n_entries >= (1ull << 28);
    if ((!use_lines && v.mm.direct) || front_is_hashed(v.w)) v.front.lines = nullptr; // (hash-addressed lines serve point queries; reads of such an index resolve one MPHF lookup per run)
    const bool grouped = opts.reads_verify == 2; // grouped verify-and-extend (tile_finish_grouped) instead of one sequence-record gather per candidate
#define MZ_LAUNCH_READS(KERNEL, GRID) hipLaunchKernelGGL(KERNEL, GRID, dim3(kBlock), 0, stream, v, d_bases, n_bases, d_read_offsets, d_kmer_offsets, n_reads, d_out, d_counts)
    if (v.k == 31 && v.w == 15 && opts.reads_kernel != 2) {
        const uint64_t n_wave_tiles = (n_bases + kWaveTile - 1) / kWaveTile;
        // workgroups per CU: 3 are resident; 24 give every slot eight shorter runs of tiles to even the finish out (measured 3: 31.0, 8: 30.3 --
        // 2.67 rounds --, 24: 31.7, 96: 30.7 G windows/s on s-txome; every wave starts with a search for its first read)
        const dim3 wgrid(tile_grid(d, (n_wave_tiles + kBlock / 64 - 1) / (kBlock / 64), opts.reads_grid > 0 ? (unsigned)opts.reads_grid : 24u));
        const bool per_item = v.front.lines != nullptr && opts.reads_resolve != 2;
        if (per_item && grouped) MZ_LAUNCH_READS((k2u_reads_wave_kernel<kKW3115, true, true>), wgrid);
        else if (per_item) MZ_LAUNCH_READS((k2u_reads_wave_kernel<kKW3115, true, false>), wgrid);
        else if (grouped) MZ_LAUNCH_READS((k2u_reads_wave_kernel<kKW3115, false, true>), wgrid);
        else MZ_LAUNCH_READS((k2u_reads_wave_kernel<kKW3115, false, false>), wgrid);
    } else if (v.k == 31 && (v.w == 17 || v.w == 19 || v.w == 21) && d.view.front.lines != nullptr && opts.reads_kernel != 2 && opts.reads_lines >= 0) {
        // minimizers beyond 32 bits through the hash-addressed lines: the wave-scoped kernel, every window scanning its own line
        v.front = d.view.front;
        const uint64_t n_wave_tiles = (n_bases + kWaveTile - 1) / kWaveTile;
        const dim3 wgrid(tile_grid(d, (n_wave_tiles + kBlock / 64 - 1) / (kBlock / 64), opts.reads_grid > 0 ? (unsigned)opts.reads_grid : 24u));
        if (v.w == 17) MZ_LAUNCH_READS((k2u_reads_wave_kernel<(31u << 8) | 17u, true, false>), wgrid);
        else if (v.w == 19) MZ_LAUNCH_READS((k2u_reads_wave_kernel<(31u << 8) | 19u, true, false>), wgrid);
        else MZ_LAUNCH_READS((k2u_reads_wave_kernel<(31u << 8) | 21u, true, false>), wgrid);
    } else if (v.k == 31 && v.w == 15) {
        MZ_LAUNCH_READS((k2u_reads_kernel<kKW3115>), grid);
    } else {
        MZ_LAUNCH_READS((k2u_reads_kernel<0>), grid);
    }
#undef MZ_LAUNCH_READS
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_validate_self(const DeviceIndex &d, uint64_t *d_report, uint64_t *d_dup_list, uint64_t dup_cap, hipStream_t stream) {
    MZ_HIP_CHECK(hipSetDevice(d.device));
    uint64_t n_tiles = ((d.view.total_len + 64 * kQpt / 2 - 1) / (64 * kQpt / 2) + kBlock / 64 - 1) / (kBlock / 64); // four wave tiles per workgroup
    if (d.view.k == 31 && d.view.w == 15)
        hipLaunchKernelGGL(validate_self_kernel<kKW3115>, dim3(tile_grid(d, n_tiles, 16)), dim3(kBlock), 0, stream, d.view,
                           (unsigned long long *)d_report, (unsigned long long *)d_dup_list, dup_cap);
    else
        hipLaunchKernelGGL(validate_self_kernel<0>, dim3(tile_grid(d, n_tiles, 16)), dim3(kBlock), 0, stream, d.view,
                           (unsigned long long *)d_report, (unsigned long long *)d_dup_list, dup_cap);
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_line_stats(const DeviceIndex &d, uint64_t *d_out, hipStream_t stream) {
    MZ_HIP_CHECK(hipSetDevice(d.device));
    const uint64_t n_slots = d.view.front.lines ? d.view.front.n_lines * kFrontSlots : 0;
    if (n_slots) hipLaunchKernelGGL(line_stats_kernel, dim3(grid_for(d, n_slots, 16)), dim3(kBlock), 0, stream, d.view.front.lines, n_slots, (unsigned long long *)d_out);
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_extract_kmers(const DeviceIndex &d, const uint64_t *d_positions, const uint8_t *d_flip, uint64_t n,
                         uint64_t *d_fw, hipStream_t stream) {
    if (n == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    hipLaunchKernelGGL(extract_kmers_kernel, dim3(grid_for(d, n, 32)), dim3(kBlock), 0, stream, d.view, d_positions, d_flip, n, d_fw);
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

} // namespace mz
