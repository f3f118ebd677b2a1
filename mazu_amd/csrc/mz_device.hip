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
//   S0  canonical encode + minimizer (ALU only)                      dense over queries
//   S1  MPHF probe, one level (pair) per round; unresolved keys are ballot-compacted into an
//       LDS list for the next round (geometric shrink)               dense over pending keys
//   S2  Elias-Fano get(h), get(h+1) -> bucket range                  dense over queries
//   S3  bucket entries flattened by an LDS prefix sum: one (query, entry) item per lane
//       iteration: packed (pos, strand class) -> the one candidate k-mer of that class ->
//       verify; the first verified entry in bucket order is kept with an LDS atomicMin
//   S4  heavy buckets: skew MPHF (same level loop) -> packed pos -> verify
//   S5  unitig locate + boundary check of the winner (exact rescan if it straddles two unitigs),
//       results stored as coalesced 16-byte records
// ------------------------------------------------------------------------------------
constexpr int kQpt = 4;                  // queries per thread
constexpr int kTileQ = kBlock * kQpt;    // queries per tile
constexpr unsigned long long kBestNone = ~0ull;
constexpr uint32_t kPosBits = 36, kNMiss = 0xFFFFFFFFu, kNSkew = 0xFFFFFFFEu;

struct TileLds {
    uint64_t fw[kTileQ];               // forward word of the query
    uint64_t slot[kTileQ];             // minimizer word -> bucket id h -> bucket start (skew: canonical word -> slot)
    unsigned long long best[kTileQ];   // (bucket order key << 37) | (twin << 36) | global k-mer position
    uint32_t n[kTileQ];                // bucket size, kNSkew, or kNMiss
    uint32_t eoff[kTileQ + 1];         // exclusive prefix sum of bucket sizes
    uint16_t list_a[kTileQ], list_b[kTileQ];
    uint8_t off[kTileQ];               // minimizer offset inside g*
    uint32_t cnt_a, cnt_b, cnt_skew, wave_tot[kBlock / 64];
};

// append `idx` of the lanes with pred to an LDS list (one LDS atomic per wave)
__device__ __forceinline__ void wave_push(uint16_t *list, uint32_t *cnt, bool pred, uint32_t idx) {
    unsigned long long b = __ballot(pred);
    if (b == 0) return;
    unsigned lane = __lane_id();
    unsigned leader = (unsigned)(__ffsll((long long)b) - 1);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(cnt, (uint32_t)__popcll(b));
    base = __shfl(base, (int)leader);
    if (pred) list[base + __popcll(b & ((1ull << lane) - 1))] = (uint16_t)idx;
}

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
// through every level get n[i] = kNMiss.  `dense_first` == all kTileQ items (level 0, no list).
__device__ __forceinline__ void tile_mphf(TileLds &S, const MphfView &m, bool dense_first, uint32_t n_valid) {
    const uint32_t t = threadIdx.x;
    uint32_t lv = 0;
    if (dense_first) {
        if (m.n_levels == 0) {
            for (int q = 0; q < kQpt; ++q) { uint32_t i = q * kBlock + t; if (i < n_valid) S.n[i] = kNMiss; }
            return;
        }
        const MphfLevel L = m.levels[0];
        MphfProbe pr[kQpt];
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            pr[q] = mphf_issue(m, L, i < n_valid ? S.slot[i] : 0);
        }
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            uint64_t idx;
            bool hit = mphf_eval(pr[q], idx);
            if (hit && i < n_valid) S.slot[i] = idx;
            wave_push(S.list_a, &S.cnt_a, !hit && i < n_valid, i);
        }
        lv = 1;
    }
    uint16_t *cur = S.list_a, *nxt = S.list_b;
    uint32_t *ccur = &S.cnt_a, *cnxt = &S.cnt_b;
    for (;;) {
        __syncthreads();
        const uint32_t count = *ccur;
        if (count == 0) break;
        if (lv >= m.n_levels) { // fell through every level: boomphf try_hash -> None
            for (uint32_t j = t; j < count; j += kBlock) S.n[cur[j]] = kNMiss;
            __syncthreads();
            if (t == 0) *ccur = 0;
            break;
        }
        const bool two = lv + 1 < m.n_levels; // probe two levels per round (speculative second)
        const MphfLevel L0 = m.levels[lv], L1 = m.levels[two ? lv + 1 : lv];
        for (uint32_t j = t; j < ((count + 63) & ~63u); j += kBlock) {
            bool act = j < count;
            uint32_t i = act ? cur[j] : 0;
            uint64_t key = S.slot[i];
            MphfProbe p0 = mphf_issue(m, L0, key), p1 = mphf_issue(m, L1, key);
            uint64_t idx;
            bool hit = mphf_eval(p0, idx) || (two && mphf_eval(p1, idx));
            if (act && hit) S.slot[i] = idx;
            wave_push(nxt, cnxt, act && !hit, i);
        }
        lv += two ? 2 : 1;
        __syncthreads();
        if (t == 0) *ccur = 0;
        uint16_t *tl = cur; cur = nxt; nxt = tl;
        uint32_t *tc = ccur; ccur = cnxt; cnxt = tc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k2u_tile_kernel(IndexView ix, const uint64_t *__restrict__ fw_words, uint64_t n,
                                                          mz_k2upos *__restrict__ out, mz_counts *counts) {
    __shared__ TileLds S;
    const uint32_t t = threadIdx.x, k = ix.k, w = ix.w;
    const uint64_t kmask = kmer_mask(k);
    const uint64_t n_tiles = (n + kTileQ - 1) / kTileQ;
    Tally tally;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * kTileQ;
        const uint32_t n_valid = (uint32_t)(n - base < (uint64_t)kTileQ ? n - base : (uint64_t)kTileQ);
        if (t == 0) { S.cnt_a = 0; S.cnt_b = 0; S.cnt_skew = 0; }
        // ---- S0: canonical encode + minimizer
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            uint64_t fw = i < n_valid ? (fw_words[base + i] & kmask) : 0;
            uint64_t rc = kmer_rc(fw, k);
            Minimizer mn = canonical_minimizer(fw, rc, k, w, ix.seed);
            S.fw[i] = fw;
            S.slot[i] = mn.word;
            S.off[i] = (uint8_t)mn.offset;
            S.best[i] = kBestNone;
            S.n[i] = 0;
        }
        __syncthreads();
        // ---- S1: minimizer MPHF
        tile_mphf(S, ix.mm, true, n_valid);
        // ---- S2: bucket range from the Elias-Fano prefix sums
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            bool act = i < n_valid && S.n[i] != kNMiss;
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
        __syncthreads();
        // ---- S3: flatten bucket entries (exclusive scan of n over the tile, blocked by thread)
        {
            uint32_t v[kQpt], sum = 0;
#pragma unroll
            for (int q = 0; q < kQpt; ++q) {
                uint32_t nn = S.n[t * kQpt + q];
                v[q] = sum;
                sum += (nn >= kNSkew) ? 0u : nn;
            }
            uint32_t incl = sum;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t o = __shfl_up(incl, d);
                if ((int)(t & 63) >= d) incl += o;
            }
            if ((t & 63) == 63) S.wave_tot[t >> 6] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (uint32_t wv = 0; wv < (t >> 6); ++wv) wbase += S.wave_tot[wv];
            uint32_t excl = wbase + incl - sum;
#pragma unroll
            for (int q = 0; q < kQpt; ++q) S.eoff[t * kQpt + q] = excl + v[q];
            if (t == kBlock - 1) S.eoff[kTileQ] = excl + sum;
        }
        __syncthreads();
        {
            const uint32_t E = S.eoff[kTileQ];
            const uint64_t last_km_start = ix.total_len - k;
            for (uint32_t e = t; e < E; e += kBlock) {
                // owner item: largest i with eoff[i] <= e
                uint32_t lo = 0, hi = kTileQ;
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (S.eoff[mid] <= e) lo = mid; else hi = mid;
                }
                const uint32_t i = lo, j = e - S.eoff[i];
                const uint64_t ent = read_packed(ix.pos, S.slot[i] + j, ix.pos_width);     // sshash.rs:495
                const uint64_t mm_pos = ent >> 1;
                const uint64_t fw = S.fw[i], rc = kmer_rc(fw, k);
                const uint64_t off_a = S.off[i];
                const uint64_t off = (ent & 1) ? k - off_a - w : off_a;                  // class 1: candidate B (:527), class 0: A (:498)
                if (mm_pos >= off && mm_pos - off <= last_km_start) {
                    const uint64_t p = mm_pos - off;
                    const uint32_t mt = word_equivalency(fw, rc, read_kmer(ix.useq, p, k));
                    const uint64_t jj = j < 0x7FFFFFFu ? j : 0x7FFFFFFu;
                    // optimistic: the unitig-boundary check (:509-514) is done once, in S5
                    if (mt != MZ_NO_MATCH)
                        atomicMin(&S.best[i], (jj << (kPosBits + 1)) | ((uint64_t)(mt == MZ_TWIN_MATCH) << kPosBits) | p);
                }
            }
        }
        __syncthreads();
        // ---- S4: heavy buckets go through the skew index (sshash.rs:415-433)
        if (S.cnt_skew != 0) { // uniform
            const uint32_t ns = S.cnt_skew;
            __syncthreads();
            if (ix.has_skew) {
                for (uint32_t j = t; j < ns; j += kBlock) {
                    uint32_t i = S.list_b[j];
                    uint64_t fw = S.fw[i], rc = kmer_rc(fw, k);
                    S.slot[i] = fw < rc ? fw : rc;
                    S.list_a[j] = (uint16_t)i;
                    S.eoff[j] = i; // eoff is free now: remember the heavy items across the level loop
                }
                if (t == 0) { S.cnt_a = ns; S.cnt_b = 0; }
                __syncthreads();
                tile_mphf(S, ix.skew, false, n_valid);
                for (uint32_t j = t; j < ns; j += kBlock) {
                    uint32_t i = S.eoff[j];
                    if (S.n[i] == kNMiss) continue;
                    uint64_t p = read_packed(ix.skew_pos, S.slot[i], ix.skew_pos_width);
                    uint64_t fw = S.fw[i], rc = kmer_rc(fw, k);
                    uint32_t mt = word_equivalency(fw, rc, read_kmer(ix.useq, p, k));
                    if (mt != MZ_NO_MATCH) S.best[i] = ((uint64_t)(mt == MZ_TWIN_MATCH) << kPosBits) | p;
                }
            }
            __syncthreads();
        }
        // ---- S5: materialise results, coalesced 16-byte stores, one tally atomic per wave
#pragma unroll
        for (int q = 0; q < kQpt; ++q) {
            uint32_t i = q * kBlock + t;
            bool act = i < n_valid;
            unsigned long long b = S.best[i];
            mz_k2upos r = make_none();
            if (act && b != kBestNone) {
                uint64_t p = b & ((1ull << kPosBits) - 1);
                UnitigSpan s = locate_unitig(ix.bv_blocks, p);
                if (p + k <= s.end)
                    r = make_hit(s, p, ((b >> kPosBits) & 1) ? MZ_TWIN_MATCH : MZ_IDENTITY_MATCH);
                else // first verified candidate straddles two unitigs (sshash.rs:509-514): exact in-order rescan
                    r = lookup(ix, S.fw[i]);
            }
            if (act) out[base + i] = r;
            tally.add(act && r.match != MZ_NO_MATCH, act && r.match == MZ_NO_MATCH, false);
        }
        __syncthreads();
    }
    tally.flush(counts);
}

// hits only, compacted per wavefront with a ballot + prefix popcount (K7)
__global__ __launch_bounds__(kBlock) void k2u_compact_kernel(IndexView ix, const uint64_t *__restrict__ fw_words, uint64_t n,
                                                             uint64_t *__restrict__ hit_idx, mz_k2upos *__restrict__ hits,
                                                             unsigned long long *n_hits) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        mz_k2upos r = lookup(ix, fw_words[i]);
        bool hit = r.match != MZ_NO_MATCH;
        unsigned long long b = __ballot(hit);
        unsigned long long active = __ballot(1);
        unsigned lane = __lane_id();
        unsigned leader = (unsigned)(__ffsll((long long)active) - 1);
        unsigned long long base = 0;
        if (lane == leader && b) base = atomicAdd(n_hits, (unsigned long long)__popcll(b));
        base = __shfl(base, (int)leader);
        if (hit) {
            unsigned long long slot = base + __popcll(b & ((1ull << lane) - 1));
            hit_idx[slot] = i;
            hits[slot] = r;
        }
    }
}

// ------------------------------------------------------------------------------------
// K: reads (mz_sshash_k2u_reads_device).  One workgroup = 256 consecutive base positions of
// the concatenated read buffer.  The window is staged in LDS as three bit planes (low code
// bit, high code bit, invalid) built with wavefront ballots; every w-mer hash of the tile is
// computed ONCE into LDS (forward and reverse-complement strand) and each lane takes the
// minimum over its k-w+1 window instead of re-hashing -- the sliding-window minimizer.
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

constexpr int kTile = kBlock;           // base positions per workgroup
constexpr int kTileWords = kTile / 64 + 2; // plane words incl. tail + pad

__global__ __launch_bounds__(kBlock) void k2u_reads_kernel(IndexView ix, const uint8_t *__restrict__ bases, uint64_t n_bases,
                                                           const uint64_t *__restrict__ read_offsets,
                                                           const uint64_t *__restrict__ kmer_offsets, uint64_t n_reads,
                                                           mz_k2upos *__restrict__ out, mz_counts *counts) {
    __shared__ uint64_t plane_lo[kTileWords], plane_hi[kTileWords], plane_inv[kTileWords];
    __shared__ uint64_t hf[kTile + 32], hr[kTile + 32];
    __shared__ uint64_t s_read0;
    const uint32_t k = ix.k, w = ix.w, n_win = k - w + 1;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t B0 = (uint64_t)blockIdx.x * kTile;

    if (t < kTileWords) { plane_lo[t] = 0; plane_hi[t] = 0; plane_inv[t] = 0; }
    if (t == 0) { // read holding the first base of the tile: largest r with read_offsets[r] <= B0
        uint64_t lo = 0, hi = n_reads; // invariant: read_offsets[lo] <= B0 (read_offsets[0] == 0)
        while (hi - lo > 1) {
            uint64_t mid = (lo + hi) >> 1;
            if (read_offsets[mid] <= B0) lo = mid; else hi = mid;
        }
        s_read0 = lo;
    }
    __syncthreads();
    // stage: main 256 positions (one ballot triple per wave) then the k-1 tail (wave 0)
    {
        uint64_t b = B0 + t;
        uint32_t c = b < n_bases ? base_code_dev(bases[b]) : 4u;
        unsigned long long blo = __ballot(c & 1u), bhi = __ballot((c >> 1) & 1u), binv = __ballot(c >> 2);
        if (lane == 0) { plane_lo[wave] = blo; plane_hi[wave] = bhi; plane_inv[wave] = binv; }
        if (wave == 0) {
            uint64_t bt = B0 + kTile + lane;
            uint32_t ct = (lane < k - 1 && bt < n_bases) ? base_code_dev(bases[bt]) : 4u;
            unsigned long long tlo = __ballot(ct & 1u), thi = __ballot((ct >> 1) & 1u), tinv = __ballot(ct >> 2);
            if (lane == 0) { plane_lo[kTile / 64] = tlo; plane_hi[kTile / 64] = thi; plane_inv[kTile / 64] = tinv; }
        }
    }
    __syncthreads();
    // w-mer hashes of the tile, both strands (positions t and, for the first k-w lanes, kTile+t)
    for (uint32_t j = t; j < kTile + n_win - 1; j += kBlock) {
        uint64_t m = spread_bits(plane_window(plane_lo, j, w)) | (spread_bits(plane_window(plane_hi, j, w)) << 1);
        hf[j] = hash_mmer(m, ix.seed, w);
        hr[j] = hash_mmer(kmer_rc(m, w), ix.seed, w);
    }
    __syncthreads();

    const uint64_t b = B0 + t;
    bool has_window = false, valid = false;
    uint64_t out_idx = 0;
    if (b < n_bases) {
        uint64_t r = s_read0;
        while (r + 1 < n_reads && read_offsets[r + 1] <= b) ++r;
        uint64_t rs = read_offsets[r], re = read_offsets[r + 1];
        if (b >= rs && b + k <= re) {
            has_window = true;
            out_idx = kmer_offsets[r] + (b - rs);
            valid = plane_window(plane_inv, t, k) == 0;
        }
    }
    mz_k2upos res = make_none();
    if (has_window) {
        if (valid) {
            uint64_t fw = spread_bits(plane_window(plane_lo, t, k)) | (spread_bits(plane_window(plane_hi, t, k)) << 1);
            uint64_t rc = kmer_rc(fw, k);
            Minimizer mn;
            if (fw <= rc) { // g* = fw: leftmost minimum of the forward hashes
                uint32_t bo = 0;
                uint64_t bh = hf[t];
                for (uint32_t o = 1; o < n_win; ++o) {
                    uint64_t h = hf[t + o];
                    if (h < bh) { bh = h; bo = o; }
                }
                mn.offset = bo;
                mn.word = (fw >> (2 * bo)) & kmer_mask(w);
            } else { // g* = rc: offset o of g* is forward position t + (k-w-o); leftmost o == rightmost j
                uint32_t bj = 0;
                uint64_t bh = hr[t];
                for (uint32_t j = 1; j < n_win; ++j) {
                    uint64_t h = hr[t + j];
                    if (h <= bh) { bh = h; bj = j; }
                }
                mn.offset = (k - w) - bj;
                mn.word = (rc >> (2 * mn.offset)) & kmer_mask(w);
            }
            res = lookup_with_minimizer(ix, fw, rc, mn);
        } else {
            res.match = MZ_SKIPPED;
        }
        out[out_idx] = res;
    }
    Tally tally;
    tally.add(has_window && valid && res.match != MZ_NO_MATCH, has_window && valid && res.match == MZ_NO_MATCH, has_window && !valid);
    tally.flush(counts);
}

// ------------------------------------------------------------------------------------
// K: validate_self on the device (kphf/mod.rs:69-139): thread per global base position
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void validate_self_kernel(IndexView ix, unsigned long long *report) {
    __shared__ uint64_t s_uid0;
    const uint32_t k = ix.k;
    const uint64_t P0 = (uint64_t)blockIdx.x * kBlock;
    if (threadIdx.x == 0) {
        uint64_t lo = 0, hi = ix.n_unitigs;
        while (hi - lo > 1) {
            uint64_t mid = (lo + hi) >> 1;
            if (ix.starts[mid] <= P0) lo = mid; else hi = mid;
        }
        s_uid0 = lo;
    }
    __syncthreads();
    const uint64_t p = P0 + threadIdx.x;
    unsigned n_k = 0, n_fail = 0, n_dup = 0;
    if (p + k <= ix.total_len) {
        uint64_t ui = s_uid0;
        while (ix.starts[ui + 1] <= p) ++ui;
        uint64_t us = ix.starts[ui], ue = ix.starts[ui + 1];
        if (p + k <= ue) {
            n_k = 1;
            uint64_t fw = read_kmer(ix.useq, p, k), rc = kmer_rc(fw, k);
            for (int o = 0; o < 2; ++o) { // km, then km.swap()
                uint64_t q = o ? rc : fw;
                mz_k2upos r = lookup(ix, q);
                uint32_t want = o ? MZ_TWIN_MATCH : MZ_IDENTITY_MATCH;
                bool ok = r.match == want && r.unitig_id == (uint32_t)ui && r.pos == (uint32_t)(p - us) &&
                          r.unitig_len == (uint32_t)(ue - us);
                if (!ok) {
                    bool dup = false;
                    if (r.match != MZ_NO_MATCH && r.unitig_id < ix.n_unitigs) {
                        uint64_t at = ix.starts[r.unitig_id] + r.pos;
                        if (at + k <= ix.total_len) {
                            uint64_t kw = read_kmer(ix.useq, at, k);
                            dup = (kw == fw || kw == rc) && at != p;
                        }
                    }
                    if (dup) ++n_dup; else { ++n_fail; atomicMin(&report[3], (unsigned long long)p); }
                }
            }
        }
    }
    // wave reduce with ballots over the three small counters
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

// forward words at given global positions (positive-query synthesis, stays in HBM)
__global__ __launch_bounds__(kBlock) void extract_kmers_kernel(IndexView ix, const uint64_t *__restrict__ positions,
                                                               const uint8_t *__restrict__ flip, uint64_t n,
                                                               uint64_t *__restrict__ fw_words) {
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t p = positions[i];
        uint64_t x = p + ix.k <= ix.total_len ? read_kmer(ix.useq, p, ix.k) : 0;
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
    if ((rc = upload(d, h.us.useq.data(), h.us.useq.size(), &v.useq)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.us.bv_blocks.data(), h.us.bv_blocks.size(), &v.bv_blocks)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.us.starts.data(), h.us.starts.size(), &v.starts)) != MZ_OK) return fail(rc);
    if ((rc = upload_mphf(d, h.mm, &v.mm)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.buckets.blocks.data(), h.buckets.blocks.size(), &v.buckets.blocks)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.buckets.ovf.data(), h.buckets.ovf.size(), &v.buckets.ovf)) != MZ_OK) return fail(rc);
    if ((rc = upload(d, h.pos.w.data(), h.pos.w.size(), &v.pos)) != MZ_OK) return fail(rc);
    if (h.has_skew) {
        if ((rc = upload_mphf(d, h.skew, &v.skew)) != MZ_OK) return fail(rc);
        if ((rc = upload(d, h.skew_pos.w.data(), h.skew_pos.w.size(), &v.skew_pos)) != MZ_OK) return fail(rc);
    }
    d->view = v;
    *out = d;
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

static bool use_v1_kernel() {
    static const bool v = [] { const char *e = getenv("MZ_K2U_KERNEL"); return e && e[0] == 'v' && e[1] == '1'; }();
    return v;
}

int launch_k2u_batch(const DeviceIndex &d, const uint64_t *d_fw, uint64_t n, mz_k2upos *d_out, mz_counts *d_counts,
                     hipStream_t stream) {
    if (n == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    if (d.view.total_len >= (1ull << kPosBits)) {
        set_error("unitig set longer than 2^%u bases is not supported by the tile kernel", kPosBits);
        return MZ_EINVAL;
    }
    if (use_v1_kernel()) { // lane-per-query kernel (kept for A/B measurements)
        hipLaunchKernelGGL(k2u_batch_kernel, dim3(grid_for(d, n, 32)), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
    } else {
        uint64_t n_tiles = (n + kTileQ - 1) / kTileQ;
        unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)d.n_cu * 16));
        hipLaunchKernelGGL(k2u_tile_kernel, dim3(grid), dim3(kBlock), 0, stream, d.view, d_fw, n, d_out, d_counts);
    }
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_k2u_compact(const DeviceIndex &d, const uint64_t *d_fw, uint64_t n, uint64_t *d_hit_idx, mz_k2upos *d_hits,
                       uint64_t *d_n_hits, hipStream_t stream) {
    if (n == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    hipLaunchKernelGGL(k2u_compact_kernel, dim3(grid_for(d, n, 32)), dim3(kBlock), 0, stream, d.view, d_fw, n, d_hit_idx, d_hits,
                       (unsigned long long *)d_n_hits);
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_k2u_reads(const DeviceIndex &d, const uint8_t *d_bases, uint64_t n_bases, const uint64_t *d_read_offsets,
                     const uint64_t *d_kmer_offsets, uint64_t n_reads, int /*streaming*/, mz_k2upos *d_out,
                     mz_counts *d_counts, hipStream_t stream) {
    if (n_bases == 0 || n_reads == 0) return MZ_OK;
    MZ_HIP_CHECK(hipSetDevice(d.device));
    uint64_t n_tiles = (n_bases + kTile - 1) / kTile;
    if (n_tiles > 0x7FFFFFFFull) {
        set_error("reads batch too large (%llu bases)", (unsigned long long)n_bases);
        return MZ_EINVAL;
    }
    hipLaunchKernelGGL(k2u_reads_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, stream, d.view, d_bases, n_bases,
                       d_read_offsets, d_kmer_offsets, n_reads, d_out, d_counts);
    MZ_HIP_CHECK(hipGetLastError());
    return MZ_OK;
}

int launch_validate_self(const DeviceIndex &d, uint64_t *d_report, hipStream_t stream) {
    MZ_HIP_CHECK(hipSetDevice(d.device));
    uint64_t n_blocks = (d.view.total_len + kBlock - 1) / kBlock;
    if (n_blocks > 0x7FFFFFFFull) {
        set_error("validate_self: sequence too long for one launch");
        return MZ_EINVAL;
    }
    hipLaunchKernelGGL(validate_self_kernel, dim3((unsigned)n_blocks), dim3(kBlock), 0, stream, d.view,
                       (unsigned long long *)d_report);
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
