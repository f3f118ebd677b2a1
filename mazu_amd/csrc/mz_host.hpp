// mz_host.hpp -- host-side owner of the index layout (the role of the Rust host code in the reference):
// UnitigSet, Elias-Fano, rank-interleaved MPHF and the SSHash builder.
// Follows the construction steps of reference src/kphf/sshash.rs:86-329 but lays every
// structure out for 128-byte-line gathers on MI355X (one dependent step = one line; layouts in mz_common.hpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mz_common.hpp"

namespace mz {

int hw_threads();

// Layout choices of a build that are not part of the reference's (w, skew_param, seed): arguments of
// mz_sshash_from_unitig_set_opts.  0 = decide by the size rules below.  The environment variables MZ_FRONT_TABLE (0 / 1),
// MZ_FRONT_MAX_LOAD and MZ_DIRECT_BYTES_PER_KEY of the A/B tools fill the fields a caller left at 0 (with_env, once per build).
struct BuildOpts {
    int minimizer_lines = 0;           // 1: always build the minimizer lines (2w <= 32), -1: never, 0: by the size rule
    uint32_t lines_max_load = 0;       // mean entries per line of 16 slots the table may carry; 0: 3, else 5, else 10 (front_choose_shape)
    uint64_t direct_bytes_per_key = 0; // the direct-address minimizer records may take this many bytes per minimizer; 0: the rule of use_direct_records
    BuildOpts with_env() const;
};
// Kernel choices of the query entry points that do not change results: arguments of mz_sshash_set_query_opts, kept per index
// handle.  0 = decide by the rules in launch_k2u_batch / launch_k2u_reads.  The A/B tools' MZ_K2U_KERNEL, MZ_READS_LINES,
// MZ_READS_RESOLVE, MZ_READS_KERNEL, MZ_READS_VERIFY and MZ_READS_GRID fill a NEW handle's fields (from_env, once per handle).
struct QueryOpts {
    int point_kernel = 0;  // 1: wave-scoped tiles (the rule), 2: workgroup-scoped tiles, 3: one lane per query
    int reads_lines = 0;   // reads resolve through 1: the minimizer lines, -1: the minimizer records; rule: lines from 2^28 bucket entries on
    int reads_resolve = 0; // through the lines, 1: every window scans its line (the rule), 2: one lookup per run of windows
    int reads_kernel = 0;  // 1: wave-scoped (the rule at k = 31, w = 15), 2: workgroup-staged
    int reads_verify = 0;  // 1: one sequence-record gather per candidate (the rule), 2: grouped verify-and-extend (wave-scoped kernel)
    int reads_grid = 0;    // workgroups per CU of the wave-scoped reads kernel; rule: 24
    static QueryOpts from_env();
};

// fixed-width packed integers (the reference's IntVector::pack, sshash.rs:314-315)
struct PackedVec {
    std::vector<uint64_t> w; // padded with 2 words
    uint64_t n = 0;
    uint32_t width = 1;
    void assign(const uint64_t *xs, uint64_t n_, uint64_t max_value, int n_threads);
    uint64_t get(uint64_t i) const { return read_packed(w.data(), i, width); }
    uint64_t num_bits() const { return n * width; }
};

// Elias-Fano monotone sequence (reference src/elias_fano.rs:21-122): l = max(1, msb(u/n)),
// high part in unary with n + (u >> l) bits, low part packed.  `sel` replaces simple-sds'
// select support with one sampled position per 2^kEfSelLog2 ones.
struct EFVector {
    uint64_t u = 0, n = 0, n_high_bits = 0;
    uint32_t l = 1;
    std::vector<uint64_t> high, sel, low;
    // returns MZ_OK, MZ_ENOTMONOTONE or MZ_EEMPTY (Error::EFNotMonotone / EFEmpty)
    int build(const uint64_t *xs, uint64_t n_, uint64_t u_);
    EfView view() const { return EfView{high.data(), sel.data(), low.data(), n, (uint64_t)high.size(), l, 0}; }
    uint64_t get(uint64_t i) const { return ef_get(view(), i); }
    uint64_t len() const { return n; }
    uint64_t num_bits() const { return 3 * 64 + n_high_bits + n * l; } // elias_fano.rs:31-33
};

// Partitioned Elias-Fano (one 64-byte line per 64 elements), see BefView in mz_common.hpp.
struct BlockedEF {
    uint64_t n = 0, n_overflow = 0;
    std::vector<uint64_t> blocks, ovf;
    int build(const uint64_t *xs, uint64_t n_, int n_threads);
    BefView view() const { return BefView{blocks.data(), ovf.data(), n}; }
    void get_pair(uint64_t i, uint64_t &a, uint64_t &b) const { bef_get_pair(view(), i, a, b); }
    uint64_t num_bits() const { return (blocks.size() + ovf.size()) * 64; }
};

// Minimal perfect hash in the BBHash/boomphf family (reference uses boomphf::Mphf, gamma 1.7,
// sshash.rs:177,280) stored as rank-interleaved 64-byte blocks.
struct Mphf {
    std::vector<uint64_t> blocks;
    std::vector<MphfLevel> levels;
    uint64_t n_keys = 0;
    uint32_t direct = 0;
    // keys must be distinct; returns MZ_OK or MZ_EINVAL (too many levels)
    int build(const uint64_t *keys, uint64_t n, double gamma, int n_threads);
    // direct-address variant for keys < 2^key_bits (key_bits <= 32): index = rank of the key
    int build_direct(const uint64_t *keys, uint64_t n, uint32_t key_bits, int n_threads);
    // direct mode: encode the bucket starts of every record's members into its second half; returns #records left to the global EF
    uint64_t fill_bucket_offsets(const uint64_t *psum, uint64_t n_psum, int n_threads);
    void fill_filters(const uint64_t *keys, uint64_t n, const uint64_t *psum, const uint64_t *entries, uint64_t skew_param, int n_threads);
    MphfView view() const { return MphfView{blocks.data(), levels.data(), (uint32_t)levels.size(), direct, n_keys}; }
    bool try_hash(uint64_t key, uint64_t &idx) const { return mphf_lookup(view(), key, idx); }
    uint64_t num_bits() const { return blocks.size() * 64 + levels.size() * sizeof(MphfLevel) * 8; }
};

// UnitigSet (reference src/unitig_set.rs:30-36)
struct UnitigSet {
    uint32_t k = 0;
    uint64_t total_len = 0, n_unitigs = 0;
    std::vector<uint64_t> useq;      // 2-bit packed + 2 pad words
    std::vector<uint64_t> starts;    // n_unitigs + 1 (plain copy of accum_lens)
    EFVector accum_lens;             // as in the reference
    std::vector<uint64_t> bv_blocks; // boundary bit vector, rank/select-interleaved blocks
    std::vector<uint64_t> seqbv;     // 64-byte records {128 bases | their unitig boundaries}: what the query path reads
    void build_seqbv();

    int from_packed(const uint64_t *words, uint64_t total_len_, const uint64_t *starts_, uint64_t n, uint32_t k_);
    int from_seqs(const char *const *seqs, uint64_t n, uint32_t k_);
    int from_cf_prefix(const std::string &prefix);
    uint64_t n_kmers() const { return total_len - (uint64_t)k * n_unitigs + n_unitigs; } // :212-214
    uint64_t unitig_len(uint64_t i) const { return accum_lens.get(i + 1) - accum_lens.get(i); } // :178-180
    uint64_t unitig_start_pos(uint64_t i) const { return accum_lens.get(i); }
    uint64_t pos_to_id(uint64_t pos) const { return locate_unitig(bv_blocks.data(), pos).id; } // :185-187
    uint64_t num_bits() const { return 64 + total_len * 2 + accum_lens.num_bits() + total_len; } // :41-47
private:
    int finish();
};

// SSHash (reference src/kphf/sshash.rs:39-57)
// Hash table of the skew index (layout: SkewView in mz_common.hpp).  build() takes the k-mer words in increasing order
// with their positions; the result depends only on that set (the GPU builder produces the same arrays).
struct SkewTable {
    std::vector<uint64_t> buckets, ovf_word, ovf_pos;
    uint64_t n_buckets = 0, n_keys = 0, n_ovf = 0; // n_ovf == ovf_word.size() whenever the arrays are on the host
    uint32_t fp_bits = kSkewFpBits; // fingerprint width of a slot
    uint32_t fp_mask() const { return (uint32_t)((1ull << fp_bits) - 1); }
    void build(const uint64_t *words, const uint64_t *pos, uint64_t n, int n_threads);
    SkewView view() const { return SkewView{buckets.data(), ovf_word.data(), ovf_pos.data(), n_buckets, n_ovf, n_keys, fp_mask(), 0}; }
    uint64_t n_slots() const { return n_buckets * kSkewSlots > 2 ? n_buckets * kSkewSlots : 2; }
    uint64_t num_bits() const { return 64 * (n_slots() + 2 * n_ovf); }
};

// Minimizer lines (layout: FrontView in mz_common.hpp).  Built from the minimizer records, the bucket starts and the
// entry array; the GPU builder produces the same words.
struct FrontTable {
    std::vector<uint64_t> lines;
    uint32_t g = 0, t = 0;
    bool present = false;
    // shape of the table for this index, false when none is built.  opts.minimizer_lines -1 never builds the lines, 1 builds them
    // at a load of at most 10 whatever the table size, otherwise front_choose_shape decides; opts.lines_max_load sets the load
    static bool choose(uint32_t w, uint64_t n_occ, bool direct, const BuildOpts &opts, uint32_t &g, uint32_t &t);
    // after a table turned out to have a line with more than 16 member keys: the next shape to try (half the tag), false when there is none
    static bool next_shape(uint32_t w, uint64_t n_occ, const BuildOpts &opts, uint32_t &g, uint32_t &t);
    void build(const uint64_t *minrec, uint32_t w, uint64_t n_occ, bool direct, const uint64_t *psum, const uint64_t *entries, uint64_t skew_param, int n_threads,
               const BuildOpts &opts);
    // 2w > 32: lines addressed by a hash of the word (front_hash in mz_common.hpp), built from the sorted minimizer set and each
    // word's bucket id; g = 0 and t = log2(lines) then
    static bool choose_hashed(uint32_t w, uint64_t n_occ, uint64_t n_mm, const BuildOpts &opts, uint32_t &bits);
    void build_hashed(const uint64_t *mm_set, const uint64_t *bucket_of, uint64_t n_mm, uint32_t w, uint64_t n_occ, const uint64_t *psum, const uint64_t *entries,
                      uint64_t skew_param, int n_threads, const BuildOpts &opts);
    uint64_t n_unusable = 0, n_walk = 0; // lines given up (more than 16 words); walk markers (words sharing a fingerprint in their line)
    FrontView view() const { return present ? FrontView{lines.data(), (uint64_t)lines.size() / kFrontSlots, g, t} : FrontView{nullptr, 0, 0, 0}; }
    uint64_t n_words(uint32_t w) const { return !present ? 0 : front_is_hashed(w) ? (uint64_t)kFrontSlots << t : (1ull << (2 * w - g - t)) * kFrontSlots; }
};

struct SSHashHost {
    UnitigSet us;
    uint32_t w = 0;
    uint64_t seed = 0;
    uint64_t skew_param = ~0ull;
    Mphf mm;          // mphf
    BlockedEF buckets; // occs_prefix_sum
    PackedVec pos;    // pos
    bool has_skew = false;
    SkewTable skew;   // skew index: canonical k-mer of a heavy bucket -> global k-mer position (SkewView in mz_common.hpp)
    FrontTable front; // minimizer lines (optional)
    uint64_t n_minimizers = 0, n_minrec_overflow = 0;
    double build_seconds = 0;
    // false after a GPU build that left its arrays in HBM (mm.blocks, buckets, pos.w, skew and front vectors are then empty
    // until somebody needs them on the host: save(), a replica on another GPU); the scalars and `us` are always here
    bool arrays_on_host = true;
    // sizes of the arrays above in words, from the scalars alone
    uint64_t n_mm_block_words() const { // (from the scalars: the arrays may be in HBM only)
        if (mm.direct) return (((1ull << (2 * w)) >> kMinRecKeysLog2) + 1) * kMinRecWords;
        return mm.levels.empty() ? 8 : (mm.levels.back().block_off + mm.levels.back().n_blocks) * 8;
    }
    uint64_t n_bucket_block_words() const { return (((buckets.n - 1) >> kBefShift) + 1) * 8; }
    uint64_t n_bucket_ovf_words() const { return buckets.n_overflow ? buckets.n_overflow * (kBefSpan + 1) : 2; }

    int build(const UnitigSet &unitigs, uint32_t w_, uint64_t skew_param_, uint64_t seed_, int n_threads, const BuildOpts &opts = BuildOpts());
    IndexView view() const;
    uint64_t num_bits() const;
    int save(const std::string &path) const;
    int load(const std::string &path);
};

// fingerprint width the builders give a new skew table: kSkewFpBits, or MZ_TEST_SKEW_FP_BITS (1..kSkewFpBits; a knob for
// tests that need fingerprint collisions to be common)
uint32_t skew_fp_bits_for_build();
// minimizer MPHF kind: direct-address records while 2w <= 32 and they stay small next to the key set: at most
// max(64 MiB, B bytes per key), B = 48 unless opts.direct_bytes_per_key says otherwise (a transcriptome-sized set at w = 15 then
// takes the records: one gather instead of the level loop, 1.6x the index, 1.9x the point-query rate); else BBHash levels
bool use_direct_records(uint32_t w, uint64_t n_mm, const BuildOpts &opts);

void set_error(const char *fmt, ...);
const char *last_error();

} // namespace mz
