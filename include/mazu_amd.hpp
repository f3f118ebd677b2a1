// mazu_amd.hpp -- header-only C++17 mirror of mazu's `kphf` surface for the SSHash path, over the C ABI
// of mazu_amd.h.  Same names, argument meaning and error behaviour as the reference (src/kphf/mod.rs,
// src/kphf/sshash.rs, src/unitig_set.rs, src/index/caching.rs): where the reference panics
// (`assert_eq!` on k, `assert!(w <= k)`) or returns `Err`, these throw mazu::Error.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mazu_amd.h"

namespace mazu {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("mazu_amd error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
    if (rc != MZ_OK) throw Error(rc, mz_last_error());
}

enum class MatchType : uint32_t { NoMatch = MZ_NO_MATCH, IdentityMatch = MZ_IDENTITY_MATCH, TwinMatch = MZ_TWIN_MATCH }; // kmers crate

struct K2UPos { // kphf/mod.rs:13-19
    size_t unitig_id, unitig_len, pos;
    MatchType o;
    bool operator==(const K2UPos &r) const { return unitig_id == r.unitig_id && unitig_len == r.unitig_len && pos == r.pos && o == r.o; }
    void reverse_match_type() { // kphf/mod.rs:22-29
        o = o == MatchType::IdentityMatch ? MatchType::TwinMatch : (o == MatchType::TwinMatch ? MatchType::IdentityMatch : o);
    }
};

// kmers::CanonicalKmer as used on the path: forward word + k; swap() exchanges fw and rc
class CanonicalKmer {
    uint64_t fw_ = 0, rc_ = 0;
    uint32_t k_ = 0;
    static uint64_t rc_of(uint64_t w, uint32_t k) {
        uint64_t r = 0;
        for (uint32_t i = 0; i < k; ++i) r |= (3 - ((w >> (2 * i)) & 3)) << (2 * (k - 1 - i));
        return r;
    }
public:
    CanonicalKmer(uint64_t fw_word, uint32_t k) : fw_(fw_word), rc_(rc_of(fw_word, k)), k_(k) {}
    explicit CanonicalKmer(const std::string &s) : k_((uint32_t)s.size()) {
        for (uint32_t i = 0; i < k_; ++i) {
            char c = s[i] & 0xDF;
            uint64_t code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
            if (code > 3) throw Error(MZ_EINVAL, "non-ACGT base in k-mer");
            fw_ |= code << (2 * i);
        }
        rc_ = rc_of(fw_, k_);
    }
    void swap() { std::swap(fw_, rc_); }
    uint64_t get_fw_mer() const { return fw_; }
    uint64_t get_canonical_word() const { return fw_ < rc_ ? fw_ : rc_; }
    uint32_t len() const { return k_; }
};

class UnitigSet { // unitig_set.rs:30-36
    mz_unitig_set *h_ = nullptr;
    bool owned_ = true;
    friend class SSHash;
    UnitigSet(mz_unitig_set *h, bool owned) : h_(h), owned_(owned) {}
public:
    static UnitigSet from_seqs(const std::vector<std::string> &seqs, uint32_t k) { // :74-106
        std::vector<const char *> p;
        for (auto &s : seqs) p.push_back(s.c_str());
        mz_unitig_set *h = nullptr;
        check(mz_unitig_set_from_seqs(p.data(), p.size(), k, &h));
        return UnitigSet(h, true);
    }
    static UnitigSet from_cf_prefix(const std::string &prefix) { // :109-165
        mz_unitig_set *h = nullptr;
        check(mz_unitig_set_from_cf_prefix(prefix.c_str(), &h));
        return UnitigSet(h, true);
    }
    UnitigSet(UnitigSet &&o) noexcept : h_(o.h_), owned_(o.owned_) { o.h_ = nullptr; }
    UnitigSet(const UnitigSet &) = delete;
    ~UnitigSet() { if (owned_ && h_) mz_unitig_set_free(h_); }
    size_t k() const { return mz_unitig_set_k(h_); }
    size_t n_unitigs() const { return mz_unitig_set_n_unitigs(h_); }
    size_t total_len() const { return mz_unitig_set_total_len(h_); }
    size_t n_kmers() const { return mz_unitig_set_n_kmers(h_); }
    size_t unitig_len(size_t i) const { return mz_unitig_set_unitig_len(h_, i); }
    size_t unitig_start_pos(size_t i) const { return mz_unitig_set_unitig_start_pos(h_, i); }
    size_t pos_to_id(size_t pos) const { return mz_unitig_set_pos_to_id(h_, pos); }
    std::string unitig_seq(size_t i) const { // SeqVectorSlice::to_string
        const uint64_t *w = mz_unitig_set_useq_words(h_);
        size_t s = unitig_start_pos(i), n = unitig_len(i);
        std::string out(n, 'A');
        for (size_t j = 0; j < n; ++j) out[j] = "ACGT"[(w[(s + j) >> 5] >> (2 * ((s + j) & 31))) & 3];
        return out;
    }
};

class StreamingK2U;

class SSHash { // sshash.rs:39-51; trait K2U kphf/mod.rs:58-140
    mz_sshash *h_ = nullptr;
    friend class StreamingK2U;
    explicit SSHash(mz_sshash *h) : h_(h) {}
public:
    static SSHash from_unitig_set(const UnitigSet &unitigs, size_t w, size_t skew_param, uint64_t seed = 0) { // :405-412
        mz_sshash *h = nullptr;
        check(mz_sshash_from_unitig_set(unitigs.h_, (uint32_t)w, skew_param == SIZE_MAX ? MZ_NO_SKEW_INDEX : skew_param, seed, 0, &h));
        return SSHash(h);
    }
    // the same construction with this library's layout choices (mz_build_opts: where to build, minimizer lines, record budget); none changes a result
    static SSHash from_unitig_set_opts(const UnitigSet &unitigs, size_t w, size_t skew_param, uint64_t seed, mz_build_opts opts) {
        mz_sshash *h = nullptr;
        opts.struct_size = (uint32_t)sizeof(mz_build_opts);
        check(mz_sshash_from_unitig_set_opts(unitigs.h_, (uint32_t)w, skew_param == SIZE_MAX ? MZ_NO_SKEW_INDEX : skew_param, seed, &opts, &h));
        return SSHash(h);
    }
    bool built_on_device() const { return mz_sshash_built_on_device(h_) != 0; }
    bool has_minimizer_lines() const { return mz_sshash_has_minimizer_lines(h_) != 0; }
    void set_query_opts(mz_query_opts opts) { // kernel choices of this handle's query entry points
        opts.struct_size = (uint32_t)sizeof(mz_query_opts);
        check(mz_sshash_set_query_opts(h_, &opts));
    }
    static SSHash from_unitig_set_no_skew_index(const UnitigSet &unitigs, size_t w, uint64_t seed = 0) { // :397-403
        return from_unitig_set(unitigs, w, SIZE_MAX, seed);
    }
    SSHash(SSHash &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    SSHash(const SSHash &) = delete;
    ~SSHash() { if (h_) mz_sshash_free(h_); }
    size_t k() const { return mz_sshash_k(h_); }
    size_t n_kmers() const { return mz_sshash_n_kmers(h_); }
    size_t sum_unitigs_len() const { return mz_sshash_sum_unitigs_len(h_); }
    size_t unitig_len(size_t id) const { return mz_sshash_unitig_len(h_, id); }
    size_t n_unitigs() const { return mz_sshash_n_unitigs(h_); }
    UnitigSet as_ref() const { return UnitigSet(const_cast<mz_unitig_set *>(mz_sshash_as_unitig_set(h_)), false); } // AsRef<UnitigSet>
    size_t n_minimizers() const { return mz_sshash_n_minimizers(h_); }
    size_t n_kmers_in_skew_index() const { return mz_sshash_n_kmers_in_skew_index(h_); }
    size_t num_bits() const { return mz_sshash_num_bits(h_); }
    static std::optional<K2UPos> to_opt(const mz_k2upos &r) {
        if (r.match != MZ_IDENTITY_MATCH && r.match != MZ_TWIN_MATCH) return std::nullopt;
        return K2UPos{r.unitig_id, r.unitig_len, r.pos, (MatchType)r.match};
    }
    std::optional<K2UPos> k2u(const CanonicalKmer &km) const { // :471-554; wrong k throws (the reference panics)
        mz_k2upos r{};
        check(mz_sshash_k2u(h_, km.get_fw_mer(), km.len(), &r));
        return to_opt(r);
    }
    std::vector<mz_k2upos> k2u_batch(const std::vector<uint64_t> &fw_words, mz_counts *counts = nullptr) const {
        std::vector<mz_k2upos> out(fw_words.size());
        check(mz_sshash_k2u_batch(h_, fw_words.data(), fw_words.size(), out.data(), counts));
        return out;
    }
    void validate_self() const { // kphf/mod.rs:69-103, on the device
        mz_validate_report rep{};
        check(mz_sshash_validate_self(h_, &rep));
        if (rep.n_fail != 0 || rep.n_kmers != n_kmers())
            throw Error(MZ_EINVAL, "validate_self failed at global position " + std::to_string(rep.first_fail_pos));
    }
    void validate_self_parallel() const { validate_self(); } // kphf/mod.rs:105-139
    inline StreamingK2U as_streaming() const;                // :435-437
};

class StreamingK2U { // index/caching.rs:12-17
    mz_streaming *h_ = nullptr;
    size_t k_;
public:
    explicit StreamingK2U(const SSHash &k2u) : k_(k2u.k()) { check(mz_streaming_new(k2u.h_, &h_)); }
    StreamingK2U(StreamingK2U &&o) noexcept : h_(o.h_), k_(o.k_) { o.h_ = nullptr; }
    ~StreamingK2U() { if (h_) mz_streaming_free(h_); }
    size_t k() const { return k_; }
    std::optional<K2UPos> k2u_streaming(const CanonicalKmer &km) { // :65-71
        mz_k2upos r{};
        check(mz_streaming_k2u(h_, km.get_fw_mer(), km.len(), &r));
        return SSHash::to_opt(r);
    }
};
inline StreamingK2U SSHash::as_streaming() const { return StreamingK2U(*this); }

} // namespace mazu
