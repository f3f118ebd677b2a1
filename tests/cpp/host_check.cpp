// host_check.cpp -- TEST-ONLY debugging aid (never linked into libmazu_amd.so).
// Compiles the shared lookup chain of mazu_amd/csrc/mz_common.hpp for the HOST so the index
// BUILDER (mz_host.cpp) can be checked in a container without a GPU.  The product library
// exports no host query entry point; GPU parity is established by the `-m gpu` tests.
//
//   host_check validate <cf_prefix> <w> <skew_param|none> <seed>
//   host_check batch    <cf_prefix> <w> <skew_param|none> <seed> <queries.u64> <out.k2upos>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../mazu_amd/csrc/mz_host.hpp"

using namespace mz;

// The same lookup answered through the minimizer lines, as stage SF and front_walk of the kernels read them (a plain restatement for
// the host: line and key field of the query's minimizer -- the word itself at 2w <= 32, a hash of it beyond --, the key's slots in bucket
// order, flank test, verification against the sequence).  A marker (bucket not inline) or a line the builder gave up defers to lookup().
// *via: 0 = answered from the line, 1 = deferred.
static mz_k2upos lookup_through_lines(const IndexView &ix, uint64_t fw, int *via) {
    const uint32_t k = ix.k, w = ix.w;
    const uint64_t rc = kmer_rc(fw, k), g = fw <= rc ? fw : rc;
    const Minimizer mn = canonical_minimizer(fw, rc, k, w, ix.seed);
    uint64_t line;
    uint32_t kf;
    if (front_is_hashed(w)) {
        const uint64_t h = front_hash(mn.word);
        line = front_hashed_line(h, ix.front.n_lines);
        kf = front_hashed_keyfield(h);
    } else {
        line = front_line_of((uint32_t)mn.word, ix.front.g, ix.front.n_lines);
        kf = front_keyfield((uint32_t)mn.word, ix.front.g, 2 * w - ix.front.t);
    }
    const uint64_t *L = ix.front.lines + line * kFrontSlots;
    *via = 1;
    if (front_is_hashed(w) && L[0] == kFrontUnusable) return lookup(ix, fw);
    for (uint32_t j = 0; j < kFrontSlots; ++j) { // a marker of this key: the bucket lives elsewhere
        const uint32_t kind = (uint32_t)L[j] & 3u, key = ((uint32_t)L[j] & kFrontKeyMask) >> kFrontKeyShift;
        if ((kind == kFrontHeavy || kind == kFrontSpill) && key == kf) return lookup(ix, fw);
    }
    *via = 0;
    for (uint32_t j = 0; j < kFrontSlots; ++j) {
        const uint64_t e = L[j];
        if (((uint32_t)e & 3u) != kFrontEntry || (((uint32_t)e & kFrontKeyMask) >> kFrontKeyShift) != kf) continue;
        if (!flank_ok(g, mn.offset, k, w, (uint32_t)(e >> kFrontFlankShift) & 0xFFFFu)) continue;
        const uint32_t cls = (uint32_t)(e >> kFrontClsShift) & 1u;
        const uint64_t p = (e >> kFrontPosShift) - (cls ? k - w - mn.offset : mn.offset);
        if (p > ix.total_len - k) continue;
        const uint32_t mt = word_equivalency(fw, rc, read_kmer_i(ix.seqbv, p, k));
        if (mt == MZ_NO_MATCH) continue;
        const UnitigSpan sp = locate_unitig_i(ix.seqbv, p);
        if (p + k <= sp.end) return make_hit(sp, p, mt);
    }
    return make_none();
}
static bool same(const mz_k2upos &a, const mz_k2upos &b) { return a.match == b.match && (a.match == MZ_NO_MATCH || (a.unitig_id == b.unitig_id && a.pos == b.pos && a.unitig_len == b.unitig_len)); }

static uint64_t parse_skew(const char *s) { return std::strcmp(s, "none") == 0 ? ~0ull : std::strtoull(s, nullptr, 10); }

int main(int argc, char **argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    std::string mode = argv[1];
    UnitigSet us;
    if (us.from_cf_prefix(argv[2]) != MZ_OK) { std::fprintf(stderr, "load: %s\n", last_error()); return 1; }
    SSHashHost h;
    int rc = h.build(us, (uint32_t)std::atoi(argv[3]), parse_skew(argv[4]), std::strtoull(argv[5], nullptr, 10), 0);
    if (rc != MZ_OK) { std::fprintf(stderr, "build: rc=%d %s\n", rc, last_error()); return 1; }
    IndexView v = h.view();
    const uint32_t k = us.k;
    if (mode == "validate") { // kphf/mod.rs:69-103 evaluated with the shared chain on the host
        uint64_t fails = 0, n = 0, line_fails = 0, from_line = 0;
        const uint64_t line_stride = h.pos.n / (h.n_minimizers ? h.n_minimizers : 1) > 64 ? 32 : 1; // (buckets of thousands of entries and no skew index: every lookup walks one; a sample will do)
        for (uint64_t ui = 0; ui < us.n_unitigs; ++ui) {
            uint64_t s = us.starts[ui], e = us.starts[ui + 1];
            if (us.unitig_len(ui) != e - s || us.unitig_start_pos(ui) != s) ++fails;
            for (uint64_t p = s; p + k <= e; ++p, ++n) {
                uint64_t fw = read_kmer_i(v.seqbv, p, k);
                mz_k2upos r = lookup(v, fw);
                if (!(r.match == MZ_IDENTITY_MATCH && r.unitig_id == ui && r.pos == p - s && r.unitig_len == e - s)) ++fails;
                r = lookup(v, kmer_rc(fw, k));
                if (!(r.match == MZ_TWIN_MATCH && r.unitig_id == ui && r.pos == p - s && r.unitig_len == e - s)) ++fails;
                if (v.front.lines && (p % line_stride) == 0) { // the minimizer lines must answer what the chain answers: this k-mer, its reverse complement, and a neighbour with one base changed
                    const uint64_t qs[3] = {fw, kmer_rc(fw, k), fw ^ (1ull << (2 * (p % k)))};
                    for (uint64_t q : qs) {
                        int via = 0;
                        if (!same(lookup_through_lines(v, q, &via), lookup(v, q))) ++line_fails;
                        from_line += via == 0;
                    }
                }
            }
        }
        fails += line_fails;
        std::printf("{\"n_kmers\": %llu, \"n_fail\": %llu, \"n_minimizers\": %llu, \"n_occs\": %llu, \"n_skew\": %llu, \"mphf_levels\": %zu, "
                    "\"ef_overflow_blocks\": %llu, \"direct\": %u, \"minrec_overflow\": %llu, \"num_bits\": %llu, \"build_s\": %.3f, \"lines\": %u, \"line_fail\": %llu, "
                    "\"answered_from_line\": %llu, \"lines_unusable\": %llu, \"walk_markers\": %llu}\n",
                    (unsigned long long)n, (unsigned long long)fails, (unsigned long long)h.n_minimizers, (unsigned long long)h.pos.n,
                    (unsigned long long)(h.has_skew ? h.skew.n_keys : 0), h.mm.levels.size(), (unsigned long long)h.buckets.n_overflow, h.mm.direct, (unsigned long long)h.n_minrec_overflow,
                    (unsigned long long)h.num_bits(), h.build_seconds, h.front.present ? 1u : 0u, (unsigned long long)line_fails, (unsigned long long)from_line,
                    (unsigned long long)h.front.n_unusable, (unsigned long long)h.front.n_walk);
        return fails ? 3 : 0;
    }
    if (mode == "batch" && argc >= 8) {
        FILE *f = std::fopen(argv[6], "rb");
        if (!f) return 1;
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        std::vector<uint64_t> q(sz / 8);
        if (std::fread(q.data(), 8, q.size(), f) != q.size()) return 1;
        std::fclose(f);
        std::vector<mz_k2upos> out(q.size());
        for (size_t i = 0; i < q.size(); ++i) out[i] = lookup(v, q[i]);
        f = std::fopen(argv[7], "wb");
        std::fwrite(out.data(), sizeof(mz_k2upos), out.size(), f);
        std::fclose(f);
        return 0;
    }
    return 2;
}
