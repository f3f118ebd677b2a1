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
        uint64_t fails = 0, n = 0;
        for (uint64_t ui = 0; ui < us.n_unitigs; ++ui) {
            uint64_t s = us.starts[ui], e = us.starts[ui + 1];
            if (us.unitig_len(ui) != e - s || us.unitig_start_pos(ui) != s) ++fails;
            for (uint64_t p = s; p + k <= e; ++p, ++n) {
                uint64_t fw = read_kmer_i(v.seqbv, p, k);
                mz_k2upos r = lookup(v, fw);
                if (!(r.match == MZ_IDENTITY_MATCH && r.unitig_id == ui && r.pos == p - s && r.unitig_len == e - s)) ++fails;
                r = lookup(v, kmer_rc(fw, k));
                if (!(r.match == MZ_TWIN_MATCH && r.unitig_id == ui && r.pos == p - s && r.unitig_len == e - s)) ++fails;
            }
        }
        std::printf("{\"n_kmers\": %llu, \"n_fail\": %llu, \"n_minimizers\": %llu, \"n_occs\": %llu, \"n_skew\": %llu, \"mphf_levels\": %zu, "
                    "\"ef_overflow_blocks\": %llu, \"direct\": %u, \"minrec_overflow\": %llu, \"num_bits\": %llu, \"build_s\": %.3f}\n",
                    (unsigned long long)n, (unsigned long long)fails, (unsigned long long)h.n_minimizers, (unsigned long long)h.pos.n,
                    (unsigned long long)(h.has_skew ? h.skew.n_keys : 0), h.mm.levels.size(), (unsigned long long)h.buckets.n_overflow, h.mm.direct, (unsigned long long)h.n_minrec_overflow,
                    (unsigned long long)h.num_bits(), h.build_seconds);
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
