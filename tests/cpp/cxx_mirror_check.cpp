// cxx_mirror_check.cpp -- exercises include/mazu_amd.hpp (the C++ mirror of the reference's kphf surface).
// Reads like the reference's own tests (src/kphf/sshash.rs:645-884).  With no GPU it checks the host side and
// that queries refuse to run (exit 0, prints "nogpu"); with a GPU it checks the tiny k=7 known answers.
#include <cstdio>
#include <string>

#include "../../include/mazu_amd.hpp"

using namespace mazu;

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv) {
    std::string prefix = argc > 1 ? argv[1] : "tests/golden/cf/tiny/tiny";
    UnitigSet unitigs = UnitigSet::from_cf_prefix(prefix);
    REQUIRE(unitigs.k() == 7 && unitigs.n_unitigs() == 2 && unitigs.total_len() == 20);
    REQUIRE(unitigs.unitig_seq(0) == "CACACACCAC" && unitigs.unitig_seq(1) == "CCTCAATACG");
    bool threw = false;
    try { SSHash::from_unitig_set(unitigs, 8, 0); } catch (const Error &e) { threw = e.code == MZ_EINVAL; } // assert!(w <= k)
    REQUIRE(threw);
    if (mz_device_count() < 1) {
        SSHash sshash = SSHash::from_unitig_set_no_skew_index(unitigs, 3);
        threw = false;
        try { sshash.k2u(CanonicalKmer("CACACAC")); } catch (const Error &e) { threw = e.code == MZ_EHIP; }
        REQUIRE(threw); // no CPU fallback
        std::puts("nogpu");
        return 0;
    }
    for (size_t w = 1; w <= 7; ++w) {
        SSHash sshash = SSHash::from_unitig_set_no_skew_index(unitigs, w);
        struct { const char *km; size_t uid, pos; } kats[] = {{"CACACAC", 0, 0}, {"ACACACC", 0, 1}, {"ACACCAC", 0, 3}, {"CCTCAAT", 1, 0}, {"CAATACG", 1, 3}};
        for (auto &t : kats) {
            CanonicalKmer km(t.km);
            K2UPos ans{t.uid, 10, t.pos, MatchType::IdentityMatch};
            REQUIRE(sshash.k2u(km) == ans);
            km.swap();
            ans.reverse_match_type();
            REQUIRE(sshash.k2u(km) == ans);
        }
        CanonicalKmer miss("AAAAAAA");
        REQUIRE(!sshash.k2u(miss).has_value());
        sshash.validate_self();
        threw = false;
        try { sshash.k2u(CanonicalKmer(0, 8)); } catch (const Error &e) { threw = e.code == MZ_EINVAL; } // sshash.rs:826-838
        REQUIRE(threw);
    }
    SSHash all_skew = SSHash::from_unitig_set(unitigs, 3, 0);
    REQUIRE(all_skew.n_kmers_in_skew_index() == all_skew.n_kmers());
    StreamingK2U st = all_skew.as_streaming();
    REQUIRE(st.k2u_streaming(CanonicalKmer("CACACAC")).has_value());
    {   // construction with layout choices: built by the GPU builder, minimizer lines forced, the workgroup-scoped point kernel; same answers
        mz_build_opts bo{};
        bo.device = 0;
        bo.minimizer_lines = 1;
        SSHash opt = SSHash::from_unitig_set_opts(unitigs, 3, 0, 0, bo);
        REQUIRE(opt.built_on_device() && opt.has_minimizer_lines());
        mz_query_opts qo{};
        qo.point_kernel = 2;
        opt.set_query_opts(qo);
        REQUIRE(opt.k2u(CanonicalKmer("CACACAC")).has_value() && !opt.k2u(CanonicalKmer("AAAAAAA")).has_value());
        opt.validate_self();
    }
    std::puts("gpu-ok");
    return 0;
}
