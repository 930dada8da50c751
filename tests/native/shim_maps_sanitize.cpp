// Sanitizer harness for the two containers of the SfM mirror (spherical_sfm_amd/csrc/shim/sfm.h: IndexedMap = dense storage, FlatMap = sorted vector) against the
// std::map whose interface they implement: random operator[] / find / count / erase / iteration, in ascending and in random key order, under -fsanitize=address,undefined.
// Built and run by tests/test_sanitizers_cpu.py.
#include <cstdio>
#include <map>
#include <random>
#include "../../spherical_sfm_amd/csrc/shim/sfm.h"
#ifdef SSFM_REF_SPARSE      // build container only: the reference's own containers (include/sphericalsfm/sparse.hpp, std-only, compiled as it stands) are the yardstick
#include <sphericalsfm/sparse.hpp>
#endif
using namespace sphericalsfm;

template <class M> static void compare(M& m, std::map<int, double>& ref, const char* what) {
    if (m.size() != ref.size() || m.empty() != ref.empty()) { std::printf("%s: size %zu vs %zu\n", what, m.size(), ref.size()); std::abort(); }
    auto r = ref.begin();
    for (auto&& kv : m) { if (r == ref.end() || kv.first != r->first || kv.second != r->second) { std::printf("%s: iteration differs at key %d\n", what, kv.first); std::abort(); } ++r; }
    if (r != ref.end()) { std::printf("%s: iteration ended early\n", what); std::abort(); }
    const M& cm = m;
    for (const auto& kv : cm) if (ref.at(kv.first) != kv.second) { std::printf("%s: const iteration differs\n", what); std::abort(); }
}
template <class M> static long run(const char* what, unsigned seed, bool ascending_first) {
    std::mt19937 rng(seed); M m; std::map<int, double> ref; long sum = 0;
    if (ascending_first) for (int k = 0; k < 300; k += 1 + (int)(rng() % 3)) { const double v = (double)(rng() % 1000); m[k] = v; ref[k] = v; }      // the common case: ids in order
    for (int step = 0; step < 20000; step++) {
        const int k = (int)(rng() % 400), op = (int)(rng() % 6);
        if (op == 0 || op == 1) { const double v = (double)(rng() % 1000); m[k] = v; ref[k] = v; }
        else if (op == 2) { if (m.erase(k) != ref.erase(k)) { std::printf("%s: erase(%d) differs\n", what, k); std::abort(); } }
        else if (op == 3) { auto it = m.find(k); auto rt = ref.find(k); if ((it == m.end()) != (rt == ref.end()) || (rt != ref.end() && (it->first != k || it->second != rt->second))) { std::printf("%s: find(%d) differs\n", what, k); std::abort(); } if (it != m.end()) { it->second += 1.0; rt->second += 1.0; } }
        else if (op == 4) { if (m.count(k) != ref.count(k)) { std::printf("%s: count(%d) differs\n", what, k); std::abort(); } }
        else { const double a = m[k]; const double b = ref[k]; if (a != b) { std::printf("%s: operator[](%d) differs\n", what, k); std::abort(); } sum += (long)a; }      // (inserts a default value like std::map)
        if (step % 997 == 0) compare(m, ref, what);
    }
    compare(m, ref, what);
    return sum;
}
#ifdef SSFM_REF_SPARSE
// The mirror's tables against the REFERENCE's SparseVector / SparseMatrix on the operations SfM uses (src/sfm.cpp: operator(), exists, erase, ordered iteration):
// points = IndexedMap<double> vs SparseVector<double>; observations = std::map<int, FlatMap<double>> vs SparseMatrix<double>.
static long run_reference(unsigned seed) {
    std::mt19937 rng(seed); long sum = 0;
    IndexedMap<double> pts; SparseVector<double> rpts;
    std::map<int, FlatMap<double>> obs; SparseMatrix<double> robs;
    auto same_rows = [&]() {
        auto r = robs.begin();
        for (auto& row : obs) {
            if (row.second.empty()) continue;                                  // (an emptied row may stay behind in either table: SfM only walks entries)
            while (r != robs.end() && r->second.empty()) ++r;
            if (r == robs.end() || r->first != row.first || r->second.size() != row.second.size()) { std::printf("reference: row %d differs\n", row.first); std::abort(); }
            auto c = r->second.begin();
            for (auto&& kv : row.second) { if (kv.first != c->first || kv.second != c->second) { std::printf("reference: entry (%d, %d) differs\n", row.first, kv.first); std::abort(); } ++c; }
            ++r;
        }
        while (r != robs.end() && r->second.empty()) ++r;
        if (r != robs.end()) { std::printf("reference: rows missing\n"); std::abort(); }
        auto q = rpts.begin();
        for (auto&& kv : pts) { if (q == rpts.end() || kv.first != q->first || kv.second != q->second) { std::printf("reference: point %d differs\n", kv.first); std::abort(); } ++q; }
        if (q != rpts.end()) { std::printf("reference: points missing\n"); std::abort(); }
    };
    for (int step = 0; step < 30000; step++) {
        const int r = (int)(rng() % 40), c = (int)(rng() % 300), op = (int)(rng() % 8);
        const double v = (double)(rng() % 1000);
        if (op <= 1) { obs[r][c] = v; robs(r, c) = v; }                         // AddObservation
        else if (op == 2) { const bool a = obs.count(r) && obs[r].count(c), b = robs.exists(r, c); if (a != b) { std::printf("reference: exists(%d, %d)\n", r, c); std::abort(); } if (a && obs[r][c] != robs(r, c)) std::abort(); }
        else if (op == 3) { if (obs.count(r)) obs[r].erase(c); robs.erase(r, c); }   // RemoveObservation
        else if (op == 4) { pts[c] = v; rpts(c) = v; }                          // AddPoint / SetPoint
        else if (op == 5) { if ((pts.count(c) != 0) != rpts.exists(c)) { std::printf("reference: point exists(%d)\n", c); std::abort(); } if (rpts.exists(c)) sum += (long)(pts[c] - rpts(c)); }
        else if (op == 6) { pts.erase(c); rpts.erase(c); }                      // RemovePoint
        else { obs.erase(r); robs.erase(r); }                                   // RemoveCamera
        if (step % 1499 == 0) same_rows();
    }
    same_rows();
    return sum;
}
#endif
int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    long s = 0;
#ifdef SSFM_REF_SPARSE
    for (unsigned seed = 1; seed <= 4; seed++) s += run_reference(seed);
    std::printf("SHIM_VS_REFERENCE_SPARSE_OK\n");
#endif
    for (unsigned seed = 1; seed <= 6; seed++) { s += run<IndexedMap<double>>("IndexedMap", seed, seed & 1); s += run<FlatMap<double>>("FlatMap", 100 + seed, seed & 1); }
    // erase through an iterator while walking a row (FilterObservations)
    FlatMap<double> f; std::map<int, double> r;
    for (int k = 0; k < 50; k++) { f[2 * k] = k; r[2 * k] = k; }
    for (auto it = f.begin(); it != f.end();) { if (it->first % 3 == 0) it = f.erase(it); else ++it; }
    for (auto it = r.begin(); it != r.end();) { if (it->first % 3 == 0) it = r.erase(it); else ++it; }
    compare(f, r, "FlatMap erase(iterator)");
    std::printf("SHIM_MAPS_OK %ld\n", s);
    return 0;
}
