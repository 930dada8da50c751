// Sanitizer harness for the two containers of the SfM mirror (spherical_sfm_amd/csrc/shim/sfm.h: IndexedMap = dense storage, FlatMap = sorted vector) against the
// std::map whose interface they implement: random operator[] / find / count / erase / iteration, in ascending and in random key order, under -fsanitize=address,undefined.
// Built and run by tests/test_sanitizers_cpu.py.
#include <cstdio>
#include <map>
#include <random>
#include "../../spherical_sfm_amd/csrc/shim/sfm.h"
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
int main() {
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    long s = 0;
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
